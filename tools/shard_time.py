#!/usr/bin/env python3
"""Run on the GPU box: step time of rank 0's shard of the C4 batch for W ranks (what bench.py reports as
extra.strong_scaling_ms), for the environment it is started in.   python tools/shard_time.py [W ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                   # noqa: E402
from phoenix_amd import engine, parallel       # noqa: E402

wl = bench.WORKLOADS["breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
T, B, N = t.shape[-1], wl["B"], wl["N"]
G = (torch.randn(T, B, 1, N) / (B * N)).to(dev)
G[0].zero_()
engine.set_status_mode("deferred")
for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    lo, hi = parallel.shard_range(B, 0, W)
    ys, ts, Gs = y0[lo:hi].contiguous(), t[lo:hi].contiguous(), G[:, lo:hi].contiguous()
    for _ in range(50):
        bench.one_step(net, ys, ts, Gs, wl["method"], 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        bench.one_step(net, ys, ts, Gs, wl["method"], 1)
    engine.check_pending_status(wait=True)
    torch.cuda.synchronize()
    print("W=%d (%d trajectories): %.4f ms/step" % (W, hi - lo, (time.perf_counter() - t0) / 200 * 1e3), flush=True)
