#!/usr/bin/env python3
"""Run on the GPU box: host-side duration of every bench step (no per-step synchronisation, deferred status), to see
whether single steps stall.   python tools/step_jitter.py [workload] [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                   # noqa: E402
from phoenix_amd import engine                 # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "insilico"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
T, B, N = t.shape[-1], wl["B"], wl["N"]
G = (torch.randn(T, B, 1, N) / (B * N)).to(dev)
G[0].zero_()
engine.set_status_mode("deferred")
times = []
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    bench.one_step(net, y0, t, G, wl["method"], 1)
    times.append((time.perf_counter() - t0) * 1e3)
engine.check_pending_status(wait=True)
torch.cuda.synchronize()
tot = (time.perf_counter() - t_all) * 1e3
print("host ms per step issue:", " ".join("%.2f" % x for x in times))
print("total %.2f ms for %d steps = %.3f ms/step" % (tot, steps, tot / steps))
