#!/usr/bin/env python3
"""Where the HOST time of a bench step goes (cProfile over 300 steps, deferred status), on the GPU box.
usage: python tools/host_overhead.py [trajectories]      (breast-cancer shape; 17 = the reference's batch size)"""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from phoenix_amd import engine
wl = dict(bench.WORKLOADS["breast"]); B = int(sys.argv[1]) if len(sys.argv) > 1 else 17
wl["B"] = B
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
T = t.shape[1]
G = torch.randn(T, B, 1, wl["N"], device=dev) / (B * wl["N"])
engine.set_status_mode("deferred")
for _ in range(50):
    bench.one_step(net, y0, t, G, wl["method"], 1)
engine.check_pending_status(wait=True); torch.cuda.synchronize()
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    bench.one_step(net, y0, t, G, wl["method"], 1)
t1 = time.perf_counter()
engine.check_pending_status(wait=True); torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(reps):
    bench.one_step(net, y0, t, G, wl["method"], 1)
pr.disable()
engine.check_pending_status(wait=True); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
