#!/usr/bin/env python3
"""Diagnostic: bench.one_step throughput with and without the per-step status read-back (host sync)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, phoenix_amd
from phoenix_amd import engine
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N, H, B = wl["N"], wl["H"], wl["B"]
T = t.shape[1]
G = (torch.randn(T, B, 1, N) / (B * N)).to(dev)
def run(K, label):
    for _ in range(5):
        bench.one_step(net, y0, t, G, wl["method"], 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        bench.one_step(net, y0, t, G, wl["method"], 1)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s %.3f ms/step (host issue %.3f ms/step)" % (label, dt / K * 1e3, t_host / K * 1e3))
run(50, "default (status read-back)")
orig = engine.raise_for_status
engine.raise_for_status = lambda s: None
run(50, "no status read-back")
engine.raise_for_status = orig
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    bench.one_step(net, y0, t, G, wl["method"], 1)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
