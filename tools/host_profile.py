#!/usr/bin/env python3
"""Diagnostic: python-side cost of one training step's engine calls (cProfile by tottime)."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, phoenix_amd
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N, H, B = wl["N"], wl["H"], wl["B"]
G = (torch.randn(t.shape[1], B, 1, N) / (B * N)).to(dev)
for _ in range(5):
    bench.one_step(net, y0, t, G, wl["method"], 1)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    y = y0.detach().requires_grad_(True)
    sol = phoenix_amd.odeint_adjoint(net, y, t, method=wl["method"])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("forward issue: %.1f us/call" % ((t1 - t0) / K * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(K):
    y = y0.detach().requires_grad_(True)
    sol = phoenix_amd.odeint_adjoint(net, y, t, method=wl["method"])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
