#!/usr/bin/env python3
"""Diagnostic: cost of the prior branch (prior_only_forward on K rows + its backward) at breast scale."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, phoenix_amd
wl = bench.WORKLOADS["breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N = wl["N"]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
X = torch.rand(K, 1, N, device=dev) - 0.5
target = torch.randn(K, 1, N, device=dev) * 0.01
def S():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    for p in net.parameters(): p.grad = None
    t0 = S()
    pred = net.prior_only_forward(t, X)
    t1 = S()
    loss = torch.mean((pred - target) ** 2)
    t2 = S()
    loss.backward()
    t3 = S()
    print("K=%d prior fwd %.2f ms | loss %.2f ms | bwd %.2f ms" % (K, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
