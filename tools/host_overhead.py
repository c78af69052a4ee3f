#!/usr/bin/env python3
"""Diagnostic: host-side (CPU) time of each engine call vs GPU time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from phoenix_amd import _lib, engine
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N, H, B = wl["N"], wl["H"], wl["B"]
args = (net.net_sums.linear_out.weight, net.net_sums.linear_out.bias, net.net_prods.linear_out.weight,
        net.net_prods.linear_out.bias, net.net_alpha_combine.linear_out.weight, net.gene_multipliers)
y2 = y0.reshape(B, N).contiguous(); t64 = t.double().contiguous(); T = t64.shape[1]
G = torch.randn(T, B, N, device=dev) / (B * N)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = engine.Params(*args); torch.cuda.synchronize(); t1 = time.perf_counter()
    sol, st, nfe, ns = engine.solve_forward(p, y2, t64, wl["method"], _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    adj, grads, st2, nfe2, ns2 = engine.solve_adjoint(p, t64, sol, G, wl["method"], _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
    t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print("params %.3f ms | fwd host %.3f ms, +sync %.3f | adj host %.3f ms, +sync %.3f" %
          ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3))
