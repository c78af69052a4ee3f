#!/usr/bin/env python3
"""Diagnostic: wall time of the pieces of bench.one_step (with device syncs between them)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, phoenix_amd
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N, H, B = wl["N"], wl["H"], wl["B"]
T = t.shape[1]
G = (torch.randn(T, B, 1, N) / (B * N)).to(dev)
def S():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t0 = S()
    for p in net.parameters(): p.grad = None
    y = y0.detach().requires_grad_(True)
    t1 = S()
    sol = phoenix_amd.odeint_adjoint(net, y, t, method=wl["method"])
    t2 = S()
    loss = (sol * G).sum()
    t3 = S()
    loss.backward()
    t4 = S()
    print("prep %.3f | odeint_adjoint fwd %.3f | loss %.3f | backward %.3f | total %.3f ms" %
          ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3))
for rep in range(3):
    t0 = S()
    bench.one_step(net, y0, t, G, wl["method"], 1)
    t1 = S()
    print("one_step %.3f ms" % ((t1-t0)*1e3))
