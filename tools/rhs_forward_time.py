#!/usr/bin/env python3
"""Diagnostic: time of ONE standalone RHS evaluation (ODENet.forward / prior_only_forward on a small batch: the
pass-structured k1_eval_fwd below 1024 rows, the kernel chain above) at breast scale.  usage: rhs_forward_time.py [rows...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, phoenix_amd
from phoenix_amd import engine
wl = bench.WORKLOADS["breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
P = engine.params_cached(*phoenix_amd.odenet.params_of(net))
for B in [int(a) for a in sys.argv[1:]] or [16, 64, 256, 512, 1000, 1024, 4096]:
    y = torch.rand(B, wl["N"], device=dev)
    for _ in range(5): engine.rhs_forward(P, y, False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): engine.rhs_forward(P, y, False)
    torch.cuda.synchronize()
    print("B=%5d  %.1f us per evaluation" % (B, (time.perf_counter() - t0) / n * 1e6))
