#!/usr/bin/env python3
"""Diagnostic: saves the forward trajectories and dL/dy0 of one per-sample dopri5 solve to gpurun_out/<tag>.npz (compare two
library builds with numpy afterwards).  usage: python tools/dump_case.py tag N H B"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import phoenix_amd as pa
import test_gpu_parity as T
tag, N, H, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
p = T.rand_params(N, H, seed=7 * N + H, std=min(0.05, 2.0 / np.sqrt(N)))
net = T.make_net(pa, dev, p)
r = np.random.RandomState(2)
y0 = r.rand(B, N).astype(np.float32)
t = np.stack([np.array([0.0, 0.01]) for b in range(B)]).astype(np.float32)
G = r.randn(2, B, 1, N).astype(np.float32)
y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method="dopri5")
(sol * torch.from_numpy(G).to(dev)).sum().backward()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", tag + ".npz"), sol=sol.detach().cpu().numpy(), adj=y0t.grad.cpu().numpy())
print("saved", tag)
