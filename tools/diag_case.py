#!/usr/bin/env python3
"""Diagnostic: one per-sample dopri5 solve + adjoint of shape (N, H, B) against the oracle, per trajectory, for the kernel
variants named on the command line (env assignments like PHX_ADJ=v1,PHX_FWD=v1; '-' = default).
usage: python tools/diag_case.py N H B [variant ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phoenix_amd as pa  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
N, H, B = (int(v) for v in sys.argv[1:4])
variants = sys.argv[4:] or ["-"]
dev = torch.device("cuda:0")
p = T.rand_params(N, H, seed=7 * N + H, std=0.05)
net, onet = T.make_net(pa, dev, p), T.onet_of(orc, p)
r = np.random.RandomState(2)
y0 = r.rand(B, N).astype(np.float32)
t = np.stack([np.array([0.1 * b, 0.1 * b + 0.4 + 0.05 * (b % 64)]) for b in range(B)]).astype(np.float32)
G = r.randn(B, 2, N).astype(np.float32)
ref = orc.odeint_per_sample(onet, y0, t, method="dopri5")
adj_ref, gr_ref = orc.adjoint_backward_per_sample(onet, t, ref, G, method="dopri5", theta_in_norm=False)
for var in variants:
    env = dict(kv.split("=") for kv in var.split(",")) if var != "-" else {}
    os.environ.update(env)
    try:
        for rep in range(2):
            T.zero_grads(net)
            y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
            sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method="dopri5")
            Gt = torch.from_numpy(G.transpose(1, 0, 2).copy()).to(dev).reshape(2, B, 1, N)
            (sol * Gt).sum().backward()
            got = sol.detach().cpu().numpy().reshape(2, B, N).transpose(1, 0, 2)
            ga = y0t.grad.cpu().numpy().reshape(B, N)
            et = np.abs(got - ref).max(axis=(1, 2)) / np.abs(ref).max()
            ea = np.abs(ga - adj_ref.reshape(B, N)).max(axis=1) / np.abs(adj_ref).max()
            print("%-24s rep %d  traj err max %.2e  adj err max %.2e  bad trajectories (adj > 1e-4): %s" %
                  (var, rep, et.max(), ea.max(), np.nonzero(ea > 1e-4)[0].tolist()))
            bad = np.nonzero(ea > 1e-4)[0]
            if len(bad):
                b = bad[0]
                dg = np.abs(ga[b] - adj_ref.reshape(B, N)[b])
                blk = dg.reshape(-1)[: (N // 32) * 32].reshape(-1, 32).max(axis=1)
                print("   trajectory %d: gene blocks with error > 1e-4 x max: %s" % (b, np.nonzero(blk > 1e-4 * np.abs(adj_ref).max())[0].tolist()[:40]))
        # step statistics of the backward solve, per trajectory (engine entry point)
        from phoenix_amd import engine, _lib
        P = engine.params_cached(*pa.odenet.params_of(net))
        t64 = torch.from_numpy(t).to(dev).double().contiguous()
        ysv = sol.detach().reshape(2, B, N).contiguous()
        gy = torch.from_numpy(G.transpose(1, 0, 2).copy()).to(dev).contiguous()
        adj, _, st, nfe, ns = engine.solve_adjoint(P, t64, ysv, gy, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, False)
        print("   nfe   ", nfe.cpu().numpy().tolist())
        print("   nsteps", ns.cpu().numpy().tolist())
    finally:
        for k in env:
            os.environ.pop(k, None)
