#!/usr/bin/env python3
"""Run on the GPU box: many identical bench steps on kept workspaces (phx_solve_opts.ws_keep: no fills, the two sets of
exchange buffers alternate), every step's trajectories and gradients compared BITWISE with the first (freshly filled) one.
A stale tag, a missed clean-up of the idle set or a store hazard shows up as a mismatch (or a hang -> status 7).
usage: python tools/soak.py [workload] [steps] [shard trajectories]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench                                   # noqa: E402
import phoenix_amd                             # noqa: E402
from phoenix_amd import engine                 # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "breast"])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if len(sys.argv) > 3:
    wl["B"] = int(sys.argv[3])
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
T, B, N = t.shape[-1], wl["B"], wl["N"]
G = (torch.randn(T, B, 1, N) / (B * N)).to(dev)
G[0].zero_()
engine.set_status_mode("deferred")
params = list(net.parameters())


def step():
    for p in params:
        p.grad = None
    y = y0.detach().requires_grad_(True)
    sol = phoenix_amd.odeint_adjoint(net, y, t, method=wl["method"])
    torch.autograd.backward(sol, G)
    return sol.detach(), y.grad, [p.grad for p in params]


engine.forget_workspaces()
ref = step()
torch.cuda.synchronize()
bad = 0
t0 = time.perf_counter()
for i in range(steps):
    got = step()
    ok = torch.equal(got[0], ref[0]) & torch.equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        ok = ok & torch.equal(a, b)
    if not ok:
        bad += 1
        print("step", i, "differs from the first step", flush=True)
        if bad > 5:
            break
engine.check_pending_status(wait=True)
torch.cuda.synchronize()
print("%s, %d trajectories: %d steps, %d mismatches, %.3f ms/step incl. comparison" %
      (sys.argv[1] if len(sys.argv) > 1 else "breast", B, steps, bad, (time.perf_counter() - t0) / steps * 1e3))
sys.exit(1 if bad else 0)
