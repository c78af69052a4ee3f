#!/usr/bin/env python3
"""Eager step against the same step recorded into ONE HIP graph (phoenix_amd.GraphedStep), on the GPU box.
usage: python tools/graph_step.py [workload] [trajectories] [--training-step [prior rows]]
  default: the bench step (forward solve + backward solve + re-layout + gradient reduction) of `workload` at `trajectories`
  --training-step: the reference's whole training_step (both losses, K-row prior branch, Adam with capturable=True)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import phoenix_amd  # noqa: E402
from phoenix_amd import engine  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = dict(bench.WORKLOADS[args[0] if args else "breast"])
if len(args) > 1:
    wl["B"] = int(args[1])
full = "--training-step" in sys.argv
K = int(args[2]) if full and len(args) > 2 else 10000
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
B, N, T = wl["B"], wl["N"], t.shape[1]
if full:
    class Handler:
        def __init__(self, b, tt, y):
            self.b, self.device = (b, tt, y), b.device

        def get_batch(self, bs):
            return self.b
    g = torch.Generator(device="cpu").manual_seed(7)
    h = Handler(y0, t, y0 + 0.01 * torch.randn(y0.shape, generator=g).to(dev))
    X = (torch.rand(K, 1, N, generator=g) - 0.5).to(dev)
    pg = (torch.randn(K, 1, N, generator=g) * 0.05).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-6, capturable=True)

    def step():
        return phoenix_amd.training_step(net, h, opt, wl["method"], B, False, False, X, pg, 0.99)
else:
    G = torch.randn(T, B, 1, N, device=dev) / (B * N)

    def step():
        return bench.one_step(net, y0, t, G, wl["method"], 1)

engine.set_status_mode("deferred")
for _ in range(20):
    step()
engine.check_pending_status(wait=True)
torch.cuda.synchronize()
reps = 200 if not full else 40
t0 = time.perf_counter()
for _ in range(reps):
    step()
t1 = time.perf_counter()
engine.check_pending_status(wait=True)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager, back to back (deferred status): %.4f ms/step (the host issued them in %.4f ms/step)" %
      ((t2 - t0) / reps * 1e3, (t1 - t0) / reps * 1e3), flush=True)
engine.set_status_mode("immediate")
gs = phoenix_amd.GraphedStep(step)
for _ in range(10):
    gs()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    gs()
torch.cuda.synchronize()
print("one graph per step: %.4f ms/step" % ((time.perf_counter() - t0) / reps * 1e3), flush=True)
gs.check_status()
