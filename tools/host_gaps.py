#!/usr/bin/env python3
"""Diagnostic: host-side critical path of a training step between the status read-back and the next forward launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, phoenix_amd
from phoenix_amd import engine
od = sys.modules["phoenix_amd.odeint"]
wl = bench.WORKLOADS["breast"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N, H, B = wl["N"], wl["H"], wl["B"]
G = (torch.randn(t.shape[1], B, 1, N) / (B * N)).to(dev)
T = {}
orig_raise, orig_fwd, orig_params = engine.raise_for_status, engine.solve_forward, engine.Params
def raise_(s):
    orig_raise(s); T["sync_ret"] = time.perf_counter()
def fwd_(*a, **k):
    T["fwd_entry"] = time.perf_counter()
    r = orig_fwd(*a, **k)
    T["fwd_ret"] = time.perf_counter()
    return r
class P_(orig_params):
    def __init__(self, *a):
        T.setdefault("params_entry", time.perf_counter())
        super().__init__(*a)
        T.setdefault("params_ret", time.perf_counter())
engine.raise_for_status = raise_; engine.solve_forward = fwd_; engine.Params = P_
from phoenix_amd import _lib
_l = _lib.load()
class _LibProxy:
    def __getattr__(self, name):
        f = getattr(_l, name)
        if name != "phx_odeint":
            return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); T["c_call"] = time.perf_counter() - t0
            return r
        return g
_lib.load = lambda: _LibProxy()
orig_bwd = od._OdeintAdjointFn.backward
def bwd_(ctx, *g):
    r = orig_bwd(ctx, *g)
    T["bwd_ret"] = time.perf_counter()
    return r
od._OdeintAdjointFn.backward = staticmethod(bwd_)
orig_one = bench.one_step
def one_(net, y0, t, G, method, world):
    for p in net.parameters():
        p.grad = None
    y = y0.detach().requires_grad_(True)
    T["pre_fwd"] = time.perf_counter()
    sol = phoenix_amd.odeint_adjoint(net, y, t, method=method)
    T["post_fwd"] = time.perf_counter()
    loss = (sol * G).sum()
    T["post_loss"] = time.perf_counter()
    loss.backward()
    T["post_bwd"] = time.perf_counter()
    return sol
bench.one_step = one_
acc = {}
for it in range(60):
    T.pop("params_entry", None); T.pop("params_ret", None)
    T["step"] = time.perf_counter()
    bench.one_step(net, y0, t, G, wl["method"], 1)
    if it >= 10 and "prev_sync" in T:
        for k, v in (("sync_ret->bwd_ret", T["prev_bwd_ret"] - T["prev_sync"]), ("bwd_ret->backward() returns", T["prev_post_bwd"] - T["prev_bwd_ret"]),
                     ("backward() returns->step_start", T["step"] - T["prev_post_bwd"]), ("grad=None, detach", T["pre_fwd"] - T["step"]),
                     ("sync_ret->step_start", T["step"] - T["prev_sync"]), ("step_start->Params", T["params_entry"] - T["step"]),
                     ("Params()", T["params_ret"] - T["params_entry"]), ("Params->solve_forward", T["fwd_entry"] - T["params_ret"]),
                     ("solve_forward (allocs+C call)", T["fwd_ret"] - T["fwd_entry"]), ("  of which phx_odeint C call", T.get("c_call", 0))):
            acc.setdefault(k, []).append(v)
    T["prev_sync"] = T["sync_ret"]; T["prev_bwd_ret"] = T["bwd_ret"]; T["prev_post_bwd"] = T["post_bwd"]
for k, v in acc.items():
    print("%-32s %7.1f us" % (k, 1e6 * sum(v) / len(v)))
