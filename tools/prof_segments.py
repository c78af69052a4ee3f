#!/usr/bin/env python3
"""Diagnostic: per-segment time of the v1 persistent kernels (PHX_PROF=1), averaged over workgroups.
usage: PHX_PROF=1 python tools/prof_segments.py [workload | N,H,B[,t1]] [fwd|adj] [trajectories]
PHX_PROF=2 (per-block timers inside the sweeps of the second / third-generation kernels) needs the diagnostic build of the library
(the marks are compiled out of the regular one: they cost 1.5-2.3 % of a launch): `python -m phoenix_amd.build --prof`
here, or it is built on first use; this script then loads libphoenix_prof.so."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PHX_PROF", "1")
if os.environ["PHX_PROF"] == "2" and "PHX_LIB" not in os.environ:
    from phoenix_amd import build as _build
    os.environ["PHX_DIAG"] = "1"
    os.environ["PHX_LIB"] = _build.build_prof(check_stale=False)   # build it here first: python -m phoenix_amd.build --prof
import bench  # noqa: E402
from phoenix_amd import _lib, engine  # noqa: E402

_w = sys.argv[1] if len(sys.argv) > 1 else "breast"
if "," in _w:      # a shape of one's own: N,H,B[,t1]  (dopri5, t = [0, t1])
    _p = _w.split(",")
    wl = dict(N=int(_p[0]), H=int(_p[1]), B=int(_p[2]), method="dopri5", t=[0.0, float(_p[3]) if len(_p) > 3 else 1.0], desc=_w)
else:
    wl = dict(bench.WORKLOADS[_w])
if len(sys.argv) > 3:
    wl["B"] = int(sys.argv[3])
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
N, H, B = wl["N"], wl["H"], wl["B"]
p = engine.Params(net.net_sums.linear_out.weight, net.net_sums.linear_out.bias, net.net_prods.linear_out.weight,
                  net.net_prods.linear_out.bias, net.net_alpha_combine.linear_out.weight, net.gene_multipliers)
y2 = y0.reshape(B, N).contiguous()
t64 = t.double().contiguous()
T = t64.shape[1]
lib = _lib.load()
off, nwg = C.c_size_t(0), C.c_int(0)
plan = (C.c_int * 6)()
op = _lib.OP_ODEINT if which == "fwd" else _lib.OP_ADJOINT
rc = lib.phx_debug_profile_region(op, N, H, B, T, _lib.CTRL_PER_TRAJECTORY, C.byref(off), C.byref(nwg), plan)
assert rc == 0, "no v1 plan for this shape"
print("plan NW=%d TPW=%d NB=%d G=%d TG=%d HT=%d  workgroups=%d" % (*plan, nwg.value))
_k = lib.phx_debug_adjoint_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, _lib.METHODS[wl["method"]]) if which == "adj" else 0
_third = _k == 3 or (which == "fwd" and wl["method"] == "dopri5" and H <= 48 and os.environ.get("PHX_FWD") != "v1")
if os.environ["PHX_PROF"] == "2" and not _third:
    os.environ["PHX_PROF"] = "1"   # the second / first-generation kernels know one level; the diagnostic build stays loaded
G = torch.randn(T, B, N, device=dev) / (B * N)
for rep in range(3):
    sol, st, nfe, ns = engine.solve_forward(p, y2, t64, wl["method"], _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
    if which == "adj":
        engine.solve_adjoint(p, t64, sol, G, wl["method"], _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, True)
torch.cuda.synchronize()
ws = engine._ws_cache[(dev.index, torch.cuda.current_stream().cuda_stream, op)]
raw = ws[off.value: off.value + nwg.value * 16 * 8].cpu().numpy().view(np.uint64).reshape(nwg.value, 16)
us = raw.astype(np.float64) / 100.0
names = ["other", "P1", "syncA", "reduce", "syncB", "P2", "init(weights,y0)", "norm sync", "norm gather", "controller", "accept pass", "quad: rest", "quad: mfma+loop", "quad: loads+act", "quad: stores", "chunk weight staging"]
kern = lib.phx_debug_adjoint_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, _lib.METHODS[wl["method"]]) if which == "adj" else 0
second = kern == 2
fwd3 = which == "fwd" and wl["method"] == "dopri5" and H <= 48 and os.environ.get("PHX_FWD") != "v1"
if fwd3:
    names = ["other (barriers, tails)", "sweeps: P1 only", "sweeps: fused P2+P1", "reduce-scatter", "gather hidden rows (1st tile)",
             "sweeps: P2 only", "init(weights,y0)", "-", "norms (reduce+gather)", "controller + ring", "accept pass (dense output)",
             "-", "block: tile wait", "block: consume tiles + requests", "block: P2 + k", "block: input + act + P1"]
if kern == 3:
    names = ["other (barriers, tails)", "sweeps: P1' only", "sweeps: fused P2'+P1'", "reduce-scatter (+FSAL rows)", "gather hidden rows (1st tile)",
             "sweeps: P2' only", "init(weights,y0)", "-", "norms (reduce+gather)", "controller + ring", "quadrature + accept", "-",
             "block: tile wait", "block: consume tiles + requests", "block: P2' + k", "block: input + act + P1'"]
    if os.environ["PHX_PROF"] == "3":
        names[10:16] = ["quad: other (loop, requests)", "quad: tile wait", "quad: seven stages", "quad: partial stores", "quad: accept pass", "quad: acquire"]
fk = lib.phx_debug_forward_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, _lib.METHODS[wl["method"]])
if which == "fwd" and fk == 4:      # chunked third generation (phx_mfma_fwd3c.inc)
    names = ["other (barriers, tails)", "phase F: P1 only", "phase F: fused P2+P1", "reduce-scatter", "gather hidden rows (phase F, 1st tile)",
             "phase F: P2 only", "init(weights,y0)", "chunk staging (+barriers)", "norms (reduce+gather)", "controller + ring",
             "accept pass (dense output)", "phase A: P2 of chunks 0..HC-2 (+gathers)", "phase C: P1 of chunks HC-2..0", "-", "-", "-"]
if kern == 4:                       # chunked third generation (phx_mfma_adj3c.inc)
    names = ["other (barriers, tails)", "phase F: P1' only", "phase F: fused P2'+P1'", "reduce-scatter (+FSAL rows)",
             "gather hidden rows (phase F, 1st tile)", "phase F: P2' only", "init(weights,y0)", "-", "norms (reduce+gather)",
             "controller + ring", "quadrature", "phase A: P2' of chunks 0..HC-2 (+gathers)", "phase C: P1' of chunks HC-2..0",
             "chunk staging (+barriers)", "accept pass", "-"]
if second:
    names = ["other (between blocks)", "sweep tail P1-only", "drain+flag+publish", "reduce owned rows", "gather hidden rows", "sweep tail fused", "init(weights,y0)", "-", "norms (gather+pair sync)", "controller", "accept pass", "quadrature", "block: tile wait", "block: finish (VALU)", "block: requests+P1+P2 MFMA", "-"]
clk = None
if second:
    clk = raw[:, 15].astype(np.float64).copy()
    us[:, 15] = 0
if kern in (3, 4):
    clk = raw[:, 7].astype(np.float64).copy()
    us[:, 7] = 0
tot = us.sum(1)
if clk is not None:
    print("in-kernel shader clock: %.0f MHz (median over workgroups)" % np.median(clk / tot))
print("per-workgroup total: mean %.1f us  min %.1f  max %.1f" % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    if us[:, i].max() > 0:
        print("  %-24s mean %8.1f us  min %8.1f  max %8.1f   (%4.1f%%)" % (n, us[:, i].mean(), us[:, i].min(), us[:, i].max(),
                                                                       100 * us[:, i].mean() / tot.mean()))
print("nfe/traj", int(nfe[0]), "batch evals", int(nfe.sum()) / B)
