#!/usr/bin/env python3
"""Diagnostic (GPU box): the third backward kernel (PHX_ADJ=v3) against the first one (PHX_ADJ=v1) and the CPU oracle
on a few shapes -- single- and multi-step intervals, rejected steps, ragged batches, several intervals --, then the
launch times of both at the breast-cancer shape.  usage: python tools/adj3_check.py [quick]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phoenix_amd as pa  # noqa: E402
from phoenix_amd import _lib, engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = torch.device("cuda:0")


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rand_params(N, H, seed, std):
    r = np.random.RandomState(seed)
    g = r.rand(N).astype(np.float32)
    g[r.rand(N) < 0.1] *= -1
    return {"Ws": (r.randn(H, N) * std).astype(np.float32), "bs": r.uniform(-.2, .2, H).astype(np.float32),
            "Wp": (r.randn(H, N) * std).astype(np.float32), "bp": r.uniform(-.2, .2, H).astype(np.float32),
            "Wa": (r.randn(N, 2 * H) * std).astype(np.float32), "g": g}


def make_net(p):
    H, N = p["Ws"].shape
    net = pa.ODENet(dev, N, neurons=H)
    with torch.no_grad():
        net.net_sums.linear_out.weight.copy_(torch.from_numpy(p["Ws"]))
        net.net_sums.linear_out.bias.copy_(torch.from_numpy(p["bs"]))
        net.net_prods.linear_out.weight.copy_(torch.from_numpy(p["Wp"]))
        net.net_prods.linear_out.bias.copy_(torch.from_numpy(p["bp"]))
        net.net_alpha_combine.linear_out.weight.copy_(torch.from_numpy(p["Wa"]))
        net.gene_multipliers.copy_(torch.from_numpy(p["g"]).reshape(1, N))
    return net


def run(net, y0, t, G, variant):
    for k in ("PHX_ADJ", "PHX_ADJ2_NP"):
        os.environ.pop(k, None)
    if variant:
        os.environ["PHX_ADJ"] = variant
    P = engine.params_cached(*pa.odenet.params_of(net))
    sol, st, nfe, ns = engine.solve_forward(P, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
    adj, grads, st2, nfe2, ns2 = engine.solve_adjoint(P, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
    torch.cuda.synchronize()
    assert int(st.max()) == 0, "forward status %d" % int(st.max())
    return sol, adj.clone(), grads.flat.clone(), int(st2.max()), nfe2.clone(), ns2.clone()


def case(name, N, H, B, tgrid, std, seed=0, gscale=1.0, with_oracle=True):
    p = rand_params(N, H, seed + N, std)
    net = make_net(p)
    rs = np.random.RandomState(seed + 1)
    y0 = torch.from_numpy((rs.rand(B, N) * 0.9 + 0.05).astype(np.float32)).to(dev)
    t = torch.from_numpy(np.asarray(tgrid, np.float64)).to(dev)
    T = t.shape[1]
    G = torch.from_numpy((rs.randn(T, B, N) * gscale).astype(np.float32)).to(dev)
    kern = _lib.load().phx_debug_adjoint_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, 3)
    sol, a1, g1, s1, nfe1, ns1 = run(net, y0, t, G, "v1")
    _, a3, g3, s3, nfe3, ns3 = run(net, y0, t, G, "v3")
    os.environ["PHX_ADJ"] = "v3"
    k3 = _lib.load().phx_debug_adjoint_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, 3)
    os.environ.pop("PHX_ADJ")
    msg = "%-28s N=%5d H=%3d B=%4d T=%d default-kernel=%d forced=%d status v1/v3 %d/%d  steps %d..%d (v1 %d..%d)" % (
        name, N, H, B, T, kern, k3, s1, s3, int(ns3.min()), int(ns3.max()), int(ns1.min()), int(ns1.max()))
    e_a, e_g = relerr(a3.cpu().numpy(), a1.cpu().numpy()), relerr(g3.cpu().numpy(), g1.cpu().numpy())
    msg += "  v3-v1: adj_y0 %.2e grads %.2e nfe-equal %s" % (e_a, e_g, bool(torch.equal(nfe1, nfe3)))
    if with_oracle:
        onet = orc.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])
        tn = t.cpu().numpy()
        ref = sol.cpu().numpy().reshape(T, B, N).transpose(1, 0, 2).copy()
        adj_ref, gr = orc.adjoint_backward_per_sample(onet, tn, ref, G.cpu().numpy().transpose(1, 0, 2).copy(), method="dopri5",
                                                      theta_in_norm=False)
        HN = H * N
        flat = g3.cpu().numpy()
        got = {"Ws": flat[:HN].reshape(H, N), "Wp": flat[HN + H:2 * HN + H].reshape(H, N)}   # Ws | bs | Wp | bp | Wa | g
        msg += "  v3-oracle: adj_y0 %.2e Ws %.2e Wp %.2e" % (relerr(a3.cpu().numpy(), adj_ref), relerr(got["Ws"], gr["Ws"]),
                                                           relerr(got["Wp"], gr["Wp"]))
    print(msg, flush=True)
    return e_a, e_g


def grids(B, t0, t1, T=2, spread=0.0):
    return [[t0 + spread * b + (t1 - t0) * i / (T - 1) for i in range(T)] for b in range(B)]


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    case("single step", 777, 12, 21, grids(21, 0.0, 0.01), 0.06)
    case("multi step", 777, 12, 21, grids(21, 0.0, 0.5, spread=0.05), 0.06)
    case("three intervals", 350, 40, 64, grids(64, 0.0, 0.9, T=4, spread=0.01), 0.05)
    case("ragged, several groups", 350, 40, 150, grids(150, 0.0, 0.3, spread=0.002), 0.05)
    case("rejections (stiff-ish)", 96, 8, 40, grids(40, 0.0, 1.0, spread=0.02), 0.5 / np.sqrt(96), gscale=1.0)
    case("wide tile count (TPW>1)", 700, 40, 600, grids(600, 0.0, 0.05), 0.03, with_oracle=False)
    case("mixed directions", 350, 30, 6, [[0.0, 0.3, 0.7] if b % 2 == 0 else [1.0, 0.6, 0.1] for b in range(6)], 0.6 / np.sqrt(350))
    if not quick:
        # the breast-cancer shape: launch times of the two kernels (HIP events around the solve kernel)
        N, H, B = 11165, 40, 256
        p = rand_params(N, H, 11, 0.02)
        net = make_net(p)
        rs = np.random.RandomState(4)
        y0 = torch.from_numpy(np.clip(rs.randn(B, N) * 0.15 + 0.5, 0.03, 1.07).astype(np.float32)).to(dev)
        t = torch.tensor([[0.0, 0.0051]] * B, dtype=torch.float64, device=dev)
        G = torch.from_numpy((rs.randn(2, B, N) / (B * N)).astype(np.float32)).to(dev)
        outs = {}
        for variant in ("v1", "v3", "v1", "v3"):
            os.environ["PHX_ADJ"] = variant
            P = engine.params_cached(*pa.odenet.params_of(net))
            sol, st, nfe, ns = engine.solve_forward(P, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
            for _ in range(3):
                engine.solve_adjoint(P, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                adj, grads, st2, nfe2, ns2 = engine.solve_adjoint(P, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
            ev[1].record()
            torch.cuda.synchronize()
            outs[variant] = (adj.clone(), grads.flat.clone())
            print("C4 backward (solve + reduce) %s: %.3f ms per call, status %d, nfe %d" % (
                variant, ev[0].elapsed_time(ev[1]) / 10, int(st2.max()), int(nfe2[0])), flush=True)
        print("C4 v3 vs v1: adj_y0 %.2e grads %.2e" % (relerr(outs["v3"][0].cpu().numpy(), outs["v1"][0].cpu().numpy()),
                                                      relerr(outs["v3"][1].cpu().numpy(), outs["v1"][1].cpu().numpy())))
