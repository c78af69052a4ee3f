#!/usr/bin/env python3
"""Diagnostic (GPU box): the hidden-chunked third-generation kernels (k1_solve_fwd3c / k1_solve_adj3c, H > 48) against the
older kernels for the same shapes (PHX_V3C=0: k1_solve_fwd + k1_solve_adj2 / k1_solve_adj) and the CPU oracle -- resident
and re-staged chunk slots, single- and multi-step intervals, several intervals, ragged batches, TPW > 1 --, then the
launch times of both at the yeast and B-cell shapes.  usage: python tools/v3c_check.py [quick|time]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import phoenix_amd as pa  # noqa: E402
from phoenix_amd import _lib, engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
ENVK = ("PHX_V3C", "PHX_V3C_HB", "PHX_V3C_NTG", "PHX_V3C_SPLIT", "PHX_V3C_NB", "PHX_V3C_TPW", "PHX_V3C_RES", "PHX_V3C_SLOTS", "PHX_ADJ", "PHX_FWD")


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rand_params(N, H, seed, std, sparse=0.0):
    r = np.random.RandomState(seed)
    g = r.rand(N).astype(np.float32)
    g[r.rand(N) < 0.1] *= -1

    def w(shape):
        x = (r.randn(*shape) * std).astype(np.float32)
        if sparse > 0:
            x[r.rand(*shape) < sparse] = 0.0
        return x
    return {"Ws": w((H, N)), "bs": r.uniform(-.2, .2, H).astype(np.float32),
            "Wp": w((H, N)), "bp": r.uniform(-.2, .2, H).astype(np.float32),
            "Wa": w((N, 2 * H)), "g": g}


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


def setenv(env):
    for k in ENVK:
        os.environ.pop(k, None)
    os.environ.update(env)


def run(net, y0, t, G, env, strict=True):
    setenv(env)
    P = engine.params_cached(*pa.odenet.params_of(net))
    sol, st, nfe, ns = engine.solve_forward(P, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
    torch.cuda.synchronize()
    assert int(st.max()) == 0 or not strict, "forward status %d (%s)" % (int(st.max()), env)
    if int(st.max()) != 0:      # (fuzz: a problem that fails -- every plan must fail it the same way)
        return sol.clone(), sol.clone(), sol.clone(), 100 + int(st.max()), nfe.clone(), nfe.clone(), ns.clone(), ns.clone()
    adj, grads, st2, nfe2, ns2 = engine.solve_adjoint(P, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
    torch.cuda.synchronize()
    return sol.clone(), adj.clone(), grads.flat.clone(), int(st2.max()), nfe.clone(), nfe2.clone(), ns.clone(), ns2.clone()


def case(name, N, H, B, tgrid, std, seed=0, gscale=1.0, with_oracle=True, env=None, sparse=0.0, yrange=(0.05, 0.95)):
    env = env or {}
    p = rand_params(N, H, seed + N, std, sparse)
    net = make_net(p)
    rs = np.random.RandomState(seed + 1)
    y0 = torch.from_numpy((rs.rand(B, N) * (yrange[1] - yrange[0]) + yrange[0]).astype(np.float32)).to(dev)
    t = torch.from_numpy(np.asarray(tgrid, np.float64)).to(dev)
    T = t.shape[1]
    G = torch.from_numpy((rs.randn(T, B, N) * gscale).astype(np.float32)).to(dev)
    s_old, a_old, g_old, st_old, nf_old, nb_old, nsf_old, nsb_old = run(net, y0, t, G, {"PHX_V3C": "0"})
    s_new, a_new, g_new, st_new, nf_new, nb_new, nsf_new, nsb_new = run(net, y0, t, G, env)
    setenv(env)
    kf = _lib.load().phx_debug_forward_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, 3)
    ka = _lib.load().phx_debug_adjoint_kernel_m(N, H, B, T, _lib.CTRL_PER_TRAJECTORY, 3)
    plan = (C6 * 6)()
    off, nwg = C.c_size_t(0), C.c_int(0)
    _lib.load().phx_debug_profile_region(_lib.OP_ADJOINT, N, H, B, T, _lib.CTRL_PER_TRAJECTORY, C.byref(off), C.byref(nwg), plan)
    msg = "%-26s N=%5d H=%3d B=%4d T=%d kernels f%d/a%d plan[NW,TPW,NB,G,TG,HC*10+res]=%s status old/new %d/%d fwd steps %d..%d (old %d..%d) bwd steps %d..%d (old %d..%d)" % (
        name, N, H, B, T, kf, ka, list(plan), st_old, st_new, int(nsf_new.min()), int(nsf_new.max()), int(nsf_old.min()),
        int(nsf_old.max()), int(nsb_new.min()), int(nsb_new.max()), int(nsb_old.min()), int(nsb_old.max()))
    msg += "\n    new-old: sol %.2e adj_y0 %.2e grads %.2e nfe-equal f %s b %s" % (
        relerr(s_new.cpu().numpy(), s_old.cpu().numpy()), relerr(a_new.cpu().numpy(), a_old.cpu().numpy()),
        relerr(g_new.cpu().numpy(), g_old.cpu().numpy()), bool(torch.equal(nf_old, nf_new)), bool(torch.equal(nb_old, nb_new)))
    if with_oracle:
        onet = orc.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])
        tn = t.cpu().numpy()
        ref = orc.odeint_per_sample(onet, y0.cpu().numpy(), tn, method="dopri5")     # [B, T, N]
        got = s_new.cpu().numpy().reshape(T, B, N).transpose(1, 0, 2)
        adj_ref, gr = orc.adjoint_backward_per_sample(onet, tn, ref, G.cpu().numpy().transpose(1, 0, 2).copy(), method="dopri5",
                                                      theta_in_norm=True)
        HN = H * N
        flat = g_new.cpu().numpy()          # Ws | bs | Wp | bp | Wa | g
        o = 0
        parts = {}
        for k, n in (("Ws", HN), ("bs", H), ("Wp", HN), ("bp", H), ("Wa", 2 * HN), ("g", N)):
            parts[k] = flat[o:o + n]
            o += n
        msg += "\n    new-oracle: sol %.2e adj_y0 %.2e" % (relerr(got, ref), relerr(a_new.cpu().numpy(), adj_ref))
        for k in ("Ws", "bs", "Wp", "bp", "Wa", "g"):
            msg += " %s %.2e" % (k, relerr(parts[k].reshape(np.asarray(gr[k]).shape), gr[k]))
    print(msg, flush=True)


def grids(B, t0, t1, T=2, spread=0.0):
    return [[t0 + spread * b + (t1 - t0) * i / (T - 1) for i in range(T)] for b in range(B)]


def timing(name, N, H, B, tg, std, sparse, envs, yrange=(0.05, 0.95), reps=5):
    p = rand_params(N, H, 11, std, sparse)
    net = make_net(p)
    rs = np.random.RandomState(4)
    y0 = torch.from_numpy((rs.rand(B, N) * (yrange[1] - yrange[0]) + yrange[0]).astype(np.float32)).to(dev)
    t = torch.tensor([tg] * B, dtype=torch.float64, device=dev)
    G = torch.from_numpy((rs.randn(2, B, N) / (B * N)).astype(np.float32)).to(dev)
    base = None
    for env in envs:
        setenv(env)
        P = engine.params_cached(*pa.odenet.params_of(net))
        try:
            for _ in range(2):
                sol, st, nfe, ns = engine.solve_forward(P, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
                adj, grads, st2, nfe2, ns2 = engine.solve_adjoint(P, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
            torch.cuda.synchronize()
        except Exception as exc:   # noqa: BLE001
            print("%s %s: FAILED %r" % (name, env, exc), flush=True)
            continue
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(reps):
            sol, st, nfe, ns = engine.solve_forward(P, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
        ev[1].record()
        for _ in range(reps):
            adj, grads, st2, nfe2, ns2 = engine.solve_adjoint(P, t, sol, G, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
        ev[2].record()
        torch.cuda.synchronize()
        plan = (C6 * 6)()
        off, nwg = C.c_size_t(0), C.c_int(0)
        _lib.load().phx_debug_profile_region(_lib.OP_ADJOINT, N, H, B, 2, _lib.CTRL_PER_TRAJECTORY, C.byref(off), C.byref(nwg), plan)
        cur = (sol.clone(), adj.clone(), grads.flat.clone())
        extra = ""
        if base is None:
            base = cur
        else:
            extra = "  vs first: sol %.2e adj %.2e grads %.2e" % tuple(relerr(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(cur, base))
        print("%s %-44s fwd %.3f ms  adj(+reduce) %.3f ms  status %d/%d nfe f %d b %d plan %s%s" % (
            name, env, ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps, int(st.max()), int(st2.max()),
            int(nfe.max()), int(nfe2.max()), list(plan), extra), flush=True)


import ctypes as C  # noqa: E402
C6 = C.c_int

if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "quick"):
        # resident chunk slots (small gene tiles)
        case("H=120 single step", 777, 120, 21, grids(21, 0.0, 0.01), 0.03)
        case("H=120 half-block tiles, empty half", 322, 120, 37, grids(37, 0.0, 0.6, spread=0.002), 0.03)
        case("H=120 whole-block tiles (HB=0)", 322, 120, 37, grids(37, 0.0, 0.6, spread=0.002), 0.03, env={"PHX_V3C_HB": "0"})
        case("H=200 half-block tiles, groups", 500, 200, 100, grids(100, 0.0, 0.3), 0.02)
        case("H=120 multi step", 777, 120, 21, grids(21, 0.0, 0.5, spread=0.05), 0.03)
        case("H=64 three intervals", 350, 64, 64, grids(64, 0.0, 0.9, T=4, spread=0.01), 0.04)
        case("H=100 ragged, groups", 350, 100, 150, grids(150, 0.0, 0.3, spread=0.002), 0.04)
        case("H=131 (not HALF)", 500, 131, 37, grids(37, 0.0, 0.4, spread=0.01), 0.03)
        case("H=200 rejections", 200, 200, 40, grids(40, 0.0, 1.0, spread=0.02), 0.3 / np.sqrt(200), gscale=1.0)
        case("mixed directions", 350, 120, 6, [[0.0, 0.3, 0.7] if b % 2 == 0 else [1.0, 0.6, 0.1] for b in range(6)], 0.5 / np.sqrt(350))
        # one re-staged slot, forced
        case("H=120 restaged NB=2", 777, 120, 70, grids(70, 0.0, 0.3, spread=0.003), 0.03, env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "2"})
        case("H=64 (HC even) NB=2 slots", 350, 64, 40, grids(40, 0.0, 0.5, spread=0.01), 0.04, env={"PHX_V3C_NB": "2"})
        case("H=200 restaged NB=4", 1500, 200, 50, grids(50, 0.0, 0.2, spread=0.002), 0.02, env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "4"})
        case("H=200 restaged TPW=2 NB=4", 1500, 200, 100, grids(100, 0.0, 0.2, spread=0.002), 0.02,
             env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "4", "PHX_V3C_TPW": "2"})
        case("H=200 restaged TPW=4 NB=2", 900, 200, 200, grids(200, 0.0, 0.2, spread=0.001), 0.02,
             env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "2", "PHX_V3C_TPW": "4"}, with_oracle=False)
    if mode == "hb":        # half-block gene tiles against whole-block ones
        timing("yeast N=2000 H=120 B=23", 2000, 120, 23, [0.0, 5.0], 0.05, 0.95, [{"PHX_V3C_HB": "0"}, {}], yrange=(-0.3, 0.9))
        timing("yeast N=2000 H=120 B=4", 2000, 120, 4, [0.0, 5.0], 0.05, 0.95, [{"PHX_V3C_HB": "0"}, {}], yrange=(-0.3, 0.9))
        timing("yeast N=2000 H=120 B=128", 2000, 120, 128, [0.0, 5.0], 0.05, 0.95, [{"PHX_V3C_HB": "0"}, {}], yrange=(-0.3, 0.9))
        timing("N=350 H=100 B=64", 350, 100, 64, [0.0, 2.0], 0.05, 0.95, [{"PHX_V3C_HB": "0"}, {}])
    if mode == "split":     # block split of small batches on multi-block gene tiles
        case("H=200 restaged NB=4, one tile, split 4", 1500, 200, 10, grids(10, 0.0, 0.3, spread=0.002), 0.02,
             env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "4"})
        case("H=120 resident NB=2, two tiles, split 2", 777, 120, 20, grids(20, 0.0, 0.5, T=3), 0.03, env={"PHX_V3C_NB": "2"})
        case("H=200 restaged NB=2, ragged last tile", 1511, 200, 7, grids(7, 0.4, 0.0), 0.02, env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "2"})
        case("same, split off", 1511, 200, 7, grids(7, 0.4, 0.0), 0.02, env={"PHX_V3C_RES": "0", "PHX_V3C_NB": "2", "PHX_V3C_SPLIT": "0"})
        case("B-cell shape, 2 trajectories", 14691, 200, 2, grids(2, 0.0, 1.0), 0.003, with_oracle=False, sparse=0.95)
        timing("bcell N=14691 H=200 B=2", 14691, 200, 2, [0.0, 1.0], 0.05, 0.95, [{"PHX_V3C_SPLIT": "0"}, {}, {"PHX_V3C_NB": "4"}], reps=5)
        timing("N=9000 H=120 B=16", 9000, 120, 16, [0.0, 2.0], 0.05, 0.95, [{"PHX_V3C_SPLIT": "0"}, {}], reps=5)
    if mode == "fuzz":      # random shapes: the default plans against whole-block tiles in four-tile groups and the older kernels
        rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
        worst = 0.0
        for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
            N = int(rs.choice([rs.randint(20, 64), rs.randint(64, 700), rs.randint(700, 3000)]))
            H = int(rs.choice([rs.randint(49, 97), rs.randint(97, 145), rs.randint(145, 257)]))
            B = int(rs.choice([rs.randint(1, 17), rs.randint(17, 65), rs.randint(65, 200)]))
            T = int(rs.randint(2, 5))
            sgn = -1.0 if rs.rand() < 0.3 else 1.0
            tg = [[sgn * (0.05 * (b % 5) + 0.25 * i * (1 + 0.1 * (b % 3))) for i in range(T)] for b in range(B)]
            p = rand_params(N, H, 1000 + it, 0.03 + 0.03 * rs.rand())
            net = make_net(p)
            y0 = torch.from_numpy(rs.rand(B, N).astype(np.float32)).to(dev)
            t = torch.from_numpy(np.asarray(tg, np.float64)).to(dev)
            G = torch.from_numpy(rs.randn(T, B, N).astype(np.float32)).to(dev)
            try:
                base = run(net, y0, t, G, {"PHX_V3C_HB": "0", "PHX_V3C_NTG": "4"}, strict=False)
                new = run(net, y0, t, G, {}, strict=False)
                old = run(net, y0, t, G, {"PHX_V3C": "0"}, strict=False)
                if max(base[3], new[3], old[3]) >= 100:
                    same = base[3] == new[3] == old[3]
                    print("fuzz %2d N=%4d H=%3d B=%3d T=%d: the forward solve fails (status whole-block %d, default %d, older %d)%s" % (
                        it, N, H, B, T, base[3] - 100, new[3] - 100, old[3] - 100, "" if same else "   <-- LOOK"), flush=True)
                    if not same:
                        worst = float("inf")
                    continue
            except Exception as exc:   # noqa: BLE001
                print("fuzz %d N=%d H=%d B=%d T=%d: FAILED %r" % (it, N, H, B, T, exc), flush=True)
                worst = float("inf")
                continue
            setenv({})
            plan = (C6 * 6)()
            off, nwg = C.c_size_t(0), C.c_int(0)
            _lib.load().phx_debug_profile_region(_lib.OP_ADJOINT, N, H, B, T, _lib.CTRL_PER_TRAJECTORY, C.byref(off), C.byref(nwg), plan)
            e1 = max(relerr(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(new[:3], base[:3]))
            e2 = max(relerr(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(new[:3], old[:3]))
            worst = max(worst, e1, e2)
            print("fuzz %2d N=%4d H=%3d B=%3d T=%d dir %+d plan %s status %d: vs whole-block four-tile plan %.2e, vs older kernels %.2e%s" % (
                it, N, H, B, T, int(sgn), list(plan), new[3], e1, e2, "   <-- LOOK" if max(e1, e2) > 6e-5 or new[3] != 0 else ""), flush=True)
        print("fuzz worst %.2e" % worst, flush=True)
    if mode == "ntg":       # trajectory tiles per batch group (more groups of fewer tiles: more helper waves per tile)
        for B in (23, 40, 64):
            timing("yeast N=2000 H=120 B=%d" % B, 2000, 120, B, [0.0, 5.0], 0.05, 0.95,
                   [{}, {"PHX_V3C_NTG": "1"}, {"PHX_V3C_NTG": "1", "PHX_V3C_HB": "1"}, {"PHX_V3C_NTG": "2"}], yrange=(-0.3, 0.9))
        timing("N=600 H=120 B=64", 600, 120, 64, [0.0, 2.0], 0.05, 0.95, [{}, {"PHX_V3C_NTG": "1"}, {"PHX_V3C_NTG": "2"}])
        timing("N=1000 H=200 B=32", 1000, 200, 32, [0.0, 2.0], 0.05, 0.95, [{}, {"PHX_V3C_NTG": "1"}])
    if mode == "ntggrid":   # the tiles-per-group rule against the plan before it (PHX_V3C_NTG=4)
        for H in (120, 200):
            for N in (250, 350, 600, 1000, 2000, 3500) if H == 120 else (350, 1000, 2000):
                for B in (16, 23, 32, 40, 48, 64, 100, 128):
                    timing("N=%d H=%d B=%d" % (N, H, B), N, H, B, [0.0, 2.0], 0.05, 0.95, [{"PHX_V3C_NTG": "4"}, {}], reps=3)
    if mode == "hbgrid":    # where half-block tiles pay: gene blocks x trajectory tiles
        for N in (250, 350, 600, 800, 1000, 2000, 3500):
            for B in (16, 32, 48, 64, 128):
                timing("N=%d H=120 B=%d" % (N, B), N, 120, B, [0.0, 2.0], 0.05, 0.95, [{"PHX_V3C_HB": "0"}, {"PHX_V3C_HB": "1"}], reps=3)
        for N in (350, 2000):
            for B in (16, 64):
                timing("N=%d H=200 B=%d" % (N, B), N, 200, B, [0.0, 2.0], 0.05, 0.95, [{"PHX_V3C_HB": "0"}, {"PHX_V3C_HB": "1"}], reps=3)
    if mode == "plans":     # plan alternatives of the chunked kernels at the two bench shapes (environment overrides)
        timing("yeast N=2000 H=120 B=23", 2000, 120, 23, [0.0, 5.0], 0.05, 0.95,
               [{}, {"PHX_V3C_NB": "2"}, {"PHX_V3C_NB": "2", "PHX_V3C_RES": "0"}, {"PHX_V3C_RES": "0"}], yrange=(-0.3, 0.9))
        timing("yeast N=2000 H=120 B=4", 2000, 120, 4, [0.0, 5.0], 0.05, 0.95, [{}, {"PHX_V3C_NB": "2"}], yrange=(-0.3, 0.9))
        timing("bcell N=14691 H=200 B=256", 14691, 200, 256, [0.0, 1.0], 0.05, 0.95,
               [{}, {"PHX_V3C_NB": "2"}, {"PHX_V3C_NB": "8"}, {"PHX_V3C_TPW": "1"}], reps=3)
    if mode in ("all", "time"):
        timing("yeast N=2000 H=120 B=23", 2000, 120, 23, [0.0, 5.0], 0.05, 0.95, [{"PHX_V3C": "0"}, {}], yrange=(-0.3, 0.9))
        timing("bcell N=14691 H=200 B=256", 14691, 200, 256, [0.0, 1.0], 0.05, 0.95,
               [{"PHX_V3C": "0"}, {}, {"PHX_V3C_NB": "2", "PHX_V3C_TPW": "4"}, {"PHX_V3C_SLOTS": "4"}], reps=3)
