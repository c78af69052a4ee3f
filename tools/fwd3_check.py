#!/usr/bin/env python3
"""Diagnostic (GPU box): the third-generation forward kernel against the first one (PHX_FWD=v1) on a few shapes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import phoenix_amd as pa  # noqa: E402
from phoenix_amd import _lib, engine  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from adj3_check import rand_params, make_net, relerr, grids  # noqa: E402

dev = torch.device("cuda:0")


def run(net, y0, t, env):
    for k in ("PHX_FWD", "PHX_V1_MAXNW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    P = engine.params_cached(*pa.odenet.params_of(net))
    sol, st, nfe, ns = engine.solve_forward(P, y0, t, "dopri5", _lib.CTRL_PER_TRAJECTORY, 1e-7, 1e-9, True, 0)
    torch.cuda.synchronize()
    return sol.clone(), int(st.max()), nfe.clone(), ns.clone()


def case(name, N, H, B, tgrid, std, seed=0):
    p = rand_params(N, H, seed + N, std)
    net = make_net(p)
    rs = np.random.RandomState(seed + 1)
    y0 = torch.from_numpy((rs.rand(B, N) * 0.9 + 0.05).astype(np.float32)).to(dev)
    t = torch.from_numpy(np.asarray(tgrid, np.float64)).to(dev)
    s1, st1, nfe1, ns1 = run(net, y0, t, {"PHX_FWD": "v1"})
    out = "%-22s N=%5d H=%3d B=%4d T=%d" % (name, N, H, B, t.shape[1])
    for label, env in (("nw4", {}), ("nw2", {"PHX_V1_MAXNW": "2"}), ("nw1", {"PHX_V1_MAXNW": "1"})):
        s3, st3, nfe3, ns3 = run(net, y0, t, env)
        out += "  | %s: status %d err %.2e steps %d..%d (v1 %d..%d)" % (label, st3, relerr(s3.cpu().numpy(), s1.cpu().numpy()),
                                                                     int(ns3.min()), int(ns3.max()), int(ns1.min()), int(ns1.max()))
    print(out, flush=True)


if __name__ == "__main__":
    case("single step", 777, 12, 21, grids(21, 0.0, 0.01), 0.06)
    case("multi step", 777, 12, 21, grids(21, 0.0, 0.5, spread=0.05), 0.06)
    case("T=4", 350, 40, 64, grids(64, 0.0, 0.9, T=4, spread=0.01), 0.05)
    case("B=150", 350, 40, 150, grids(150, 0.0, 0.3, spread=0.002), 0.05)
    for nblk in (47, 65, 117):
        case("G=%d" % nblk, 32 * nblk, 40, 256, grids(256, 0.0, 0.02), 0.03)
    case("B=600", 700, 40, 600, grids(600, 0.0, 0.05), 0.03)
    case("C4", 11165, 40, 256, grids(256, 0.0, 0.0051), 0.02)
