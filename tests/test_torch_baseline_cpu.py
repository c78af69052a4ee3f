"""Pins the PyTorch-CPU batched baseline of bench.py (oracle/torch_baseline.py) against the C oracle, which is itself
pinned by the goldens captured from the reference.  CPU only."""
import numpy as np
import torch

from conftest import relerr


def test_batched_torch_baseline_matches_c_oracle(oracle):
    from oracle import torch_baseline as tb
    r = np.random.RandomState(3)
    N, H, B = 48, 6, 5
    p = dict(Ws=r.randn(H, N) * 0.08, bs=r.randn(H) * 0.1, Wp=r.randn(H, N) * 0.08, bp=r.randn(H) * 0.1,
             Wa=r.randn(N, 2 * H) * 0.08, g=r.rand(1, N))
    p = {k: v.astype(np.float32) for k, v in p.items()}
    onet = oracle.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])
    tnet = tb.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])
    y0 = r.rand(B, N).astype(np.float32)
    t = np.array([0.0, 0.4, 0.9])
    G = r.randn(3, B, N).astype(np.float32)
    ref = oracle.odeint(onet, y0, t, method="dopri5")                         # [T,B,N], one controller for the batch
    adj_ref, gr_ref = oracle.adjoint_backward(onet, t, ref, G, method="dopri5", theta_in_norm=True)
    sol, nfe = tb.odeint(tnet, torch.from_numpy(y0).reshape(B, 1, N), torch.from_numpy(t))
    assert nfe > 8
    assert relerr(sol.numpy().reshape(3, B, N), ref) < 1e-5
    gy, gp, nfe_b = tb.adjoint_backward(tnet, torch.from_numpy(t), sol, torch.from_numpy(G).reshape(3, B, 1, N))
    assert nfe_b > 8
    assert relerr(gy.numpy().reshape(B, N), adj_ref) < 5e-5
    names = ("g", "Wp", "bp", "Ws", "bs", "Wa")
    for k, gt in zip(names, gp):
        assert relerr(gt.numpy().reshape(gr_ref[k].shape), gr_ref[k]) < 5e-5, k
