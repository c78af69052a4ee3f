"""Diagnostic: SHA-1 of (trajectories, dL/dy0, parameter gradients) of a few per-sample dopri5 solves -- run it under two
builds of the library (PHX_DIAG=1 PHX_LIB=...) to see whether they agree bit for bit."""
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import phoenix_amd as pa
import test_gpu_parity as T
dev = torch.device("cuda:0")
for (N, H, B) in [(1100, 16, 37), (777, 12, 21), (2080, 40, 64), (11165, 40, 256), (350, 40, 150)]:
    p = T.rand_params(N, H, seed=7 * N + H, std=min(0.05, 2.0 / np.sqrt(N)))   # keeps the big shapes non-stiff
    net = T.make_net(pa, dev, p)
    r = np.random.RandomState(2)
    y0 = r.rand(B, N).astype(np.float32)
    t = np.stack([np.array([0.1 * (b % 7), 0.1 * (b % 7) + 0.3 + 0.02 * (b % 16)]) for b in range(B)]).astype(np.float32)
    G = r.randn(2, B, 1, N).astype(np.float32)
    T.zero_grads(net)
    y0t = torch.from_numpy(y0).to(dev).reshape(B, 1, N).requires_grad_(True)
    sol = pa.odeint_adjoint(net, y0t, torch.from_numpy(t).to(dev), method="dopri5")
    (sol * torch.from_numpy(G).to(dev)).sum().backward()
    h = hashlib.sha1()
    h.update(sol.detach().cpu().numpy().tobytes()); h.update(y0t.grad.cpu().numpy().tobytes())
    for k, v in sorted(T.grads_of(net).items()): h.update(np.ascontiguousarray(v).tobytes())
    print(N, H, B, h.hexdigest()[:16])
