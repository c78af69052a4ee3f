#!/usr/bin/env python3
"""Run on the GPU box: time of phx_rhs_vjp with dL/dy (full RHS) on a batch, MFMA kernel chain vs the VALU engine.
    python tools/rhs_vjp_time.py [N] [H] [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import phoenix_amd as pa                     # noqa: E402
from phoenix_amd import engine               # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11165
H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = pa.ODENet(dev, N, neurons=H)
P = engine.params_cached(*pa.odenet.params_of(net))
y = torch.rand(B, N, device=dev)
cot = torch.randn(B, N, device=dev)
for name, env in (("MFMA kernel chain", None), ("VALU engine (PHX_ENGINE=v0)", "v0")):
    if env:
        os.environ["PHX_ENGINE"] = env
    else:
        os.environ.pop("PHX_ENGINE", None)
    engine.rhs_vjp(P, y, cot)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        engine.rhs_vjp(P, y, cot)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    flops = 24.0 * B * N * H
    print("N=%d H=%d B=%d  full RHS VJP (dL/dy + 6 parameter gradients), %s: %.2f ms  (%.1f TFLOP/s algorithmic)" %
          (N, H, B, name, dt * 1e3, flops / dt / 1e12))
