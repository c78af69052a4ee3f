#!/usr/bin/env python3
"""Run on the GPU box: time of the prior-target SpMM (row f1) at breast-cancer size.
    python tools/prior_spmm_time.py [N] [K] [density]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from phoenix_amd.prior import PriorMatrix, prior_targets      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11165
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
rng = np.random.default_rng(0)
nnz = int(dens * N * N)
rows, cols = rng.integers(0, N, nnz), rng.integers(0, N, nnz)
vals = rng.choice(np.array([-1.0, 1.0], np.float32), nnz)
P = PriorMatrix(rows, cols, vals, N, "cuda")
X = torch.rand(K, 1, N, device="cuda")
prior_targets(X, P)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = prior_targets(X, P)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
alg = 8.0 * P.nnz + 8.0 * K * N          # CSC entries once + X read + out written
print("prior targets X[%d,%d] @ P (nnz %d = %.2f %%): %.2f ms, %.3g gathers/s, algorithmic %.1f GB/s (HBM view), "
      "dense torch.matmul would be %.1f GFLOP" % (K, N, P.nnz, 100.0 * P.nnz / N / N, dt * 1e3, K * P.nnz / dt, alg / dt / 1e9,
                                                  2.0 * K * N * N / 1e9))
t0 = time.perf_counter()
D = P.to_dense().to("cuda")
torch.cuda.synchronize()
t1 = time.perf_counter()
ref = torch.matmul(X.reshape(K, N), D)
torch.cuda.synchronize()
t2 = time.perf_counter()
ref = torch.matmul(X.reshape(K, N), D)
torch.cuda.synchronize()
t3 = time.perf_counter()
print("reference formulation on the same GPU: dense matrix build %.0f ms (%.0f MB), torch.matmul %.2f ms; max |diff| %.2e" %
      ((t1 - t0) * 1e3, N * N * 4 / 1e6, (t3 - t2) * 1e3, float((ref - out.reshape(K, N)).abs().max())))
