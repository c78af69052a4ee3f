#!/usr/bin/env python3
"""Row f3 throughput (run on the GPU box): influence-scan solves per second at breast-cancer size.
    python tools/f3_throughput.py [N] [H] [genes] -> one line per mode (one call per launch / batched calls)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import phoenix_amd as pa                                           # noqa: E402
from phoenix_amd.analysis import gene_influence_scores             # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 11165
H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
G = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = pa.ODENet(dev, N, neurons=H)
with torch.no_grad():
    for prm in net.parameters():
        prm.mul_(0.5)
genes = list(range(0, N, max(1, N // G)))[:G]
# the loop as round 1 ran it: one odeint call per launch
t = torch.from_numpy(np.arange(0, 1, 0.1)).to(dev)
y = torch.rand(60, 1, N, device=dev) - 0.5
with torch.no_grad():
    pa.odeint(net, y, t, method="dopri5")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2 * len(genes)):
        pa.odeint(net, y, t, method="dopri5")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("N=%d H=%d  one odeint([60,1,N]) call per launch: %.1f solves/s" % (N, H, 2 * len(genes) / dt))
for gpl in (1, 2, 4, 8):
    torch.manual_seed(1)
    gene_influence_scores(net, N, "dopri5", device=dev, genes=genes[:gpl], genes_per_launch=gpl)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc = gene_influence_scores(net, N, "dopri5", device=dev, genes=genes, genes_per_launch=gpl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("N=%d H=%d  %d genes, %d per launch (%d calls of [60,1,N], 10 outputs): %.1f ms per gene, %.1f solves/s, "
          "%.3g trajectory-gene RHS rows integrated/s  (mean score %.3e)" %
          (N, H, len(genes), gpl, 2 * gpl, 1e3 * dt / len(genes), 2 * len(genes) / dt, 2 * len(genes) * 60 * N / dt,
           float(np.mean(sc))))
