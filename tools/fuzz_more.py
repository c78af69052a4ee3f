#!/usr/bin/env python3
"""One-off verification (GPU box): the seeded shape fuzz of tests/test_gpu_parity.py on seeds outside the suite's range.
usage: python tools/fuzz_more.py first last"""
import os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import phoenix_amd as pa
import test_gpu_parity as T
dev = torch.device("cuda:0")
_relerr = T.relerr
_last = []
def relerr(a, b):
    v = _relerr(a, b)
    _last.append(float(v))
    return v
T.relerr = relerr
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    _last.clear()
    try:
        T.test_random_shapes_mfma_engine_agrees_with_valu_engine(pa, dev, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e).split("\n")[0][:200], "| last relerr %.3e (largest so far %.3e)" % (_last[-1], max(_last)), flush=True)
    except Exception as e:
        bad += 1
        print("seed", seed, "ERROR:", repr(e)[:200], flush=True)
print("seeds %d..%d: %d failures" % (lo, hi - 1, bad))
