import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def sub(d, prefix):
    """keys under `prefix` with the prefix stripped"""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def relerr(a, b):
    """max |a-b| normalised by max |b| (the tolerance convention of SURVEY.md section 8c / G4)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / den


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def net_from(orc, d, prefix="p_"):
    p = sub(d, prefix)
    return orc.Net(p["Ws"], p["bs"], p["Wp"], p["bp"], p["Wa"], p["g"])
