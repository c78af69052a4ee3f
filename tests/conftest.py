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


# ---- engine-vs-reference dopri5 gradient errors of the whole GPU suite: every comparison of a
# multi-step dopri5 gradient against the oracle / a golden goes through grad_err(), which records the value; the last test
# of tests/test_gpu_parity.py holds the MEDIAN of all of them to the north-star bar (1e-5) -- the per-case tolerance
# TOL_DOPRI_GRAD (2.5e-5) is the accept/reject noise ceiling of single cases -- and the terminal summary prints the
# distribution.
GRAD_ERRORS = []


def grad_err(a, b, tag=""):
    e = relerr(a, b)
    GRAD_ERRORS.append((e, tag))
    return e


def pytest_terminal_summary(terminalreporter):
    if not GRAD_ERRORS:
        return
    v = np.sort(np.array([e for e, _ in GRAD_ERRORS]))
    q = lambda x: float(v[min(len(v) - 1, int(x * len(v)))])   # noqa: E731
    terminalreporter.write_line(
        "dopri5 gradient errors vs oracle / goldens over the suite: n=%d median %.2e p90 %.2e p99 %.2e max %.2e (%s)"
        % (len(v), q(0.5), q(0.9), q(0.99), float(v[-1]), max(GRAD_ERRORS)[1]))
