"""The unfused stepper behind `odeint` / `odeint_adjoint` for modules that are not a PHOENIX ODENet
(phoenix_amd/generic.py, torch ops on the device): against the batched PyTorch restatement of the reference in
oracle/torch_baseline.py (CPU; itself pinned to the C oracle and the reference's goldens), against closed forms, and
adjoint vs backpropagation.  Device tensors only -- the package has no CPU compute path."""
import numpy as np
import pytest
import torch

import phoenix_amd as pa
from oracle import torch_baseline as tb

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Mlp(torch.nn.Module):
    def __init__(self, n, h, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.a = torch.nn.Parameter((torch.randn(h, n, generator=g) * 0.4).to(DEV))
        self.b = torch.nn.Parameter((torch.randn(n, h, generator=g) * 0.4).to(DEV))

    def forward(self, t, y):
        return torch.tanh(y @ self.a.t()) @ self.b.t() - 0.5 * y + 0.1 * torch.sin(3 * t).to(y.dtype)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("tgrid", [[0.0, 0.3, 0.9, 2.0], [2.0, 1.1, 0.0]])
def test_generic_dopri5_matches_the_restatement(tgrid):
    f = Mlp(6, 9)
    y0 = torch.randn(4, 1, 6, generator=torch.Generator().manual_seed(1))
    t = torch.tensor(tgrid)
    got = pa.odeint(f, y0.to(DEV), t.to(DEV))
    assert got.shape == (len(tgrid), 4, 1, 6) and torch.equal(got[0].cpu(), y0)
    got = got.cpu()
    tt = t.double() if tgrid[0] < tgrid[-1] else -t.double()
    a_c, b_c = f.a.detach().cpu(), f.b.detach().cpu()
    fc = lambda s, y: torch.tanh(y @ a_c.t()) @ b_c.t() - 0.5 * y + 0.1 * torch.sin(3 * s).to(y.dtype)   # noqa: E731
    ff = fc if tgrid[0] < tgrid[-1] else (lambda s, y: -fc(-s, y))
    solver = tb.Dopri5(lambda s, y: ff(s, y.view(4, 1, 6)).reshape(-1), y0.reshape(-1))
    with torch.no_grad():
        ref = solver.integrate(tt).view(len(tgrid), 4, 1, 6)
    assert rel(got.detach(), ref) < 1e-5        # the north-star bar; measured 4e-6 (summation order at rtol 1e-7)


@pytest.mark.parametrize("method,order", [("euler", 1), ("midpoint", 2), ("rk4", 4)])
def test_generic_fixed_grid_takes_one_step_per_interval_at_the_methods_order(method, order):
    lam = torch.tensor([[-0.7, 0.2], [0.1, -0.4]], device=DEV)
    f = lambda t, y: y @ lam.t()     # noqa: E731
    y0 = torch.tensor([[1.0, -2.0]], device=DEV)
    errs = []
    for n in ((1, 2) if order == 4 else (8, 16)):        # fp32: keep the rk4 errors above rounding
        t = torch.linspace(0, 2 if order == 4 else 1, n + 1, device=DEV)
        sol = pa.odeint(f, y0, t, method=method)
        assert sol.shape == (n + 1, 1, 2)
        exact = y0 @ torch.linalg.matrix_exp(lam.double() * float(t[-1])).t().float()
        errs.append(float((sol[-1] - exact).abs().max()))
    assert errs[0] / errs[1] > 2 ** order * 0.7      # halving the step divides the error by ~2^order
    # euler: literally y + dt f
    t = torch.tensor([0.0, 0.25, 1.0], device=DEV)
    s = pa.odeint(f, y0, t, method="euler")
    y1 = y0 + 0.25 * f(0, y0)
    assert torch.allclose(s[1], y1) and torch.allclose(s[2], y1 + 0.75 * f(0, y1))


def test_generic_adjoint_matches_backpropagation_through_the_solver():
    f = Mlp(5, 7, seed=3)
    y0 = torch.randn(3, 5, generator=torch.Generator().manual_seed(2)).to(DEV).requires_grad_(True)
    t = torch.tensor([0.0, 0.4, 1.0], device=DEV)
    w = torch.randn(3, 3, 5, generator=torch.Generator().manual_seed(4)).to(DEV)
    (pa.odeint(f, y0, t) * w).sum().backward()                      # plain backprop through the torch ops
    ref = [y0.grad.clone(), f.a.grad.clone(), f.b.grad.clone()]
    y0.grad = None; f.a.grad = None; f.b.grad = None
    sol = pa.odeint_adjoint(f, y0, t)
    (sol * w).sum().backward()
    got = [y0.grad, f.a.grad, f.b.grad]
    for a, b in zip(got, ref):
        assert rel(a, b) < 2e-5
    # error behaviour of the reference (rk_common.py:174-176, misc.py:184-186)
    with pytest.raises(AssertionError, match="max_num_steps"):
        pa.odeint(f, y0.detach(), torch.tensor([0.0, 50.0], device=DEV), options={"max_num_steps": 2})
    with pytest.raises(ValueError):
        pa.odeint(f, y0.detach(), t, method="nope")


def test_host_tensors_are_refused_for_every_kind_of_func():
    """no CPU compute path: an ODENet on the CPU raises, and so does any other callable"""
    net = pa.ODENet("cpu", 12, neurons=4)
    with pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):
        pa.odeint(net, torch.rand(2, 1, 12), torch.tensor([0.0, 1.0]))
    with pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):
        pa.odeint(lambda t, y: -y, torch.rand(2, 1, 12), torch.tensor([0.0, 1.0]))
    with pytest.raises(RuntimeError, match="no CPU path|must live on the GPU"):
        pa.odeint_adjoint(torch.nn.Linear(12, 12), torch.rand(2, 12), torch.tensor([0.0, 1.0]))
