"""TEST / BENCH INFRASTRUCTURE ONLY -- never imported by phoenix_amd.

PyTorch-CPU restatement of the reference's path in its BATCHED shape: one `odeint_adjoint(ODENet, y0[B,1,N], t)`
call over the whole batch with shared step control, forward solve + adjoint backward.  BASELINE.md section 3 asks for
this formulation next to the per-sample loop (which oracle/phx_oracle.c restates): it is the stronger CPU baseline,
because every RHS evaluation is three dense GEMMs over the batch on all host cores.  The reference itself is pure
Python and cannot travel to the GPU box; this file follows it function by function:

    ODENet.forward                          ode_net/code/odenet.py:85-91
    Dopri5 tableau                          torchdiffeq/_impl/dopri5.py:5-30
    _runge_kutta_step, _adaptive_step       torchdiffeq/_impl/rk_common.py:39-77, 150-228
    _select_initial_step, error ratio,
    _optimal_step_size                      torchdiffeq/_impl/misc.py:47-103
    quartic dense output                    torchdiffeq/_impl/interp.py:1-47
    OdeintAdjointMethod.backward            torchdiffeq/_impl/adjoint.py:32-162 (mixed Linf/RMS norm :72-78)

It is pinned by tests/test_torch_baseline_cpu.py against the C oracle (itself pinned by the reference's goldens)."""
import torch

ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
         -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.]
C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
         187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


class Net:
    """the six ODENet parameters as torch CPU tensors (reference layouts)"""

    def __init__(self, Ws, bs, Wp, bp, Wa, g):
        t = lambda x: torch.as_tensor(x, dtype=torch.float32).clone()
        self.Ws, self.bs, self.Wp, self.bp, self.Wa, self.g = t(Ws), t(bs), t(Wp), t(bp), t(Wa), t(g).reshape(1, -1)

    def params(self):   # parameters() order of the reference class
        return (self.g, self.Wp, self.bp, self.Ws, self.bs, self.Wa)

    def __call__(self, t, y):   # odenet.py:85-91
        s = y - 0.5
        a = s / (1 + torch.abs(s))
        l = torch.log1p(a)
        u = torch.nn.functional.linear(a, self.Ws, self.bs)
        p = torch.exp(torch.nn.functional.linear(l, self.Wp, self.bp))
        j = torch.nn.functional.linear(torch.cat((u, p), dim=-1), self.Wa)
        return torch.relu(self.g) * (j - y)


def _rms(x):
    return x.pow(2).mean().sqrt()


def _mixed_norm(sizes):   # misc.py:14-24
    def norm(x):
        out, tot = [], 0
        for n in sizes:
            out.append(_rms(x[tot:tot + n]))
            tot += n
        return max(out)
    return norm


class Dopri5:
    """RKAdaptiveStepsizeODESolver with the Dormand-Prince tableau on a FLAT state vector (rk_common.py:106-228)"""

    def __init__(self, func, y0, rtol=1e-7, atol=1e-9, norm=_rms, max_num_steps=2 ** 31 - 1):
        self.func, self.y0, self.rtol, self.atol, self.norm, self.max_num_steps = func, y0, rtol, atol, norm, max_num_steps
        self.nfe = 0

    def _f(self, t, y):
        self.nfe += 1
        return self.func(t, y)

    def _initial_step(self, t0, y0, f0):   # misc.py:47-86, order 4
        scale = self.atol + torch.abs(y0) * self.rtol
        d0, d1 = self.norm(y0 / scale), self.norm(f0 / scale)
        h0 = torch.tensor(1e-6, dtype=torch.float64) if (d0 < 1e-5 or d1 < 1e-5) else (0.01 * d0 / d1).double()
        y1 = y0 + h0.to(y0.dtype) * f0
        f1 = self._f(t0 + h0, y1)
        d2 = self.norm((f1 - f0) / scale) / h0.to(y0.dtype)
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = torch.max(torch.tensor(1e-6, dtype=torch.float64), h0 * 1e-3)
        else:
            h1 = ((0.01 / max(d1, d2)) ** (1. / 5)).double()
        return torch.min(100 * h0, h1)

    def integrate(self, t):
        t = t.double()
        y0 = self.y0
        f0 = self._f(t[0], y0)
        dt = self._initial_step(t[0], y0, f0)
        t0 = t1 = t[0]
        y1, f1 = y0, f0
        coeffs = None
        out = [y0]
        beta = [torch.tensor(b, dtype=y0.dtype) for b in BETA]
        c_sol, c_err, c_mid = (torch.tensor(c, dtype=y0.dtype) for c in (C_SOL, C_ERR, C_MID))
        for i in range(1, len(t)):
            n_steps = 0
            while t[i] > t1:
                assert n_steps < self.max_num_steps, "max_num_steps exceeded"
                assert t1 + dt > t1, "underflow in dt"
                # ---- _runge_kutta_step (rk_common.py:39-77)
                dtf = dt.to(y0.dtype)
                k = torch.empty(y1.shape + (7,), dtype=y0.dtype)
                k[..., 0] = f1
                for s, (al, be) in enumerate(zip(ALPHA, beta)):
                    yi = y1 + k[..., :s + 1].matmul(be * dtf)
                    k[..., s + 1] = self._f(t1 + al * dt, yi)
                y_new = y1 + k.matmul(dtf * c_sol)
                err = k.matmul(dtf * c_err)
                tol = self.atol + self.rtol * torch.max(y1.abs(), y_new.abs())
                ratio = self.norm(err / tol)
                accept = bool(ratio <= 1)
                if accept:
                    y_mid = y1 + k.matmul(dtf * c_mid)
                    coeffs = _interp_fit(y1, y_new, y_mid, k[..., 0], k[..., -1], dtf)
                    t0, t1 = t1, t1 + dt
                    y1, f1 = y_new, k[..., -1]
                # _optimal_step_size (misc.py:94-103)
                if ratio == 0:
                    dt = dt * 10
                else:
                    dfactor = 1.0 if ratio < 1 else 0.2
                    factor = min(10.0, max(0.9 / float(ratio) ** 0.2, dfactor))
                    dt = dt * factor
                n_steps += 1
            out.append(_interp_eval(coeffs, t0, t1, t[i]))
        return torch.stack(out)


def _interp_fit(y0, y1, ym, f0, f1, dt):   # interp.py:1-22
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * ym
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * ym
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * ym
    return [y0, dt * f0, c, b, a]


def _interp_eval(co, t0, t1, t):   # interp.py:25-47
    x = ((t - t0) / (t1 - t0)).to(co[0].dtype)
    tot, xp = co[0] + x * co[1], x
    for c in co[2:]:
        xp = xp * x
        tot = tot + xp * c
    return tot


def odeint(net, y0, t, rtol=1e-7, atol=1e-9):
    """batched forward solve, shared control; returns (sol [T, *y0.shape], nfe)"""
    shape = y0.shape
    solver = Dopri5(lambda tt, y: net(tt, y.view(shape)).reshape(-1), y0.reshape(-1), rtol, atol)
    with torch.no_grad():
        sol = solver.integrate(t)
    return sol.view((len(t),) + tuple(shape)), solver.nfe


def adjoint_backward(net, t, sol, grad_sol, rtol=1e-7, atol=1e-9):
    """OdeintAdjointMethod.backward (adjoint.py:32-162): returns (dL/dy0, [dL/dparam ...] in parameters() order, nfe)"""
    params = tuple(p.requires_grad_(True) for p in net.params())
    shape = sol.shape[1:]
    ny = sol[0].numel()
    npar = sum(p.numel() for p in params)
    norm = _mixed_norm([1, ny, ny, npar])
    nfe = 0

    def aug(tt, s):   # adjoint.py:94-127 on the flat state [vjp_t, y, adj_y, adj_params]
        y = s[1:1 + ny].view(shape)
        a = s[1 + ny:1 + 2 * ny].view(shape)
        with torch.enable_grad():
            y = y.detach().requires_grad_(True)
            f = net(tt, y)
            vj = torch.autograd.grad(f, (y,) + params, -a, allow_unused=True)
        vj = [torch.zeros_like(x) if v is None else v for v, x in zip(vj, (y,) + params)]
        return torch.cat([torch.zeros(1, dtype=s.dtype), f.detach().reshape(-1), vj[0].reshape(-1)] +
                         [v.reshape(-1) for v in vj[1:]])

    state = torch.cat([torch.zeros(1), sol[-1].reshape(-1), grad_sol[-1].reshape(-1), torch.zeros(npar)])
    for i in range(len(t) - 1, 0, -1):
        # t[i-1:i+1].flip(0) is decreasing: torchdiffeq integrates y(-t) forwards (misc.py:210-221)
        solver = Dopri5(lambda tt, s: -aug(-tt, s), state, rtol, atol, norm=norm)
        with torch.no_grad():
            state = solver.integrate(-t[i - 1:i + 1].flip(0).double())[1]
        nfe += solver.nfe
        state = state.clone()
        state[1:1 + ny] = sol[i - 1].reshape(-1)
        state[1 + ny:1 + 2 * ny] += grad_sol[i - 1].reshape(-1)
    gy = state[1 + ny:1 + 2 * ny].view(shape)
    gp, off = [], 1 + 2 * ny
    for p in params:
        gp.append(state[off:off + p.numel()].view_as(p))
        off += p.numel()
    return gy, gp, nfe
