"""Unfused stepper for `func` objects that are NOT a PHOENIX ODENet ("the surface is total").

PHOENIX's drivers only ever pass ODENet, which the fused HIP engine integrates; any other `func(t, y)` gets this plain
PyTorch formulation on the caller's device: the same algorithms (fixed grid euler / midpoint / 3/8-rule rk4, one step per
interval, solvers.py:77-95, fixed_grid.py, rk_common.py:96-103; dopri5 with the reference's initial step, FSAL, error
ratio, step-size rule and quartic dense output, rk_common.py:39-228, misc.py:47-103, interp.py), one batch-wide step
controller like the reference's `odeint`, every operation a torch op (so `odeint` is differentiable by plain
backpropagation, as the reference's is, and `odeint_adjoint` integrates the augmented system of adjoint.py:32-162 with
`torch.autograd.grad` supplying the vector-Jacobian products).  Nothing here is a fallback of the engine: an ODENet never
reaches this module, no engine kernel is replaced, and like every other entry point of the package it takes device
tensors only (phoenix_amd.odeint raises for host tensors before calling in).
"""
import torch

_A = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)                                   # dopri5.py:5-30
_B = ((1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9),
      (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
      (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
      (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_CSOL = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
_CERR = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
         -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0)
_CMID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
         187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms(x):
    return x.pow(2).mean().sqrt()                                              # misc.py:38-39


def _combo(y, ks, coeffs, dt):
    acc = None
    for k, c in zip(ks, coeffs):
        if c != 0.0:
            acc = k * (c * dt) if acc is None else acc + k * (c * dt)
    return y if acc is None else y + acc


def _fixed_step(func, method, t0, dt, y):
    dtf = dt.to(y.dtype)
    k1 = func(t0, y)
    if method == "euler":                                                      # fixed_grid.py:6-10
        return y + dtf * k1
    if method == "midpoint":                                                   # fixed_grid.py:13-19
        return y + dtf * func(t0 + dt / 2, y + k1 * (dtf / 2))
    k2 = func(t0 + dt / 3, y + k1 * (dtf / 3))                                 # rk4_alt_step_func, rk_common.py:96-103
    k3 = func(t0 + dt * 2 / 3, y + dtf * (k2 - k1 / 3))
    k4 = func(t0 + dt, y + dtf * (k1 - k2 + k3))
    return y + (k1 + 3 * (k2 + k3) + k4) * (dtf / 8)


def _interp(y0, y1, ym, f0, f1, dt, x):                                        # interp.py:1-47
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * ym
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * ym
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * ym
    return y0 + x * (dt * f0 + x * (c + x * (b + x * a)))


def integrate(func, y0, t, rtol, atol, method, norm=_rms, max_num_steps=2 ** 31 - 1):
    """[len(t), *y0.shape]; `t` strictly monotone (either direction), time arithmetic in t's dtype promoted to fp64."""
    t = t.to(torch.float64)
    if t.numel() > 1 and bool(t[0] > t[1]):                                    # misc.py:210-221: integrate y(-t) forwards
        f = func
        func = lambda tt, y: -f(-tt, y)                                        # noqa: E731
        t = -t
    assert bool((t[1:] > t[:-1]).all()), "t must be strictly increasing or decreasing"
    out = [y0]
    if method != "dopri5":
        y = y0
        for i in range(len(t) - 1):                                            # grid = t: one step per interval
            y = _fixed_step(func, method, t[i], t[i + 1] - t[i], y)
            out.append(y)
        return torch.stack(out)
    f0 = func(t[0], y0)
    # _select_initial_step, misc.py:47-86 (order 4)
    scale = atol + y0.abs() * rtol
    d0, d1 = norm(y0 / scale), norm(f0 / scale)
    h0 = torch.tensor(1e-6, dtype=torch.float64, device=y0.device) if bool((d0 < 1e-5) | (d1 < 1e-5)) else (0.01 * d0 / d1).double()
    f1 = func(t[0] + h0, y0 + h0.to(y0.dtype) * f0)
    d2 = norm((f1 - f0) / scale) / h0.to(y0.dtype)
    if bool((d1 <= 1e-15) & (d2 <= 1e-15)):
        h1 = torch.maximum(torch.tensor(1e-6, dtype=torch.float64, device=y0.device), h0 * 1e-3)
    else:
        h1 = ((0.01 / torch.maximum(d1, d2)) ** (1.0 / 5)).double()
    dt = torch.minimum(100 * h0, h1)
    t0 = t1 = t[0]
    y1, fy1 = y0, f0
    seg = None
    for i in range(1, len(t)):
        n_steps = 0
        while bool(t[i] > t1):                                                 # _adaptive_step, rk_common.py:150-228
            assert n_steps < max_num_steps, "max_num_steps exceeded ({}>={})".format(n_steps, max_num_steps)
            assert bool(t1 + dt > t1), "underflow in dt {}".format(float(dt))
            dtf = dt.to(y0.dtype)
            ks = [fy1]
            for al, be in zip(_A, _B):
                ks.append(func(t1 + al * dt, _combo(y1, ks, be, dtf)))
            y_new = _combo(y1, ks, _CSOL, dtf)
            err = _combo(torch.zeros_like(y1), ks, _CERR, dtf)
            tol = atol + rtol * torch.maximum(y1.abs(), y_new.abs())
            ratio = norm(err / tol)
            assert bool(torch.isfinite(y_new).all()), "non-finite values in state `y`"
            if bool(ratio <= 1):
                y_mid = _combo(y1, ks, _CMID, dtf)
                seg = (y1, y_new, y_mid, ks[0], ks[-1], dtf, t1, t1 + dt)
                t0, t1 = t1, t1 + dt
                y1, fy1 = y_new, ks[-1]
            r = float(ratio.detach())                                               # _optimal_step_size, misc.py:94-103
            if r == 0:
                dt = dt * 10
            else:
                dt = dt * min(10.0, max(0.9 / r ** 0.2, 1.0 if r < 1 else 0.2))
            n_steps += 1
        ya, yb, ym, fa, fb, h, ta, tb = seg
        out.append(_interp(ya, yb, ym, fa, fb, h, ((t[i] - ta) / (tb - ta)).to(y0.dtype)))
    return torch.stack(out)


def _mixed_norm(sizes):                                                        # misc.py:14-24, adjoint.py:72-78
    def norm(x):
        parts, off = [], 0
        for n in sizes:
            parts.append(_rms(x[off:off + n]) if n > 0 else x.new_zeros(()))
            off += n
        return torch.stack(parts).max()
    return norm


class _AdjointFn(torch.autograd.Function):
    """OdeintAdjointMethod (adjoint.py:9-162) for an arbitrary module: forward solve without a graph, backward = the
    augmented system [vjp_t, y, adj_y, adj_params] integrated interval by interval in reverse time."""

    @staticmethod
    def forward(ctx, func, y0, t, rtol, atol, method, adj, max_steps, *params):
        with torch.no_grad():
            sol = integrate(func, y0, t, rtol, atol, method, max_num_steps=max_steps)
        ctx.func, ctx.cfg, ctx.max_steps = func, adj, max_steps
        ctx.save_for_backward(t, sol, *params)
        return sol

    @staticmethod
    def backward(ctx, grad_sol):
        t, sol, *params = ctx.saved_tensors
        func = ctx.func
        rtol, atol, method = ctx.cfg
        shape, ny = sol.shape[1:], sol[0].numel()
        npar = sum(p.numel() for p in params)
        norm = _mixed_norm([1, ny, ny, npar])

        def aug(tt, s):                                                        # adjoint.py:94-127
            y = s[1:1 + ny].view(shape)
            a = s[1 + ny:1 + 2 * ny].view(shape)
            with torch.enable_grad():
                y = y.detach().requires_grad_(True)
                f = func(tt, y)
                vj = torch.autograd.grad(f, (y,) + tuple(params), -a, allow_unused=True)
            vj = [torch.zeros_like(x) if v is None else v for v, x in zip(vj, (y,) + tuple(params))]
            return torch.cat([s.new_zeros(1), f.detach().reshape(-1), vj[0].reshape(-1)] + [v.reshape(-1) for v in vj[1:]])

        with torch.no_grad():
            state = torch.cat([sol.new_zeros(1), sol[-1].reshape(-1), grad_sol[-1].reshape(-1), sol.new_zeros(npar)])
            for i in range(len(t) - 1, 0, -1):
                state = integrate(aug, state, t[i - 1:i + 1].flip(0), rtol, atol, method, norm=norm,
                                  max_num_steps=ctx.max_steps)[1].clone()
                state[1:1 + ny] = sol[i - 1].reshape(-1)
                state[1 + ny:1 + 2 * ny] += grad_sol[i - 1].reshape(-1)
        gy = state[1 + ny:1 + 2 * ny].view(shape)
        gp, off = [], 1 + 2 * ny
        for p in params:
            gp.append(state[off:off + p.numel()].view_as(p))
            off += p.numel()
        return (None, gy, None, None, None, None, None, None) + tuple(gp)


def odeint(func, y0, t, rtol, atol, method, options):
    return integrate(func, y0, t.to(y0.device), rtol, atol, method,
                     max_num_steps=int(options.get("max_num_steps", 2 ** 31 - 1)) or 2 ** 31 - 1)


def odeint_adjoint(func, y0, t, rtol, atol, method, options, adjoint_rtol, adjoint_atol, adjoint_method, adjoint_params):
    if adjoint_params is None:                                                 # find_parameters, adjoint.py:207-218
        adjoint_params = tuple(p for p in func.parameters() if p.requires_grad) if isinstance(func, torch.nn.Module) else ()
    max_steps = int(options.get("max_num_steps", 2 ** 31 - 1)) or 2 ** 31 - 1
    return _AdjointFn.apply(func, y0, t.to(y0.device), rtol, atol, method, (adjoint_rtol, adjoint_atol, adjoint_method),
                            max_steps, *adjoint_params)
