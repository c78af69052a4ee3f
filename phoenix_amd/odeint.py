"""Drop-in `odeint` / `odeint_adjoint` for PHOENIX's ODENet on MI355X.

Mirrors the reference surface (vendored torchdiffeq 0.1.1):
    odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None)          _impl/odeint.py:30
    odeint_adjoint(func, y0, t, rtol, atol, method, options, adjoint_rtol, adjoint_atol,
                   adjoint_method, adjoint_options, adjoint_params)                 _impl/adjoint.py:165
Same argument meaning, output shape `[len(t), *y0.shape]`, default method 'dopri5', the same
exceptions (ValueError for an unknown method, AssertionError for the solver asserts).

Extension (what the reference's training loop *means*, train_insilico.py:128-130): `t` may be 2-D
`[B, T]` -- one time grid per sample of `y0 [B, 1, N]` -- which integrates all B samples in ONE launch
with an independent step controller per sample (`odeint_per_sample` is the explicit spelling)."""
import warnings

import torch

from . import _lib, engine, generic
from .odenet import params_of

# torchdiffeq's registry (odeint.py:14-27); the engine implements the ones PHOENIX's configs use.
SOLVERS = ("dopri8", "dopri5", "bosh3", "adaptive_heun", "euler", "midpoint", "rk4", "explicit_adams",
           "implicit_adams", "fixed_adams")
_ENGINE_METHODS = ("dopri5", "euler", "midpoint", "rk4")
_KNOWN_UNSUPPORTED_OPTIONS = ("step_size", "grid_constructor", "grid_points", "eps", "first_step", "safety",
                              "ifactor", "dfactor", "norm", "dtype")


def _check_inputs(func, y0, t, rtol, atol, method, options):
    """misc.py:165-241, restricted to what the engine supports."""
    if not torch.is_tensor(y0):
        raise NotImplementedError("phoenix_amd: tuple states are not supported (PHOENIX passes a single tensor)")
    if not torch.is_floating_point(y0):
        raise TypeError("`y0` must be a floating point Tensor but is a {}".format(y0.type()))
    if y0.dtype != torch.float32:
        raise NotImplementedError("phoenix_amd: the engine computes in float32 like the reference's configs; "
                                  "got y0 dtype %s" % y0.dtype)
    options = {} if options is None else dict(options)
    if method is None:
        method = "dopri5"
    if method not in SOLVERS:
        raise ValueError('Invalid method "{}". Must be one of {}'.format(
            method, '{"' + '", "'.join(SOLVERS) + '"}.'))
    if method not in _ENGINE_METHODS:
        raise NotImplementedError("phoenix_amd: method '%s' is registered in torchdiffeq but not on PHOENIX's "
                                  "path (configs use dopri5; rk4/euler/midpoint are implemented)" % method)
    assert torch.is_tensor(t), "t must be a torch.Tensor"
    assert t.ndimension() in (1, 2), "t must be one dimensional"
    if not torch.is_floating_point(t):
        raise TypeError("`t` must be a floating point Tensor but is a {}".format(t.type()))
    for k in list(options):
        if k in _KNOWN_UNSUPPORTED_OPTIONS:
            raise NotImplementedError("phoenix_amd: solver option '%s' is not supported by the fused stepper" % k)
        if k not in ("max_num_steps", "batch_control"):
            warnings.warn("phoenix_amd: Unexpected arguments {}".format({k: options.pop(k)}))
    if torch.is_tensor(rtol) or torch.is_tensor(atol):
        rtol, atol = float(rtol), float(atol)
    return y0, t, float(rtol), float(atol), method, options


def _is_odenet(func):
    try:
        params_of(func)
        return True
    except TypeError:
        return False


def _prepare(func, y0, t, options):
    ws, bs, wp, bp, wa, g = params_of(func)
    N = ws.shape[1]
    if y0.shape[-1] != N:
        raise ValueError("phoenix_amd: y0 last dimension %d != number of genes %d" % (y0.shape[-1], N))
    engine._require_gpu(y0, "y0")
    y2 = y0.reshape(-1, N)
    B = y2.shape[0]
    if t.device != y0.device:
        warnings.warn("t is not on the same device as y0. Coercing to y0.device.")   # misc.py:232-235
    if t.dtype == torch.float32:     # the engine reads a float32 grid as it is (time arithmetic stays fp64 inside)
        t_is_f32 = 2
        t64 = t.detach().to(device=y0.device).contiguous()
    else:
        t_is_f32 = 0
        t64 = t.detach().to(device=y0.device, dtype=torch.float64).contiguous()
    per_sample = t.ndimension() == 2
    if per_sample and t.shape[0] != B:
        raise ValueError("phoenix_amd: per-sample t must be [B, T] with B = %d trajectories" % B)
    ctl = options.get("batch_control", "per_trajectory" if per_sample else "shared")
    if ctl not in ("shared", "per_trajectory"):
        raise ValueError("batch_control must be 'shared' or 'per_trajectory'")
    if per_sample and ctl == "shared":
        raise ValueError("per-sample time grids need batch_control='per_trajectory'")
    control = _lib.CTRL_SHARED if ctl == "shared" else _lib.CTRL_PER_TRAJECTORY
    return (ws, bs, wp, bp, wa, g), y2, t64, B, N, per_sample, t_is_f32, control


def _out_shape(sol, y0, per_sample):
    # [T, B, N] -> [T, *y0.shape]
    return sol.reshape((sol.shape[0],) + tuple(y0.shape))


class _OdeintAdjointFn(torch.autograd.Function):
    """OdeintAdjointMethod (adjoint.py:9-162): forward = no-grad solve, saves (t, y, params);
    backward = reverse-time augmented solve on the engine."""

    @staticmethod
    def forward(ctx, y2, t64, cfg, ws, bs, wp, bp, wa, g):
        (method, control, rtol, atol, per_sample, t_is_f32, max_steps, adj, defer) = cfg
        engine.check_pending_status()
        p = engine.params_cached(ws, bs, wp, bp, wa, g)
        # one stats block for both launches of the step, [status | nfe | nsteps][launch: forward, backward][B]: the two
        # status rows are contiguous (one copy to the host), and nothing is zero-filled -- every solve kernel writes the
        # three entries of every trajectory of its launch
        stats = torch.empty((3, 2 if defer else 1, y2.shape[0]), dtype=torch.int32, device=y2.device)
        ctx.set_materialize_grads(False)      # no zero tensor for the (non-differentiable) nfe output in backward
        sol, status, nfe, nsteps = engine.solve_forward(p, y2.detach().contiguous(), t64, method, control, rtol,
                                                        atol, per_sample, t_is_f32, max_steps, stats=stats[:, 0])
        # The reference raises the solver's AssertionErrors synchronously.  When a backward pass is coming
        # (some input requires grad) the forward status is read together with the backward solve's status, in
        # ONE host<->device round trip per training step after both launches are queued: same exception (the
        # forward's first), raised from backward().  Outputs a failed forward never reached are NaN, which stops
        # the backward solve at once.  Without autograd (validation, analysis callers) the check is immediate.
        ctx.phx_stats = stats if defer else None
        ctx.phx_fwd_status = status if defer else None
        if ctx.phx_fwd_status is None:
            engine.raise_for_status(status)
        ctx.cfg = cfg
        ctx.phx_params = p   # engine-layout views (incl. the transposed Wa copy) reused by backward
        ctx.phx_versions = tuple(x._version for x in (ws, bs, wp, bp, wa, g))
        ctx.save_for_backward(t64, sol, ws, bs, wp, bp, wa, g)
        ctx.mark_non_differentiable(nfe)
        return sol, nfe

    @staticmethod
    def backward(ctx, grad_sol, _grad_nfe):
        t64, sol, ws, bs, wp, bp, wa, g = ctx.saved_tensors
        if grad_sol is None:
            grad_sol = torch.zeros_like(sol)
        (method, control, rtol, atol, per_sample, t_is_f32, max_steps, adj, _defer) = ctx.cfg
        a_method, a_rtol, a_atol, via_odeint = adj
        if via_odeint and a_method != "dopri5":
            raise NotImplementedError(
                "phoenix_amd: backward through odeint(method='%s') on an ODENet would need backpropagation through the "
                "fixed-grid steps, which the fused stepper does not record (the continuous adjoint differs from it at "
                "O(1) for one unconverged step per interval); use odeint_adjoint -- the reference's training path, "
                "adjoint.py:165" % a_method)
        p = ctx.phx_params
        if ctx.phx_versions != tuple(x._version for x in (ws, bs, wp, bp, wa, g)):
            p = engine.params_cached(ws, bs, wp, bp, wa, g)   # parameters were modified in place since forward
        need_p = any(ctx.needs_input_grad[3:])
        adj_y0, grads, status, _nfe, _ns = engine.solve_adjoint(
            p, t64, sol, grad_sol.contiguous(), a_method, control, a_rtol, a_atol, per_sample, t_is_f32,
            want_grads=need_p, max_num_steps=max_steps, stats=None if ctx.phx_stats is None else ctx.phx_stats[:, 1])
        # The status read-back is the one host<->device round trip of a training step.  It is queued as an
        # end-of-backward callback of the autograd engine (the mechanism DDP finalises with): the exception still comes
        # out of loss.backward(), but the host returns the gradients and runs its accumulation nodes while the kernel
        # is still executing instead of after it.
        stats = status if ctx.phx_stats is None else ctx.phx_stats[0]          # [2, B]: forward and backward status
        if engine.status_mode() == "deferred":
            engine.defer_status(stats)      # checked at a later engine call, once the kernel has finished
        else:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda: engine.raise_for_status(stats))
            except Exception:   # noqa: BLE001  (no running graph task: check right here)
                engine.raise_for_status(stats)
        if need_p:
            gws, gbs, gwp, gbp, gwa, gg = grads.as_reference_layout(g.shape)
        else:
            gws = gbs = gwp = gbp = gwa = gg = None
        return adj_y0, None, None, gws, gbs, gwp, gbp, gwa, gg


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, return_stats=False):
    """odeint.py:30-74.  The reference's `odeint` returns an autograd-tracked solution (backpropagation through the
    solver's own operations).  The fused stepper keeps no such history, so for an ODENet:
      * nothing requires grad, autograd is off (validation, analysis callers), or `return_stats`: plain forward solve;
      * something requires grad: the call is routed through `odeint_adjoint`, so the result IS differentiable.  With
        dopri5 the continuous adjoint agrees with backpropagation through the converged solve to the solver tolerance
        (<= 2e-6 relative).  With a fixed grid (euler / midpoint / rk4: ONE step per interval,
        far from converged) the two differ at O(1), and returning the adjoint's gradient silently would not be the
        reference's `odeint`: the BACKWARD pass of such a call raises NotImplementedError (the forward result is
        exact either way; PHOENIX itself trains through `odeint_adjoint`, train_insilico.py:15-18, which is
        reproduced operation for operation)."""
    y0, t, rtol, atol, method, options = _check_inputs(func, y0, t, rtol, atol, method, options)
    if not _is_odenet(func):      # any other module: unfused torch stepper, differentiable by plain backpropagation
        engine._require_gpu(y0, "y0")          # like every other entry point: no CPU compute path in this package
        assert t.ndimension() == 1 and not return_stats, "per-sample grids / solver statistics need a PHOENIX ODENet"
        return generic.odeint(func, y0, t, rtol, atol, method, options)
    if not return_stats and torch.is_grad_enabled() and (y0.requires_grad or any(x.requires_grad for x in params_of(func))):
        return odeint_adjoint(func, y0, t, rtol=rtol, atol=atol, method=method, options=options, _via_odeint=True)
    params, y2, t64, B, N, per_sample, t_is_f32, control = _prepare(func, y0, t, options)
    engine.check_pending_status()
    p = engine.params_cached(*params)
    sol, status, nfe, nsteps = engine.solve_forward(p, y2.detach().contiguous(), t64, method, control, rtol, atol,
                                                    per_sample, t_is_f32, int(options.get("max_num_steps", 0)))
    engine.raise_for_status(status)
    out = _out_shape(sol, y0, per_sample)
    return (out, nfe, nsteps) if return_stats else out


def odeint_calls(func, y0s, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    """K forward-only `odeint(func, y0s[k], t, ...)` calls in as few launches as the device has room for:
        y0s [K, *shape] (shape = [B,1,N] or [B,N]), t [T]  ->  [K, T, *shape]  (a view of the engine's [T, K*B, N]).
    Every call keeps the reference's batch semantics -- ONE adaptive step size shared by its B trajectories, chosen
    from its own error norm -- so the result equals the K separate calls (the analysis loop of
    find_gene_influences.py:64-77 issues 2 per gene).  Falls back to K launches where the batched plan does not
    exist (shapes served by the VALU engine)."""
    K = y0s.shape[0]
    if K == 1:
        with torch.no_grad():     # forward-only by contract, like the batched launch below
            return odeint(func, y0s[0], t, rtol, atol, method, options).unsqueeze(0)
    y0, t, rtol, atol, method, options = _check_inputs(func, y0s, t, rtol, atol, method, options)
    params, y2, t64, B, N, per_sample, t_is_f32, control = _prepare(func, y0, t, options)
    if per_sample or control != _lib.CTRL_SHARED:
        raise ValueError("odeint_calls: one shared time grid, batch_control='shared'")
    engine.check_pending_status()
    p = engine.params_cached(*params)
    try:
        sol, status, _, _ = engine.solve_forward(p, y2.detach().contiguous(), t64, method, control, rtol, atol, per_sample,
                                                 t_is_f32, int(options.get("max_num_steps", 0)), calls=K)
    except ValueError:
        with torch.no_grad():
            return torch.stack([odeint(func, y0s[k], t, rtol, atol, method, options) for k in range(K)])
    engine.raise_for_status(status)
    return sol.reshape((sol.shape[0], K) + tuple(y0s.shape[1:])).transpose(0, 1)


def odeint_adjoint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, adjoint_rtol=None,
                   adjoint_atol=None, adjoint_method=None, adjoint_options=None, adjoint_params=None, _via_odeint=False):
    """adjoint.py:165-204.  Gradients flow to y0 and to the six ODENet parameters."""
    if not _is_odenet(func):      # any other module: the reference's augmented system on the unfused torch stepper
        if adjoint_options:
            raise NotImplementedError("phoenix_amd: adjoint_options are not supported")
        y0, t, rtol, atol, method, options = _check_inputs(func, y0, t, rtol, atol, method, options)
        engine._require_gpu(y0, "y0")          # like every other entry point: no CPU compute path in this package
        assert t.ndimension() == 1, "per-sample time grids need a PHOENIX ODENet"
        return generic.odeint_adjoint(func, y0, t, rtol, atol, method, options,
                                      rtol if adjoint_rtol is None else adjoint_rtol,
                                      atol if adjoint_atol is None else adjoint_atol,
                                      method if adjoint_method is None else adjoint_method, adjoint_params)
    if adjoint_params is not None:
        raise NotImplementedError("phoenix_amd: adjoint_params is fixed to ODENet's six parameters")
    if adjoint_options:
        raise NotImplementedError("phoenix_amd: adjoint_options are not supported by the fused stepper")
    if adjoint_rtol is None:
        adjoint_rtol = rtol
    if adjoint_atol is None:
        adjoint_atol = atol
    y0, t, rtol, atol, method, options = _check_inputs(func, y0, t, rtol, atol, method, options)
    if adjoint_method is None:
        adjoint_method = method
    if adjoint_method not in _ENGINE_METHODS:
        raise NotImplementedError("phoenix_amd: adjoint_method '%s' not supported" % adjoint_method)
    params, y2, t64, B, N, per_sample, t_is_f32, control = _prepare(func, y0, t, options)
    # The forward status is read together with the backward solve's only when a backward pass can come: autograd must
    # be recording AND something must require grad (needs_input_grad mirrors .requires_grad even under no_grad: the
    # reference's validation code calls odeint_adjoint inside torch.no_grad(), train_insilico.py:77-106).
    # (a plain `odeint` call routed here keeps the reference's synchronous asserts: its caller may never run backward)
    defer = torch.is_grad_enabled() and (y0.requires_grad or any(x.requires_grad for x in params)) and not _via_odeint
    cfg = (method, control, rtol, atol, per_sample, t_is_f32, int(options.get("max_num_steps", 0)),
           (adjoint_method, float(adjoint_rtol), float(adjoint_atol), bool(_via_odeint)), defer)
    sol, _nfe = _OdeintAdjointFn.apply(y2, t64, cfg, *params)
    return _out_shape(sol, y0, per_sample)


def odeint_per_sample(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None, adjoint=True):
    """The reference's per-sample python loop as ONE launch:
        for time, batch_point in zip(t, batch): odeint(odenet, batch_point, time, method)[1]
    y0 [B,1,N] (or [B,N]), t [B,T]; returns [T, *y0.shape]; independent step control per sample."""
    assert t.ndimension() == 2, "odeint_per_sample needs t of shape [B, T]"
    fn = odeint_adjoint if adjoint else odeint
    return fn(func, y0, t, rtol=rtol, atol=atol, method=method, options=options)
