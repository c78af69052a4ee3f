"""Thin torch <-> C-ABI glue: device pointers, the current HIP stream, caller-owned workspaces.

PyTorch here is plumbing (device memory, streams, autograd bookkeeping); all arithmetic of the
path happens in libphoenix_hip.so."""
import ctypes as C
import os

import torch

from . import _lib

_ws_cache = {}


def _require_gpu(x, name):
    if not x.is_cuda:
        raise RuntimeError("phoenix_amd: `%s` must live on the GPU (cuda/HIP device); this engine has no CPU path"
                           % name)
    if x.is_floating_point() and x.dtype != torch.float32:
        raise TypeError("phoenix_amd: `%s` must be float32 (got %s); the engine computes in float32 like the "
                        "reference's configs" % (name, x.dtype))


# The host side of a step is a few hundred microseconds of Python, which is what a small-batch step costs on a slow host
# (17 trajectories at the breast-cancer shape: 0.32 ms of kernels): the current stream and the plan switches are read
# through the cheapest route there is (torch.cuda.current_stream() builds a Stream object through four Python frames, 8 us;
# os.environ.get encodes and decodes, 0.8 us per switch, and a solve reads all of them several times).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_raw(device_index=None):
    """handle (an int) of the current HIP stream of `device_index` (default: the current device)"""
    if _raw_stream is None or _raw_device is None:
        return torch.cuda.current_stream(device_index).cuda_stream
    return _raw_stream(_raw_device() if device_index is None else device_index)


def _stream_ptr():
    return C.c_void_p(_stream_raw())


_ws_bytes = {}
_PLAN_ENV = ("PHX_ENGINE", "PHX_ADJ", "PHX_ADJ2_NP", "PHX_V1_MAXNW", "PHX_PGRAD", "PHX_EVAL_NBC", "PHX_PGRAD_KS", "PHX_FWD", "PHX_V3_NB",
             "PHX_V3_HALF", "PHX_BATCH_MIN_ROWS", "PHX_PGRAD_WGS", "PHX_PGRAD_G4", "PHX_BATCH_CHUNK_MIN", "PHX_V3C", "PHX_V3C_NB",
             "PHX_V3C_TPW", "PHX_V3C_RES", "PHX_V3C_SLOTS", "PHX_V3C_HB", "PHX_V3C_NTG", "PHX_V3C_SPLIT")


_PLAN_ENV_B = tuple(os.fsencode(k) for k in _PLAN_ENV)
_env_data = getattr(os.environ, "_data", None)


def _plan_env():
    """the values of the plan switches (part of every plan / workspace key: the tests and tools flip them between calls)"""
    if isinstance(_env_data, dict) and os.name == "posix":
        return tuple(map(_env_data.get, _PLAN_ENV_B))     # bytes or None: only compared with itself
    return tuple(os.environ.get(k) for k in _PLAN_ENV)


# phx_solve_opts.ws_keep: what the previous solve on a cached workspace was (plan key), so that the next identical one can
# vouch for the workspace and the library skips the fill of its exchange buffers (the third-generation kernels alternate
# between two sets and clean the idle one themselves).  Any other use drops the entry: the next call fills as usual.
_ws_last = {}


def forget_params():
    """drops the cached engine layouts of parameter tensors (`params_cached`): for a caller that changed parameters behind
    PyTorch's version counters -- a replayed graph that contains the optimizer step does"""
    _params_cache.clear()


def forget_workspaces():
    """for a caller that writes into the cached workspaces itself (the tests poison them): the next solves zero-fill
    their exchange buffers again instead of vouching for what the workspace holds"""
    _ws_last.clear()


def _workspace(op, N, H, B, T, device, calls=1):
    # the size query re-plans the launch on the host: remembered per shape (and per diagnostic switch setting)
    key = (op, N, H, B, T, calls) + _plan_env()
    nbytes = _ws_bytes.get(key)
    if nbytes is None:
        if calls > 1:
            nbytes = _lib.load().phx_odeint_calls_workspace_bytes(N, H, B, T, calls)
            if nbytes == 0:
                raise ValueError("a batch of %d calls x %d trajectories cannot be planned for N=%d, H=%d" %
                                 (calls, B // calls, N, H))
        else:
            nbytes = _lib.load().phx_workspace_bytes(op, N, H, B, T)
        _ws_bytes[key] = nbytes
    key = (device.index, _stream_raw(device.index), op)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        _ws_last.pop(key, None)
    return buf, nbytes


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Params:
    """Device-side view of ODENet parameters in the engine layout (include/phoenix_hip.h)."""

    def __init__(self, Ws, bs, Wp, bp, Wa, g):
        for n, x in (("Ws", Ws), ("bs", bs), ("Wp", Wp), ("bp", bp), ("Wa", Wa), ("g", g)):
            _require_gpu(x, n)
            if x.dtype != torch.float32:
                raise TypeError("phoenix_amd: parameter %s must be float32 (got %s)" % (n, x.dtype))
        self.H, self.N = Ws.shape
        assert Wp.shape == (self.H, self.N) and Wa.shape == (self.N, 2 * self.H), "not a PHOENIX ODENet"
        self.Ws = Ws.detach().contiguous()
        self.bs = bs.detach().contiguous()
        self.Wp = Wp.detach().contiguous()
        self.bp = bp.detach().contiguous()
        self.g = g.detach().reshape(-1).contiguous()
        self.device = Ws.device
        # The engine layout of this parameter version in ONE kernel (phx_layout_params, ABI 6), straight from
        # net_alpha_combine.linear_out.weight as PyTorch stores it ([N, 2H]): the gene-contiguous transpose WaT [2H, N]
        # and the packed LDS weight images (phx_params.wimg) that the forward and the backward solve of a step share.
        wa = Wa.detach().contiguous()
        self.WaT = torch.empty((2 * self.H, self.N), dtype=torch.float32, device=self.device)
        nbytes = _lib.load().phx_weight_image_bytes(self.N, self.H)
        self.wimg = torch.empty(nbytes, dtype=torch.uint8, device=self.device) if nbytes else None
        _check_call(_lib.load().phx_layout_params(_p(self.Ws), _p(self.Wp), _p(wa), _p(self.g), self.N, self.H,
                                                  _p(self.WaT), _p(self.wimg), _stream_ptr()))
        self.c = _lib.PhxParams(_p(self.Ws), _p(self.bs), _p(self.Wp), _p(self.bp), _p(self.WaT), _p(self.g),
                                self.N, self.H, self.wimg.data_ptr() if nbytes else None)
        # ready event, unconditionally: the contiguous / transposed copies above are written on this stream too
        self._ready_stream = torch.cuda.current_stream()
        self._ready_raw = self._ready_stream.cuda_stream
        self._ready_event = torch.cuda.Event()
        self._ready_event.record()

    def on_current_stream(self):
        """a call on another stream than the one that laid these parameters out waits for the layout (copies and
        packed images) and keeps their memory alive for that stream"""
        if _stream_raw(self.device.index) == self._ready_raw:
            return self
        cur = torch.cuda.current_stream()
        if cur != self._ready_stream:
            cur.wait_event(self._ready_event)
            for x in (self.Ws, self.bs, self.Wp, self.bp, self.WaT, self.g, self.wimg):
                if x is not None:
                    x.record_stream(cur)
        return self

    def new_grads(self):
        return Grads(self)


_params_cache = {}


def params_cached(Ws, bs, Wp, bp, Wa, g):
    """`Params` of these six tensors, rebuilt only when one of them changed (storage or in-place version): the engine
    layout needs a transposed copy of Wa (7 MB at breast scale), which a forward/backward pair, a validation loop or an
    analysis scan over fixed weights should not redo per call.  An optimizer step bumps the versions."""
    key = tuple((x.data_ptr(), x._version, tuple(x.shape)) for x in (Ws, bs, Wp, bp, Wa, g))
    dkey = (Ws.device.index, Ws.data_ptr())
    hit = _params_cache.get(dkey)
    if hit is not None and hit[0] == key:
        return hit[1]
    p = Params(Ws, bs, Wp, bp, Wa, g)
    if len(_params_cache) > 8:
        _params_cache.clear()
    _params_cache[dkey] = (key, p)
    return p


def invalidate_params():
    """forget the cached engine-layout copies of parameters.  Needed only after writing a parameter through a view that
    does not bump its version counter (`p.data.copy_(...)`, a raw pointer): optimizers, `copy_`, `add_` under
    `no_grad`, `load_state_dict` all bump it and need nothing."""
    _params_cache.clear()


class Grads:
    """One flat buffer carved into the six gradient tensors, every one in the REFERENCE's layout (dWa as [N, 2H],
    phx_grads.Wa): what the engine writes is what autograd hands the optimizer, no transposed copy in between.  It is
    NOT zero-filled: the engine call it is handed to writes every element (`phx_grads.overwrite`)."""

    def __init__(self, p):
        H, N = p.H, p.N
        sizes = (H * N, H, H * N, H, 2 * H * N, N)
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=p.device)
        parts = torch.split(self.flat, sizes)
        self.Ws, self.bs, self.Wp, self.bp = parts[0].view(H, N), parts[1], parts[2].view(H, N), parts[3]
        self.Wa, self.g = parts[4].view(N, 2 * H), parts[5]
        self.c = _lib.PhxGrads(_p(self.Ws), _p(self.bs), _p(self.Wp), _p(self.bp), None, _p(self.g), 1, _p(self.Wa))

    def as_reference_layout(self, g_shape):
        """(Ws, bs, Wp, bp, Wa[N,2H], g[1,N]) gradients in the reference's parameter layouts"""
        return self.Ws, self.bs, self.Wp, self.bp, self.Wa, self.g.reshape(g_shape)


def _check_call(rc):
    if rc != 0:
        raise RuntimeError("phoenix_amd: %s" % _lib.load().phx_status_string(rc).decode())


def _raise(status, worst):
    # A launch that ended in an error (above all PHX_ERR_SYNC_TIMEOUT, whose 5 s bail-out leaves the workspace header --
    # exchange-set parity, `started` counters -- in an undefined state) must not be vouched for: the next solve on every
    # cached workspace fills its exchange buffers again (phx_solve_opts.ws_keep = 0).
    forget_workspaces()
    bad = int((status != 0).nonzero()[0].item())
    msg = "%s (trajectory %d)" % (_lib.STATUS_TEXT.get(worst, "status %d" % worst), bad)
    if worst in (1, 2, 3, 4):
        raise AssertionError(msg)
    raise RuntimeError("phoenix_amd: " + msg)


_status_mode = "immediate"
_pending = []          # (event, pinned status copy [L, B]) of solves whose status has not been read yet
captured_status = []   # "captured" mode: the device status blocks of the solves issued since the mode was set


def set_status_mode(mode):
    """"immediate" (default): a training step reads the solver status of its two launches in one blocking round trip at
    the end of backward(), as the reference raises its asserts synchronously.  "deferred": the status block is copied to
    pinned host memory behind the backward kernel and checked at a later engine call (or `check_pending_status(True)`)
    once its event has completed -- the same AssertionError, raised up to two engine calls late, and the host no longer
    idles the GPU between steps.  "captured": nothing is read at all -- every status block stays on the device and is
    appended to `captured_status`; for steps recorded into a HIP graph (phoenix_amd.graphs.GraphedStep), whose caller
    reads them after a replay (`check_captured_status`)."""
    global _status_mode
    assert mode in ("immediate", "deferred", "captured")
    if mode != "deferred":
        check_pending_status(wait=True)
    _status_mode = mode


def status_mode():
    return _status_mode


_free_host = {}        # (shape, dtype) -> pinned buffers of checked solves, reused (no pinned allocation in steady state)


def defer_status(status):
    """queue a device status block [L, B] (or [B]) for a later check"""
    if _status_mode == "captured":
        captured_status.append(status)
        return
    key = (tuple(status.shape), status.dtype)
    pool = _free_host.setdefault(key, [])
    host = pool.pop() if pool else torch.empty(status.shape, dtype=status.dtype, pin_memory=True)
    host.copy_(status, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _pending.append((ev, host))


def check_pending_status(wait=False):
    """raise for every queued solve that has finished (all of them with wait=True)"""
    if _status_mode == "captured":
        return                   # (event queries and synchronisation have no place inside a capture)
    while _pending:
        ev, host = _pending[0]
        if not wait and not ev.query():
            return
        if wait:
            ev.synchronize()
        _pending.pop(0)
        pool = _free_host.setdefault((tuple(host.shape), host.dtype), [])
        if len(pool) < 8:
            pool.append(host)
        raise_for_status(host)


def check_captured_status(blocks=None):
    """reads the status blocks a captured step left on the device (default: `captured_status`) and raises like
    `raise_for_status`; one device->host read per block, outside any capture"""
    for st in (captured_status if blocks is None else blocks):
        raise_for_status(st, _force=True)


def raise_for_status(status, _force=False):
    """Maps per-trajectory solver status onto the reference's exceptions (rk_common.py:154,175-176,
    misc.py:114-115).  One device->host read.  `status` is [B], or [L, B] for L consecutive launches (the forward
    and backward solve of a training step share one stats block): the earliest failing launch is reported.
    In "captured" status mode (a step being recorded into a graph, or one of its warm-up runs) the block is kept for
    `check_captured_status` instead."""
    if _status_mode == "captured" and status.is_cuda and not _force:
        captured_status.append(status)
        return
    if status.dim() == 1:
        status = status.unsqueeze(0)
    if not status.is_cuda and not bool(status.any()):
        return
    for launch, worst in enumerate(status.amax(dim=1).tolist()):
        if worst != 0:
            _raise(status[launch], int(worst))


def rhs_forward(p, y, prior_only=False):
    _require_gpu(y, "y")
    y2 = y.detach().reshape(-1, p.N).contiguous()
    out = torch.empty_like(y2)
    B = y2.shape[0]
    p.on_current_stream()
    ws, nb = _workspace(_lib.OP_RHS_FORWARD, p.N, p.H, B, 0, y.device)
    _check_call(_lib.load().phx_rhs_forward(C.byref(p.c), _p(y2), _p(out), B, int(prior_only), _p(ws), nb,
                                            _stream_ptr()))
    return out.reshape(y.shape)


def rhs_vjp(p, y, cot, prior_only=False, want_grads=True, want_vjp_y=True, f_out=None):
    """phx_rhs_vjp on a batch; `f_out` (optional, a contiguous float32 tensor shaped like `y`) also receives f(y) --
    the C entry point's `f_out` argument."""
    _require_gpu(y, "y")
    _require_gpu(cot, "cot")
    y2 = y.detach().reshape(-1, p.N).contiguous()
    c2 = cot.detach().reshape(-1, p.N).contiguous()
    B = y2.shape[0]
    vjp = torch.empty_like(y2) if want_vjp_y else None
    grads = p.new_grads() if want_grads else None
    p.on_current_stream()
    ws, nb = _workspace(_lib.OP_RHS_VJP, p.N, p.H, B, 0, y.device)
    if f_out is not None:
        _require_gpu(f_out, "f_out")
        if f_out.dtype != torch.float32 or not f_out.is_contiguous() or f_out.numel() != y2.numel():
            raise ValueError("f_out must be a contiguous float32 tensor with as many elements as y")
    _check_call(_lib.load().phx_rhs_vjp(C.byref(p.c), _p(y2), _p(c2), _p(vjp), C.byref(grads.c) if grads else None,
                                        _p(f_out) if f_out is not None else C.c_void_p(0), B, int(prior_only), _p(ws), nb,
                                        _stream_ptr()))
    return (vjp.reshape(y.shape) if want_vjp_y else None), grads


PRIOR_MSE_MIN_ROWS = 1024   # the fused loss head is for the large prior batches (train_insilico.py:134); smaller ones take the plain formula


def prior_mse(p, X, target, keep_hidden=False):
    """fused mean((prior_only_forward(X) - target)^2) and its cotangent; returns (loss [1], cot like X) -- with
    `keep_hidden` (loss, cot, z): z = what `prior_vjp_saved` needs of the hidden layer (since ABI 7 the four sections
    du | dv | z_u | z_p of every row, formed with THIS cotangent), or None where the library has no such path (H > 128) -- or None when the engine cannot plan the batch chain for this shape
    (caller falls back to the unfused formula)."""
    _require_gpu(X, "X")
    _require_gpu(target, "target")
    x2 = X.detach().reshape(-1, p.N).contiguous()
    t2 = target.detach().reshape(-1, p.N).contiguous()
    B = x2.shape[0]
    if B < PRIOR_MSE_MIN_ROWS or t2.shape != x2.shape:
        return None
    cot = torch.empty_like(x2)
    loss = torch.empty(1, dtype=torch.float32, device=x2.device)
    p.on_current_stream()
    ws, nb = _workspace(_lib.OP_RHS_FORWARD, p.N, p.H, B, 0, X.device)
    zbytes = _lib.load().phx_prior_z_bytes(p.N, p.H, B) if keep_hidden else 0
    if zbytes:
        z = torch.empty(zbytes // 4, dtype=torch.float32, device=x2.device)
        rc = _lib.load().phx_prior_mse_save(C.byref(p.c), _p(x2), _p(t2), B, _p(cot), _p(loss), _p(z), _p(ws), nb,
                                            _stream_ptr())
    else:
        z = None
        rc = _lib.load().phx_prior_mse(C.byref(p.c), _p(x2), _p(t2), B, _p(cot), _p(loss), _p(ws), nb, _stream_ptr())
    if rc == 4:
        return None
    _check_call(rc)
    return (loss, cot, z) if keep_hidden else (loss, cot)


def prior_vjp_saved(p, X, cot, z):
    """parameter gradients of sum(cot * prior_only_forward(X)) from the hidden sections `z` that `prior_mse(...,
    keep_hidden=True)` kept (phx_prior_vjp_saved): what rhs_vjp(prior_only=True, want_vjp_y=False) returns, with the
    gradient contraction alone.  `cot` must be the cotangent that same call returned (z holds du, dv formed with it)."""
    _require_gpu(X, "X")
    _require_gpu(cot, "cot")
    x2 = X.detach().reshape(-1, p.N).contiguous()
    c2 = cot.detach().reshape(-1, p.N).contiguous()
    B = x2.shape[0]
    # `z` must be the buffer prior_mse(keep_hidden=True) filled for THIS batch size: the kernels index it by tile
    zbytes = _lib.load().phx_prior_z_bytes(p.N, p.H, B)
    if (z is None or not z.is_cuda or z.dtype != torch.float32 or not z.is_contiguous() or zbytes == 0 or
            z.numel() * 4 != zbytes):
        raise ValueError("prior_vjp_saved: `z` is not the hidden-row buffer of prior_mse(keep_hidden=True) for a batch "
                         "of %d rows (N=%d, H=%d)" % (B, p.N, p.H))
    grads = p.new_grads()
    p.on_current_stream()
    ws, nb = _workspace(_lib.OP_RHS_VJP, p.N, p.H, B, 0, X.device)
    _check_call(_lib.load().phx_prior_vjp_saved(C.byref(p.c), _p(x2), _p(c2), _p(z), C.byref(grads.c), B, _p(ws), nb,
                                                _stream_ptr()))
    return grads


def _opts(method, control, rtol, atol, t_per_sample, t_is_f32, max_num_steps, calls=1, ws_keep=0):
    return _lib.PhxSolveOpts(_lib.METHODS[method], control, float(rtol), float(atol), int(t_per_sample),
                             int(t_is_f32), int(max_num_steps), int(calls), int(ws_keep))


def _solve_call(op, pkey, device, call):
    """runs `call(ws_keep)` (the C entry point) and keeps the workspace bookkeeping of ws_keep"""
    wkey = (device.index, _stream_raw(device.index), op)
    keep = 1 if _ws_last.get(wkey) == pkey else 0
    _ws_last.pop(wkey, None)              # whatever happens below, the workspace is no longer in its previous state
    _check_call(call(keep))
    _ws_last[wkey] = pkey


def solve_forward(p, y0, t64, method, control, rtol, atol, t_per_sample, t_is_f32, max_num_steps=0, stats=None,
                  calls=1):
    """y0 [B,N] f32, t64 [T] or [B,T] f64 (device) -> sol [T,B,N], status[B], nfe[B], nsteps[B].
    The engine writes NaN into the outputs a failed trajectory never reached (a backward solve launched before the
    status is read then stops at once).  calls > 1 (shared control): the rows are `calls` independent odeint calls
    of B/calls rows each, every call with its own step controller (include/phoenix_hip.h, phx_solve_opts.calls)."""
    B, N = y0.shape
    if calls > 1 and (control != _lib.CTRL_SHARED or B % calls):
        raise ValueError("calls > 1 needs shared step control and B divisible by calls")
    T = t64.shape[-1]
    sol = torch.empty((T, B, N), dtype=torch.float32, device=y0.device)
    if stats is None:      # not zero-filled: every solve kernel writes status / nfe / nsteps of every trajectory
        stats = torch.empty((3, B), dtype=torch.int32, device=y0.device)
    p.on_current_stream()
    ws, nb = _workspace(_lib.OP_ODEINT, p.N, p.H, B, T, y0.device, calls)
    pkey = (p.N, p.H, B, T, method, control, int(t_per_sample), calls) + _plan_env()

    def call(keep):
        o = _opts(method, control, rtol, atol, t_per_sample, t_is_f32, max_num_steps, calls, keep)
        return _lib.load().phx_odeint(C.byref(p.c), _p(y0), _p(t64), B, T, C.byref(o), _p(sol), _p(stats[0]),
                                      _p(stats[1]), _p(stats[2]), _p(ws), nb, _stream_ptr())

    _solve_call(_lib.OP_ODEINT, pkey, y0.device, call)
    return sol, stats[0], stats[1], stats[2]


def solve_adjoint(p, t64, y_saved, grad_y, method, control, rtol, atol, t_per_sample, t_is_f32, want_grads=True,
                  max_num_steps=0, stats=None):
    """y_saved, grad_y [T,B,N] -> adj_y0 [B,N], Grads, status, nfe, nsteps.  `stats`: an int32 [3,B] to use (rows contiguous)."""
    T, B, N = y_saved.shape
    adj = torch.empty((B, N), dtype=torch.float32, device=y_saved.device)
    if stats is None:
        stats = torch.empty((3, B), dtype=torch.int32, device=y_saved.device)
    grads = p.new_grads() if want_grads else None
    p.on_current_stream()
    ws, nb = _workspace(_lib.OP_ADJOINT, p.N, p.H, B, T, y_saved.device)
    pkey = (p.N, p.H, B, T, method, control, int(t_per_sample), bool(want_grads)) + _plan_env()

    def call(keep):
        o = _opts(method, control, rtol, atol, t_per_sample, t_is_f32, max_num_steps, 1, keep)
        return _lib.load().phx_odeint_adjoint_backward(
            C.byref(p.c), _p(t64), B, T, C.byref(o), _p(y_saved), _p(grad_y), _p(adj),
            C.byref(grads.c) if grads else None, _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(ws), nb, _stream_ptr())

    _solve_call(_lib.OP_ADJOINT, pkey, y_saved.device, call)
    return adj, grads, stats[0], stats[1], stats[2]
