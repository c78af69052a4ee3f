"""ctypes binding of the CPU oracle (oracle/libphx_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (phoenix_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libphx_oracle.so")

METHODS = {"euler": 0, "midpoint": 1, "rk4": 2, "dopri5": 3}
STATUS = {
    1: "max_num_steps exceeded",
    2: "underflow in dt",
    3: "non-finite values in state `y`",
    4: "bad argument",
}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("phx_oracle.c", "phx_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libphx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Net(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int)] + [(k, C.c_void_p) for k in ("Ws", "bs", "Wp", "bp", "Wa", "g")]


class _Grads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("Ws", "bs", "Wp", "bp", "Wa", "g")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.phxo_rms_norm.restype = C.c_float
        _lib.phxo_mixed_norm.restype = C.c_float
        _lib.phxo_optimal_step_size.restype = C.c_double
        _lib.phxo_optimal_step_size.argtypes = [C.c_double, C.c_float, C.c_double, C.c_double, C.c_double, C.c_int]
    return _lib


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Net:
    """Parameters in the reference layouts: Ws[H,N] bs[H] Wp[H,N] bp[H] Wa[N,2H] g[N]."""

    KEYS = ("Ws", "bs", "Wp", "bp", "Wa", "g")

    def __init__(self, Ws, bs, Wp, bp, Wa, g):
        self.Ws, self.bs, self.Wp, self.bp, self.Wa = _f32(Ws), _f32(bs), _f32(Wp), _f32(bp), _f32(Wa)
        self.g = _f32(g).reshape(-1)
        self.H, self.N = self.Ws.shape
        assert self.Wp.shape == (self.H, self.N) and self.Wa.shape == (self.N, 2 * self.H)
        assert self.bs.shape == (self.H,) and self.bp.shape == (self.H,) and self.g.shape == (self.N,)
        self._c = _Net(self.N, self.H, *[_ptr(getattr(self, k)) for k in self.KEYS])

    @property
    def P(self):
        return 4 * self.H * self.N + 2 * self.H + self.N

    def zero_grads(self):
        return {k: np.zeros_like(getattr(self, k)) for k in self.KEYS}

    def _cgr(self, grads):
        return _Grads(*[_ptr(grads[k]) for k in self.KEYS])


def _check(st):
    if st != 0:
        raise AssertionError(STATUS.get(st, "oracle status %d" % st))


def rhs(net, y, prior_only=False):
    y = _f32(y)
    shp = y.shape
    y2 = y.reshape(-1, net.N)
    f = np.empty_like(y2)
    _check(lib().phxo_rhs(C.byref(net._c), _ptr(y2), _ptr(f), C.c_int(y2.shape[0]), C.c_int(int(prior_only))))
    return f.reshape(shp)


def rhs_vjp(net, y, cot, prior_only=False, want_grads=True):
    y = _f32(y)
    shp = y.shape
    y2 = y.reshape(-1, net.N)
    c2 = _f32(cot).reshape(-1, net.N)
    vjp = np.empty_like(y2)
    f = np.empty_like(y2)
    grads = net.zero_grads() if want_grads else None
    cg = net._cgr(grads) if want_grads else None
    _check(lib().phxo_rhs_vjp(C.byref(net._c), _ptr(y2), _ptr(c2), C.c_int(y2.shape[0]), _ptr(vjp),
                              C.byref(cg) if want_grads else None, _ptr(f), C.c_int(int(prior_only))))
    return vjp.reshape(shp), grads, f.reshape(shp)


def _t64(t):
    t = np.asarray(t)
    is32 = int(t.dtype == np.float32)
    return np.ascontiguousarray(t.astype(np.float64)), is32


def odeint(net, y0, t, method="dopri5", rtol=1e-7, atol=1e-9, return_stats=False):
    """Reference odeint(ODENet, y0, t): one solve of the flattened state (shared step control).
    y0: [..., N]; returns [T, *y0.shape]."""
    y0 = _f32(y0)
    shp = y0.shape
    y2 = y0.reshape(-1, net.N)
    B = y2.shape[0]
    t64, is32 = _t64(t)
    T = t64.shape[0]
    sol = np.empty((T, B, net.N), np.float32)
    nfe, nst = C.c_longlong(0), C.c_longlong(0)
    _check(lib().phxo_odeint(C.byref(net._c), _ptr(y2), C.c_int(B), _ptr(t64), C.c_int(T), C.c_int(is32),
                             C.c_int(METHODS[method]), C.c_double(rtol), C.c_double(atol), _ptr(sol),
                             C.byref(nfe), C.byref(nst)))
    sol = sol.reshape((T,) + shp)
    return (sol, nfe.value, nst.value) if return_stats else sol


def adjoint_backward(net, t, y_saved, grad_y, method="dopri5", rtol=1e-7, atol=1e-9, theta_in_norm=True,
                     return_stats=False):
    """OdeintAdjointMethod.backward: y_saved, grad_y: [T, ..., N] -> (adj_y0 [..., N], grads dict)."""
    y_saved = _f32(y_saved)
    grad_y = _f32(grad_y)
    T = y_saved.shape[0]
    shp = y_saved.shape[1:]
    ys = y_saved.reshape(T, -1, net.N)
    gy = grad_y.reshape(T, -1, net.N)
    B = ys.shape[1]
    t64, is32 = _t64(t)
    adj = np.empty((B, net.N), np.float32)
    grads = net.zero_grads()
    cg = net._cgr(grads)
    nfe, nst = C.c_longlong(0), C.c_longlong(0)
    _check(lib().phxo_adjoint_backward(C.byref(net._c), C.c_int(B), _ptr(t64), C.c_int(T), C.c_int(is32),
                                       C.c_int(METHODS[method]), C.c_double(rtol), C.c_double(atol),
                                       _ptr(ys), _ptr(gy), C.c_int(int(theta_in_norm)), _ptr(adj),
                                       C.byref(cg), C.byref(nfe), C.byref(nst)))
    adj = adj.reshape(shp)
    return (adj, grads, nfe.value, nst.value) if return_stats else (adj, grads)


def odeint_per_sample(net, y0, t, method="dopri5", rtol=1e-7, atol=1e-9, nthreads=0, return_stats=False):
    """B independent solves (the reference's python loop): y0 [B,N], t [B,T] -> [B,T,N]."""
    y0 = _f32(y0).reshape(-1, net.N)
    B = y0.shape[0]
    t = np.asarray(t)
    is32 = int(t.dtype == np.float32)
    t64 = np.ascontiguousarray(np.broadcast_to(t.astype(np.float64), (B, t.shape[-1])))
    T = t64.shape[1]
    sol = np.empty((B, T, net.N), np.float32)
    nfe, nst = C.c_longlong(0), C.c_longlong(0)
    _check(lib().phxo_odeint_per_sample(C.byref(net._c), _ptr(y0), C.c_int(B), _ptr(t64), C.c_int(T),
                                        C.c_int(is32), C.c_int(METHODS[method]), C.c_double(rtol),
                                        C.c_double(atol), _ptr(sol), C.byref(nfe), C.byref(nst),
                                        C.c_int(nthreads)))
    return (sol, nfe.value, nst.value) if return_stats else sol


def adjoint_backward_per_sample(net, t, y_saved, grad_y, method="dopri5", rtol=1e-7, atol=1e-9,
                                theta_in_norm=True, nthreads=0, return_stats=False):
    """y_saved, grad_y: [B,T,N]; t [B,T] -> (adj_y0 [B,N], grads summed over samples)."""
    ys = _f32(y_saved)
    gy = _f32(grad_y)
    B, T, N = ys.shape
    t = np.asarray(t)
    is32 = int(t.dtype == np.float32)
    t64 = np.ascontiguousarray(np.broadcast_to(t.astype(np.float64), (B, T)))
    adj = np.empty((B, N), np.float32)
    grads = net.zero_grads()
    cg = net._cgr(grads)
    nfe, nst = C.c_longlong(0), C.c_longlong(0)
    _check(lib().phxo_adjoint_backward_per_sample(
        C.byref(net._c), C.c_int(B), _ptr(t64), C.c_int(T), C.c_int(is32), C.c_int(METHODS[method]),
        C.c_double(rtol), C.c_double(atol), _ptr(ys), _ptr(gy), C.c_int(int(theta_in_norm)), _ptr(adj),
        C.byref(cg), C.byref(nfe), C.byref(nst), C.c_int(nthreads)))
    return (adj, grads, nfe.value, nst.value) if return_stats else (adj, grads)


# ---- controller unit functions (golden G6) ----
def rms_norm(x):
    x = _f32(x).reshape(-1)
    return float(lib().phxo_rms_norm(_ptr(x), C.c_longlong(x.size)))


def mixed_norm(x, blocks):
    x = _f32(x).reshape(-1)
    bl = np.asarray(blocks, dtype=np.int64)
    return float(lib().phxo_mixed_norm(_ptr(x), _ptr(bl), C.c_int(len(bl))))


def optimal_step_size(last_step, error_ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    return float(lib().phxo_optimal_step_size(last_step, error_ratio, safety, ifactor, dfactor, order))


def interp_fit(y0, y1, y_mid, f0, f1, dt):
    arrs = [_f32(a).reshape(-1) for a in (y0, y1, y_mid, f0, f1)]
    n = arrs[0].size
    coef = np.empty((5, n), np.float32)
    lib().phxo_interp_fit(*[_ptr(a) for a in arrs], C.c_float(dt), C.c_longlong(n), _ptr(coef))
    return coef


def interp_eval(coef, t0, t1, t):
    coef = _f32(coef)
    n = coef.shape[1]
    out = np.empty(n, np.float32)
    lib().phxo_interp_eval(_ptr(coef), C.c_longlong(n), C.c_double(t0), C.c_double(t1), C.c_double(t), _ptr(out))
    return out
