"""ctypes binding of the C ABI declared in include/phoenix_hip.h.

There is NO fallback: if libphoenix_hip.so is missing the import fails loudly (build it with
`python -m phoenix_amd.build` or `__graft_entry__.build()`), and every entry point requires a GPU."""
import ctypes as C
import os

from . import build as _build

_LIB = None


class PhxParams(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("Ws", "bs", "Wp", "bp", "WaT", "g")] + [("N", C.c_int), ("H", C.c_int),
                                                                                 ("wimg", C.c_void_p)]


class PhxGrads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("Ws", "bs", "Wp", "bp", "WaT", "g")] + [("overwrite", C.c_int), ("Wa", C.c_void_p)]


class PhxSolveOpts(C.Structure):
    _fields_ = [("method", C.c_int), ("control", C.c_int), ("rtol", C.c_double), ("atol", C.c_double),
                ("t_per_sample", C.c_int), ("t_is_f32", C.c_int), ("max_num_steps", C.c_longlong), ("calls", C.c_int),
                ("ws_keep", C.c_int)]


EXPORTS = ("phx_abi_version", "phx_status_string", "phx_device_cus", "phx_workspace_bytes", "phx_rhs_forward",
           "phx_rhs_vjp", "phx_odeint", "phx_odeint_adjoint_backward", "phx_debug_profile_region", "phx_debug_set_kernel_events",
           "phx_prior_targets", "phx_hill_rhs", "phx_hill_simulate", "phx_prior_mse", "phx_debug_adjoint_kernel",
           "phx_odeint_calls_workspace_bytes", "phx_weight_image_bytes", "phx_pack_weight_images", "phx_prior_targets_sell",
           "phx_debug_adjoint_kernel_m", "phx_prior_z_bytes", "phx_prior_mse_save", "phx_prior_vjp_saved",
           "phx_layout_params", "phx_debug_forward_kernel_m", "phx_debug_queue_kernel_events", "phx_debug_solve_launches")

OP_RHS_FORWARD, OP_RHS_VJP, OP_ODEINT, OP_ADJOINT = 0, 1, 2, 3
METHODS = {"euler": 0, "midpoint": 1, "rk4": 2, "dopri5": 3}
CTRL_SHARED, CTRL_PER_TRAJECTORY = 0, 1
STATUS_TEXT = {
    1: "max_num_steps exceeded",
    2: "underflow in dt",
    3: "non-finite values in state `y`",
    4: "bad argument (t must be strictly increasing or decreasing)",
    5: "workspace too small",
    6: "HIP launch failure",
    7: "in-kernel grid barrier timed out",
}


def lib_path():
    # PHX_LIB names a diagnostic build of the same ABI (A/B timing of two builds); it is honoured only together with
    # PHX_DIAG=1 so that a stray variable in a production environment cannot swap the engine
    if os.environ.get("PHX_DIAG") == "1" and os.environ.get("PHX_LIB"):
        return os.environ["PHX_LIB"]
    return _build.LIB


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "phoenix_amd: %s not found. The HIP engine is the only execution path (no CPU fallback); "
            "build it with `python -m phoenix_amd.build`." % path)
    lib = C.CDLL(path)
    lib.phx_status_string.restype = C.c_char_p
    lib.phx_workspace_bytes.restype = C.c_size_t
    lib.phx_workspace_bytes.argtypes = [C.c_int] * 5
    vp = C.c_void_p
    lib.phx_rhs_forward.argtypes = [C.POINTER(PhxParams), vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp]
    lib.phx_rhs_vjp.argtypes = [C.POINTER(PhxParams), vp, vp, vp, C.POINTER(PhxGrads), vp, C.c_int, C.c_int, vp,
                                C.c_size_t, vp]
    lib.phx_odeint.argtypes = [C.POINTER(PhxParams), vp, vp, C.c_int, C.c_int, C.POINTER(PhxSolveOpts), vp, vp, vp,
                               vp, vp, C.c_size_t, vp]
    lib.phx_odeint_adjoint_backward.argtypes = [C.POINTER(PhxParams), vp, C.c_int, C.c_int, C.POINTER(PhxSolveOpts),
                                                vp, vp, vp, C.POINTER(PhxGrads), vp, vp, vp, vp, C.c_size_t, vp]
    lib.phx_prior_targets.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.phx_prior_targets_sell.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.phx_prior_mse.argtypes = [C.POINTER(PhxParams), vp, vp, C.c_int, vp, vp, vp, C.c_size_t, vp]
    lib.phx_prior_z_bytes.restype = C.c_size_t
    lib.phx_prior_z_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.phx_prior_mse_save.argtypes = [C.POINTER(PhxParams), vp, vp, C.c_int, vp, vp, vp, vp, C.c_size_t, vp]
    lib.phx_prior_vjp_saved.argtypes = [C.POINTER(PhxParams), vp, vp, vp, C.POINTER(PhxGrads), C.c_int, vp, C.c_size_t, vp]
    lib.phx_hill_rhs.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.phx_hill_simulate.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_double, vp, C.c_int, C.c_int, vp]
    lib.phx_debug_set_kernel_events.argtypes = [vp, vp]
    lib.phx_debug_set_kernel_events.restype = None
    lib.phx_debug_queue_kernel_events.argtypes = [vp, vp]
    lib.phx_debug_queue_kernel_events.restype = None
    lib.phx_debug_adjoint_kernel.argtypes = [C.c_int] * 5
    lib.phx_debug_adjoint_kernel_m.argtypes = [C.c_int] * 6
    lib.phx_odeint_calls_workspace_bytes.argtypes = [C.c_int] * 5
    lib.phx_odeint_calls_workspace_bytes.restype = C.c_size_t
    lib.phx_weight_image_bytes.argtypes = [C.c_int, C.c_int]
    lib.phx_weight_image_bytes.restype = C.c_size_t
    lib.phx_pack_weight_images.argtypes = [C.POINTER(PhxParams), vp, vp]
    lib.phx_layout_params.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.phx_debug_forward_kernel_m.argtypes = [C.c_int] * 6
    lib.phx_debug_solve_launches.argtypes = [C.c_int] * 7
    assert lib.phx_abi_version() == 7
    _LIB = lib
    return lib
