"""ctypes binding of librqp_hip.so (include/rqp_abi.h).

This is the only module that touches the shared library.  There is NO fallback:
if the library is missing, or no HIP device is visible, the product path raises
(``RqpUnavailable``) -- it never routes to a CPU implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RQP_LIB", os.path.join(_HERE, "lib", "librqp_hip.so"))

RQP_F32, RQP_F64 = 0, 1
RQP_TILE_SAME, RQP_TILE_F16, RQP_TILE_BF16 = 0, 1, 2
KERNELS = {"auto": 0, "generic": 1, "resident": 2, "resident2": 2, "wave": 3, "mfma": 4}     # enum rqp_kernel
RQP_ERR_UNSUPPORTED = -5
FLAG_LOW_MEMORY = 1          # rqp_dims.flags
FLAG_FULL_LADDER = 2
STATUS_STR = {0: "solved", 1: "max_iters_reached", 2: "nan_detected", 3: "primal_infeasible", 4: "dual_infeasible",
              -1: "unsolved"}

# every symbol include/rqp_abi.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = (
    "rqp_default_settings", "rqp_create", "rqp_setup", "rqp_update", "rqp_update_mats", "rqp_update_affine",
    "rqp_update_settings",
    "rqp_warm_start", "rqp_clear_primal_dual", "rqp_solve", "rqp_iterate", "rqp_compute_residuals",
    "rqp_get_state", "rqp_get_rhos", "rqp_get_K", "rqp_dispatch_history", "rqp_get_dispatch", "rqp_get_window", "rqp_kernel_name", "rqp_destroy", "rqp_strerror",
    "rqp_last_error", "rqp_version",
)


class RqpUnavailable(RuntimeError):
    """librqp_hip.so cannot be loaded / no MI355X visible."""


class RqpError(RuntimeError):
    """A C-ABI call returned a negative rqp_error (``code``)."""
    code = 0


class Dims(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("m", ctypes.c_int32), ("batch", ctypes.c_int32),
                ("shared_mats", ctypes.c_int32), ("dtype", ctypes.c_int32), ("kernel", ctypes.c_int32),
                ("tile_dtype", ctypes.c_int32), ("flags", ctypes.c_int32)]


class CSettings(ctypes.Structure):
    _fields_ = [("rho", ctypes.c_double), ("rho_min", ctypes.c_double), ("rho_max", ctypes.c_double),
                ("sigma", ctypes.c_double), ("adaptive_rho_tolerance", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eq_tol", ctypes.c_double),
                ("adaptive_rho", ctypes.c_int32), ("max_iter", ctypes.c_int32),
                ("check_interval", ctypes.c_int32), ("warm_starting", ctypes.c_int32),
                ("eps_rel", ctypes.c_double), ("eps_prim_inf", ctypes.c_double), ("eps_dual_inf", ctypes.c_double),
                ("scaling", ctypes.c_int32), ("check_infeasibility", ctypes.c_int32)]


class CInfo(ctypes.Structure):
    _fields_ = [("iter", ctypes.c_void_p), ("status", ctypes.c_void_p), ("rho_ind", ctypes.c_void_p),
                ("pri_res", ctypes.c_void_p), ("dua_res", ctypes.c_void_p), ("rho_estimate", ctypes.c_void_p),
                ("obj_val", ctypes.c_void_p), ("trace", ctypes.c_void_p), ("trace_cap", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


_lib = None


def load():
    """Load the library once; raise RqpUnavailable if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RqpUnavailable(
            "librqp_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C reluqp-py_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # torch ships its own libamdhip64 (SONAME libamdhip64.so.7).  It must be mapped BEFORE our
    # library so that both bind to ONE HIP runtime (streams and pointers are shared with torch);
    # loading ours first would pull a second runtime from /opt/rocm into the process.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RqpUnavailable("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double
    H = ctypes.c_void_p  # opaque handle
    sig = {
        "rqp_default_settings": (ctypes.c_int, [ctypes.POINTER(CSettings)]),
        "rqp_create": (ctypes.c_int, [ctypes.POINTER(H), ctypes.POINTER(Dims), ctypes.POINTER(CSettings), ctypes.c_int]),
        "rqp_setup": (ctypes.c_int, [H, vp, vp, vp, vp, vp, vp]),
        "rqp_update": (ctypes.c_int, [H, vp, vp, vp, vp]),
        "rqp_update_mats": (ctypes.c_int, [H, vp, vp, vp]),
        "rqp_update_affine": (ctypes.c_int, [H, vp, ctypes.c_int32, vp, vp, vp, vp, vp]),
        "rqp_update_settings": (ctypes.c_int, [H, ctypes.POINTER(CSettings)]),
        "rqp_warm_start": (ctypes.c_int, [H, vp, vp, vp, ctypes.c_int, dbl, vp]),
        "rqp_clear_primal_dual": (ctypes.c_int, [H, vp]),
        "rqp_solve": (ctypes.c_int, [H, vp, vp, vp, ctypes.POINTER(CInfo), vp]),
        "rqp_iterate": (ctypes.c_int, [H, i32, vp]),
        "rqp_compute_residuals": (ctypes.c_int, [H, dbl, vp, vp, vp, vp, vp]),
        "rqp_get_state": (ctypes.c_int, [H, vp, vp, vp, vp, vp]),
        "rqp_get_rhos": (ctypes.c_int, [H, ctypes.POINTER(dbl), i32, ctypes.POINTER(i32)]),
        "rqp_get_K": (ctypes.c_int, [H, i32, i32, vp, vp]),
        "rqp_dispatch_history": (ctypes.c_int, [H, i32]),
        "rqp_get_dispatch": (ctypes.c_int, [H, vp, vp, ctypes.POINTER(i32), vp]),
        "rqp_get_window": (ctypes.c_int, [H, ctypes.POINTER(i32), vp, vp]),
        "rqp_kernel_name": (ctypes.c_char_p, [H]),
        "rqp_destroy": (ctypes.c_int, [H]),
        "rqp_strerror": (ctypes.c_char_p, [ctypes.c_int]),
        "rqp_last_error": (ctypes.c_char_p, [H]),
        "rqp_version": (ctypes.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(handle, rc, what):
    if rc == 0:
        return
    lib = load()
    msg = lib.rqp_strerror(rc).decode()
    detail = lib.rqp_last_error(handle).decode() if handle else ""
    err = RqpError("%s failed: %s (%d)%s" % (what, msg, rc, (": " + detail) if detail else ""))
    err.code = rc
    raise err


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
