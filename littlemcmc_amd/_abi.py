"""ctypes binding of the C ABI in include/lmc_hip.h (liblmc_hip.so).

There is no CPU fallback: if the HIP library is missing this module raises, and every product
entry point fails loudly with it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMC_HIP_LIB") or os.path.join(_HERE, "liblmc_hip.so")

ABI_VERSION = 8
OK = 0

KIND_NUTS, KIND_HMC = 0, 1
POT_DIAG_ADAPT, POT_DIAG, POT_FULL, POT_FULL_INV, POT_FULL_ADAPT, POT_FULL_F64 = range(6)
TARGET_STD_NORMAL, TARGET_DIAG_GAUSSIAN, TARGET_AR1, TARGET_FUNNEL, TARGET_NORMAL1D, TARGET_USER, TARGET_EXTERNAL = range(7)
STATUS_BAD_INITIAL_ENERGY = 1
SDOT_NATIVE, SDOT_OPENBLAS_SKYLAKEX, SDOT_OPENBLAS_HASWELL = 0, 1, 2
RNG_NUMPY, RNG_PHILOX = 0, 1
LDS_PLAN_AUTO, LDS_PLAN_SHALLOW, LDS_PLAN_DEEP = 0, 1, 2
PLANE_F64, PLANE_I32, PLANE_U8 = 0, 1, 2
AS_NATIVE, AS_F64, AS_I64 = 0, 1, 2
MAX_PLANES = 16
MAX_RUN_STREAMS = 8
(STAT_STEP_SIZE, STAT_STEP_SIZE_BAR, STAT_ACCEPT, STAT_ENERGY_ERROR, STAT_ENERGY, STAT_MAX_ENERGY_ERROR,
 STAT_MODEL_LOGP) = range(7)
STAT_DEPTH, STAT_TREE_SIZE = 0, 1
STAT_DIVERGING, STAT_TUNE, STAT_ACCEPTED = 0, 1, 2
CT_REACHED_MAX_TREEDEPTH, CT_DIVS_AFTER_TUNE, CT_SAMPLES_AFTER_TUNE, CT_LEAPFROGS, CT_WAVE_TICKS = range(5)
NUM_COUNTERS = 5
STAT_RECORD_BYTES = 64   # include/lmc_hip.h: one record of sampler statistics per draw


class Tuning(C.Structure):
    """struct lmc_tuning (include/lmc_hip.h): kernel-selection and layout knobs, 0 = the engine decides."""

    _fields_ = [("sub_blocks", C.c_int32), ("force_general", C.c_int32), ("general_team", C.c_int32), ("run_ns", C.c_int32),
                ("run_w", C.c_int32), ("dense_coop_off", C.c_int32), ("dense_cache_rows_p1", C.c_int32),
                ("dense_lds_slots_p1", C.c_int32), ("chol_hbm", C.c_int32), ("reserved", C.c_int32 * 3)]


class Config(C.Structure):
    """struct lmc_config (include/lmc_hip.h)."""

    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("chains", C.c_int32), ("dim", C.c_int32),
        ("kind", C.c_int32), ("target_family", C.c_int32), ("potential", C.c_int32),
        ("adapt_step_size", C.c_int32),
        ("target_accept", C.c_double), ("emax", C.c_double), ("step_scale", C.c_double),
        ("gamma", C.c_double), ("k", C.c_double), ("t0", C.c_double),
        ("max_treedepth", C.c_int32), ("early_max_treedepth", C.c_int32),
        ("path_length", C.c_double), ("max_steps", C.c_int32), ("adaptation_window", C.c_int32),
        ("lds_levels", C.c_int32), ("start_energy_sdot", C.c_int32), ("adaptation_window_multiplier", C.c_double),
        ("rng_mode", C.c_int32), ("mass_f64", C.c_int32), ("lds_plan", C.c_int32), ("reserved0", C.c_int32),
        ("tuning", Tuning),
    ]


class WindowPlane(C.Structure):
    """struct lmc_window_plane (include/lmc_hip.h)."""

    _fields_ = [("dst", C.c_void_p), ("kind", C.c_int32), ("idx", C.c_int32), ("as_", C.c_int32), ("reserved", C.c_int32)]


class WindowDst(C.Structure):
    """struct lmc_window_dst (include/lmc_hip.h): where lmc_engine_copy_window_async() puts a window of iterations."""

    _fields_ = [("n_out", C.c_int64), ("first", C.c_int64), ("trace", C.c_void_p), ("n_planes", C.c_int32),
                ("copy_workgroups", C.c_int32), ("plane", WindowPlane * MAX_PLANES)]


_P = C.c_void_p


class ChainState(C.Structure):
    """struct lmc_chain_state (include/lmc_hip.h): pointers, NULL = skip."""

    FIELDS = (("var", np.float32, True), ("fore_mean", np.float64, True), ("fore_raw_var", np.float64, True),
              ("back_mean", np.float64, True), ("back_raw_var", np.float64, True), ("fore_w_sum", np.float64, False),
              ("back_w_sum", np.float64, False), ("n_samples", np.int32, False), ("log_step", np.float64, False),
              ("log_bar", np.float64, False), ("hbar", np.float64, False), ("da_count", np.int32, False),
              ("iter_count", np.int32, False), ("window", np.int32, False), ("var64", np.float64, True))
    _fields_ = [(name, C.c_void_p) for name, _dt, _vec in FIELDS]


class DenseState(C.Structure):
    """struct lmc_dense_state (include/lmc_hip.h): (name, dtype, shape kind) with shape kind "m" = [chains, d, d],
    "v" = [chains, d], "s" = [chains]; pointers, NULL = skip."""

    FIELDS = (("cov", np.float32, "m"), ("chol", np.float32, "m"), ("fore_mean", np.float64, "v"),
              ("fore_raw_cov", np.float64, "m"), ("fore_n", np.float64, "s"), ("back_mean", np.float64, "v"),
              ("back_raw_cov", np.float64, "m"), ("back_n", np.float64, "s"), ("window", np.int32, "s"),
              ("previous_update", np.int32, "s"), ("chol_failures", np.int32, "s"),
              ("cov64", np.float64, "m"), ("chol64", np.float64, "m"))
    _fields_ = [(name, C.c_void_p) for name, _dt, _k in FIELDS]


_SIGNATURES = {
    # name: (restype, [argtypes])
    "lmc_config_defaults": (None, [C.POINTER(Config), C.c_int32, C.c_int32]),
    "lmc_last_error": (C.c_char_p, [_P]),
    "lmc_abi_version": (C.c_int32, []),
    "lmc_build_hash": (C.c_char_p, []),
    "lmc_device_count": (C.c_int32, []),
    "lmc_has_target": (C.c_int32, [C.c_int32]),
    "lmc_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "lmc_engine_destroy": (None, [_P]),
    "lmc_engine_set_stream": (C.c_int, [_P, _P]),
    "lmc_engine_synchronize": (C.c_int, [_P]),
    "lmc_engine_set_target_params": (C.c_int, [_P, _P, C.c_int64]),
    "lmc_engine_set_potential": (C.c_int, [_P, _P, _P, C.c_double, C.c_int32]),
    "lmc_engine_set_dense_potential": (C.c_int, [_P, _P, _P, C.c_double, C.c_int32, C.c_double, C.c_int32]),
    "lmc_engine_get_dense_state": (C.c_int, [_P, C.POINTER(DenseState)]),
    "lmc_engine_set_dense_state": (C.c_int, [_P, C.POINTER(DenseState)]),
    "lmc_engine_dense_update": (C.c_int, [_P, C.c_int32]),
    "lmc_engine_tick_begin": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32]),
    "lmc_engine_tick_positions": (_P, [_P]),
    "lmc_engine_tick": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int32)]),
    "lmc_engine_get_dense_chain": (C.c_int, [_P, C.c_int32, _P, _P]),
    "lmc_engine_get_dense_chain_f64": (C.c_int, [_P, C.c_int32, _P, _P]),
    "lmc_engine_get_dense_factor_f64": (C.c_int, [_P, _P]),
    "lmc_engine_seed": (C.c_int, [_P, _P]),
    "lmc_engine_set_rng_state": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_double]),
    "lmc_engine_get_rng_state": (C.c_int, [_P, C.c_int32, _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double)]),
    "lmc_engine_set_position": (C.c_int, [_P, _P, C.c_int32]),
    "lmc_engine_get_position": (C.c_int, [_P, _P]),
    "lmc_engine_reset_tuning": (C.c_int, [_P]),
    "lmc_engine_set_dual_average": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_int32]),
    "lmc_engine_reserve": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "lmc_engine_attach_trace": (C.c_int, [_P, _P, C.c_int64]),
    "lmc_engine_run": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32]),
    "lmc_engine_run_streams": (C.c_int, [_P, _P, C.c_int32]),
    "lmc_engine_last_run_plan": (C.c_int32, [_P]),
    "lmc_engine_copy_window_async": (C.c_int, [_P, C.POINTER(WindowDst), C.c_int64, C.c_int64]),
    "lmc_engine_copy_wait": (C.c_int, [_P]),
    "lmc_host_alloc": (_P, [C.c_uint64]),
    "lmc_host_free": (None, [_P]),
    "lmc_host_register": (C.c_int, [_P, C.c_uint64]),
    "lmc_host_unregister": (C.c_int, [_P]),
    "lmc_engine_set_step_jitter": (C.c_int, [_P, C.c_int32, C.c_double, C.c_double]),
    "lmc_engine_set_step_sizes": (C.c_int, [_P, _P]),
    "lmc_engine_diag_update": (C.c_int, [_P, C.c_int32]),
    "lmc_engine_get_trace": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "lmc_engine_get_stat_f64": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int64]),
    "lmc_engine_get_stat_i32": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int64]),
    "lmc_engine_get_stat_u8": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int64]),
    "lmc_engine_trace_device_ptr": (_P, [_P]),
    "lmc_engine_stat_records_device_ptr": (_P, [_P]),
    "lmc_engine_trace_begin": (C.c_int64, [_P]),
    "lmc_engine_capacity": (C.c_int64, [_P]),
    "lmc_engine_get_adapt_state": (C.c_int, [_P, _P, _P, _P, _P]),
    "lmc_engine_get_chain_state": (C.c_int, [_P, C.POINTER(ChainState)]),
    "lmc_engine_set_chain_state": (C.c_int, [_P, C.POINTER(ChainState)]),
    "lmc_engine_keep_moments": (C.c_int, [_P, C.c_int32]),
    "lmc_engine_get_moments": (C.c_int, [_P, _P, _P, _P]),
    "lmc_engine_get_status": (C.c_int, [_P, _P]),
    "lmc_engine_get_counters": (C.c_int, [_P, _P]),
    "lmc_engine_trajectory": (C.c_int, [_P, _P, _P, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                        _P, _P, _P, _P, _P, _P]),
    "lmc_engine_logp_dlogp": (C.c_int, [_P, _P, _P, _P]),
    "lmc_engine_rng_draw": (C.c_int, [_P, _P, C.c_int32, _P]),
    "lmc_engine_draw_momentum": (C.c_int, [_P, _P]),
    "lmc_engine_kernel_shape": (C.c_int, [_P, _P, _P, _P]),
    "lmc_engine_uses_general_kernels": (C.c_int32, [_P]),
    "lmc_engine_occupancy": (C.c_int, [_P, _P, _P, _P]),
    "lmc_engine_request_stop": (C.c_int, [_P, C.c_int32]),
    "lmc_engine_progress": (C.c_int64, [_P]),
    "lmc_engine_run_lds_bytes": (C.c_int32, [_P]),
    "lmc_engine_load_user_kernels": (C.c_int, [_P, _P, C.c_char_p, C.c_char_p, C.c_char_p]),
    "lmc_engine_load_user_run_plan1": (C.c_int, [_P, C.c_char_p]),
    "lmc_diag_lags_per_pass": (C.c_int, []),
    "lmc_diag_chain_stats": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_libs = {}


class HipLibraryError(RuntimeError):
    """The HIP library is missing, stale or failed; there is no CPU path to fall back to."""


def load(path=None):
    """dlopen liblmc_hip.so (or a user-target build of it) and type its entry points."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise HipLibraryError(
            "%s not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (needs hipcc). littlemcmc_amd has no CPU fallback." % path)
    try:
        lib = C.CDLL(path)
    except OSError as err:
        raise HipLibraryError("cannot load %s: %s" % (path, err))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError("%s does not export %s (stale build?)" % (path, name))
        fn.restype = res
        fn.argtypes = args
    if lib.lmc_abi_version() != ABI_VERSION:
        raise HipLibraryError("ABI version mismatch: python %d, library %d" % (ABI_VERSION, lib.lmc_abi_version()))
    _libs[path] = lib
    return lib


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or pass through an int device pointer / None)."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return C.c_void_p(a.ctypes.data)
