"""ctypes binding of libonephase_kkt.so (C ABI: include/okkt.h).

This is the Python counterpart of the `ccall` stubs in julia/linear_solver_hip.jl: plain
pointers and sizes, no torch types.  The library is built in-tree by `__graft_entry__.build()`
(hipcc, gfx950); there is no CPU fallback -- if the shared object is missing or no HIP device is
present the compute entry points raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OKKT_LIB_PATH", os.path.join(_HERE, "libonephase_kkt.so"))   # override: A/B builds

OKKT_SYM_DEFINITE = 0
OKKT_SYM_SYMMETRIC = 1
OKKT_KKT_SCHUR = 0
OKKT_KKT_SYMMETRIC = 1
OKKT_KKT_CLEVER_SYMMETRIC = 2
OKKT_KKT_SCHUR_DIRECT = 3
OKKT_RESCALE = {"none": 0, "u_only": 1, "u_and_x": 2}

OKKT_OK = 0
OKKT_ERR_INVALID = -1
OKKT_ERR_NO_DEVICE = -2
OKKT_ERR_HIP = -3
OKKT_ERR_ALLOC = -4
OKKT_ERR_INTERNAL = -5


class OkktOpts(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("host_symbolic_only", C.c_int32),
        ("ordering", C.c_int32),
        ("relax_always", C.c_int32),
        ("relax_small", C.c_int32),
        ("relax_mid", C.c_int32),
        ("relax_small_frac", C.c_double),
        ("relax_mid_frac", C.c_double),
        ("relax_any_frac", C.c_double),
        ("inertia_tol", C.c_double),
        ("small_front_max", C.c_int32),
        ("panel_nb", C.c_int32),
        ("early_exit", C.c_int32),
        ("reserved", C.c_int32),
    ]


class OkktInertia(C.Structure):
    _fields_ = [("pos", C.c_int64), ("neg", C.c_int64), ("zero", C.c_int64), ("nonfinite", C.c_int64)]

    def as_tuple(self):
        return (self.pos, self.neg, self.zero, self.nonfinite)


class OkktStats(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("nnz_lower", C.c_int64),
        ("nnzL", C.c_int64),
        ("nnzL_stored", C.c_int64),
        ("flops_exact", C.c_double),
        ("flops_stored", C.c_double),
        ("arena_bytes", C.c_int64),
        ("nsuper", C.c_int64),
        ("nlevels", C.c_int64),
        ("max_front", C.c_int64),
        ("n_small_fronts", C.c_int64),
        ("n_big_fronts", C.c_int64),
        ("sum_rowidx", C.c_int64),
        ("analyze_seconds", C.c_double),
        ("last_factor_ms", C.c_double),
        ("last_solve_ms", C.c_double),
        ("pattern_hash", C.c_uint64),
        ("n_analyze_calls", C.c_int64),
        ("ordering_used", C.c_int64),
        ("critical_pivots", C.c_int64),
        ("top_separator", C.c_int64),
        ("amd_skipped", C.c_int64),
        ("flops_other", C.c_double),
        ("arena_dense_bytes", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class OkktKktPars(C.Structure):
    _fields_ = [
        ("delta_start", C.c_double),
        ("delta_min", C.c_double),
        ("delta_max", C.c_double),
        ("delta_inc", C.c_double),
        ("delta_dec", C.c_double),
        ("delta_zero", C.c_double),
        ("ItRefine_Num", C.c_int32),
        ("max_it", C.c_int32),
    ]


class OkktKktError(C.Structure):
    _fields_ = [
        ("error_D", C.c_double),
        ("error_P", C.c_double),
        ("error_mu", C.c_double),
        ("overall", C.c_double),
        ("rhs_norm", C.c_double),
        ("ratio", C.c_double),
    ]


class OkktKktTimers(C.Structure):
    _fields_ = [
        ("assemble_ms", C.c_double),
        ("upload_ms", C.c_double),
        ("shift_ms", C.c_double),
        ("factor_ms", C.c_double),
        ("rhs_ms", C.c_double),
        ("solve_ms", C.c_double),
        ("refine_ms", C.c_double),
        ("kkt_err_ms", C.c_double),
        ("direction_ms", C.c_double),
        ("n_solves", C.c_int32),
        ("reserved", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


# every symbol include/okkt.h declares, with its signature
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p
SIGNATURES = {
    "okkt_default_opts": (C.c_int, [C.POINTER(OkktOpts)]),
    "okkt_create": (C.c_int, [C.POINTER(_vp), C.POINTER(OkktOpts)]),
    "okkt_set_early_exit": (C.c_int, [_vp, C.c_int]),
    "okkt_destroy": (C.c_int, [_vp]),
    "okkt_last_error": (C.c_char_p, [_vp]),
    "okkt_version": (C.c_char_p, []),
    "okkt_set_perm": (C.c_int, [_vp, _i64p, C.c_int64]),
    "okkt_analyze": (C.c_int, [_vp, C.c_int64, _i64p, _i64p, C.c_int]),
    "okkt_get_perm": (C.c_int, [_vp, _i64p]),
    "okkt_get_stats": (C.c_int, [_vp, C.POINTER(OkktStats)]),
    "okkt_get_etree": (C.c_int, [_vp, _i64p, _i64p]),
    "okkt_factor": (C.c_int, [_vp, _f64p, C.c_int64, C.c_int64, C.c_int, C.POINTER(OkktInertia)]),
    "okkt_factor_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(OkktInertia)]),
    "okkt_solve": (C.c_int, [_vp, _f64p, _f64p, C.c_int64]),
    "okkt_solve_dev": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "okkt_get_diag": (C.c_int, [_vp, _f64p]),
    "okkt_get_factor_csc": (C.c_int, [_vp, _i64p, _i64p, _f64p, _i64p]),
    "okkt_dev_alloc": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "okkt_dev_free": (C.c_int, [_vp, _vp]),
    "okkt_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "okkt_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "okkt_get_stream": (_vp, [_vp]),
    "okkt_profile_dominant": (C.c_int, [_vp, C.c_int]),
    "okkt_get_profile": (C.c_int, [_vp, _i64p, _f64p, _f64p]),
    "okkt_debug_dataflow_queue": (C.c_int64, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int64, _f64p]),
    "okkt_dist_set_partition": (C.c_int, [_vp, C.c_int, C.c_int]),
    "okkt_dist_info": (C.c_int, [_vp, _i64p, _i64p, _i64p, _f64p, _f64p]),
    "okkt_dist_get_owner": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "okkt_dist_factor_local": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int]),
    "okkt_dist_cb": (C.c_int, [_vp, _vp, C.c_int]),
    "okkt_dist_factor_top": (C.c_int, [_vp]),
    "okkt_dist_counts": (C.c_int, [_vp, _i64p]),
    "okkt_dist_finish": (C.c_int, [_vp, _i64p]),
    "okkt_dist_solve_begin": (C.c_int, [_vp, _vp]),
    "okkt_dist_cv": (C.c_int, [_vp, _vp, C.c_int]),
    "okkt_dist_solve_top": (C.c_int, [_vp]),
    "okkt_dist_x": (C.c_int, [_vp, _vp, C.c_int]),
    "okkt_dist_solve_end": (C.c_int, [_vp]),
    "okkt_dist_unique_id": (C.c_int, [_vp]),
    "okkt_dist_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "okkt_dist_comm_destroy": (C.c_int, [_vp]),
    "okkt_dist_factor": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(OkktInertia)]),
    "okkt_dist_solve": (C.c_int, [_vp, _vp, _vp]),
    "okkt_kkt_default_pars": (C.c_int, [C.POINTER(OkktKktPars)]),
    "okkt_kkt_create": (C.c_int, [C.POINTER(_vp), C.POINTER(OkktOpts), C.c_int]),
    "okkt_kkt_destroy": (C.c_int, [_vp]),
    "okkt_kkt_last_error": (C.c_char_p, [_vp]),
    "okkt_kkt_linear_solver": (_vp, [_vp]),
    "okkt_kkt_set_structure": (C.c_int, [_vp, C.c_int64, C.c_int64, _i64p, _i64p, _i64p, _i64p, C.c_int]),
    "okkt_kkt_form_system": (C.c_int, [_vp, _f64p, _f64p, _f64p, _f64p]),
    "okkt_kkt_diag_min": (C.c_int, [_vp, _f64p]),
    "okkt_kkt_factor": (C.c_int, [_vp, C.c_double, C.POINTER(OkktInertia)]),
    "okkt_kkt_factor_trial": (C.c_int, [_vp, C.c_double, C.POINTER(OkktInertia)]),
    "okkt_kkt_get_timers": (C.c_int, [_vp, C.POINTER(OkktKktTimers)]),
    "okkt_kkt_get_direction": (C.c_int, [_vp, _f64p, _f64p, _f64p]),
    "okkt_kkt_ipopt_strategy": (C.c_int, [_vp, C.c_double, C.POINTER(OkktKktPars), C.POINTER(C.c_int32), _f64p]),
    "okkt_kkt_compute_directions": (C.c_int, [_vp, C.c_int32, _f64p, C.c_int32, _f64p, _f64p, _f64p, C.POINTER(OkktKktError)]),
    "okkt_kkt_is_diag_dom": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "okkt_kkt_diag_dom_warnings": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "okkt_kkt_estimate_y_tilde": (C.c_int, [_vp, _f64p, _f64p]),
    "okkt_kkt_system_rhs": (C.c_int, [_vp, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _f64p, _f64p, _f64p]),
    "okkt_kkt_compute_direction": (C.c_int, [_vp, _f64p, _f64p, _f64p, C.c_int32, _f64p, _f64p, _f64p, C.POINTER(OkktKktError)]),
    "okkt_kkt_get_matrix": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p, _f64p]),
    "okkt_kkt_get_schur_diag": (C.c_int, [_vp, _f64p]),
    "okkt_kkt_compute_indicies": (C.c_int, [_vp, _f64p, C.POINTER(C.c_int64)]),
    "okkt_kkt_get_indicies": (C.c_int, [_vp, _i64p, _i64p, _i64p, _f64p, _f64p, _f64p, _f64p]),
    "okkt_kkt_set_rescale": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double]),
    "okkt_kkt_set_direction": (C.c_int, [_vp, _f64p, _f64p, _f64p]),
    "okkt_kkt_max_step_primal": (C.c_int, [_vp, _f64p, C.c_double, _f64p, _f64p]),
    "okkt_kkt_s_bound_ok": (C.c_int, [_vp, _f64p, _f64p, C.c_double, C.POINTER(C.c_int32)]),
    "okkt_kkt_dual_step_range": (C.c_int, [_vp, _f64p, _f64p, C.c_double, C.c_double, _f64p, _f64p, _f64p]),
    "okkt_kkt_predicted_reduction": (C.c_int, [_vp, _f64p, C.c_double, C.c_double, C.c_double, C.c_double, _f64p]),
    "okkt_kkt_dual_step": (C.c_int, [_vp, _f64p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                     C.c_int, C.c_double, C.c_double, _f64p]),
}

_lib = None
MISSING = []


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the KKT path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:  # reported by tests/test_abi.py; calling it raises AttributeError
            MISSING.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def p_i64(a):
    return a.ctypes.data_as(_i64p)


def p_f64(a):
    return a.ctypes.data_as(_f64p)
