"""CPU ORACLE package -- test infrastructure only (see okkt_oracle.c header).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Nothing under
onephase.jl_amd/ imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libokkt_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "okkt_oracle.c")):
            build()
        L = C.CDLL(_SO)
        i64p, f64p, vp = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        L.oracle_analyze.restype = vp
        L.oracle_analyze.argtypes = [C.c_int64, i64p, i64p, C.c_int, i64p]
        L.oracle_free.argtypes = [vp]
        L.oracle_numeric.restype = C.c_int
        L.oracle_numeric.argtypes = [vp, f64p]
        L.oracle_ls_factor.restype = C.c_int
        L.oracle_ls_factor.argtypes = [vp, f64p, C.c_int64, C.c_int64, C.c_int]
        L.oracle_inertia.argtypes = [vp, C.c_double, i64p, i64p, i64p, i64p]
        L.oracle_solve.argtypes = [vp, f64p, f64p]
        for name, res in (("oracle_n", C.c_int64), ("oracle_lnz", C.c_int64), ("oracle_flops", C.c_double),
                          ("oracle_npiv_done", C.c_int64)):
            getattr(L, name).restype = res
            getattr(L, name).argtypes = [vp]
        L.oracle_get_D.argtypes = [vp, f64p]
        L.oracle_get_parent.argtypes = [vp, i64p]
        L.oracle_get_colcounts.argtypes = [vp, i64p]
        L.oracle_get_L.argtypes = [vp, i64p, i64p, f64p]
        _lib = L
    return _lib


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class linear_solver_ORACLE:
    """Restatement of linear_solver_JULIA (/root/reference/src/linear_system_solvers/julia.jl:1-113)
    on top of the C up-looking LDL^T.  `perm` (perm[new] = old) stands in for CHOLMOD's AMD."""

    def __init__(self, sym, safe_mode=False, recycle=False, perm=None):
        assert sym in ("definite", "symmetric")
        self.sym = sym
        self.perm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self._F = None
        self._pattern = None

    def __del__(self):
        try:
            if self._F is not None:
                lib().oracle_free(self._F)
        except Exception:
            pass

    def _analyze(self, A):
        A = sp.csc_matrix(A)
        A.sort_indices()
        key = (A.shape[0], A.indptr.tobytes(), A.indices.tobytes())
        if self._pattern != key:
            if self._F is not None:
                lib().oracle_free(self._F)
            colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
            rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
            F = lib().oracle_analyze(A.shape[0], _pi(colptr), _pi(rowval), 0, None if self.perm is None else _pi(self.perm))
            if not F:
                raise ValueError("oracle_analyze rejected the input")
            self._F = C.c_void_p(F)
            self._pattern = key
            self.n = A.shape[0]
        return np.ascontiguousarray(A.data, dtype=np.float64)

    def ls_factor_b(self, A, n, m, timer=None):
        vals = self._analyze(A)
        rc = lib().oracle_ls_factor(self._F, _pf(vals), n, m, 0 if self.sym == "definite" else 1)
        if rc < 0:
            raise ValueError("oracle_ls_factor: bad arguments")
        return rc

    def ls_solve(self, rhs, timer=None):
        rhs = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64).ravel())
        sol = np.empty(self.n)
        lib().oracle_solve(self._F, _pf(rhs), _pf(sol))
        return sol

    def ls_solve_b(self, rhs, sol, timer=None):
        sol[:] = self.ls_solve(rhs)

    # diagnostics
    def diag(self):
        d = np.empty(self.n)
        lib().oracle_get_D(self._F, _pf(d))
        return d

    def inertia(self, tol=1e-20):
        v = [C.c_int64() for _ in range(4)]
        lib().oracle_inertia(self._F, tol, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def etree(self):
        par = np.empty(self.n, dtype=np.int64)
        cnt = np.empty(self.n, dtype=np.int64)
        lib().oracle_get_parent(self._F, _pi(par))
        lib().oracle_get_colcounts(self._F, _pi(cnt))
        return par, cnt

    def lnz(self):
        return lib().oracle_lnz(self._F)

    def flops(self):
        return lib().oracle_flops(self._F)

    def L(self):
        n, lnz = self.n, self.lnz()
        Lp = np.empty(n + 1, dtype=np.int64)
        Li = np.empty(max(lnz, 1), dtype=np.int64)
        Lx = np.empty(max(lnz, 1))
        lib().oracle_get_L(self._F, _pi(Lp), _pi(Li), _pf(Lx))
        M = sp.csc_matrix((Lx[:lnz], Li[:lnz], Lp), shape=(n, n))
        M.sort_indices()
        return M


# ---------------------------------------------------------------------------------------------------------------
# Supernodal multifrontal CPU baseline (okkt_oracle_mf.c): the algorithm class of CHOLMOD's supernodal numeric phase,
# OpenMP over the elimination tree and inside the large fronts.  bench.py's cpu_baseline leg times it on all host
# cores and on one; the tests check it against the simplicial oracle above.
# ---------------------------------------------------------------------------------------------------------------
_SO_MF = os.path.join(_HERE, "libokkt_oracle_mf.so")
_lib_mf = None


def lib_mf():
    global _lib_mf
    if _lib_mf is None:
        if not os.path.exists(_SO_MF) or os.path.getmtime(_SO_MF) < os.path.getmtime(os.path.join(_HERE, "okkt_oracle_mf.c")):
            build()
        L = C.CDLL(_SO_MF)
        i64p, f64p, vp = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        L.mf_analyze.restype = vp
        L.mf_analyze.argtypes = [C.c_int64, i64p, i64p, C.c_int64, i64p]
        L.mf_free.argtypes = [vp]
        L.mf_factor_numeric.restype = C.c_int
        L.mf_factor_numeric.argtypes = [vp, f64p, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, i64p]
        L.mf_solve.argtypes = [vp, f64p, f64p]
        L.mf_get_D.argtypes = [vp, f64p]
        for name, res in (("mf_flops", C.c_double), ("mf_nsuper", C.c_int64), ("mf_max_front", C.c_int64), ("mf_lnz", C.c_int64)):
            getattr(L, name).restype = res
            getattr(L, name).argtypes = [vp]
        _lib_mf = L
    return _lib_mf


class linear_solver_ORACLE_MF:
    """Same interface as linear_solver_ORACLE, multifrontal and multi-threaded (nthreads <= 0: OpenMP's default)."""

    def __init__(self, sym, safe_mode=False, recycle=False, perm=None, nthreads=0):
        assert sym in ("definite", "symmetric")
        self.sym = sym
        self.perm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self.nthreads = int(nthreads)
        self._F = None
        self._pattern = None
        self._counts = None

    def __del__(self):
        try:
            if self._F is not None:
                lib_mf().mf_free(self._F)
        except Exception:
            pass

    def _analyze(self, A):
        A = sp.csc_matrix(A)
        A.sort_indices()
        key = (A.shape[0], A.indptr.tobytes(), A.indices.tobytes())
        if self._pattern != key:
            if self._F is not None:
                lib_mf().mf_free(self._F)
            colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
            rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
            F = lib_mf().mf_analyze(A.shape[0], _pi(colptr), _pi(rowval), 0, None if self.perm is None else _pi(self.perm))
            if not F:
                raise ValueError("mf_analyze rejected the input")
            self._F = C.c_void_p(F)
            self._pattern = key
            self.n = A.shape[0]
        return np.ascontiguousarray(A.data, dtype=np.float64)

    def ls_factor_b(self, A, n, m, timer=None):
        vals = self._analyze(A)
        counts = np.zeros(4, dtype=np.int64)
        rc = lib_mf().mf_factor_numeric(self._F, _pf(vals), n, m, 0 if self.sym == "definite" else 1, 1e-20, self.nthreads, _pi(counts))
        self._counts = tuple(int(v) for v in counts)
        return int(rc)

    def ls_solve(self, rhs, timer=None):
        rhs = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64).ravel())
        sol = np.empty(self.n)
        lib_mf().mf_solve(self._F, _pf(rhs), _pf(sol))
        return sol

    def inertia(self, tol=1e-20):
        return self._counts

    def diag(self):
        d = np.empty(self.n)
        lib_mf().mf_get_D(self._F, _pf(d))
        return d

    def flops(self):
        return lib_mf().mf_flops(self._F)

    def stats(self):
        L = lib_mf()
        return dict(flops=L.mf_flops(self._F), nsuper=L.mf_nsuper(self._F), max_front=L.mf_max_front(self._F), lnz_stored=L.mf_lnz(self._F))
