"""Host-side mirror of the reference's linear-solver plug-in interface, bound to the HIP library.

Mirrors /root/reference/src/linear_system_solvers/linear_system_solvers.jl (abstract type,
initialize!/finalize!, inertia_status) and the back-end pattern of hsl.jl / julia.jl
(`X(sym, safe_mode, recycle)`, ls_factor!, ls_solve!, ls_solve).  Julia's `f!` becomes `f_b`
("bang") here; argument order and meaning are the reference's.  The Julia glue a maintainer
would add is julia/linear_solver_hip.jl; this Python class calls the same C ABI through ctypes
so that the parity tests can read like test/linear_system_solvers.jl.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib as L


class OkktError(RuntimeError):
    pass


class abstract_linear_system_solver:  # linear_system_solvers.jl:11
    pass


def initialize_b(solver):  # linear_system_solvers.jl:40
    solver._initialize()


def finalize_b(solver):  # linear_system_solvers.jl:44
    solver._finalize()


def inertia_status(pos_eigs, neg_eigs, zero_eigs, num_vars, num_constraints):
    """linear_system_solvers.jl:48-91 -- is the inertia (num_vars, num_constraints, 0)?"""
    if pos_eigs + neg_eigs + zero_eigs != num_vars + num_constraints:
        raise OkktError("pos_eigs + neg_eigs + zero_eigs != num_vars + num_constraints")
    return pos_eigs == num_vars and neg_eigs == num_constraints


def csc_arrays(A):
    """(dim, colptr, rowval, nzval, index_base) of a square sparse matrix, SparseMatrixCSC-like."""
    if isinstance(A, tuple):
        dim, colptr, rowval, nzval, base = A
        return int(dim), L.i64(colptr), L.i64(rowval), L.f64(nzval), int(base)
    A = sp.csc_matrix(A)
    if A.shape[0] != A.shape[1]:
        raise OkktError("matrix must be square")
    if not A.has_sorted_indices:
        A = A.copy()
        A.sort_indices()
    return A.shape[0], L.i64(A.indptr), L.i64(A.indices), L.f64(A.data), 0


class linear_solver_HIP(abstract_linear_system_solver):
    """`linear_solver_HIP(sym, safe_mode, recycle)` -- constructor shape of julia.jl:11 / hsl.jl:17."""

    def __init__(self, sym, safe_mode=False, recycle=False, **opts):
        if sym not in ("definite", "symmetric"):
            # julia.jl:95: error("this.options.sym = ... not supported")
            raise OkktError(f"this.options.sym = {sym} not supported")
        self.sym = sym
        self.safe_mode = safe_mode
        self.recycle = recycle
        self._opts = opts
        self._h = None
        self._lib = None
        self.inertia = None  # (pos, neg, zero, nonfinite) of the last factorisation
        self._dim = 0

    # -- initialize! / finalize!
    def _initialize(self):
        if self._h is not None:
            return
        self._lib = L.load()
        o = L.OkktOpts()
        self._lib.okkt_default_opts(C.byref(o))
        for k, v in self._opts.items():
            if not hasattr(o, k):
                raise OkktError(f"unknown option {k}")
            setattr(o, k, v)
        h = C.c_void_p()
        rc = self._lib.okkt_create(C.byref(h), C.byref(o))
        if rc != L.OKKT_OK:
            raise OkktError(
                f"okkt_create failed with code {rc}"
                + (" (no HIP device: the KKT path has no CPU fallback)" if rc == L.OKKT_ERR_NO_DEVICE else "")
            )
        self._h = h

    def _finalize(self):
        if self._h is not None:
            self._lib.okkt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._finalize()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            msg = self._lib.okkt_last_error(self._h)
            raise OkktError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    def _need(self):
        if self._h is None:
            raise OkktError("initialize_b(solver) has not been called")

    # -- analysis helpers (not part of the reference interface)
    def set_perm(self, perm):
        self._need()
        p = L.i64(perm)
        self._check(self._lib.okkt_set_perm(self._h, L.p_i64(p), len(p)), "okkt_set_perm")

    def analyze(self, A):
        self._need()
        dim, colptr, rowval, _, base = csc_arrays(A)
        self._check(self._lib.okkt_analyze(self._h, dim, L.p_i64(colptr), L.p_i64(rowval), base), "okkt_analyze")
        self._dim = dim

    def perm(self):
        out = np.zeros(self._dim, dtype=np.int64)
        self._check(self._lib.okkt_get_perm(self._h, L.p_i64(out)), "okkt_get_perm")
        return out

    def etree(self):
        par = np.zeros(self._dim, dtype=np.int64)
        cnt = np.zeros(self._dim, dtype=np.int64)
        self._check(self._lib.okkt_get_etree(self._h, L.p_i64(par), L.p_i64(cnt)), "okkt_get_etree")
        return par, cnt

    def stats(self):
        st = L.OkktStats()
        self._check(self._lib.okkt_get_stats(self._h, C.byref(st)), "okkt_get_stats")
        return st.as_dict()

    def diag(self):
        """diag(F) (julia.jl:72): D in pivot order."""
        out = np.zeros(self._dim)
        self._check(self._lib.okkt_get_diag(self._h, L.p_f64(out)), "okkt_get_diag")
        return out

    def factor_csc(self):
        """L (strictly lower, permuted numbering) as scipy CSC -- parity tests only."""
        nnz = C.c_int64()
        self._check(self._lib.okkt_get_factor_csc(self._h, None, None, None, C.byref(nnz)), "okkt_get_factor_csc")
        colptr = np.zeros(self._dim + 1, dtype=np.int64)
        rowval = np.zeros(max(nnz.value, 1), dtype=np.int64)
        val = np.zeros(max(nnz.value, 1))
        self._check(self._lib.okkt_get_factor_csc(self._h, L.p_i64(colptr), L.p_i64(rowval), L.p_f64(val), C.byref(nnz)),
                    "okkt_get_factor_csc")
        return sp.csc_matrix((val[: nnz.value], rowval[: nnz.value], colptr), shape=(self._dim, self._dim))

    # -- device-resident variants: inputs already in HBM (bench.py, the KKT layer)
    def dev_upload(self, arr):
        """Copy a contiguous numpy array into a fresh HBM buffer; returns the device pointer (int)."""
        self._need()
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._check(self._lib.okkt_dev_alloc(self._h, arr.nbytes, C.byref(p)), "okkt_dev_alloc")
        self._check(self._lib.okkt_dev_upload(self._h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "okkt_dev_upload")
        return p.value

    def dev_alloc(self, nbytes):
        self._need()
        p = C.c_void_p()
        self._check(self._lib.okkt_dev_alloc(self._h, nbytes, C.byref(p)), "okkt_dev_alloc")
        return p.value

    def dev_download(self, ptr, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.okkt_dev_download(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes), "okkt_dev_download")
        return out

    def dev_free(self, ptr):
        self._check(self._lib.okkt_dev_free(self._h, C.c_void_p(ptr)), "okkt_dev_free")

    def ls_factor_dev(self, d_nzval, n, m):
        """ls_factor! with nzval resident in HBM (pattern analysed beforehand with analyze())."""
        self._need()
        inert = L.OkktInertia()
        kind = L.OKKT_SYM_DEFINITE if self.sym == "definite" else L.OKKT_SYM_SYMMETRIC
        rc = self._check(self._lib.okkt_factor_dev(self._h, C.c_void_p(d_nzval), n, m, kind, C.byref(inert)), "okkt_factor_dev")
        self.inertia = inert.as_tuple()
        return int(rc)

    def ls_solve_dev(self, d_rhs, d_sol, nrhs=1):
        self._need()
        self._check(self._lib.okkt_solve_dev(self._h, C.c_void_p(d_rhs), C.c_void_p(d_sol), nrhs), "okkt_solve_dev")

    def profile_dominant(self, enable):
        self._check(self._lib.okkt_profile_dominant(self._h, 1 if enable else 0), "okkt_profile_dominant")

    def get_profile(self):
        n = C.c_int64(); ms = C.c_double(); fl = C.c_double()
        self._check(self._lib.okkt_get_profile(self._h, C.byref(n), C.byref(ms), C.byref(fl)), "okkt_get_profile")
        return n.value, ms.value, fl.value

    # -- the reference interface
    def ls_factor_b(self, SparseMatrix, n, m, timer=None):
        """ls_factor!(solver, A, n, m, timer) -> 1 if the inertia is (n, m, 0), else 0 (julia.jl:21-97)."""
        self._need()
        dim, colptr, rowval, nzval, base = csc_arrays(SparseMatrix)
        if timer is not None:
            timer.start("HIP/factorize")
        try:
            self._check(self._lib.okkt_analyze(self._h, dim, L.p_i64(colptr), L.p_i64(rowval), base), "okkt_analyze")
            self._dim = dim
            inert = L.OkktInertia()
            kind = L.OKKT_SYM_DEFINITE if self.sym == "definite" else L.OKKT_SYM_SYMMETRIC
            rc = self._check(self._lib.okkt_factor(self._h, L.p_f64(nzval), n, m, kind, C.byref(inert)), "okkt_factor")
            self.inertia = inert.as_tuple()
        finally:
            if timer is not None:
                timer.pause("HIP/factorize")
        return int(rc)

    def ls_solve_b(self, my_rhs, my_sol, timer=None):
        """ls_solve!(solver, rhs, sol, timer): sol[1:end] = F \\ rhs (julia.jl:99-103)."""
        self._need()
        rhs = L.f64(my_rhs)
        if rhs.shape != (self._dim,) or my_sol.shape != (self._dim,) or my_sol.dtype != np.float64:
            raise OkktError("rhs/sol must be float64 vectors of the factorised dimension")
        if timer is not None:
            timer.start("HIP/ls_solve")
        try:
            out = np.empty(self._dim)
            self._check(self._lib.okkt_solve(self._h, L.p_f64(rhs), L.p_f64(out), 1), "okkt_solve")
            my_sol[:] = out
        finally:
            if timer is not None:
                timer.pause("HIP/ls_solve")

    def ls_solve(self, my_rhs, timer=None):
        """ls_solve(solver, rhs, timer) -> F \\ Vector(rhs); sparse vectors are densified (julia.jl:105-113)."""
        if sp.issparse(my_rhs):
            my_rhs = np.asarray(my_rhs.todense()).ravel()
        sol = np.empty(self._dim)
        self.ls_solve_b(np.asarray(my_rhs, dtype=np.float64).ravel(), sol, timer)
        return sol
