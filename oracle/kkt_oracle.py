"""CPU ORACLE (test infrastructure): restatement of the reference's KKT-system layer with scipy.

Follows, function by function:
  /root/reference/src/kkt_system_solver/kkt_system_solver.jl:27-47,67-113,167-204,291-300
  /root/reference/src/kkt_system_solver/schur.jl:47-182
  /root/reference/src/kkt_system_solver/schur_direct.jl:32-66
  /root/reference/src/kkt_system_solver/symmetric.jl:35-102
  /root/reference/src/kkt_system_solver/system_rhs.jl:3-24,57-73
  /root/reference/src/IPM/delta_strategy.jl:37-121
  /root/reference/src/utils/eval.jl:53-63,85-108,136-142,221-234
Linear algebra underneath = oracle.linear_solver_ORACLE (C up-looking LDL^T).
"""
import math
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from . import linear_solver_ORACLE


@dataclass
class Iterate:
    """Class_iterate + Class_cache subset (Class_iterate.jl:4-20,40-84; Class_point.jl:1-12)."""
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    mu: float
    J: sp.csc_matrix            # m x n
    H: sp.csc_matrix            # n x n, LOWER TRIANGLE ONLY (Class_cutest.jl:548)
    grad: np.ndarray            # gradient of f
    cons: np.ndarray            # a(x)
    a_norm_penalty_par: float = 1e-4
    delta: float = 0.0
    primal_scale: float = 1.0

    def dim(self):
        return len(self.x)

    def ncon(self):
        return len(self.s)


@dataclass
class Class_reduction_factors:  # system_rhs.jl:3-13
    P: float = math.nan
    D: float = math.nan
    mu: float = math.nan


def Reduct_affine():  # system_rhs.jl:16-19
    return Class_reduction_factors(0.0, 0.0, 0.0)


def Reduct_stable():  # system_rhs.jl:21-24
    return Class_reduction_factors(1.0, 0.0, 1.0)


# ---- eval.jl subset
def eval_jac_prod(it, x):  # eval.jl:102-104
    return it.J @ x


def eval_jac_T_prod(it, y):  # eval.jl:106-108
    return it.J.T @ y


def vector_product(Lmat, v):  # eval.jl:221-230
    return Lmat @ v + Lmat.T @ v - Lmat.diagonal() * v


def hess_product(it, v):  # eval.jl:232-234
    return vector_product(it.H, v)


def eval_grad_r(it):  # eval.jl:59-63
    return it.a_norm_penalty_par * eval_jac_T_prod(it, np.ones(it.ncon()))


def eval_grad_lag(it, mu, y=None):  # eval.jl:136-142
    y = it.y if y is None else y
    return it.grad - eval_jac_T_prod(it, y) + mu * eval_grad_r(it)


def eval_J_T_J(it, diag_vals):  # eval.jl:85-87
    return (it.J.T @ sp.diags(diag_vals) @ it.J).tocsc()


def eval_diag_J_T_J(it, diag_vals):  # eval.jl:89-100
    J2 = it.J.copy()
    J2.data = J2.data ** 2
    return np.asarray(J2.T @ diag_vals).ravel()


def compute_schur_diag(it):  # kkt_system_solver.jl:296-300
    return it.H.diagonal() + eval_diag_J_T_J(it, it.y / it.s)


@dataclass
class System_rhs:  # system_rhs.jl:34-73
    dual_r: np.ndarray
    primal_r: np.ndarray
    comp_r: np.ndarray

    @staticmethod
    def build(it, reduct):
        dual_target = -eval_grad_lag(it, it.mu * reduct.mu) * (1.0 - reduct.D)
        primal_target = -(it.cons - it.s) * (1.0 - reduct.P)
        mu_target = it.mu * reduct.mu
        return System_rhs(dual_target, primal_target, mu_target - it.s * it.y)


@dataclass
class Class_kkt_error:  # kkt_system_solver.jl:49-65
    error_D: float = 0.0
    error_P: float = 0.0
    error_mu: float = 0.0
    overall: float = 0.0
    rhs_norm: float = 0.0
    ratio: float = 0.0


@dataclass
class Direction:  # Class_point as a direction
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    mu: float = 0.0
    primal_scale: float = 0.0


@dataclass
class KKTPars:
    ItRefine_Num: int = 3          # parameters.jl:20
    ItRefine_BigFloat: bool = False   # parameters.jl:21; true cannot run in the reference: schur.jl:167 -> hess_product(::Array{BigFloat}) has no method (eval.jl:232)
    delta_start: float = 1e-6      # parameters.jl:147-158
    delta_min: float = 1e-12
    delta_max: float = 1e50
    delta_inc: float = 8.0
    delta_dec: float = 1.0 / math.pi
    delta_zero: float = 0.0


class _KKTBase:
    def __init__(self, ls_solver, pars=None):
        self.ls_solver = ls_solver
        self.pars = pars or KKTPars()
        self.ready = "not_ready"
        self.kkt_err_norm = Class_kkt_error()
        self.factor_it = None
        self.dir = None
        self.rhs = None
        self.Q = None

    # kkt_system_solver.jl:21-25
    def initialize_b(self, it):
        self.dir = Direction(np.zeros(it.dim()), np.zeros(it.ncon()), np.zeros(it.ncon()))

    # kkt_system_solver.jl:109-113
    def update_delta_b(self, delta_x, delta_s):
        delta_x_vec = delta_x * np.ones(self.factor_it.dim())
        delta_s_vec = delta_s * self.factor_it.s ** (-2.0)
        self.update_delta_vecs_b(delta_x_vec, delta_s_vec)

    # kkt_system_solver.jl:98-107,190-204
    def factor_b(self, delta_x=None, delta_s=0.0):
        if delta_x is not None:
            self.update_delta_b(delta_x, delta_s)
        if self.ready != "delta_updated":
            raise RuntimeError(f"kkt solver not ready to factor kkt_solver.ready = {self.ready} != :delta_updated")
        self.ready = "factored"
        return self.factor_implementation_b()

    # kkt_system_solver.jl:167-176
    def kkt_associate_rhs_b(self, it, eta):
        self.rhs = System_rhs.build(it, eta)
        self.dir.mu = -(1.0 - eta.mu) * it.mu
        self.dir.primal_scale = -(1.0 - eta.P) * it.primal_scale
        self.reduct_factors = eta          # schur.jl:42-43
        self.current_it = it

    # kkt_system_solver.jl:178-188
    def compute_direction_b(self):
        if self.ready != "factored":
            raise RuntimeError("kkt solver not ready to compute direction!")
        self.compute_direction_implementation_b()
        for v in (self.dir.x, self.dir.y, self.dir.s):
            if not np.all(np.isfinite(v)):
                raise FloatingPointError("NaN in direction")  # check_for_nan, IPM_tools.jl:32-49

    # kkt_system_solver.jl:27-47
    def predicted_lag_change(self):
        fi = self.factor_it
        tmp1 = eval_jac_prod(fi, self.dir.x)
        tmp2 = self.delta_s_vec * tmp1
        tmp3 = eval_jac_T_prod(fi, tmp2)
        delta_err = self.delta_x_vec * self.dir.x + tmp3
        J_err = eval_jac_T_prod(fi, self.dir.y)
        H_err = hess_product(fi, self.dir.x)
        return delta_err + H_err - J_err

    # kkt_system_solver.jl:67-96 (p = Inf)
    def update_kkt_error_b(self):
        fi, rhs, d = self.factor_it, self.rhs, self.dir
        error_D = self.predicted_lag_change() - rhs.dual_r
        error_P = eval_jac_prod(fi, d.x) - d.s - rhs.primal_r
        error_mu = fi.s * d.y + fi.y * d.s - rhs.comp_r
        inf = lambda v: float(np.max(np.abs(v))) if len(v) else 0.0
        overall = max(inf(error_D), inf(error_P), inf(error_mu))
        rhs_norm = max(inf(rhs.dual_r), inf(rhs.primal_r), inf(rhs.comp_r))
        self.kkt_err_norm = Class_kkt_error(inf(error_D), inf(error_P), inf(error_mu), overall, rhs_norm,
                                            overall / rhs_norm if rhs_norm != 0 else math.inf if overall > 0 else math.nan)

    def diag_min(self):  # kkt_system_solver.jl:291-294
        return float(np.min(self.schur_diag))


class Schur_KKT_solver(_KKTBase):
    """schur.jl:3-182"""

    def form_system_b(self, it):  # schur.jl:47-62
        self.Q = (eval_J_T_J(it, it.y / it.s) + it.H).tolil()
        self.schur_diag = np.asarray(self.Q.diagonal()).copy()
        self.factor_it = it
        self.ready = "system_formed"

    def update_delta_vecs_b(self, delta_x_vec, delta_s_vec):  # schur.jl:64-83
        self.delta_x_vec, self.delta_s_vec = delta_x_vec, delta_s_vec
        if np.sum(np.abs(delta_s_vec)) > 0.0:
            raise NotImplementedError("Not implemented")
        self.Q.setdiag(self.schur_diag + delta_x_vec)
        self.ready = "delta_updated"

    def factor_implementation_b(self):  # schur.jl:85-87
        return self.ls_solver.ls_factor_b(self.Q.tocsc(), self.factor_it.dim(), 0)

    def compute_direction_implementation_b(self):  # schur.jl:89-128
        fi, rhs = self.factor_it, self.rhs
        y_org, s_org = fi.y, fi.s
        symmetric_primal_rhs = rhs.primal_r + rhs.comp_r / y_org
        S_vec = y_org / s_org
        y_ = rhs.primal_r * S_vec + rhs.comp_r / s_org
        schur_rhs = rhs.dual_r + eval_jac_T_prod(fi, y_)
        d = self.dir
        d.x = self.solver_schur_rhs(schur_rhs)
        d.y = -(eval_jac_prod(fi, d.x) - symmetric_primal_rhs) * S_vec
        d.s = eval_jac_prod(fi, d.x) - rhs.primal_r
        self.update_kkt_error_b()

    def solver_schur_rhs(self, schur_rhs):  # schur.jl:131-182
        fit = self.factor_it
        if self.pars.ItRefine_BigFloat:
            raise TypeError("MethodError: no method matching hess_product(::Class_iterate, ::Array{BigFloat,1})")   # schur.jl:154-167, eval.jl:232
        S_vec = fit.y / fit.s
        res_old = schur_rhs
        dir_x = np.zeros(fit.dim())
        for _ in range(self.pars.ItRefine_Num):
            dir_x = dir_x + self.ls_solver.ls_solve(res_old)
            jac_res = eval_jac_T_prod(fit, S_vec * eval_jac_prod(fit, dir_x))
            hess_res = hess_product(fit, dir_x) + self.delta_x_vec * dir_x
            res_old = schur_rhs - (jac_res + hess_res)
        return dir_x


class Schur_KKT_solver_direct(Schur_KKT_solver):
    """schur_direct.jl:3-66 -- an abstract_schur_solver: form_system!, update_delta_vecs!, factor_implementation! and
    solver_schur_rhs are Schur_KKT_solver's (schur.jl:47-87,131-182, all on factor_it); only the direction differs: the
    rhs terms, dy and ds are taken at current_it, and ds comes from the complementarity row."""

    def compute_direction_implementation_b(self):  # schur_direct.jl:32-66
        it, rhs = self.current_it, self.rhs
        y, s = it.y, it.s
        symmetric_primal_rhs = rhs.primal_r + rhs.comp_r / y
        S_vec = y / s
        y_ = rhs.primal_r * S_vec + rhs.comp_r / s
        schur_rhs = rhs.dual_r + eval_jac_T_prod(it, y_)
        d = self.dir
        d.x = self.solver_schur_rhs(schur_rhs)
        d.y = -(eval_jac_prod(it, d.x) - symmetric_primal_rhs) * S_vec
        d.s = (rhs.comp_r - d.y * s) / y
        self.update_kkt_error_b()


class Symmetric_KKT_solver(_KKTBase):
    """symmetric.jl:2-102"""

    def form_system_b(self, it):  # symmetric.jl:35-53
        B = sp.diags(-it.s / it.y)
        M = sp.bmat([[it.H, it.J.T], [it.J, B]], format="lil")
        self.Q = M
        self.factor_it = it
        self.schur_diag = compute_schur_diag(it)
        self.true_x_diag = np.asarray(M.diagonal())[: it.dim()].copy()
        self.ready = "system_formed"

    def update_delta_vecs_b(self, delta_x_vec, delta_s_vec):  # symmetric.jl:85-102
        self.delta_x_vec, self.delta_s_vec = delta_x_vec, delta_s_vec
        if np.sum(np.abs(delta_s_vec)) > 0.0:
            raise NotImplementedError("not implemented")
        dg = np.asarray(self.Q.diagonal()).copy()
        dg[: len(delta_x_vec)] = self.true_x_diag + delta_x_vec
        self.Q.setdiag(dg)
        self.ready = "delta_updated"

    def factor_implementation_b(self):  # symmetric.jl:55-57
        return self.ls_solver.ls_factor_b(self.Q.tocsc(), self.factor_it.dim(), self.factor_it.ncon())

    def compute_direction_implementation_b(self):  # symmetric.jl:59-83
        fi, rhs = self.factor_it, self.rhs
        symmetric_rhs = np.concatenate([rhs.dual_r, rhs.primal_r + rhs.comp_r / fi.y])
        sol = self.ls_solver.ls_solve(symmetric_rhs)
        n = len(rhs.dual_r)
        d = self.dir
        d.x = sol[:n]
        d.y = -sol[n:]
        d.s = eval_jac_prod(fi, d.x) - rhs.primal_r
        self.update_kkt_error_b()


# ---------------------------------------------------------------------------------------------------------------
# Clever_Symmetric_KKT_solver (clever_symmetric.jl): parallel rows of J are merged before the LDL^T.
# Index work (integer outputs pinned by the reference's test_compute_indicies / test_compare_columns,
# test/kkt_system_solvers.jl:5-58) is restated literally; indices are 0-based here, the tests add 1.
# ---------------------------------------------------------------------------------------------------------------
def _cols_of(A):
    """Columns of a CSC matrix as (sorted index array, value array) pairs, explicit zeros dropped like Julia's
    sparse() constructor does for the test matrices."""
    A = sp.csc_matrix(A)
    A.sort_indices()
    out = []
    for j in range(A.shape[1]):
        idx = A.indices[A.indptr[j]:A.indptr[j + 1]]
        val = A.data[A.indptr[j]:A.indptr[j + 1]]
        out.append((np.array(idx, dtype=np.int64), np.array(val, dtype=float)))
    return out


def rescale_cols(cols):  # clever_symmetric.jl:90-105: every column divided by its first stored value
    return [(idx, val / val[0] if len(val) else val) for idx, val in cols]


def compare_columns(cols, i, j):  # clever_symmetric.jl:107-155: the strict order "column i before column j"
    ai, vi = cols[i]
    aj, vj = cols[j]
    if len(aj) == 0:
        return False
    if len(ai) == 0:
        return True
    if ai[0] != aj[0]:
        return bool(aj[0] < ai[0])            # the LARGER first row index sorts first
    if len(ai) != len(aj):
        return len(ai) < len(aj)
    for a, b in zip(ai, aj):
        if a != b:
            return bool(a < b)
    for a, b in zip(vi, vj):
        if a < b:
            return True
        if a > b:
            return False
    return i < j


def sorted_col_list(cols):  # clever_symmetric.jl:157-166
    import functools
    rc = rescale_cols(cols)
    order = list(range(len(cols)))
    order.sort(key=functools.cmp_to_key(lambda i, j: -1 if compare_columns(rc, i, j) else (1 if compare_columns(rc, j, i) else 0)))
    return order


def columns_are_same(cols, i, j):  # clever_symmetric.jl:63-88 (on the UNscaled matrix, tolerance 1e-16 in the 2-norm)
    ai, vi = cols[i]
    aj, vj = cols[j]
    if len(ai) != len(aj) or not np.array_equal(ai, aj):
        return False
    if len(vi) == 0:
        return True
    ratio = vi[0] / vj[0]
    return bool(np.sqrt(np.sum((vi - vj * ratio) ** 2)) < 1e-16)


def compute_breakpoints(cols, sorted_cols):  # clever_symmetric.jl:168-198
    bps = []
    for bp in range(len(sorted_cols)):
        if bp == 0 or not columns_are_same(cols, sorted_cols[bp - 1], sorted_cols[bp]):
            bps.append(bp)
    return bps


@dataclass
class Parallel_row:   # clever_symmetric.jl:4-13
    ind: int
    ratio: float
    u: float = math.nan
    g: float = math.nan


@dataclass
class Parallel_row_group:   # clever_symmetric.jl:15-19
    ls: list
    first: int
    u: float = math.nan


def compute_indicies(J, diag_vals=None):  # clever_symmetric.jl:200-260
    cols = _cols_of(sp.csc_matrix(J).T)          # columns of J' = rows of J
    sorted_cols = sorted_col_list(cols)
    bps = compute_breakpoints(cols, sorted_cols)
    no_para = sorted(sorted_cols[b] for b in bps)
    groups = []
    for k, bp in enumerate(bps):
        end = bps[k + 1] if k + 1 < len(bps) else len(sorted_cols)
        ind_ls = sorted_cols[bp:end]
        ls = []
        for i in ind_ls:
            if i == ind_ls[0]:
                ratio = 1.0
            else:
                ratio = cols[i][1][0] / cols[ind_ls[0]][1][0] if len(cols[i][1]) else 1.0
                if ratio == 0.0 or not math.isfinite(ratio):
                    raise ArithmeticError(f"clever_symmetric.jl: ratio = {ratio}")
            ls.append(Parallel_row(i, ratio))
        groups.append(Parallel_row_group(ls, ind_ls[0]))
    groups.sort(key=lambda g: g.first)
    if diag_vals is not None:
        update_indicies(groups, diag_vals)
    return no_para, groups


def update_indicies(groups, diag_vals):  # clever_symmetric.jl:262-287
    for grp in groups:
        u_inv = 0.0
        for row in grp.ls:
            row.u = float(diag_vals[row.ind])
            u_inv += row.ratio ** 2 * row.u ** (-1.0)
        if not u_inv > 0.0:
            raise ArithmeticError(f"clever_symmetric.jl: u_inf = {u_inv} !> 0.0")
        grp.u = 1.0 / u_inv
        for row in grp.ls:
            row.g = grp.u * row.ratio * row.u ** (-1.0)


class Clever_Symmetric_KKT_solver(_KKTBase):
    """clever_symmetric.jl:25-519.  kkt_system_rescale in {"none", "u_only", "u_and_x"} (parameters.jl:24-28)."""

    def __init__(self, ls_solver, pars=None, kkt_system_rescale="none"):
        super().__init__(ls_solver, pars)
        self.kkt_system_rescale = kkt_system_rescale

    def initialize_b(self, it):  # clever_symmetric.jl:53-61
        super().initialize_b(it)
        self.first_para_indicies, self.para_row_info = compute_indicies(it.J)

    def form_system_b(self, it):  # clever_symmetric.jl:341-393
        u = it.s / it.y
        assert np.all(u > 0.0)
        update_indicies(self.para_row_info, u)
        J_new = sp.csr_matrix(it.J)[self.first_para_indicies, :]
        u_new = np.array([g.u for g in self.para_row_info])
        n = it.dim()
        M = sp.bmat([[it.H, None], [J_new, -sp.diags(u_new)]], format="csc") if len(u_new) else sp.csc_matrix(it.H)
        self.factor_it = it
        if self.kkt_system_rescale == "none":
            self.diag_rescale = np.ones(n + len(u_new))
        elif self.kkt_system_rescale == "u_only":
            self.diag_rescale = np.concatenate([np.ones(n), it.mu / np.sqrt(u_new)])
        else:
            self.diag_rescale = np.concatenate([np.ones(n) / math.sqrt(1.0 + float(np.max(np.abs(it.x)))), it.mu / np.sqrt(u_new)])
        D = sp.diags(self.diag_rescale)
        self.Q = (D @ M @ D).tolil()
        self.schur_diag = compute_schur_diag(it)
        self.true_x_diag = np.asarray(M.diagonal())[:n].copy()
        self.ready = "system_formed"

    def update_delta_vecs_b(self, delta_x_vec, delta_s_vec):  # clever_symmetric.jl:494-519
        self.delta_x_vec, self.delta_s_vec = delta_x_vec, delta_s_vec
        if np.sum(np.abs(delta_s_vec)) > 0.0:
            raise NotImplementedError("not implemented")
        if np.sum(np.abs(delta_x_vec)) > 0.0:
            dg = np.asarray(self.Q.diagonal()).copy()
            dg[: len(delta_x_vec)] = self.true_x_diag + delta_x_vec   # the UNscaled diagonal, as the reference does
            self.Q.setdiag(dg)
        self.ready = "delta_updated"

    def factor_implementation_b(self):  # clever_symmetric.jl:395-400
        n = self.factor_it.dim()
        return self.ls_solver.ls_factor_b(self.Q.tocsc(), n, len(self.para_row_info))

    def compute_direction_implementation_b(self):  # clever_symmetric.jl:417-492
        fi, rhs = self.factor_it, self.rhs
        symmetric_primal_rhs = rhs.primal_r + rhs.comp_r / fi.y
        m = len(self.para_row_info)
        crhs = np.zeros(m)
        for i, grp in enumerate(self.para_row_info):
            for row in grp.ls:
                crhs[i] += row.g * symmetric_primal_rhs[row.ind]
        my_rhs = np.concatenate([rhs.dual_r, crhs]) * self.diag_rescale
        Qc = sp.csc_matrix(self.Q)
        sol = np.zeros(len(my_rhs))
        for i in range(self.pars.ItRefine_Num):                     # ls_solve with refinement, clever_symmetric.jl:402-415
            err = my_rhs.copy() if i == 0 else my_rhs - vector_product(Qc, sol)
            sol = sol + self.ls_solver.ls_solve(err)
        dir_x_and_y = sol * self.diag_rescale
        n = len(rhs.dual_r)
        d = self.dir
        d.x = dir_x_and_y[:n]
        v = dir_x_and_y[n:]
        y = (fi.s / fi.y) ** (-1.0) * symmetric_primal_rhs
        for i, grp in enumerate(self.para_row_info):
            tmp = -(crhs[i] + grp.u * v[i])
            for row in grp.ls:
                y[row.ind] += row.u ** (-1.0) * row.ratio * tmp
        d.y = y
        d.s = eval_jac_prod(fi, d.x) - rhs.primal_r
        self.update_kkt_error_b()


def estimate_y_tilde(J, g):  # guess-vars.jl:128-169 (cholesky branch), dense restatement
    J = sp.csc_matrix(J)
    n = J.shape[1]
    Hd = 1e-4 * np.eye(n) + (J.T @ J).toarray()
    Lc = np.linalg.cholesky(Hd)
    dx = np.linalg.solve(Lc.T, np.linalg.solve(Lc, -np.asarray(g, dtype=float)))
    return -(J @ dx)


def is_diag_dom(Q):
    """delta_strategy.jl:1-9: false as soon as 3 Q[i,i] < sum(Q[:,i]) + sum(Q[i,:]) for some column i (plain sums of the STORED
    entries -- H is lower-stored, the Schur matrix carries the full J'SJ on top)."""
    Q = sp.csc_matrix(Q)
    col = np.asarray(Q.sum(axis=0)).ravel()
    row = np.asarray(Q.sum(axis=1)).ravel()
    d = Q.diagonal()
    for i in range(Q.shape[1]):
        if 3 * d[i] < col[i] + row[i]:
            return False
    return True


def ipopt_strategy_b(it, kkt_solver, pars=None):
    """delta_strategy.jl:37-114.  Returns (status, num_fac, delta) and the list of deltas tried."""
    pars = pars or kkt_solver.pars
    MAX_IT = 500
    num_fac = 0
    tried = []
    tau = 1.5 * kkt_solver.diag_min()
    delta = pars.delta_zero
    if tau > 0.0:
        tau = 0.0
        inertia = kkt_solver.factor_b(delta)
        tried.append(delta)
        num_fac += 1
        if inertia == 1:
            return "success", num_fac, delta, tried
    for i in range(1, MAX_IT + 1):
        if i == 1:
            if it.delta != 0.0:
                delta = max(pars.delta_min - tau, it.delta * pars.delta_dec)
            else:
                delta = pars.delta_start - tau
        else:
            delta = delta * pars.delta_inc
        inertia = kkt_solver.factor_b(delta)
        tried.append(delta)
        num_fac += 1
        if inertia == 1:
            return "success", num_fac, delta, tried
        if delta > pars.delta_max:
            return "failure", num_fac, delta, tried
    raise RuntimeError("max it")


def pick_KKT_solver(kkt_solver_type, perm=None, pars=None):
    """kkt_system_solver.jl:232-287 with linear_solver_type = :julia"""
    if kkt_solver_type == "symmetric":
        return Symmetric_KKT_solver(linear_solver_ORACLE("symmetric", perm=perm), pars)
    if kkt_solver_type == "schur":
        return Schur_KKT_solver(linear_solver_ORACLE("definite", perm=perm), pars)
    if kkt_solver_type == "schur_direct":
        return Schur_KKT_solver_direct(linear_solver_ORACLE("definite", perm=perm), pars)
    if kkt_solver_type == "clever_symmetric":
        return Clever_Symmetric_KKT_solver(linear_solver_ORACLE("symmetric", perm=perm), pars)
    raise ValueError("pick a solver!")


def step_failure_delta(it, dir, old_delta, pars=None, response_to_failure="lag_delta_inc"):
    """The delta of the refactorisation after a failed step (one_phase.jl:231-242); parameters.jl:209 selects
    :lag_delta_inc.  it.delta is the delta of the failed step, old_delta the one of the previous outer iteration."""
    pars = pars or KKTPars()
    floor = max(pars.delta_start, old_delta * pars.delta_dec)
    if response_to_failure == "lag_delta_inc":
        # Julia float semantics: x / 0 = Inf, 0 / 0 = NaN, norm of an empty vector = 0; max propagates NaN
        g = eval_grad_lag(it, it.mu)
        gnorm = np.float64(np.max(np.abs(g))) if len(g) else np.float64(0.0)
        dnorm = np.float64(np.max(np.abs(dir.x))) if len(dir.x) else np.float64(0.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = float(gnorm / dnorm)
        out = ratio
        for v in (it.delta * pars.delta_inc, floor):
            out = v if (v != v or (out == out and v > out)) else out
        return out
    if response_to_failure == "default":
        return max(it.delta * pars.delta_inc, floor)
    raise ValueError("pars.test.response_to_failure parameter incorrectly set")
