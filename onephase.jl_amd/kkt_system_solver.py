"""Host-side mirror of the reference's KKT-system-solver interface, bound to the HIP library.

Mirrors /root/reference/src/kkt_system_solver/kkt_system_solver.jl (abstract_KKT_system_solver:
initialize!, form_system!, factor!, kkt_associate_rhs!, compute_direction!, update_kkt_error!,
pick_KKT_solver), schur.jl / symmetric.jl (the two solver kinds), system_rhs.jl and
src/IPM/delta_strategy.jl (ipopt_strategy!).  `f!` becomes `f_b`.  Everything numeric happens in
libonephase_kkt.so through the level-2 entry points of include/okkt.h; this file only keeps the
reference's state machine (`ready`), field names (`dir`, `rhs`, `kkt_err_norm`, `schur_diag`,
`delta_x_vec`, ...) and error behaviour.
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from . import _lib as L
from .linear_system_solvers import OkktError


@dataclass
class Class_iterate:
    """Subset of Class_iterate + Class_cache the path reads (Class_iterate.jl:4-20,40-84)."""
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    mu: float
    J: sp.csc_matrix           # m x n
    H: sp.csc_matrix           # n x n, lower triangle only (Class_cutest.jl:548)
    grad: np.ndarray
    cons: np.ndarray
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


def Reduct_affine():   # system_rhs.jl:16-19
    return Class_reduction_factors(0.0, 0.0, 0.0)


def Reduct_stable():   # system_rhs.jl:21-24
    return Class_reduction_factors(1.0, 0.0, 1.0)


@dataclass
class System_rhs:      # system_rhs.jl:34-73
    dual_r: np.ndarray
    primal_r: np.ndarray
    comp_r: np.ndarray


@dataclass
class Class_kkt_error:  # kkt_system_solver.jl:49-65
    error_D: float = 0.0
    error_P: float = 0.0
    error_mu: float = 0.0
    overall: float = 0.0
    rhs_norm: float = 0.0
    ratio: float = 0.0


@dataclass
class Class_point:      # Class_point.jl:1-12 (as a direction)
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    mu: float = 0.0
    primal_scale: float = 0.0


@dataclass
class Class_kkt_solver_options:   # parameters.jl:4-46 (the entries the path reads)
    kkt_solver_type: str = "schur"
    linear_solver_type: str = "HIP"
    ItRefine_Num: int = 3
    # parameters.jl:6,21 (default false).  With true, solver_schur_rhs keeps dir_x in BigFloat (schur.jl:154-155) and then calls
    # hess_product(fit, dir_x) (schur.jl:167), whose only method takes vector::Array{Float64,1} (eval.jl:232): a MethodError.
    # The option cannot run in the reference; the mirror keeps the field and fails the same way instead of inventing a meaning.
    ItRefine_BigFloat: bool = False
    kkt_system_rescale: str = "none"    # parameters.jl:24 (:none | :u_only | :u_and_x), clever_symmetric only
    # The HIP back-end's knobs, carried the way the reference carries a back-end's own option (`ma97_u`, parameters.jl:13,25, handed to
    # linear_solver_HSL by pick_KKT_solver, kkt_system_solver.jl:247): plain fields of pars.kkt, reachable through the JuMP option
    # strings "kkt!hip_ordering" etc. (create_pars_JuMP, JuMPinterface.jl:570-586).  They map one to one onto okkt_opts
    # (include/okkt.h); -1 / 0.0 = the library's default.
    hip_device: int = -1                # okkt_opts.device: HIP device ordinal, -1 = the current device
    hip_ordering: int = 0               # okkt_opts.ordering: 0 automatic, 3 minimum degree always (the reference's class of order), 4 / 5 dissections
    hip_relax_always: int = -1          # okkt_opts.relax_always / relax_small / relax_mid: supernode amalgamation widths
    hip_relax_small: int = -1
    hip_relax_mid: int = -1
    hip_relax_small_frac: float = 0.0   # okkt_opts.relax_small_frac / relax_mid_frac / relax_any_frac: explicit-zero fractions
    hip_relax_mid_frac: float = 0.0
    hip_relax_any_frac: float = 0.0
    hip_inertia_tol: float = 1e-20      # okkt_opts.inertia_tol (julia.jl:73)


def okkt_opts_from_pars(kkt):
    """pars.kkt.hip_* -> the keyword options of HIP_KKT_solver / linear_solver_HIP (= fields of okkt_opts).  Only what differs from
    the library's defaults is passed, so okkt_default_opts stays the single source of the defaults."""
    o = {}
    if kkt.hip_device >= 0:
        o["device"] = int(kkt.hip_device)
    if kkt.hip_ordering != 0:
        o["ordering"] = int(kkt.hip_ordering)
    for name in ("relax_always", "relax_small", "relax_mid"):
        v = getattr(kkt, "hip_" + name)
        if v > 0:
            o[name] = int(v)
    for name in ("relax_small_frac", "relax_mid_frac", "relax_any_frac"):
        v = getattr(kkt, "hip_" + name)
        if v > 0.0:
            o[name] = float(v)
    if kkt.hip_inertia_tol != 1e-20:
        o["inertia_tol"] = float(kkt.hip_inertia_tol)
    return o


@dataclass
class Class_delta_parameters:     # parameters.jl:138-159
    max: float = 1e50
    start: float = 1e-6
    zero: float = 0.0
    min: float = 1e-12
    inc: float = 8.0
    dec: float = 1.0 / math.pi


@dataclass
class Class_parameters:
    kkt: Class_kkt_solver_options = None
    delta: Class_delta_parameters = None
    output_level: int = 0

    def __post_init__(self):
        self.kkt = self.kkt or Class_kkt_solver_options()
        self.delta = self.delta or Class_delta_parameters()


def _julia_max(*vals):
    """max(a, b, c) as Julia evaluates it: NaN wins (Python's max would drop it depending on the position)."""
    out = vals[0]
    for v in vals[1:]:
        out = v if (v != v or (out == out and v > out)) else out
    return out


def _csc(A):
    # (an iterate that already holds canonical CSC -- what the reference's cache holds, Class_iterate.jl:4-20 -- is taken as it is: building a
    # new scipy matrix around the same arrays cost 1 ms per matrix at the metric size, twice per form_system_b)
    if sp.isspmatrix_csc(A) and A.has_sorted_indices:
        return A
    A = sp.csc_matrix(A)
    A.sort_indices()
    return A


class HIP_KKT_solver:
    """abstract_KKT_system_solver backed by the device-resident KKT path
    (kind = 'schur' | 'schur_direct' | 'symmetric' | 'clever_symmetric')."""

    _KINDS = {"schur": L.OKKT_KKT_SCHUR, "symmetric": L.OKKT_KKT_SYMMETRIC, "clever_symmetric": L.OKKT_KKT_CLEVER_SYMMETRIC,
              "schur_direct": L.OKKT_KKT_SCHUR_DIRECT}

    def __init__(self, kind, pars=None, **opts):
        if kind not in self._KINDS:
            raise OkktError("pick a solver!")          # kkt_system_solver.jl:280
        self.kind = kind
        self.pars = pars or Class_parameters()
        self._opts = opts
        self._lib = None
        self._k = None
        self.ready = "not_ready"
        self.kkt_err_norm = Class_kkt_error()
        self.factor_it = None
        self.dir = None
        self.rhs = None
        self.schur_diag = None
        self.delta_x_vec = None
        self.delta_s_vec = None
        self.inertia = None
        self._pattern = None
        self._resident_rhs = None

    # ---- initialize! (kkt_system_solver.jl:21-25)
    def initialize_b(self, intial_it):
        if self._k is None:
            self._lib = L.load()
            o = L.OkktOpts()
            self._lib.okkt_default_opts(C.byref(o))
            for key, v in self._opts.items():
                setattr(o, key, v)
            k = C.c_void_p()
            rc = self._lib.okkt_kkt_create(C.byref(k), C.byref(o), self._KINDS[self.kind])
            if rc != L.OKKT_OK:
                raise OkktError(f"okkt_kkt_create failed with code {rc}"
                                + (" (no HIP device: the KKT path has no CPU fallback)" if rc == L.OKKT_ERR_NO_DEVICE else ""))
            self._k = k
        self.dir = Class_point(np.zeros(intial_it.dim()), np.zeros(intial_it.ncon()), np.zeros(intial_it.ncon()))
        if self.kind == "clever_symmetric" and self._pattern is None:
            # initialize!(::Clever_Symmetric_KKT_solver, it): compute_indicies(get_jac(it)), clever_symmetric.jl:53-61
            self._set_structure(intial_it)
            J = _csc(intial_it.J)
            Jx = L.f64(J.data)
            m_new = C.c_int64()
            self._check(self._lib.okkt_kkt_compute_indicies(self._k, L.p_f64(Jx), C.byref(m_new)), "okkt_kkt_compute_indicies")
            self.m_new = m_new.value
            self.first_para_indicies, self.para_row_info = self.get_indicies()

    def _set_structure(self, it):
        H, J = _csc(it.H), _csc(it.J)
        n, m = it.dim(), it.ncon()
        if self._pattern is None:
            Hp, Hi, Jp, Ji = L.i64(H.indptr), L.i64(H.indices), L.i64(J.indptr), L.i64(J.indices)
            self._check(self._lib.okkt_kkt_set_structure(self._k, n, m, L.p_i64(Hp), L.p_i64(Hi), L.p_i64(Jp), L.p_i64(Ji), 0),
                        "okkt_kkt_set_structure")
            # the analysed pattern: the arrays themselves (kept alive, so `is` stays meaningful) -- a later iterate that carries the same
            # index arrays (new values in the same structure, the usual case) is recognised without touching them; one with arrays of
            # its own is compared entry by entry (round 5: byte strings of all four arrays were built and compared in every call,
            # 1.5 ms of the 1.9 ms form_system_b took at the metric size)
            self._pattern = (n, m, H.indptr, H.indices, J.indptr, J.indices)
            self._m = m
        else:
            pn, pm, hp, hi, jp, ji = self._pattern
            same = (n, m) == (pn, pm)
            for a, b in ((H.indptr, hp), (H.indices, hi), (J.indptr, jp), (J.indices, ji)):
                same = same and (a is b or (a.shape == b.shape and np.array_equal(a, b)))
            if not same:
                raise OkktError("the sparsity pattern of H / J changed: create a new HIP_KKT_solver")
        return H, J, n, m

    def get_indicies(self, with_values=False):
        """(first_para_indicies, para_row_info) of the clever-symmetric solver, 0-based; para_row_info[g] =
        dict(first, u, ls=[dict(ind, ratio, u, g)]) like Parallel_row_group / Parallel_row (clever_symmetric.jl:4-19)."""
        m = self._m
        mn = self.m_new
        first, gptr, ind = np.zeros(mn, np.int64), np.zeros(mn + 1, np.int64), np.zeros(m, np.int64)
        ratio, mu, mg, gu = np.zeros(m), np.full(m, np.nan), np.full(m, np.nan), np.full(mn, np.nan)
        self._check(self._lib.okkt_kkt_get_indicies(self._k, L.p_i64(first), L.p_i64(gptr), L.p_i64(ind), L.p_f64(ratio),
                                                    L.p_f64(mu) if with_values else None, L.p_f64(mg) if with_values else None,
                                                    L.p_f64(gu) if with_values else None), "okkt_kkt_get_indicies")
        info = [dict(first=int(first[g]), u=float(gu[g]),
                     ls=[dict(ind=int(ind[t]), ratio=float(ratio[t]), u=float(mu[t]), g=float(mg[t])) for t in range(gptr[g], gptr[g + 1])])
                for g in range(mn)]
        return [int(v) for v in first], info

    def finalize_b(self):
        if self._k is not None:
            self._lib.okkt_kkt_destroy(self._k)
            self._k = None

    def __del__(self):
        try:
            self.finalize_b()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            msg = self._lib.okkt_kkt_last_error(self._k)
            raise OkktError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    # ---- form_system! (schur.jl:47-62, symmetric.jl:35-53)
    def form_system_b(self, it, timer=None):
        if self._k is None:
            raise OkktError("initialize_b has not been called")
        H, J, n, m = self._set_structure(it)
        if self.kind == "clever_symmetric":
            if not np.all(it.s / it.y > 0.0):
                raise OkktError("assert all(u .> 0.0)")          # clever_symmetric.jl:352
            mode = L.OKKT_RESCALE[self.pars.kkt.kkt_system_rescale]
            self._check(self._lib.okkt_kkt_set_rescale(self._k, mode, float(it.mu), float(np.max(np.abs(it.x))) if n else 0.0), "okkt_kkt_set_rescale")
        Hx, Jx, s, y = L.f64(H.data), L.f64(J.data), L.f64(it.s), L.f64(it.y)
        self._check(self._lib.okkt_kkt_form_system(self._k, L.p_f64(Hx), L.p_f64(Jx), L.p_f64(s), L.p_f64(y)), "okkt_kkt_form_system")
        sd = np.zeros(n)
        self._check(self._lib.okkt_kkt_get_schur_diag(self._k, L.p_f64(sd)), "okkt_kkt_get_schur_diag")
        self.schur_diag = sd
        self.factor_it = it
        self.ready = "system_formed"

    def diag_min(self):   # kkt_system_solver.jl:291-294
        out = C.c_double()
        self._check(self._lib.okkt_kkt_diag_min(self._k, C.byref(out)), "okkt_kkt_diag_min")
        return out.value

    # ---- update_delta! / factor! (kkt_system_solver.jl:98-113,190-204)
    def update_delta_b(self, delta_x, delta_s=0.0, timer=None):
        self.delta_x_vec = delta_x * np.ones(self.factor_it.dim())
        self.delta_s_vec = delta_s * self.factor_it.s ** (-2.0)
        if np.sum(np.abs(self.delta_s_vec)) > 0.0:
            raise OkktError("Not implemented")          # schur.jl:71, symmetric.jl:92
        if self.ready == "not_ready":
            raise OkktError("form_system! has not been called")
        self._delta = float(delta_x)
        self.ready = "delta_updated"

    def factor_b(self, delta_x=None, delta_s=0.0, timer=None, trial=False):
        """factor!(kkt_solver, delta_x[, delta_s]) -> inertia flag.  trial=True is the delta loop's use of it: the caller
        discards a factorisation whose flag is 0, so it may stop early (okkt_kkt_factor_trial); the default is a complete
        factorisation that a direction can be computed from whatever the flag (one_phase.jl:231-242)."""
        if delta_x is not None:
            self.update_delta_b(delta_x, delta_s)
        if self.ready != "delta_updated":
            raise OkktError(f"kkt solver not ready to factor kkt_solver.ready = {self.ready} != :delta_updated")
        self.ready = "factored"
        inert = L.OkktInertia()
        fn = self._lib.okkt_kkt_factor_trial if trial else self._lib.okkt_kkt_factor
        rc = self._check(fn(self._k, self._delta, C.byref(inert)), "okkt_kkt_factor")
        self.inertia = inert.as_tuple()
        return int(rc)

    def timers(self):
        """Device times of the phases of the last calls (okkt_kkt_timers; SURVEY.md section 5, tracing)."""
        t = L.OkktKktTimers()
        self._check(self._lib.okkt_kkt_get_timers(self._k, C.byref(t)), "okkt_kkt_get_timers")
        return t.as_dict()

    # ---- kkt_associate_rhs! (kkt_system_solver.jl:167-176, schur.jl:34-45)
    def kkt_associate_rhs_b(self, it, eta, timer=None):
        n, m = it.dim(), it.ncon()
        # J of the iterate the system was formed at is already resident: NULL = use it (no 11 MB upload at S-metric);
        # a different iterate (correction steps, one_phase.jl:262-279) sends its own Jacobian values
        Jx = None if it is self.factor_it else L.f64(_csc(it.J).data)
        grad, cons, s, y = L.f64(it.grad), L.f64(it.cons), L.f64(it.s), L.f64(it.y)
        rD, rP, rC = np.zeros(n), np.zeros(m), np.zeros(m)
        self._check(self._lib.okkt_kkt_system_rhs(self._k, L.p_f64(Jx) if Jx is not None else None, L.p_f64(grad), L.p_f64(cons), L.p_f64(s), L.p_f64(y),
                                                  it.mu, it.a_norm_penalty_par, eta.P, eta.D, eta.mu,
                                                  L.p_f64(rD), L.p_f64(rP), L.p_f64(rC)), "okkt_kkt_system_rhs")
        self.rhs = System_rhs(rD, rP, rC)
        self._resident_rhs = self.rhs
        self.dir.mu = -(1.0 - eta.mu) * it.mu
        self.dir.primal_scale = -(1.0 - eta.P) * it.primal_scale
        self.reduct_factors = eta
        self.current_it = it

    # ---- several directions of the same factor in one pass (probe + candidates of take_step2!, take_step.jl:2-66)
    def compute_directions_b(self, etas):
        """[(dir, kkt_err_norm)] for the reduction factors `etas` at the iterate of the last kkt_associate_rhs_b: System_rhs on
        the device per triple, the solves batched (up to four right-hand sides per sweep over L).  Every returned direction passes
        the checks of compute_direction_b (the BigFloat guard of the Schur solvers, check_for_nan of IPM_tools.jl:32-49); self.dir and
        self.kkt_err_norm are NOT updated -- the caller picks one of the candidates."""
        if self.ready != "factored":
            raise OkktError("kkt solver not ready to compute direction!")
        if self.pars.kkt.ItRefine_BigFloat and self.kind in ("schur", "schur_direct"):
            raise OkktError("MethodError: no method matching hess_product(::Class_iterate, ::Array{BigFloat,1}) "
                            "(schur.jl:167 with pars.kkt.ItRefine_BigFloat = true; eval.jl:232 only accepts Array{Float64,1})")
        n, m = self.factor_it.dim(), self.factor_it.ncon()
        q = len(etas)
        e = L.f64(np.array([[t.P, t.D, t.mu] for t in etas], dtype=float).ravel())
        dx, dy, ds = np.zeros((q, n)), np.zeros((q, m)), np.zeros((q, m))
        err = (L.OkktKktError * q)()
        self._check(self._lib.okkt_kkt_compute_directions(self._k, q, L.p_f64(e), self.pars.kkt.ItRefine_Num,
                                                          L.p_f64(dx), L.p_f64(dy), L.p_f64(ds), err), "okkt_kkt_compute_directions")
        out = []
        for i in range(q):
            d = Class_point(x=dx[i].copy(), y=dy[i].copy(), s=ds[i].copy())
            d.mu = -(1.0 - etas[i].mu) * self.current_it.mu
            d.primal_scale = -(1.0 - etas[i].P) * self.current_it.primal_scale
            for name, v in (("x", d.x), ("y", d.y), ("s", d.s)):       # check_for_nan, IPM_tools.jl:32-49 -- as the single-rhs path
                if not np.all(np.isfinite(v)):
                    raise OkktError(f"NaN in {name} (candidate {i})")
            ke = err[i]
            out.append((d, Class_kkt_error(ke.error_D, ke.error_P, ke.error_mu, ke.overall, ke.rhs_norm, ke.ratio)))
        return out

    # ---- compute_direction! (kkt_system_solver.jl:178-188)
    def compute_direction_b(self, timer=None):
        if self.ready != "factored":
            raise OkktError("kkt solver not ready to compute direction!")
        if self.pars.kkt.ItRefine_BigFloat and self.kind in ("schur", "schur_direct"):
            raise OkktError("MethodError: no method matching hess_product(::Class_iterate, ::Array{BigFloat,1}) "
                            "(schur.jl:167 with pars.kkt.ItRefine_BigFloat = true; eval.jl:232 only accepts Array{Float64,1})")
        n, m = self.factor_it.dim(), self.factor_it.ncon()
        dx, dy, ds = np.zeros(n), np.zeros(m), np.zeros(m)
        err = L.OkktKktError()
        if self.rhs is self._resident_rhs:
            # the triple that kkt_associate_rhs_b left in HBM: nothing crosses PCIe on the way in
            a = (None, None, None)
        else:
            rD, rP, rC = L.f64(self.rhs.dual_r), L.f64(self.rhs.primal_r), L.f64(self.rhs.comp_r)
            a = (L.p_f64(rD), L.p_f64(rP), L.p_f64(rC))
        self._check(self._lib.okkt_kkt_compute_direction(self._k, a[0], a[1], a[2], self.pars.kkt.ItRefine_Num,
                                                         L.p_f64(dx), L.p_f64(dy), L.p_f64(ds), C.byref(err)),
                    "okkt_kkt_compute_direction")
        self.dir.x, self.dir.y, self.dir.s = dx, dy, ds
        self.kkt_err_norm = Class_kkt_error(err.error_D, err.error_P, err.error_mu, err.overall, err.rhs_norm, err.ratio)
        for name, v in (("x", dx), ("y", dy), ("s", ds)):       # check_for_nan, IPM_tools.jl:32-49
            if not np.all(np.isfinite(v)):
                raise OkktError(f"NaN in {name}")

    # ---- the refactorisation after a failed step (one_phase.jl:231-242): delta from the Lagrangian gradient and the direction
    def refactor_after_step_failure_b(self, it, old_delta, response_to_failure="lag_delta_inc", timer=None):
        """set_delta(iter, max(norm(eval_grad_lag(iter, mu), Inf) / norm(dir.x, Inf), delta * inc, max(start, old_delta * dec)));
        inertia = factor!(kkt_solver, delta) -- returns (inertia, delta) and stores delta in it.delta.  The gradient norm is
        taken from the device rhs kernel (dual_r of System_rhs with eta_D = 0, eta_mu = 1 is -grad L_mu)."""
        d = self.pars.delta
        floor = max(d.start, old_delta * d.dec)
        if response_to_failure == "lag_delta_inc":
            n, m = it.dim(), it.ncon()
            Jx = None if it is self.factor_it else L.f64(_csc(it.J).data)
            grad, cons, s, y = L.f64(it.grad), L.f64(it.cons), L.f64(it.s), L.f64(it.y)
            rD, rP, rC = np.zeros(n), np.zeros(m), np.zeros(m)
            self._check(self._lib.okkt_kkt_system_rhs(self._k, L.p_f64(Jx) if Jx is not None else None, L.p_f64(grad), L.p_f64(cons), L.p_f64(s),
                                                      L.p_f64(y), it.mu, it.a_norm_penalty_par, 0.0, 0.0, 1.0, L.p_f64(rD), L.p_f64(rP), L.p_f64(rC)),
                        "okkt_kkt_system_rhs")
            self._resident_rhs = None      # the device rhs now holds this gradient triple, not self.rhs
            # norm(eval_grad_lag, Inf) / norm(dir.x, Inf) with Julia's float semantics: x / 0 = Inf (the delta loop then
            # ends with MAX_DELTA), 0 / 0 = NaN, norm of an empty vector = 0 (one_phase.jl:233)
            gnorm = np.float64(np.max(np.abs(rD))) if n else np.float64(0.0)
            dnorm = np.float64(np.max(np.abs(self.dir.x))) if n else np.float64(0.0)
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = float(gnorm / dnorm)
            delta = _julia_max(ratio, it.delta * d.inc, floor)
        elif response_to_failure == "default":
            delta = max(it.delta * d.inc, floor)
        else:
            raise OkktError("pars.test.response_to_failure parameter incorrectly set")
        it.delta = delta
        return self.factor_b(delta), delta

    # ---- ipopt_strategy! (delta_strategy.jl:37-114): the whole loop runs behind the C ABI
    def ipopt_strategy_b(self, it, timer=None):
        p = L.OkktKktPars()
        self._lib.okkt_kkt_default_pars(C.byref(p))
        d = self.pars.delta
        p.delta_start, p.delta_min, p.delta_max, p.delta_inc, p.delta_dec, p.delta_zero = d.start, d.min, d.max, d.inc, d.dec, d.zero
        nfac = C.c_int32()
        delta = C.c_double()
        rc = self._check(self._lib.okkt_kkt_ipopt_strategy(self._k, float(it.delta), C.byref(p), C.byref(nfac), C.byref(delta)),
                         "okkt_kkt_ipopt_strategy")
        self._delta = delta.value
        self.delta_x_vec = delta.value * np.ones(it.dim())
        self.delta_s_vec = np.zeros(it.ncon())
        self.ready = "factored"
        # delta_strategy.jl:94-98: after a failed attempt on a diagonally dominant x-block the reference prints a warning
        nwarn = C.c_int32()
        if self._lib.okkt_kkt_diag_dom_warnings(self._k, C.byref(nwarn)) == 0:
            self.diag_dom_warnings = int(nwarn.value)
            for _ in range(self.diag_dom_warnings):
                print("WARNING: Inertia calculation incorrect")
        return ("success" if rc == 1 else "failure"), int(nfac.value), float(delta.value)

    def is_diag_dom(self):
        """is_diag_dom(kkt_solver.Q[1:n,1:n]) (delta_strategy.jl:1-9) at the delta of the last factor!, scanned on the device."""
        out = C.c_int32()
        self._check(self._lib.okkt_kkt_is_diag_dom(self._k, C.byref(out)), "okkt_kkt_is_diag_dom")
        return None if out.value < 0 else bool(out.value)

    def estimate_y_tilde_tail(self, g):
        """y = -J (F \\ (-g)) on the device (guess-vars.jl:155-160)."""
        g = L.f64(np.asarray(g, dtype=float))
        y = np.zeros(self._m)
        self._check(self._lib.okkt_kkt_estimate_y_tilde(self._k, L.p_f64(g), L.p_f64(y)), "okkt_kkt_estimate_y_tilde")
        return y

    # ---- diagnostics
    def matrix(self):
        dim, nnz = C.c_int64(), C.c_int64()
        self._check(self._lib.okkt_kkt_get_matrix(self._k, C.byref(dim), C.byref(nnz), None, None, None), "okkt_kkt_get_matrix")
        colptr = np.zeros(dim.value + 1, dtype=np.int64)
        rowval = np.zeros(max(nnz.value, 1), dtype=np.int64)
        val = np.zeros(max(nnz.value, 1))
        self._check(self._lib.okkt_kkt_get_matrix(self._k, C.byref(dim), C.byref(nnz), L.p_i64(colptr), L.p_i64(rowval), L.p_f64(val)),
                    "okkt_kkt_get_matrix")
        return sp.csc_matrix((val[: nnz.value], rowval[: nnz.value], colptr), shape=(dim.value, dim.value))

    def linear_solver_stats(self):
        st = L.OkktStats()
        h = self._lib.okkt_kkt_linear_solver(self._k)
        rc = self._lib.okkt_get_stats(C.c_void_p(h), C.byref(st))
        if rc != 0:
            raise OkktError("okkt_get_stats failed")
        return st.as_dict()

    def linear_solver_perm(self):
        h = C.c_void_p(self._lib.okkt_kkt_linear_solver(self._k))
        n = self.linear_solver_stats()["n"]
        out = np.zeros(n, dtype=np.int64)
        if self._lib.okkt_get_perm(h, L.p_i64(out)) != 0:
            raise OkktError("okkt_get_perm failed")
        return out


    def ls_solve(self, rhs):
        """ls_solve(kkt_solver.ls_solver, rhs, timer) with the factor this KKT solver holds (julia.jl:105-113)."""
        if self.ready != "factored":
            raise OkktError("kkt solver not ready: factor! first")
        h = C.c_void_p(self._lib.okkt_kkt_linear_solver(self._k))
        b = L.f64(np.asarray(rhs, dtype=float))
        x = np.zeros_like(b)
        rc = self._lib.okkt_solve(h, L.p_f64(b), L.p_f64(x), 1)
        if rc != 0:
            raise OkktError(f"okkt_solve failed ({rc})")
        return x


def estimate_y_tilde(J, g, pars=None, **opts):
    """Initialisation-time Cholesky routed through the same handle (guess-vars.jl:128-169):
    M = cholesky(lambda I + J^T J), dx = M^-1 (-g), y = -J dx with lambda = 1e-4.  J'J + lambda I is assembled on the
    device by the Schur kernel (Sigma = I, H = lambda I) and factored with Cholesky semantics; like the reference,
    any failure returns ones(m)."""
    J = _csc(J)
    m, n = J.shape
    lam = 1e-4
    it = Class_iterate(x=np.zeros(n), y=np.ones(m), s=np.ones(m), mu=1.0, J=J, H=sp.identity(n, format="csc") * lam,
                       grad=np.asarray(g, dtype=float), cons=np.zeros(m))
    p = pars or Class_parameters()
    k = HIP_KKT_solver("schur", p, **opts)
    try:
        k.initialize_b(it)
        k.form_system_b(it)
        if k.factor_b(0.0) != 1:
            return np.ones(m)
        return k.estimate_y_tilde_tail(it.grad)     # dx = M \ (-g), y = -J dx: both on the device
    except OkktError:
        return np.ones(m)
    finally:
        k.finalize_b()


def pick_KKT_solver(pars):
    """kkt_system_solver.jl:232-287 with the `linear_solver_type == :HIP` branch."""
    if pars.kkt.linear_solver_type != "HIP":
        raise OkktError("pick a valid solver!")
    if pars.kkt.kkt_solver_type in ("schur", "schur_direct", "symmetric", "clever_symmetric"):
        return HIP_KKT_solver(pars.kkt.kkt_solver_type, pars, **okkt_opts_from_pars(pars.kkt))
    raise OkktError("pick a solver!")
