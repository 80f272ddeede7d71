"""Host-side mirror of the step-side functions of the reference's line search, bound to the HIP library.

Mirrors /root/reference/src/line_search/frac_boundary.jl (lb_s_predict + simple_max_step as used by simple_ls,
line_search.jl:40-41), move.jl (the s-bound test of move_primal :15-17, dual_bounds :28-80, move_dual's step size
:82-118) and src/utils/eval.jl:236-273 (merit_function_predicted_reduction and its parts).  `iter` is always the
iterate the KKT solver was formed at (`kkt_solver.factor_it`) and `dir` the direction of its last
compute_direction_b: both are already resident on the device, so the functions take the KKT solver where the
reference takes (iter, dir).  Everything numeric happens in libonephase_kkt.so (okkt_kkt_max_step_primal, ...);
there is no host fallback.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from .kkt_system_solver import _csc
from .linear_system_solvers import OkktError


@dataclass
class Class_ls_parameters:      # parameters.jl:50-105 (the entries these functions read)
    fraction_to_boundary_predict_exp: float = 0.5
    comp_feas: float = 1.0 / 100.0
    move_primal_seperate_to_dual: bool = True
    dual_ls: int = 1


def _need_dir(kkt_solver):
    if kkt_solver.factor_it is None or kkt_solver.dir.x is None:
        raise OkktError("no direction: form_system_b / factor_b / compute_direction_b first")


def set_direction(kkt_solver, dir):
    """Make `dir` the resident direction (scale_direction, line_search.jl:10-19; corrections): kkt_solver.dir follows."""
    if kkt_solver.factor_it is None:
        raise OkktError("form_system_b first")
    dx, dy, ds = L.f64(dir.x), L.f64(dir.y), L.f64(dir.s)
    kkt_solver._check(kkt_solver._lib.okkt_kkt_set_direction(kkt_solver._k, L.p_f64(dx), L.p_f64(dy), L.p_f64(ds)), "okkt_kkt_set_direction")
    kkt_solver.dir.x, kkt_solver.dir.y, kkt_solver.dir.s = dx.copy(), dy.copy(), ds.copy()
    kkt_solver.dir.mu, kkt_solver.dir.primal_scale = dir.mu, dir.primal_scale


def max_step_primal(kkt_solver, frac_bd_predict, pars=None):
    """step_size_P = simple_max_step(iter.point.s, dir.s, lb_s_predict(iter, dir, pars)) (line_search.jl:40-41).
    Returns (step_size_P, norm(dir.x, Inf))."""
    pars = pars or Class_ls_parameters()
    _need_dir(kkt_solver)
    step, nx = C.c_double(), C.c_double()
    f = L.f64(frac_bd_predict)
    kkt_solver._check(kkt_solver._lib.okkt_kkt_max_step_primal(kkt_solver._k, L.p_f64(f), pars.fraction_to_boundary_predict_exp,
                                                               C.byref(step), C.byref(nx)), "okkt_kkt_max_step_primal")
    return step.value, nx.value


def s_bound_ok(kkt_solver, s_new, frac_bd, pars=None):
    """all(new_it.point.s .>= lb_s(it, dir, pars)) -- move_primal's :s_bound test (move.jl:15-17)."""
    pars = pars or Class_ls_parameters()
    _need_dir(kkt_solver)
    ok = C.c_int32()
    s_new, f = L.f64(s_new), L.f64(frac_bd)
    kkt_solver._check(kkt_solver._lib.okkt_kkt_s_bound_ok(kkt_solver._k, L.p_f64(s_new), L.p_f64(f), pars.fraction_to_boundary_predict_exp,
                                                          C.byref(ok)), "okkt_kkt_s_bound_ok")
    return bool(ok.value)


def dual_step_range(kkt_solver, candidate, frac_bd, pars=None):
    """lb, ub = dual_bounds(candidate, candidate.point.y, dir.y, comp_feas); ub = min(ub, simple_max_step(candidate.point.y,
    dir.y, lb_y(iter, dir, pars))) (line_search.jl:84-86)."""
    pars = pars or Class_ls_parameters()
    _need_dir(kkt_solver)
    lb, ub = C.c_double(), C.c_double()
    s, y, f = L.f64(candidate.s), L.f64(candidate.y), L.f64(frac_bd)
    kkt_solver._check(kkt_solver._lib.okkt_kkt_dual_step_range(kkt_solver._k, L.p_f64(s), L.p_f64(y), float(candidate.mu), pars.comp_feas,
                                                               L.p_f64(f), C.byref(lb), C.byref(ub)), "okkt_kkt_dual_step_range")
    return lb.value, ub.value


def predicted_reduction_terms(kkt_solver, step_size):
    """(phi_predicted_reduction_primal_dual, norm(comp(iter), Inf), norm(comp_predicted(iter, dir, step_size), Inf),
    merit_function_predicted_reduction) (eval.jl:236-273)."""
    _need_dir(kkt_solver)
    it = kkt_solver.factor_it
    out = np.zeros(4)
    g = L.f64(it.grad)
    kkt_solver._check(kkt_solver._lib.okkt_kkt_predicted_reduction(kkt_solver._k, L.p_f64(g), float(it.mu), float(kkt_solver.dir.mu),
                                                                   float(it.a_norm_penalty_par), float(step_size), L.p_f64(out)),
                      "okkt_kkt_predicted_reduction")
    return tuple(float(v) for v in out)


def merit_function_predicted_reduction(kkt_solver, step_size):   # eval.jl:257-273
    return predicted_reduction_terms(kkt_solver, step_size)[3]


def phi_predicted_reduction_primal_dual(kkt_solver, step_size):  # eval.jl:236-249
    return predicted_reduction_terms(kkt_solver, step_size)[0]


def move_dual_step(kkt_solver, new_it, step_size_P, lb, ub, scale_D, scale_mu, pars=None):
    """step_size_D of move_dual (move.jl:82-118); new_it = the candidate with J and grad re-evaluated, y not yet moved."""
    pars = pars or Class_ls_parameters()
    _need_dir(kkt_solver)
    if not pars.move_primal_seperate_to_dual:
        return float(step_size_P)
    if pars.dual_ls == 2:
        raise OkktError("dual_ls == 2 evaluates the KKT error at trial points on the host (eval_kkt_err): not a device function")
    Jx = None if new_it is kkt_solver.factor_it else L.f64(_csc(new_it.J).data)
    g, s, y = L.f64(new_it.grad), L.f64(new_it.s), L.f64(new_it.y)
    out = C.c_double()
    kkt_solver._check(kkt_solver._lib.okkt_kkt_dual_step(kkt_solver._k, L.p_f64(Jx) if Jx is not None else None, L.p_f64(g), L.p_f64(s), L.p_f64(y),
                                                         float(new_it.mu), float(new_it.a_norm_penalty_par), float(step_size_P), float(lb),
                                                         float(ub), int(pars.dual_ls), float(scale_D), float(scale_mu), C.byref(out)),
                      "okkt_kkt_dual_step")
    return out.value
