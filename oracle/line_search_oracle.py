"""CPU ORACLE (test infrastructure): the vector work either side of a primal-dual step, restated with numpy.

Follows, function by function:
  /root/reference/src/line_search/frac_boundary.jl:3-35     lb_s_thres, lb_s_predict, lb_y, lb_s, simple_max_step
  /root/reference/src/line_search/move.jl:15-17,28-80       the s-bound test of move_primal, dual_bounds
  /root/reference/src/line_search/move.jl:100-118           move_dual's least-squares dual step (dual_ls 1 / 3)
  /root/reference/src/line_search/line_search.jl:84-86      how simple_ls combines dual_bounds with lb_y
  /root/reference/src/utils/eval.jl:11-13,117-120,236-273   comp, eval_grad_phi, phi_predicted_reduction_primal_dual,
                                                            comp_predicted, merit_function_predicted_reduction
The loops that the reference writes as loops (dual_bounds) stay loops: their result depends on the order.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

from .kkt_oracle import eval_grad_lag, eval_jac_prod, eval_jac_T_prod, vector_product


def _norm_inf(v):
    return float(np.max(np.abs(v))) if len(v) else 0.0


def lb_s_thres(it, dir, ex):  # frac_boundary.jl:3-10
    nx = _norm_inf(dir.x)
    x_thres = nx ** ex
    return np.minimum(it.s, nx * x_thres)


def lb_s_predict(it, dir, frac_bd_predict, ex):  # frac_boundary.jl:12-15
    return frac_bd_predict * lb_s_thres(it, dir, ex)


def lb_y(it, dir, frac_bd):  # frac_boundary.jl:17-20
    return frac_bd * it.y * min(1.0, _norm_inf(dir.x))


def lb_s(it, dir, frac_bd, ex):  # frac_boundary.jl:22-28
    return frac_bd * lb_s_thres(it, dir, ex)


def simple_max_step(val, dir, lb):  # frac_boundary.jl:31-35
    with np.errstate(divide="ignore", invalid="ignore"):
        r = -dir / (val - lb)
    ratio = 1.0
    for v in r:              # Julia's maximum propagates NaN
        if v != v:
            ratio = math.nan
            break
        ratio = max(ratio, float(v))
    return 1.0 / ratio


def s_bound_ok(it, dir, s_new, frac_bd, ex):  # move.jl:15-17
    return bool(np.all(s_new >= lb_s(it, dir, frac_bd, ex)))


def _jmax(a, b):  # Julia's max / min return NaN if either argument is NaN
    return math.nan if (a != a or b != b) else max(a, b)


def _jmin(a, b):
    return math.nan if (a != a or b != b) else min(a, b)


def dual_bounds(s, mu, y, dy, comp_feas):  # move.jl:28-80 (s, mu of the candidate)
    lb, ub = 0.0, 1.0
    safety_factor, safety_add = 1.001, 0.0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for i in range(len(y)):
            si, dyi, yi = np.float64(s[i]), np.float64(dy[i]), np.float64(y[i])
            assert si > 0.0
            ub_dyi = float(np.float64(mu) / (np.float64(comp_feas) * si * dyi) - yi / dyi)
            lb_dyi = float(np.float64(mu) * np.float64(comp_feas) / (si * dyi) - yi / dyi)
            if dyi > 0.0:
                lb = _jmax(lb_dyi * safety_factor + safety_add, lb)
                ub = _jmin(ub_dyi / safety_factor - safety_add, ub)
            elif dyi < 0.0:
                lb = _jmax(ub_dyi * safety_factor + safety_add, lb)
                ub = _jmin(lb_dyi / safety_factor - safety_add, ub)
            elif lb_dyi >= 0.0 or ub_dyi <= 0.0:
                lb, ub = 0.0, -1.0
    if not (math.isfinite(lb) and math.isfinite(ub)):
        return 0.0, -1.0
    return lb, ub


def dual_step_range(it, dir, s_cand, y_cand, mu_cand, comp_feas, frac_bd):  # line_search.jl:84-86
    lb, ub = dual_bounds(s_cand, mu_cand, y_cand, dir.y, comp_feas)
    ub = _jmin(ub, simple_max_step(y_cand, dir.y, lb_y(it, dir, frac_bd)))
    return lb, ub


def comp(it):  # eval.jl:11-13
    return it.s * it.y - it.mu


def eval_grad_phi(it, mu):  # eval.jl:117-120
    return eval_grad_lag(it, mu, it.mu / it.s)


def phi_predicted_reduction_primal_dual(it, dir, step_size):  # eval.jl:236-249
    v = eval_jac_prod(it, dir.x)
    J_gain = float(np.dot(v * v, it.y / it.s))
    h_prod = vector_product(it.H, dir.x)
    return step_size * float(np.dot(dir.x, eval_grad_phi(it, it.mu))) + step_size ** 2 * 0.5 * (float(np.dot(dir.x, h_prod)) + J_gain)


def comp_predicted(it, dir, step_size):  # eval.jl:251-255
    return it.s * it.y + dir.y * it.s * step_size + dir.s * it.y * step_size - (it.mu + dir.mu * step_size)


def merit_function_predicted_reduction(it, dir, step_size):  # eval.jl:257-273
    C_k = _norm_inf(comp(it))
    P_k = _norm_inf(comp_predicted(it, dir, step_size))
    comp_penalty = (P_k ** 3 - C_k ** 3) / it.mu ** 2 if it.ncon() > 0 else 0.0
    return phi_predicted_reduction_primal_dual(it, dir, step_size) + comp_penalty


def predicted_reduction_terms(it, dir, step_size):
    """(phi reduction, C_k, P_k, merit reduction) -- the four numbers the device entry point returns."""
    return (phi_predicted_reduction_primal_dual(it, dir, step_size), _norm_inf(comp(it)),
            _norm_inf(comp_predicted(it, dir, step_size)), merit_function_predicted_reduction(it, dir, step_size))


def move_dual_step(new_it, dir, step_size_P, lb, ub, dual_ls, scale_D, scale_mu):  # move.jl:82-118
    """step_size_D of move_dual for pars.ls.move_primal_seperate_to_dual with dual_ls in (1, 3), else ub.
    new_it is the candidate (x, s moved; J, grad re-evaluated; y still the old multipliers)."""
    small_step = max(lb, min(ub, step_size_P))
    if dual_ls not in (1, 3):
        return ub
    dual_res = eval_grad_lag(new_it, new_it.mu)
    q = np.concatenate([scale_D * eval_jac_T_prod(new_it, dir.y), scale_mu * new_it.s * dir.y])
    res = np.concatenate([scale_D * dual_res, -scale_mu * comp(new_it)])
    step_size_D = float(np.sum(res * q) / np.sum(q * q))
    step_size_D = _jmin(step_size_D, ub)
    step_size_D = _jmax(step_size_D, small_step)
    return step_size_D
