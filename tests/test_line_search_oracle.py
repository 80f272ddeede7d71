"""CPU tests: the oracle's step-side functions (oracle/line_search_oracle.py) against hand-worked cases and the
known answers of tests/golden/make_golden.py (the reference holds no tests or vectors for these functions)."""
import math

import numpy as np
import scipy.sparse as sp

from conftest import iterate_from_record
from oracle import kkt_oracle as KO
from oracle import line_search_oracle as LO


def _dir(rec, ls):
    return KO.Direction(np.array(rec["dx"]), np.array(rec["dy"]), np.array(rec["ds"]), mu=ls["dir_mu"])


def _records(golden):
    by_name = {r["name"]: r for r in [golden["readme_toy"], golden["indef5"], golden["posdiag_indef5"]] + golden["toy_lps"]}
    return [(by_name[ls["name"]], ls) for ls in golden["line_search"]]


def test_simple_max_step_by_hand():
    # ratios -dir ./ (val - lb) = [0.5, 2.0] -> maximum([1; ...]) = 2 -> 1 / 2
    assert LO.simple_max_step(np.array([1.0, 2.0]), np.array([-0.5, -4.0]), np.zeros(2)) == 0.5
    # nothing shrinks: the leading 1.0 wins
    assert LO.simple_max_step(np.array([1.0, 2.0]), np.array([0.5, 4.0]), np.zeros(2)) == 1.0
    assert LO.simple_max_step(np.zeros(0), np.zeros(0), np.zeros(0)) == 1.0
    assert math.isnan(LO.simple_max_step(np.array([1.0]), np.array([math.nan]), np.zeros(1)))
    # val == lb with a shrinking direction: ratio +Inf, step 0
    assert LO.simple_max_step(np.array([1.0]), np.array([-1.0]), np.array([1.0])) == 0.0


def test_dual_bounds_by_hand():
    # one row, s = mu = y = 1, comp_feas = 0.01, dy = 1: ub_dyi = 100 - 1, lb_dyi = 0.01 - 1
    assert LO.dual_bounds([1.0], 1.0, [1.0], [1.0], 0.01) == (0.0, 1.0)
    lb, ub = LO.dual_bounds([1.0], 1.0, [1.0], [-1.0], 0.01)        # ub_dyi = -99, lb_dyi = 0.99
    assert lb == 0.0 and ub == 0.99 / 1.001
    lb, ub = LO.dual_bounds([1.0], 1.0, [0.001], [1.0], 0.01)       # y below mu * comp_feas / s: lb_dyi = 0.009 > 0
    assert lb == (1.0 * 0.01 / 1.0 - 0.001 / 1.0) * 1.001 and ub == 1.0
    # dy == 0 normally gives Inf - Inf = NaN in both bounds and the row is skipped ...
    assert LO.dual_bounds([1.0, 1.0], 1.0, [1.0, 1.0], [0.0, -1.0], 0.01) == (0.0, 0.99 / 1.001)
    # ... but with y < 0 the bounds are +Inf: the interval is reset to (0, -1) and only LATER rows count
    lb, ub = LO.dual_bounds([1.0, 1.0, 1.0], 1.0, [0.001, -1.0, 1.0], [1.0, 0.0, -1.0], 0.01)
    assert (lb, ub) == (0.0, -1.0)
    lb, ub = LO.dual_bounds([1.0, 1.0, 1.0], 1.0, [1.0, -1.0, 0.001], [-1.0, 0.0, 1.0], 0.01)
    assert lb == (0.01 - 0.001) * 1.001 and ub == -1.0
    # overflow to Inf - Inf inside a counted row: isbad -> (0, -1)
    assert LO.dual_bounds([1e-300], 1e300, [1e300], [1e-300], 0.01) == (0.0, -1.0)


def test_line_search_known_answers(golden):
    for rec, ls in _records(golden):
        it = iterate_from_record(rec, KO.Iterate)
        d = _dir(rec, ls)
        m = rec["m"]
        fp, fb = np.full(m, ls["frac_bd_predict"]), np.full(m, ls["frac_bd"])
        step_P = LO.simple_max_step(it.s, d.s, LO.lb_s_predict(it, d, fp, ls["ex"]))
        assert step_P == ls["step_size_P"], ls["name"]
        assert LO.s_bound_ok(it, d, np.array(ls["s_cand"]), fb, ls["ex"]) == ls["s_bound_ok"]
        assert not LO.s_bound_ok(it, d, np.array(ls["s_bad"]), fb, ls["ex"])
        lb, ub = LO.dual_step_range(it, d, np.array(ls["s_cand"]), it.y, ls["mu_cand"], ls["comp_feas"], fb)
        assert (lb, ub) == (ls["dual_lb"], ls["dual_ub"]), ls["name"]
        phi, C_k, P_k, merit = LO.predicted_reduction_terms(it, d, 1.0)
        assert (C_k, P_k) == (ls["C_k"], ls["P_k"])
        assert abs(phi - ls["phi_red"]) <= 1e-12 * max(1.0, abs(ls["phi_red"]))
        assert abs(merit - ls["merit_red"]) <= 1e-12 * max(1.0, abs(ls["merit_red"]))
        cand = iterate_from_record(rec, KO.Iterate)
        cand.s, cand.mu = np.array(ls["s_cand"]), ls["mu_cand"]
        sd = LO.move_dual_step(cand, d, ls["alpha"], lb, ub, 1, ls["scale_D"], ls["scale_mu"])
        assert abs(sd - ls["step_size_D"]) <= 1e-12 * max(1.0, abs(sd)), ls["name"]
        assert LO.move_dual_step(cand, d, ls["alpha"], lb, ub, 0, ls["scale_D"], ls["scale_mu"]) == ub
