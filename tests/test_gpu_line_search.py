"""GPU parity tests, step-side functions through the C ABI (okkt_kkt_max_step_primal, ...).
max / min results are compared bit for bit with the known answers and the oracle, sums to 1e-12."""
import math

import numpy as np
import pytest

from conftest import iterate_from_record
from onephase_jl_amd import kkt_system_solver as KS
from onephase_jl_amd import line_search as LS
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import OkktError
from oracle import kkt_oracle as KO
from oracle import line_search_oracle as LO

pytestmark = pytest.mark.gpu


def _records(golden):
    by_name = {r["name"]: r for r in [golden["readme_toy"], golden["indef5"], golden["posdiag_indef5"]] + golden["toy_lps"]}
    return [(by_name[ls["name"]], ls) for ls in golden["line_search"]]


def formed_solver(it, kind="symmetric", delta=1e-8):
    k = KS.HIP_KKT_solver(kind)
    k.initialize_b(it)
    k.form_system_b(it)
    assert k.factor_b(delta) in (0, 1)
    return k


def close(a, b, tol=1e-12):
    return abs(a - b) <= tol * max(1.0, abs(b))


def test_known_answers_with_the_golden_direction(golden):
    for rec, ls in _records(golden):
        it = iterate_from_record(rec, KS.Class_iterate)
        k = formed_solver(it, delta=rec["delta"])
        m = rec["m"]
        LS.set_direction(k, KS.Class_point(np.array(rec["dx"]), np.array(rec["dy"]), np.array(rec["ds"]), mu=ls["dir_mu"]))
        pars = LS.Class_ls_parameters(fraction_to_boundary_predict_exp=ls["ex"], comp_feas=ls["comp_feas"])
        fp, fb = np.full(m, ls["frac_bd_predict"]), np.full(m, ls["frac_bd"])
        step_P, nx = LS.max_step_primal(k, fp, pars)
        assert (step_P, nx) == (ls["step_size_P"], ls["dx_norm_inf"]), ls["name"]
        assert LS.s_bound_ok(k, ls["s_cand"], fb, pars) == ls["s_bound_ok"]
        assert not LS.s_bound_ok(k, ls["s_bad"], fb, pars)
        cand = iterate_from_record(rec, KS.Class_iterate)
        cand.s, cand.mu = np.array(ls["s_cand"]), ls["mu_cand"]
        lb, ub = LS.dual_step_range(k, cand, fb, pars)
        assert (lb, ub) == (ls["dual_lb"], ls["dual_ub"]), ls["name"]
        phi, C_k, P_k, merit = LS.predicted_reduction_terms(k, 1.0)
        assert (C_k, P_k) == (ls["C_k"], ls["P_k"]), ls["name"]
        assert close(phi, ls["phi_red"]) and close(merit, ls["merit_red"]), ls["name"]
        assert LS.merit_function_predicted_reduction(k, 1.0) == merit
        sd = LS.move_dual_step(k, cand, ls["alpha"], lb, ub, ls["scale_D"], ls["scale_mu"], pars)
        assert close(sd, ls["step_size_D"]), ls["name"]
        pars.dual_ls = 0
        assert LS.move_dual_step(k, cand, ls["alpha"], lb, ub, ls["scale_D"], ls["scale_mu"], pars) == ub
        k.finalize_b()


def _synth_iterates(name, seed):
    prob = synth.make_config(name, seed=seed, well_scaled=True)
    rng = np.random.default_rng(seed + 100)
    n, m = prob["n"], prob["m"]
    kw = dict(x=rng.normal(size=n), y=prob["y"], s=prob["s"], mu=float(np.mean(prob["s"] * prob["y"])), J=prob["J"], H=prob["H"],
              grad=rng.normal(size=n), cons=prob["s"] + 0.01 * rng.normal(size=m), a_norm_penalty_par=1e-4)
    return prob, KS.Class_iterate(**kw), KO.Iterate(**kw)


@pytest.mark.parametrize("name,seed", [("S-small", 1), ("S-C3", 0)])
def test_synthetic_direction_vs_oracle(name, seed):
    # the direction the device computed stays resident; the oracle evaluates the same functions on its download
    prob, it, oit = _synth_iterates(name, seed)
    n, m = prob["n"], prob["m"]
    k = formed_solver(it, delta=1e-6)
    k.kkt_associate_rhs_b(it, KS.Reduct_stable())
    k.compute_direction_b()
    d = KO.Direction(k.dir.x.copy(), k.dir.y.copy(), k.dir.s.copy(), mu=k.dir.mu)
    rng = np.random.default_rng(7)
    fp, fb = rng.uniform(0.05, 0.3, m), rng.uniform(0.05, 0.3, m)
    pars = LS.Class_ls_parameters()
    step_P, nx = LS.max_step_primal(k, fp, pars)
    assert nx == np.max(np.abs(d.x))
    assert step_P == LO.simple_max_step(oit.s, d.s, LO.lb_s_predict(oit, d, fp, 0.5))
    assert 0.0 < step_P <= 1.0
    for alpha in (0.5 * step_P, step_P, 1.5 * step_P):
        s_new = oit.s + alpha * d.s
        assert LS.s_bound_ok(k, s_new, fb, pars) == LO.s_bound_ok(oit, d, s_new, fb, 0.5)
    alpha = 0.5 * step_P
    cand = KS.Class_iterate(x=it.x + alpha * d.x, y=it.y, s=it.s + alpha * d.s, mu=it.mu + alpha * d.mu, J=it.J, H=it.H,
                            grad=it.grad + 0.1 * alpha, cons=it.cons, a_norm_penalty_par=1e-4)
    ocand = KO.Iterate(x=cand.x, y=cand.y, s=cand.s, mu=cand.mu, J=cand.J, H=cand.H, grad=cand.grad, cons=cand.cons, a_norm_penalty_par=1e-4)
    for cf in (0.01, 0.3):
        pars.comp_feas = cf
        assert LS.dual_step_range(k, cand, fb, pars) == LO.dual_step_range(oit, d, ocand.s, ocand.y, ocand.mu, cf, fb)
    for step in (1.0, 0.37):
        got, exp = LS.predicted_reduction_terms(k, step), LO.predicted_reduction_terms(oit, d, step)
        assert got[1] == exp[1] and got[2] == exp[2]
        scale = abs(step * np.dot(d.x, LO.eval_grad_phi(oit, oit.mu))) + abs(exp[0]) + 1.0
        assert abs(got[0] - exp[0]) <= 1e-11 * scale and abs(got[3] - exp[3]) <= 1e-11 * (scale + abs(exp[3]))
    lb, ub = LO.dual_step_range(oit, d, ocand.s, ocand.y, ocand.mu, 0.01, fb)
    got = LS.move_dual_step(k, cand, alpha, lb, ub, 0.7, 1.3, LS.Class_ls_parameters())
    exp = LO.move_dual_step(ocand, d, alpha, lb, ub, 1, 0.7, 1.3)
    assert close(got, exp, 1e-10)
    # twice the same call: identical bits (fixed reduction partition)
    assert LS.predicted_reduction_terms(k, 0.37) == LS.predicted_reduction_terms(k, 0.37)
    k.finalize_b()


def test_reset_nan_and_empty_semantics():
    # dual_bounds' loop is order dependent (a dy == 0 row with infinite bounds resets the interval): rows placed so that
    # the reset sits in the middle of a multi-workgroup launch, and NaN / Inf inputs
    rng = np.random.default_rng(3)
    m, n = 5000, 3
    import scipy.sparse as sp
    J = sp.random(m, n, density=0.5, random_state=1, format="csc") + sp.csc_matrix((np.ones(n), (np.arange(n), np.arange(n))), shape=(m, n))
    H = sp.identity(n, format="csc")
    s, y = rng.uniform(0.5, 2.0, m), rng.uniform(0.5, 2.0, m)
    kw = dict(x=np.zeros(n), y=y, s=s, mu=1.0, J=J.tocsc(), H=H, grad=np.ones(n), cons=s.copy(), a_norm_penalty_par=1e-4)
    it, oit = KS.Class_iterate(**kw), KO.Iterate(**kw)
    k = formed_solver(it, delta=1e-8)
    dy = rng.normal(size=m)
    fb = np.full(m, 0.1)
    pars = LS.Class_ls_parameters(comp_feas=0.01)
    for variant in range(5):
        yc, d_y = y.copy(), dy.copy()
        if variant == 1:      # plain dy == 0 rows: NaN bounds, skipped
            d_y[[10, 2500, 4999]] = 0.0
        elif variant == 2:    # resets at rows 1700 and 3300 (y < 0, dy == 0): only rows behind 3300 count
            d_y[[1700, 3300]] = 0.0; yc[[1700, 3300]] = -1.0
        elif variant == 3:    # reset on the last row
            d_y[m - 1] = -0.0; yc[m - 1] = -2.0
        elif variant == 4:    # NaN in a counted row
            d_y[4000] = math.nan
        d = KO.Direction(np.full(n, 0.3), d_y, rng.normal(size=m), mu=-0.5)
        LS.set_direction(k, KS.Class_point(d.x, d.y, d.s, mu=d.mu))
        cand = KS.Class_iterate(**{**kw, "y": yc})
        got = LS.dual_step_range(k, cand, fb, pars)
        exp = LO.dual_step_range(oit, d, s, yc, 1.0, 0.01, fb)
        assert got == exp or (all(math.isnan(v) for v in (got[1], exp[1])) and got[0] == exp[0]), (variant, got, exp)
        if variant == 2:
            assert exp[1] <= -1.0
    # NaN and Inf in the primal ratio
    d_s = rng.normal(size=m); d_s[123] = math.nan
    LS.set_direction(k, KS.Class_point(np.full(n, 0.3), dy, d_s, mu=0.0))
    assert math.isnan(LS.max_step_primal(k, fb, pars)[0])
    d_s[123] = -math.inf
    LS.set_direction(k, KS.Class_point(np.full(n, 0.3), dy, d_s, mu=0.0))
    assert LS.max_step_primal(k, fb, pars)[0] == 0.0
    k.finalize_b()


def test_no_constraints_and_call_order():
    import scipy.sparse as sp
    n = 4
    H = sp.csc_matrix(np.tril(np.array([[4.0, 1, 0, 0], [1, 3, 0, 0], [0, 0, 2, 0], [0, 0, 0, 5]])))
    kw = dict(x=np.zeros(n), y=np.zeros(0), s=np.zeros(0), mu=0.5, J=sp.csc_matrix((0, n)), H=H, grad=np.arange(1.0, n + 1), cons=np.zeros(0),
              a_norm_penalty_par=1e-4)
    it, oit = KS.Class_iterate(**kw), KO.Iterate(**kw)
    k = KS.HIP_KKT_solver("schur")
    k.initialize_b(it)
    k.form_system_b(it)
    with pytest.raises(OkktError):           # no direction yet for this system
        LS.max_step_primal(k, np.zeros(0))
    assert k.factor_b(0.0) == 1
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    k.compute_direction_b()
    d = KO.Direction(k.dir.x, k.dir.y, k.dir.s, mu=k.dir.mu)
    assert LS.max_step_primal(k, np.zeros(0)) == (1.0, float(np.max(np.abs(d.x))))
    assert LS.s_bound_ok(k, np.zeros(0), np.zeros(0))
    assert LS.dual_step_range(k, it, np.zeros(0)) == (0.0, 1.0)
    got, exp = LS.predicted_reduction_terms(k, 1.0), LO.predicted_reduction_terms(oit, d, 1.0)
    assert got[1] == 0.0 and got[2] == 0.0 and close(got[0], exp[0]) and close(got[3], exp[3])
    k.form_system_b(it)                      # a new system invalidates the direction
    with pytest.raises(OkktError):
        LS.predicted_reduction_terms(k, 1.0)
    k.finalize_b()
