"""GPU parity tests, KKT-system level.  Reads like the reference's test/kkt_system_solvers.jl:
test_kkt_solver = initialize! -> form_system! -> factor!(kkt, 1e-8) -> kkt_associate_rhs!(Reduct_affine)
-> compute_direction! -> dir, compared between solver kinds, with the oracle and with the golden answers."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import iterate_from_record
from onephase_jl_amd import synth
from onephase_jl_amd import kkt_system_solver as KS
from onephase_jl_amd.linear_system_solvers import OkktError
from oracle import kkt_oracle as KO

pytestmark = pytest.mark.gpu


def test_kkt_solver(rec_or_it, kind, delta=None, Iterate=KS.Class_iterate, **opts):
    pars = KS.Class_parameters()
    pars.kkt.kkt_solver_type = kind
    it = iterate_from_record(rec_or_it, Iterate) if isinstance(rec_or_it, dict) else rec_or_it
    kkt_solver = KS.pick_KKT_solver(pars) if not opts else KS.HIP_KKT_solver(kind, pars, **opts)
    kkt_solver.initialize_b(it)
    kkt_solver.form_system_b(it)
    inertia = kkt_solver.factor_b(1e-8 if delta is None else delta)
    kkt_solver.kkt_associate_rhs_b(it, KS.Reduct_affine())
    kkt_solver.compute_direction_b()
    return inertia, kkt_solver
test_kkt_solver.__test__ = False


def oracle_solver(rec_or_it, kind, delta, perm=None):
    it = iterate_from_record(rec_or_it, KO.Iterate) if isinstance(rec_or_it, dict) else rec_or_it
    k = KO.pick_KKT_solver(kind, perm=perm)
    k.initialize_b(it)
    k.form_system_b(it)
    inertia = k.factor_b(delta)
    k.kkt_associate_rhs_b(it, KO.Reduct_affine())
    k.compute_direction_b()
    return inertia, k


def test_kkt_solvers_toy_lps(golden):
    # test/kkt_system_solvers.jl:91-181: schur vs symmetric directions agree to 1e-6
    for rec in golden["toy_lps"]:
        i_s, ks = test_kkt_solver(rec, "schur")
        i_y, ky = test_kkt_solver(rec, "symmetric")
        assert i_s == 1 and i_y == 1, rec["name"]
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(ks.dir, a) - getattr(ky.dir, a)) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(ks.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(ky.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
        assert ks.kkt_err_norm.ratio < 1e-8
        assert np.allclose(ks.rhs.dual_r, rec["rD"], atol=1e-14) and np.allclose(ks.rhs.primal_r, rec["rP"], atol=1e-14)
        assert np.allclose(ks.rhs.comp_r, rec["rC"], atol=1e-14)


def test_readme_toy(golden):
    rec = golden["readme_toy"]
    for kind in ("schur", "symmetric"):
        inertia, k = test_kkt_solver(rec, kind, delta=rec["delta"])
        assert inertia == 1
        assert np.allclose(k.dir.x, rec["dx"], rtol=1e-12)
        assert np.allclose(k.dir.y, rec["dy"], rtol=1e-11)
        assert np.allclose(k.dir.s, rec["ds"], rtol=1e-12)
        i0, _ = None, None
        it = iterate_from_record(rec, KS.Class_iterate)
        k2 = KS.HIP_KKT_solver(kind)
        k2.initialize_b(it)
        k2.form_system_b(it)
        assert k2.factor_b(0.0) == 0              # K(0) has inertia (0, 3, 0)
        status, num_fac, delta = k2.ipopt_strategy_b(it)
        exp = rec["delta_loop_" + kind]
        assert (status, num_fac, delta) == (exp["status"], exp["num_fac"], exp["delta"])


@pytest.mark.parametrize("prob", ["indef5", "posdiag_indef5"])
def test_delta_loop_traces(golden, prob):
    rec = golden[prob]
    for key, exp in rec["delta_loops"].items():
        kind, prev = key.split("_prev")
        it = iterate_from_record(rec, KS.Class_iterate)
        it.delta = float(prev)
        k = KS.HIP_KKT_solver(kind)
        k.initialize_b(it)
        k.form_system_b(it)
        status, num_fac, delta = k.ipopt_strategy_b(it)
        assert status == exp["status"] and num_fac == exp["num_fac"]     # '#fac' column
        assert delta == exp["delta"]                                     # same scalar arithmetic, bit for bit
        k.kkt_associate_rhs_b(it, KS.Reduct_affine())
        k.compute_direction_b()
        if abs(delta - rec["delta"]) < 1e-12:
            assert np.allclose(k.dir.x, rec["dx"], rtol=1e-8, atol=1e-10)


def test_state_machine_errors(golden):
    rec = golden["toy_lps"][1]
    it = iterate_from_record(rec, KS.Class_iterate)
    k = KS.HIP_KKT_solver("schur")
    k.initialize_b(it)
    k.form_system_b(it)
    with pytest.raises(OkktError):
        k.factor_b()                 # not :delta_updated
    with pytest.raises(OkktError):
        k.compute_direction_b()      # not :factored
    with pytest.raises(OkktError):
        k.factor_b(1e-8, 1.0)        # delta_s != 0: "Not implemented"
    with pytest.raises(OkktError):
        KS.HIP_KKT_solver("clever")
    # pars.kkt.ItRefine_BigFloat = true is a MethodError in the reference (schur.jl:167 -> eval.jl:232): same failure here
    pars = KS.Class_parameters(); pars.kkt.ItRefine_BigFloat = True
    kb = KS.HIP_KKT_solver("schur", pars)
    kb.initialize_b(it); kb.form_system_b(it); kb.factor_b(1e-8); kb.kkt_associate_rhs_b(it, KS.Reduct_affine())
    with pytest.raises(OkktError):
        kb.compute_direction_b()
    kb.finalize_b()


def synth_iterate(prob, Iterate, seed=0):
    rng = np.random.default_rng(seed)
    n, m = prob["n"], prob["m"]
    return Iterate(x=rng.normal(size=n), y=prob["y"].copy(), s=prob["s"].copy(), mu=prob["mu"], J=prob["J"], H=prob["H"],
                   grad=rng.normal(size=n), cons=prob["s"] + 0.1 * rng.normal(size=m), a_norm_penalty_par=1e-4)


@pytest.mark.parametrize("name,seed", [("S-tiny", 0), ("S-small", 0), ("S-small", 3)])
@pytest.mark.parametrize("kind", ["schur", "symmetric"])
def test_synthetic_directions_vs_oracle(name, seed, kind):
    prob = synth.make_config(name, seed=seed, well_scaled=True)
    inertia, k = test_kkt_solver(synth_iterate(prob, KS.Class_iterate, seed), kind, delta=1e-6)
    io, ko = oracle_solver(synth_iterate(prob, KO.Iterate, seed), kind, 1e-6, perm=k.linear_solver_perm())
    assert inertia == io == 1
    # assembled matrix: lower triangle equals the reference-shaped scipy assembly
    A = k.matrix()
    Aref = sp.tril(sp.csc_matrix(ko.Q)).tolil()          # the oracle's copy carries delta on the x-diagonal
    nn = prob["n"]
    dg = Aref.diagonal()
    dg[:nn] -= 1e-6
    Aref.setdiag(dg)
    Aref = Aref.tocsc()
    assert abs(A - Aref).max() <= 1e-12 * max(1.0, abs(Aref).max())
    assert np.allclose(k.schur_diag, ko.schur_diag, rtol=1e-13)
    assert np.allclose(k.rhs.dual_r, ko.rhs.dual_r, rtol=1e-12, atol=1e-13)
    for a in ("x", "y", "s"):
        da, db = getattr(k.dir, a), getattr(ko.dir, a)
        assert np.max(np.abs(da - db)) <= 1e-9 * max(1.0, np.max(np.abs(db))), (a, np.max(np.abs(da - db)))
    assert k.kkt_err_norm.ratio < 1e-8 and ko.kkt_err_norm.ratio < 1e-8
    assert abs(k.kkt_err_norm.rhs_norm - ko.kkt_err_norm.rhs_norm) <= 1e-12 * ko.kkt_err_norm.rhs_norm
    k.finalize_b()


def test_nonconvex_delta_loop_matches_oracle():
    prob = synth.make_config("S-small", seed=5, convex=False, well_scaled=True)
    for kind in ("schur", "symmetric"):
        it = synth_iterate(prob, KS.Class_iterate)
        k = KS.HIP_KKT_solver(kind)
        k.initialize_b(it)
        k.form_system_b(it)
        status, num_fac, delta = k.ipopt_strategy_b(it)
        ito = synth_iterate(prob, KO.Iterate)
        ko = KO.pick_KKT_solver(kind, perm=k.linear_solver_perm())
        ko.initialize_b(ito)
        ko.form_system_b(ito)
        so, nfo, do, tried = KO.ipopt_strategy_b(ito, ko)
        assert (status, num_fac) == (so, nfo) and num_fac >= 1
        assert abs(delta - do) <= 1e-12 * abs(do)
        assert k.linear_solver_stats()["n_analyze_calls"] == 1       # refactor never re-analyses
        k.finalize_b()


def test_second_form_system_same_pattern_new_values():
    prob = synth.make_config("S-small", seed=7, well_scaled=True)
    it = synth_iterate(prob, KS.Class_iterate)
    k = KS.HIP_KKT_solver("symmetric")
    k.initialize_b(it)
    k.form_system_b(it)
    assert k.factor_b(1e-6) == 1
    it2 = synth_iterate(prob, KS.Class_iterate, seed=9)
    it2.s = it2.s * 1.7
    it2.H = it2.H * 0.5
    k.form_system_b(it2)
    assert k.factor_b(1e-6) == 1
    k.kkt_associate_rhs_b(it2, KS.Reduct_stable())
    k.compute_direction_b()
    ito = synth_iterate(prob, KO.Iterate, seed=9)
    ito.s = ito.s * 1.7
    ito.H = ito.H * 0.5
    ko = KO.pick_KKT_solver("symmetric", perm=k.linear_solver_perm())
    ko.initialize_b(ito); ko.form_system_b(ito); ko.factor_b(1e-6)
    ko.kkt_associate_rhs_b(ito, KO.Reduct_stable()); ko.compute_direction_b()
    assert np.max(np.abs(k.dir.x - ko.dir.x)) <= 1e-9 * np.max(np.abs(ko.dir.x))
    assert k.linear_solver_stats()["n_analyze_calls"] == 1


# ---- Clever_Symmetric (clever_symmetric.jl) -----------------------------------------------
def _clever(rec_or_it, rescale="none", delta=None):
    pars = KS.Class_parameters()
    pars.kkt.kkt_solver_type = "clever_symmetric"
    pars.kkt.kkt_system_rescale = rescale
    it = iterate_from_record(rec_or_it, KS.Class_iterate) if isinstance(rec_or_it, dict) else rec_or_it
    k = KS.pick_KKT_solver(pars)
    k.initialize_b(it)
    k.form_system_b(it)
    inertia = k.factor_b(1e-8 if delta is None else delta)
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    k.compute_direction_b()
    return inertia, k


def test_compute_indicies_reference_goldens(golden):
    # test/kkt_system_solvers.jl:5-41: the integer answers of the reference's own unit test, through the C ABI
    g = golden["compute_indicies"]
    J = sp.csc_matrix(np.array(g["J"]))
    m, n = J.shape
    u = 1.0 / np.arange(1, m + 1)
    it = KS.Class_iterate(x=np.zeros(n), y=np.ones(m), s=u.copy(), mu=0.1, J=J, H=sp.csc_matrix((n, n)), grad=np.zeros(n), cons=np.zeros(m))
    pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = "clever_symmetric"
    k = KS.pick_KKT_solver(pars)
    k.initialize_b(it)
    assert [i + 1 for i in k.first_para_indicies] == g["no_para_indicies"]
    assert [grp["first"] + 1 for grp in k.para_row_info] == g["no_para_indicies"]
    for grp_no, members in g["group_members"].items():
        for pos, ind, ratio in members:
            row = k.para_row_info[int(grp_no) - 1]["ls"][pos - 1]
            assert row["ind"] + 1 == ind and row["ratio"] == ratio
    k.form_system_b(it)
    _, info = k.get_indicies(with_values=True)
    for grp_no in g["singletons"]:
        grp = info[grp_no - 1]
        assert grp["u"] == u[grp["first"]] and grp["ls"][0]["g"] == 1.0
    # and everything equals the oracle's restatement
    no_para, oinfo = KO.compute_indicies(J, u)
    assert no_para == k.first_para_indicies
    for a, b in zip(info, oinfo):
        assert a["first"] == b.first and abs(a["u"] - b.u) <= 1e-15 * abs(b.u)
        assert [(r["ind"], r["ratio"]) for r in a["ls"]] == [(r.ind, r.ratio) for r in b.ls]
        assert np.allclose([r["g"] for r in a["ls"]], [r.g for r in b.ls], rtol=1e-15, atol=0)
    k.finalize_b()


@pytest.mark.parametrize("rescale", ["none", "u_only", "u_and_x"])
def test_clever_symmetric_toy_lps(golden, rescale):
    # test/kkt_system_solvers.jl:141-150: clever_symmetric agrees with schur to 1e-6 on toy_lp0-8
    for rec in golden["toy_lps"]:
        i_c, kc = _clever(rec, rescale)
        i_s, ks = test_kkt_solver(rec, "schur")
        assert i_c == 1 and i_s == 1, rec["name"]
        _, ko = oracle_solver(rec, "clever_symmetric", 1e-8)
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(kc.dir, a) - getattr(ks.dir, a)) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(kc.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
        assert kc.kkt_err_norm.ratio < 1e-8
        assert kc.m_new == len(ko.para_row_info)
        assert kc.first_para_indicies == ko.first_para_indicies
        kc.finalize_b(); ks.finalize_b()


def test_clever_symmetric_synthetic_with_duplicated_rows():
    # a well-scaled synthetic KKT whose J carries exact multiples of existing rows (two-sided constraints):
    # grouping, reduced matrix, inertia (n, m_new, 0) and directions against the oracle and the plain symmetric solver
    prob = synth.make_problem(n=300, m=200, seed=5, well_scaled=True)
    rng = np.random.default_rng(7)
    J0 = sp.csr_matrix(prob["J"])
    dup = rng.choice(200, size=60, replace=False)
    scale = rng.choice([-1.0, 2.0, -0.5, 4.0], size=60)
    J = sp.vstack([J0, sp.diags(scale) @ J0[dup, :]]).tocsc()
    m = J.shape[0]
    s = np.concatenate([prob["s"], rng.uniform(0.5, 2.0, 60)]); y = np.concatenate([prob["y"], rng.uniform(0.5, 2.0, 60)])
    n = 300
    it = KS.Class_iterate(x=rng.normal(size=n), y=y, s=s, mu=float(prob["mu"]), J=J, H=prob["H"], grad=rng.normal(size=n),
                          cons=s + 1e-3 * rng.normal(size=m))
    ito = KO.Iterate(x=it.x, y=y, s=s, mu=it.mu, J=J, H=prob["H"], grad=it.grad, cons=it.cons)
    for rescale in ("none", "u_only"):
        i_c, kc = _clever(it, rescale, delta=1e-6)
        assert i_c == 1 and kc.m_new == 200
        ko = KO.pick_KKT_solver("clever_symmetric", perm=kc.linear_solver_perm())
        ko.kkt_system_rescale = rescale
        ko.initialize_b(ito); ko.form_system_b(ito)
        assert ko.factor_b(1e-6) == 1
        ko.kkt_associate_rhs_b(ito, KO.Reduct_affine()); ko.compute_direction_b()
        assert kc.first_para_indicies == ko.first_para_indicies
        A = kc.matrix()                                   # Q without delta (the factorisation adds it as a shift)
        Qo = sp.csc_matrix(sp.tril(sp.csc_matrix(ko.Q)))
        Qo.setdiag(Qo.diagonal() - np.concatenate([1e-6 * np.ones(n), np.zeros(kc.m_new)]))
        assert abs(A - Qo).max() < 1e-12 * abs(Qo).max()
        for a in ("x", "y", "s"):
            ref = getattr(ko.dir, a)
            assert np.max(np.abs(getattr(kc.dir, a) - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), (rescale, a)
        assert kc.kkt_err_norm.ratio < 1e-9
        i_y, ky = test_kkt_solver(it, "symmetric", delta=1e-6)
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(kc.dir, a) - getattr(ky.dir, a)) < 1e-6 * max(1.0, np.linalg.norm(getattr(ky.dir, a)))
        kc.finalize_b(); ky.finalize_b()


def test_estimate_y_tilde_through_the_handle():
    # guess-vars.jl:128-169, cholesky(lambda I + J'J) \ -g routed through the HIP handle
    for seed, (n, m) in enumerate([(40, 25), (300, 420)]):
        prob = synth.make_problem(n=n, m=m, seed=seed, well_scaled=True)
        g = np.random.default_rng(seed).normal(size=n)
        y_hip = KS.estimate_y_tilde(prob["J"], g)
        y_ref = KO.estimate_y_tilde(prob["J"], g)
        assert np.max(np.abs(y_hip - y_ref)) <= 1e-9 * max(1.0, np.max(np.abs(y_ref)))


def test_trial_factorisation_stops_early_complete_factorisation_can_be_solved_with():
    # A strongly nonconvex system.  (a) factor! as the delta loop uses it (trial): the pivot counts of the lower levels
    # already exceed m negative pivots, the top of the tree is skipped (partial counts), the flag is 0 as the oracle's and a
    # direction is refused.  (b) plain factor! (what one_phase.jl:241 calls after a failed step) runs to the end: complete
    # counts equal to the oracle's, flag 0, and a direction CAN be computed from it, as in the reference, where
    # take_step2! follows whatever the flag was.  (c) the delta loop's trace is the oracle's.
    prob = synth.make_config("S-small", seed=2, convex=False, neg_shift=50.0, well_scaled=True)
    it, oit = synth_iterate(prob, KS.Class_iterate, 2), synth_iterate(prob, KO.Iterate, 2)
    n, m = prob["n"], prob["m"]
    k = KS.HIP_KKT_solver("symmetric")
    k.initialize_b(it)
    k.form_system_b(it)
    assert k.factor_b(0.0, trial=True) == 0
    pos, neg, zero, bad = k.inertia
    assert neg > m and pos + neg + zero + bad < n + m          # stopped before the last level
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    with pytest.raises(OkktError):
        k.compute_direction_b()
    ko = KO.pick_KKT_solver("symmetric", perm=k.linear_solver_perm())
    ko.initialize_b(oit)
    ko.form_system_b(oit)
    assert ko.factor_b(0.0) == 0
    # (b)
    k.form_system_b(it)
    assert k.factor_b(0.0) == 0
    assert sum(k.inertia) == n + m and k.inertia[:3] == ko.ls_solver.inertia()[:3]
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    k.compute_direction_b()
    ko.kkt_associate_rhs_b(oit, KO.Reduct_affine())
    ko.compute_direction_b()
    for a in ("x", "y", "s"):
        ref = getattr(ko.dir, a)
        assert np.max(np.abs(getattr(k.dir, a) - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref))), a
    # (c)
    k.form_system_b(it)
    ko.form_system_b(oit)
    status, num_fac, delta = k.ipopt_strategy_b(it)
    ostatus, onum_fac, odelta, _ = KO.ipopt_strategy_b(oit, ko)
    # tau = 1.5 * min(schur_diag) < 0 enters delta: schur_diag is a sum (order differs from scipy's), so delta agrees to rounding
    assert (status, num_fac) == (ostatus, onum_fac) and status == "success"
    assert abs(delta - odelta) <= 1e-13 * abs(odelta)
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    k.compute_direction_b()
    assert k.kkt_err_norm.ratio < 1e-8
    k.finalize_b()


def test_refactor_with_wrong_inertia_still_gives_a_direction(golden):
    # one_phase.jl:241 ignores the flag of the refactorisation after a failed step; take_step2! then calls
    # compute_direction!.  A delta too small for the indefinite H: flag 0, and the direction equals the oracle's.
    rec = golden["indef5"]
    for kind in ("schur", "symmetric"):
        it, oit = iterate_from_record(rec, KS.Class_iterate), iterate_from_record(rec, KO.Iterate)
        k = KS.HIP_KKT_solver(kind)
        k.initialize_b(it); k.form_system_b(it)
        ko = KO.pick_KKT_solver(kind)
        ko.initialize_b(oit); ko.form_system_b(oit)
        small = 1e-3
        assert k.factor_b(small) == ko.factor_b(small) == 0
        k.kkt_associate_rhs_b(it, KS.Reduct_affine()); ko.kkt_associate_rhs_b(oit, KO.Reduct_affine())
        k.compute_direction_b(); ko.compute_direction_b()
        for a in ("x", "y", "s"):
            ref = getattr(ko.dir, a)
            assert np.max(np.abs(getattr(k.dir, a) - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref))), (kind, a)
        k.finalize_b()


def test_refactor_after_step_failure_zero_direction(golden):
    # norm(grad, Inf) / norm(dir.x, Inf) with dir.x = 0: Julia gives Inf, delta = Inf, the factorisation of K + Inf I
    # reports non-finite pivots (flag 0) and the caller's loop ends with MAX_DELTA -- no Python exception on the way
    rec = golden["indef5"]
    it = iterate_from_record(rec, KS.Class_iterate)
    k = KS.HIP_KKT_solver("symmetric")
    k.initialize_b(it); k.form_system_b(it)
    status, num_fac, delta = k.ipopt_strategy_b(it)
    it.delta = delta
    k.kkt_associate_rhs_b(it, KS.Reduct_affine()); k.compute_direction_b()
    k.dir.x = np.zeros_like(k.dir.x)
    oit = iterate_from_record(rec, KO.Iterate); oit.delta = delta
    want = KO.step_failure_delta(oit, KO.Direction(k.dir.x, k.dir.y, k.dir.s), 0.0)
    inertia, got = k.refactor_after_step_failure_b(it, 0.0)
    assert np.isinf(want) and np.isinf(got) and it.delta == got and inertia == 0
    k.finalize_b()


# ---- Schur_KKT_solver_direct (schur_direct.jl; kkt_solver_type = :schur_direct, kkt_system_solver.jl:270-276) ----------
def test_schur_direct_toy_lps(golden):
    # same iterate for factor_it and current_it: the direction solves the same Newton system as schur / symmetric
    for rec in golden["toy_lps"]:
        i_d, kd = test_kkt_solver(rec, "schur_direct")
        i_s, ks = test_kkt_solver(rec, "schur")
        _, ko = oracle_solver(rec, "schur_direct", 1e-8)
        assert i_d == 1 and i_s == 1, rec["name"]
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(kd.dir, a) - getattr(ks.dir, a)) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(kd.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(kd.dir, a) - getattr(ko.dir, a)) < 1e-9 * max(1.0, np.linalg.norm(getattr(ko.dir, a))), (rec["name"], a)
        assert kd.kkt_err_norm.ratio < 1e-8
        kd.finalize_b(); ks.finalize_b()


@pytest.mark.parametrize("name,seed", [("S-tiny", 1), ("S-small", 4)])
def test_schur_direct_current_iterate_differs_from_factor_iterate(name, seed):
    # the case that separates schur_direct from schur (schur_direct.jl:35-56 reads current_it, schur.jl:93-116 factor_it):
    # factor at one iterate, kkt_associate_rhs! at a moved one (new s, y, Jacobian values, gradient): inner iterations of
    # one_phase.jl:262-279.  Both solvers against their oracle restatements, and they must differ from each other.
    prob = synth.make_config(name, seed=seed, well_scaled=True)
    rng = np.random.default_rng(seed + 100)
    dirs = {}
    for kind in ("schur_direct", "schur"):
        it, oit = synth_iterate(prob, KS.Class_iterate, seed), synth_iterate(prob, KO.Iterate, seed)
        pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = kind
        k = KS.pick_KKT_solver(pars)
        k.initialize_b(it); k.form_system_b(it)
        assert k.factor_b(1e-6) == 1
        ko = KO.pick_KKT_solver(kind, perm=k.linear_solver_perm())
        ko.initialize_b(oit); ko.form_system_b(oit)
        assert ko.factor_b(1e-6) == 1
        rng2 = np.random.default_rng(seed + 100)
        def moved(cls, base):
            J2 = base.J.copy(); J2.data = J2.data * (1.0 + 0.05 * rng2.normal(size=J2.nnz))
            return cls(x=base.x + 0.01, y=base.y * rng2.uniform(0.7, 1.4, size=len(base.y)), s=base.s * rng2.uniform(0.7, 1.4, size=len(base.s)),
                       mu=0.5 * base.mu, J=J2, H=base.H, grad=base.grad + 0.1 * rng2.normal(size=len(base.x)), cons=base.cons + 0.01,
                       a_norm_penalty_par=base.a_norm_penalty_par)
        cur = moved(KS.Class_iterate, it)
        rng2 = np.random.default_rng(seed + 100)
        ocur = moved(KO.Iterate, oit)
        k.kkt_associate_rhs_b(cur, KS.Reduct_stable()); ko.kkt_associate_rhs_b(ocur, KO.Reduct_stable())
        assert np.allclose(k.rhs.dual_r, ko.rhs.dual_r, rtol=1e-12, atol=1e-12)
        k.compute_direction_b(); ko.compute_direction_b()
        for a in ("x", "y", "s"):
            ref = getattr(ko.dir, a)
            assert np.max(np.abs(getattr(k.dir, a) - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), (kind, a)
        assert abs(k.kkt_err_norm.ratio - ko.kkt_err_norm.ratio) <= 1e-6 * max(ko.kkt_err_norm.ratio, 1e-12) + 1e-12
        dirs[kind] = (k.dir.y.copy(), k.dir.s.copy())
        k.finalize_b()
    assert np.max(np.abs(dirs["schur_direct"][1] - dirs["schur"][1])) > 1e-6      # ds really comes from another row of the system


def test_resident_rhs_and_timers():
    # compute_direction! with the rhs triple kkt_associate_rhs! left in HBM (NULL pointers) == with the same triple
    # uploaded from the host; the per-phase device timers of the four reference methods are filled
    prob = synth.make_config("S-small", seed=1, well_scaled=True)
    for kind in ("schur", "symmetric"):
        it = synth_iterate(prob, KS.Class_iterate, 1)
        k = KS.HIP_KKT_solver(kind)
        k.initialize_b(it); k.form_system_b(it)
        assert k.factor_b(1e-6) == 1
        k.kkt_associate_rhs_b(it, KS.Reduct_affine())
        k.compute_direction_b()
        d1 = (k.dir.x.copy(), k.dir.y.copy(), k.dir.s.copy())
        k.rhs = KS.System_rhs(k.rhs.dual_r.copy(), k.rhs.primal_r.copy(), k.rhs.comp_r.copy())    # a host copy: uploaded
        k.compute_direction_b()
        for a, b in zip(d1, (k.dir.x, k.dir.y, k.dir.s)):
            assert np.array_equal(a, b)
        t = k.timers()
        assert t["n_solves"] == (3 if kind == "schur" else 1)
        assert t["factor_ms"] > 0 and t["solve_ms"] > 0 and t["assemble_ms"] > 0 and t["kkt_err_ms"] > 0 and t["rhs_ms"] > 0
        assert t["direction_ms"] >= t["solve_ms"] and t["direction_ms"] < 1e3
        k.finalize_b()


@pytest.mark.parametrize("kind", ["schur", "symmetric"])
def test_refactor_after_step_failure(golden, kind):
    # one_phase.jl:231-242: delta <- max(|grad L_mu|_inf / |dx|_inf, 8 delta, max(1e-6, old_delta / pi)); factor!
    rec = golden["indef5"]
    it, oit = iterate_from_record(rec, KS.Class_iterate), iterate_from_record(rec, KO.Iterate)
    k = KS.HIP_KKT_solver(kind)
    k.initialize_b(it)
    k.form_system_b(it)
    status, num_fac, delta = k.ipopt_strategy_b(it)
    assert status == "success"
    it.delta = oit.delta = delta
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    k.compute_direction_b()
    d = KO.Direction(k.dir.x, k.dir.y, k.dir.s)
    for old_delta, response in ((0.0, "lag_delta_inc"), (5.0 * delta, "lag_delta_inc"), (0.0, "default")):
        it.delta = oit.delta = delta
        want = KO.step_failure_delta(oit, d, old_delta, response_to_failure=response)
        inertia, got = k.refactor_after_step_failure_b(it, old_delta, response)
        assert inertia == 1 and it.delta == got
        assert abs(got - want) <= 1e-14 * want and got >= 8.0 * delta
        k.kkt_associate_rhs_b(it, KS.Reduct_affine())
        k.compute_direction_b()                      # the new factorisation is usable
        assert k.kkt_err_norm.ratio < 1e-8
    with pytest.raises(OkktError):
        k.refactor_after_step_failure_b(it, 0.0, "nonsense")
    k.finalize_b()


@pytest.mark.parametrize("kind", ["schur", "symmetric"])
def test_is_diag_dom_scan_matches_the_reference_rule(golden, kind):
    # delta_strategy.jl:1-9,94-98: the scan the reference runs after every failed attempt of the delta loop, here on the device.
    # Instances on both sides of the rule: toy LPs (H = 0: dominant only through delta), a convex H = B'B-like synthetic (dominant
    # columns) and a nonconvex one; several deltas each.
    cases = [iterate_from_record(rec, KS.Class_iterate) for rec in golden["toy_lps"][:3]]
    ocases = [iterate_from_record(rec, KO.Iterate) for rec in golden["toy_lps"][:3]]
    for seed, convex in ((1, True), (2, False)):
        prob = synth.make_config("S-small", seed=seed, convex=convex, well_scaled=True)
        cases.append(synth_iterate(prob, KS.Class_iterate, seed)); ocases.append(synth_iterate(prob, KO.Iterate, seed))
    seen = set()
    for it, oit in zip(cases, ocases):
        n = it.dim()
        k = KS.HIP_KKT_solver(kind)
        k.initialize_b(it); k.form_system_b(it)
        ko = KO.pick_KKT_solver(kind)
        ko.initialize_b(oit); ko.form_system_b(oit)
        for delta in (0.0, 1e-3, 10.0, 1e6):
            k.factor_b(delta); ko.factor_b(delta)
            want = KO.is_diag_dom(sp.csc_matrix(ko.Q)[:n, :n])
            assert k.is_diag_dom() == want, (kind, n, delta)
            seen.add(want)
        k.finalize_b()
    assert seen == {True, False}


@pytest.mark.parametrize("kind", ["schur", "symmetric"])
def test_delta_loop_failure_still_leaves_a_factor_to_solve_with(kind):
    # gertz_init.jl:25-27 calls ipopt_strategy!, kkt_associate_rhs! and compute_direction! without looking at the status: on
    # :failure the reference solves with the failed factor of the last delta.  delta_max far below what the nonconvex H needs
    # forces :failure at the first attempt; the direction must equal the oracle's, computed from the same (wrong-inertia) factor.
    prob = synth.make_config("S-small", seed=2, convex=False, neg_shift=50.0, well_scaled=True)
    it, oit = synth_iterate(prob, KS.Class_iterate, 2), synth_iterate(prob, KO.Iterate, 2)
    pars = KS.Class_parameters(); pars.delta.max = 1e-9
    k = KS.HIP_KKT_solver(kind, pars)
    k.initialize_b(it); k.form_system_b(it)
    # the first delta of the loop is delta_start - tau with tau = 1.5 * diag_min < 0 here: delta_start is chosen so that it comes
    # out as 1.0 -- far too small for a diagonal shifted by -50 (wrong inertia), far above delta_max (:failure at once)
    start = 1.0 + 1.5 * k.diag_min()
    pars.delta.start = start
    status, num_fac, delta = k.ipopt_strategy_b(it)
    opars = KO.KKTPars(delta_max=1e-9, delta_start=start)
    ko = KO.pick_KKT_solver(kind, perm=k.linear_solver_perm(), pars=opars)
    ko.initialize_b(oit); ko.form_system_b(oit)
    ostatus, onum_fac, odelta, _ = KO.ipopt_strategy_b(oit, ko)
    assert status == ostatus == "failure" and num_fac == onum_fac
    assert abs(delta - odelta) <= 1e-12 * abs(odelta)
    k.kkt_associate_rhs_b(it, KS.Reduct_affine()); k.compute_direction_b()       # must not raise
    ko.kkt_associate_rhs_b(oit, KO.Reduct_affine()); ko.compute_direction_b()
    for a in ("x", "y", "s"):
        ref = getattr(ko.dir, a)
        assert np.max(np.abs(getattr(k.dir, a) - ref)) <= 1e-7 * max(1.0, np.max(np.abs(ref))), a
    k.finalize_b()


@pytest.mark.parametrize("kind", ["schur", "schur_direct", "symmetric"])
def test_batched_directions_equal_the_one_by_one_ones(kind):
    # take_step.jl:2-66: the probe (Reduct_affine) and the candidates of take_step2! (gamma-triples, Reduct_stable, the constant
    # (0.2, 0, 0.2)) are right-hand sides of ONE factorised system: okkt_kkt_compute_directions solves them in one pass over L
    # (five triples = one sweep of four + one of one).  Every direction must equal compute_direction!'s for the same triple
    # (1e-9) and the oracle's; the N-err of every triple comes back too.
    prob = synth.make_config("S-small", seed=4, well_scaled=True)
    it, oit = synth_iterate(prob, KS.Class_iterate, 4), synth_iterate(prob, KO.Iterate, 4)
    etas = [KS.Reduct_affine(), KS.Class_reduction_factors(0.3, 0.3, 0.3), KS.Reduct_stable(), KS.Class_reduction_factors(0.2, 0.0, 0.2),
            KS.Class_reduction_factors(0.05, 0.0, 0.05)]
    k = KS.HIP_KKT_solver(kind)
    k.initialize_b(it); k.form_system_b(it)
    assert k.factor_b(1e-6) == 1
    k.kkt_associate_rhs_b(it, etas[0])
    batch = k.compute_directions_b(etas)
    assert len(batch) == len(etas)
    ko = KO.pick_KKT_solver(kind, perm=k.linear_solver_perm())
    ko.initialize_b(oit); ko.form_system_b(oit)
    assert ko.factor_b(1e-6) == 1
    for eta, (d, kerr) in zip(etas, batch):
        k.kkt_associate_rhs_b(it, eta); k.compute_direction_b()
        ko.kkt_associate_rhs_b(oit, KO.Class_reduction_factors(eta.P, eta.D, eta.mu)); ko.compute_direction_b()
        for a in ("x", "y", "s"):
            one, ref, got = getattr(k.dir, a), getattr(ko.dir, a), getattr(d, a)
            scale = max(1.0, np.max(np.abs(ref)))
            assert np.max(np.abs(got - one)) <= 1e-9 * scale, (kind, a)
            assert np.max(np.abs(got - ref)) <= 1e-8 * scale, (kind, a)
        assert abs(kerr.ratio - k.kkt_err_norm.ratio) <= 1e-6 * max(k.kkt_err_norm.ratio, 1e-12) + 1e-14
        assert d.mu == k.dir.mu and d.primal_scale == k.dir.primal_scale
    k.finalize_b()


def test_hip_options_travel_through_pick_KKT_solver():
    # The back-end's knobs are fields of pars.kkt (kkt.hip_*), read by pick_KKT_solver's :HIP branch and handed to okkt_kkt_create as an
    # okkt_opts struct -- the way the reference hands pars.kkt.ma97_u to linear_solver_HSL (src/kkt_system_solver/kkt_system_solver.jl:247;
    # option plumbing src/parameters.jl:4-46, src/JuMPinterface.jl:570-586).  Two knobs whose effect can be read back from the handle:
    # the ordering (ordering_used of the plan) and the zero-pivot tolerance (the inertia counts).
    prob = synth.make_config("S-C3-random-small", seed=0, well_scaled=True)
    it = synth_iterate(prob, KS.Class_iterate, 0)
    seen = {}
    for ordering in (0, 1, 3):
        pars = KS.Class_parameters()
        pars.kkt.kkt_solver_type = "symmetric"
        pars.kkt.hip_ordering = ordering
        k = KS.pick_KKT_solver(pars)
        k.initialize_b(it); k.form_system_b(it)
        assert k.factor_b(1e-8) == 1
        st = k.linear_solver_stats()
        seen[ordering] = (st["ordering_used"], st["nnzL"])
        k.finalize_b()
    assert seen[1][0] == 1 and seen[3][0] == 0                     # natural order / minimum degree, as asked
    assert seen[1][1] > seen[3][1]                                 # ... and it is a different plan: the natural order fills more
    # inertia_tol: with an absurdly large tolerance every pivot counts as zero -> inertia flag 0 (julia.jl:73-78)
    pars = KS.Class_parameters()
    pars.kkt.kkt_solver_type = "symmetric"
    pars.kkt.hip_inertia_tol = 1e300
    k = KS.pick_KKT_solver(pars)
    k.initialize_b(it); k.form_system_b(it)
    assert k.factor_b(1e-8) == 0
    assert k.inertia[2] == prob["n"] + prob["m"], k.inertia
    k.finalize_b()
