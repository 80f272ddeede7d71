"""Pins the CPU oracle: against the reference's own unit-test matrices and acceptance rules, against
dense LAPACK known answers (tests/golden), and against dense numpy on seeded random sparse systems.
CPU only."""
import math
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import kkt_oracle as KO
from conftest import iterate_from_record


def full_sym(A):
    A = sp.csc_matrix(A)
    return (sp.tril(A) + sp.tril(A, -1).T).toarray()


# ---- reference test/linear_system_solvers.jl:58-116
def run_linear_solvers(A, b, n, m, inertia, x_expected):
    tol = 1e-9
    res = {}
    for sym in ("symmetric", "definite"):
        s = oracle.linear_solver_ORACLE(sym)
        assert s.ls_factor_b(A, n, m) == inertia
        r1 = np.zeros(len(b))
        s.ls_solve_b(b, r1)
        r2 = s.ls_solve(b)
        assert np.array_equal(r1, r2)  # @test res1 == res2
        res[sym] = r1
    assert np.linalg.norm(res["symmetric"] - res["definite"]) < tol
    assert np.linalg.norm(res["symmetric"] - x_expected) < tol
    # A_2 = A + A' with the diagonal of A: upper entries must be ignored (:74-84)
    A2 = sp.lil_matrix(A + A.T)
    A2.setdiag(A.diagonal())
    for sym in ("symmetric", "definite"):
        s = oracle.linear_solver_ORACLE(sym)
        assert s.ls_factor_b(A2.tocsc(), n, m) == inertia
        assert np.linalg.norm(s.ls_solve(b) - res[sym]) < tol


def test_linear_solvers_reference_matrices(golden):
    for rec in golden["linear_solvers"]:
        A = sp.csc_matrix(np.array(rec["A_lower"]))
        run_linear_solvers(A, np.array(rec["b"]), rec["n"], rec["m"], rec["inertia"], np.array(rec["x"]))


def test_inertia_rule():
    # julia.jl:70-90 + linear_system_solvers.jl:48-91
    A = sp.diags([2.0, -1.0, 3.0, -4.0]).tocsc()
    s = oracle.linear_solver_ORACLE("symmetric")
    assert s.ls_factor_b(A, 2, 2) == 1
    assert s.inertia() == (2, 2, 0, 0)
    assert s.ls_factor_b(A, 3, 1) == 0
    assert oracle.linear_solver_ORACLE("definite").ls_factor_b(A, 4, 0) == 0
    Z = sp.csc_matrix(np.array([[1.0, 0.0], [0.0, 0.0]]))
    assert oracle.linear_solver_ORACLE("symmetric").ls_factor_b(Z, 1, 1) == 0  # zero pivot -> 0
    tiny = sp.diags([1.0, 1e-21]).tocsc()
    s = oracle.linear_solver_ORACLE("symmetric")
    assert s.ls_factor_b(tiny, 2, 0) == 0 and s.inertia() == (1, 0, 1, 0)      # tol = 1e-20
    assert oracle.linear_solver_ORACLE("definite").ls_factor_b(tiny, 2, 0) == 1


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 7, 60, 200])
def test_random_sparse_vs_dense(seed, n):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=min(1.0, 4.0 / n), random_state=np.random.RandomState(seed), format="csc")
    A = sp.tril(A, -1) + sp.diags(rng.normal(size=n) + np.sign(rng.normal(size=n)) * 3.0)
    A = sp.csc_matrix(A)
    M = full_sym(A)
    w = np.linalg.eigvalsh(M)
    npos, nneg = int((w > 0).sum()), int((w < 0).sum())
    b = rng.normal(size=n)
    for perm in (None, rng.permutation(n)):
        s = oracle.linear_solver_ORACLE("symmetric", perm=perm)
        assert s.ls_factor_b(A, npos, nneg) == 1          # Sylvester: sign(D) = inertia
        x = s.ls_solve(b)
        assert np.linalg.norm(M @ x - b) <= 1e-9 * max(1.0, np.linalg.norm(b))
        # L D L' reproduces P A P'
        p = s.perm if s.perm is not None else np.arange(n)
        Lm = s.L().toarray() + np.eye(n)
        assert np.allclose(Lm @ np.diag(s.diag()) @ Lm.T, M[np.ix_(p, p)], atol=1e-9)


# ---- known answers for the KKT layer
def _perm_for(rec, kind):
    # CHOLMOD's AMD is not reproducible here; natural order would pivot on the delta-sized (1,1)
    # block first (element growth 1e8 on LPs).  Eliminating the -S/Y block first is the order a
    # minimum-degree heuristic takes on these tiny problems and the one the Schur form implies.
    n, m = rec["n"], rec["m"]
    return np.concatenate([np.arange(n, n + m), np.arange(n)]) if kind == "symmetric" else None


def _solve_kind(rec, kind):
    it = iterate_from_record(rec, KO.Iterate)
    k = KO.pick_KKT_solver(kind, perm=_perm_for(rec, kind))
    k.initialize_b(it)
    k.form_system_b(it)
    inertia = k.factor_b(rec["delta"])
    k.kkt_associate_rhs_b(it, KO.Reduct_affine())
    k.compute_direction_b()
    return inertia, k


def test_rhs_matches_known_answers(golden):
    for rec in golden["toy_lps"] + [golden["readme_toy"], golden["indef5"]]:
        it = iterate_from_record(rec, KO.Iterate)
        rhs = KO.System_rhs.build(it, KO.Reduct_affine())
        assert np.allclose(rhs.dual_r, rec["rD"], rtol=0, atol=1e-14)
        assert np.allclose(rhs.primal_r, rec["rP"], rtol=0, atol=1e-14)
        assert np.allclose(rhs.comp_r, rec["rC"], rtol=0, atol=1e-14)


def test_toy_lps_directions(golden):
    # reference test/kkt_system_solvers.jl:91-181: schur vs symmetric agree to 1e-6; here both must
    # also match the dense known answer
    for rec in golden["toy_lps"]:
        i1, ks = _solve_kind(rec, "schur")
        i2, ky = _solve_kind(rec, "symmetric")
        assert i1 == 1 and i2 == 1
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(ks.dir, a) - getattr(ky.dir, a)) < 1e-6
            assert np.linalg.norm(getattr(ky.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(ks.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
        assert ks.kkt_err_norm.ratio < 1e-8 and ky.kkt_err_norm.ratio < 1e-6


def test_readme_toy_known_answer(golden):
    rec = golden["readme_toy"]
    for kind in ("schur", "symmetric"):
        inertia, k = _solve_kind(rec, kind)
        assert inertia == 1
        assert np.allclose(k.dir.x, rec["dx"], rtol=1e-12)
        assert np.allclose(k.dir.y, rec["dy"], rtol=1e-11)
        assert np.allclose(k.dir.s, rec["ds"], rtol=1e-12)
        # K(0) has inertia (0,3,0): rejected
        it = iterate_from_record(rec, KO.Iterate)
        k2 = KO.pick_KKT_solver(kind, perm=_perm_for(rec, kind))
        k2.initialize_b(it)
        k2.form_system_b(it)
        assert k2.factor_b(0.0) == 0


@pytest.mark.parametrize("prob", ["readme_toy", "indef5", "posdiag_indef5"])
def test_delta_loop_traces(golden, prob):
    rec = golden[prob]
    loops = rec.get("delta_loops") or {"schur_prev0.0": rec["delta_loop_schur"], "symmetric_prev0.0": rec["delta_loop_symmetric"]}
    for key, exp in loops.items():
        kind, prev = key.split("_prev")
        it = iterate_from_record(rec, KO.Iterate)
        it.delta = float(prev)
        k = KO.pick_KKT_solver(kind, perm=_perm_for(rec, kind))
        k.initialize_b(it)
        k.form_system_b(it)
        status, num_fac, delta, tried = KO.ipopt_strategy_b(it, k)
        assert status == exp["status"] and num_fac == exp["num_fac"]
        assert tried == exp["tried"]          # same arithmetic, bit for bit
        assert delta == exp["delta"]


def test_state_machine_errors(golden):
    rec = golden["toy_lps"][1]
    it = iterate_from_record(rec, KO.Iterate)
    k = KO.pick_KKT_solver("schur")
    k.initialize_b(it)
    k.form_system_b(it)
    with pytest.raises(RuntimeError):
        k.factor_b()             # not :delta_updated (kkt_system_solver.jl:195-199)
    with pytest.raises(RuntimeError):
        k.compute_direction_b()  # not :factored (kkt_system_solver.jl:181-183)


# ---- Clever_Symmetric (clever_symmetric.jl): index work pinned on the reference's own unit-test data
def test_compute_indicies_reference_goldens(golden):
    g = golden["compute_indicies"]
    J = sp.csc_matrix(np.array(g["J"]))
    cols = KO._cols_of(J.T)
    sorted_cols = KO.sorted_col_list(cols)
    assert [c + 1 for c in sorted_cols] == g["sorted_cols"]                       # test/kkt_system_solvers.jl:17-18
    assert [b + 1 for b in KO.compute_breakpoints(cols, sorted_cols)] == g["break_points"]   # :19-20
    u = 1.0 / np.arange(1, J.shape[0] + 1)
    no_para, info = KO.compute_indicies(J, u)
    assert [i + 1 for i in no_para] == g["no_para_indicies"]                      # :24
    assert [grp.first + 1 for grp in info] == g["no_para_indicies"]               # :25-27
    for grp_no, members in g["group_members"].items():                            # :28-33 (exact float equality, as there)
        for pos, ind, ratio in members:
            row = info[int(grp_no) - 1].ls[pos - 1]
            assert row.ind + 1 == ind and row.ratio == ratio
    for grp_no in g["singletons"]:                                                # :36-39
        grp = info[grp_no - 1]
        assert grp.u == u[grp.first] and grp.ls[0].g == 1.0


def test_compare_columns_reference_goldens(golden):
    g = golden["compute_indicies"]
    cols = KO._cols_of(sp.csc_matrix(np.array(g["compare_columns_A_rows"]).T))
    for i, j, expect in g["compare_columns"]:                                     # test/kkt_system_solvers.jl:49-58
        assert KO.compare_columns(cols, i - 1, j - 1) is expect


def _oracle_direction(rec, kind, delta=1e-8, **kw):
    it = iterate_from_record(rec, KO.Iterate)
    k = KO.pick_KKT_solver(kind)
    for a, b in kw.items():
        setattr(k, a, b)
    k.initialize_b(it)
    k.form_system_b(it)
    inertia = k.factor_b(delta)
    k.kkt_associate_rhs_b(it, KO.Reduct_affine())
    k.compute_direction_b()
    return inertia, k


@pytest.mark.parametrize("rescale", ["none", "u_only", "u_and_x"])
def test_clever_symmetric_toy_lps(golden, rescale):
    # test/kkt_system_solvers.jl:141-150: clever_symmetric directions agree with schur to 1e-6 on toy_lp0-8
    merged = 0
    for rec in golden["toy_lps"]:
        i_c, kc = _oracle_direction(rec, "clever_symmetric", kkt_system_rescale=rescale)
        i_s, ks = _oracle_direction(rec, "schur")
        assert i_c == 1 and i_s == 1, rec["name"]
        merged += rec["m"] - len(kc.para_row_info)
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(kc.dir, a) - getattr(ks.dir, a)) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(kc.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
        assert kc.kkt_err_norm.ratio < 1e-8
    assert merged > 0      # the toy LPs do contain parallel rows (two-sided constraints / duplicated bounds)


def test_step_failure_delta_by_hand(golden):
    # one_phase.jl:231-242 on the README toy: grad L_mu = 1 - J'y + mu*pen*J'1 with J = [2x; 1], x = -0.1
    from oracle import kkt_oracle as KO
    it = iterate_from_record(golden["readme_toy"], KO.Iterate)
    glag = 1.0 - (-0.2 * 2.0 + 0.1) + 1.0 * 1e-4 * (-0.2 + 1.0)
    assert abs(KO.eval_grad_lag(it, it.mu)[0] - glag) < 1e-15
    d = KO.Direction(np.array([0.25]), np.zeros(2), np.zeros(2))
    it.delta = 0.01
    assert KO.step_failure_delta(it, d, 0.0) == max(abs(glag) / 0.25, 0.08, 1e-6)            # the gradient term wins
    it.delta = 10.0
    assert KO.step_failure_delta(it, d, 0.0) == 80.0                                            # delta * inc wins
    it.delta = 1e-9
    d.x = np.array([1e9])
    assert KO.step_failure_delta(it, d, 3.0) == 3.0 / math.pi                                   # old_delta * dec wins
    assert KO.step_failure_delta(it, d, 0.0, response_to_failure="default") == 1e-6             # delta.start


def test_step_failure_delta_zero_direction(golden):
    # Julia: norm(g, Inf) / 0.0 = Inf (delta = Inf, the caller's loop then ends with MAX_DELTA); 0 / 0 = NaN and
    # max(NaN, ...) = NaN; an empty x gives norm 0.  No ZeroDivisionError.
    it = iterate_from_record(golden["readme_toy"], KO.Iterate)
    it.delta = 0.01
    assert KO.step_failure_delta(it, KO.Direction(np.zeros(1), np.zeros(2), np.zeros(2)), 0.0) == math.inf
    it.grad = it.grad * 0.0; it.y = it.y * 0.0; it.a_norm_penalty_par = 0.0            # grad L_mu exactly zero
    assert np.all(KO.eval_grad_lag(it, it.mu) == 0.0)
    assert math.isnan(KO.step_failure_delta(it, KO.Direction(np.zeros(1), np.zeros(2), np.zeros(2)), 0.0))


# ---- Schur_KKT_solver_direct (schur_direct.jl:32-66)
def test_schur_direct_directions(golden):
    # factor_it == current_it: the same Newton system as schur / symmetric, ds from the complementarity row instead of
    # the primal row -- the dense known answers pin it
    for rec in golden["toy_lps"] + [golden["readme_toy"]]:
        i_d, kd = _solve_kind(rec, "schur_direct")
        i_s, ks = _solve_kind(rec, "schur")
        assert i_d == 1 and i_s == 1
        for a in ("x", "y", "s"):
            assert np.linalg.norm(getattr(kd.dir, a) - np.array(rec["d" + a])) < 1e-6, (rec["name"], a)
            assert np.linalg.norm(getattr(kd.dir, a) - getattr(ks.dir, a)) < 1e-6, (rec["name"], a)
        assert kd.kkt_err_norm.ratio < 1e-8


def test_schur_direct_reads_current_iterate(golden):
    # current_it != factor_it: dy, ds and the rhs terms of schur_direct use y, s, J of the CURRENT iterate
    # (schur_direct.jl:35-56), the linear system is still the factorised one.  Dense restatement by hand.
    rec = golden["toy_lps"][3]
    it = iterate_from_record(rec, KO.Iterate)
    k = KO.pick_KKT_solver("schur_direct")
    k.initialize_b(it); k.form_system_b(it)
    assert k.factor_b(rec["delta"]) == 1
    cur = iterate_from_record(rec, KO.Iterate)
    rng = np.random.default_rng(0)
    cur.y = cur.y * rng.uniform(0.5, 1.5, size=len(cur.y)); cur.s = cur.s * rng.uniform(0.5, 1.5, size=len(cur.s))
    cur.J = cur.J.copy(); cur.J.data = cur.J.data * rng.uniform(0.9, 1.1, size=cur.J.nnz)
    k.kkt_associate_rhs_b(cur, KO.Reduct_stable())
    k.compute_direction_b()
    r = k.rhs
    Q = (it.J.T @ sp.diags(it.y / it.s) @ it.J + it.H + it.H.T - sp.diags(it.H.diagonal())).toarray() + rec["delta"] * np.eye(it.dim())
    Jc = cur.J.toarray()
    sig = cur.y / cur.s
    dx = np.linalg.solve(Q, r.dual_r + Jc.T @ (r.primal_r * sig + r.comp_r / cur.s))
    dy = -(Jc @ dx - (r.primal_r + r.comp_r / cur.y)) * sig
    ds = (r.comp_r - dy * cur.s) / cur.y
    for got, want in ((k.dir.x, dx), (k.dir.y, dy), (k.dir.s, ds)):
        assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))


# ---- the supernodal multifrontal CPU baseline (okkt_oracle_mf.c) against the simplicial oracle
@pytest.mark.parametrize("name,seed,threads", [("S-tiny", 0, 1), ("S-small", 1, 1), ("S-small", 2, 4)])
def test_multifrontal_baseline_matches_simplicial_oracle(name, seed, threads):
    from onephase_jl_amd import synth
    prob = synth.make_config(name, seed=seed, well_scaled=True)
    n, m = prob["n"], prob["m"]
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n + m)
    for delta, want in ((1e-7, 1), (-50.0, 0)):
        K = synth.augmented_matrix(prob, delta=delta)
        o1 = oracle.linear_solver_ORACLE("symmetric", perm=perm)
        o2 = oracle.linear_solver_ORACLE_MF("symmetric", perm=perm, nthreads=threads)
        assert o1.ls_factor_b(K, n, m) == o2.ls_factor_b(K, n, m) == want
        assert o1.inertia()[:3] == o2.inertia()[:3]
        d1, d2 = o1.diag(), o2.diag()
        assert np.array_equal(np.sign(d1), np.sign(d2)) and np.allclose(d1, d2, rtol=1e-9, atol=0)
        b = rng.normal(size=n + m)
        x1, x2 = o1.ls_solve(b), o2.ls_solve(b)
        assert np.max(np.abs(x1 - x2)) <= 1e-10 * np.max(np.abs(x1))
    # Cholesky semantics on the Schur complement
    Q = synth.schur_matrix(prob, delta=1e-6)
    o1 = oracle.linear_solver_ORACLE("definite"); o2 = oracle.linear_solver_ORACLE_MF("definite", nthreads=threads)
    assert o1.ls_factor_b(Q, n, 0) == o2.ls_factor_b(Q, n, 0) == 1
    b = rng.normal(size=n)
    assert np.max(np.abs(o1.ls_solve(b) - o2.ls_solve(b))) <= 1e-10 * np.max(np.abs(o1.ls_solve(b)))


def test_itrefine_bigfloat_is_a_method_error_in_the_reference(golden):
    # parameters.jl:21 ItRefine_BigFloat: solver_schur_rhs converts dir_x to BigFloat (schur.jl:154-155) and then calls
    # hess_product(fit, dir_x) (schur.jl:167); the only method is hess_product(it, vector::Array{Float64,1}) (eval.jl:232), so the
    # option throws a MethodError in the reference.  The restatement (and the product's mirror) fail the same way.
    rec = golden["toy_lps"][0]
    it = iterate_from_record(rec, KO.Iterate)
    k = KO.pick_KKT_solver("schur", pars=KO.KKTPars(ItRefine_BigFloat=True))
    k.initialize_b(it); k.form_system_b(it); k.factor_b(rec["delta"]); k.kkt_associate_rhs_b(it, KO.Reduct_affine())
    with pytest.raises(TypeError):
        k.compute_direction_b()
