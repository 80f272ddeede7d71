"""GPU parity tests, linear-solver level: HIP path (through the C ABI) vs the CPU oracle with the
SAME permutation, the golden known answers, and size-independent properties at larger sizes.
Reads like the reference's test/linear_system_solvers.jl."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import OkktError, finalize_b, initialize_b, linear_solver_HIP

pytestmark = pytest.mark.gpu


def hip_solver(sym, **o):
    s = linear_solver_HIP(sym, False, False, **o)
    initialize_b(s)
    return s


def full_sym(A):
    A = sp.csc_matrix(A)
    return (sp.tril(A) + sp.tril(A, -1).T).tocsc()


# ---- reference test/linear_system_solvers.jl:18-116 on the HIP solver
def _test_hip(sym, A, b, n, m, inertia):
    solver = hip_solver(sym)
    assert inertia == solver.ls_factor_b(A, n, m)
    res1 = np.zeros(len(b))
    solver.ls_solve_b(b, res1)
    res2 = solver.ls_solve(b)
    assert np.array_equal(res1, res2)  # @test res1 == res2
    finalize_b(solver)
    return res1


def run_linear_solvers(A, b, n, m, inertia, x_expected):
    tol = 1e-9
    dir_sym = _test_hip("symmetric", A, b, n, m, inertia)
    dir_chol = _test_hip("definite", A, b, n, m, inertia)
    assert np.linalg.norm(dir_sym - dir_chol) < tol
    assert np.linalg.norm(dir_sym - x_expected) < tol
    A_2 = sp.lil_matrix(A + A.T)
    A_2.setdiag(A.diagonal())
    A_2 = A_2.tocsc()
    assert np.linalg.norm(dir_sym - _test_hip("symmetric", A_2, b, n, m, inertia)) < tol
    assert np.linalg.norm(dir_chol - _test_hip("definite", A_2, b, n, m, inertia)) < tol


def test_linear_solvers(golden):
    for rec in golden["linear_solvers"]:
        A = sp.csc_matrix(np.array(rec["A_lower"]))
        run_linear_solvers(A, np.array(rec["b"]), rec["n"], rec["m"], rec["inertia"], np.array(rec["x"]))


def test_inertia_rule_matches_oracle():
    cases = [
        (sp.diags([2.0, -1.0, 3.0, -4.0]).tocsc(), 2, 2),
        (sp.diags([2.0, -1.0, 3.0, -4.0]).tocsc(), 3, 1),
        (sp.csc_matrix(np.array([[1.0, 0.0], [0.0, 0.0]])), 1, 1),   # exact zero pivot
        (sp.diags([1.0, 1e-21]).tocsc(), 2, 0),                      # below tol = 1e-20
        (sp.diags([1.0, -1e-21, 5.0]).tocsc(), 2, 1),
    ]
    for A, n, m in cases:
        h = hip_solver("symmetric")
        o = oracle.linear_solver_ORACLE("symmetric")
        assert h.ls_factor_b(A, n, m) == o.ls_factor_b(A, n, m)
        finalize_b(h)
    for A in (sp.diags([2.0, -1.0, 3.0]).tocsc(), sp.diags([1.0, 1e-21]).tocsc(), sp.diags([1.0, 0.0]).tocsc()):
        h = hip_solver("definite")
        o = oracle.linear_solver_ORACLE("definite")
        assert h.ls_factor_b(A, A.shape[0], 0) == o.ls_factor_b(A, A.shape[0], 0)
        finalize_b(h)
    # NaN / Inf in the values -> 0 (julia.jl:77-89)
    h = hip_solver("symmetric")
    assert h.ls_factor_b(sp.csc_matrix(np.array([[1.0, 0.0], [0.5, np.nan]])), 2, 0) == 0
    assert h.inertia[3] >= 1
    assert h.ls_factor_b(sp.csc_matrix(np.array([[np.inf, 0.0], [0.5, 1.0]])), 2, 0) == 0
    with pytest.raises(OkktError):
        hip_solver("definite").ls_factor_b(sp.identity(3, format="csc"), 2, 1)   # @assert(m == 0)
    with pytest.raises(OkktError):
        hip_solver("symmetric").ls_factor_b(sp.identity(3, format="csc"), 2, 2)  # n + m != dim


def compare_with_oracle(A, n, m, sym="symmetric", nrhs=2, seed=0, tol=1e-10, **opts):
    rng = np.random.default_rng(seed)
    h = hip_solver(sym, **opts)
    rc = h.ls_factor_b(A, n, m)
    perm = h.perm()
    o = oracle.linear_solver_ORACLE(sym, perm=perm)
    rco = o.ls_factor_b(A, n, m)
    assert rc == rco
    assert h.inertia[:3] == o.inertia(0.0 if sym == "definite" else 1e-20)[:3]   # counts, bit-exact
    d_h, d_o = h.diag(), o.diag()
    assert np.array_equal(np.sign(d_h), np.sign(d_o))
    assert np.allclose(d_h, d_o, rtol=max(1e-9, 10 * tol), atol=0)
    M = full_sym(A)
    for _ in range(nrhs):
        b = rng.normal(size=A.shape[0])
        x = h.ls_solve(b)
        xo = o.ls_solve(b)
        assert np.max(np.abs(x - xo)) <= tol * np.max(np.abs(xo)), np.max(np.abs(x - xo)) / np.max(np.abs(xo))
        assert np.max(np.abs(M @ x - b)) <= 1e-9 * max(1.0, np.max(np.abs(b))) * max(1.0, abs(M).sum(axis=1).max())
    st = h.stats()
    finalize_b(h)
    return st, h, o


@pytest.mark.parametrize("n", [1, 2, 5, 33, 64, 65, 130, 300])
@pytest.mark.parametrize("seed", [0, 1])
def test_random_sparse_vs_oracle(n, seed):
    rng = np.random.default_rng(100 * n + seed)
    A = sp.random(n, n, density=min(1.0, 5.0 / n), random_state=np.random.RandomState(seed), format="csc")
    A = sp.csc_matrix(sp.tril(A, -1) + sp.diags(rng.normal(size=n) + np.sign(rng.normal(size=n)) * 4.0))
    w = np.linalg.eigvalsh(full_sym(A).toarray())
    compare_with_oracle(A, int((w > 0).sum()), int((w < 0).sum()), seed=seed)


@pytest.mark.parametrize("n", [40, 150, 400])
def test_dense_matrix_single_front(n):
    # a dense symmetric matrix is one supernode: exercises the LDS kernel (n <= 128) and the
    # blocked big-front kernels (diag / trsm / MFMA syrk) with ragged tails
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n))
    M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
    A = sp.csc_matrix(np.tril(M))
    w = np.linalg.eigvalsh(M)
    compare_with_oracle(A, int((w > 0).sum()), int((w < 0).sum()), tol=1e-9)


@pytest.mark.parametrize("nb", [16, 32, 64])
def test_big_front_panel_widths(nb):
    n = 200
    rng = np.random.default_rng(7)
    B = rng.normal(size=(n, n))
    M = B @ B.T + n * np.eye(n)
    A = sp.csc_matrix(np.tril(M))
    compare_with_oracle(A, n, 0, sym="definite", panel_nb=nb, tol=1e-9)
    compare_with_oracle(A, n, 0, sym="symmetric", panel_nb=nb, small_front_max=64, tol=1e-9)


@pytest.mark.parametrize("name,seed", [("S-tiny", 0), ("S-small", 0), ("S-small", 1)])
@pytest.mark.parametrize("kind", ["augmented", "schur"])
@pytest.mark.parametrize("well_scaled", [True, False])
def test_synthetic_kkt_vs_oracle(name, seed, kind, well_scaled):
    # tolerance (fp64): 1e-10 relative on well-scaled systems (BASELINE.md parity gate); the default
    # generator spreads s/y over 16 orders of magnitude (late-iteration IPM conditioning), where two
    # correct factorisations with different summation order agree to ~1e-7 only.
    prob = synth.make_config(name, seed=seed, well_scaled=well_scaled)
    n, m = prob["n"], prob["m"]
    tol = 1e-10 if well_scaled else 1e-7
    if kind == "augmented":
        st, h, o = compare_with_oracle(synth.augmented_matrix(prob, delta=1e-8), n, m, "symmetric", tol=tol)
    else:
        st, h, o = compare_with_oracle(synth.schur_matrix(prob, delta=1e-8), n, 0, "definite", tol=tol)
    assert st["n_analyze_calls"] == 1


def test_factor_values_match_oracle_L():
    prob = synth.make_config("S-small", seed=4)
    K = synth.augmented_matrix(prob, delta=1e-6)
    n, m = prob["n"], prob["m"]
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, n, m) == 1
    o = oracle.linear_solver_ORACLE("symmetric", perm=h.perm())
    assert o.ls_factor_b(K, n, m) == 1
    Lh = h.factor_csc()
    Lo = o.L()
    # same structure up to the explicit zeros of relaxed supernodes
    D = (Lh - Lo).tocsc()
    scale = max(1.0, abs(Lo).max())
    assert abs(D).max() <= 1e-9 * scale
    finalize_b(h)


def test_nonconvex_wrong_inertia_then_shift():
    # delta loop shape: same pattern, new values, no re-analysis; flag flips 0 -> 1
    prob = synth.make_config("S-small", seed=6, convex=False, well_scaled=True)
    n, m = prob["n"], prob["m"]
    h = hip_solver("symmetric")
    K0 = synth.augmented_matrix(prob, delta=1e-8)     # keeps the diagonal structurally present
    rc0 = h.ls_factor_b(K0, n, m)
    o = oracle.linear_solver_ORACLE("symmetric", perm=h.perm())
    assert rc0 == o.ls_factor_b(K0, n, m) == 0
    assert h.inertia[:3] == o.inertia()[:3]
    K1 = synth.augmented_matrix(prob, delta=50.0)
    rc1 = h.ls_factor_b(K1, n, m)
    assert rc1 == o.ls_factor_b(K1, n, m) == 1
    assert h.stats()["n_analyze_calls"] == 1
    b = np.ones(n + m)
    assert np.allclose(h.ls_solve(b), o.ls_solve(b), rtol=1e-9, atol=1e-12)
    finalize_b(h)


def test_duplicate_entries_are_summed():
    # SparseMatrixCSC never holds duplicates, but a raw CSC handed over the C ABI may
    colptr = np.array([0, 3, 4], dtype=np.int64)
    rowval = np.array([0, 1, 1, 1], dtype=np.int64)
    nzval = np.array([4.0, 1.0, 0.5, 3.0])
    h = hip_solver("symmetric")
    assert h.ls_factor_b((2, colptr, rowval, nzval, 0), 2, 0) == 1
    x = h.ls_solve(np.array([1.0, 2.0]))
    M = np.array([[4.0, 1.5], [1.5, 3.0]])
    assert np.allclose(M @ x, [1.0, 2.0], atol=1e-12)
    finalize_b(h)


def test_empty_matrix_and_solve_before_factor():
    h = hip_solver("symmetric")
    assert h.ls_factor_b(sp.csc_matrix((0, 0)), 0, 0) == 1
    h2 = hip_solver("symmetric")
    h2.analyze(sp.identity(3, format="csc"))
    with pytest.raises(OkktError):
        h2.ls_solve(np.ones(3))


def test_sc3_size_properties():
    # BASELINE config 3 (n = 1e4, m = 2e4, nnz ~ 5e5): too large for the scalar oracle in a unit
    # test, so check size-independent properties: inertia (n, m, 0), tiny residual, linearity.
    prob = synth.make_config("S-C3", seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, n, m) == 1
    assert h.inertia == (n, m, 0, 0)
    M = full_sym(K)
    rng = np.random.default_rng(1)
    b1, b2 = rng.normal(size=n + m), rng.normal(size=n + m)
    x1, x2, x12 = h.ls_solve(b1), h.ls_solve(b2), h.ls_solve(b1 + 2.0 * b2)
    for b, x in ((b1, x1), (b2, x2)):
        r = M @ x - b
        assert np.max(np.abs(r)) <= 1e-8 * np.max(np.abs(b)) * max(1.0, np.max(np.abs(x)))
    assert np.max(np.abs(x12 - (x1 + 2.0 * x2))) <= 1e-8 * np.max(np.abs(x12))
    st = h.stats()
    print("S-C3 stats:", {k: st[k] for k in ("nnzL", "flops_exact", "nsuper", "nlevels", "max_front", "last_factor_ms", "last_solve_ms", "analyze_seconds")})
    finalize_b(h)


def test_smetric_size_properties_and_reproducibility():
    # BASELINE's metric configuration (n = 4e4, m = 6e4; 16641-column root front): the look-ahead streams, the
    # adaptive super-step width and the super-block solves all engage only at this size.  Properties checked:
    # inertia (n, m, 0), residual, linearity, and -- because every tile is updated in a fixed order whatever the
    # streams' relative timing -- bit-identical D and solutions from a second factorisation of the same values.
    prob = synth.make_config("S-metric", seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, n, m) == 1
    assert h.inertia == (n, m, 0, 0)
    M = full_sym(K)
    rng = np.random.default_rng(2)
    b1, b2 = rng.normal(size=n + m), rng.normal(size=n + m)
    d_first = h.diag().copy()
    # solve 1 takes the 128-column steps, solve 2 prepares the super-block inverses, 3+ use them
    x1, x2, x12, x1_again = h.ls_solve(b1), h.ls_solve(b2), h.ls_solve(b1 + 2.0 * b2), h.ls_solve(b1)
    for b, x in ((b1, x1), (b2, x2), (b1, x1_again)):
        r = M @ x - b
        assert np.max(np.abs(r)) <= 1e-7 * np.max(np.abs(b)) * max(1.0, np.max(np.abs(x)))
    assert np.max(np.abs(x12 - (x1 + 2.0 * x2))) <= 1e-7 * np.max(np.abs(x12))
    assert np.max(np.abs(x1_again - x1)) <= 1e-7 * np.max(np.abs(x1))
    assert h.ls_factor_b(K, n, m) == 1
    assert np.array_equal(h.diag(), d_first)
    assert np.array_equal(h.ls_solve(b1), x1)
    st = h.stats()
    print("S-metric stats:", {k: st[k] for k in ("nnzL", "flops_exact", "max_front", "last_factor_ms", "last_solve_ms")})
    finalize_b(h)


def test_long_columns_chunked_assembly_with_duplicates():
    # one dense front of 2300 rows: its columns are assembled in 1024-row LDS chunks (k_big_assemble_chunked), three
    # panel super-steps with look-ahead; a raw CSC with duplicated entries exercises the serial has_dup scatter
    n = 2300
    rng = np.random.default_rng(23)
    B = rng.normal(size=(n, n))
    M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
    A = sp.csc_matrix(np.tril(M))
    w = np.linalg.eigvalsh(M)
    compare_with_oracle(A, int((w > 0).sum()), int((w < 0).sum()), tol=1e-9)
    # duplicates: split 200 random lower entries (some on chunk boundaries and on the diagonal) into two halves
    A.sort_indices()
    colptr, rowval, nzval = [A.indptr.astype(np.int64)], A.indices.astype(np.int64), A.data.copy()
    cols = np.repeat(np.arange(n), np.diff(A.indptr))
    pick = np.concatenate([rng.choice(len(nzval), size=196, replace=False),
                           [A.indptr[0], A.indptr[5] + 1019, A.indptr[5] + 1020, A.indptr[1030]]]).astype(np.int64)
    pick = np.unique(pick)
    keep = np.ones(len(nzval), dtype=bool)
    extra_r, extra_c, extra_v = rowval[pick], cols[pick], 0.25 * nzval[pick]
    nz2 = nzval.copy(); nz2[pick] *= 0.75
    rr = np.concatenate([rowval, extra_r]); cc = np.concatenate([cols, extra_c]); vv = np.concatenate([nz2, extra_v])
    order = np.lexsort((rr, cc))
    rr, cc, vv = rr[order], cc[order], vv[order]
    cp = np.zeros(n + 1, dtype=np.int64); np.add.at(cp, cc + 1, 1); cp = np.cumsum(cp)
    h = hip_solver("symmetric")
    assert h.ls_factor_b((n, cp, rr, vv, 0), int((w > 0).sum()), int((w < 0).sum())) == 1
    b = rng.normal(size=n)
    x = h.ls_solve(b)
    assert np.max(np.abs(M @ x - b)) <= 1e-9 * np.max(np.abs(b)) * np.sqrt(n)
    finalize_b(h)


def test_super_block_solves_follow_the_factorisation():
    # fronts with >= 2048 pivot columns switch to the super-block solves (explicit 1024-column inverses, prepared lazily):
    # the first solve after a factorisation takes the 128-column steps, the second prepares the inverses, the later ones
    # use them -- and a NEW factorisation must invalidate them
    n = 2300
    rng = np.random.default_rng(5)
    h = hip_solver("symmetric")
    for trial in range(2):
        B = rng.normal(size=(n, n))
        M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
        w = np.linalg.eigvalsh(M)
        assert h.ls_factor_b(sp.csc_matrix(np.tril(M)), int((w > 0).sum()), int((w < 0).sum())) == 1
        xs = []
        for i in range(4):
            b = rng.normal(size=n) if i < 3 else bs
            bs = b
            x = h.ls_solve(b)
            assert np.max(np.abs(M @ x - b)) <= 1e-9 * np.max(np.abs(b)) * np.sqrt(n), (trial, i)
            xs.append(x)
        # the same right-hand side through the prepared inverses twice: identical bits (deterministic kernels)
        assert np.array_equal(xs[2], xs[3])
    finalize_b(h)


def test_early_exit_of_a_failed_factorisation_is_opt_in():
    # okkt_set_early_exit / opts.early_exit: ls_factor! only returns the flag, so a factorisation that has already counted
    # more than m negative pivots may skip the top of the tree; by default the counts are complete
    prob = synth.make_config("S-small", seed=2, convex=False, neg_shift=50.0, well_scaled=True)
    n, m = prob["n"], prob["m"]
    K0, K1 = synth.augmented_matrix(prob, delta=0.0), synth.augmented_matrix(prob, delta=80.0)
    b = np.random.default_rng(0).normal(size=n + m)
    full = hip_solver("symmetric")
    assert full.ls_factor_b(K0, n, m) == 0 and sum(full.inertia) == n + m
    early = hip_solver("symmetric", early_exit=1)
    assert early.ls_factor_b(K0, n, m) == 0
    assert early.inertia[1] > m and sum(early.inertia) < n + m
    with pytest.raises(OkktError):
        early.ls_solve(b)
    assert early.ls_factor_b(K1, n, m) == 1 and early.inertia == (n, m, 0, 0)
    assert full.ls_factor_b(K1, n, m) == 1
    assert np.array_equal(early.ls_solve(b), full.ls_solve(b))
    finalize_b(full); finalize_b(early)


@pytest.mark.parametrize("n", [129, 130, 191, 192, 193, 255, 256, 257, 320, 383, 384, 385, 449])
def test_one_front_of_every_mid_size_against_lapack(n):
    """A single dense front whose pivot block takes the solves through block substitution in 64-column steps (129 .. 384 columns: csrc/solve.hip,
    k_fwd_mid / k_bwd_mid -- every count of full and partial steps, odd and even row counts below a step) and, from 385 on, through the
    explicit inverse again: the solution of a well-conditioned indefinite system against LAPACK's to 1e-11, one and four right-hand sides."""
    rng = np.random.default_rng(1000 + n)
    B = rng.normal(size=(n, n))
    M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
    w = np.linalg.eigvalsh(M)
    h = hip_solver("symmetric")
    assert h.ls_factor_b(sp.csc_matrix(np.tril(M)), int((w > 0).sum()), int((w < 0).sum())) == 1
    from onephase_jl_amd import _lib as L
    Bm = rng.normal(size=(4, n))
    X = np.zeros_like(Bm)
    h._check(h._lib.okkt_solve(h._h, L.p_f64(Bm), L.p_f64(X), 4), "okkt_solve")
    for r in range(4):
        xt = np.linalg.solve(M, Bm[r])
        assert np.max(np.abs(X[r] - xt)) <= 1e-11 * np.max(np.abs(xt)), (n, r)
        assert np.max(np.abs(h.ls_solve(Bm[r]) - xt)) <= 1e-11 * np.max(np.abs(xt)), (n, r)
    finalize_b(h)


@pytest.mark.parametrize("case", ["S-small", "dense-257", "dense-700", "dense-2600", "S-C3"])
def test_batched_right_hand_sides(case):
    # okkt_solve with nrhs > 1 carries up to four right-hand sides through one pass over L (solve.hip): every column of
    # the batch must equal the single solve of that column (same summation order: to rounding of the compiler's FMA
    # choices) -- thin fronts, wide fronts with one and two inverted diagonal blocks, small fronts, batches 1..6
    rng = np.random.default_rng(42)
    if case.startswith("dense"):
        n = int(case.split("-")[1])
        B = rng.normal(size=(n, n))
        M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
        A = sp.csc_matrix(np.tril(M))
        w = np.linalg.eigvalsh(M)
        npos, nneg = int((w > 0).sum()), int((w < 0).sum())
    else:
        prob = synth.make_config(case, seed=0, **({"well_scaled": True} if case == "S-small" else {}))
        A = synth.augmented_matrix(prob, delta=1e-8)
        npos, nneg = prob["n"], prob["m"]
    h = hip_solver("symmetric")
    assert h.ls_factor_b(A, npos, nneg) == 1
    dim = A.shape[0]
    Mfull = full_sym(A)
    from onephase_jl_amd import _lib as L
    for nrhs in (1, 2, 3, 4, 5, 6):
        Bm = rng.normal(size=(nrhs, dim))
        X = np.zeros_like(Bm)
        h._check(h._lib.okkt_solve(h._h, L.p_f64(Bm), L.p_f64(X), nrhs), "okkt_solve")
        for r in range(nrhs):
            x1 = h.ls_solve(Bm[r])
            assert np.max(np.abs(X[r] - x1)) <= 1e-13 * max(1.0, np.max(np.abs(x1))), (nrhs, r)
            assert np.max(np.abs(Mfull @ X[r] - Bm[r])) <= 1e-7 * np.max(np.abs(Bm[r])) * max(1.0, np.max(np.abs(X[r])))
    finalize_b(h)


def _kkt_from(J, n, rng, h_pattern=None):
    """Quasi-definite K = [[H, J'], [J, -D]] (lower triangle stored) from a constraint pattern: H diagonally dominant positive."""
    m = J.shape[0]
    H = sp.diags(1.0 + rng.random(n)) if h_pattern is None else h_pattern
    K = sp.bmat([[sp.tril(H), None], [J, -sp.diags(0.5 + rng.random(m))]], format="csc")
    return K, n, m


@pytest.mark.parametrize("shape", ["arrow", "band", "blocks_with_isolated_nodes", "star", "dense_rows", "two_by_two_chain", "all_in_one_column"])
@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("ordering", [0, 4])
def test_structure_zoo_vs_oracle(shape, seed, ordering):
    # sparsity shapes that stress different parts of the plan (one huge separator, a path-shaped tree, forests with singleton
    # fronts, dense rows set aside by the orderings, a 3 x 3 system, a single dense column), each under the
    # default ordering and under forced nested dissection: inertia counts, sign(D), D and solutions against the oracle with
    # the library's permutation
    rng = np.random.default_rng(17 * seed + len(shape))
    if shape == "arrow":
        n, m = 60, 90
        J = sp.lil_matrix((m, n)); J[:, 0] = rng.normal(size=(m, 1)); J.setdiag(rng.normal(size=min(m, n)) + 2.0)
    elif shape == "band":
        n, m = 300, 299
        J = sp.diags([rng.normal(size=m) + 2.0, rng.normal(size=m)], [0, 1], shape=(m, n)).tolil()
    elif shape == "blocks_with_isolated_nodes":
        n, m = 120, 40
        J = sp.lil_matrix((m, n))
        for i in range(m):
            J[i, 3 * (i % 20): 3 * (i % 20) + 3] = rng.normal(size=3)       # columns 60..119 touch no constraint: singleton fronts
    elif shape == "star":
        n, m = 1, 150
        J = sp.lil_matrix(rng.normal(size=(m, 1)))
    elif shape == "dense_rows":
        n, m = 400, 30
        J = sp.lil_matrix((m, n))
        for i in range(m):
            cols = rng.choice(n, size=220 if i < 3 else 4, replace=False)
            J[i, cols] = rng.normal(size=len(cols))
    elif shape == "two_by_two_chain":
        n, m = 2, 1
        J = sp.lil_matrix(np.array([[1.5, -0.5]]))
    else:   # all_in_one_column
        n, m = 50, 50
        J = sp.lil_matrix((m, n)); J[:, 7] = rng.normal(size=(m, 1)) + 3.0
    K, n, m = _kkt_from(sp.csc_matrix(J), n, rng)
    compare_with_oracle(K, n, m, seed=seed, tol=1e-9, ordering=ordering)


@pytest.mark.parametrize("case", ["chain", "small_fronts_below_a_big_front", "small_fronts_only"])
@pytest.mark.parametrize("nrhs", [1, 4])
def test_levels_of_small_fronts_in_one_launch(case, nrhs):
    """The leading levels of the tree that hold small fronts only run as ONE launch per phase (factorisation, forward sweep,
    backward sweep), a task waiting in the launch for its children (its parent in the backward sweep).  Answers equal a dense
    solve, repeated calls agree bit for bit, and no hand-off runs into its time-out (a wait that is never answered ends after
    about 0.7 s instead of hanging the GPU: the device times below would show it)."""
    if case == "chain":
        prob = synth.hanging_chain(N_h=300, seed=2)
    elif case == "small_fronts_below_a_big_front":
        prob = synth.make_config("S-small", seed=1, h_per_col=2, j_per_row=3)    # one front of 158 rows above three levels of tasks
    else:
        prob = synth.make_config("S-tiny", seed=3)
    K = synth.augmented_matrix(prob, delta=0.5)
    Ms = synth.symmetrize_lower(K).toarray()
    w = np.linalg.eigvalsh(Ms)
    h = hip_solver("symmetric")
    rng = np.random.default_rng(5)
    B = rng.normal(size=(nrhs, K.shape[0]))
    X = []
    for rep in range(3):
        h.ls_factor_b(K, prob["n"], prob["m"])
        assert h.inertia == (int((w > 0).sum()), int((w < 0).sum()), 0, 0)
        if nrhs > 1:
            from onephase_jl_amd import _lib as L
            x = np.zeros_like(B)
            h._check(h._lib.okkt_solve(h._h, L.p_f64(B), L.p_f64(x), nrhs), "okkt_solve")
        else:
            x = h.ls_solve(B[0])[None, :]
        st = h.stats()
        assert st["last_factor_ms"] < 300.0 and st["last_solve_ms"] < 300.0, st      # (a hand-off that ran into its time-out costs >= 700 ms)
        X.append(np.array(x))
    xd = np.linalg.solve(Ms, B.T).T
    assert np.max(np.abs(X[0] - xd)) <= 1e-8 * max(1.0, np.max(np.abs(xd)))
    assert np.array_equal(X[0], X[1]) and np.array_equal(X[1], X[2])
    finalize_b(h)
