"""GPU parity at BASELINE's stated sizes against the ORACLE (not against another HIP code path).

  * config 3 (S-C3, n = 1e4, m = 2e4): the scalar oracle needs ~16 s for one factorisation -- affordable once per run:
    inertia counts equal, sign(D) equal entry by entry, D and the solution within the stated fp64 tolerance.
  * config 5 (S-C5, 8 x (5 000, 7 500) + 200 linking columns): the single handle against the oracle, and the sharded
    path (8 virtual ranks on one GPU, the code a torchrun job runs per GPU) against the same oracle solution.
The tolerance: 1e-8 relative on x (well inside the reference's own 1e-6 bar for directions); these matrices carry
s / y over six orders of magnitude, two correct factorisations with different summation orders differ by ~1e-10.
"""
import numpy as np
import pytest

import oracle
from onephase_jl_amd import _lib as L
from onephase_jl_amd import synth
from onephase_jl_amd.distributed import LocalComm, ShardedLinearSolver
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

pytestmark = pytest.mark.gpu

TOL_X = 1e-8
TOL_D = 1e-7     # D itself: pivots down to 1e-6 beside entries of 1e5 (measured 1.0e-8 at S-C5 between the two summation orders)


def hip_solver(sym, **o):
    s = linear_solver_HIP(sym, False, False, **o)
    initialize_b(s)
    return s


def hip_vs_oracle(K, n, m, nrhs=2, seed=0):
    h = hip_solver("symmetric")
    rc = h.ls_factor_b(K, n, m)
    o = oracle.linear_solver_ORACLE("symmetric", perm=h.perm())      # same pivot order by construction
    rco = o.ls_factor_b(K, n, m)
    assert rc == rco == 1
    assert h.inertia[:3] == o.inertia(1e-20)[:3] == (n, m, 0)
    d_h, d_o = h.diag(), o.diag()
    assert np.array_equal(np.sign(d_h), np.sign(d_o))
    assert np.max(np.abs(d_h - d_o) / np.abs(d_o)) <= TOL_D
    rng = np.random.default_rng(seed)
    bs = rng.normal(size=(nrhs, n + m))
    xo = [o.ls_solve(b) for b in bs]
    for b, x_ref in zip(bs, xo):
        x = h.ls_solve(b)
        err = np.max(np.abs(x - x_ref)) / np.max(np.abs(x_ref))
        assert err <= TOL_X, err
    return h, o, bs, xo


def test_sc3_full_size_against_the_oracle():
    prob = synth.make_config("S-C3", seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    h, o, bs, xo = hip_vs_oracle(K, n, m)
    # the batched entry point: both right-hand sides in one call, identical to the single solves bit for bit
    X = np.zeros_like(bs)
    h._check(h._lib.okkt_solve(h._h, L.p_f64(bs), L.p_f64(X), 2), "okkt_solve")
    for r in range(2):
        assert np.max(np.abs(X[r] - xo[r])) <= TOL_X * np.max(np.abs(xo[r]))
    finalize_b(h)


def test_sc5_full_size_single_handle_and_eight_parts_against_the_oracle():
    prob = synth.make_config("S-C5", seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    h, o, bs, xo = hip_vs_oracle(K, n, m, nrhs=1)
    st = h.stats()
    M = synth.symmetrize_lower(K)
    # the sharded path at the stated size: 8 parts, the cut keeps the eight blocks whole
    sh = ShardedLinearSolver(LocalComm(8), "symmetric")
    info = sh.analyze(K)
    total = sum(info["part_flops"]) + info["top_flops"]
    assert abs(total - st["flops_stored"]) <= 1e-6 * total
    assert info["top_flops"] <= 0.08 * total and min(info["part_flops"]) > 0
    d_vals = [s.dev_upload(K.data) for s in sh.solvers]
    d_rhs = [s.dev_upload(bs[0]) for s in sh.solvers]
    assert sh.factor(d_vals, n, m) == 1
    assert sh.inertia[:3] == o.inertia(1e-20)[:3]                      # pivot counts against the ORACLE, exact
    x = sh.solve(d_rhs)
    assert np.max(np.abs(x - xo[0])) <= TOL_X * np.max(np.abs(xo[0]))
    assert np.max(np.abs(M @ x - bs[0])) <= 1e-7 * np.max(np.abs(bs[0])) * max(1.0, np.max(np.abs(x)))
    sh.finalize()
    finalize_b(h)


@pytest.mark.parametrize("nparts", [2, 8])
def test_block_angular_sharded_against_the_oracle(nparts):
    # a block-angular case small enough for the oracle in a blink, with the wrong-inertia branch through the same
    # reduction: the comparator of the sharded path is the oracle, not the unsharded HIP path
    prob = synth.block_angular(nblocks=8, n_b=400, m_b=600, n_link=10, seed=2, j_per_row=4, h_per_col=3, w=6.0, p_far=0.0, well_scaled=True)
    n, m = prob["n"], prob["m"]
    b = np.random.default_rng(3).normal(size=n + m)
    sh = ShardedLinearSolver(LocalComm(nparts), "symmetric")
    for delta, want in ((1e-7, 1), (-50.0, 0)):
        K = synth.augmented_matrix(prob, delta=delta)
        sh.analyze(K)
        o = oracle.linear_solver_ORACLE("symmetric", perm=sh.solvers[0].perm())
        assert o.ls_factor_b(K, n, m) == want
        d_vals = [s.dev_upload(K.data) for s in sh.solvers]
        assert sh.factor(d_vals, n, m) == want
        assert sh.inertia[:3] == o.inertia(1e-20)[:3]
        if want == 1:
            x = sh.solve([s.dev_upload(b) for s in sh.solvers])
            xo = o.ls_solve(b)
            assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    sh.finalize()


def test_sc3_fully_random_variant_against_the_oracle():
    """BASELINE config 3 says "random sparse": the p_far = 1 variant of SURVEY 8d (every coupling uniformly random) has no
    separators and ends in one near-dense front -- the MFMA tile tasks carry it.  Parity at the size the scalar oracle affords:
    inertia and sign(D) equal, D and two solutions within the stated tolerances; the automatic ordering must not be worse than
    minimum degree (the reference's class of ordering) by more than the 10 % of its acceptance rule."""
    # s, y = O(1): with the default six orders of magnitude in s / y this near-dense matrix has pivots from 1e-8 to 1e8 and two
    # correct factorisations differ by more than the 1e-8 of TOL_X (measured 7e-9 ... 2e-8); the numerics are what is tested here
    prob = synth.make_config("S-C3-random-small", seed=0, well_scaled=True)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    h, o, bs, xo = hip_vs_oracle(K, n, m)
    st = h.stats()
    assert st["max_front"] >= 0.25 * (n + m)                 # the near-dense case it is meant to be
    ha = hip_solver("symmetric", ordering=3)
    ha.analyze(K)
    assert st["flops_exact"] <= 1.1 * ha.stats()["flops_exact"]
    finalize_b(ha)
    finalize_b(h)


def _forward_errors(name, nthreads):
    """forward error of the HIP solve and of the CPU restatement's against the extended-precision solution of the same fp64 matrix"""
    prob = synth.make_config(name, seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    M = synth.symmetrize_lower(K).tocsr()
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, n, m) == 1
    o = oracle.linear_solver_ORACLE_MF("symmetric", perm=h.perm(), nthreads=nthreads)
    o._analyze(K)
    assert o.ls_factor_b(K, n, m) == 1
    # the factor itself against the multithreaded CPU port on the same pivot order (round-5 review, 5b: at the metric size only the
    # forward error was compared): inertia counts equal, sign(D) equal entry by entry, D within TOL_D
    assert tuple(h.inertia[:3]) == tuple(o.inertia()[:3]) == (n, m, 0), (h.inertia, o.inertia())
    d_h, d_o = h.diag(), o.diag()
    assert np.array_equal(np.sign(d_h), np.sign(d_o))
    assert np.max(np.abs(d_h - d_o) / np.abs(d_o)) <= TOL_D
    rng = np.random.default_rng(3)
    out = []
    for b in rng.normal(size=(2, n + m)):
        xo = o.ls_solve(b)
        xt = xo.copy()
        for _ in range(4):
            prod = M.data.astype(np.longdouble) * xt.astype(np.longdouble)[M.indices]
            r = (b.astype(np.longdouble) - np.add.reduceat(prod, M.indptr[:-1])).astype(np.float64)
            xt = xt + o.ls_solve(r)
        xh = h.ls_solve(b)
        sc = np.max(np.abs(xt))
        out.append((np.max(np.abs(xo - xt)) / sc, np.max(np.abs(xh - xt)) / sc))
    finalize_b(h)
    return out


def test_smetric_forward_error_against_the_extended_precision_solution():
    """The metric workload itself (round-3 review: its widest fronts take the solves through the 1024-column explicit inverses and had
    only a residual check): the same bound as S-C3 below -- the HIP solution may be at most 2 x less accurate than the CPU
    restatement's (8 x until round 5), both measured against the extended-precision solution.  One multithreaded CPU factorisation (5 - 25 s)."""
    import os
    # the CPU restatement itself is at 7e-9 ... 1e-8 on this matrix; HIP measured 1.9e-8 ... 2.0e-8 in round 4, 6.9e-9 / 8.9e-9 with the block
    # substitution of round 5, 4.7e-9 / 4.9e-9 since the thin fronts of the upper levels go through it as well (round 6)
    for e_o, e_h in _forward_errors("S-metric", max(1, min(os.cpu_count() or 1, 32))):
        assert e_h <= 1e-7 and e_h <= 2.0 * e_o + 1e-12, (e_h, e_o)


def test_sc5_forward_error_against_the_extended_precision_solution():
    """BASELINE config 5 (round-4 review: it had no forward-error test).  Its many fronts of 129 - 384 pivot columns used to take the
    solves through explicit inverses of the whole pivot block (recursive doubling): 1.18e-8 and 3.4e-9 on the two vectors of this test
    against 3.7e-10 and 6.9e-10 for the CPU restatement -- outside the 1e-8 of the other comparisons.  `scripts/solve_emulation.py`
    located the loss (explicit inverses are as good as substitution up to 64 columns and 5 - 40 x worse from 128 on), and these fronts
    now go through block substitution in 64-column steps (csrc/solve.hip, k_fwd_mid / k_bwd_mid): measured 9.4e-10 and 4.0e-10 (round 6).
    Asserted: the stated tolerance on x for every vector, and at most 2 x the restatement's error (worst against worst; 4 x until round 5)."""
    errs = _forward_errors("S-C5", 8)
    worst_o = max(e_o for e_o, _ in errs)
    for e_o, e_h in errs:
        assert e_h <= TOL_X, (e_h, e_o)
    assert max(e_h for _, e_h in errs) <= 2.0 * worst_o + 1e-12, errs


def test_sc3_forward_error_against_the_extended_precision_solution():
    """Both solutions against the TRUE solution of the fp64 matrix (the oracle's solve refined with long-double residuals until the
    correction is at rounding level) instead of against each other: the HIP path may not be less accurate than the CPU restatement by
    more than a small factor.  This is what caught an amalgamation setting that was 12 % faster on S-C5 and wrong by 3.5e-8 instead of
    2e-9 (fronts of 256-512 pivot columns take the solves through the explicit inverses of the 2048-column blocks).
    Round 6 (the round-5 review's bound): at most 2 x the restatement's error -- measured 1.9e-9 / 1.6e-9 against 1.5e-9 / 2.1e-9 since the
    pivot blocks of up to 128 columns of the upper levels are solved by block substitution as well (5.0e-9 / 3.2e-9 before: 8 x was the bound)."""
    prob = synth.make_config("S-C3", seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    M = synth.symmetrize_lower(K).tocsr()
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, n, m) == 1
    o = oracle.linear_solver_ORACLE_MF("symmetric", perm=h.perm(), nthreads=8)
    o._analyze(K)
    assert o.ls_factor_b(K, n, m) == 1
    rng = np.random.default_rng(3)
    for b in rng.normal(size=(2, n + m)):
        xo = o.ls_solve(b)
        xt = xo.copy()
        for _ in range(4):
            prod = M.data.astype(np.longdouble) * xt.astype(np.longdouble)[M.indices]
            r = (b.astype(np.longdouble) - np.add.reduceat(prod, M.indptr[:-1])).astype(np.float64)
            xt = xt + o.ls_solve(r)
        xh = h.ls_solve(b)
        sc = np.max(np.abs(xt))
        e_o, e_h = np.max(np.abs(xo - xt)) / sc, np.max(np.abs(xh - xt)) / sc
        assert e_h <= TOL_X and e_h <= 2.0 * e_o + 1e-12, (e_h, e_o)
    finalize_b(h)


def test_contribution_blocks_are_released_and_a_four_times_larger_system_factors():
    """Round 6: only the L panels of finished fronts stay in the arena, the contribution blocks share a region by lifetime
    (csrc/numeric.hip, numeric_setup).  At the metric size the arena shrinks from the 6.0 GB of f x f buffers to 4.1 GB (the factor itself
    is 0.88 GB); a system of the same family with four times the unknowns (n + m = 4e5, nnz(L) = 1.3e9, 2.9e13 factor flops: the
    largest front has 37 860 rows, just below what the 32-bit local offsets allow) factors and solves on one GPU: inertia (n, m, 0),
    residual at rounding level."""
    prob = synth.make_config("S-metric", seed=0)
    K = synth.augmented_matrix(prob, delta=1e-8)
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, prob["n"], prob["m"]) == 1
    st = h.stats()
    assert st["arena_bytes"] <= 0.70 * st["arena_dense_bytes"], (st["arena_bytes"], st["arena_dense_bytes"])      # measured 4.09 of 6.04 GB
    assert st["arena_bytes"] <= 4.8 * 8 * st["nnzL_stored"], (st["arena_bytes"], st["nnzL_stored"])                # ... = 4.6 x the stored factor
    finalize_b(h)
    prob = synth.make_config("S-metric-4x", seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    h = hip_solver("symmetric")
    assert h.ls_factor_b(K, n, m) == 1
    assert tuple(h.inertia[:3]) == (n, m, 0)
    st = h.stats()
    assert st["arena_bytes"] < st["arena_dense_bytes"]
    b = np.random.default_rng(0).normal(size=n + m)
    x = h.ls_solve(b)
    M = synth.symmetrize_lower(K)
    assert np.max(np.abs(M @ x - b)) / np.max(np.abs(b)) < 2e-6
    finalize_b(h)
