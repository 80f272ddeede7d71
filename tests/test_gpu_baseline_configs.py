"""GPU parity tests on the stand-ins for BASELINE configs 2 and 4 (CUTEst CHAIN and the Netlib-infeasible
LPs are not in the repo): the hanging chain (banded, small fronts, indefinite H) and a rank-deficient infeasible LP
(H = 0).  Both need the delta loop; compared with the oracle: the '#fac' trace of ipopt_strategy!, the direction, N err."""
import numpy as np
import pytest

from onephase_jl_amd import kkt_system_solver as KS
from onephase_jl_amd import synth
from oracle import kkt_oracle as KO

pytestmark = pytest.mark.gpu

CASES = {"S-C2": lambda: synth.hanging_chain(N_h=400, seed=0), "S-C4": lambda: synth.infeasible_lp(seed=0)}


def iterates(prob, seed=5):
    rng = np.random.default_rng(seed)
    n, m = prob["n"], prob["m"]
    kw = dict(x=rng.normal(size=n), y=prob["y"].copy(), s=prob["s"].copy(), mu=prob["mu"], J=prob["J"], H=prob["H"],
              grad=rng.normal(size=n), cons=prob["s"] + 0.1 * rng.normal(size=m), a_norm_penalty_par=1e-4)
    return KS.Class_iterate(**kw), KO.Iterate(**kw)


@pytest.mark.parametrize("name", ["S-C2", "S-C4"])
@pytest.mark.parametrize("kind", ["schur", "symmetric", "clever_symmetric"])
def test_delta_loop_and_direction_vs_oracle(name, kind):
    prob = CASES[name]()
    it, oit = iterates(prob)
    k = KS.HIP_KKT_solver(kind)
    k.initialize_b(it)
    k.form_system_b(it)
    status, num_fac, delta = k.ipopt_strategy_b(it)
    ko = KO.pick_KKT_solver(kind, perm=k.linear_solver_perm())
    ko.initialize_b(oit)
    ko.form_system_b(oit)
    ostatus, onum_fac, odelta, tried = KO.ipopt_strategy_b(oit, ko)
    assert (status, num_fac, delta) == (ostatus, onum_fac, odelta), (tried,)
    assert status == "success" and num_fac >= 2          # delta = 0 (or the first shift) does not give the inertia (n, m, 0)
    k.kkt_associate_rhs_b(it, KS.Reduct_stable())
    k.compute_direction_b()
    ko.kkt_associate_rhs_b(oit, KO.Reduct_stable())
    ko.compute_direction_b()
    for a in ("x", "y", "s"):
        ref = getattr(ko.dir, a)
        err = np.linalg.norm(getattr(k.dir, a) - ref) / max(np.linalg.norm(ref), 1e-300)
        assert err < 1e-6, (name, kind, a, err)           # the reference's own bar (test/kkt_system_solvers.jl:118-120)
    assert k.kkt_err_norm.ratio < 1e-6 and ko.kkt_err_norm.ratio < 1e-6
    if kind == "clever_symmetric":
        # [cons >= l ; -cons <= u] rows are pairwise parallel: the reduced system is much smaller
        assert k.m_new <= prob["m"] - min(400, prob["m"] // 4)
    k.finalize_b()


def test_chain_has_only_small_fronts():
    prob = synth.hanging_chain(N_h=400, seed=0)
    it, _ = iterates(prob)
    k = KS.HIP_KKT_solver("symmetric")
    k.initialize_b(it)
    k.form_system_b(it)
    st = k.linear_solver_stats()
    assert st["max_front"] <= 128 and st["n_big_fronts"] == 0, st["max_front"]
    k.finalize_b()


def test_chain_nested_dissection_matches_amd_order_on_the_device():
    # the same banded system factored with the path-shaped AMD tree (ordering = 3) and with the automatic choice (nested
    # dissection): equal inertia, solutions equal to fp64 accuracy, and the oracle agrees with both
    import oracle
    from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP
    prob = synth.hanging_chain(N_h=2000, seed=1)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1.0)
    b = np.random.default_rng(3).normal(size=(n + m, 2))
    out = {}
    for ordering in (3, 0):
        h = linear_solver_HIP("symmetric", ordering=ordering)
        initialize_b(h)
        assert h.ls_factor_b(K, n, m) == 1
        st = h.stats()
        out[ordering] = (h.inertia, np.column_stack([h.ls_solve(b[:, 0]), h.ls_solve(b[:, 1])]), st["ordering_used"], st["critical_pivots"], h.perm())
        finalize_b(h)
    assert out[3][2] == 0 and out[0][2] == 4 and out[0][3] * 10 <= out[3][3]
    assert out[3][0] == out[0][0]
    assert np.max(np.abs(out[3][1] - out[0][1])) <= 1e-9 * np.max(np.abs(out[3][1]))
    o = oracle.linear_solver_ORACLE("symmetric", perm=out[0][4])
    assert o.ls_factor_b(K, n, m) == 1
    xo = np.column_stack([o.ls_solve(b[:, 0]), o.ls_solve(b[:, 1])])
    assert np.max(np.abs(out[0][1] - xo)) <= 1e-9 * np.max(np.abs(xo))
