"""GPU parity test of the multi-GPU (subtree-sharded) path on ONE device: nparts virtual ranks driven by
LocalComm run exactly the code a torchrun job runs per GPU; the result must match the unsharded solver."""
import numpy as np
import pytest
import scipy.sparse as sp

from onephase_jl_amd import synth
from onephase_jl_amd.distributed import LocalComm, RcclShardedLinearSolver, ShardedLinearSolver
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

pytestmark = pytest.mark.gpu


def problems():
    yield "block-angular", synth.block_angular(nblocks=4, n_b=150, m_b=220, n_link=12, seed=1, j_per_row=5, h_per_col=3, w=8.0, p_far=0.0, well_scaled=True)
    yield "S-small", synth.make_config("S-small", seed=2, well_scaled=True)


@pytest.mark.parametrize("nparts", [2, 3, 8])
def test_sharded_matches_single(nparts):
    for name, prob in problems():
        n, m = prob["n"], prob["m"]
        K = synth.augmented_matrix(prob, delta=1e-7)
        ref = linear_solver_HIP("symmetric")
        initialize_b(ref)
        assert ref.ls_factor_b(K, n, m) == 1
        b = np.random.default_rng(5).normal(size=n + m)
        x_ref = ref.ls_solve(b)
        sh = ShardedLinearSolver(LocalComm(nparts), "symmetric")
        info = sh.analyze(K)
        assert abs(sum(info["part_flops"]) + info["top_flops"] - ref.stats()["flops_stored"]) <= 1e-6 * ref.stats()["flops_stored"]
        d_vals = [s.dev_upload(K.data) for s in sh.solvers]
        d_rhs = [s.dev_upload(b) for s in sh.solvers]
        assert sh.factor(d_vals, n, m) == 1
        assert sh.inertia == ref.inertia                       # pivot counts, exact
        x = sh.solve(d_rhs)
        assert np.max(np.abs(x - x_ref)) <= 1e-10 * np.max(np.abs(x_ref)), (name, nparts)
        # wrong-inertia case travels through the same reduction
        K2 = synth.augmented_matrix(prob, delta=-50.0)
        d_vals2 = [s.dev_upload(K2.data) for s in sh.solvers]
        assert sh.factor(d_vals2, n, m) == ref.ls_factor_b(K2, n, m) == 0
        assert sh.inertia == ref.inertia
        sh.finalize()
        finalize_b(ref)


def test_block_angular_subtrees_are_balanced():
    # blocks large against the amalgamation width (supernodes up to 128 columns): 1000 columns each
    prob = synth.block_angular(nblocks=8, n_b=400, m_b=600, n_link=10, seed=0, j_per_row=4, h_per_col=3, w=6.0, p_far=0.0)
    K = synth.augmented_matrix(prob, delta=1e-8)
    sh = ShardedLinearSolver(LocalComm(8), "symmetric")
    info = sh.analyze(K)
    pf = np.array(info["part_flops"])
    assert pf.min() > 0 and pf.max() <= 1.6 * pf.mean()        # 8 independent blocks -> 8 busy parts
    assert info["top_flops"] < 1.0 * pf.sum()
    sh.finalize()


# ---- speculative delta loop: W virtual ranks on one GPU must reproduce the serial ipopt_strategy! exactly
@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_speculative_delta_loop_matches_serial(golden, world):
    from conftest import iterate_from_record
    from onephase_jl_amd import kkt_system_solver as KS
    from onephase_jl_amd.distributed import LocalComm, speculative_ipopt_strategy
    for prob in ("indef5", "posdiag_indef5"):
        rec = golden[prob]
        for key, want in rec["delta_loops"].items():
            kind, dprev = key.split("_prev")
            dprev = float(dprev)
            if True:
                it = iterate_from_record(rec, KS.Class_iterate)
                it.delta = dprev
                pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = kind
                ks = []
                for _ in range(world):
                    k = KS.pick_KKT_solver(pars); k.initialize_b(it); k.form_system_b(it); ks.append(k)
                status, num_fac, delta, owner = speculative_ipopt_strategy(LocalComm(world), ks, it)
                serial = KS.pick_KKT_solver(pars); serial.initialize_b(it); serial.form_system_b(it)
                s_status, s_fac, s_delta = serial.ipopt_strategy_b(it)
                assert (status, num_fac, delta) == (s_status, s_fac, s_delta)              # bit-for-bit the serial loop
                assert num_fac == want["num_fac"] and delta == want["delta"]               # and the golden trace
                assert owner == (num_fac - 1) % world
                # every replica ends up with the chosen factor: same direction everywhere
                dirs = []
                for k in ks:
                    k.kkt_associate_rhs_b(it, KS.Reduct_affine()); k.compute_direction_b(); dirs.append(k.dir.x.copy())
                for d in dirs[1:]:
                    assert np.max(np.abs(d - dirs[0])) <= 1e-12 * max(1.0, np.max(np.abs(dirs[0])))
                for k in ks + [serial]:
                    k.finalize_b()


def test_rccl_transport_single_rank_matches_unsharded():
    # okkt_dist_factor / okkt_dist_solve (collectives inside the library, on RCCL) with a one-rank communicator: the whole
    # sequence -- RCCL id, ncclCommInitRank, pack / unpack, device-side count sum, one synchronisation -- against the
    # unsharded solver.  (Two ranks on one GPU are refused by RCCL; the multi-rank protocol is the LocalComm / gloo tests'.)
    prob = synth.make_config("S-small", seed=4, well_scaled=True)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-7)
    b = np.random.default_rng(6).normal(size=n + m)
    ref = linear_solver_HIP("symmetric")
    initialize_b(ref)
    assert ref.ls_factor_b(K, n, m) == 1
    x_ref = ref.ls_solve(b)
    sh = RcclShardedLinearSolver(0, 1, "symmetric")
    sh.analyze(K)
    s = sh.solver
    d_vals, d_rhs, d_sol = s.dev_upload(K.data), s.dev_upload(b), s.dev_alloc(8 * (n + m))
    for _ in range(2):
        assert sh.factor(d_vals, n, m) == 1
        assert sh.inertia == ref.inertia
        sh.solve(d_rhs, d_sol)
        x = s.dev_download(d_sol, (n + m,))
        assert np.max(np.abs(x - x_ref)) <= 1e-12 * np.max(np.abs(x_ref))
    K2 = synth.augmented_matrix(prob, delta=-50.0)
    assert sh.factor(s.dev_upload(K2.data), n, m) == ref.ls_factor_b(K2, n, m) == 0
    assert sh.inertia == ref.inertia
    sh.finalize()
    finalize_b(ref)
