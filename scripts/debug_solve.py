"""GPU debugging aid: solve errors against dense / oracle answers for a sweep of front shapes."""
import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
import oracle
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP

rng = np.random.default_rng(0)
def dense_case(n, **opts):
    B = rng.normal(size=(n, n))
    M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
    w = np.linalg.eigvalsh(M)
    h = linear_solver_HIP("symmetric", **opts); initialize_b(h)
    rc = h.ls_factor_b(sp.csc_matrix(np.tril(M)), int((w > 0).sum()), int((w < 0).sum()))
    b = rng.normal(size=n)
    x = h.ls_solve(b)
    xd = np.linalg.solve(M, b)
    print(f"dense n={n} {opts} rc={rc} err={np.max(np.abs(x - xd)) / np.max(np.abs(xd)):.2e}", flush=True)
    finalize_b(h)
for n in (100, 129, 130, 200, 256, 257, 300, 513, 700, 1025, 1485, 1500, 2049, 2600):
    dense_case(n)
for nb in (32, 64):
    for n in (100, 200, 300):
        dense_case(n, panel_nb=nb)

def sparse_case(name, prob, delta=1e-8):
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=delta)
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    rc = h.ls_factor_b(K, n, m)
    o = oracle.linear_solver_ORACLE("symmetric", perm=h.perm())
    rco = o.ls_factor_b(K, n, m)
    b = rng.normal(size=n + m)
    x, xo = h.ls_solve(b), o.ls_solve(b)
    st = h.stats()
    print(f"{name}: rc={rc}/{rco} err={np.max(np.abs(x - xo)) / np.max(np.abs(xo)):.2e} max_front={st['max_front']} big={st['n_big_fronts']} solve_ms={st['last_solve_ms']:.3f}", flush=True)
    finalize_b(h)
sparse_case("blocks-1000", synth.block_angular(nblocks=8, n_b=400, m_b=600, n_link=10, seed=2, j_per_row=4, h_per_col=3, w=6.0, p_far=0.0, well_scaled=True), 1e-7)
sparse_case("blocks-2500", synth.block_angular(nblocks=4, n_b=1000, m_b=1500, n_link=40, seed=1), 1e-8)
sparse_case("blocks-5000", synth.block_angular(nblocks=2, n_b=2000, m_b=3000, n_link=60, seed=1), 1e-8)
sparse_case("S-C5-2blocks", synth.block_angular(nblocks=2, seed=0), 1e-8)
sparse_case("S-C3", synth.make_config("S-C3", seed=0))
