"""Accuracy and time of the solves for one build of the library (OKKT_LIB_PATH): run once per kSolveBlock variant."""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
import oracle
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
tag = os.environ.get("OKKT_LIB_PATH", "default").split("/")[-1]
rng = np.random.default_rng(0)
def case(name, prob, use_oracle=True):
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    M = synth.symmetrize_lower(K)
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    rc = h.ls_factor_b(K, n, m)
    b = np.random.default_rng(1).normal(size=n + m)
    x = h.ls_solve(b); x = h.ls_solve(b)
    st = h.stats()
    res = np.max(np.abs(M @ x - b)) / (np.max(np.abs(b)) * max(1.0, np.max(np.abs(x))))
    err = float("nan")
    if use_oracle:
        o = oracle.linear_solver_ORACLE("symmetric", perm=h.perm()); o.ls_factor_b(K, n, m)
        xo = o.ls_solve(b)
        err = np.max(np.abs(x - xo)) / np.max(np.abs(xo))
        reso = np.max(np.abs(M @ xo - b)) / (np.max(np.abs(b)) * max(1.0, np.max(np.abs(xo))))
    else:
        reso = float("nan")
    np.save(f"/tmp/x_{name}_{tag}.npy", x)
    print(f"[{tag}] {name}: rc={rc} err_vs_oracle={err:.2e} resid={res:.2e} (oracle resid {reso:.2e}) factor_ms={st['last_factor_ms']:.2f} solve_ms={st['last_solve_ms']:.3f}", flush=True)
    finalize_b(h)
case("S-C5-2blocks", synth.block_angular(nblocks=2, seed=0))
case("S-C5", synth.make_config("S-C5", seed=0))
case("S-C3", synth.make_config("S-C3", seed=0))
case("S-metric", synth.make_config("S-metric", seed=0), use_oracle=False)
