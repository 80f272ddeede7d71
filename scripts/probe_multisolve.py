"""Timing probe (GPU box): one factorisation followed by several solves (the IPM pattern): device ms of each solve."""
import sys
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP

name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
nsolve = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
M = synth.symmetrize_lower(K)
h = linear_solver_HIP("symmetric"); initialize_b(h)
rng = np.random.default_rng(0)
for rep in range(2):
    h.ls_factor_b(K, n, m)
    out = [f"factor {h.stats()['last_factor_ms']:.2f}"]
    for i in range(nsolve):
        b = rng.normal(size=n + m)
        x = h.ls_solve(b)
        out.append(f"solve{i + 1} {h.stats()['last_solve_ms']:.2f} (res {np.max(np.abs(M @ x - b)) / np.max(np.abs(b)):.1e})")
    print(name, "rep", rep, " | ".join(out))
finalize_b(h)
