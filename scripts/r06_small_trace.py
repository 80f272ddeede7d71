"""Kernel trace target (GPU box, under rocprofv3 --kernel-trace): the CUTEst-size configurations (S-C2 at N_h = 400 / 20000, S-C4) factor + solve a few
times, so that the trace shows which launches a small system pays for."""
import sys
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP

which = sys.argv[1] if len(sys.argv) > 1 else "400"
prob = synth.infeasible_lp(seed=0) if which == "lp" else synth.hanging_chain(N_h=int(which), seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-4)
h = linear_solver_HIP("symmetric"); initialize_b(h)
b = np.random.default_rng(0).normal(size=n + m)
for r in range(6):
    rc = h.ls_factor_b(K, n, m)
    x = h.ls_solve(b)
    st = h.stats()
    print(f"rep {r}: rc {rc} factor dev {st['last_factor_ms']:.3f} ms, solve dev {st['last_solve_ms']:.3f} ms")
print({k: st[k] for k in ("nnzL", "nsuper", "nlevels", "max_front", "n_big_fronts")})
finalize_b(h)
