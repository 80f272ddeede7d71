"""Factor / solve device times on the stand-ins of BASELINE configs 2 and 4 (small fronts only / mixed)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
for name, prob in (("S-C2 chain N_h=400", synth.hanging_chain(400)), ("S-C2 chain N_h=20000", synth.hanging_chain(20000)),
                   ("S-C4 lp 600x900", synth.infeasible_lp()), ("S-C4 lp 3000x4500", synth.infeasible_lp(3000, 4500)),
                   ("S-small", synth.make_config("S-small", seed=0))):
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-2)
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    h.ls_factor_b(K, n, m)
    b = np.random.default_rng(0).normal(size=n + m)
    for _ in range(3):
        rc = h.ls_factor_b(K, n, m); x = h.ls_solve(b)
    st = h.stats()
    print(f"{name:24s} n+m={n+m:6d} nnzL={st['nnzL']:9d} levels={st['nlevels']:3d} fronts small/big={st['n_small_fronts']}/{st['n_big_fronts']} max_front={st['max_front']:5d} "
          f"factor {1e3*st['last_factor_ms']:.0f} us  solve {1e3*st['last_solve_ms']:.0f} us  analyze {st['analyze_seconds']*1e3:.1f} ms  rc={rc}")
    finalize_b(h)
