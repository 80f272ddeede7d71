import sys, time
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth, kkt_system_solver as KS
prob = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "S-metric", seed=0)
n, m = prob["n"], prob["m"]
rng = np.random.default_rng(1)
it = KS.Class_iterate(x=rng.normal(size=n), y=prob["y"], s=prob["s"], mu=float(prob["mu"]), J=prob["J"], H=prob["H"],
                      grad=rng.normal(size=n), cons=prob["s"] + 1e-3 * rng.normal(size=m))
for kind in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("symmetric", "schur")):
    k = KS.HIP_KKT_solver(kind)
    k.initialize_b(it)
    k.form_system_b(it)
    for r in range(3):
        try:
            print(kind, r, "inertia flag", k.factor_b(1e-8), k.linear_solver_stats()["last_factor_ms"])
        except Exception as e:
            print(kind, r, "FAILED", e)
    st = k.linear_solver_stats()
    print({q: st[q] for q in ("nnz_lower", "nnzL", "flops_exact", "flops_stored", "max_front", "n_big_fronts", "nlevels")})
    k.finalize_b()
