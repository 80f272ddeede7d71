"""GPU probe: the delta loop at scale -- a nonconvex S-C3 / S-metric problem (H shifted by -lambda) through
HIP_KKT_solver.ipopt_strategy_b, with #fac, delta, wall time, then one direction."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd import kkt_system_solver as KS

name = sys.argv[1] if len(sys.argv) > 1 else "S-C3"
shift = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
prob = synth.make_config(name, seed=0, convex=False, neg_shift=shift)
n, m = prob["n"], prob["m"]
rng = np.random.default_rng(1)
it = KS.Class_iterate(x=rng.normal(size=n), y=prob["y"], s=prob["s"], mu=float(prob["mu"]), J=prob["J"], H=prob["H"],
                      grad=rng.normal(size=n), cons=prob["s"] + 1e-3 * rng.normal(size=m))
for kind in ("symmetric", "schur"):
    pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = kind
    k = KS.pick_KKT_solver(pars)
    k.initialize_b(it); k.form_system_b(it)
    k.ipopt_strategy_b(it)                       # warm (analysis, module loads)
    k.form_system_b(it)
    t = time.time(); status, nfac, delta = k.ipopt_strategy_b(it); dt = time.time() - t
    k.kkt_associate_rhs_b(it, KS.Reduct_affine())
    t = time.time(); k.compute_direction_b(); dd = time.time() - t
    print(f"{name} {kind}: diag_min {k.diag_min():.3g}  status {status}  #fac {nfac}  delta {delta:.3g}  loop {1e3 * dt:.1f} ms ({1e3 * dt / nfac:.1f} ms/fac)  direction {1e3 * dd:.1f} ms  N-err {k.kkt_err_norm.ratio:.2e}")
    k.finalize_b()
