"""Timing probe (GPU box), KKT level: set_structure / form_system / factor / rhs / direction for one synthetic config."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd import kkt_system_solver as KS

name = sys.argv[1] if len(sys.argv) > 1 else "S-C3"
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["symmetric", "schur"]
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
rng = np.random.default_rng(1)
it = KS.Class_iterate(x=rng.normal(size=n), y=prob["y"], s=prob["s"], mu=float(prob["mu"]), J=prob["J"], H=prob["H"],
                      grad=rng.normal(size=n), cons=prob["s"] + 1e-3 * rng.normal(size=m))
for kind in kinds:
    pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = kind
    k = KS.pick_KKT_solver(pars)
    t = time.time(); k.initialize_b(it); k.form_system_b(it); t_first = time.time() - t
    tm = {}
    for rep in range(3):
        t = time.time(); k.form_system_b(it); tm["form_system"] = time.time() - t
        t = time.time(); inertia = k.factor_b(1e-8); tm["factor"] = time.time() - t
        t = time.time(); k.kkt_associate_rhs_b(it, KS.Reduct_affine()); tm["rhs"] = time.time() - t
        t = time.time(); k.compute_direction_b(); tm["direction"] = time.time() - t
    st = k.linear_solver_stats()
    print(f"{name} {kind}: first form (incl. analyse) {t_first:.2f} s; " + "  ".join(f"{a} {1e3 * b:.2f} ms" for a, b in tm.items()) +
          f"  | inertia {inertia}  N-err {k.kkt_err_norm.ratio:.2e}  dev factor {st['last_factor_ms']:.2f} ms solve {st['last_solve_ms']:.2f} ms nnzL {st['nnzL']:.3g} flops {st['flops_exact']:.3g}")
    k.finalize_b()
