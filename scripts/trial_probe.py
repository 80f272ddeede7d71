"""GPU box: what a wrong-inertia trial factorisation of the delta loop costs (okkt_set_early_exit(1); the second and later
attempts are retries: the kernels obey the device-side stop flag) beside a complete factorisation of the same matrix."""
import sys
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
shift = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
prob = synth.make_config(name, seed=0, convex=False, neg_shift=shift)
n, m = prob["n"], prob["m"]
h = linear_solver_HIP("symmetric"); initialize_b(h)
K = synth.augmented_matrix(prob, delta=1e-6)
h.analyze(K)
d = h.dev_upload(K.data)
for early in (0, 1):
    h._check(h._lib.okkt_set_early_exit(h._h, early), "early")
    for rep in range(4):
        rc = h.ls_factor_dev(d, n, m)
        st = h.stats()
        print(f"{name} shift {shift} early_exit={early} attempt {rep}: rc {rc} inertia {h.inertia} factor {st['last_factor_ms']:.3f} ms", flush=True)
for delta in (1e-6 * 8 ** k for k in range(0, 10, 3)):
    K2 = synth.augmented_matrix(prob, delta=delta)
    d2 = h.dev_upload(K2.data)
    rc = h.ls_factor_dev(d2, n, m); st = h.stats()
    print(f"  delta {delta:.3g}: rc {rc} inertia {h.inertia} factor {st['last_factor_ms']:.3f} ms")
    h.dev_free(d2)
finalize_b(h)
