"""Target of bench.py's live PMC passes: one analysis, two numeric factorisations of the metric workload (values resident)."""
import sys
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h)
h.analyze(K)
d_vals = h.dev_upload(K.data)
for _ in range(2):
    rc = h.ls_factor_dev(d_vals, n, m)
print("rc", rc)
finalize_b(h)
