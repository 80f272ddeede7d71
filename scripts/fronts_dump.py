import sys, numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, linear_solver_HIP
name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
prob = synth.make_config(name, seed=0)
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h); h.analyze(K)
print(h.stats())
n, m = prob["n"], prob["m"]
d_vals = h.dev_upload(K.data)
h.ls_factor_dev(d_vals, n, m)      # the numeric set-up (and OKKT_DEBUG_FRONTS=1's list of the big fronts) happens at the first factorisation
print(h.stats())
