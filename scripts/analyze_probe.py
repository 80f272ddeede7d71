import sys, time
sys.path.insert(0,'.')
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import linear_solver_HIP, initialize_b
prob = synth.make_config(sys.argv[1] if len(sys.argv)>1 else "S-metric", seed=0)
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric", host_symbolic_only=1); initialize_b(h)
t=time.time(); h.analyze(K); print("analyze", time.time()-t)
st=h.stats(); print({k: st[k] for k in ("nnzL","flops_exact","flops_stored","nsuper","max_front","analyze_seconds")})
