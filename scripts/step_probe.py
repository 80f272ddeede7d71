import sys, time, numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
prob = synth.make_config(name, seed=0); n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h); h.analyze(K)
d_vals = h.dev_upload(K.data); rhs = np.random.default_rng(1).normal(size=n + m); d_rhs = h.dev_upload(rhs); d_sol = h.dev_alloc(8 * (n + m))
for _ in range(3): h.ls_factor_dev(d_vals, n, m); h.ls_solve_dev(d_rhs, d_sol)
t0 = time.perf_counter(); f = s = 0
N = 10
for _ in range(N):
    h.ls_factor_dev(d_vals, n, m); h.ls_solve_dev(d_rhs, d_sol); st = h.stats(); f += st["last_factor_ms"]; s += st["last_solve_ms"]
dt = (time.perf_counter() - t0) / N * 1e3
print(f"{name}: step {dt:.2f} ms wall, factor {f / N:.2f} ms, solve {s / N:.2f} ms")
