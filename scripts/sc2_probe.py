"""GPU box: factor + solve of the S-C2 stand-in (hanging chain, N_h = 400 by default): device times, wall rate."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, linear_solver_HIP
N_h = int(sys.argv[1]) if len(sys.argv) > 1 else 400
prob = synth.hanging_chain(N_h=N_h, seed=0); n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h); h.analyze(K)
d_vals = h.dev_upload(K.data); rhs = np.random.default_rng(1).normal(size=n + m); d_rhs = h.dev_upload(rhs); d_sol = h.dev_alloc(8 * (n + m))
for _ in range(5): h.ls_factor_dev(d_vals, n, m); h.ls_solve_dev(d_rhs, d_sol)
N = 200
t0 = time.perf_counter(); f = s = 0
for _ in range(N):
    h.ls_factor_dev(d_vals, n, m); h.ls_solve_dev(d_rhs, d_sol); st = h.stats(); f += st["last_factor_ms"]; s += st["last_solve_ms"]
dt = (time.perf_counter() - t0) / N
st = h.stats()
print({k: st[k] for k in ("n", "nnz_lower", "nnzL", "nsuper", "nlevels", "max_front", "n_small_fronts", "n_big_fronts")})
print(f"S-C2 N_h={N_h}: {1.0 / dt:.0f} factor+solve/s ({dt * 1e6:.1f} us wall per pair incl. the stats call), factor {f / N * 1e3:.1f} us, solve {s / N * 1e3:.1f} us (device events)")
