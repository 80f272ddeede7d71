"""Where the wall-clock of the level-1 host call ls_factor!(A, n, m) goes beyond the device time (S-metric)."""
import sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth, _lib as L
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP, csc_arrays
prob = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "S-metric", seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
print("indices dtype", K.indices.dtype, "nnz", K.nnz)
h = linear_solver_HIP("symmetric"); initialize_b(h)
h.ls_factor_b(K, n, m)
for rep in range(3):
    t0 = time.perf_counter(); dim, colptr, rowval, nzval, base = csc_arrays(K); t1 = time.perf_counter()
    h._lib.okkt_analyze(h._h, dim, L.p_i64(colptr), L.p_i64(rowval), base); t2 = time.perf_counter()
    inert = L.OkktInertia()
    h._lib.okkt_factor(h._h, L.p_f64(nzval), n, m, L.OKKT_SYM_SYMMETRIC, C.byref(inert)); t3 = time.perf_counter()
    st = h.stats()
    print(f"csc_arrays {1e3*(t1-t0):.2f} ms  okkt_analyze (cached pattern) {1e3*(t2-t1):.2f} ms  okkt_factor {1e3*(t3-t2):.2f} ms  (device {st['last_factor_ms']:.2f} ms)")
    t0 = time.perf_counter(); h.ls_factor_b(K, n, m); print(f"ls_factor_b total {1e3*(time.perf_counter()-t0):.2f} ms")
b = np.random.default_rng(0).normal(size=n + m)
for rep in range(3):
    t0 = time.perf_counter(); x = h.ls_solve(b); print(f"ls_solve total {1e3*(time.perf_counter()-t0):.2f} ms (device {h.stats()['last_solve_ms']:.2f})")
finalize_b(h)
