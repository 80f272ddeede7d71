"""Soak: many factor + solve cycles (and handle create / destroy cycles) -- timings must not drift, device memory must not grow."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import ctypes as C
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
prob = synth.make_config("S-C3", seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
b = np.random.default_rng(0).normal(size=n + m)
h = linear_solver_HIP("symmetric"); initialize_b(h)
h.ls_factor_b(K, n, m); x0 = h.ls_solve(b)
f0 = free_mb()
ts = []
for i in range(300):
    t = time.perf_counter(); assert h.ls_factor_b(K, n, m) == 1; x = h.ls_solve(b); ts.append(time.perf_counter() - t)
    assert np.array_equal(x, x0)
print(f"300 cycles: first 10 avg {1e3*np.mean(ts[:10]):.2f} ms, last 10 avg {1e3*np.mean(ts[-10:]):.2f} ms, free memory change {free_mb() - f0:.1f} MiB")
finalize_b(h)
f1 = free_mb()
for i in range(30):
    h = linear_solver_HIP("symmetric"); initialize_b(h); assert h.ls_factor_b(K, n, m) == 1; finalize_b(h)
print(f"30 handle cycles: free memory change {free_mb() - f1:.1f} MiB")
