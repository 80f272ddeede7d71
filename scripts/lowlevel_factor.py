"""N numeric factorisations through the C ABI, return codes ignored (timing-only ablation builds give wrong pivot counts)."""
import sys, ctypes as C
sys.path.insert(0, ".")
from onephase_jl_amd import synth, _lib as L
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP, csc_arrays
prob = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "S-C3", seed=0); n, m = prob["n"], prob["m"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h)
dim, colptr, rowval, nzval, base = csc_arrays(K)
h._lib.okkt_analyze(h._h, dim, L.p_i64(colptr), L.p_i64(rowval), base)
inert = L.OkktInertia()
for _ in range(reps):
    h._lib.okkt_factor(h._h, L.p_f64(nzval), n, m, 1, C.byref(inert))
print("done", h.stats()["last_factor_ms"])
finalize_b(h)
