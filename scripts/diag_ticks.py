"""OKKT_DEBUG_DIAG_STOP=9: k_big_diag reports phase times through the inertia counters (ticks of 10 ns x 1000)."""
import sys, numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
prob = synth.make_config("S-C3", seed=0); n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h)
import ctypes as C
from onephase_jl_amd import _lib as L
from onephase_jl_amd.linear_system_solvers import csc_arrays
h.ls_factor_b(K, n, m) if False else None
dim, colptr, rowval, nzval, base = csc_arrays(K)
h._lib.okkt_analyze(h._h, dim, L.p_i64(colptr), L.p_i64(rowval), base)
inert = L.OkktInertia()
for _ in range(2):
    h._lib.okkt_factor(h._h, L.p_f64(nzval), n, m, 1, C.byref(inert))
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 40
c = [int(v) // 1000 for v in inert.as_tuple()]
print("per launch (us): load %.2f  extract+barrier %.2f  compute+barrier %.2f  mfma(wave0) %.2f" % tuple(v * 0.01 / nd for v in c))
finalize_b(h)
