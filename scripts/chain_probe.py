"""Banded KKT (BASELINE config 2 stand-in): factor / solve device times under the orderings and amalgamation widths."""
import sys, numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ordering = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prob = synth.hanging_chain(N_h=N, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1.0)
h = linear_solver_HIP("symmetric", ordering=ordering); initialize_b(h)
b = np.random.default_rng(0).normal(size=n + m)
import scipy.sparse as sp
M = (sp.tril(K) + sp.tril(K, -1).T).tocsc()
fm, sm = [], []
for _ in range(6):
    rc = h.ls_factor_b(K, n, m); x = h.ls_solve(b)
    st = h.stats(); fm.append(st["last_factor_ms"]); sm.append(st["last_solve_ms"])
res = np.max(np.abs(M @ x - b)) / np.max(np.abs(b))
print(f"N_h {N} ordering {ordering} used {st['ordering_used']} chain {st['critical_pivots']} nlevels {st['nlevels']} nsuper {st['nsuper']} rc {rc} factor {np.median(fm):.3f} ms solve {np.median(sm):.3f} ms resid {res:.1e}")
finalize_b(h)
