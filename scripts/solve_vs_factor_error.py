"""GPU box: is the distance of the HIP solution from the extended-precision one in the FACTOR or in the SOLVE?  The HIP factor (L, D) is
downloaded and the triangular solves are repeated on the CPU by plain column substitution in fp64; forward errors of (a) the HIP solve,
(b) substitution with the HIP factor, (c) the CPU restatement (its own factor and substitution).  usage: solve_vs_factor_error.py <config>"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from onephase_jl_amd import synth
import oracle
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
name = sys.argv[1] if len(sys.argv) > 1 else "S-C5"
prob = synth.make_config(name, seed=0); n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
M = synth.symmetrize_lower(K).tocsr()
h = linear_solver_HIP("symmetric"); initialize_b(h)
assert h.ls_factor_b(K, n, m) == 1
perm = np.asarray(h.perm()); L = h.factor_csc().tocsc(); d = h.diag().copy()
N = n + m
o = oracle.linear_solver_ORACLE_MF("symmetric", perm=perm, nthreads=8); o._analyze(K); assert o.ls_factor_b(K, n, m) == 1
Lp, Li, Lx = L.indptr, L.indices, L.data
def subst(b):
    y = b[perm].copy()
    for j in range(N):
        lo, hi = Lp[j], Lp[j + 1]
        rows = Li[lo:hi]; vals = Lx[lo:hi]
        msk = rows > j
        if msk.any(): y[rows[msk]] -= vals[msk] * y[j]
    y /= d
    for j in range(N - 1, -1, -1):
        lo, hi = Lp[j], Lp[j + 1]
        rows = Li[lo:hi]; vals = Lx[lo:hi]
        msk = rows > j
        if msk.any(): y[j] -= np.dot(vals[msk], y[rows[msk]])
    x = np.empty(N); x[perm] = y
    return x
rng = np.random.default_rng(3)
for b in rng.normal(size=(2, N)):
    xo = o.ls_solve(b); xt = xo.copy()
    for _ in range(4):
        prod = M.data.astype(np.longdouble) * xt.astype(np.longdouble)[M.indices]
        r = (b.astype(np.longdouble) - np.add.reduceat(prod, M.indptr[:-1])).astype(np.float64)
        xt = xt + o.ls_solve(r)
    xh = h.ls_solve(b); xs = subst(b)
    sc = np.max(np.abs(xt))
    print(f"{name}: forward error  HIP solve {np.max(np.abs(xh - xt)) / sc:.2e} | substitution with the HIP factor {np.max(np.abs(xs - xt)) / sc:.2e} | CPU restatement {np.max(np.abs(xo - xt)) / sc:.2e}")
finalize_b(h)
