"""Forward error of the HIP solve and of the oracle's against the extended-precision solution of the same fp64 matrix
(oracle solve + refinement with long-double residuals until the correction is at rounding level).  Usage: forward_error.py <config> [opt=value ...]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP
name = sys.argv[1] if len(sys.argv) > 1 else "S-C3"
opts = {}
for a in sys.argv[2:]:
    k, v = a.split("="); opts[k] = float(v) if "." in v else int(v)
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
M = synth.symmetrize_lower(K).tocsr()
h = linear_solver_HIP("symmetric", **opts); initialize_b(h)
assert h.ls_factor_b(K, n, m) == 1
o = oracle.linear_solver_ORACLE_MF("symmetric", perm=h.perm(), nthreads=32)
o._analyze(K); assert o.ls_factor_b(K, n, m) == 1
rng = np.random.default_rng(0)
for b in rng.normal(size=(2, n + m)):
    xo = o.ls_solve(b); xt = xo.copy()
    for it in range(4):
        prod = M.data.astype(np.longdouble) * xt.astype(np.longdouble)[M.indices]
        r = (b.astype(np.longdouble) - np.add.reduceat(prod, M.indptr[:-1])).astype(np.float64)
        xt = xt + o.ls_solve(r)
    xh = h.ls_solve(b)
    sc = np.max(np.abs(xt))
    print(name, opts, "forward error: oracle %.2e  HIP %.2e  | HIP vs oracle %.2e | residual HIP %.2e oracle %.2e" % (
        np.max(np.abs(xo - xt)) / sc, np.max(np.abs(xh - xt)) / sc, np.max(np.abs(xh - xo)) / sc,
        np.max(np.abs(M @ xh - b)) / np.max(np.abs(b)), np.max(np.abs(M @ xo - b)) / np.max(np.abs(b))), flush=True)
finalize_b(h)
