"""GPU debugging aid: forward sweep of the HIP solve against a host triangular solve with the SAME L and D."""
import os, sys
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
name = sys.argv[1]
if name == "S-C5": prob = synth.make_config("S-C5", seed=0)
elif name.startswith("blk"):
    nb, n_b = [int(v) for v in name[3:].split("x")]
    prob = synth.block_angular(nblocks=nb, n_b=n_b, m_b=n_b * 3 // 2, n_link=200, seed=0)
else: prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h)
rc = h.ls_factor_b(K, n, m)
st = h.stats()
print(name, "rc", rc, {k: st[k] for k in ("n", "nsuper", "max_front", "n_big_fronts", "nlevels")}, flush=True)
b = np.random.default_rng(1).normal(size=n + m)
x = h.ls_solve(b)
perm = h.perm()
if os.environ.get("OKKT_DEBUG_SOLVE_PHASE") == "1":
    L = h.factor_csc(); D = h.diag()
    Lu = (L + sp.identity(n + m, format="csc")).tocsr()
    z_ref = spla.spsolve_triangular(Lu, b[perm], lower=True, unit_diagonal=True) / D
    z = x[perm]
    err = np.abs(z - z_ref) / (np.abs(z_ref) + 1e-300)
    bad = np.flatnonzero(err > 1e-6)
    print("forward: max rel err %.2e, bad entries %d of %d, first bad permuted columns %s" % (err.max(), len(bad), n + m, bad[:12]))
    par, cnt = h.etree()
    if len(bad):
        j = bad[0]
        print("first bad column", j, "colcount", cnt[j], "parent", par[j], "z", z[j], "ref", z_ref[j])
        # columns of the same supernode chain around it
        for jj in range(max(0, j - 3), min(n + m, j + 4)): print("   col", jj, "count", cnt[jj], "par", par[jj], "err %.2e" % err[jj])
else:
    M = synth.symmetrize_lower(K)
    print("resid %.2e" % (np.max(np.abs(M @ x - b)) / (np.max(np.abs(b)) * max(1.0, np.max(np.abs(x))))))
finalize_b(h)
