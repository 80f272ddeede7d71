"""Timing probe (GPU box): factor / solve device times and residuals for one synthetic config."""
import sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP

name = sys.argv[1] if len(sys.argv) > 1 else "S-C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
opts = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    opts[k] = float(v) if "." in v else int(v)
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
schur = opts.pop("schur", 0)
if schur:
    K = synth.schur_matrix(prob, delta=1e-8); n, m = n, 0
else:
    K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("definite" if schur else "symmetric", **opts); initialize_b(h)
t = time.time(); rc = h.ls_factor_b(K, n, m); t1 = time.time() - t
st = h.stats()
print(name, "rc", rc, "inertia", h.inertia, "first call s", round(t1, 3), "analyze s", round(st["analyze_seconds"], 3))
print({k: st[k] for k in ("nnz_lower", "nnzL", "nnzL_stored", "flops_exact", "flops_stored", "arena_bytes", "nsuper", "nlevels", "max_front", "n_big_fronts")})
M = (sp.tril(K) + sp.tril(K, -1).T).tocsc()
b = np.random.default_rng(0).normal(size=n + m)
for r in range(reps):
    t = time.time(); rc = h.ls_factor_b(K, n, m); tf = time.time() - t
    t = time.time(); x = h.ls_solve(b); ts = time.time() - t
    st = h.stats()
    res = np.max(np.abs(M @ x - b)) / np.max(np.abs(b))
    print(f"rep {r}: factor dev {st['last_factor_ms']:.3f} ms (wall {tf*1e3:.1f}), solve dev {st['last_solve_ms']:.3f} ms (wall {ts*1e3:.1f}), "
          f"TF/s {st['flops_stored']/st['last_factor_ms']/1e9:.2f}, resid {res:.2e}")
finalize_b(h)
