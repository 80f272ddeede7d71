"""Profile target: one factorisation, then `reps` solves (rocprofv3 --kernel-trace --stats -- python3 scripts/solve_profile.py S-metric 20 [nrhs])."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h)
h.analyze(K)
d_vals = h.dev_upload(K.data)
rc = h.ls_factor_dev(d_vals, n, m)
B = np.random.default_rng(0).normal(size=(nrhs, n + m))
d_rhs = h.dev_upload(B); d_sol = h.dev_alloc(8 * nrhs * (n + m))
ms = []
for r in range(reps):
    h.ls_solve_dev(d_rhs, d_sol, nrhs)
    ms.append(h.stats()["last_solve_ms"])
st = h.stats()
X = h.dev_download(d_sol, (nrhs, n + m))
M = synth.symmetrize_lower(K)
res = max(np.max(np.abs(M @ X[r] - B[r])) / np.max(np.abs(B[r])) for r in range(nrhs))
print(f"{name} nrhs={nrhs}: rc={rc} factor {st['last_factor_ms']:.2f} ms, solve min {min(ms):.3f} median {np.median(ms):.3f} ms ({min(ms)/nrhs:.3f} per rhs), resid {res:.2e}, nnzL_stored {st['nnzL_stored']}, "
      f"effective {2*8*st['nnzL_stored']*nrhs/min(ms)/1e9:.2f} TB/s")
finalize_b(h)
