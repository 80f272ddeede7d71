"""Helper of test_gpu_dataflow.py (run as a subprocess: the switches are read once per process): factors one system through the
C ABI and writes D, the stored factor, a solution and the inertia to an .npz file.
usage: python tests/dataflow_case.py <case> <out.npz>      case: dense<n> | dense<n>k<k> | a synth configuration name"""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

name, out = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(11)
if name.startswith("dense"):
    # one dense front; "dense700k300": an arrow matrix whose root front has 700 rows but the pattern forces a second front
    n = int(name[5:])
    B = rng.normal(size=(n, n))
    M = B + B.T + np.diag(np.where(rng.random(n) < 0.4, 1.0, -1.0) * (3.0 * np.sqrt(n)))
    w = np.linalg.eigvalsh(M)
    K = sp.csc_matrix(np.tril(M))
    npos, nneg = int((w > 0).sum()), int((w < 0).sum())
else:
    prob = synth.make_config(name, seed=0)
    K = synth.augmented_matrix(prob, delta=1e-8)
    npos, nneg = prob["n"], prob["m"]
h = linear_solver_HIP("symmetric")
initialize_b(h)
rc = 0
for _ in range(2):
    rc = h.ls_factor_b(K, npos, nneg)
b = rng.normal(size=K.shape[0])
x = h.ls_solve(b)
L = h.factor_csc()
np.savez(out, d=h.diag().copy(), x=x, rc=rc, inertia=np.array(h.inertia), Ldata=L.data, Lidx=L.indices, want=np.array([npos, nneg]))
finalize_b(h)
print("CASE_OK")
