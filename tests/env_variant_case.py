"""Helper of test_gpu_env_variants.py (run as a subprocess: the tuning switches are read once per process).
Factors and solves a dense 2600 x 2600 indefinite matrix (one front, 21 block columns, two inverted diagonal blocks) and
the S-C3 system, checks inertia, residuals and that two solves of the same right-hand side agree bit for bit."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

rng = np.random.default_rng(11)
n = 2600
B = rng.normal(size=(n, n))
M = B + B.T + np.diag(np.where(rng.random(n) < 0.4, 1.0, -1.0) * (3.0 * np.sqrt(n)))
w = np.linalg.eigvalsh(M)
h = linear_solver_HIP("symmetric")
initialize_b(h)
assert h.ls_factor_b(sp.csc_matrix(np.tril(M)), int((w > 0).sum()), int((w < 0).sum())) == 1
b = rng.normal(size=n)
xs = [h.ls_solve(b) for _ in range(4)]
for x in xs:
    assert np.max(np.abs(M @ x - b)) <= 1e-9 * np.sqrt(n) * np.max(np.abs(b))
assert np.array_equal(xs[2], xs[3])
finalize_b(h)

import os
ordering = int(os.environ.get("OKKT_ORDERING_TEST", "0"))
prob = synth.make_config("S-C3", seed=0)
K = synth.augmented_matrix(prob, delta=1e-8)
Ms = synth.symmetrize_lower(K)
h = linear_solver_HIP("symmetric", ordering=ordering)
initialize_b(h)
for rep in range(2):
    assert h.ls_factor_b(K, prob["n"], prob["m"]) == 1
    assert h.inertia == (prob["n"], prob["m"], 0, 0)
    b = rng.normal(size=prob["n"] + prob["m"])
    x = h.ls_solve(b)
    assert np.max(np.abs(Ms @ x - b)) <= 1e-7 * np.max(np.abs(b)) * max(1.0, np.max(np.abs(x)))
    d = h.diag().copy()
    if rep:
        assert np.array_equal(d, d_prev)
    d_prev = d
finalize_b(h)
# a banded system (elimination tree = one long path of small fronts) and a tree of small fronts only
for prob in (synth.hanging_chain(N_h=300, seed=2), synth.make_config("S-small", seed=1, h_per_col=2, j_per_row=3)):
    K = synth.augmented_matrix(prob, delta=0.5)
    Ms = synth.symmetrize_lower(K)
    h = linear_solver_HIP("symmetric")
    initialize_b(h)
    h.ls_factor_b(K, prob["n"], prob["m"])
    w = np.linalg.eigvalsh(Ms.toarray())
    assert h.inertia == (int((w > 0).sum()), int((w < 0).sum()), 0, 0)
    b = rng.normal(size=K.shape[0])
    x = h.ls_solve(b)
    xd = np.linalg.solve(Ms.toarray(), b)
    assert np.max(np.abs(x - xd)) <= 1e-8 * max(1.0, np.max(np.abs(xd)))
    finalize_b(h)
# one dense front of every shape of the last 64-column step of the block substitution (one row, an odd and an even short step, a full one):
# a front without rows below its pivot block, where a pair load behind the last row would reach a never-written entry (OKKT_DEBUG_POISON)
for n in (129, 130, 191, 193, 256, 257, 321, 383):
    B = rng.normal(size=(n, n))
    M = B + B.T + np.diag(np.where(rng.random(n) < 0.5, 1.0, -1.0) * (3.0 * np.sqrt(n)))
    w = np.linalg.eigvalsh(M)
    h = linear_solver_HIP("symmetric")
    initialize_b(h)
    assert h.ls_factor_b(sp.csc_matrix(np.tril(M)), int((w > 0).sum()), int((w < 0).sum())) == 1
    b = rng.normal(size=n)
    x = h.ls_solve(b)
    xd = np.linalg.solve(M, b)
    assert np.max(np.abs(x - xd)) <= 1e-9 * max(1.0, np.max(np.abs(xd))), (n, np.max(np.abs(x - xd)))
    finalize_b(h)
print("VARIANT_OK")
