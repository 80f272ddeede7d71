"""A mesh-like KKT system (H = 3-D grid Laplacian + I, a few constraints per node group): the automatic ordering against minimum
degree and the multilevel dissection -- flops, device times, residual, agreement with the CPU port.  Usage: grid_probe.py [g]"""
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
import oracle
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP
g = int(sys.argv[1]) if len(sys.argv) > 1 else 36
n = g ** 3
e = np.ones(g); T = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1]); I = sp.identity(g)
H = sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + sp.identity(n)
rng = np.random.default_rng(0)
m = n // 8
rows = np.repeat(np.arange(m), 2); cols = np.minimum(n - 1, 8 * rows + rng.integers(0, 8, size=2 * m))
J = sp.csc_matrix((rng.normal(size=2 * m), (rows, cols)), shape=(m, n))
K = sp.bmat([[sp.tril(H), None], [J, -sp.diags(0.5 + rng.random(m))]], format="csc")
M = (sp.tril(K) + sp.tril(K, -1).T).tocsr()
b = rng.normal(size=n + m)
for ordering in (0, 3, 5, 4):
    h = linear_solver_HIP("symmetric", ordering=ordering); initialize_b(h)
    rc = h.ls_factor_b(K, n, m)
    for _ in range(3): rc = h.ls_factor_b(K, n, m); x = h.ls_solve(b)
    st = h.stats()
    res = np.max(np.abs(M @ x - b)) / np.max(np.abs(b))
    line = f"g={g} n+m={n + m} ordering {ordering} used {st['ordering_used']} rc {rc} flops {st['flops_exact']:.3g} levels {st['nlevels']} max_front {st['max_front']} analyze {st['analyze_seconds']:.2f} s factor {st['last_factor_ms']:.2f} ms solve {st['last_solve_ms']:.2f} ms resid {res:.1e}"
    if ordering == 0:
        o = oracle.linear_solver_ORACLE_MF("symmetric", perm=h.perm(), nthreads=32); o._analyze(K)
        assert o.ls_factor_b(K, n, m) == rc
        xo = o.ls_solve(b)
        line += f" | vs CPU port {np.max(np.abs(x - xo)) / np.max(np.abs(xo)):.1e}"
    print(line, flush=True)
    finalize_b(h)
