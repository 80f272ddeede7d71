"""GPU box: which part of the HIP solve loses accuracy?  The HIP factor is downloaded (stored pattern = the supernodes with their relaxation
zeros) and the two sweeps are repeated on the CPU supernode by supernode, the pivot block of a supernode solved (a) by substitution, (b) by an
explicit inverse of blocks of at most 32 / 64 / 128 columns + block substitution between them, (c) by an explicit inverse of blocks of at most
1024 columns built by recursive doubling (what csrc/solve.hip does; 128-column blocks for the fronts of at most 128 pivot columns).  Forward errors against the extended-precision
solution.  usage: solve_emulation.py <config>"""
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
Lp, Li, Lx = L.indptr, L.indices, L.data
# supernodes of the stored pattern
sn = [0]
for j in range(N - 1):
    a = Li[Lp[j]:Lp[j + 1]]; b = Li[Lp[j + 1]:Lp[j + 2]]
    a = a[a > j]; b = b[b >= j + 1]
    bb = b[b > j + 1] if (len(b) and b[0] == j + 1) else b
    same = len(a) >= 1 and a[0] == j + 1 and len(a) - 1 == len(bb) and np.array_equal(a[1:], bb)
    if not same: sn.append(j + 1)
sn.append(N)
ks = np.diff(sn)
print(f"{name}: {len(ks)} supernodes in the stored pattern, widest {ks.max()}, with more than 128 columns: {(ks > 128).sum()}, more than 1024: {(ks > 1024).sum()}")
blocks = []
for a in range(len(ks)):
    j0, j1 = sn[a], sn[a + 1]; k = j1 - j0
    rows = Li[Lp[j0]:Lp[j0 + 1]]; rows = rows[rows >= j0]
    if len(rows) == 0 or rows[0] != j0: rows = np.concatenate(([j0], rows))
    f = len(rows)
    F = np.zeros((f, k))
    pos = {int(r): i for i, r in enumerate(rows)} if k > 1 else None
    for c in range(k):
        lo, hi = Lp[j0 + c], Lp[j0 + c + 1]
        rr = Li[lo:hi]; vv = Lx[lo:hi]; msk = rr > j0 + c
        if k == 1: F[np.searchsorted(rows, rr[msk]), c] = vv[msk]
        else: F[[pos[int(r)] for r in rr[msk]], c] = vv[msk]
    blocks.append((j0, k, rows, F))
def inv_sub(B):
    nb = B.shape[0]; X = np.eye(nb)
    for p in range(nb): X[p + 1:, :] -= np.outer(B[p + 1:, p], X[p, :])
    return X
def inv_doubling(B, base=32):
    nb = B.shape[0]
    if nb <= base: return inv_sub(B)
    hlf = (nb // 2 + base - 1) // base * base
    if hlf >= nb: hlf = nb // 2
    XA, XD = inv_doubling(B[:hlf, :hlf], base), inv_doubling(B[hlf:, hlf:], base)
    X = np.zeros_like(B); X[:hlf, :hlf] = XA; X[hlf:, hlf:] = XD; X[hlf:, :hlf] = -XD @ (B[hlf:, :hlf] @ XA)
    return X
cache = {}
def pivot_solve(a, B, r, mode, trans):
    k = len(r)
    if k == 1: return r
    if mode == "subst":
        y = r.copy()
        if not trans:
            for p in range(k): y[p + 1:] -= B[p + 1:, p] * y[p]
        else:
            for p in range(k - 1, -1, -1): y[p] -= B[p + 1:, p] @ y[p + 1:]
        return y
    bs = int(mode[3:])
    key = (a, bs)
    if key not in cache: cache[key] = [(i0, min(i0 + bs, k), inv_doubling(B[i0:min(i0 + bs, k), i0:min(i0 + bs, k)])) for i0 in range(0, k, bs)]
    y = r.copy()
    if not trans:
        for i0, i1, X in cache[key]:
            y[i0:i1] = X @ y[i0:i1]
            y[i1:] -= B[i1:, i0:i1] @ y[i0:i1]
    else:
        for i0, i1, X in reversed(cache[key]):
            y[i0:i1] -= B[i1:, i0:i1].T @ y[i1:]
            y[i0:i1] = X.T @ y[i0:i1]
    return y
def solve(b, mode):
    y = b[perm].copy()
    for a, (j0, k, rows, F) in enumerate(blocks):
        B = np.tril(F[:k, :k], -1) + np.eye(k)
        yk = pivot_solve(a, B, y[j0:j0 + k], mode, False)
        y[j0:j0 + k] = yk
        if len(rows) > k: y[rows[k:]] -= F[k:, :] @ yk
    y /= d
    for a in range(len(blocks) - 1, -1, -1):
        j0, k, rows, F = blocks[a]
        B = np.tril(F[:k, :k], -1) + np.eye(k)
        t = y[j0:j0 + k].copy()
        if len(rows) > k: t -= F[k:, :].T @ y[rows[k:]]
        y[j0:j0 + k] = pivot_solve(a, B, t, mode, True)
    x = np.empty(N); x[perm] = y
    return x
o = oracle.linear_solver_ORACLE_MF("symmetric", perm=perm, nthreads=8); o._analyze(K); assert o.ls_factor_b(K, n, m) == 1
rng = np.random.default_rng(3)
for b in rng.normal(size=(2, N)):
    xo = o.ls_solve(b); xt = xo.copy()
    for _ in range(4):
        prod = M.data.astype(np.longdouble) * xt.astype(np.longdouble)[M.indices]
        r = (b.astype(np.longdouble) - np.add.reduceat(prod, M.indptr[:-1])).astype(np.float64)
        xt = xt + o.ls_solve(r)
    sc = np.max(np.abs(xt)); e = lambda x: np.max(np.abs(x - xt)) / sc
    print(f"{name}: HIP solve {e(h.ls_solve(b)):.2e} | CPU, HIP factor: substitution {e(solve(b, 'subst')):.2e}, inverses of <= 32 / 64 / 128 / 1024 columns {e(solve(b, 'inv32')):.2e} / {e(solve(b, 'inv64')):.2e} / {e(solve(b, 'inv128')):.2e} / {e(solve(b, 'inv1024')):.2e} | CPU restatement {e(xo):.2e}")
finalize_b(h)
