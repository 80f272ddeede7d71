"""Randomised structures through the C ABI against dense LAPACK: quasi-definite KKT matrices [[H, J'], [J, -D]] of random size,
pattern family, ordering and amalgamation; inertia from the dense eigenvalues, solutions (1 and 3 right-hand sides) from a dense solve,
two solves bit for bit equal.  Usage: python scripts/fuzz_gpu.py [cases] [seed] [scale].  Prints one line per failure and a summary."""
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
from onephase_jl_amd import _lib as L
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
fails = 0
fam_count = {}
import os
only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None      # FUZZ_ONLY=<case>: that case alone (same seeds)
for case in range(cases):
    if only is not None and case != only: continue
    rng = np.random.default_rng(1000 * seed0 + case)
    fam = ["random", "band", "arrow", "blocks", "grid", "dense_rows", "wide_band", "tree"][case % 8]
    n = int(rng.integers(1, int(700 * scale)))
    m = int(rng.integers(0, int(900 * scale)))
    if fam == "random":
        J = sp.random(m, n, density=min(1.0, rng.uniform(1, 6) / max(n, 1)), random_state=rng, format="lil")
    elif fam in ("band", "wide_band"):
        offs = list(range(int(rng.integers(1, 6)))) if fam == "band" else list(range(0, int(rng.integers(20, 90)), 7))
        m = min(m, n)
        J = sp.lil_matrix((m, n))
        for k in offs:
            ln = min(m, n - k)
            if ln > 0: J.setdiag(rng.normal(size=ln) + (2.0 if k == 0 else 0.0), k)
    elif fam == "arrow":
        J = sp.lil_matrix((m, n))
        if m and n:
            J[:, 0] = rng.normal(size=(m, 1)); J.setdiag(rng.normal(size=min(m, n)) + 2.0)
    elif fam == "blocks":
        J = sp.lil_matrix((m, n)); b = int(rng.integers(2, 40))
        for i in range(m):
            c0 = (i * b) % max(n - b, 1); J[i, c0:c0 + min(b, n)] = rng.normal(size=min(b, n - c0) if c0 + b <= n else n - c0)
    elif fam == "grid":
        g = int(rng.integers(3, int(26 * scale ** 0.5))); n = g * g
        ex = np.ones(g); T = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
        Hg = sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g))
        J = sp.random(m, n, density=min(1.0, 2.0 / n), random_state=rng, format="lil")
    elif fam == "dense_rows":
        J = sp.random(m, n, density=min(1.0, 2.0 / max(n, 1)), random_state=rng, format="lil")
        for i in range(min(m, int(rng.integers(1, 4)))):
            cols = rng.choice(n, size=max(1, int(n * rng.uniform(0.3, 0.9))), replace=False); J[i, cols] = rng.normal(size=len(cols))
    else:  # tree: each row couples a node with its parent in a random tree
        m = max(n - 1, 0); J = sp.lil_matrix((m, n))
        for i in range(m):
            J[i, i + 1] = 1.0 + rng.random(); J[i, int(rng.integers(0, i + 1))] = -1.0
    J = sp.csc_matrix(J)
    if fam == "grid":
        H = sp.tril(Hg + sp.diags(0.5 + rng.random(n)))
    else:
        Hs = sp.random(n, n, density=min(1.0, rng.uniform(0, 3) / max(n, 1)), random_state=rng)
        Hs = Hs + Hs.T
        H = sp.tril(Hs + sp.diags(np.asarray(abs(Hs).sum(axis=1)).ravel() + 0.5 + rng.random(n)))
    K = sp.bmat([[H, None], [J, -sp.diags(0.5 + rng.random(m))]], format="csc") if m else sp.csc_matrix(H)
    N = n + m
    Kd = K.toarray(); Md = np.tril(Kd) + np.tril(Kd, -1).T
    w = np.linalg.eigvalsh(Md)
    ordering = [0, 0, 3, 4, 5][int(rng.integers(0, 5))]
    opts = {"ordering": ordering}
    if rng.random() < 0.3: opts.update(relax_small=int(rng.choice([64, 128, 512, 1024])), relax_small_frac=float(rng.choice([0.1, 0.3, 0.6])))
    tag = f"case {case} {fam} n={n} m={m} {opts}"
    # how the matrix is handed over: scipy CSC, or a raw CSC through the C ABI's contract -- entries split into duplicates that have
    # to be summed, an upper triangle full of garbage that has to be ignored (julia.jl:34: Symmetric(A, :L)), 1-based indices
    form = ["scipy", "scipy", "duplicates", "upper_garbage", "one_based"][int(rng.integers(0, 5))]
    Karg = K
    if form != "scipy" and N > 0:
        Kc = sp.csc_matrix(sp.tril(K)); Kc.sort_indices()
        cols = np.repeat(np.arange(N), np.diff(Kc.indptr)); rows = Kc.indices.copy(); vals = Kc.data.copy()
        if form == "duplicates":
            pick = rng.random(len(vals)) < 0.3
            frac = rng.uniform(0.2, 0.8, size=int(pick.sum()))
            rows = np.concatenate([rows, rows[pick]]); cols = np.concatenate([cols, cols[pick]])
            extra = vals[pick] * frac; vals[pick] -= extra; vals = np.concatenate([vals, extra])
        elif form == "upper_garbage":
            ng = max(1, len(vals) // 3)
            gr = rng.integers(0, N, size=ng); gc = rng.integers(0, N, size=ng); keep = gr < gc
            rows = np.concatenate([rows, gr[keep]]); cols = np.concatenate([cols, gc[keep]]); vals = np.concatenate([vals, 1e3 * rng.normal(size=int(keep.sum()))])
        order = np.lexsort((rng.random(len(rows)) if form == "duplicates" else rows, cols))
        rows, cols, vals = rows[order], cols[order], vals[order]
        colptr = np.zeros(N + 1, dtype=np.int64); np.add.at(colptr, cols + 1, 1); colptr = np.cumsum(colptr)
        base = 1 if form == "one_based" else 0
        Karg = (N, colptr + base, rows.astype(np.int64) + base, vals, base)
    tag += f" {form}"
    try:
        h = linear_solver_HIP("symmetric", **opts); initialize_b(h)
        rc = h.ls_factor_b(Karg, n, m)
        want = (int((w > 0).sum()), int((w < 0).sum()), 0, 0)
        ok = h.inertia == want and rc == (1 if want[:2] == (n, m) else 0)
        B = rng.normal(size=(3, N))
        xd = np.linalg.solve(Md, B.T).T
        x1 = h.ls_solve(B[0]); x2 = h.ls_solve(B[0])
        X = np.zeros_like(B)
        h._check(h._lib.okkt_solve(h._h, L.p_f64(B), L.p_f64(X), 3), "okkt_solve")
        sc = max(1.0, np.max(np.abs(xd)))
        e1 = np.max(np.abs(x1 - xd[0])) / sc; e3 = np.max(np.abs(X - xd)) / sc
        tol = 1e-9 * max(1.0, np.linalg.cond(Md) * 1e-3)
        ok = ok and np.array_equal(x1, x2) and e1 <= tol and e3 <= tol
        if not ok:
            fails += 1
            print("FAIL", tag, "inertia", h.inertia, "want", want, "rc", rc, "e1 %.2e e3 %.2e tol %.2e" % (e1, e3, tol), "bitwise", np.array_equal(x1, x2), flush=True)
        finalize_b(h)
    except Exception as ex:   # noqa
        fails += 1
        print("EXC", tag, repr(ex)[:300], flush=True)
    fam_count[fam] = fam_count.get(fam, 0) + 1
print("FUZZ cases", cases, "failures", fails, fam_count)
