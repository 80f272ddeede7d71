"""Ordering-quality probe (host only): factor flops / nnz(L) / largest front of the S-metric and S-C3
patterns under AMD and under prototype nested-dissection orderings fed in as user permutations."""
import sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
import onephase_jl_amd as pk
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import linear_solver_HIP, initialize_b, finalize_b


def stats_for(K, ordering, perm=None):
    s = linear_solver_HIP("symmetric", host_symbolic_only=1, ordering=ordering)
    initialize_b(s)
    if perm is not None:
        s.set_perm(perm)
    t = time.time()
    s.analyze(K)
    dt = time.time() - t
    st = s.stats()
    p = s.perm()
    finalize_b(s)
    return st, p, dt


def show(tag, st, dt):
    print(f"{tag:28s} flops_exact {st['flops_exact']:.3e} stored {st['flops_stored']:.3e} nnzL {st['nnzL']:.3e} "
          f"max_front {st['max_front']} nsuper {st['nsuper']} levels {st['nlevels']} analyze {dt:.2f}s", flush=True)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
    prob = synth.make_config(name, seed=0)
    K = sp.tril(synth.augmented_matrix(prob, delta=0.0, with_upper=False), format="csc")
    st, p, dt = stats_for(K, 3)
    show(name + " AMD", st, dt)
    np.save(f"/tmp/{name}_amd_perm.npy", p)


# ---------------- prototype: geometric nested dissection with Koenig vertex separators ----------------
from scipy.sparse.csgraph import maximum_bipartite_matching


def vertex_cover_separator(G, A, B):
    """Smallest vertex separator contained in the edge cut between vertex sets A and B (Koenig)."""
    inA = np.zeros(G.shape[0], bool); inA[A] = True
    inB = np.zeros(G.shape[0], bool); inB[B] = True
    sub = G[A][:, B].tocsr()          # |A| x |B| bipartite crossing edges
    ra = np.unique(sub.nonzero()[0]); cb = np.unique(sub.nonzero()[1])
    if len(ra) == 0:
        return np.zeros(0, np.int64)
    sub2 = sub[ra][:, cb].tocsr()
    match_r = maximum_bipartite_matching(sub2, perm_type="column")   # for each row: matched column or -1
    nr, nc = sub2.shape
    match_c = -np.ones(nc, np.int64)
    for r, c in enumerate(match_r):
        if c >= 0: match_c[c] = r
    # Koenig: Z = vertices reachable from unmatched rows by alternating paths; cover = (rows \ Z) + (cols & Z)
    visited_r = np.zeros(nr, bool); visited_c = np.zeros(nc, bool)
    stack = [r for r in range(nr) if match_r[r] < 0]
    for r in stack: visited_r[r] = True
    indptr, indices = sub2.indptr, sub2.indices
    while stack:
        r = stack.pop()
        for c in indices[indptr[r]:indptr[r + 1]]:
            if not visited_c[c]:
                visited_c[c] = True
                r2 = match_c[c]
                if r2 >= 0 and not visited_r[r2]:
                    visited_r[r2] = True; stack.append(r2)
    cover_r = ra[~visited_r]; cover_c = cb[visited_c]
    return np.concatenate([np.asarray(A)[cover_r], np.asarray(B)[cover_c]])


def leaf_order(Klow, verts):
    sub = Klow[verts][:, verts]
    sub = sp.tril(sub + sub.T, format="csc")
    sub = (sub + sp.eye(len(verts), format="csc")).tocsc()
    sub.sort_indices()
    st, p, _ = stats_for(sub, 3)
    return np.asarray(verts)[p]


def geo_nd(G, Ksym, verts, pos, depth):
    if depth == 0 or len(verts) < 2000:
        return leaf_order(Ksym, verts)
    med = np.median(pos[verts])
    A = verts[pos[verts] <= med]; B = verts[pos[verts] > med]
    S = vertex_cover_separator(G, A, B)
    inS = np.zeros(G.shape[0], bool); inS[S] = True
    A = A[~inS[A]]; B = B[~inS[B]]
    print(f"  depth {depth}: |A| {len(A)} |B| {len(B)} |S| {len(S)}", flush=True)
    return np.concatenate([geo_nd(G, Ksym, A, pos, depth - 1), geo_nd(G, Ksym, B, pos, depth - 1), S])


def run_geo(name, depth):
    prob = synth.make_config(name, seed=0)
    n, m = prob["n"], prob["m"]
    K = sp.tril(synth.augmented_matrix(prob, delta=0.0, with_upper=False), format="csc")
    Ksym = (K + K.T).tocsr()
    G = Ksym.copy(); G.setdiag(0); G.eliminate_zeros()
    pos = np.concatenate([np.arange(n, dtype=float), np.arange(m, dtype=float) * n / m])
    perm = geo_nd(G, Ksym, np.arange(n + m), pos, depth)
    assert len(np.unique(perm)) == n + m
    st, p, dt = stats_for(K, 2, perm)
    show(f"{name} geoND depth {depth}", st, dt)


if len(sys.argv) > 2 and sys.argv[2] == "geo":
    for d in [int(x) for x in sys.argv[3:]]:
        run_geo(sys.argv[1], d)
