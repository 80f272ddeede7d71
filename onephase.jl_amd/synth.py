"""Seeded synthetic KKT inputs.

The reference's real inputs (CUTEst SIF files, Netlib lpi_*.mat) are not in the repository and
cannot be loaded here, so the path is exercised on generated (H, J, s, y) with the layouts the
reference's iterate cache uses (/root/reference/src/utils/Class_iterate.jl:4-20): H is n x n,
LOWER TRIANGLE ONLY (Class_cutest.jl:548), J is m x n, s, y > 0.

Locality model: variable i couples to columns i + round(N(0, w)); a fraction p_far of the
couplings is uniformly random, which controls fill.
"""
import numpy as np
import scipy.sparse as sp

CONFIGS = {
    # name: n, m, J nnz/row, tril(H) nnz/col (incl. diagonal)
    "S-metric": dict(n=40_000, m=60_000, j_per_row=24, h_per_col=10),
    "S-C3": dict(n=10_000, m=20_000, j_per_row=20, h_per_col=8),
    # the fully random variant of BASELINE config 3 ("random sparse", SURVEY 8d: p_far = 1): no locality, no separators for a
    # dissection to find, near-dense fronts -- the case where the ordering gain vanishes and the MFMA kernels carry the number;
    # the small one is what the scalar oracle affords in a test
    "S-C3-random": dict(n=10_000, m=20_000, j_per_row=20, h_per_col=8, p_far=1.0),
    "S-C3-random-small": dict(n=1_000, m=2_000, j_per_row=20, h_per_col=8, p_far=1.0),
    # the metric family at four times the size (n + m = 4e5; the largest front stays below the 46 000 rows the 32-bit local offsets allow): what the release of the contribution blocks (round 6) is for -- with an
    # f x f buffer per front the arena of this instance does not fit one GPU
    "S-metric-4x": dict(n=160_000, m=240_000, j_per_row=24, h_per_col=10),
    "S-small": dict(n=400, m=600, j_per_row=6, h_per_col=4),
    "S-tiny": dict(n=40, m=60, j_per_row=4, h_per_col=3),
}


def _local_cols(rng, centers, per, ncols, w, p_far):
    """`per` column indices for every center (may repeat; duplicates are merged later)."""
    k = len(centers)
    off = np.rint(rng.normal(0.0, w, size=(k, per))).astype(np.int64)
    cols = centers[:, None] + off
    far = rng.random((k, per)) < p_far
    cols = np.where(far, rng.integers(0, ncols, size=(k, per)), cols)
    return np.clip(cols, 0, ncols - 1)


def make_problem(n, m, j_per_row=8, h_per_col=4, w=50.0, p_far=0.01, seed=0, convex=True,
                 neg_shift=0.0, mu=1e-2, well_scaled=False):
    """Returns dict(H=csc lower, J=csc, s, y, mu, n, m)."""
    rng = np.random.default_rng(seed)
    # J: row r is centred on column ~ r * n / m
    centers = (np.arange(m, dtype=np.int64) * n) // max(m, 1)
    jc = _local_cols(rng, centers, j_per_row, n, w, p_far)
    jr = np.repeat(np.arange(m, dtype=np.int64), j_per_row)
    jv = rng.normal(size=m * j_per_row)
    J = sp.coo_matrix((jv, (jr, jc.ravel())), shape=(m, n)).tocsc()
    J.sum_duplicates()
    J.sort_indices()
    # H: strictly lower couplings + diagonal
    hp = max(h_per_col - 1, 0)
    if hp > 0:
        hc = np.repeat(np.arange(n, dtype=np.int64), hp)
        hr = _local_cols(rng, np.arange(n, dtype=np.int64), hp, n, w, p_far).ravel()
        lo = np.minimum(hr, hc)
        hi = np.maximum(hr, hc)
        keep = hi != lo
        hv = rng.normal(size=keep.sum())
        Hoff = sp.coo_matrix((hv, (hi[keep], lo[keep])), shape=(n, n)).tocsc()
        Hoff.sum_duplicates()
    else:
        Hoff = sp.csc_matrix((n, n))
    absrow = np.asarray(abs(Hoff).sum(axis=0)).ravel() + np.asarray(abs(Hoff).sum(axis=1)).ravel()
    if convex:
        d = absrow + 1.0 + rng.random(n)
    else:
        d = rng.normal(size=n)
    d = d - neg_shift
    H = (Hoff + sp.diags(d, format="csc")).tocsc()
    H.sort_indices()
    # s, y: log-uniform slack, complementarity products spread around mu (parameters.jl:97)
    if well_scaled:  # s, y = O(1): condition number of K stays moderate
        s = np.exp(rng.uniform(np.log(0.2), np.log(5.0), size=m))
        y = np.exp(rng.uniform(np.log(0.2), np.log(5.0), size=m))
    else:
        s = np.exp(rng.uniform(np.log(1e-4), np.log(1e2), size=m))
        y = mu * np.exp(rng.uniform(np.log(1e-2), np.log(1e2), size=m)) / s
    return dict(H=H, J=J, s=s, y=y, mu=mu, n=n, m=m)


def make_config(name, seed=0, **over):
    if name == "S-C5":   # block-angular: 8 independent blocks + 200 linking columns (wide elimination tree)
        return block_angular(seed=seed, **over)
    if name == "S-C5-wide":   # the same tree shape with blocks of the S-C3 size (n + m = 240 200): a part of the sharded run is
        # throughput-bound instead of a chain of small fronts (round-3 review, item 3; labelled: NOT BASELINE's config 5)
        cfg = dict(n_b=10_000, m_b=20_000, j_per_row=20, h_per_col=8)
        cfg.update(over)
        return block_angular(seed=seed, **cfg)
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    return make_problem(seed=seed, **cfg)


def augmented_matrix(prob, delta=0.0, with_upper=True):
    """K = [[H + delta I, J'], [J, -diag(s/y)]] as CSC (symmetric.jl:35-53).  H stays lower-only;
    with_upper=True keeps the J' block exactly like the reference's hvcat (it is ignored by the
    factorisation, Symmetric(K,:L))."""
    H, J, s, y, n, m = prob["H"], prob["J"], prob["s"], prob["y"], prob["n"], prob["m"]
    Hd = H + sp.diags(np.full(n, delta), format="csc") if delta != 0.0 else H
    B = sp.diags(-s / y, format="csc")
    top = sp.hstack([Hd, J.T if with_upper else sp.csc_matrix((n, m))], format="csc")
    bot = sp.hstack([J, B], format="csc")
    K = sp.vstack([top, bot], format="csc")
    K.sort_indices()
    return K


def schur_matrix(prob, delta=0.0):
    """Q = J' diag(y/s) J + H (+ delta I): full-symmetric J'SJ plus lower-only H (schur.jl:55)."""
    H, J, s, y, n = prob["H"], prob["J"], prob["s"], prob["y"], prob["n"]
    Q = (J.T @ sp.diags(y / s) @ J + H).tocsc()
    if delta != 0.0:
        Q = (Q + sp.diags(np.full(n, delta))).tocsc()
    Q.sort_indices()
    return Q


def symmetrize_lower(A):
    """Full symmetric matrix from the lower triangle of A (what Symmetric(A,:L) means)."""
    L = sp.tril(A, format="csc")
    return (L + sp.tril(A, -1, format="csc").T).tocsc()


def block_angular(nblocks=8, n_b=5000, m_b=7500, n_link=200, seed=0, **kw):
    """S-C5: independent diagonal blocks + linking columns that couple all blocks."""
    rng = np.random.default_rng(seed + 1000)
    Hs, Js, ss, ys = [], [], [], []
    for b in range(nblocks):
        p = make_problem(n_b, m_b, seed=seed * 100 + b, **kw)
        Hs.append(p["H"]); Js.append(p["J"]); ss.append(p["s"]); ys.append(p["y"])
    n = nblocks * n_b + n_link
    m = nblocks * m_b
    H = sp.block_diag(Hs + [sp.diags(1.0 + rng.random(n_link))], format="csc")
    Jb = sp.block_diag(Js, format="csc")
    # each linking column touches ~0.2% of the rows of every block
    dens = 0.002
    link = sp.random(m, n_link, density=dens, random_state=np.random.RandomState(seed + 7), format="csc",
                     data_rvs=lambda k: rng.normal(size=k))
    J = sp.hstack([Jb, link], format="csc")
    J.sort_indices()
    return dict(H=H.tocsc(), J=J, s=np.concatenate(ss), y=np.concatenate(ys), mu=1e-2, n=n, m=m)


def hanging_chain(N_h=400, seed=0):
    """S-C2, stand-in for CUTEst CHAIN (BASELINE config 2; CUTEst is not in the repo): the hanging chain
        min  int_0^1 x sqrt(1 + u^2) dt   s.t.  x' = u,  int_0^1 sqrt(1 + u^2) dt = L,  x(0) = a, x(1) = b
    on N_h trapezoidal intervals, variables z = [x_0..x_N, u_0..u_N] (n = 2 N_h + 2).  Rows of a(z) >= 0 in the
    order the reference's adapter builds them (Class_cutest.jl:454-459): [cons >= l ; -cons <= u] for the N_h
    dynamics rows, the length row and the two end conditions (every equality is a row pair).  J is banded except
    for the two length rows (dense in u), H = Hessian of the Lagrangian at a seeded point: 2 x 2 blocks coupling
    x_i and u_i with a zero (x, x) diagonal -- indefinite, the delta loop has work to do.  Small fronts throughout:
    the HBM-bound assembly case."""
    rng = np.random.default_rng(seed + 2000)
    N = N_h
    h = 1.0 / N
    t = np.linspace(0.0, 1.0, N + 1)
    x = 1.0 + 3.0 * (t - 0.5) ** 2 + 0.01 * rng.normal(size=N + 1)
    u = 6.0 * (t - 0.5) + 0.01 * rng.normal(size=N + 1)
    n = 2 * N + 2
    ix, iu = np.arange(N + 1), N + 1 + np.arange(N + 1)
    wq = np.full(N + 1, h); wq[0] = wq[-1] = h / 2            # trapezoid weights
    r1 = np.sqrt(1.0 + u * u)
    # equality rows c(z) = 0: dynamics (N), length (1), ends (2)
    rows, cols, vals = [], [], []
    for i in range(N):
        rows += [i, i, i, i]; cols += [ix[i + 1], ix[i], iu[i], iu[i + 1]]; vals += [1.0, -1.0, -h / 2, -h / 2]
    rows += [N] * (N + 1); cols += list(iu); vals += list(wq * u / r1)
    rows += [N + 1, N + 2]; cols += [ix[0], ix[N]]; vals += [1.0, 1.0]
    C = sp.coo_matrix((vals, (rows, cols)), shape=(N + 3, n)).tocsc()
    J = sp.vstack([C, -C], format="csc")
    J.sort_indices()
    m = J.shape[0]
    # Lagrangian Hessian: objective sum w x sqrt(1 + u^2), minus lam * length constraint (lam seeded)
    lam = 0.3 + 0.1 * rng.random()
    duu = wq * (x - lam) / r1 ** 3
    dxu = wq * u / r1
    H = sp.coo_matrix((np.concatenate([duu, dxu]), (np.concatenate([iu, iu]), np.concatenate([iu, ix]))), shape=(n, n)).tocsc()
    H.sort_indices()
    s = np.exp(rng.uniform(np.log(1e-3), np.log(1e1), size=m))
    mu = 1e-2
    y = mu * np.exp(rng.uniform(np.log(1e-1), np.log(1e1), size=m)) / s
    return dict(H=H, J=J, s=s, y=y, mu=mu, n=n, m=m)


def infeasible_lp(rows=600, cols=900, per_col=5, frac_bounded=0.7, n_dependent=6, seed=0):
    """S-C4, stand-in for the Netlib-infeasible LPs (BASELINE config 4; the lpi_*.mat files are not in the repo):
    a random sparse LP  l_c <= A x <= u_c, bounds on a fraction of the variables, made infeasible by a contradictory
    row pair (the last row repeats the first with an incompatible range -- for the KKT matrix: two parallel rows).
    H = 0, and `n_dependent` free columns repeat other free columns, so J'SJ is singular: factor!(delta = 0) has the
    wrong inertia and ipopt_strategy! must shift."""
    rng = np.random.default_rng(seed + 4000)
    A = sp.random(rows, cols, density=per_col / rows, random_state=np.random.RandomState(seed + 11), format="lil",
                  data_rvs=lambda k: rng.normal(size=k))
    A = A.tocsc()
    empty = np.flatnonzero(np.diff(A.indptr) == 0)              # a real LP has no empty columns
    A = A.tolil()
    for c in empty:
        A[int(rng.integers(rows)), c] = 1.0
    bounded = rng.random(cols) < frac_bounded
    free = np.flatnonzero(~bounded)
    for q in range(min(n_dependent, len(free) // 2)):
        A[:, free[2 * q + 1]] = A[:, free[2 * q]]                # linearly dependent free columns
    A = A.tocsr()
    A = sp.vstack([A, A[0]], format="csr")                        # the contradictory twin of row 0
    lo_b = np.flatnonzero(bounded)
    up_b = lo_b[rng.random(len(lo_b)) < 0.5]
    I = sp.identity(cols, format="csr")
    J = sp.vstack([A, -A, I[lo_b], -I[up_b]], format="csc")       # [cons >= l ; -cons <= u ; I_l ; -I_u]
    J.sort_indices()
    n, m = cols, J.shape[0]
    s = np.exp(rng.uniform(np.log(1e-3), np.log(1e1), size=m))
    mu = 1e-2
    y = mu * np.exp(rng.uniform(np.log(1e-1), np.log(1e1), size=m)) / s
    return dict(H=sp.csc_matrix((n, n)), J=J, s=s, y=y, mu=mu, n=n, m=m)
