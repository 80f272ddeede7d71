"""Generates tests/golden/kkt_known_answers.json.

The reference (Julia + CHOLMOD) cannot run here and stores no golden vectors, so the fixtures are
KNOWN ANSWERS derived independently of both the oracle and the product: dense numpy/LAPACK solves
of the full 3-block Newton system (layout /root/reference/src/kkt_system_solver/system_rhs.jl:34-43),
eigenvalue inertias, and a dense re-enactment of the delta loop.  Problems:
  * the two 10 x 10 matrices of test_linear_solvers (reference test/linear_system_solvers.jl:94-116),
    with a seeded b instead of the reference's unseeded rand(10);
  * the README toy (min x s.t. x^2 >= 1, x >= -1) at the point of SURVEY.md Appendix B;
  * toy_lp0 ... toy_lp8 (reference test/problems.jl:108-287) written in the a(x) >= 0 form the
    reference's adapter builds (Class_cutest.jl:385-420,451-503: rows [cons>=l; -cons<=u; I_l; -I_u]),
    at hand-chosen interior points, delta = 1e-8 and the affine rhs exactly like
    test_kkt_solver (test/kkt_system_solvers.jl:61-89);
  * a 5-variable indefinite problem whose delta-loop trace (delta values, #fac) is enacted densely
    following delta_strategy.jl:37-114.
Run:  python tests/golden/make_golden.py
"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def sym_from_lower(Hl):
    return Hl + np.tril(Hl, -1).T


def newton_dense(Hl, J, s, y, delta, rD, rP, rC):
    """Solve [(H+dI) dx - J'dy = rD ; J dx - ds = rP ; S dy + Y ds = rC] densely."""
    n, m = Hl.shape[0], J.shape[0]
    H = sym_from_lower(Hl) + delta * np.eye(n)
    A = np.zeros((n + 2 * m, n + 2 * m))
    A[:n, :n] = H
    A[:n, n + m:] = -J.T
    A[n:n + m, :n] = J
    A[n:n + m, n:n + m] = -np.eye(m)
    A[n + m:, n:n + m] = np.diag(y)
    A[n + m:, n + m:] = np.diag(s)
    b = np.concatenate([rD, rP, rC])
    # solve in extended precision by refinement
    sol = np.linalg.solve(A, b)
    for _ in range(3):
        r = (b.astype(np.longdouble) - A.astype(np.longdouble) @ sol.astype(np.longdouble)).astype(np.float64)
        sol = sol + np.linalg.solve(A, r)
    return sol[:n], sol[n + m:], sol[n:n + m]  # dx, dy, ds


def inertia_dense(M, tol=1e-12):
    w = np.linalg.eigvalsh(M)
    return int((w > tol).sum()), int((w < -tol).sum()), int((np.abs(w) <= tol).sum())


def affine_rhs(grad, J, cons, s, y, mu, a_norm_penalty):
    # system_rhs.jl:57-73 with Reduct_affine (P = D = mu = 0); eval.jl:59-63,136-142
    grad_lag = grad - J.T @ y + (mu * 0.0) * a_norm_penalty * (J.T @ np.ones(len(s)))
    rD = -grad_lag
    rP = -(cons - s)
    rC = 0.0 * mu - s * y
    return rD, rP, rC


def lp(c, rows_l, rows_u, lb, ub):
    """rows_l: list of (coeffs, l) for c'x >= l; rows_u: (coeffs, u) for c'x <= u."""
    n = len(c)
    Jr, off = [], []
    for a, l in rows_l:
        Jr.append(np.array(a, float)); off.append(-l)
    for a, u in rows_u:
        Jr.append(-np.array(a, float)); off.append(u)
    for i in range(n):
        if lb[i] is not None:
            e = np.zeros(n); e[i] = 1.0; Jr.append(e); off.append(-lb[i])
    for i in range(n):
        if ub[i] is not None:
            e = np.zeros(n); e[i] = -1.0; Jr.append(e); off.append(ub[i])
    return np.array(c, float), np.array(Jr), np.array(off)


def toy_lps():
    N = None
    P = {}
    P["toy_lp0"] = lp([1.0], [([1.0], 4.0)], [], [N], [N])
    P["toy_lp1"] = lp([-1.0, -100.0], [], [([1.0, 1.0], 1.0)], [0.0, 0.0], [N, N])
    P["toy_lp2"] = lp([-1.0, -100.0], [], [([1.0, 1.0], 2.0)], [0.0, 0.0], [1.0, 1.0])
    P["toy_lp3"] = lp([1.0, 0.0], [([1.0, 1.0], 1.0)], [([1.0, 1.0], 2.0)], [0.0, 0.0], [1.0, 1.0])
    P["toy_lp4"] = P["toy_lp3"]
    P["toy_lp5"] = lp([1.0, 0.0], [([1.0, 1.0], 1.0), ([32.5, 32.5], 32.5)],
                      [([1.0, 1.0], 1.0), ([32.5, 32.5], 32.5), ([3.0, 3.0], 3.0)], [0.0, 0.0], [1.0, 1.0])
    P["toy_lp6"] = lp([1.0, 0.0], [([1.0, 1.0], 1.0), ([5.5, 5.5], 5.5)],
                      [([1.0, 1.0], 1.0), ([5.5, 5.5], 5.5)], [0.0, 0.0], [1.0, 1.0])
    P["toy_lp7"] = lp([1.0, 0.0], [([2.0, 1.0], 1.0)], [([2.0, 1.0], 1.0)], [0.0, 0.0], [1.0, 1.0])
    P["toy_lp8"] = lp([1.0, 0.0], [([1.0, 1.0], 1.0)], [([5.5, 5.5], 5.5)], [0.0, 0.0], [1.0, 1.0])
    return P


def problem_record(name, Hl, J, grad, cons, x, s, y, mu, delta, a_norm_penalty=1e-4):
    n, m = len(x), len(s)
    rD, rP, rC = affine_rhs(grad, J, cons, s, y, mu, a_norm_penalty)
    dx, dy, ds = newton_dense(Hl, J, s, y, delta, rD, rP, rC)
    H = sym_from_lower(Hl)
    K = np.block([[H + delta * np.eye(n), J.T], [J, -np.diag(s / y)]])
    Q = H + J.T @ np.diag(y / s) @ J + delta * np.eye(n)
    return dict(name=name, n=n, m=m, H_lower=Hl.tolist(), J=J.tolist(), grad=grad.tolist(), cons=cons.tolist(),
                x=x.tolist(), s=s.tolist(), y=y.tolist(), mu=mu, delta=delta, a_norm_penalty=a_norm_penalty,
                rD=rD.tolist(), rP=rP.tolist(), rC=rC.tolist(), dx=dx.tolist(), dy=dy.tolist(), ds=ds.tolist(),
                inertia_K=inertia_dense(K), inertia_Q=inertia_dense(Q))


def jl_max(a, b):
    return math.nan if (a != a or b != b) else max(a, b)


def jl_min(a, b):
    return math.nan if (a != a or b != b) else min(a, b)


def line_search_record(rec, frac_bd_predict=0.2, frac_bd=0.1, ex=0.5, comp_feas=0.01, scale_D=0.7, scale_mu=1.3):
    """Known answers for the step-side functions (frac_boundary.jl:3-35, move.jl:15-17,28-118, line_search.jl:84-86,
    eval.jl:236-273), written out entry by entry with Python floats from the record's dense direction; affine
    direction, so dir.mu = -mu.  The candidate is the half step to the boundary."""
    n, m = rec["n"], rec["m"]
    Hl, J = np.array(rec["H_lower"], float).reshape(n, n), np.array(rec["J"], float).reshape(m, n)
    s, y, mu, pen = rec["s"], rec["y"], rec["mu"], rec["a_norm_penalty"]
    dx, dy, ds, grad = rec["dx"], rec["dy"], rec["ds"], rec["grad"]
    dmu = -mu
    nx = max(abs(v) for v in dx)
    thr = nx * nx ** ex
    ratio = 1.0
    for i in range(m):
        ratio = jl_max(ratio, -ds[i] / (s[i] - frac_bd_predict * min(s[i], thr)))
    step_P = 1.0 / ratio
    alpha = 0.5 * step_P
    s_c = [s[i] + alpha * ds[i] for i in range(m)]
    mu_c = mu + alpha * dmu
    s_ok = all(s_c[i] >= frac_bd * min(s[i], thr) for i in range(m))
    s_bad = list(s_c)
    if m:
        s_bad[m - 1] = 0.5 * frac_bd * min(s[m - 1], thr)
    lb, ub = 0.0, 1.0
    for i in range(m):
        if dy[i] == 0.0:
            continue   # no entry of these fixtures has dy == 0 (asserted below)
        ub_i = mu_c / (comp_feas * s_c[i] * dy[i]) - y[i] / dy[i]
        lb_i = mu_c * comp_feas / (s_c[i] * dy[i]) - y[i] / dy[i]
        if dy[i] > 0.0:
            lb, ub = jl_max(lb_i * 1.001 + 0.0, lb), jl_min(ub_i / 1.001 - 0.0, ub)
        else:
            lb, ub = jl_max(ub_i * 1.001 + 0.0, lb), jl_min(lb_i / 1.001 - 0.0, ub)
    assert all(v != 0.0 for v in dy)
    ratio_y = 1.0
    for i in range(m):
        ratio_y = jl_max(ratio_y, -dy[i] / (y[i] - frac_bd * y[i] * min(1.0, nx)))
    ub = jl_min(ub, 1.0 / ratio_y)
    # predicted reductions at step 1 (long double accumulation: the expected values are the exact sums, rounded once)
    ld = np.longdouble
    H = sym_from_lower(Hl).astype(ld)
    Jl, dxl = J.astype(ld), np.array(dx, ld)
    v = Jl @ dxl
    J_gain = sum(v[i] * v[i] * (ld(y[i]) / ld(s[i])) for i in range(m))
    w = np.array([ld(mu) / ld(s[i]) - ld(mu) * ld(pen) for i in range(m)], ld)
    gphi = np.array(grad, ld) - (Jl.T @ w if m else 0)
    step = 1.0
    phi_red = step * float(dxl @ gphi) + step * step * 0.5 * float(dxl @ (H @ dxl) + J_gain)
    C_k = max([abs(s[i] * y[i] - mu) for i in range(m)], default=0.0)
    P_k = max([abs(s[i] * y[i] + dy[i] * s[i] * step + ds[i] * y[i] * step - (mu + dmu * step)) for i in range(m)], default=0.0)
    merit = phi_red + ((P_k ** 3 - C_k ** 3) / mu ** 2 if m else 0.0)
    # move_dual's least-squares step on the candidate (same J and grad: the toys are linear or evaluated at iter)
    dyl = np.array(dy, ld)
    q = np.concatenate([ld(scale_D) * (Jl.T @ dyl), np.array([ld(scale_mu) * ld(s_c[i]) * dyl[i] for i in range(m)], ld)])
    dres = np.array(grad, ld) - Jl.T @ (np.array(y, ld) - ld(mu_c) * ld(pen))
    res = np.concatenate([ld(scale_D) * dres, np.array([-ld(scale_mu) * (ld(s_c[i]) * ld(y[i]) - ld(mu_c)) for i in range(m)], ld)])
    sd = float((res @ q) / (q @ q))
    small = max(lb, min(ub, alpha))
    step_D = jl_max(jl_min(sd, ub), small)
    return dict(name=rec["name"], frac_bd_predict=frac_bd_predict, frac_bd=frac_bd, ex=ex, comp_feas=comp_feas, scale_D=scale_D,
                scale_mu=scale_mu, dir_mu=dmu, dx_norm_inf=nx, step_size_P=step_P, alpha=alpha, s_cand=s_c, mu_cand=mu_c,
                s_bound_ok=bool(s_ok), s_bad=s_bad, dual_lb=lb, dual_ub=ub, phi_red=phi_red, C_k=C_k, P_k=P_k, merit_red=merit,
                ls_step_unclamped=sd, step_size_D=step_D)


def delta_loop_dense(Hl, J, s, y, delta_prev, kind):
    """delta_strategy.jl:37-114 with dense inertia; kind = 'schur' or 'symmetric'."""
    n, m = Hl.shape[0], J.shape[0]
    H = sym_from_lower(Hl)
    Q0 = H + J.T @ np.diag(y / s) @ J
    schur_diag = np.diag(Q0).copy()

    def ok(delta):
        if kind == "schur":
            w = np.linalg.eigvalsh(Q0 + delta * np.eye(n))
            return bool((w > 0).all())
        K = np.block([[H + delta * np.eye(n), J.T], [J, -np.diag(s / y)]])
        w = np.linalg.eigvalsh(K)
        return int((w > 0).sum()) == n and int((w < 0).sum()) == m

    tried, num_fac = [], 0
    tau = 1.5 * schur_diag.min()
    delta = 0.0
    if tau > 0.0:
        tau = 0.0
        tried.append(delta); num_fac += 1
        if ok(delta):
            return dict(status="success", num_fac=num_fac, delta=delta, tried=tried)
    for i in range(1, 501):
        if i == 1:
            delta = max(1e-12 - tau, delta_prev / math.pi) if delta_prev != 0.0 else 1e-6 - tau
        else:
            delta = delta * 8.0
        tried.append(delta); num_fac += 1
        if ok(delta):
            return dict(status="success", num_fac=num_fac, delta=delta, tried=tried)
        if delta > 1e50:
            return dict(status="failure", num_fac=num_fac, delta=delta, tried=tried)
    raise RuntimeError("max it")


def main():
    out = {}
    rng = np.random.default_rng(20240917)
    # --- test_linear_solvers matrices
    A1 = np.eye(10)
    A2 = np.eye(10); A2[9, 0] = 0.1; A2[8, 1] = 0.1   # A[10,1] = A[9,2] = 0.1 (1-based)
    lin = []
    for A in (A1, A2):
        b = rng.random(10)
        Afull = sym_from_lower(np.tril(A))
        lin.append(dict(A_lower=np.tril(A).tolist(), b=b.tolist(), x=np.linalg.solve(Afull, b).tolist(), n=10, m=0, inertia=1))
    out["linear_solvers"] = lin
    # --- README toy at the SURVEY Appendix-B point
    x = np.array([-0.1]); s = np.array([8.0, 1.0]); y = np.array([2.0, 0.1]); mu = 1.0; pen = 1e-4
    J = np.array([[2 * x[0]], [1.0]])
    cons = np.array([x[0] ** 2 - 1.0, x[0] + 1.0])
    grad = np.array([1.0])
    Hl = np.array([[-2.0 * (y[0] + mu * pen)]])
    tau = 1.5 * (Hl[0, 0] + (J[:, 0] ** 2 * (y / s)).sum())
    delta1 = 1e-6 - tau
    rec = problem_record("readme_toy", Hl, J, grad, cons, x, s, y, mu, delta1, pen)
    rec["delta_loop_schur"] = delta_loop_dense(Hl, J, s, y, 0.0, "schur")
    rec["delta_loop_symmetric"] = delta_loop_dense(Hl, J, s, y, 0.0, "symmetric")
    out["readme_toy"] = rec
    # --- toy LPs
    pts = {1: np.array([4.7]), 2: np.array([0.3, 0.55])}
    toys = []
    for name, (c, Jm, off) in toy_lps().items():
        n = len(c); m = Jm.shape[0]
        xx = pts[n]
        cons = Jm @ xx + off
        s = 0.2 + rng.random(m) * 1.5
        y = 0.1 + rng.random(m) * 2.0
        toys.append(problem_record(name, np.zeros((n, n)), Jm, c, cons, xx, s, y, 0.37, 1e-8))
    out["toy_lps"] = toys
    # --- 5-variable indefinite problem, delta loop
    n, m = 5, 4
    B = rng.normal(size=(n, n))
    Hl = np.tril(B + B.T) - 2.5 * np.eye(n)
    J = rng.normal(size=(m, n)) * (rng.random((m, n)) < 0.6)
    s = 0.5 + rng.random(m); y = 0.05 + 0.2 * rng.random(m)
    xx = rng.normal(size=n); cons = J @ xx - 0.3; grad = rng.normal(size=n)
    loops = {}
    for kind in ("schur", "symmetric"):
        for dprev in (0.0, 0.5):
            loops[f"{kind}_prev{dprev}"] = delta_loop_dense(Hl, J, s, y, dprev, kind)
    dl = loops["symmetric_prev0.0"]["delta"]
    rec = problem_record("indef5", Hl, J, grad, cons, xx, s, y, 0.1, dl)
    rec["delta_loops"] = loops
    out["indef5"] = rec
    # --- positive diagonal but indefinite: tau > 0, so delta = 0 is tried first, then 1e-6 * 8^k
    n, m = 5, 3
    Hl = np.tril(np.full((n, n), 2.0), -1) + np.eye(n)
    J = np.array([[1.0, 0, 0, 0, 0], [0, 0, 1.0, 0, -1.0], [0, 1.0, 0, 0.5, 0]])
    s = np.array([1.0, 2.0, 0.5]); y = np.array([0.02, 0.01, 0.03])
    xx = np.linspace(-1, 1, n); cons = J @ xx + 0.2; grad = np.ones(n)
    loops = {}
    for kind in ("schur", "symmetric"):
        for dprev in (0.0, 3.0):
            loops[f"{kind}_prev{dprev}"] = delta_loop_dense(Hl, J, s, y, dprev, kind)
    rec = problem_record("posdiag_indef5", Hl, J, grad, cons, xx, s, y, 0.1, loops["symmetric_prev0.0"]["delta"])
    rec["delta_loops"] = loops
    out["posdiag_indef5"] = rec
    # --- Clever_Symmetric index work: the data and the integer answers of the reference's own unit tests
    # (test/kkt_system_solvers.jl:5-41 test_compute_indicies, :49-58 test_compare_columns), 1-based as written there
    out["compute_indicies"] = {
        "J": [[1.0, 1.0, 1.0], [-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 4.0, 3.0], [0.0, 2.0, 1.5], [3.0, 2.0, 1.0], [-2.0, -2.0, -2.0]],
        "sorted_cols": [4, 5, 3, 6, 1, 2, 7],
        "break_points": [1, 3, 4, 5],
        "no_para_indicies": [1, 3, 4, 6],
        "group_members": {"1": [[2, 2, -1.0], [3, 7, -2.0]], "3": [[2, 5, 0.5]]},   # group -> [position in ls, ind, ratio]
        "singletons": [2, 4],                                                       # groups with u == u[first], g == 1
        "compare_columns_A_rows": [[0.0, 10.0, 0.0], [1.0, 0.0, 1.0], [1.0, 0.0, 0.0], [2.0, 0.0, 0.0]],   # A = sparse(rows')'
        "compare_columns": [[1, 2, True], [2, 1, False], [2, 3, False], [3, 2, True], [3, 1, False], [1, 3, True]],
    }
    # --- step-side functions (line search), known answers on the records above
    out["line_search"] = [line_search_record(r) for r in [out["readme_toy"], out["indef5"], out["posdiag_indef5"]] + out["toy_lps"]]
    # a tight complementarity window (comp_feas = 0.5, 0.9): positive lower bounds and empty ranges (lb >= ub)
    for cf in (0.5, 0.9):
        out["line_search"] += [line_search_record(r, frac_bd_predict=0.05, frac_bd=0.3, ex=1.0, comp_feas=cf, scale_D=1.0, scale_mu=1.0)
                               for r in (out["indef5"], out["toy_lps"][3], out["toy_lps"][8])]
    with open(os.path.join(HERE, "kkt_known_answers.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "kkt_known_answers.json"))


if __name__ == "__main__":
    main()
