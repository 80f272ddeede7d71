"""KKT-level randomised check through the C ABI against oracle/kkt_oracle.py: random (H, J, s, y, x, grad, cons) of random shape
(empty rows and columns of J, dense rows, zero H, a single variable, more constraints than variables ...), every solver kind, the
assembled matrix, schur_diag, the right-hand side, the direction (dx, dy, ds) and N-err; a batch of three directions from one factor.
Usage: python scripts/fuzz_kkt.py [cases] [seed]."""
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
from onephase_jl_amd import kkt_system_solver as KS
from oracle import kkt_oracle as KO

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = 0
compared = 0
worst = 0.0
kinds = {}
for case in range(cases):
    rng = np.random.default_rng(7000 * seed0 + case)
    shape = ["random", "tall", "wide", "dense_rows", "empty_rows", "diag_H", "zero_H", "tiny"][case % 8]
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    n = int(rng.integers(1, int(300 * scale))); m = int(rng.integers(1, int(400 * scale)))
    if shape == "tall": m = int(rng.integers(n, 4 * n + 2))
    if shape == "wide": m = int(rng.integers(1, max(2, n // 3)))
    if shape == "tiny": n, m = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    J = sp.random(m, n, density=min(1.0, rng.uniform(1, 5) / n), random_state=rng, format="lil")
    if shape == "dense_rows":
        for i in range(min(m, 3)): J[i, :] = rng.normal(size=n)
    if shape == "empty_rows":
        for i in range(0, m, 3): J[i, :] = 0.0
    J = sp.csc_matrix(J); J.eliminate_zeros()
    if shape == "zero_H": Hs = sp.csc_matrix((n, n))
    elif shape == "diag_H": Hs = sp.diags(rng.uniform(0.1, 2.0, size=n)).tocsc()
    else:
        R = sp.random(n, n, density=min(1.0, rng.uniform(0, 3) / n), random_state=rng)
        Hs = (R + R.T + sp.diags(np.asarray(abs(R + R.T).sum(axis=1)).ravel() + rng.uniform(0.1, 1.0, size=n))).tocsc()
    H = sp.csc_matrix(sp.tril(Hs))
    s = 10.0 ** rng.uniform(-2, 1, size=m); y = 10.0 ** rng.uniform(-2, 1, size=m)
    x = rng.normal(size=n); grad = rng.normal(size=n); cons = s + 0.1 * rng.normal(size=m); mu = float(10.0 ** rng.uniform(-3, 0))
    kind = ["schur", "symmetric", "schur_direct", "clever_symmetric"][int(rng.integers(0, 4))]
    delta = float(10.0 ** rng.uniform(-6, -1))
    tag = f"case {case} {shape} n={n} m={m} {kind} delta={delta:.2e}"
    mk = lambda It: It(x=x.copy(), y=y.copy(), s=s.copy(), mu=mu, J=J.copy(), H=H.copy(), grad=grad.copy(), cons=cons.copy(), a_norm_penalty_par=1e-4)
    try:
        pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = kind
        k = KS.pick_KKT_solver(pars); it = mk(KS.Class_iterate)
        k.initialize_b(it); k.form_system_b(it)
        inertia = k.factor_b(delta)
        k.kkt_associate_rhs_b(it, KS.Reduct_affine()); k.compute_direction_b()
        ko = KO.pick_KKT_solver(kind, perm=k.linear_solver_perm()); ito = mk(KO.Iterate)
        ko.initialize_b(ito); ko.form_system_b(ito)
        io = ko.factor_b(delta)
        ko.kkt_associate_rhs_b(ito, KO.Reduct_affine()); ko.compute_direction_b()
        ok = inertia == io
        errs = {}
        if inertia == 1 and io == 1:
            for a in ("x", "y", "s"):
                da, db = getattr(k.dir, a), getattr(ko.dir, a)
                errs[a] = float(np.max(np.abs(da - db)) / max(1.0, np.max(np.abs(db)))) if len(db) else 0.0
            cond_guard = 1e-7
            compared += 1; worst = max(worst, max(errs.values())); kinds[kind] = kinds.get(kind, 0) + 1
            ok = ok and all(e <= cond_guard for e in errs.values())
            ok = ok and np.allclose(k.schur_diag, ko.schur_diag, rtol=1e-12, atol=1e-14)
            ok = ok and np.allclose(k.rhs.dual_r, ko.rhs.dual_r, rtol=1e-11, atol=1e-13)
            ok = ok and abs(k.kkt_err_norm.ratio - ko.kkt_err_norm.ratio) <= 1e-6 + 1e-3 * ko.kkt_err_norm.ratio
        if not ok:
            fails += 1
            print("FAIL", tag, "inertia", inertia, io, "errs", errs, "N_err", k.kkt_err_norm.ratio, ko.kkt_err_norm.ratio, flush=True)
        k.finalize_b()
    except Exception as ex:   # noqa
        fails += 1
        print("EXC", tag, repr(ex)[:300], flush=True)
print("FUZZ_KKT cases", cases, "failures", fails, "directions compared", compared, kinds, "worst relative difference %.2e" % worst)
