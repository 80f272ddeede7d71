"""Soak of the round-2 paths: batched solves, the KKT layer (schur / symmetric) with delta loops, and the metric workload --
results must repeat bit for bit, device memory must not grow."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import ctypes as C
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
for name, cycles in (("S-C3", 100), ("S-metric", 25)):
    prob = synth.make_config(name, seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    B = np.random.default_rng(0).normal(size=(4, n + m))
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    h.analyze(K)
    d_vals = h.dev_upload(K.data); d_rhs = h.dev_upload(B); d_sol = h.dev_alloc(8 * 4 * (n + m))
    assert h.ls_factor_dev(d_vals, n, m) == 1
    h.ls_solve_dev(d_rhs, d_sol, 4); X0 = h.dev_download(d_sol, (4, n + m)).copy()
    x1 = h.ls_solve(B[1])
    assert np.max(np.abs(x1 - X0[1])) <= 1e-12 * np.max(np.abs(x1))
    f0 = free_mb(); ts = []
    for i in range(cycles):
        t = time.perf_counter()
        assert h.ls_factor_dev(d_vals, n, m) == 1
        h.ls_solve_dev(d_rhs, d_sol, 4 if i % 2 else 1)
        ts.append(time.perf_counter() - t)
        X = h.dev_download(d_sol, (4, n + m))
        assert np.array_equal(X[0], X0[0]) and (i % 2 == 0 or np.array_equal(X, X0)), (name, i)
    print(f"{name}: {cycles} factor + batched-solve cycles bit-identical; first 5 avg {1e3*np.mean(ts[:5]):.2f} ms, last 5 avg {1e3*np.mean(ts[-5:]):.2f} ms, free memory change {free_mb() - f0:.1f} MiB", flush=True)
    h.dev_free(d_vals); h.dev_free(d_rhs); h.dev_free(d_sol); finalize_b(h)
# KKT layer: nonconvex instance, delta loop + direction, repeated
sys.path.insert(0, "tests")
from onephase_jl_amd import kkt_system_solver as KS
prob = synth.make_config("S-C3", seed=0, convex=False)
rng = np.random.default_rng(5)
n, m = prob["n"], prob["m"]
def make_it():
    return KS.Class_iterate(x=rng.normal(size=n), y=prob["y"].copy(), s=prob["s"].copy(), mu=prob["mu"], J=prob["J"], H=prob["H"],
                            grad=np.random.default_rng(6).normal(size=n), cons=prob["s"] + 0.1 * np.random.default_rng(7).normal(size=m), a_norm_penalty_par=1e-4)
for kind in ("symmetric", "schur"):
    pars = KS.Class_parameters(); pars.kkt.kkt_solver_type = kind
    it = make_it()
    k = KS.pick_KKT_solver(pars); k.initialize_b(it); f0 = None; ref = None
    for i in range(20):
        it.delta = 0.0
        k.form_system_b(it); st = k.ipopt_strategy_b(it); k.kkt_associate_rhs_b(it, KS.Reduct_affine()); k.compute_direction_b()
        d = k.dir.x.copy()
        if ref is None: ref, f0 = (st, d), free_mb()
        assert st == ref[0] and np.array_equal(d, ref[1]), (kind, i)
    print(f"KKT {kind}: 20 x (form, delta loop {ref[0]}, direction) bit-identical, free memory change {free_mb() - f0:.1f} MiB", flush=True)
    k.finalize_b()
