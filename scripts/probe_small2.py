"""Small systems (the reference's own test sizes): device times and wall-clock per call beside the CPU port on one thread."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import oracle
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP
cases = [("S-tiny", synth.make_config("S-tiny", seed=0), 1e-8), ("S-small", synth.make_config("S-small", seed=0), 1e-8),
         ("n=2000 m=3000", synth.make_problem(seed=0, n=2000, m=3000, j_per_row=8, h_per_col=4), 1e-8),
         ("n=4000 m=6000", synth.make_problem(seed=0, n=4000, m=6000, j_per_row=10, h_per_col=5), 1e-8),
         ("chain N_h=400", synth.hanging_chain(400), 1.0), ("lp 600x900", synth.infeasible_lp(), 1.0)]
for name, prob, delta in cases:
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=delta)
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    h.ls_factor_b(K, n, m)
    b = np.random.default_rng(0).normal(size=n + m)
    wf, ws, fm, sm = [], [], [], []
    for _ in range(10):
        t = time.perf_counter(); rc = h.ls_factor_b(K, n, m); wf.append(time.perf_counter() - t)
        t = time.perf_counter(); x = h.ls_solve(b); ws.append(time.perf_counter() - t)
        st = h.stats(); fm.append(st["last_factor_ms"]); sm.append(st["last_solve_ms"])
    o = oracle.linear_solver_ORACLE_MF("symmetric", perm=h.perm(), nthreads=1); o._analyze(K)
    cf, cs = [], []
    for _ in range(5):
        t = time.perf_counter(); o.ls_factor_b(K, n, m); cf.append(time.perf_counter() - t)
        t = time.perf_counter(); o.ls_solve(b); cs.append(time.perf_counter() - t)
    print(f"{name:16s} n+m={n+m:6d} nnzL={st['nnzL']:8d} levels={st['nlevels']:3d} maxfront={st['max_front']:5d} | GPU factor {1e3*np.median(fm):6.0f} us dev / {1e6*np.median(wf):6.0f} us wall, "
          f"solve {1e3*np.median(sm):5.0f} us dev / {1e6*np.median(ws):5.0f} us wall | CPU 1 thread factor {1e6*np.median(cf):6.0f} us solve {1e6*np.median(cs):5.0f} us", flush=True)
    finalize_b(h)
