"""Device times of factor + solve on the configurations that are all small fronts or mostly launch latency (BASELINE config 2
stand-ins, S-C4, S-small), checked against a dense solve where the size allows.  Usage: python scripts/tiny_probe.py [reps]"""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
cfgs = [("chain400", lambda: synth.hanging_chain(N_h=400, seed=0)),
        ("chain20000", lambda: synth.hanging_chain(N_h=20000, seed=0)),
        ("S-C4", lambda: synth.infeasible_lp(seed=0)),
        ("S-small", lambda: synth.make_config("S-small", seed=1, h_per_col=2, j_per_row=3))]
for name, gen in cfgs:
    prob = gen()
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=float(prob.get("delta_ok", 1.0)))
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    rc = h.ls_factor_b(K, n, m)
    b = np.random.default_rng(0).normal(size=n + m)
    fm, sm = [], []
    for _ in range(reps):
        rc = h.ls_factor_b(K, n, m); x = h.ls_solve(b)
        st = h.stats(); fm.append(st["last_factor_ms"]); sm.append(st["last_solve_ms"])
    Ms = synth.symmetrize_lower(K)
    res = float(np.max(np.abs(Ms @ x - b)) / max(1.0, np.max(np.abs(x))))
    row = {"config": name, "n": n, "m": m, "rc": int(rc), "factor_ms": float(np.median(fm)), "solve_ms": float(np.median(sm)), "factor_min": float(np.min(fm)),
           "solve_min": float(np.min(sm)), "residual": res, "inertia": list(h.inertia)}
    row["fs_per_s"] = 1000.0 / (row["factor_ms"] + row["solve_ms"])
    finalize_b(h)
    print(json.dumps(row), flush=True)
