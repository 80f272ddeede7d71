"""solve of one dense front with the flow sweeps against the two-launch sweeps: where do the solutions differ?
usage: python scripts/flow_debug.py n  (runs itself twice, OKKT_SOLVE_FLOW=0 / 1)"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2600
if len(sys.argv) > 2:
    import scipy.sparse as sp
    from onephase_jl_amd.linear_system_solvers import initialize_b, linear_solver_HIP
    rng = np.random.default_rng(3)
    B = rng.normal(size=(n, n)); M = B + B.T + np.diag(np.full(n, 3.0 * np.sqrt(n)))
    h = linear_solver_HIP("symmetric"); initialize_b(h)
    rc = h.ls_factor_b(sp.csc_matrix(np.tril(M)), n, 0)
    b = rng.normal(size=n)
    x = np.array(h.ls_solve(b))
    np.save(sys.argv[2], np.stack([x, M @ x - b]))
    sys.exit(0)
outs = []
for flow in ("0", "1"):
    e = dict(os.environ); e["OKKT_SOLVE_FLOW"] = flow
    if len(sys.argv) > 1 and os.environ.get("PHASE"): e["OKKT_DEBUG_SOLVE_PHASE"] = os.environ["PHASE"]
    out = f"/tmp/flowdbg_{flow}.npy"
    r = subprocess.run([sys.executable, __file__, str(n), out], env=e, capture_output=True, text=True)
    if r.returncode != 0: print(r.stdout[-500:], r.stderr[-1500:]); sys.exit(1)
    outs.append(np.load(out))
a, b = outs
print("resid legacy", np.abs(a[1]).max(), "flow", np.abs(b[1]).max())
d = np.abs(a[0] - b[0])
bad = np.where(d > 1e-9 * np.abs(a[0]).max())[0]
print("differing entries", len(bad), "first", bad[:10], "last", bad[-10:] if len(bad) else [])
