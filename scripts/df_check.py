"""GPU box: the dataflow launch (OKKT_DATAFLOW=1, csrc/dataflow.hip) against the per-step launches (OKKT_DATAFLOW=0) --
D and the stored factor must agree BIT FOR BIT (same bodies, same order of the panels per tile), then timings.
usage: python scripts/df_check.py [case ...]      cases: dense129 dense700 dense2600 S-C3 S-C5 S-metric (default: all but S-metric)
       python scripts/df_check.py --run <case> <out.npz>     (one process per setting: the switches are read once)"""
import os
import subprocess
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")


def run_case(name, out):
    from onephase_jl_amd import synth
    from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

    rng = np.random.default_rng(11)
    if name.startswith("dense"):
        n = int(name[5:])
        B = rng.normal(size=(n, n))
        M = B + B.T + np.diag(np.where(rng.random(n) < 0.4, 1.0, -1.0) * (3.0 * np.sqrt(n)))
        w = np.linalg.eigvalsh(M)
        K = sp.csc_matrix(np.tril(M))
        npos, nneg = int((w > 0).sum()), int((w < 0).sum())
    else:
        prob = synth.make_config(name, seed=0)
        K = synth.augmented_matrix(prob, delta=1e-8)
        npos, nneg = prob["n"], prob["m"]
    h = linear_solver_HIP("symmetric")
    initialize_b(h)
    rc = h.ls_factor_b(K, npos, nneg)
    tf, ts = [], []
    b = rng.normal(size=K.shape[0])
    for _ in range(4):
        rc = h.ls_factor_b(K, npos, nneg)
        x = h.ls_solve(b)
        st = h.stats()
        tf.append(st["last_factor_ms"])
        ts.append(st["last_solve_ms"])
    Ms = (sp.tril(K) + sp.tril(K, -1).T).tocsc()
    res = float(np.max(np.abs(Ms @ x - b)) / np.max(np.abs(b)))
    d = h.diag().copy()
    save = {"d": d, "x": x, "rc": rc, "inertia": np.array(h.inertia), "tf": np.array(tf), "ts": np.array(ts), "res": res}
    if K.shape[0] <= 40000:
        L = h.factor_csc()
        save["Ldata"] = L.data
        save["Lidx"] = L.indices
    np.savez(out, **save)
    finalize_b(h)


def main():
    cases = sys.argv[1:] or ["dense129", "dense300", "dense700", "dense2600", "S-C3", "S-C5"]
    os.makedirs("gpurun_out", exist_ok=True)
    bad = 0
    for c in cases:
        outs = {}
        for df in ("0", "1"):
            env = dict(os.environ)
            env["OKKT_DATAFLOW"] = df
            out = f"/tmp/df_{c}_{df}.npz"
            t = time.time()
            r = subprocess.run([sys.executable, "scripts/df_check.py", "--run", c, out], env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(f"{c} dataflow={df}: FAILED rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}")
                bad += 1
                continue
            outs[df] = dict(np.load(out))
            o = outs[df]
            print(f"{c} dataflow={df}: rc {int(o['rc'])} inertia {o['inertia'].tolist()} factor ms {np.round(o['tf'], 3).tolist()} solve ms {np.round(o['ts'], 3).tolist()} resid {float(o['res']):.2e}  ({time.time() - t:.0f} s)", flush=True)
        if len(outs) == 2:
            a, b = outs["0"], outs["1"]
            same_d = np.array_equal(a["d"], b["d"])
            same_x = np.array_equal(a["x"], b["x"])
            same_l = np.array_equal(a["Ldata"], b["Ldata"]) if "Ldata" in a and "Ldata" in b else None
            dd = np.max(np.abs(a["d"] - b["d"]) / np.maximum(np.abs(a["d"]), 1e-300))
            print(f"{c}: D bitwise {same_d} (max rel diff {dd:.2e}), L bitwise {same_l}, x bitwise {same_x}, inertia equal {a['inertia'].tolist() == b['inertia'].tolist()}", flush=True)
            if not (same_d and same_l is not False and a["inertia"].tolist() == b["inertia"].tolist()):
                bad += 1
    print("DF_CHECK", "FAILED" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--run":
        run_case(sys.argv[2], sys.argv[3])
    else:
        sys.exit(main())
