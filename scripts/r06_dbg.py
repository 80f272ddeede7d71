"""round 6 debugging: where does a variant's factor differ from the per-step launches?  usage: r06_dbg.py <case> VAR=VAL ..."""
import os, subprocess, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
case = sys.argv[1]
env = dict(a.split("=") for a in sys.argv[2:])
def run(e, tag):
    out = f"/tmp/dbg_{case}_{tag}.npz"
    ee = dict(os.environ); ee.update(e)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dataflow_case.py"), case, out], cwd=ROOT, env=ee, capture_output=True, text=True, timeout=600)
    if "CASE_OK" not in r.stdout:
        print("FAILED", e, r.stdout[-300:], r.stderr[-800:]); sys.exit(1)
    return dict(np.load(out))
a = run({"OKKT_DATAFLOW": "0"}, "steps")
e = {"OKKT_DATAFLOW": "1"}; e.update(env)
b = run(e, "var")
n = len(a["d"])
dd = np.flatnonzero(a["d"] != b["d"])
print(case, env, "n", n, "d differs at", len(dd), "positions", dd[:10], "max rel", np.max(np.abs(a["d"] - b["d"]) / np.abs(a["d"])) if len(dd) else 0)
if np.array_equal(a["Lidx"], b["Lidx"]):
    df = np.flatnonzero(a["Ldata"] != b["Ldata"])
    print("L entries differing", len(df), "of", len(a["Ldata"]))
    if len(df) and case.startswith("dense"):
        # dense front: column-major strictly lower
        rows = a["Lidx"][df]
        # column of each entry: entries are column by column, n-1-j entries in column j
        ptr = np.concatenate([[0], np.cumsum(np.arange(n - 1, -1, -1))])
        cols = np.searchsorted(ptr, df, side="right") - 1
        rel = np.abs(a["Ldata"][df] - b["Ldata"][df]) / np.maximum(np.abs(a["Ldata"][df]), 1e-300)
        print("rows", rows.min(), rows.max(), "cols", cols.min(), cols.max(), "max rel diff", rel.max())
        H = np.zeros(((n + 127) // 128, (n + 127) // 128), dtype=int)
        np.add.at(H, (rows // 128, cols // 128), 1)
        print("differing entries per 128 x 128 tile:\n", H)
        H32 = np.zeros((4, 4), dtype=int)
        m = (rows // 128 == rows.min() // 128) & (cols // 128 == cols.min() // 128)
        np.add.at(H32, ((rows[m] % 128) // 32, (cols[m] % 128) // 32), 1)
        print("first differing tile, per 32 x 32 block:\n", H32)
        m2 = m & ((cols % 128) < 32)
        print("  in its first block column: columns", sorted(set((cols[m2] % 128).tolist())), "rows (mod 128)", sorted(set((rows[m2] % 128).tolist()))[:40])
        for cc in sorted(set((cols[m2] % 128).tolist())):
            mm3 = m2 & ((cols % 128) == cc)
            print("   col", cc, "max rel diff", rel[mm3].max(), "n", mm3.sum())
print("x equal", np.array_equal(a["x"], b["x"]), "max rel x diff", np.max(np.abs(a["x"] - b["x"])) / np.max(np.abs(a["x"])))
