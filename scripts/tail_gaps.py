"""Gaps between consecutive kernels of the chain-bound tail of the last big front (kernel trace of scripts/trace_probe.sh):
for the last N launches of the last factorisation: kernel, duration, gap to the previous kernel's end."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void okkt::", "").replace("okkt::", "") for r in rows]
ends = [i for i, n in enumerate(names) if n.startswith("k_permute_in")]
hi = ends[-1]
seg = [(names[i], int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])) for i in range(max(0, hi - 400), hi)]
# keep the chain kernels only (not the block inversions of the auxiliary stream)
chain = [x for x in seg if not x[0].startswith(("k_xinv", "k_big_invert"))]
tail = chain[-N:]
tot_k = sum(e - s for _, s, e in tail); span = tail[-1][2] - tail[0][1]
print(f"last {N} chain launches: span {span / 1e3:.1f} us, kernel time {tot_k / 1e3:.1f} us, gaps {(span - tot_k) / 1e3:.1f} us")
prev = None
agg = {}
for n, s, e in tail:
    fam = n.split("<")[0]
    a = agg.setdefault(fam, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    if prev is not None: a[2] += max(0, s - prev) / 1e3
    prev = e
for fam, (c, d, g) in agg.items(): print(f"  {fam:28s} {c:3d} launches, avg {d / c:6.1f} us, avg gap before it {g / c:5.1f} us")
