"""Look-ahead balance of the last big front (kernel trace of scripts/trace_probe.sh): for every trailing update of the last level,
its duration and the summed duration of the chain kernels (look-ahead update, diagonal blocks, trsm, in-group updates) that started
between its start and the next trailing update's start -- which of the two the super-step waited for."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void okkt::", "").replace("okkt::", "")
ends = [i for i, r in enumerate(rows) if nm(r).startswith("k_permute_in")]
hi = ends[-1]
asm = [i for i in range(hi) if nm(rows[i]).startswith("k_big_assemble")]
lo = asm[-1]                                    # the last level (the root front)
seg = rows[lo:hi]
trail = [r for r in seg if nm(r).startswith("k_big_syrk<0, 0")]
chain = [r for r in seg if nm(r).startswith(("k_big_syrk<0, 1", "k_big_syrk<0, 2", "k_big_diag", "k_big_trsm", "k_diag_trsm"))]
t0 = int(seg[0]["Start_Timestamp"])
print("trailing update:  start us   dur us | chain kernels until the next one: count  sum us | gap to next trailing start us")
for i, r in enumerate(trail):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nxt = int(trail[i + 1]["Start_Timestamp"]) if i + 1 < len(trail) else int(seg[-1]["End_Timestamp"])
    ch = [c for c in chain if s <= int(c["Start_Timestamp"]) < nxt]
    cs = sum(int(c["End_Timestamp"]) - int(c["Start_Timestamp"]) for c in ch) / 1e3
    print(f"{i:3d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} | {len(ch):3d} {cs:8.1f} | {(nxt - s) / 1e3:8.1f}  {'chain-bound' if cs > (e - s) / 1e3 * 1.1 else ('balanced' if cs > (e - s) / 1e3 * 0.9 else 'update-bound')}")
