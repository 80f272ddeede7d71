"""Launches that cannot fill the chip: from a rocprofv3 kernel trace (scripts/trace_probe.sh), per kernel family the launches of
fewer than 256 workgroups and their summed duration in the last factorisation + solve.  Usage: small_grids.py <trace dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void okkt::", "").replace("okkt::", "") for r in rows]
outs = [i for i, n in enumerate(names) if n.startswith("k_permute_out")]
hi = outs[-1] + 1; lo = outs[-2] + 1 if len(outs) > 1 else 0
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0, 0])
for r, n in zip(rows[lo:hi], names[lo:hi]):
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    wgs = g // max(w, 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[n.split("<")[0] + ("<" + n.split("<")[1] if "<" in n else "")]
    a[0] += 1; a[1] += d
    if wgs < 256: a[2] += 1; a[3] += d; a[4] = max(a[4], wgs)
print(f"{'kernel':44s} launches  total us | < 256 workgroups: launches  total us  (largest grid)")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    if a[3] > 0: print(f"{k[:44]:44s} {a[0]:8d} {a[1]:9.1f} | {a[2]:8d} {a[3]:9.1f}  ({a[4]})")
