"""Print a kernel timeline excerpt from a rocprofv3 --kernel-trace csv: the launches around the longest k_big_syrk."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    return n.replace("void okkt::", "").replace("okkt::", "")[:28]
# last factorisation: find the last k_big_assemble with the biggest grid... simply take the last longest syrk
syrk = [i for i, r in enumerate(rows) if "k_big_syrk" in r["Kernel_Name"]]
longest = max(syrk, key=lambda i: (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]), i))
asm = [i for i, r in enumerate(rows) if "k_big_assemble" in r["Kernel_Name"]]
lo = asm[-1] if (len(sys.argv) > 2 and sys.argv[2] == "root") else max(longest - int(sys.argv[2]) if len(sys.argv) > 2 else longest - 12, 0)
hi = min(lo + (int(sys.argv[3]) if len(sys.argv) > 3 else 60), len(rows))
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {short(r['Kernel_Name'])}")
