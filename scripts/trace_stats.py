"""Per-kernel, per-queue totals of a rocprofv3 --kernel-trace csv, plus idle gaps of the main queue between the
first k_big_assemble and the last kernel of each factorisation (delimited by k_set_shift / gaps > 1 ms)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
nfac = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void okkt::", "")[:40]
    k = (r["Queue_Id"], n)
    tot[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[k][1] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"q{k[0]:>2} {k[1]:42s} {v[0] / nfac / 1e3:9.3f} ms/factor  {v[1] / nfac:8.1f} launches  avg {v[0] / v[1]:8.1f} us")
