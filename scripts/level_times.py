"""Per-level breakdown of ONE factorisation from a rocprofv3 --kernel-trace csv (scripts/trace_probe.sh): the launches of
the last factorisation are cut into segments at every k_big_assemble* launch (each level of big fronts starts with one);
per segment: wall span and the summed duration of each kernel family."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0].replace("void okkt::", "").replace("okkt::", "")
    return n
names = [short(r["Kernel_Name"]) for r in rows]
# factorisations start with k_set_shift (or the first k_front_small after a solve); take the last k_set_shift
ends = [i for i, n in enumerate(names) if n.startswith("k_permute_in")]   # a solve follows every factorisation
hi = ends[-1]
prev = [i for i, n in enumerate(names[:hi]) if n.startswith("k_permute_out")]
lo = prev[-1] + 1 if prev else 0
seg = rows[lo:hi]; segn = names[lo:hi]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
print(f"factorisation: {len(seg)} launches, span {(t1 - t0) / 1e6:.3f} ms")
cuts = [i for i, n in enumerate(segn) if n.startswith("k_big_assemble")]
# merge consecutive assemble launches (several batches per level)
levels = []
for c in cuts:
    if levels and all(segn[j].startswith("k_big_assemble") or segn[j].startswith("k_count") for j in range(levels[-1], c)): continue
    levels.append(c)
bounds = [0] + levels + [len(seg)]
fam = lambda n: n.split("<")[0]
for a, b in zip(bounds[:-1], bounds[1:]):
    if a == b: continue
    s = int(seg[a]["Start_Timestamp"]); e = max(int(r["End_Timestamp"]) for r in seg[a:b])
    nxt = int(seg[b]["Start_Timestamp"]) if b < len(seg) else e
    d = collections.defaultdict(float); c = collections.defaultdict(int)
    for r, n in zip(seg[a:b], segn[a:b]):
        d[fam(n)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; c[fam(n)] += 1
    parts = "  ".join(f"{k.replace('k_big_', '').replace('k_', '')} {v:.0f}us/{c[k]}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:7])
    print(f"  t={(s - t0) / 1e3:9.1f} us  span {(max(e, nxt) - s) / 1e3:8.1f} us  launches {b - a:4d} | {parts}")
