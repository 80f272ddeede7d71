#!/bin/bash
# GPU box: durations of the assembly launches (and everything else by family) of the last factorisation of a configuration.
# usage: scripts/r05_asm_trace.sh <tag> <config> [env...]
export TMPDIR=/tmp
tag=$1; c=${2:-S-metric}; shift 2
D=gpurun_out/asm_$tag; rm -rf $D; mkdir -p $D
env "$@" timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 scripts/probe.py $c 3 > $D/log.txt 2>&1
tail -1 $D/log.txt
python3 - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void okkt::", "").replace("okkt::", "")
names = [short(r["Kernel_Name"]) for r in rows]
ends = [i for i, n in enumerate(names) if n.startswith("k_permute_in")]
hi = ends[-1]
prev = [i for i, n in enumerate(names[:hi]) if n.startswith("k_permute_out")]
lo = prev[-1] + 1 if prev else 0
tot = 0.0
for r, n in zip(rows[lo:hi], names[lo:hi]):
    if n.startswith("k_big_assemble"):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        print(f"  {n:32s} grid {r['Grid_Size_X']:>7s} x {r['Grid_Size_Y']:>4s} x {r['Grid_Size_Z']:>3s}  {d:8.1f} us")
print(f"  assembly launches of one factorisation: {tot:.1f} us")
PY
