#!/bin/bash
# GPU box: kernel time line of one factor + solve of the S-C2 stand-in (banded KKT, N_h = 400).
export TMPDIR=/tmp
D=gpurun_out/sc2_trace; rm -rf $D; mkdir -p $D
OKKT_DEBUG_FRONTS=1 timeout 120 python3 scripts/sc2_probe.py 2>&1 | tail -4
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 scripts/sc2_probe.py > $D/log.txt 2>&1
python3 - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void okkt::", "").replace("okkt::", "")
rows = rows[-24:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {short(r['Kernel_Name']):40s} grid {r['Grid_Size_X']:>7s} wg {r['Workgroup_Size_X']:>4s} {d:8.1f} us")
PY
