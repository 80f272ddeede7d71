#!/bin/bash
# round 6: what a CUTEst-size system pays for -- kernel traces of S-C2 (N_h = 400, 20000) and S-C4
export TMPDIR=/tmp
mkdir -p gpurun_out/small
for w in 400 20000 lp; do
  rm -rf /tmp/tr_$w
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$w -- python3 scripts/r06_small_trace.py $w > gpurun_out/small/run_$w.log 2>&1
  f=$(find /tmp/tr_$w -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > gpurun_out/small/trace_$w.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factor + solve: print the final 60 launches with gaps
t0 = None
for r in rows[-70:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s; prev = s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}  {r['Kernel_Name'][:90]}")
    prev = e
PY
  tail -3 gpurun_out/small/run_$w.log
done
