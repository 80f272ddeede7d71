#!/bin/bash
# solve kernels of step_probe.py for one config under rocprofv3: kstats_solve.sh <tag> <config>; env passes through
export TMPDIR=/tmp
TAG=$1; CFG=${2:-S-metric}
D=gpurun_out/kss_$TAG; rm -rf $D; mkdir -p $D
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/step_probe.py $CFG > $D/log.txt 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r["Name"]
    if any(x in n for x in ("k_fwd", "k_bwd", "k_fs_", "k_bs_", "permute", "_mid")):
        tot += float(r["TotalDurationNs"])
        print("%9.1f us total  %6d calls  avg %8.2f us  %s" % (float(r["TotalDurationNs"]) / 1e3, int(r["Calls"]), float(r["AverageNs"]) / 1e3, n[:80]))
print("solve kernels total %.1f us over 13 solves = %.1f us per solve" % (tot / 1e3, tot / 1e3 / 13))
PY
