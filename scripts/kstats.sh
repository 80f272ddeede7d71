#!/bin/bash
# kernel statistics of probe.py for one config under rocprofv3: kstats.sh <tag> <config> [reps]; env passes through
export TMPDIR=/tmp
TAG=$1; CFG=${2:-S-C3}; REPS=${3:-5}
D=gpurun_out/ks_$TAG; rm -rf $D; mkdir -p $D
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/probe.py $CFG $REPS > $D/log.txt 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(12)]:
    print("%9.1f us total  %6d calls  avg %8.2f us  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e3, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:70]))
PY
