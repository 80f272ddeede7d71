#!/bin/bash
# like kstats.sh but through scripts/lowlevel_factor.py (ablation builds): kstats2.sh <tag> <config> <pattern>
export TMPDIR=/tmp
TAG=$1; CFG=${2:-S-C3}; PAT=${3:-diag}
D=gpurun_out/ks_$TAG; rm -rf $D; mkdir -p $D
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/lowlevel_factor.py $CFG 5 > $D/log.txt 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$PAT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%s calls %s avg %.2f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
