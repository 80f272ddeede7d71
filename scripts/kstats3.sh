#!/bin/bash
# per-kernel calls per solve / factorisation of a small config: kstats3.sh <config> (solve_profile.py: 1 factorisation, 20 solves)
export TMPDIR=/tmp
CFG=${1:-S-small}
D=gpurun_out/ks3; rm -rf $D; mkdir -p $D
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/solve_profile.py $CFG 20 1 > $D/log.txt 2>&1
F=$(find $D -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    c = int(r["Calls"])
    print("%7.2f calls/solve  avg %7.2f us  total/solve %7.1f us  %s" % (c / 20.0, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 20e3, r["Name"][:60]))
PY
tail -1 $D/log.txt
