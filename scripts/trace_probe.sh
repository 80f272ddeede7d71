#!/bin/bash
# kernel trace (csv) of probe.py for one config: trace_probe.sh <tag> <config> [reps] [probe options]; the csv comes back in gpurun_out/tr_<tag>/
export TMPDIR=/tmp
TAG=$1; CFG=${2:-S-C3}; REPS=${3:-2}; shift 3
D=gpurun_out/tr_$TAG; rm -rf $D; mkdir -p $D
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 scripts/probe.py $CFG $REPS "$@" > $D/log.txt 2>&1
find $D -name "*agent_info.csv" -delete
ls -la $D/*/ | head
