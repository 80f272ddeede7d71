#!/bin/bash
# GPU box: per-task log of a configuration in the two-kernel form.  usage: scripts/r05_split_log.sh <tag> <config> [env...]
tag=$1; c=$2; shift 2
mkdir -p gpurun_out; rm -f /tmp/dflog.txt
env "$@" OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
OKKT_DF_LOG_SAVE=gpurun_out/dflog_${tag}_$c.npz python scripts/df_log.py /tmp/dflog.txt > gpurun_out/dflog_${tag}_$c.txt 2>&1
grep -n "^launch" gpurun_out/dflog_${tag}_$c.txt | tail -12
tail -52 gpurun_out/dflog_${tag}_$c.txt | head -36
