#!/bin/bash
# GPU box: per-task log, summary of a chosen launch index.  usage: scripts/r05_split_log2.sh <tag> <config> <launch> [env...]
tag=$1; c=$2; li=$3; shift 3
mkdir -p gpurun_out; rm -f /tmp/dflog.txt
env "$@" OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
OKKT_DF_LOG_SAVE=gpurun_out/dflog_${tag}_${c}_$li.npz python scripts/df_log.py /tmp/dflog.txt $li > gpurun_out/dflog_${tag}_${c}_$li.txt 2>&1
grep -v "^launch  " gpurun_out/dflog_${tag}_${c}_$li.txt | head -60
