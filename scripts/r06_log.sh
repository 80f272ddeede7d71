#!/bin/bash
# per-task log of one configuration under given env: r06_log.sh <case> <tag> [VAR=VAL ...]
mkdir -p gpurun_out
c=$1; tag=$2; shift 2
rm -f /tmp/dflog.txt
env "$@" OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
python scripts/df_log.py /tmp/dflog.txt > gpurun_out/r06_dflog_${tag}_$c.txt 2>&1
grep -E "^launch|  U K=|  TL|  TU|  D  |  T  |workers:|phases" gpurun_out/r06_dflog_${tag}_$c.txt | tail -40
