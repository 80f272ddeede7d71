#!/bin/bash
# GPU box: bitwise check of the dataflow launch against the per-step launches, then the per-task log of the named cases
# usage: scripts/df_run.sh "<check cases>" "<log cases>" tag
mkdir -p gpurun_out
timeout 900 python scripts/df_check.py $1 > gpurun_out/dfcheck_$3.log 2>&1
grep -v "^$" gpurun_out/dfcheck_$3.log | tail -40
for c in $2; do
  rm -f /tmp/dflog.txt
  OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 300 python scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
  python scripts/df_log.py /tmp/dflog.txt > gpurun_out/dflog_$3_$c.txt 2>&1
  echo "=== $c"; tail -32 gpurun_out/dflog_$3_$c.txt
done
