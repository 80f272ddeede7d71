#!/bin/bash
# the task logs of the evidence (scripts/r06_final_evidence.sh has the same lines): the product kernel + the log instrumentation
OUT=gpurun_out/profiles_r06
mkdir -p $OUT
for c in S-metric S-C3 S-C5; do
  rm -f /tmp/dflog.txt
  OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_log.so OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python3 scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
  python3 scripts/df_log.py /tmp/dflog.txt > $OUT/r06_dataflow_tasks_$c.txt 2>&1
done
head -10 $OUT/r06_dataflow_tasks_S-metric.txt; grep "U K=512  n" $OUT/r06_dataflow_tasks_S-metric.txt | head -2
