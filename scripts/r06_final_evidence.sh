#!/bin/bash
# GPU box: the round-6 evidence in one call -- rocprofv3 statistics and PMC passes (profile_r06.sh), the default bench line, the task
# logs of the three configurations, the forward errors.  Everything lands in gpurun_out/profiles_r06/; copy what is to be judged to profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r06
mkdir -p $OUT
bash scripts/profile_r06.sh r06 > gpurun_out/profile_r06.log 2>&1
timeout 900 python3 bench.py > $OUT/r06_bench.json 2> gpurun_out/r06_bench.err
for c in S-metric S-C3 S-C5; do
  rm -f /tmp/dflog.txt
  # (the task log is instrumentation the product library does not carry: libonephase_kkt_log.so = the product kernel + the log)
  OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_log.so OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python3 scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
  python3 scripts/df_log.py /tmp/dflog.txt > $OUT/r06_dataflow_tasks_$c.txt 2>&1
done
: > $OUT/r06_forward_error.txt
for c in S-metric S-C3 S-C5; do timeout 600 python3 scripts/forward_error.py $c 2>&1 | grep "forward error" >> $OUT/r06_forward_error.txt; done
timeout 300 python3 scripts/sc5_err_probe.py S-C5 2>&1 | tail -2 >> $OUT/r06_forward_error.txt
ls -la $OUT; tail -3 $OUT/r06_bench.json | cut -c1-600
timeout 1200 python3 scripts/results_table.py $OUT/r06_results_table.json > gpurun_out/r06_results_table.log 2>&1
# the metric family at four times the size: what the released contribution blocks make room for, and what the launch reaches on large fronts
OKKT_DEBUG_FRONTS=1 timeout 600 python3 scripts/probe.py S-metric-4x 3 2>&1 | grep -E "front arena|rep |S-metric-4x|nnz_lower" > $OUT/r06_four_times_size.txt
