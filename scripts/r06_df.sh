#!/bin/bash
# round 6: the dataflow tests (bitwise against the per-step launches), timings, the per-task log
mkdir -p gpurun_out
T=${1:-df}
timeout 900 python -m pytest tests/test_gpu_dataflow.py -x -q > gpurun_out/r06_${T}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_${T}_pytest.log
tail -15 gpurun_out/r06_${T}_pytest.log
for c in S-C3 S-C5 S-metric; do timeout 300 python scripts/probe.py $c 4 > gpurun_out/r06_${T}_probe_$c.log 2>&1; tail -2 gpurun_out/r06_${T}_probe_$c.log; done
for c in ${2:-S-C3}; do
  rm -f /tmp/dflog.txt
  OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 300 python scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
  python scripts/df_log.py /tmp/dflog.txt > gpurun_out/r06_${T}_dflog_$c.txt 2>&1
  grep -A14 "^chain of front" gpurun_out/r06_${T}_dflog_$c.txt | head -40
  grep "phases" gpurun_out/r06_${T}_dflog_$c.txt | head
done
