#!/bin/bash
# round 6, first contact: the GPU suite (tier-1 changes), timings of the BASELINE configurations, the per-task log of S-C3
mkdir -p gpurun_out
T=${1:-base}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_${T}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_${T}_pytest.log
tail -5 gpurun_out/r06_${T}_pytest.log
for c in S-C3 S-C5 S-metric; do timeout 300 python scripts/probe.py $c 4 > gpurun_out/r06_${T}_probe_$c.log 2>&1; tail -2 gpurun_out/r06_${T}_probe_$c.log; done
for c in S-C3; do
  rm -f /tmp/dflog.txt
  OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 300 python scripts/df_check.py --run $c /tmp/x.npz > /dev/null 2>&1
  python scripts/df_log.py /tmp/dflog.txt > gpurun_out/r06_${T}_dflog_$c.txt 2>&1
  tail -40 gpurun_out/r06_${T}_dflog_$c.txt
done
