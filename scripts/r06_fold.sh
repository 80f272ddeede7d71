#!/bin/bash
mkdir -p gpurun_out
T=${1:-fold}
for c in S-metric S-C3 S-C5; do
  for v in 0 3; do echo "== $c OKKT_FOLD_LONE=$v"; OKKT_FOLD_LONE=$v timeout 300 python scripts/probe.py $c 4 2>&1 | tail -2; done
done > gpurun_out/r06_${T}_probe.log 2>&1
cat gpurun_out/r06_${T}_probe.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_${T}_pytest.log 2>&1; tail -5 gpurun_out/r06_${T}_pytest.log
