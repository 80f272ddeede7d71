#!/bin/bash
# round 6: pairs of row tiles as one stream (sequential, product library) with every row tile published as its stores drain
mkdir -p gpurun_out
{
if [ -z "$SKIPTEST" ]; then echo "== dataflow tests"; timeout 1500 python -m pytest tests/test_gpu_dataflow.py -x -q 2>&1 | tail -3; fi
for c in ${CFGS:-S-metric S-C3 S-C5}; do
  for v in "1 24 4 1" "2 24 4 1" "2 24 4 0" "2 12 4 1" "2 4 2 1" "2 0 0 1" "1 24 4 1" "2 24 4 1"; do
    set -- $v
    echo "== $c ROWS_BIG=$1 AHEAD=$2 COLDIST=$3 EARLY_PUB=$4"
    OKKT_DF_ROWS_BIG=$1 OKKT_DF_ROWS_AHEAD=$2 OKKT_DF_ROWS_COLDIST=$3 OKKT_DF_EARLY_PUB=$4 timeout 300 python scripts/probe.py $c 5 2>&1 | grep -E "rep [34]"
  done
done
} > gpurun_out/r06_pairs.log 2>&1
cat gpurun_out/r06_pairs.log
