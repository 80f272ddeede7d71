#!/bin/bash
mkdir -p gpurun_out
{
for c in S-metric S-C3; do for v in "0.6 1.0" "0.3 1.0" "1.0 1.0" "0.6 0.8" "0.6 1.3" "0.6 1.0"; do set -- $v; echo "== $c MODEL_CHAIN=$1 MODEL_BULK=$2"; OKKT_DF_MODEL_CHAIN=$1 OKKT_DF_MODEL_BULK=$2 timeout 300 python scripts/probe.py $c 5 2>&1 | grep -E "rep [34]"; done; done
} > gpurun_out/r06_model.log 2>&1
cat gpurun_out/r06_model.log
