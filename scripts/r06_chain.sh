#!/bin/bash
# round 6: chained update tasks (OKKT_DF_CHAIN = least column distance; 0 = off)
mkdir -p gpurun_out
{
if [ -z "$SKIPTEST" ]; then echo "== dataflow tests"; timeout 1200 python -m pytest tests/test_gpu_dataflow.py -x -q 2>&1 | tail -4; fi
for c in ${CFGS:-S-metric S-C3 S-C5}; do for v in ${CHAINS:-0 3 0 3}; do echo "== $c OKKT_DF_CHAIN=$v"; OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_exp.so OKKT_DF_CHAIN=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
} > gpurun_out/r06_chain.log 2>&1
cat gpurun_out/r06_chain.log
