#!/bin/bash
mkdir -p gpurun_out
{
for c in S-metric S-C3 S-C5; do for v in 4 5 6 8 4 6; do echo "== $c OKKT_DF_GROUP=$v (chain off)"; OKKT_DF_CHAIN=0 OKKT_DF_GROUP=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
} > gpurun_out/r06_group.log 2>&1
cat gpurun_out/r06_group.log
