#!/bin/bash
mkdir -p gpurun_out
{
for c in S-metric S-C3 S-C5; do for v in 384 128 192 256 320 384; do echo "== $c OKKT_SOLVE_MID=$v"; OKKT_SOLVE_MID=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
} > gpurun_out/r06_mid2.log 2>&1
cat gpurun_out/r06_mid2.log
