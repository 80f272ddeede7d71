#!/bin/bash
# round 6: widest pivot block solved by block substitution (OKKT_SOLVE_MID) with the thin fronts routed through the same launches
mkdir -p gpurun_out
{
for c in S-metric S-C3 S-C5; do for v in 384 512 768 1024 384; do echo "== $c OKKT_SOLVE_MID=$v"; OKKT_SOLVE_MID=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
for v in 384 1024; do echo "== forward errors MID=$v"; for c in S-metric S-C3 S-C5; do OKKT_SOLVE_MID=$v timeout 600 python3 scripts/forward_error.py $c 2>&1 | grep "forward error"; done; done
} > gpurun_out/r06_mid.log 2>&1
cat gpurun_out/r06_mid.log
