#!/bin/bash
# round 6: how the factorisation time depends on the number of dataflow workers (is the root bound by the workers' throughput or by the chain under load?)
mkdir -p gpurun_out
{
for w in 256 224 192 160 128 96; do echo "== S-metric OKKT_DF_WORKERS=$w"; OKKT_DF_WORKERS=$w timeout 300 python scripts/probe.py S-metric 4 2>&1 | grep -E "rep [23]"; done
for w in 256 128 64; do echo "== S-C3 OKKT_DF_WORKERS=$w"; OKKT_DF_WORKERS=$w timeout 300 python scripts/probe.py S-C3 4 2>&1 | grep -E "rep [23]"; done
} > gpurun_out/r06_workers.log 2>&1
cat gpurun_out/r06_workers.log
