#!/bin/bash
mkdir -p gpurun_out
{
for v in 0 3; do echo "== S-metric OKKT_FOLD_LONE=$v"; OKKT_FOLD_LONE=$v OKKT_DEBUG_FRONTS=1 timeout 300 python scripts/probe.py S-metric 4 2>&1 | grep -E "level [01] units|rep [123]"; done
for c in S-C3 S-C5; do for v in 0 3; do echo "== $c OKKT_FOLD_LONE=$v"; OKKT_FOLD_LONE=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
echo "== fuzz"; timeout 900 python scripts/fuzz_gpu.py 240 7 2>&1 | tail -3
echo "== fuzz x3"; timeout 900 python scripts/fuzz_gpu.py 60 8 3.0 2>&1 | tail -3
} > gpurun_out/r06_lone.log 2>&1
cat gpurun_out/r06_lone.log
