#!/bin/bash
mkdir -p gpurun_out
{
for c in S-metric S-C3 S-C5; do
  for v in 0 1; do echo "== $c OKKT_RELEASE_CB=$v"; OKKT_RELEASE_CB=$v OKKT_DEBUG_FRONTS=1 timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "front arena|arena_bytes|rep [23]"; done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
} > gpurun_out/r06_mem.log 2>&1
cat gpurun_out/r06_mem.log
