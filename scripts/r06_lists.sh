#!/bin/bash
# round 6: assembly launches over work lists (OKKT_ASM_LISTS=0: the full 2-D / 3-D grids)
mkdir -p gpurun_out
{
if [ -z "$SKIPTEST" ]; then timeout 1500 python -m pytest tests/test_gpu_linear_solver.py tests/test_gpu_dataflow.py tests/test_gpu_full_size.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -3; fi
for c in S-metric S-C3 S-C5; do for v in 0 1 0 1 0 1; do echo "== $c OKKT_ASM_LISTS=$v"; OKKT_ASM_LISTS=$v timeout 300 python scripts/probe.py $c 5 2>&1 | grep -E "rep [34]"; done; done
} > gpurun_out/r06_lists.log 2>&1
cat gpurun_out/r06_lists.log
