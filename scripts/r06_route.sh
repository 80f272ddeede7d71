#!/bin/bash
mkdir -p gpurun_out
{
for c in S-metric S-C3 S-C5; do for v in 0 1; do echo "== $c OKKT_SOLVE_ROUTE_THIN=$v"; OKKT_SOLVE_ROUTE_THIN=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
for v in 0 1; do echo "== forward errors ROUTE_THIN=$v"; for c in S-metric S-C3 S-C5; do OKKT_SOLVE_ROUTE_THIN=$v timeout 600 python3 scripts/forward_error.py $c 2>&1 | grep "forward error"; done; done
echo "== fuzz"; timeout 900 python scripts/fuzz_gpu.py 240 11 2>&1 | tail -2
echo "== fuzz x3"; timeout 900 python scripts/fuzz_gpu.py 80 12 3.0 2>&1 | tail -2
echo "== fuzz poison"; OKKT_DEBUG_POISON=1 timeout 900 python scripts/fuzz_gpu.py 160 13 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_linear_solver.py tests/test_gpu_env_variants.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
} > gpurun_out/r06_route.log 2>&1
cat gpurun_out/r06_route.log
