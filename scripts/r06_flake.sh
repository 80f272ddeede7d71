#!/bin/bash
# round 6: stress the one test that failed once in a full run (never again): 30 repetitions, with and without the NaN-poisoned arena
mkdir -p gpurun_out
: > gpurun_out/r06_flake.log
for i in $(seq 1 30); do
  if [ $((i % 3)) -eq 0 ]; then export OKKT_DEBUG_POISON=1; else unset OKKT_DEBUG_POISON; fi
  timeout 300 python -m pytest tests/test_gpu_linear_solver.py -x -q -k "test_levels_of_small_fronts_in_one_launch or test_structure_zoo" > /tmp/fl.log 2>&1
  rc=$?
  echo "run $i poison=${OKKT_DEBUG_POISON:-0} rc $rc $(tail -1 /tmp/fl.log)" >> gpurun_out/r06_flake.log
  if [ $rc -ne 0 ]; then cat /tmp/fl.log >> gpurun_out/r06_flake.log; fi
done
cat gpurun_out/r06_flake.log | cut -c1-300 | tail -60
