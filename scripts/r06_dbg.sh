#!/bin/bash
mkdir -p gpurun_out
{
for c in dense129 dense300 dense640 dense2600; do
  echo "== default build, lockstep"; timeout 300 python scripts/r06_dbg.py $c | tail -3
  echo "== default build, old TU"; timeout 300 python scripts/r06_dbg.py $c OKKT_DF_LOCKSTEP=0 | tail -3
done
} > gpurun_out/r06_dbg.log 2>&1
grep -c "x equal True" gpurun_out/r06_dbg.log; grep -B3 "x equal False" gpurun_out/r06_dbg.log | head -20
