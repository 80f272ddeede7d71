#!/bin/bash
# round 6: column fragments of the update loop by lane rotation (OKKT_DF_ROT, compile time): micro-probe, bitwise tests, A/B timings
mkdir -p gpurun_out
{
echo "== probe"; timeout 120 scripts/_bin/lds_dpp_probe
echo "== dataflow tests (bitwise against the per-step launches)"; timeout 1200 python -m pytest tests/test_gpu_dataflow.py -x -q 2>&1 | tail -4
for c in S-metric S-C3 S-C5; do for v in rot0 rot1 rot0 rot1; do
  if [ $v = rot0 ]; then export OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_rot0.so; else unset OKKT_LIB_PATH; fi
  echo "== $c $v"; timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"
done; done
unset OKKT_LIB_PATH
} > gpurun_out/r06_rot.log 2>&1
cat gpurun_out/r06_rot.log
timeout 900 python -m pytest tests/test_gpu_env_variants.py tests/test_gpu_linear_solver.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r06_rot.log
