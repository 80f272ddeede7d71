#!/bin/bash
# round 6: one zeroing launch instead of three fills; extend-add with 8 instead of 4 row groups in flight (OKKT_ASM_UNROLL, build)
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_linear_solver.py tests/test_gpu_c_driver.py -x -q 2>&1 | tail -2
for w in 400 20000 lp; do echo "== small $w"; timeout 120 python3 scripts/r06_small_trace.py $w 2>&1 | grep -E "rep [345]"; done
for c in S-metric S-C3 S-C5; do for v in u4 u8 u4 u8; do
  if [ $v = u8 ]; then export OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_u8.so; else unset OKKT_LIB_PATH; fi
  echo "== $c $v"; timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"
done; done
unset OKKT_LIB_PATH
} > gpurun_out/r06_asm.log 2>&1
cat gpurun_out/r06_asm.log
