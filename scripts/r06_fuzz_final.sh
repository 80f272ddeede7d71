#!/bin/bash
mkdir -p gpurun_out
{
echo "== fuzz"; timeout 900 python scripts/fuzz_gpu.py 240 21 2>&1 | tail -2
echo "== fuzz x3"; timeout 900 python scripts/fuzz_gpu.py 80 22 3.0 2>&1 | tail -2
echo "== fuzz poison"; OKKT_DEBUG_POISON=1 timeout 900 python scripts/fuzz_gpu.py 160 23 2>&1 | tail -2
echo "== fuzz, experiments library"; OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_exp.so OKKT_DF_CHAIN=1 timeout 900 python scripts/fuzz_gpu.py 80 24 2>&1 | tail -2
} > gpurun_out/r06_fuzz_final.log 2>&1
cat gpurun_out/r06_fuzz_final.log
