#!/bin/bash
# round 6, timing experiment: how much faster is the factorisation when every bulk update task does HALF of its products (wrong numbers)?
mkdir -p gpurun_out
{
for c in S-metric; do for v in 0 1 0 1; do echo "== $c OKKT_DEBUG_DF_HALF=$v (chain off)"; OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_exp.so OKKT_DF_CHAIN=0 OKKT_DEBUG_DF_HALF=$v timeout 300 python scripts/probe.py $c 4 2>&1 | grep -E "rep [23]"; done; done
for v in 0 1; do
  rm -f /tmp/dflog.txt
  OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_exp.so OKKT_DF_CHAIN=0 OKKT_DEBUG_DF_HALF=$v OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python3 scripts/df_check.py --run S-metric /tmp/x.npz > /dev/null 2>&1
  echo "== task log, OKKT_DEBUG_DF_HALF=$v"; python3 scripts/df_log.py /tmp/dflog.txt | grep -E "^launch +[0-9]:|U K=512  n|mean distance|^ +(0|4|8|16|24|32|40|48|56|64) \|"
done
} > gpurun_out/r06_half.log 2>&1
cat gpurun_out/r06_half.log
