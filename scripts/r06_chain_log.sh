#!/bin/bash
mkdir -p gpurun_out
for v in ${CHV:-3 0}; do
  rm -f /tmp/dflog.txt
  OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_exp.so OKKT_DF_CHAIN=$v OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python3 scripts/df_check.py --run S-metric /tmp/x.npz > /dev/null 2>&1
  echo "== OKKT_DF_CHAIN=$v"; python3 scripts/r06_chain_log.py /tmp/dflog.txt
  python3 scripts/df_log.py /tmp/dflog.txt | grep -E "^launch +[0-9]:|U K=512"
done > gpurun_out/r06_chain_log.txt 2>&1
cat gpurun_out/r06_chain_log.txt
