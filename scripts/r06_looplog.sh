#!/bin/bash
rm -f /tmp/dflog.txt
OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_log.so OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 600 python3 scripts/df_check.py --run S-metric /tmp/x.npz > /dev/null 2>&1
python3 scripts/df_log.py /tmp/dflog.txt | grep -E "U K=512|launch +9:" | cut -c1-330
