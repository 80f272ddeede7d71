#!/bin/bash
# A/B on one box: library variants (onephase.jl_amd/libonephase_kkt_<name>.so) against the tree's
mkdir -p gpurun_out
{
for c in ${CFGS:-S-metric S-C3 S-C5}; do for v in ${VARS:-old new lean old new lean}; do
  if [ $v = new ]; then unset OKKT_LIB_PATH; else export OKKT_LIB_PATH=$PWD/onephase.jl_amd/libonephase_kkt_$v.so; fi
  echo "== $c $v"; timeout 300 python scripts/probe.py $c 5 2>&1 | grep -E "rep [234]"
done; done
unset OKKT_LIB_PATH
} > gpurun_out/r06_ab_old.log 2>&1
cat gpurun_out/r06_ab_old.log
