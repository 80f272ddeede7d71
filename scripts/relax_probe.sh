#!/bin/bash
t() { timeout 200 python3 scripts/probe.py $CFG 3 "$@" 2>/dev/null | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
for CFG in S-C3 S-metric; do
  for i in 1 2; do
  echo "$CFG a96s128: $(t relax_always=96 relax_small=128)  a64s128: $(t relax_always=64 relax_small=128)  a96s192m256: $(t relax_always=96 relax_small=192 relax_mid=256)  a96s128f.7: $(t relax_always=96 relax_small=128 relax_small_frac=0.7) a96s128m256f.25: $(t relax_always=96 relax_small=128 relax_mid=256 relax_mid_frac=0.25)"
  done
done
