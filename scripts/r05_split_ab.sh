#!/bin/bash
# GPU box, round 5: the two-kernel form of the dataflow levels (k_front_bulk + chain kernel) against the per-step launches (bitwise)
# and against the one-kernel form (time).  usage: scripts/r05_split_ab.sh <tag> ["check cases"] ["probe configs"]
tag=${1:-a}; cases=${2:-"dense700 dense2600 S-C3"}; cfgs=${3:-"S-metric"}
mkdir -p gpurun_out
{
echo "== bitwise check, split forced on every eligible level"
OKKT_DF_SPLIT_MIN_TASKS=0 timeout 900 python scripts/df_check.py $cases 2>&1 | grep -v "^$" | tail -30
for c in $cfgs; do
  echo "== $c: two-kernel form (default)"
  OKKT_DEBUG_FRONTS=1 timeout 300 python scripts/step_probe.py $c 2>&1 | grep -v "big front level\|^$" | tail -16
  echo "== $c: one-kernel form"
  OKKT_DF_SPLIT_FRONTS=0 timeout 300 python scripts/step_probe.py $c 2>&1 | tail -2
done
} > gpurun_out/r05_split_$tag.log 2>&1
tail -60 gpurun_out/r05_split_$tag.log
