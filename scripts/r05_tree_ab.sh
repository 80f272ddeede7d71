#!/bin/bash
# GPU box, round 5: tree launches (several levels of big fronts + their assembly in one dataflow launch, default) against one assembly
# launch + one dataflow launch per level (OKKT_DF_TREE=0); bitwise check against the per-step launches.
tag=${1:-a}; cases=${2:-"dense700 dense2600 S-C3 S-C5"}; cfgs=${3:-"S-metric S-C3 S-C5"}
mkdir -p gpurun_out
{
echo "== bitwise check (default build: tree launches)"
timeout 900 python scripts/df_check.py $cases 2>&1 | grep -v "^$" | tail -30
for c in $cfgs; do
  for e in "OKKT_DF_TREE=1" "OKKT_DF_TREE=0" "OKKT_DF_TREE=1" "OKKT_DF_TREE=0"; do
    echo "== $c [$e]"
    env $e timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
  done
done
} > gpurun_out/r05_tree_$tag.log 2>&1
tail -44 gpurun_out/r05_tree_$tag.log
