#!/bin/bash
# GPU box, round 5: TL tasks (panel tile + last update fused, default) against separate tasks; bitwise check against the per-step launches.
tag=${1:-a}; cases=${2:-"dense700 dense2600 S-C3 S-C5"}; cfgs=${3:-"S-metric S-C3"}
mkdir -p gpurun_out
{
echo "== bitwise check (default build)"
timeout 900 python scripts/df_check.py $cases 2>&1 | grep -v "^$" | tail -30
for c in $cfgs; do
  for e in "OKKT_DF_FUSE_TL=1" "OKKT_DF_FUSE_TL=0"; do
    echo "== $c [$e]"
    env $e timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
  done
done
} > gpurun_out/r05_tl_$tag.log 2>&1
tail -40 gpurun_out/r05_tl_$tag.log
