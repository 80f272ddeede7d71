#!/bin/bash
# GPU box, round 5: the 1024-thread worker (OKKT_DF_W16=1) against the 512-thread worker; bitwise check against the per-step launches.
tag=${1:-a}; cases=${2:-"dense700 dense2600 S-C3"}; cfgs=${3:-"S-metric S-C3 S-C5"}
mkdir -p gpurun_out
{
echo "== bitwise check with OKKT_DF_W16=1"
OKKT_DF_W16=1 timeout 900 python scripts/df_check.py $cases 2>&1 | grep -v "^$" | tail -24
for c in $cfgs; do
  for e in "OKKT_DF_W16=1" "OKKT_DF_W16=0"; do
    echo "== $c [$e]"
    env $e timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
  done
done
} > gpurun_out/r05_w16_$tag.log 2>&1
tail -40 gpurun_out/r05_w16_$tag.log
