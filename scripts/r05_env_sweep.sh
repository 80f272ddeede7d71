#!/bin/bash
# GPU box: step_probe of one configuration under several environment settings.  usage: scripts/r05_env_sweep.sh <tag> <config> "ENV=.. ENV=.." "ENV=.." ...
tag=$1; c=$2; shift 2
mkdir -p gpurun_out
for e in "$@"; do
  echo "== $c [$e]"
  env $e timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
done > gpurun_out/r05_sweep_$tag.log 2>&1
cat gpurun_out/r05_sweep_$tag.log
