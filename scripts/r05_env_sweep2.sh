#!/bin/bash
# GPU box: factor times of S-metric for a few queue settings.  usage: r05_env_sweep2.sh "<env> <env>" ...
for v in "$@"; do
  echo -n "$v  "; env $v timeout 300 python scripts/step_probe.py S-metric 2>&1 | tail -1
done
