#!/bin/bash
# A/B of an environment setting against the default on the same GPU box (factor / solve ms of the last rep)
SETTING=$1; CFG=${2:-S-metric}; N=${3:-3}
for i in $(seq $N); do
  a=$(timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/')
  b=$(env $SETTING timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/')
  echo "default: $a   $SETTING: $b"
done
