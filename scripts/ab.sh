#!/bin/bash
# A/B of two builds on the same GPU box: alternating runs of probe.py (factor ms of the last rep)
A=${1:-scripts/_bin/lib_base.so}; CFG=${2:-S-metric}; N=${3:-4}
for i in $(seq $N); do
  a=$(OKKT_LIB_PATH=$A timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/')
  b=$(timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/')
  echo "base: $a   new: $b"
done
