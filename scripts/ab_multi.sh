#!/bin/bash
# alternating runs of probe.py over several builds (scripts/_bin/lib_<name>.so; "cur" = the in-tree library)
CFG=${CFG:-S-metric}; N=${N:-2}
for i in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = cur ]; then unset OKKT_LIB_PATH; else export OKKT_LIB_PATH=scripts/_bin/lib_$v.so; fi
    r=$(timeout 150 python3 scripts/probe.py $CFG 3 | tail -2 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*resid \(.*\)/\1 \2 \3/' | tr '\n' ' ')
    echo "$v: $r"
  done
done
