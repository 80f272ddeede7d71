#!/bin/bash
# GPU box: solve times and residuals of library builds.  usage: r05_solve_ab.sh <lib name | cur> ...
for c in S-metric S-C3 S-C5; do
  for v in "$@"; do
    if [ $v = cur ]; then unset OKKT_LIB_PATH; else export OKKT_LIB_PATH=scripts/_bin/lib_$v.so; fi
    echo -n "$v  "; timeout 300 python scripts/probe.py $c 3 2>&1 | tail -1
  done
done
