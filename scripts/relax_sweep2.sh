#!/bin/bash
# second amalgamation sweep around the round-3 default (relax_small 256 / 0.25): the other rules
CFG=${1:-S-metric}
for opt in "relax_always=64" "relax_always=32" "relax_always=96" "relax_always=128" "relax_mid=192 relax_mid_frac=0.15" "relax_mid=192 relax_mid_frac=0.2" "relax_mid=128 relax_mid_frac=0.2" "relax_any_frac=0.06" "relax_any_frac=0.1" "relax_any_frac=0.15" "relax_small=320 relax_small_frac=0.25" "relax_small=256 relax_small_frac=0.28"; do
  python scripts/probe.py $CFG 5 $opt 2>&1 | grep -E "nsuper|rep [34]" | sed -e "s/.*'flops_stored': \([0-9.e+]*\).*'nlevels': \([0-9]*\).*'n_big_fronts': \([0-9]*\).*/flops \1 levels \2 big \3/" -e "s/rep \([0-9]\): factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/f \2 s \3/" | tr "\n" " "
  echo " | $CFG $opt"
done
