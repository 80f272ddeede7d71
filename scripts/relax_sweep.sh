#!/bin/bash
# amalgamation sweep: relax_sweep.sh <config> ; prints factor / solve device ms per setting
CFG=${1:-S-metric}
for small in 256 384 512 1024; do for frac in 0.2 0.25 0.3 0.35 0.4; do for any in 0.03 0.08; do
  opt="relax_small=$small relax_small_frac=$frac relax_any_frac=$any"
  python scripts/probe.py $CFG 5 $opt 2>&1 | grep -E "nsuper|rep [34]" | sed -e "s/.*'flops_stored': \([0-9.e+]*\).*'nlevels': \([0-9]*\).*'n_big_fronts': \([0-9]*\).*/flops \1 levels \2 big \3/" -e "s/rep \([0-9]\): factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/f \2 s \3/" | tr "\n" " "
  echo " | $CFG $opt"
done; done; done
