#!/bin/bash
# amalgamation sweep around the defaults: factor / solve ms
t() { timeout 200 python3 scripts/probe.py $CFG 3 "$@" 2>/dev/null | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
for CFG in S-C3 S-metric; do
  echo "$CFG default: $(t)  any.06: $(t relax_any_frac=0.06)  any.1: $(t relax_any_frac=0.1)  mid192f.15: $(t relax_mid=192)  mid256f.1: $(t relax_mid=256 relax_mid_frac=0.1) small128f.35: $(t relax_small_frac=0.35) small160: $(t relax_small=160)"
done
