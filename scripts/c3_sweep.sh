#!/bin/bash
# medium-front tuning: factor / solve ms for group-of-one and look-ahead thresholds on S-C3 and S-metric
t() { env "$@" timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
for CFG in S-C3 S-metric; do
  echo "== $CFG"
  echo "default: $(t A=1)  $(t A=1)"
  for one in 2000 4000 6000 100000; do echo "one_rows=$one: $(t OKKT_GROUP_ONE_ROWS=$one)  la_off: $(t OKKT_GROUP_ONE_ROWS=$one OKKT_LOOKAHEAD=0)"; done
  for mt in 600 1000 1500 2500; do echo "la_min_tiles=$mt: $(t OKKT_LA_MIN_TILES=$mt)   with one_rows=4000: $(t OKKT_LA_MIN_TILES=$mt OKKT_GROUP_ONE_ROWS=4000)"; done
done
