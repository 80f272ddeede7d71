#!/bin/bash
# amalgamation sweep: factor ms and stored flops for relax_any_frac / relax_mid settings
for cfg in S-metric S-C3; do
for a in 0.03 0.06 0.1 0.15 0.25 0.4; do
  echo "$cfg relax_any_frac=$a $(python3 scripts/probe.py $cfg 2 relax_any_frac=$a 2>&1 | grep -E "^rep 1|nsuper" | sed 's/.*flops_stored.: \([0-9.e+]*\).*nsuper.: \([0-9]*\).*n_big_fronts.: \([0-9]*\).*/stored_flops \1 nsuper \2 big \3/' | cut -c1-110 | tr '\n' ' ')"
done; done
