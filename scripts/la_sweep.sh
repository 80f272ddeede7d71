#!/bin/bash
# look-ahead tuning sweep: factor ms on S-metric for (reserved CUs, min tiles)
for rc in 8 16 32; do for mt in 256 1000 2500 5000; do
  echo "reserved=$rc min_tiles=$mt $(OKKT_RESERVED_CUS=$rc OKKT_LA_MIN_TILES=$mt python3 scripts/probe.py S-metric 2 | tail -1 | cut -c1-60)"
done; done
echo "lookahead off: $(OKKT_LOOKAHEAD=0 python3 scripts/probe.py S-metric 2 | tail -1 | cut -c1-60)"
