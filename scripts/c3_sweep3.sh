#!/bin/bash
t() { env "$@" timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
export OKKT_LA_MIN_TILES=600
for CFG in S-C3 S-metric; do
echo "== $CFG (mt=600)"
for i in 1 2; do
echo "default: $(t A=1)   LA=0: $(t OKKT_LOOKAHEAD=0)  split=0: $(t OKKT_SPLIT_HEAD=0)  splitrows=5000: $(t OKKT_SPLIT_MIN_ROWS=5000) splitrows=8000: $(t OKKT_SPLIT_MIN_ROWS=8000)"
done; done
