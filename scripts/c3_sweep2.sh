#!/bin/bash
t() { env "$@" timeout 120 python3 scripts/probe.py $CFG 3 | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
CFG=S-C3
for i in 1 2; do
echo "default: $(t A=1)"
echo "LA=0: $(t OKKT_LOOKAHEAD=0)"
echo "masked, no la, no split: $(t OKKT_LA_MIN_TILES=100000000 OKKT_SPLIT_HEAD=0)"
echo "masked(1 CU), no la, no split: $(t OKKT_LA_MIN_TILES=100000000 OKKT_SPLIT_HEAD=0 OKKT_RESERVED_CUS=1)"
echo "masked(1 CU) default: $(t OKKT_RESERVED_CUS=1)"
done
