#!/bin/bash
t() { env "$@" timeout 200 python3 scripts/probe.py S-metric 3 schur=1 2>/dev/null | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
OKKT_DEBUG_FRONTS=1 timeout 200 python3 scripts/probe.py S-metric 1 schur=1 2>&1 | grep -v amdgpu | head -40
echo "default: $(t A=1)"
echo "LA=0: $(t OKKT_LOOKAHEAD=0)"
echo "mt=2500: $(t OKKT_LA_MIN_TILES=2500)"
echo "mt=100000000: $(t OKKT_LA_MIN_TILES=100000000)"
echo "split=0: $(t OKKT_SPLIT_HEAD=0)"
echo "group_big=2: $(t OKKT_GROUP_BIG=2)"
echo "reserved=16: $(t OKKT_RESERVED_CUS=16)"
