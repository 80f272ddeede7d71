#!/bin/bash
# env_sweep.sh <config> "<VAR=a VAR2=b>" "<...>" ... : factor / solve ms of the last rep of probe.py under each setting (first line: default)
CFG=$1; shift
run() { env $1 timeout 150 python3 scripts/probe.py $CFG 3 2>/dev/null | tail -1 | sed 's/.*factor dev \([0-9.]*\) ms.*solve dev \([0-9.]*\) ms.*/\1 \2/'; }
echo "default: $(run A=1)"
for s in "$@"; do echo "$s: $(run "$s")"; done
echo "default: $(run A=1)"
