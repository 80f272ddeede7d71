#!/bin/bash
export TMPDIR=/tmp
python3 scripts/probe.py S-C3 1 > /dev/null 2>&1; echo "plain exit: $?"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep1 -- python3 scripts/probe.py S-C3 1 > /dev/null 2>&1; echo "rocprof exit: $?"
OKKT_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep2 -- python3 scripts/probe.py S-C3 1 > /dev/null 2>&1; echo "rocprof LA=0 exit: $?"
