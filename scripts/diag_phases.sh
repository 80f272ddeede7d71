#!/bin/bash
# timing-only builds of k_big_diag: cumulative duration per phase (outputs are wrong for stop != 0)
export TMPDIR=/tmp
for stop in 1 7 8 2 0; do
  mkdir -p gpurun_out/dp$stop
  OKKT_LOOKAHEAD=0 OKKT_DEBUG_DIAG_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dp$stop -- python3 scripts/probe.py S-C3 1 > /dev/null 2>&1
  echo "stop=$stop $(grep k_big_diag gpurun_out/dp$stop/*/*kernel_stats.csv | cut -d, -f2-4)"
done
