#!/bin/bash
# timing-only builds of k_big_diag: cumulative duration per phase (outputs are wrong for stop != 0)
# stop=1: load only; stop=2: no inverse; stop=0: full
export TMPDIR=/tmp
for stop in 1 3 4 2 0; do
  rm -rf gpurun_out/dp$stop; mkdir -p gpurun_out/dp$stop
  OKKT_LOOKAHEAD=0 OKKT_DEBUG_DIAG_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dp$stop -- python3 scripts/probe.py S-C3 1 > /dev/null 2>&1
  echo "stop=$stop $(grep 'k_big_diag(' gpurun_out/dp$stop/*/*kernel_stats.csv | sed 's/.*)",//')"
done
