#!/bin/bash
# timing-only builds of k_big_syrk (outputs wrong for dbg != 0): bit0 no C load, bit1 no MFMA, bit2 no store, bit3 no prefetch
export TMPDIR=/tmp
for d in 85 93; do
  rm -rf gpurun_out/sp$d
  mkdir -p gpurun_out/sp$d
  OKKT_LOOKAHEAD=0 OKKT_DEBUG_SYRK=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp$d -- python3 scripts/probe.py S-metric 1 > /dev/null 2>&1
  echo "dbg=$d $(grep "k_big_syrk<[0-9]*, 0>" gpurun_out/sp$d/*/*kernel_stats.csv | awk -F'",' '{print $2}' | cut -d, -f1-3,6)"
done
