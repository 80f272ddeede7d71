#!/bin/bash
# GPU box: the dataflow launch on one small dense front with parts of the task bodies switched off (OKKT_DEBUG_DATAFLOW)
mkdir -p gpurun_out
for dbg in ${DBGS:-15 8}; do
  echo "== OKKT_DEBUG_DATAFLOW=$dbg"
  OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=$dbg OKKT_DEBUG_FRONTS=1 timeout 60 python scripts/df_check.py --run ${1:-dense129} /tmp/x.npz 2>&1 | tail -12
  echo "rc ${PIPESTATUS[0]}"
done
