#!/bin/bash
mkdir -p gpurun_out
{
echo "== baseline"; timeout 300 python scripts/probe.py S-metric 4 2>&1 | tail -3
for cd in 2 4 6 10; do for ah in 0 12; do for mk in 8; do
  echo "== MACRO ROWS_BIG=2 COLDIST=$cd AHEAD=$ah MINKB=$mk"; OKKT_DF_ROWS_BIG=2 OKKT_DF_ROWS_COLDIST=$cd OKKT_DF_ROWS_AHEAD=$ah OKKT_DF_ROWS_MINKB=$mk timeout 300 python scripts/probe.py S-metric 4 2>&1 | tail -2
done; done; done
for c in S-C3 S-C5; do
echo "== $c baseline"; timeout 300 python scripts/probe.py $c 4 2>&1 | tail -2
echo "== $c MACRO COLDIST=4 AHEAD=0 MINKB=8"; OKKT_DF_ROWS_BIG=2 OKKT_DF_ROWS_COLDIST=4 OKKT_DF_ROWS_AHEAD=0 OKKT_DF_ROWS_MINKB=8 timeout 300 python scripts/probe.py $c 4 2>&1 | tail -2
done
} > gpurun_out/r06_rows3.log 2>&1
cat gpurun_out/r06_rows3.log
