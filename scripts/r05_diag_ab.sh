#!/bin/bash
# GPU box, round 5: the diagonal-block factorisation with the 8 x 8 block factored once per micro-step (FACW, default) against the variant
# in which every row thread repeats it (scripts/_bin/lib_nofacw.so): bitwise check, step times, D phases of the S-C3 root.
tag=${1:-a}
mkdir -p gpurun_out
{
echo "== bitwise check against the per-step launches (default build)"
timeout 900 python scripts/df_check.py dense700 dense2600 S-C3 2>&1 | grep -v "^$" | tail -8
for c in S-C3 S-C5 S-metric; do
  for v in cur nofacw; do
    if [ $v = cur ]; then unset OKKT_LIB_PATH; else export OKKT_LIB_PATH=scripts/_bin/lib_$v.so; fi
    echo "== $c [$v]"; timeout 300 python scripts/step_probe.py $c 2>&1 | tail -1
  done
done
unset OKKT_LIB_PATH
rm -f /tmp/dflog.txt
OKKT_DATAFLOW=1 OKKT_DEBUG_DATAFLOW=16 OKKT_DF_LOG=/tmp/dflog.txt timeout 300 python scripts/df_check.py --run S-C3 /tmp/x.npz > /dev/null 2>&1
python scripts/df_log.py /tmp/dflog.txt > gpurun_out/dflog_diag_${tag}_S-C3.txt 2>&1
grep "D phases\|mean distance\|TU phases" gpurun_out/dflog_diag_${tag}_S-C3.txt
} > gpurun_out/r05_diag_$tag.log 2>&1
tail -30 gpurun_out/r05_diag_$tag.log
