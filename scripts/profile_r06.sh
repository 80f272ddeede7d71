#!/bin/bash
# Round-6 rocprofv3 evidence (run on the GPU box via gpurun; every pass has a hard timeout).  Writes into
# gpurun_out/profiles_$R/: kernel statistics of the bench command, of BASELINE config 3 and of the solves, and the PMC
# summaries of the dominant kernel (MFMA utilisation, LDS bank conflicts, HBM bytes) and of the streaming solve kernels.
# Counters are collected in separate passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
export TMPDIR=/tmp
R=${1:-r06}
OUT=gpurun_out/profiles_$R
W=gpurun_out/prof_work_$R
mkdir -p $OUT $W
T="timeout -s KILL 240"
prof() { d=$1; shift; mkdir -p $W/$d; $T rocprofv3 --kernel-trace --output-format csv -d $W/$d "$@" > $W/$d.log 2>&1; }

# ---- kernel statistics
$T rocprofv3 --kernel-trace --stats --output-format csv -d $W/bench_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-kkt-level --no-live-pmc --no-sharded-model > $OUT/${R}_bench_under_rocprof.json 2> $W/bench_stats.log
cp $(find $W/bench_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
$T rocprofv3 --kernel-trace --stats --output-format csv -d $W/sc3_stats -- python3 scripts/probe.py S-C3 5 > $W/sc3_stats.log 2>&1
cp $(find $W/sc3_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_sc3_kernel_stats.csv
$T rocprofv3 --kernel-trace --stats --output-format csv -d $W/solve_stats -- python3 scripts/solve_profile.py S-metric 20 1 > $W/solve_stats.log 2>&1
cp $(find $W/solve_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_solve_kernel_stats.csv
$T rocprofv3 --kernel-trace --stats --output-format csv -d $W/solve4_stats -- python3 scripts/solve_profile.py S-metric 10 4 > $W/solve4_stats.log 2>&1
cp $(find $W/solve4_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_solve_batch4_kernel_stats.csv

# ---- PMC passes on one factorisation + solves of the metric workload
prof pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -- python3 scripts/solve_profile.py S-metric 3 1
prof pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -- python3 scripts/solve_profile.py S-metric 3 1
prof pmc_fetch --pmc FETCH_SIZE -- python3 scripts/solve_profile.py S-metric 3 1
prof pmc_write --pmc WRITE_SIZE -- python3 scripts/solve_profile.py S-metric 3 1
python3 scripts/pmc_summary.py $W $OUT $R
ls -la $OUT
