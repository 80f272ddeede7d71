#!/bin/bash
# PMC passes for the dominant kernel (separate runs, kernel-trace only; see MI355X_MICROARCH.md)
export TMPDIR=/tmp
CFG=${1:-S-metric}
run() { name=$1; shift; mkdir -p gpurun_out/pmc_$name; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python3 scripts/probe.py $CFG 1 > /dev/null 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
python3 - <<'PY'
import csv, glob, collections
for name in ["sq","tcc","fetch","write","lds"]:
    files = glob.glob(f"gpurun_out/pmc_{name}/*/*counter_collection.csv")
    if not files: print(name, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    for k, d in agg.items():
        if "syrk" in k or "diag" in k or "trsm" in k or "assemble" in k:
            print(name, k, {c: f"{v:.4g}" for c, v in d.items()})
PY
