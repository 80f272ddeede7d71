#!/bin/bash
# rocprofv3 summaries of the bench command (run on the GPU box via gpurun); copies the judged files to gpurun_out/profiles_new
export TMPDIR=/tmp
R=${1:-r01}
OUT=gpurun_out/profiles_new
mkdir -p $OUT gpurun_out/pb_stats gpurun_out/pb_fetch gpurun_out/pb_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pb_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-kkt-level > $OUT/${R}_bench_under_rocprof.json 2> /dev/null
cp gpurun_out/pb_stats/*/*kernel_stats.csv $OUT/${R}_bench_kernel_stats.csv
# HBM traffic of the dominant kernel: separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pb_fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-kkt-level > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pb_write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-kkt-level > /dev/null 2>&1
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
def total(d, counter):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    tot = 0.0; n = 0
    for r in csv.DictReader(open(f)):
        if "k_big_syrk<0, 0>" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
fs, n1 = total("pb_fetch", "FETCH_SIZE")
ws, n2 = total("pb_write", "WRITE_SIZE")
# MI355X_MICROARCH.md, HBM: FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reads 1/2 of a wide coalesced
# stream -- the C tile is read with 8-byte lanes (uncalibrated width), the operands with 16-byte LDS-DMA;
# both bounds are recorded: raw (x1) and doubled (x2)
out = dict(kernel="k_big_syrk<0, 0>", launches=n1, fetch_kib_total=fs, write_kib_total=ws,
           hbm_bytes_per_launch_raw=(fs + ws) * 1024 / max(n1, 1),
           hbm_bytes_per_launch=(2 * fs + ws) * 1024 / max(n1, 1),
           note="hbm_bytes_per_launch doubles FETCH_SIZE as the guide prescribes for 16-B/lane streams; raw keeps it as reported")
json.dump(out, open(f"gpurun_out/profiles_new/{R}_syrk_pmc.json", "w"), indent=1)
print(out)
PY
head -8 $OUT/${R}_bench_kernel_stats.csv | cut -c1-150; tail -1 $OUT/${R}_bench_under_rocprof.json | cut -c1-300
