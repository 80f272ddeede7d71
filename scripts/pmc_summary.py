"""Summarise the rocprofv3 PMC passes of scripts/profile_r02.sh into profiles/<round>_pmc_summary.json (+ the syrk record
bench.py quotes).  Usage: pmc_summary.py <work dir> <out dir> <round>.

Units and corrections (MI355X_MICROARCH.md, sections "HBM" and "rocprofv3 PMC slots"): FETCH_SIZE / WRITE_SIZE are KiB
from the L2's memory-side request counters; FETCH_SIZE reports half the bytes of a 16-byte-per-lane streaming read on
gfx950 and is doubled for kernels that read that way (flag `wide` below), WRITE_SIZE is exact; SQ_* cycle counters are
summed over all SIMDs of the dispatch, GRBM_GUI_ACTIVE over the 8 XCDs (checked on the largest trailing update: the sum is
8 x duration x 2.1 GHz).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x SIMDs) with active cycles =
GRBM_GUI_ACTIVE / 8 and 1024 SIMDs -- the gfx94x derived-counter formula MfmaUtil with the per-XCD sum taken out (ROCm 7.2
ships no gfx950 formula)."""
import collections
import csv
import glob
import json
import sys

work, out, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
CUS = 256
# kernels of interest: (substring, label, reads are 16 B / lane ("wide": FETCH_SIZE doubled))
KERNELS = [
    ("k_front_dataflow", "k_front_dataflow (dominant: one persistent launch per level of big fronts)", True),
    ("k_big_syrk<0, 0,", "k_big_syrk<0,0> trailing update (per-step schedule, OKKT_DATAFLOW=0)", True),
    ("k_big_syrk<0, 1,", "k_big_syrk<0,1> in-group update", True),
    ("k_big_syrk<0, 2,", "k_big_syrk<0,2> look-ahead columns", True),
    ("k_big_trsm", "k_big_trsm", False),
    ("k_big_diag", "k_big_diag", False),
    ("k_big_assemble_chunked", "k_big_assemble_chunked", False),
    ("k_big_assemble<", "k_big_assemble", False),
    ("k_fwd_upd", "k_fwd_upd (solve: panel GEMV forward)", True),
    ("k_bwd_upd", "k_bwd_upd (solve: panel GEMV backward)", False),
    ("k_fwd_y", "k_fwd_y (solve: block product)", False),
    ("k_bwd_x", "k_bwd_x (solve: block product)", False),
    ("k_bwd_pre", "k_bwd_pre (solve)", False),
    ("k_fwd_thin_upd", "k_fwd_thin_upd (solve)", True),
    ("k_xinv_gemm", "k_xinv_gemm (block inversion)", False),
]


def read_pass(name):
    files = glob.glob(f"{work}/{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    if not files:
        return agg, ndisp
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[k].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
    return agg, ndisp


def durations(stats_dir):
    files = glob.glob(f"{work}/{stats_dir}/**/*kernel_stats.csv", recursive=True)
    d = {}
    if files:
        for row in csv.DictReader(open(files[0])):
            d[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
    return d


passes = {p: read_pass(p) for p in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write")}
dur = durations("solve_stats")
dur_bench = durations("bench_stats")
summary = {"round": rnd, "command": "python3 scripts/solve_profile.py S-metric 3 1 (one factorisation + 3 solves of the metric workload) under rocprofv3 --kernel-trace --pmc <one counter set per pass>",
           "kernels": []}
for sub, label, wide in KERNELS:
    rec = {"kernel": label, "fetch_doubled": wide}
    for pname, (agg, nd) in passes.items():
        n_pass = 0          # a kernel family may have several instantiations (k_big_syrk<0, 0, 64> and <0, 0, 128>): all of them count
        for k, ctr in agg.items():
            if sub in k:
                n_pass += len(nd[k])
                for c, v in ctr.items():
                    rec[c] = rec.get(c, 0.0) + v
        if n_pass:
            rec["dispatches"] = max(rec.get("dispatches", 0), n_pass)
    if "dispatches" not in rec:
        continue
    n = rec["dispatches"]
    for table in (dur_bench if "syrk" in sub or "trsm" in sub or "diag" in sub or "assemble" in sub or "dataflow" in sub else dur, dur, dur_bench):
        hit = [v for k, v in table.items() if sub in k]
        if hit:
            rec["avg_duration_us_unprofiled"] = sum(c * a for c, a in hit) / max(sum(c for c, a in hit), 1) / 1e3
            break
    if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("GRBM_GUI_ACTIVE", 0) > 0:
        rec["mfma_util_pct"] = 100.0 * rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 * CUS * 4)
    if rec.get("SQ_WAVE_CYCLES", 0) > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in rec:
                rec[c + "_frac_of_wave_cycles"] = rec[c] / rec["SQ_WAVE_CYCLES"]
    if rec.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        rec["lds_bank_conflict_ratio"] = rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / rec["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
        f = rec.get("FETCH_SIZE", 0.0) * 1024.0 * (2.0 if wide else 1.0)
        w = rec.get("WRITE_SIZE", 0.0) * 1024.0
        rec["hbm_bytes_per_launch"] = (f + w) / n
        rec["hbm_read_bytes_per_launch"] = f / n
        rec["hbm_write_bytes_per_launch"] = w / n
        if "avg_duration_us_unprofiled" in rec:
            rec["hbm_GBps_at_unprofiled_duration"] = rec["hbm_bytes_per_launch"] / (rec["avg_duration_us_unprofiled"] * 1e-6) / 1e9
    summary["kernels"].append(rec)
json.dump(summary, open(f"{out}/{rnd}_pmc_summary.json", "w"), indent=1)
syrk = next((r for r in summary["kernels"] if r["kernel"].startswith("k_front_dataflow")), None) or next((r for r in summary["kernels"] if r["kernel"].startswith("k_big_syrk<0,0>")), None)
if syrk:
    rec = {"kernel": syrk["kernel"], "launches": syrk["dispatches"], "fetch_kib_total": syrk.get("FETCH_SIZE"), "write_kib_total": syrk.get("WRITE_SIZE"),
           "hbm_bytes_per_launch": syrk.get("hbm_bytes_per_launch"),
           "hbm_bytes_per_launch_raw": ((syrk.get("FETCH_SIZE", 0.0) + syrk.get("WRITE_SIZE", 0.0)) * 1024.0 / syrk["dispatches"]),
           "mfma_util_pct": syrk.get("mfma_util_pct"), "lds_bank_conflict_ratio": syrk.get("lds_bank_conflict_ratio"),
           "note": "hbm_bytes_per_launch doubles FETCH_SIZE (16-byte-per-lane C tile reads and LDS-DMA operand streams, MI355X guide); raw keeps it as reported"}
    json.dump(rec, open(f"{out}/{rnd}_dataflow_pmc.json" if "dataflow" in syrk["kernel"] else f"{out}/{rnd}_syrk_pmc.json", "w"), indent=1)
for r in summary["kernels"]:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "dispatches", "mfma_util_pct", "lds_bank_conflict_ratio", "hbm_bytes_per_launch", "hbm_GBps_at_unprofiled_duration", "avg_duration_us_unprofiled", "SQ_WAIT_ANY_frac_of_wave_cycles", "SQ_WAIT_INST_ANY_frac_of_wave_cycles", "SQ_ACTIVE_INST_ANY_frac_of_wave_cycles")})
