#!/usr/bin/env python3
"""bench.py -- KKT factor+solve/sec (fp64) on synthetic KKT systems, MI355X.

A "step" is one pass of the hot path over one KKT system whose values are already resident in HBM:
numeric multifrontal LDL^T (delta-shifted augmented matrix, symbolic analysis amortised -- it is done
once per sparsity pattern) followed by one triangular solve.  Workload at N = 1: S-metric of SURVEY.md
section 8d (n = 40 000, m = 60 000, nnz(tril K) ~ 1.8e6, locality model w = 50, p_far = 1 %).

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run on 127.0.0.1,
before anything touches a GPU; under torchrun it just runs as one rank).  Every rank factors+solves its own KKT system
(different seed, same shape) -- replicas, weak scaling, no data-path collective; torch.distributed (RCCL) carries the
barrier and the max-over-ranks of the elapsed time.  Behind the timed region the same ranks also run the
strong-scaling case of BASELINE config 5: ONE block-angular system (S-C5) with its elimination-tree subtrees sharded over
the GPUs and an RCCL reduce of the parent-front contribution blocks; its numbers ride in config.sharded.

One JSON line on rank 0 with `roofline` (dominant kernel = k_front_dataflow, the persistent launch that factors the big fronts of
one level of the elimination tree -- FP64-MFMA tile tasks --, timed live with HIP events on the library's stream; plus `roofline.solve`, the HBM-bound triangular solves), `cpu_baseline` (the supernodal
multifrontal CPU port on all host cores and on one, beside the simplicial oracle) and `parity` (HIP against the CPU
oracle on BASELINE config 3 at full size).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix spec; measured ceilings: DESIGN.md


def live_traffic(config, kernel="k_front_dataflow"):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot
    share a pass -- MI355X guide, PMC slots) over a child process that factors the same workload twice.  FETCH_SIZE is doubled
    (the kernel's C tiles and operand streams are 16-byte-per-lane reads: the guide's gfx950 correction), WRITE_SIZE is exact;
    both are KiB.  The child is a separate program started after this process has finished its GPU work; every pass has a hard
    time limit.  Returns (bytes per launch or None, detail dict)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    tot, ndisp = {}, {}
    work = tempfile.mkdtemp(prefix="okkt_pmc_")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, ctr)
            cmd = ["timeout", "-s", "KILL", "150", "rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "scripts", "pmc_target.py"), config]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=200)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, {"error": f"no counter file from the {ctr} pass (rc {r.returncode})"}
            v, ids = 0.0, set()
            for row in csv.DictReader(open(files[0])):
                if kernel in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    v += float(row["Counter_Value"])
                    ids.add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
            if not ids:
                return None, {"error": f"kernel not found in the {ctr} pass"}
            tot[ctr], ndisp[ctr] = v, len(ids)
    except Exception as exc:
        return None, {"error": f"{type(exc).__name__}: {str(exc)[:120]}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    per_launch = (2.0 * tot["FETCH_SIZE"] * 1024.0) / ndisp["FETCH_SIZE"] + (tot["WRITE_SIZE"] * 1024.0) / ndisp["WRITE_SIZE"]
    return per_launch, {"source": "live rocprofv3 --pmc passes inside this bench run (two factorisations of the workload in a child process)",
                        "fetch_kib_total": tot["FETCH_SIZE"], "write_kib_total": tot["WRITE_SIZE"], "launches": ndisp["FETCH_SIZE"],
                        "raw_bytes_per_launch": (tot["FETCH_SIZE"] / ndisp["FETCH_SIZE"] + tot["WRITE_SIZE"] / ndisp["WRITE_SIZE"]) * 1024.0,
                        "correction": "FETCH_SIZE doubled (16-byte-per-lane reads on gfx950), WRITE_SIZE as reported"}


def max_over_ranks(elapsed, distributed, device="cuda"):
    """MAX over ranks of the timed region (the slowest rank defines the step time)."""
    if not distributed:
        return elapsed
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def units_for_rank(rank, world):
    """Replicas: every rank factors and solves its own KKT system (generator seed = rank)."""
    return {"seed": rank, "systems_per_step": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="S-metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="S-C3")
    ap.add_argument("--no-kkt-level", action="store_true", help="skip the (untimed-region) KKT-level breakdown in config.kkt_level")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed profiles/ summary instead of two live rocprofv3 passes")
    ap.add_argument("--mode", default="both", choices=["both", "replicas", "sharded"],
                    help="N > 1: replicas = one KKT system per rank (weak scaling: the JSON line's value); sharded = ONE system, "
                         "elimination-tree subtrees over the ranks with an RCCL reduce of the contribution blocks (strong scaling, "
                         "S-C5); both (default) = the replica line with the sharded numbers in config.sharded")
    ap.add_argument("--sharded-config", default="S-C5")
    ap.add_argument("--no-sharded-model", action="store_true",
                    help="skip config.sharded_model: the measured multi-GPU prediction for BASELINE config 5 (per-part phases timed on this one GPU)")
    ap.add_argument("--strict-sharded", action="store_true",
                    help="exit with code 3 (after printing the metric line) when the sharded leg fails or times out; without it the failure only shows as config.sharded.error")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start the N ranks as a CHILD (this process has not touched the GPU and never will) and pass
        # its exit code on; rank 0 of the child prints the JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the KKT path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from onephase_jl_amd import synth
    from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

    if args.mode == "sharded" and distributed:
        out = bench_sharded(args, rank, world, local_rank, args.config if args.config != "S-metric" else args.sharded_config)
        if rank == 0:
            print(json.dumps(out))
        dist.destroy_process_group()
        return
    prob = synth.make_config(args.config, seed=units_for_rank(rank, world)["seed"])
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    hip = linear_solver_HIP("symmetric", device=local_rank)
    initialize_b(hip)
    t0 = time.time()
    hip.analyze(K)
    t_analyze = time.time() - t0
    d_vals = hip.dev_upload(K.data)
    rhs = np.random.default_rng(1234 + rank).normal(size=n + m)
    d_rhs = hip.dev_upload(rhs)
    d_sol = hip.dev_alloc(8 * (n + m))

    def step():
        rc = hip.ls_factor_dev(d_vals, n, m)
        hip.ls_solve_dev(d_rhs, d_sol)
        return rc

    for _ in range(args.warmup):
        rc = step()
    hip.profile_dominant(True)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    fac_ms = sol_ms = 0.0
    for _ in range(args.steps):
        rc = step()
        st = hip.stats()
        fac_ms += st["last_factor_ms"]
        sol_ms += st["last_solve_ms"]
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, distributed)
    nlaunch, syrk_ms, syrk_flops = hip.get_profile()
    hip.profile_dominant(False)
    # behind the timed region: the same solve with four right-hand sides in one batch (one pass over L for all four)
    d_rhs4 = hip.dev_upload(np.tile(rhs, (4, 1)))
    d_sol4 = hip.dev_alloc(8 * 4 * (n + m))
    hip.ls_solve_dev(d_rhs4, d_sol4, 4)
    hip.ls_solve_dev(d_rhs4, d_sol4, 4)
    solve4_ms = hip.stats()["last_solve_ms"]
    hip.dev_free(d_rhs4); hip.dev_free(d_sol4)

    # correctness of what was timed: inertia flag and residual of the last solve
    x = hip.dev_download(d_sol, (n + m,))
    M = synth.symmetrize_lower(K)
    resid = float(np.max(np.abs(M @ x - rhs)) / np.max(np.abs(rhs)))
    st = hip.stats()
    ok = (rc == 1) and hip.inertia == (n, m, 0, 0) and resid < 2e-6      # the run reaches 4.6e-7 (oracle: 1.8e-7): a regression of 5 x fails the line

    if rank == 0:
        value = world * args.steps / elapsed
        achieved = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes of this same command
        # (scripts/profile_bench.sh; FETCH_SIZE doubled as the MI355X guide prescribes for 16-byte-per-lane streams):
        # counters cannot be read from inside the timed process, so the committed summary of the latest round is quoted
        traffic = None
        for name in ("r06_dataflow_pmc.json", "r05_dataflow_pmc.json", "r04_dataflow_pmc.json",):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    break
                except Exception:
                    traffic = None
        solve_ms = sol_ms / args.steps
        solve_bytes = 2 * 8 * st["nnzL_stored"] + 2 * 4 * st["sum_rowidx"] + 4 * 8 * (n + m)
        out = {
            "metric": "KKT factor+solve/sec (fp64) at n+m~1e5, nnz~2e6",
            "value": value,
            "unit": "factor+solve/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: augmented KKT [[H+dI, J'],[J, -S/Y]], n={n}, m={m}, nnz(tril K)={st['nnz_lower']}, "
                            f"locality w=50, p_far=1%, delta=1e-8; per step: numeric LDL^T (symbolic amortised) + 1 solve",
                "n": n, "m": m, "nnz_tril": st["nnz_lower"], "nnzL": st["nnzL"], "factor_flops": st["flops_exact"],
                "multi_gpu": "replicas (one KKT system per rank)" if world > 1 else "single",
                "factor_ms": fac_ms / args.steps, "solve_ms": sol_ms / args.steps, "analyze_s_once": t_analyze,
                # the reference re-runs CHOLMOD's analysis in every ls_factor! (julia.jl:34,52): the same unit with the
                # library's own host analysis (AMD ordering + symbolic factorisation, okkt_analyze) charged to every call
                "analyse_every_call": {"value": world * args.steps / (elapsed + args.steps * t_analyze), "unit": "factor+solve/s",
                                       "analyze_s": t_analyze},
                "inertia_ok": bool(ok), "residual_inf": resid,
            },
            "roofline": {
                "kernel": "k_front_dataflow (one persistent launch per level of big fronts: diagonal-block, panel-tile and FP64-MFMA update tasks on 128 x 128 tiles, csrc/dataflow.hip); algorithmic flops = the dense partial LDL^T of the level's fronts, k f^2 - k^2 f + k^3 / 3 each",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "launches": nlaunch,
                "avg_launch_ms": syrk_ms / max(nlaunch, 1),
                "algorithmic_flops_per_launch": syrk_flops / max(nlaunch, 1),
                "share_of_factor_time": syrk_ms / max(fac_ms, 1e-9),
                # the second kernel family of the step: the triangular solves, HBM-bound (L streamed once forward, once
                # backward); achieved = algorithmic bytes / device time of the solve (HIP events on the library's stream)
                "solve": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "bytes": solve_bytes, "ms": solve_ms,
                          "achieved": solve_bytes / (solve_ms * 1e-3) / 1e9, "frac": solve_bytes / (solve_ms * 1e-3) / 1e9 / 8000.0,
                          "batch4": {"ms": solve4_ms, "ms_per_rhs": solve4_ms / 4, "achieved": 4 * solve_bytes / (solve4_ms * 1e-3) / 1e9,
                                     "note": "four right-hand sides through one pass over L (okkt_solve with nrhs = 4)"}},
            },
        }
    # the handle goes back before the KKT-level handles are created: they reuse its pooled stream set (api.cpp)
    hip.dev_free(d_vals); hip.dev_free(d_rhs); hip.dev_free(d_sol)
    inertia_final = hip.inertia
    perm_metric = hip.perm()
    gpu_metric = {"d": hip.diag(), "inertia": tuple(hip.inertia), "rhs": rhs, "x": x} if rank == 0 else None
    finalize_b(hip)
    if rank == 0:
        if world == 1 and not args.no_live_pmc and "ROCPROFILER_REGISTER_LIBRARY" not in os.environ and "ROCP_TOOL_LIBRARIES" not in os.environ:
            # HBM traffic of the dominant kernel measured in this run (not when this process is itself being profiled)
            live, detail = live_traffic(args.config)
            if live is not None:
                out["roofline"]["traffic"] = live
            detail.setdefault("source", "committed summary profiles/r06_dataflow_pmc.json (live passes failed)")
            out["roofline"]["traffic_detail"] = detail
        else:
            out["roofline"]["traffic_detail"] = {"source": "committed summary profiles/r06_dataflow_pmc.json (scripts/profile_r06.sh)"}
        if world == 1 and not args.no_kkt_level:
            out["config"]["kkt_level"] = kkt_level_breakdown(prob, local_rank)
        if world == 1 and not args.no_sharded_model:
            # the strong-scaling case (BASELINE config 5) predicted from MEASURED components: every part's local phases and the
            # top of the tree timed alone on this GPU (scripts/sharded_model.py), not from flops
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                from sharded_model import measure
                out["config"]["sharded_model"] = measure(args.sharded_config, reps=3, device=local_rank)
                sm = out["config"]["sharded_model"]
                # ... and its summary as a top-level key (a prediction from measured phases on ONE GPU, not a multi-GPU measurement)
                out["sharded"] = {"config": sm.get("config"), "kind": "predicted from per-part phases measured on one GPU (scripts/sharded_model.py)",
                                  "one_gpu": sm.get("one_gpu"),
                                  "predicted": {p: {"factor_ms": v.get("predicted_factor_ms"), "solve_ms": v.get("predicted_solve_ms"), "speedup": v.get("predicted_speedup")}
                                                for p, v in sm.get("parts", {}).items()}}
            except Exception as exc:
                out["config"]["sharded_model"] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(args.cpu_sample, st, K, perm_metric, n, m, local_rank, gpu_metric)
            # BASELINE.md holds no published number for this metric (the reference publishes none), so vs_baseline stays null.  The
            # ratio against the CPU port timed beside the GPU in this run is reported under its own name: it is a stand-in for the
            # reference's CHOLMOD path, not the reference (advisor, round 3)
            if out["cpu_baseline"].get("value"):
                out["vs_cpu_port"] = {"ratio": out["value"] / out["cpu_baseline"]["value"],
                                      "note": "value / cpu_baseline.value: the GPU's steady-state rate over the repo's own OpenMP port (kind 'port', same "
                                              "permutation) on this host; not a speed-up over the reference's CHOLMOD path, which cannot run here"}
            if not args.no_kkt_level:
                out["cpu_baseline"]["step_side_port"] = step_side_port(prob)
    if distributed and args.mode == "both":
        # The sharded case rides behind the measured line and must never cost it: a watchdog prints the line without the
        # sharded numbers and ends the rank if the collectives of the sharded path do not come back (a rank that failed
        # would leave the others waiting in a reduce), and an exception is recorded instead of raised.
        import threading
        done = threading.Event()

        def watchdog():
            if done.wait(float(os.environ.get("OKKT_BENCH_SHARDED_TIMEOUT", "240"))):
                return
            if rank == 0:
                out["config"]["sharded"] = {"error": "sharded run did not finish within its time limit; replica numbers above are unaffected"}
                print(json.dumps(out), flush=True)
            os._exit((3 if args.strict_sharded else 0) if ok else 1)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            sh = bench_sharded(args, rank, world, local_rank, args.sharded_config)      # every rank takes part; rank 0 keeps the numbers
        except Exception as exc:   # recorded, not raised: the other ranks run into the watchdog
            sh = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
            if rank != 0:
                done.wait(float(os.environ.get("OKKT_BENCH_SHARDED_TIMEOUT", "240")))
        done.set()
        if rank == 0:
            out["config"]["sharded"] = sh
            # the strong-scaling numbers of the sharded case as top-level keys of the line (round-5 review): a SCALE record shows them
            # next to the replica metric without digging into config
            if isinstance(sh, dict) and "error" not in sh:
                out["sharded"] = {"config": args.sharded_config, "n_gpus": world, "factor_ms": sh.get("factor_ms"), "solve_ms": sh.get("solve_ms"),
                                  "ms_per_step": sh.get("ms_per_step"), "value": sh.get("value"), "unit": sh.get("unit"),
                                  "speedup_vs_1gpu": sh.get("speedup_vs_1gpu"), "single_gpu": sh.get("single_gpu")}
        if isinstance(sh, dict) and "error" in sh:   # the process group may be wedged: print and leave without tearing it down
            if rank == 0:
                print(json.dumps(out), flush=True)
            # the metric line is out; --strict-sharded (tests, CI) turns a failed or hung sharded leg into exit code 3
            os._exit((3 if args.strict_sharded else 0) if ok else 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"bench result failed its correctness check: rc={rc} inertia={inertia_final} resid={resid}")


def bench_sharded(args, rank, world, local_rank, config):
    """Strong scaling: ONE KKT system (BASELINE config 5, block-angular), subtrees of its elimination tree sharded over the
    ranks; RCCL reduce of the parent-front contribution blocks to part 0, broadcast of the separator solution.  Returns the
    record (rank 0) -- the same unit as the metric, ONE system per step."""
    import torch
    import torch.distributed as dist
    from onephase_jl_amd import synth
    from onephase_jl_amd.distributed import ShardedLinearSolver, TorchComm
    prob = synth.make_config(config, seed=0)            # the same system on every rank
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    dev = torch.device("cuda", local_rank)
    rhs = np.random.default_rng(1234).normal(size=n + m)
    steps, warm = max(args.steps, 5), max(args.warmup, 2)
    # transport: the collectives inside the library on RCCL (okkt_dist_factor / okkt_dist_solve, what a Julia host would
    # call; default) or the phase-by-phase C ABI with torch.distributed collectives (OKKT_DIST_TRANSPORT=torch)
    transport = os.environ.get("OKKT_DIST_TRANSPORT", "abi")
    if transport == "abi":
        from onephase_jl_amd.distributed import RcclShardedLinearSolver
        sh = RcclShardedLinearSolver(rank, world, "symmetric", device=local_rank)
        info = sh.analyze(K)
        s0 = sh.solver
        dv, dr, dsol = s0.dev_upload(K.data), s0.dev_upload(rhs), s0.dev_alloc(8 * (n + m))
        def one():
            f = sh.factor(dv, n, m)
            sh.solve(dr, dsol)
            return f
        fetch = lambda: s0.dev_download(dsol, (n + m,))
    else:
        sh = ShardedLinearSolver(TorchComm(device=dev), "symmetric", device=local_rank)
        info = sh.analyze(K)
        s0 = sh.solvers[0]
        d_vals = [s0.dev_upload(K.data)]
        d_rhs = [s0.dev_upload(rhs)]
        last = {}
        def one():
            f = sh.factor(d_vals, n, m)
            last["x"] = sh.solve(d_rhs)
            return f
        fetch = lambda: last["x"]
    for _ in range(warm):
        flag = one()
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        flag = one()
    torch.cuda.synchronize(); dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, True)
    x = fetch()
    st_last = s0.stats()
    # the same system on ONE GPU (rank 0, an ordinary handle, behind the sharded run): what the strong-scaling number is a speed-up of
    single = None
    if rank == 0:
        try:
            from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP
            h1 = linear_solver_HIP("symmetric", device=local_rank)
            initialize_b(h1)
            h1.analyze(K)
            dv1, dr1, ds1 = h1.dev_upload(K.data), h1.dev_upload(rhs), h1.dev_alloc(8 * (n + m))
            for _ in range(warm):
                h1.ls_factor_dev(dv1, n, m); h1.ls_solve_dev(dr1, ds1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                h1.ls_factor_dev(dv1, n, m); h1.ls_solve_dev(dr1, ds1)
            torch.cuda.synchronize()
            single = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / steps, "factor_ms": h1.stats()["last_factor_ms"], "solve_ms": h1.stats()["last_solve_ms"]}
            h1.dev_free(dv1); h1.dev_free(dr1); h1.dev_free(ds1)
            finalize_b(h1)
        except Exception as exc:
            single = {"error": f"{type(exc).__name__}: {str(exc)[:120]}"}
    out = None
    if rank == 0:
        M = synth.symmetrize_lower(K)
        resid = float(np.max(np.abs(M @ x - rhs)) / np.max(np.abs(rhs)))
        st = s0.stats()
        total = sum(info["part_flops"]) + info["top_flops"]
        out = {"metric": "KKT factor+solve/sec (fp64), ONE system sharded over the GPUs", "value": steps / elapsed, "unit": "factor+solve/s",
               "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "factor_ms": st_last["last_factor_ms"], "solve_ms": st_last["last_solve_ms"], "single_gpu": single,
               "speedup_vs_1gpu": (single["ms_per_step"] / (1e3 * elapsed / steps)) if single and "ms_per_step" in single else None,
               "config": {"workload": f"{config}: n={n}, m={m}, nnz(tril K)={st['nnz_lower']}, subtree-sharded multifrontal LDL^T + solve", "inertia_flag": flag,
                          "inertia": list(sh.inertia), "residual_inf": resid, "top_flops_share": info["top_flops"] / total,
                          "part_flops": info["part_flops"], "model_speedup": total / (info["top_flops"] + max(info["part_flops"])),
                          "transport": "RCCL inside the library (okkt_dist_factor / okkt_dist_solve)" if transport == "abi" else "torch.distributed collectives between C-ABI phases",
                          "exchange_MB_per_factor": info["cb_doubles"] * 8 / 1e6}}
    sh.finalize()
    return out


def kkt_level_breakdown(prob, device):
    """Outside the timed region: wall-clock of the four reference methods of the device-resident KKT solver
    (form_system! -> factor! -> kkt_associate_rhs! -> compute_direction!, kkt_system_solver.jl:13-17) on the same
    workload, for both system shapes.  Host buffers cross PCIe in form_system (H, J, s, y) and in the rhs/direction
    vectors, so these are PCIe-inclusive figures; they are reported beside `value`, never as it."""
    from onephase_jl_amd import kkt_system_solver as KS
    n, m = prob["n"], prob["m"]
    rng = np.random.default_rng(1)
    it = KS.Class_iterate(x=rng.normal(size=n), y=prob["y"], s=prob["s"], mu=float(prob["mu"]), J=prob["J"], H=prob["H"],
                          grad=rng.normal(size=n), cons=prob["s"] + 1e-3 * rng.normal(size=m))
    out = {}
    for kind in ("symmetric", "schur"):
        pars = KS.Class_parameters()
        pars.kkt.kkt_solver_type = kind
        k = KS.HIP_KKT_solver(kind, pars, device=device)
        k.initialize_b(it)
        k.form_system_b(it)                      # first call: symbolic analysis of this shape
        tm = {}
        for _ in range(2):
            t = time.perf_counter(); k.form_system_b(it); tm["form_system_ms"] = 1e3 * (time.perf_counter() - t)
            t = time.perf_counter(); inertia = k.factor_b(1e-8); tm["factor_ms"] = 1e3 * (time.perf_counter() - t)
            t = time.perf_counter(); k.kkt_associate_rhs_b(it, KS.Reduct_affine()); tm["rhs_ms"] = 1e3 * (time.perf_counter() - t)
            t = time.perf_counter(); k.compute_direction_b(); tm["direction_ms"] = 1e3 * (time.perf_counter() - t)
        tm["inertia_flag"] = int(inertia)
        tm["N_err"] = float(k.kkt_err_norm.ratio)
        tm["refinement_solves"] = 3 if kind == "schur" else 1
        # device times of the phases (HIP events on the handle's stream, okkt_kkt_get_timers) and the HBM rate they imply
        # against SURVEY.md 8d's algorithmic byte counts
        dev = k.timers()
        nnzJ, nnzH = it.J.nnz, it.H.nnz
        nnzA = k.linear_solver_stats()["nnz_lower"]
        if kind == "symmetric":
            asm_bytes = 8 * (nnzH + nnzJ + 2 * m) + 8 * nnzA + 4 * (nnzH + nnzJ)          # values in, K out, the slot map
        else:
            terms = int(np.sum(np.diff(sp_csr_indptr(it.J)) * (np.diff(sp_csr_indptr(it.J)) + 1) // 2))
            asm_bytes = 12 * nnzJ + 8 * m + 8 * nnzA + 12 * terms                        # 8d + the per-term lists this kernel reads
        resid_bytes = 2 * 12 * nnzJ + 2 * 12 * nnzH + 8 * (3 * n + 2 * m)                # one refinement residual: J dx, J' v, H dx
        dev["assemble_GBps"] = asm_bytes / max(dev["assemble_ms"], 1e-9) / 1e6
        if kind == "schur" and dev["refine_ms"] > 0:
            dev["refine_GBps"] = (2 * resid_bytes + 12 * nnzJ * 2) / dev["refine_ms"] / 1e6   # two residuals + rhs and dy / ds products
        dev["kkt_err_GBps"] = (2 * 12 * nnzJ + 2 * 12 * nnzH + 8 * (4 * n + 8 * m)) / max(dev["kkt_err_ms"], 1e-9) / 1e6
        tm["device"] = dev
        out[kind] = tm
        if kind == "symmetric":
            out["step_side"] = step_side_breakdown(k, it)
        k.finalize_b()
    # the delta-loop variant: a nonconvex instance of the same generator (H shifted by -5), the whole
    # ipopt_strategy! (delta_strategy.jl:37-114) + one direction = what one outer IPM iteration costs on the path
    from onephase_jl_amd import synth
    cfg = dict(synth.CONFIGS["S-metric"]) if n + m == 100000 else None
    if cfg is not None:
        cfg.update(convex=False, neg_shift=5.0)
        pnc = synth.make_problem(seed=0, **cfg)
        itn = KS.Class_iterate(x=it.x, y=pnc["y"], s=pnc["s"], mu=float(pnc["mu"]), J=pnc["J"], H=pnc["H"], grad=it.grad,
                               cons=pnc["s"] + 1e-3 * rng.normal(size=m))
        pars = KS.Class_parameters()
        pars.kkt.kkt_solver_type = "symmetric"
        k = KS.HIP_KKT_solver("symmetric", pars, device=device)
        k.initialize_b(itn)
        k.form_system_b(itn)
        k.ipopt_strategy_b(itn)                                   # warm-up incl. the symbolic analysis
        k.form_system_b(itn)
        t = time.perf_counter(); status, nfac, delta = k.ipopt_strategy_b(itn); t_loop = 1e3 * (time.perf_counter() - t)
        k.kkt_associate_rhs_b(itn, KS.Reduct_affine())
        t = time.perf_counter(); k.compute_direction_b(); t_dir = 1e3 * (time.perf_counter() - t)
        out["delta_loop"] = {"workload": "S-metric generator, nonconvex (neg_shift = 5), symmetric KKT", "status": status, "num_fac": int(nfac),
                             "delta": float(delta), "loop_ms": t_loop, "ms_per_factorisation": t_loop / max(nfac, 1),
                             "direction_ms": t_dir, "N_err": float(k.kkt_err_norm.ratio)}
        k.finalize_b()
    return out


def sp_csr_indptr(J):
    import scipy.sparse as sp
    return sp.csr_matrix(J).indptr


def step_side_breakdown(k, it):
    """wall-clock of the step-side functions (line_search.jl:40-41,84-86; move.jl:15-17,82-118;
    eval.jl:236-273) against the direction that compute_direction_b left on the device.  Host vectors (frac_bd, the
    candidate's s, y, grad) cross PCIe inside each call; `stream_bytes` = the length-m / length-n vectors and the
    J, H entries a call reads on the device (8 B values + 4 B indices), for the HBM figure."""
    from onephase_jl_amd import line_search as LS
    n, m = it.dim(), it.ncon()
    nnzJ, nnzH = it.J.nnz, it.H.nnz
    rng = np.random.default_rng(2)
    fp, fb = np.full(m, 0.2), np.full(m, 0.1)
    pars = LS.Class_ls_parameters()
    out = {"workload": f"n = {n}, m = {m}, nnz(J) = {nnzJ}, nnz(tril H) = {nnzH}; direction resident"}
    step_P, _ = LS.max_step_primal(k, fp, pars)
    alpha = 0.5 * step_P
    d = k.dir
    import copy
    cand = copy.copy(it)
    cand.s, cand.mu, cand.grad = it.s + alpha * d.s, it.mu + alpha * d.mu, it.grad + 1e-3 * rng.normal(size=n)
    lb, ub = LS.dual_step_range(k, cand, fb, pars)
    calls = {
        "max_step_primal": (lambda: LS.max_step_primal(k, fp, pars), 8 * (n + 4 * m)),
        "s_bound_ok": (lambda: LS.s_bound_ok(k, cand.s, fb, pars), 8 * (n + 3 * m)),
        "dual_step_range": (lambda: LS.dual_step_range(k, cand, fb, pars), 8 * (n + 5 * m)),
        "predicted_reduction": (lambda: LS.predicted_reduction_terms(k, 1.0), 12 * (2 * nnzJ + 2 * nnzH) + 8 * (8 * m + 7 * n)),
        "dual_step": (lambda: LS.move_dual_step(k, cand, alpha, lb, ub, 1.0, 1.0, pars), 12 * 2 * nnzJ + 8 * (6 * m + 6 * n)),
    }
    for name, (fn, nbytes) in calls.items():
        fn()
        t = time.perf_counter()
        for _ in range(5):
            fn()
        ms = 1e3 * (time.perf_counter() - t) / 5
        out[name] = {"ms": ms, "stream_bytes": int(nbytes)}
    out["step_size_P"], out["dual_range"] = float(step_P), [float(lb), float(ub)]
    return out


def step_side_port(prob):
    """cpu_baseline leg: the oracle's numpy / Python restatement of the same step-side functions on the same sizes."""
    from oracle import kkt_oracle as KO
    from oracle import line_search_oracle as LO
    n, m = prob["n"], prob["m"]
    rng = np.random.default_rng(1)
    it = KO.Iterate(x=rng.normal(size=n), y=prob["y"], s=prob["s"], mu=float(prob["mu"]), J=prob["J"], H=prob["H"],
                    grad=rng.normal(size=n), cons=prob["s"].copy())
    d = KO.Direction(rng.normal(size=n), 0.1 * rng.normal(size=m) * prob["y"], 0.1 * rng.normal(size=m) * prob["s"], mu=-0.5 * it.mu)
    fp, fb = np.full(m, 0.2), np.full(m, 0.1)
    out = {}
    t = time.perf_counter(); step_P = LO.simple_max_step(it.s, d.s, LO.lb_s_predict(it, d, fp, 0.5)); out["max_step_primal_ms"] = 1e3 * (time.perf_counter() - t)
    s_c = it.s + 0.5 * step_P * d.s
    t = time.perf_counter(); lb, ub = LO.dual_step_range(it, d, s_c, it.y, 0.75 * it.mu, 0.01, fb); out["dual_step_range_ms"] = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter(); LO.predicted_reduction_terms(it, d, 1.0); out["predicted_reduction_ms"] = 1e3 * (time.perf_counter() - t)
    cand = KO.Iterate(x=it.x, y=it.y, s=s_c, mu=0.75 * it.mu, J=it.J, H=it.H, grad=it.grad, cons=it.cons)
    t = time.perf_counter(); LO.move_dual_step(cand, d, 0.5 * step_P, lb, ub, 1, 1.0, 1.0); out["dual_step_ms"] = 1e3 * (time.perf_counter() - t)
    out["note"] = "numpy, 1 thread; dual_bounds is the reference's scalar loop in Python"
    return out


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(sample_cfg, st_metric, K_metric, perm_metric, n_metric, m_metric, device, gpu_metric=None):
    """cpu_baseline + parity legs (outside the timed region, rank 0 at N = 1).

    CPU side: the supernodal multifrontal port (oracle/okkt_oracle_mf.c, the algorithm class of
    CHOLMOD's supernodal numeric phase) on ONE core and on ALL host cores, and the simplicial up-looking oracle
    (oracle/okkt_oracle.c, what `ldlt` runs) on one core; each in analyse-once and analyse-every-call form.  The bounded
    sample is one factor+solve of `sample_cfg` (BASELINE config 3); the all-core port also runs the metric workload itself
    when that fits in ~45 s of CPU time, otherwise its S-C3 rate is scaled by the factor-flop ratio.
    Parity: the HIP path against the simplicial oracle on the full-size sample, same permutation."""
    import oracle
    from onephase_jl_amd import synth
    from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP
    prob = synth.make_config(sample_cfg, seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    b = np.random.default_rng(0).normal(size=n + m)
    h = linear_solver_HIP("symmetric", device=device)
    initialize_b(h)
    rc_gpu = h.ls_factor_b(K, n, m)
    perm = h.perm()          # same fill-reducing permutation as the GPU path
    st = h.stats()
    x_gpu, d_gpu, inertia_gpu = h.ls_solve(b), h.diag(), h.inertia
    finalize_b(h)
    ratio = st_metric["flops_exact"] / st["flops_exact"]
    cores = host_cores()

    def timed(solver):
        t0 = time.perf_counter(); solver._analyze(K); t_sym = time.perf_counter() - t0
        t0 = time.perf_counter(); rc = solver.ls_factor_b(K, n, m); x = solver.ls_solve(b); dt = time.perf_counter() - t0
        return rc, x, dt, t_sym

    # ---- simplicial oracle, one core: the parity reference
    ref = oracle.linear_solver_ORACLE("symmetric", perm=perm)
    rc, x, dt_simp, tsym_simp = timed(ref)
    d_ref = ref.diag()
    parity = {
        "workload": f"{sample_cfg} at full size (n={n}, m={m}), HIP path vs oracle/okkt_oracle.c with the same permutation",
        "rel_err_x": float(np.max(np.abs(x_gpu - x)) / np.max(np.abs(x))),
        "inertia_equal": bool(tuple(inertia_gpu[:3]) == tuple(ref.inertia(1e-20)[:3]) and rc_gpu == rc),
        "sign_D_equal": bool(np.array_equal(np.sign(d_gpu), np.sign(d_ref))),
        "max_rel_err_D": float(np.max(np.abs(d_gpu - d_ref) / np.abs(d_ref))),
        "tolerance_x": 1e-8,
    }
    # ---- multifrontal port: one core, all cores
    mf1 = oracle.linear_solver_ORACLE_MF("symmetric", perm=perm, nthreads=1)
    rc1, x1, dt_mf1, tsym_mf = timed(mf1)
    # "all host cores": OpenMP tasks over the tree and inside the large fronts; on a many-core host the task queue and the
    # serial top of the tree can make fewer threads faster, so the sample picks the best of {all, 64, 32} and says so
    dt_mfp, used, tried = None, cores, {}
    for nth in sorted({cores, min(cores, 64), min(cores, 32)}, reverse=True):
        cand = oracle.linear_solver_ORACLE_MF("symmetric", perm=perm, nthreads=nth)
        cand._analyze(K)
        t0 = time.perf_counter(); rcq = cand.ls_factor_b(K, n, m); xq = cand.ls_solve(b); dtq = time.perf_counter() - t0
        tried[str(nth)] = dtq
        if dt_mfp is None or dtq < dt_mfp:
            dt_mfp, used, rcp, xp = dtq, nth, rcq, xq
    agree = bool(np.max(np.abs(xp - x)) <= 1e-7 * np.max(np.abs(x)) and rc1 == rcp == rc)
    # ---- the metric workload itself on all cores, when it fits the time budget
    est = dt_mfp * ratio
    direct = None
    if est <= 45.0 and K_metric is not None:
        mfm = oracle.linear_solver_ORACLE_MF("symmetric", perm=perm_metric, nthreads=used)
        bm = np.random.default_rng(1).normal(size=n_metric + m_metric)
        t0 = time.perf_counter(); mfm._analyze(K_metric); t_sym_m = time.perf_counter() - t0
        t0 = time.perf_counter(); rcm = mfm.ls_factor_b(K_metric, n_metric, m_metric); xm = mfm.ls_solve(bm); dt_m = time.perf_counter() - t0
        direct = {"seconds": dt_m, "analyze_seconds": t_sym_m, "rc": int(rcm), "gflops": st_metric["flops_exact"] / dt_m / 1e9}
        if gpu_metric is not None:
            # parity at the metric size itself (round-5 review, 5b): the factor the timed region produced against the CPU port's on the
            # same pivot order -- inertia counts, sign(D) entry by entry, D, and the timed solve's solution
            d_cpu = mfm.diag()
            x_cpu = mfm.ls_solve(gpu_metric["rhs"])
            parity["metric_size"] = {
                "workload": f"the metric workload (n={n_metric}, m={m_metric}), HIP path vs oracle/okkt_oracle_mf.c ({used} threads), same permutation",
                "inertia_equal": bool(tuple(gpu_metric["inertia"][:3]) == tuple(mfm.inertia()[:3]) == (n_metric, m_metric, 0)),
                "sign_D_equal": bool(np.array_equal(np.sign(gpu_metric["d"]), np.sign(d_cpu))),
                "max_rel_err_D": float(np.max(np.abs(gpu_metric["d"] - d_cpu) / np.abs(d_cpu))),
                "rel_err_x": float(np.max(np.abs(gpu_metric["x"] - x_cpu)) / np.max(np.abs(x_cpu))),
                "tolerance_D": 1e-7, "tolerance_x": 1e-7,
            }
        del mfm
    value = 1.0 / direct["seconds"] if direct else 1.0 / est
    # independent datapoint: SuperLU through scipy on the same sample, its own ordering, analysis included
    try:
        import scipy.sparse.linalg as spla
        M = synth.symmetrize_lower(K).tocsc()
        t1 = time.perf_counter()
        lu = spla.splu(M, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        xs = lu.solve(b)
        indep = {"what": "scipy.sparse.linalg.splu (SuperLU, MMD_AT_PLUS_A, symmetric mode), ordering + factor + solve",
                 "sample_seconds": time.perf_counter() - t1,
                 "agrees_with_port": bool(np.max(np.abs(xs - x)) <= 1e-6 * max(1.0, float(np.max(np.abs(x)))))}
    except Exception as exc:   # scipy is optional on the box
        indep = {"what": "scipy splu unavailable", "error": str(exc)[:80]}
    base = {
        "value": value,
        "unit": "factor+solve/s",
        "cores": used,
        "host_cores": cores,
        "threads_tried_sample_seconds": tried,
        "kind": "port",
        "sample": (f"supernodal multifrontal LDL^T port (oracle/okkt_oracle_mf.c, OpenMP), {used} threads on a {cores}-core host: "
                   + (f"the metric workload itself, one factor+solve in {direct['seconds']:.2f} s ({direct['gflops']:.0f} GFLOP/s), analysis once"
                      if direct else f"one factor+solve of {sample_cfg} in {dt_mfp:.2f} s, scaled by the factor-flop ratio {ratio:.1f}")),
        "metric_workload_direct": direct,
        "analyse_every_call": {"value": (1.0 / (direct["seconds"] + direct["analyze_seconds"])) if direct else 1.0 / ((dt_mfp + tsym_mf) * ratio),
                               "unit": "factor+solve/s",
                               "note": "elimination tree, column counts, supernodes and front structure redone in every call, as the reference's "
                                       "ls_factor! does (julia.jl:34,52); the fill-reducing ordering itself is the GPU path's and is not re-timed"},
        "sample_all_cores": {"config": sample_cfg, "seconds": dt_mfp, "gflops": st["flops_exact"] / dt_mfp / 1e9, "flop_ratio_to_metric": ratio},
        "single_core": {"value": 1.0 / (dt_mf1 * ratio), "unit": "factor+solve/s", "cores": 1,
                        "sample": f"the same port on 1 core: {sample_cfg} in {dt_mf1:.2f} s ({st['flops_exact'] / dt_mf1 / 1e9:.1f} GFLOP/s), scaled by {ratio:.1f} "
                                  "(the reference's published condition is one core, docs/one-phase.tex:930)",
                        "analyse_every_call": 1.0 / ((dt_mf1 + tsym_mf) * ratio)},
        "simplicial_oracle_1core": {"value": 1.0 / (dt_simp * ratio), "unit": "factor+solve/s", "cores": 1,
                                    "sample": f"oracle/okkt_oracle.c (up-looking LDL^T, what CHOLMOD's ldlt runs): {sample_cfg} in {dt_simp:.2f} s "
                                              f"({st['flops_exact'] / dt_simp / 1e9:.1f} GFLOP/s), scaled by {ratio:.1f}",
                                    "analyse_every_call": 1.0 / ((dt_simp + tsym_simp) * ratio)},
        "ports_agree": agree,
        "independent": indep,
    }
    return base, parity


if __name__ == "__main__":
    main()
