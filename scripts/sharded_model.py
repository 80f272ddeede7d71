"""Measured multi-GPU prediction on ONE GPU (the pool hands out one-GPU boxes): the subtree-sharded factorisation + solve of a
configuration is run with P virtual ranks (LocalComm: the exact per-rank code path of the sharded solver, one handle per part, all
on this GPU, one after the other), every blocking phase is timed, and

    predicted_P_gpu_ms = max_p t(factor_local, p) + t(factor_top) + max_p t(solve_fwd_local, p) + t(solve_top)
                         + max_p t(solve_bwd_local, p) + n_collectives * 0.02 ms

stands beside the flops-only model.  Usage: python scripts/sharded_model.py [S-C5] [reps]; prints one JSON line."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.distributed import LocalComm, ShardedLinearSolver
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP


def measure(config="S-C5", reps=4, parts=(2, 4, 8), device=0):
    prob = synth.make_config(config, seed=0)
    n, m = prob["n"], prob["m"]
    K = synth.augmented_matrix(prob, delta=1e-8)
    rhs = np.random.default_rng(1234).normal(size=n + m)
    ref = linear_solver_HIP("symmetric", device=device); initialize_b(ref)
    ref.analyze(K)
    dv, dr, ds = ref.dev_upload(K.data), ref.dev_upload(rhs), ref.dev_alloc(8 * (n + m))
    f1, s1 = [], []
    for _ in range(reps + 1):
        assert ref.ls_factor_dev(dv, n, m) == 1
        ref.ls_solve_dev(dr, ds)
        st = ref.stats(); f1.append(st["last_factor_ms"]); s1.append(st["last_solve_ms"])
    x_ref = ref.dev_download(ds, (n + m,))
    one = {"factor_ms": float(np.min(f1[1:])), "solve_ms": float(np.min(s1[1:]))}
    finalize_b(ref)
    out = {"config": config, "n": n, "m": m, "one_gpu": one, "parts": {}}
    for P in parts:
        sh = ShardedLinearSolver(LocalComm(P), "symmetric", device=device)
        info = sh.analyze(K)
        d_vals = [s.dev_upload(K.data) for s in sh.solvers]
        d_rhs = [s.dev_upload(rhs) for s in sh.solvers]
        T = {}
        for r in range(reps + 1):
            tm = {} if r else None          # first round: warm-up (device plans, kernels)
            assert sh.factor(d_vals, n, m, timings=tm) == 1
            x = sh.solve(d_rhs, timings=tm)
            if tm:
                for key, per in tm.items():
                    for rank, v in per.items():
                        T.setdefault(key, {}).setdefault(rank, []).append(v[0])
        err = float(np.max(np.abs(x - x_ref)) / np.max(np.abs(x_ref)))
        best = lambda key: {rank: float(np.min(v)) for rank, v in T.get(key, {}).items()}
        ft, stp = best("factor_top"), best("solve_top")
        sh.finalize()
        # the local phases once more with ONE handle alive at a time: P handles x 4 streams on one GPU share its hardware queues
        # (a part measured 2-3x slower than its siblings in the all-handles run above), a real rank has the GPU to itself
        import ctypes as C
        from onephase_jl_amd import _lib as L
        fl, sf, sb = {}, {}, {}
        import time
        for r in range(P):
            h = linear_solver_HIP("symmetric", device=device); initialize_b(h)
            h.analyze(K)
            h._check(h._lib.okkt_dist_set_partition(h._h, P, r), "okkt_dist_set_partition")
            dvr, drr, xz = h.dev_upload(K.data), h.dev_upload(rhs), h.dev_upload(np.zeros(n + m))
            tf, t1, t2 = [], [], []
            for rep in range(reps + 1):
                t = time.perf_counter(); h._check(h._lib.okkt_dist_factor_local(h._h, C.c_void_p(dvr), n, m, L.OKKT_SYM_SYMMETRIC), "factor_local"); tf.append(1e3 * (time.perf_counter() - t))
                cnt = np.zeros(4, dtype=np.int64)
                h._check(h._lib.okkt_dist_counts(h._h, L.p_i64(cnt)), "counts")
                tot = np.array([n, m, 0, 0], dtype=np.int64)
                h._check(h._lib.okkt_dist_finish(h._h, L.p_i64(tot)), "finish")        # marks the handle factored (the counts of the other parts are not needed for timing)
                t = time.perf_counter(); h._check(h._lib.okkt_dist_solve_begin(h._h, C.c_void_p(drr)), "solve_begin"); t1.append(1e3 * (time.perf_counter() - t))
                h._check(h._lib.okkt_dist_x(h._h, C.c_void_p(xz), 1), "dist_x")
                t = time.perf_counter(); h._check(h._lib.okkt_dist_solve_end(h._h), "solve_end"); t2.append(1e3 * (time.perf_counter() - t))
            fl[r], sf[r], sb[r] = float(np.min(tf[1:])), float(np.min(t1[1:])), float(np.min(t2[1:]))
            finalize_b(h)
        ncoll_f, ncoll_s = 2, 3             # reduce(cb) + all-reduce(counts); reduce(cv) + broadcast(x) + all-reduce(solution)
        pred_f = max(fl.values()) + sum(ft.values()) + ncoll_f * 0.02
        pred_s = max(sf.values()) + sum(stp.values()) + max(sb.values()) + ncoll_s * 0.02
        tot = sum(info["part_flops"]) + info["top_flops"]
        out["parts"][str(P)] = {
            "factor_local_ms": [fl[r] for r in sorted(fl)], "factor_top_ms": sum(ft.values()),
            "solve_fwd_local_ms": [sf[r] for r in sorted(sf)], "solve_top_ms": sum(stp.values()), "solve_bwd_local_ms": [sb[r] for r in sorted(sb)],
            "predicted_factor_ms": pred_f, "predicted_solve_ms": pred_s, "predicted_ms": pred_f + pred_s,
            "predicted_speedup": (one["factor_ms"] + one["solve_ms"]) / (pred_f + pred_s),
            "flops_model_speedup": tot / (info["top_flops"] + max(info["part_flops"])), "top_flops_share": info["top_flops"] / tot,
            "rel_err_vs_unsharded": err,
            "note": "each phase timed alone on one GPU (host clock around the blocking C-ABI call, best of the repetitions; local phases with one handle alive at a time); collectives priced at 20 us each"}
    return out


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "S-C5"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    print(json.dumps(measure(cfg, reps)))
