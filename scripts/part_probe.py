"""Times okkt_dist_factor_local of single parts of a partitioned plan (debugging aid): python scripts/part_probe.py S-C5 8 1 2"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from onephase_jl_amd import synth, _lib as L
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP
cfg, P = sys.argv[1], int(sys.argv[2])
prob = synth.make_config(cfg, seed=0); n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
for r in [int(a) for a in sys.argv[3:]]:
    h = linear_solver_HIP("symmetric"); initialize_b(h); h.analyze(K)
    h._check(h._lib.okkt_dist_set_partition(h._h, P, r), "part")
    dv = h.dev_upload(K.data)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); h._check(h._lib.okkt_dist_factor_local(h._h, C.c_void_p(dv), n, m, L.OKKT_SYM_SYMMETRIC), "fl"); ts.append(1e3 * (time.perf_counter() - t))
    print("part", r, "factor_local ms", [round(x, 3) for x in ts], flush=True)
    finalize_b(h)
