"""Host only: the subtree partition of a configuration for 2 / 4 / 8 parts (okkt_dist_info): share of the flops above the cut, flops per
part, the flops-model speed-up and the size of the contribution blocks / vectors a factorisation / solve would move across the cut."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, ".")
from onephase_jl_amd import synth, _lib as L
from onephase_jl_amd.linear_system_solvers import linear_solver_HIP, initialize_b, finalize_b
for name in (sys.argv[1:] or ["S-metric"]):
    prob = synth.make_config(name, seed=0)
    K = synth.augmented_matrix(prob, delta=1e-8)
    for P in (2, 4, 8):
        h = linear_solver_HIP("symmetric", host_symbolic_only=1); initialize_b(h)
        h.analyze(K)
        h._check(h._lib.okkt_dist_set_partition(h._h, P, 0), "set_partition")
        nb = np.zeros(1, dtype=np.int64); cb = np.zeros(1, dtype=np.int64); cv = np.zeros(1, dtype=np.int64)
        pf = np.zeros(P, dtype=np.float64); tf = np.zeros(1, dtype=np.float64)
        h._check(h._lib.okkt_dist_info(h._h, L.p_i64(cb), L.p_i64(cv), L.p_i64(nb), L.p_f64(pf), L.p_f64(tf)), "dist_info")
        tot = pf.sum() + tf[0]
        print(f"{name} P={P}: boundary fronts {nb[0]}, contribution blocks {cb[0] * 8 / 1e6:.1f} MB, vectors {cv[0] * 8 / 1e3:.1f} kB, top share {tf[0] / tot:.3f}, part flops {np.round(pf / tot, 3).tolist()}, flops-model speed-up {tot / (tf[0] + pf.max()):.2f}")
        finalize_b(h)
