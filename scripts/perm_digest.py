"""Digest of what the analysis decides, per configuration: permutation, elimination tree and the structural statistics.  A change
of the ordering / symbolic code that is meant to be time-only must leave every line unchanged (also under OKKT_ANALYZE_THREADS=1, 3, 64).
usage: python scripts/perm_digest.py S-metric S-C3 ..."""
import sys, hashlib
import numpy as np
sys.path.insert(0, "/root/repo")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import linear_solver_HIP, initialize_b
for name in sys.argv[1:]:
    for seed in (0, 1):
        prob = synth.make_config(name, seed=seed)
        K = synth.augmented_matrix(prob, delta=1e-8)
        h = linear_solver_HIP("symmetric", host_symbolic_only=1)
        initialize_b(h)
        h.analyze(K)
        st = h.stats()
        et = h.etree()
        et = b"".join(np.ascontiguousarray(a).tobytes() for a in (et if isinstance(et, (tuple, list)) else [et]))
        keys = ("nnzL", "nnzL_stored", "flops_stored", "arena_bytes", "nsuper", "nlevels", "max_front", "sum_rowidx", "critical_pivots")
        print(f"{name} seed {seed}: perm sha1 {hashlib.sha1(h.perm().tobytes()).hexdigest()[:16]} etree sha1 {hashlib.sha1(et).hexdigest()[:16]} ordering {st['ordering_used']} flops {st['flops_exact']:.6g} " + " ".join(f"{k}={st[k]}" for k in keys))
