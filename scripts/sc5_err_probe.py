import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_full_size as T
from onephase_jl_amd import synth
errs = T._forward_errors(sys.argv[1] if len(sys.argv) > 1 else "S-C5", 8)
print("errors (oracle, hip):", errs)
h = T.hip_solver("symmetric"); prob = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "S-C5", seed=0)
K = synth.augmented_matrix(prob, delta=1e-8); h.analyze(K); st = h.stats()
print({k: st[k] for k in ("ordering_used", "max_front", "flops_exact", "nlevels", "top_separator", "amd_skipped", "flops_other", "analyze_seconds")})
