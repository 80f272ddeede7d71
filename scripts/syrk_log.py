"""Per-launch rate of the dominant kernel over one factorisation (GPU box): OKKT_DEBUG_SYRK_LOG=1 python scripts/syrk_log.py S-metric"""
import os, sys
os.environ["OKKT_DEBUG_SYRK_LOG"] = "1"
sys.path.insert(0, ".")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import initialize_b, finalize_b, linear_solver_HIP

name = sys.argv[1] if len(sys.argv) > 1 else "S-metric"
prob = synth.make_config(name, seed=0)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-8)
h = linear_solver_HIP("symmetric"); initialize_b(h)
assert h.ls_factor_b(K, n, m) == 1
h.profile_dominant(True)
assert h.ls_factor_b(K, n, m) == 1
print(h.get_profile(), h.stats()["last_factor_ms"])
finalize_b(h)
