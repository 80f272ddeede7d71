import sys, time
sys.path.insert(0, "/root/repo")
from onephase_jl_amd import synth
from onephase_jl_amd.linear_system_solvers import linear_solver_HIP, initialize_b
for name in sys.argv[1:]:
    prob = synth.make_config(name, seed=0)
    K = synth.augmented_matrix(prob, delta=1e-8)
    h = linear_solver_HIP("symmetric", host_symbolic_only=1)
    initialize_b(h)
    t = time.perf_counter(); h.analyze(K); dt = time.perf_counter() - t
    st = h.stats()
    print(f"{name}: analyze {dt:.3f} s (library {st['analyze_seconds']:.3f}), ordering {st['ordering_used']}, flops {st['flops_exact']:.4g}, top separator {st['top_separator']}, amd_skipped {st['amd_skipped']}, other flops {st['flops_other']:.4g}")
