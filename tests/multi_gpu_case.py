"""Helper of test_gpu_multi.py, run under torch.distributed.run with one rank per GPU: the subtree-sharded factorisation and solve
(collectives inside the library on RCCL, and the torch.distributed transport) against the unsharded solver on every rank."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
import torch
import torch.distributed as dist

torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
from onephase_jl_amd import synth
from onephase_jl_amd.distributed import RcclShardedLinearSolver, ShardedLinearSolver, TorchComm
from onephase_jl_amd.linear_system_solvers import finalize_b, initialize_b, linear_solver_HIP

prob = synth.block_angular(nblocks=8, n_b=400, m_b=600, n_link=20, seed=3, j_per_row=5, h_per_col=3, w=8.0, p_far=0.0, well_scaled=True)
n, m = prob["n"], prob["m"]
K = synth.augmented_matrix(prob, delta=1e-7)
b = np.random.default_rng(9).normal(size=n + m)
ref = linear_solver_HIP("symmetric", device=local)
initialize_b(ref)
assert ref.ls_factor_b(K, n, m) == 1
x_ref = ref.ls_solve(b)

sh = RcclShardedLinearSolver(rank, world, "symmetric", device=local)
info = sh.analyze(K)
s = sh.solver
d_vals, d_rhs, d_sol = s.dev_upload(K.data), s.dev_upload(b), s.dev_alloc(8 * (n + m))
for _ in range(2):
    assert sh.factor(d_vals, n, m) == 1
    assert sh.inertia == ref.inertia
    sh.solve(d_rhs, d_sol)
x = s.dev_download(d_sol, (n + m,))
assert np.max(np.abs(x - x_ref)) <= 1e-10 * np.max(np.abs(x_ref)), np.max(np.abs(x - x_ref))
K2 = synth.augmented_matrix(prob, delta=-50.0)
assert sh.factor(s.dev_upload(K2.data), n, m) == ref.ls_factor_b(K2, n, m) == 0
assert sh.inertia == ref.inertia
sh.finalize()

st = ShardedLinearSolver(TorchComm(device=torch.device("cuda", local)), "symmetric", device=local)
st.analyze(K)
s0 = st.solvers[0]
assert st.factor([s0.dev_upload(K.data)], n, m) == 1
xt = st.solve([s0.dev_upload(b)])
if rank == 0:
    assert np.max(np.abs(xt - x_ref)) <= 1e-10 * np.max(np.abs(x_ref))
st.finalize()
finalize_b(ref)

# BASELINE config 5 at full size (S-C5: 8 blocks of 12 500 unknowns + 200 linking columns), RCCL inside the library: per-rank
# wall-clock of the sharded factorisation and solve, so that the first multi-GPU box yields the scaling curve with no further work
import time
prob5 = synth.make_config("S-C5", seed=0)
n5, m5 = prob5["n"], prob5["m"]
K5 = synth.augmented_matrix(prob5, delta=1e-8)
b5 = np.random.default_rng(10).normal(size=n5 + m5)
sh5 = RcclShardedLinearSolver(rank, world, "symmetric", device=local)
sh5.analyze(K5)
s5 = sh5.solver
dv5, dr5, ds5 = s5.dev_upload(K5.data), s5.dev_upload(b5), s5.dev_alloc(8 * (n5 + m5))
tf, ts = [], []
for rep in range(6):
    dist.barrier()
    t = time.perf_counter(); flag5 = sh5.factor(dv5, n5, m5); tf.append(1e3 * (time.perf_counter() - t))
    t = time.perf_counter(); sh5.solve(dr5, ds5); ts.append(1e3 * (time.perf_counter() - t))
assert flag5 == 1
x5 = s5.dev_download(ds5, (n5 + m5,))
M5 = synth.symmetrize_lower(K5)
res5 = float(np.max(np.abs(M5 @ x5 - b5)) / np.max(np.abs(b5)))
assert res5 < 1e-5, res5
print(f"MULTI_GPU_SC5 rank {rank} of {world}: factor {min(tf[1:]):.3f} ms, solve {min(ts[1:]):.3f} ms, residual {res5:.1e}", flush=True)
sh5.finalize()
dist.barrier()
dist.destroy_process_group()
print(f"MULTI_GPU_OK rank {rank} of {world}, top share {info['top_flops'] / (sum(info['part_flops']) + info['top_flops']):.3f}")
