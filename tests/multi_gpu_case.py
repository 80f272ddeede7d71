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
dist.barrier()
dist.destroy_process_group()
print(f"MULTI_GPU_OK rank {rank} of {world}, top share {info['top_flops'] / (sum(info['part_flops']) + info['top_flops']):.3f}")
