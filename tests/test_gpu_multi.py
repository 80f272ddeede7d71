"""Multi-GPU parity (needs at least two GPUs on the node; skipped on the one-GPU boxes): one rank per GPU under
torch.distributed.run, the sharded path with RCCL inside the library and with the torch.distributed transport against the unsharded
solver.  The parent process never touches a GPU (device_count() does not initialise one on this image)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_path_on_real_ranks(world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "multi_gpu_case.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("MULTI_GPU_OK") == world, (r.stdout[-800:], r.stderr[-2000:])
    assert r.stdout.count("MULTI_GPU_SC5") == world          # full-size BASELINE config 5 ran on every rank
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("MULTI_GPU")))
