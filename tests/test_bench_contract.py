"""CPU checks of the benchmark contract: `python bench.py --gpus N` starts its N ranks itself (one child, torch.distributed.run on
127.0.0.1, nothing touches the GPU in the parent), and the committed JSON line of the round carries every field the driver reads."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_spawns_the_ranks_as_a_child(monkeypatch):
    import subprocess

    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7                                    # the child's exit code is passed on
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_committed_bench_line_has_the_contract_fields():
    path = os.path.join(ROOT, "profiles", "r02_bench.json")
    if not os.path.exists(path):
        pytest.skip("no committed round-2 bench line yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["solve"]["bound"] == "hbm" and 0 < r["solve"]["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    assert d["parity"]["inertia_equal"] and d["parity"]["sign_D_equal"] and d["parity"]["rel_err_x"] <= d["parity"]["tolerance_x"]
    assert abs(d["value"] - d["n_gpus"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]
