"""GPU tests of the dataflow factorisation of the big fronts (csrc/dataflow.hip: one persistent launch per level, tasks on
128 x 128 tiles handed over through per-tile states) against the per-step launches of csrc/numeric.hip (OKKT_DATAFLOW=0), which
the other GPU tests pin against the oracle.  Both run the same operations per entry in the same order, so D, the stored factor
and the solution must agree BIT FOR BIT -- any hand-off that delivered a stale tile would show up here.  The reference reaches
this arithmetic through CHOLMOD (src/linear_system_solvers/julia.jl:34,52,72)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_case(case, env, tmp_path, tag):
    out = str(tmp_path / f"{case}_{tag}.npz")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dataflow_case.py"), case, out], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CASE_OK" in r.stdout, (case, env, r.stdout[-400:], r.stderr[-1500:])
    return dict(np.load(out))


# one block column and one row more (a 1 x 1 last block), a short last pivot block, several whole blocks, 21 block columns with
# the look-ahead chain, and BASELINE config 3 (every front shape of a real elimination tree, contribution blocks included)
@pytest.mark.parametrize("case", ["dense129", "dense300", "dense640", "dense2600", "S-C3"])
def test_dataflow_equals_the_per_step_launches_bit_for_bit(case, tmp_path):
    a = run_case(case, {"OKKT_DATAFLOW": "0"}, tmp_path, "steps")
    b = run_case(case, {"OKKT_DATAFLOW": "1"}, tmp_path, "flow")
    assert int(a["rc"]) == 1 and int(b["rc"]) == 1
    assert a["inertia"].tolist() == b["inertia"].tolist() == [int(a["want"][0]), int(a["want"][1]), 0, 0]
    assert np.array_equal(a["d"], b["d"])
    assert np.array_equal(a["Lidx"], b["Lidx"]) and np.array_equal(a["Ldata"], b["Ldata"])
    assert np.array_equal(a["x"], b["x"])


EXP = {"OKKT_LIB_PATH": os.path.join(ROOT, "onephase.jl_amd", "libonephase_kkt_exp.so")}      # the experiments of round 6 live in a library of their own (csrc/Makefile)


def _exp(**e):
    d = dict(EXP)
    d.update(e)
    return d


@pytest.mark.parametrize("env", [{"OKKT_DF_GROUP": "1"}, {"OKKT_DF_GROUP": "2"}, _exp(OKKT_DF_GROUP="3", OKKT_DF_ROWS="2"), _exp(OKKT_DF_ROWS="4"),
                                 {"OKKT_DF_WORKERS": "7"}, {"OKKT_DF_WORKERS": "64", "OKKT_DF_MODEL_CHAIN": "2.0"}, {"OKKT_DF_SPLIT_TU": "0"},
                                 {"OKKT_DF_SPLIT_TU": "0", "OKKT_DF_FUSE_D": "0"}, {"OKKT_DF_FUSE_D": "0"},
                                 {"OKKT_DF_FUSE_TL": "0"}, {"OKKT_DF_FUSE_TL": "0", "OKKT_DF_FUSE_D": "0", "OKKT_DF_GROUP": "2"},
                                 _exp(OKKT_DF_ROWS="4", OKKT_DF_EARLY_PUB="0"), _exp(OKKT_DF_ROWS_BIG="2", OKKT_DF_ROWS_MINKB="8", OKKT_DF_ROWS_AHEAD="4", OKKT_DF_MACRO="0"),
                                 _exp(OKKT_DF_ROWS_BIG="3", OKKT_DF_ROWS_MINKB="4", OKKT_DF_ROWS_AHEAD="1", OKKT_DF_ROWS_COLDIST="0", OKKT_DF_WORKERS="9"),
                                 EXP,
                                 _exp(OKKT_DF_ROWS_BIG="2", OKKT_DF_ROWS_MINKB="8", OKKT_DF_ROWS_AHEAD="4"), _exp(OKKT_DF_ROWS_BIG="4", OKKT_DF_ROWS_MINKB="4", OKKT_DF_ROWS_AHEAD="1", OKKT_DF_GROUP="2"),
                                 _exp(OKKT_DF_LOCKSTEP="1"), _exp(OKKT_DF_LOCKSTEP="1", OKKT_DF_WORKERS="5"), _exp(OKKT_DF_LOCKSTEP="1", OKKT_DF_FUSE_D="0"),
                                 _exp(OKKT_DF_CHAIN="1"), _exp(OKKT_DF_CHAIN="3", OKKT_DF_WORKERS="7"), _exp(OKKT_DF_CHAIN="1", OKKT_DF_GROUP="8", OKKT_DF_WORKERS="16")],
                         ids=lambda e: ",".join(("exp" if k == "OKKT_LIB_PATH" else f"{k[8:]}={v}") for k, v in e.items()))
def test_every_queue_shape_gives_the_same_factor(env, tmp_path):
    """The grouping of the panels, the number of row tiles per task, the number of workers, the time model, one or two workers for
    the block row behind a diagonal block, the diagonal block as a task of its own, the last update of a tile inside its panel task
    (TL) and the block row behind a diagonal block in lockstep with it (round 6: OKKT_DF_LOCKSTEP=1, an experiment that is off by
    default) or behind the whole of it, and update tasks chained on a worker without leaving the operand ring (round 6: OKKT_DF_CHAIN, off
    by default) only change the ORDER of the queue and who computes what; the factor does not depend on it.  The last three kinds are
    compiled into libonephase_kkt_exp.so only ("exp": that library with its defaults against the product library).  That the variants do run there:
    test_the_experiments_library_runs_its_roles."""
    a = run_case("dense2600", {"OKKT_DATAFLOW": "1"}, tmp_path, "default")
    e = {"OKKT_DATAFLOW": "1"}
    e.update(env)
    b = run_case("dense2600", e, tmp_path, "variant")
    assert np.array_equal(a["d"], b["d"]) and np.array_equal(a["Ldata"], b["Ldata"]) and np.array_equal(a["x"], b["x"])


def test_the_experiments_library_runs_its_roles():
    """The variants above that load libonephase_kkt_exp.so would pass trivially if that library were the product library under another
    name: its version string names the roles its dataflow kernel was compiled with, the product library's says it has none."""
    import ctypes
    prod = ctypes.CDLL(os.path.join(ROOT, "onephase.jl_amd", "libonephase_kkt.so"))
    exp = ctypes.CDLL(EXP["OKKT_LIB_PATH"])
    for lib in (prod, exp):
        lib.okkt_version.restype = ctypes.c_char_p
    assert b"lockstep=0 macro=0 chain=0 log=0" in prod.okkt_version()
    assert b"lockstep=1 macro=1 chain=1 log=1" in exp.okkt_version()


def test_released_contribution_blocks_give_the_same_factor(tmp_path):
    """Round 6: the fronts' contribution blocks share a region of the arena by lifetime (numeric_setup) instead of living in an f x f
    buffer per front; OKKT_RELEASE_CB=0 keeps the buffers of rounds 1 - 5.  Same arithmetic, other addresses: bit for bit the same factor
    on BASELINE config 3 (every front shape of a real elimination tree, small-front tasks and big fronts)."""
    a = run_case("S-C3", {"OKKT_RELEASE_CB": "0"}, tmp_path, "dense_buffers")
    b = run_case("S-C3", {"OKKT_RELEASE_CB": "1"}, tmp_path, "released")
    assert int(a["rc"]) == 1 and int(b["rc"]) == 1
    assert np.array_equal(a["d"], b["d"]) and np.array_equal(a["Ldata"], b["Ldata"]) and np.array_equal(a["x"], b["x"])


def test_a_lost_hand_off_of_the_dataflow_launch_is_loud():
    """Every wait of the launch is bounded.  One that runs into its bound (forced: the first wait condition of every task asks for a
    state nobody publishes) raises the time-out word, every worker leaves, and the factorisation fails with an error instead of
    returning a factor computed from tiles that had not arrived."""
    code = (
        "import sys, numpy as np, scipy.sparse as sp\n"
        "sys.path.insert(0, '.')\n"
        "from onephase_jl_amd.linear_system_solvers import OkktError, finalize_b, initialize_b, linear_solver_HIP\n"
        "rng = np.random.default_rng(3); n = 400\n"
        "B = rng.normal(size=(n, n)); M = B + B.T + np.diag(np.full(n, 3.0 * np.sqrt(n)))\n"
        "h = linear_solver_HIP('symmetric'); initialize_b(h)\n"
        "try:\n"
        "    rc = h.ls_factor_b(sp.csc_matrix(np.tril(M)), n, 0)\n"
        "except OkktError as e:\n"
        "    print('FACTOR_ERROR', e); sys.exit(0)\n"
        "print('RC', rc)\n")
    e = dict(os.environ)
    e["OKKT_DATAFLOW"] = "1"
    e["OKKT_DEBUG_DROP_HANDOFF"] = "4"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1200:]
    assert "FACTOR_ERROR" in r.stdout and "pivot counts" in r.stdout, r.stdout[-400:]
