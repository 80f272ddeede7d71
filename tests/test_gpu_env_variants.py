"""GPU tests: every schedule the tuning switches of DESIGN.md section 9 can select gives the same answers
(the dataflow launch and the per-step schedule with look-ahead on / off / everywhere, super-step widths, split in-group updates,
when the block inversions of the solves start)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The per-step schedule of csrc/numeric.hip (OKKT_DATAFLOW=0: the round-3 path, kept as the cross-check of the dataflow launch)
# with its own switches, then the switches that both paths share.
STEPS = {"OKKT_DATAFLOW": "0"}
VARIANTS = [
    {},
    dict(STEPS),
    dict(STEPS, OKKT_LOOKAHEAD="0"),
    dict(STEPS, OKKT_LA_MIN_TILES="1"),                            # look-ahead on every big front, masked twin stream everywhere
    dict(STEPS, OKKT_LA_MIN_TILES="1", OKKT_SPLIT_MIN_ROWS="0", OKKT_GROUP="4", OKKT_GROUP_ONE_ROWS="0"),
    dict(STEPS, OKKT_GROUP="1"),
    dict(STEPS, OKKT_GROUP_BIG="4", OKKT_GROUP_BIG_MINF="0", OKKT_GROUP_SWITCH_ROWS="0", OKKT_GROUP_ONE_ROWS="0"),
    dict(STEPS, OKKT_RESERVED_CUS="32", OKKT_LA_MIN_TILES="64"),
    dict(STEPS, OKKT_SB_TAIL_ROWS="100000"),                       # block inversions start as early as a block is final (needs a front of >= 4 blocks)
    dict(STEPS, OKKT_SYRK_SMALL_TILES="0"),                        # 128 x 128 tiles everywhere
    dict(STEPS, OKKT_SYRK_SMALL_TILES="100000000"),                # 128 x 64 tiles everywhere
    dict(STEPS, OKKT_DIAG2="0"),                                   # the round-2 diagonal-block kernel
    dict(STEPS, OKKT_FUSE_DIAG_TRSM="0"),                          # diagonal block and the rows below it always in two launches
    dict(STEPS, OKKT_FUSE_DIAG_TRSM="1", OKKT_LA_MIN_TILES="1", OKKT_SPLIT_MIN_ROWS="0"),      # ... in one launch everywhere
    {"OKKT_SB_TAIL_ROWS": "-1"},
    {"OKKT_ASM_CHUNKED": "0"},
    {"OKKT_ASM_LCOL": "0"},
    {"OKKT_TASKS": "0"},                                          # one launch per level of small fronts
    {"OKKT_TASK_ABS": "1e9"},                                     # every all-small subtree is one workgroup's task
    {"OKKT_SOLVE_FUSE": "0"},                                     # the sweeps with two launches per level of thin fronts
    {"OKKT_SOLVE_FUSE_WIDE_MAX": "100000000"},                    # the wide fronts fused as well
    {"OKKT_SOLVE_FLOW": "1"},                                     # forward sweep of the wide fronts: one launch per level, vectors handed on through tile states
    {"OKKT_DF_SPLIT_TU": "0"},                                    # the block row behind a diagonal block as one task
    {"OKKT_SOLVE_MID": "0"},                                      # every pivot block of more than 128 columns through its explicit inverse (rounds 2 - 4)
    {"OKKT_SOLVE_MID": "1024"},                                   # ... through block substitution in 64-column steps up to 1024 columns (default: 384)
    {"OKKT_SOLVE_SPLIT_SMALL": "0"},                              # panel GEMVs of the wide fronts always with 64 rows / columns per workgroup
    {"OKKT_FLOW": "0"},                                           # small-front tasks: one launch per level instead of one for all levels
    {"OKKT_DF_GROUP": "2", "OKKT_DF_WORKERS": "48"},              # the dataflow launch with pairs of panels on 48 workers
    {"OKKT_ORDERING_TEST": "3"},                                  # (read by the case) AMD instead of the automatic choice
    {"OKKT_DEBUG_POISON": "1"},                                   # the front arena starts as NaNs (never-written upper triangles): no kernel may let one through
    {"OKKT_DEBUG_POISON": "1", "OKKT_SOLVE_MID": "1024"},
    {"OKKT_DEBUG_POISON": "1", "OKKT_SOLVE_MID": "0"},
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k[5:]}={v}" for k, v in e.items()) or "default")
def test_schedule_variant(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_variant_case.py")], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "VARIANT_OK" in r.stdout, (env, r.stdout[-400:], r.stderr[-1200:])


@pytest.mark.parametrize("drop,what", [("1", "factor"), ("2", "solve")])
def test_a_lost_hand_off_is_loud(drop, what):
    """The in-launch waits are bounded (a GPU never hangs on them).  When one runs into its bound -- forced here by letting the
    consumers wait for an epoch nobody raises -- the factorisation fails with an error (the lost hand-offs are counted apart, so
    the pivot counts no longer add up to the matrix order), and a solve returns NaN instead of numbers computed from data that
    had not arrived."""
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, '.')\n"
        "from onephase_jl_amd import synth\n"
        "from onephase_jl_amd.linear_system_solvers import OkktError, finalize_b, initialize_b, linear_solver_HIP\n"
        "prob = synth.hanging_chain(N_h=300, seed=2)\n"
        "K = synth.augmented_matrix(prob, delta=0.5)\n"
        "h = linear_solver_HIP('symmetric'); initialize_b(h)\n"
        "try:\n"
        "    rc = h.ls_factor_b(K, prob['n'], prob['m'])\n"
        "except OkktError as e:\n"
        "    print('FACTOR_ERROR', e); sys.exit(0)\n"
        "x = h.ls_solve(np.ones(K.shape[0]))\n"
        "print('RC', rc, 'NAN', int(np.isnan(x).all()))\n"
        "finalize_b(h)\n")
    e = dict(os.environ)
    e["OKKT_DEBUG_DROP_HANDOFF"] = drop
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1200:]
    if what == "factor":
        assert "FACTOR_ERROR" in r.stdout and "pivot counts" in r.stdout, r.stdout[-400:]
    else:
        line = [l for l in r.stdout.splitlines() if l.startswith("RC")][-1].split()
        assert int(line[1]) == 1 and int(line[3]) == 1, r.stdout[-400:]
