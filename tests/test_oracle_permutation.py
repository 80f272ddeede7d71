"""CPU: the oracle's observables -- (status, #fac, delta) of ipopt_strategy!, the direction, N err
(/root/reference/src/IPM/delta_strategy.jl:37-114, kkt_system_solver.jl:67-96) -- are the same for two pivot orders the
oracle picks itself (natural order, reverse Cuthill-McKee).  The GPU twin (test_gpu_permutation_independence.py) compares
the HIP path, which runs on a third order, with both."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import kkt_oracle as KO

_spec = importlib.util.spec_from_file_location("perm_cases", os.path.join(os.path.dirname(__file__), "test_gpu_permutation_independence.py"))
_tp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_tp)


@pytest.mark.parametrize("name", sorted(_tp.CASES))
@pytest.mark.parametrize("kind", ["schur", "symmetric"])
def test_oracle_trace_and_direction_for_two_pivot_orders(name, kind):
    prob = _tp.CASES[name]()
    res = {}
    for which in ("natural", "rcm"):
        _, oit = _tp.iterates(prob)
        ko = KO.pick_KKT_solver(kind, perm=_tp.oracle_perm(prob, kind, which))
        ko.initialize_b(oit)
        ko.form_system_b(oit)
        r = KO.ipopt_strategy_b(oit, ko)
        ko.kkt_associate_rhs_b(oit, KO.Reduct_stable())
        ko.compute_direction_b()
        res[which] = (r[:3], {a: getattr(ko.dir, a).copy() for a in ("x", "y", "s")}, ko.kkt_err_norm.ratio)
    assert res["natural"][0] == res["rcm"][0]
    assert res["natural"][0][0] == "success" and res["natural"][0][1] >= 2
    for a in ("x", "y", "s"):
        ref = res["natural"][1][a]
        assert np.linalg.norm(res["rcm"][1][a] - ref) <= _tp.DIR_TOL.get((name, kind), 1e-6) * np.linalg.norm(ref)
    assert res["natural"][2] < 1e-6 and res["rcm"][2] < 1e-6
