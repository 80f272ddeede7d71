"""CPU tests of the synthetic stand-ins for BASELINE configs 2 and 4 (generators only; the parity tests are GPU tests)."""
import numpy as np
import scipy.sparse as sp

from onephase_jl_amd import synth
from oracle import kkt_oracle as KO


def _q(prob):
    H, J = prob["H"], prob["J"]
    return (H + sp.tril(H, -1).T + J.T @ sp.diags(prob["y"] / prob["s"]) @ J).toarray()


def test_hanging_chain_shape_and_indefiniteness():
    p = synth.hanging_chain(N_h=60, seed=1)
    q = synth.hanging_chain(N_h=60, seed=1)
    assert (p["J"] != q["J"]).nnz == 0 and np.array_equal(p["s"], q["s"])            # seeded
    n, m = p["n"], p["m"]
    assert n == 2 * 60 + 2 and m == 2 * (60 + 3) and p["J"].shape == (m, n)
    assert sp.triu(p["H"], 1).nnz == 0                                               # lower triangle only (Class_cutest.jl:548)
    C = p["J"][: m // 2]
    assert abs(C + p["J"][m // 2:]).max() == 0.0                                     # [cons >= l ; -cons <= u]
    rowlen = np.diff(C.tocsr().indptr)
    assert sorted(rowlen)[-1] == 61 and sorted(rowlen)[-2] <= 4                      # banded except the length row
    w = np.linalg.eigvalsh(_q(p))
    assert (w < -1e-10).sum() > 0                                                    # nonconvex: delta = 0 cannot work


def test_infeasible_lp_is_rank_deficient_and_shifts():
    p = synth.infeasible_lp(rows=80, cols=120, seed=3)
    assert p["H"].nnz == 0 and np.all(np.diff(p["J"].tocsc().indptr) > 0)
    A = p["J"][:81].toarray()
    assert np.array_equal(A[80], A[0])                                               # the contradictory twin row
    w = np.linalg.eigvalsh(_q(p))
    assert (np.abs(w) < 1e-9 * w[-1]).sum() >= 1                                     # dependent free columns
    rng = np.random.default_rng(0)
    n, m = p["n"], p["m"]
    it = KO.Iterate(x=np.zeros(n), y=p["y"], s=p["s"], mu=p["mu"], J=p["J"], H=p["H"], grad=rng.normal(size=n), cons=p["s"].copy())
    k = KO.pick_KKT_solver("symmetric")
    k.initialize_b(it)
    k.form_system_b(it)
    status, num_fac, delta, tried = KO.ipopt_strategy_b(it, k)
    assert status == "success" and tried[0] == 0.0 and num_fac >= 2 and delta > 0.0
