import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "kkt_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The C-ABI library and the oracle are built once per session (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as g
    g.build()


def iterate_from_record(rec, Iterate):
    """Build an iterate (oracle or product flavour) from a golden problem record."""
    n, m = rec["n"], rec["m"]
    H = sp.csc_matrix(np.array(rec["H_lower"], dtype=float).reshape(n, n))
    J = sp.csc_matrix(np.array(rec["J"], dtype=float).reshape(m, n))
    return Iterate(x=np.array(rec["x"], float), y=np.array(rec["y"], float), s=np.array(rec["s"], float),
                   mu=float(rec["mu"]), J=J, H=H, grad=np.array(rec["grad"], float),
                   cons=np.array(rec["cons"], float), a_norm_penalty_par=rec["a_norm_penalty"])
