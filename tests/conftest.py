import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (runs on the MI355X box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_functions():
    return load_golden("functions")


@pytest.fixture(scope="session")
def golden_constraints():
    return load_golden("constraints")


@pytest.fixture(scope="session")
def golden_sphere():
    return load_golden("sphere")


@pytest.fixture(scope="session")
def golden_trajectories():
    return load_golden("trajectories")


@pytest.fixture(scope="session")
def golden_spectral():
    return load_golden("spectral")


@pytest.fixture(scope="session")
def golden_cycle():
    return load_golden("cycle")


def func_from_golden(g, name):
    """oracle function descriptor of a golden function case."""
    from oracle import oracle
    a1 = g[name + "__a1"] if (name + "__a1") in g.files else None
    return oracle.func(str(g[name + "__kind"]), g[name + "__a0"], a1, tuple(g[name + "__scalars"]),
                       str(g[name + "__kind_neg"]), tuple(g[name + "__scalars_neg"]))


# tolerances of SURVEY section 8c (kernel level): loss rtol 1e-5; gradient rtol 1e-4 with
# atol 1e-5 * max|grad|
LOSS_RTOL = 1e-5
GRAD_RTOL = 1e-4
GRAD_ATOL_REL = 1e-5


def assert_grad_close(got, want, rtol=GRAD_RTOL, atol_rel=GRAD_ATOL_REL):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    scale = max(float(np.abs(want).max()), 1e-30)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol_rel * scale)
