import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def cosine_rows(a, b):
    """Row-wise cosine similarity of two [..., D] arrays, computed in float64."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, np.shape(a)[-1])
    b = np.asarray(b, dtype=np.float64).reshape(-1, np.shape(b)[-1])
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1) + 1e-300)


# The parity bar of BASELINE.json:north_star: "within 1e-3 fp16 cosine tolerance".
COS_TOL = 1e-3


def assert_cosine(a, b, tol=COS_TOL, what=""):
    c = cosine_rows(a, b)
    assert np.all(np.isfinite(c)), f"{what}: non-finite cosine"
    assert (1.0 - c).max() <= tol, f"{what}: 1-cos max {(1.0 - c).max():.3e} > {tol:.1e}"


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load
