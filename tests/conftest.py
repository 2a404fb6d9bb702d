import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def frame_linf_close(got, ref, rtol=1e-4):
    """The parity gate of BASELINE.md / SURVEY.md 8(c):
    |got - ref| <= rtol * max(|ref|, ||ref frame||_inf) for every coefficient.
    Returns (ok, worst_ratio)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.size == 0:
        return True, 0.0
    linf = np.abs(ref).max(axis=-1, keepdims=True)
    scale = np.maximum(np.abs(ref), linf)
    err = np.abs(got - ref)
    # all-zero reference frames (silence) must be reproduced exactly
    zero = scale == 0
    if np.any(err[zero] != 0):
        return False, float("inf")
    ratio = np.where(zero, 0.0, err / np.where(zero, 1.0, scale))
    return bool(ratio.max() <= rtol), float(ratio.max())
