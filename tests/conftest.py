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


# Parity gate (DESIGN.md "Parity gate"):
#   |got - ref| <= RTOL * max(|ref|, ||ref frame||_inf) + ATOL_DB
# RTOL = 1e-4 is north_star's tolerance, taken relative to the frame's L-inf norm
# because the reference's own fp32 noise is up to 6e-3 element-wise.  ATOL_DB is the
# reference's measured absolute noise floor in the dB domain: its fp32 twiddle
# recurrence perturbs every mel energy by ~1e-5 relative = 4e-5 dB, which the DCT
# accumulates to <= 1.5e-4 whatever the frame's level (measured against a float64
# evaluation on every golden clip).  Without the floor even exact arithmetic fails the
# gate on low-dynamic-range frames (||frame||_inf < 1, e.g. the "tiny" clip).
RTOL = 1e-4
ATOL_DB = 3e-4


def frame_linf_close(got, ref, rtol=RTOL, atol=0.0):
    """Returns (ok, worst) with worst = max |got-ref| / (rtol*max(|ref|, ||frame||_inf) + atol) * rtol,
    i.e. the smallest rtol that would pass (for atol = 0 simply the worst relative error)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.size == 0:
        return True, 0.0
    if not np.all(np.isfinite(got)):
        return False, float("inf")
    linf = np.abs(ref).max(axis=-1, keepdims=True)
    scale = np.maximum(np.abs(ref), linf)
    err = np.abs(got - ref)
    # all-zero reference frames (silence) must be reproduced exactly
    zero = np.broadcast_to(linf == 0, ref.shape)
    if np.any(err[zero] != 0):
        return False, float("inf")
    bound = rtol * scale + atol
    ratio = np.where(zero, 0.0, err / np.where(bound == 0, 1.0, bound))
    return bool(ratio.max() <= 1.0), float(ratio.max() * rtol)
