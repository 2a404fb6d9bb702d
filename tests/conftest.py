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


# Parity gate (DESIGN.md "Parity gate"), the PURE form:
#   |got - ref| <= RTOL * max(|ref|, ||ref frame||_inf),  RTOL = 1e-4 = north_star's tolerance,
# taken relative to the frame's L-inf norm because the reference's own fp32 noise is up to 6e-3 element-wise.
# `gate()` below applies it to every comparison and records how close each case comes (gate_report.json).
#
# The absolute floor ATOL_DB is NOT part of the gate.  It is granted only to frames whose reference L-inf norm is below
# LOW_LEVEL_LINF, in tests that name the case (floor_case=...), because there the pure gate asks for less than the
# reference's own noise: its fp32 twiddle recurrence (mfcc.c:83-85) perturbs every mel energy by ~1e-5 relative = 4e-5 dB,
# which the DCT accumulates to <= 1.5e-4 whatever the frame's level.  Measured against a float64 evaluation of the chain
# on every golden clip (tools note in DESIGN.md 5): the reference itself misses the pure gate on exactly one golden,
# "tiny" (||frame||_inf = 0.46: a clip at the amin floor, flat log-mel spectrum; reference vs float64 truth 3.1e-4 of
# L-inf = 1.46e-4 absolute); every other golden sits at 1e-6 .. 2e-5.
RTOL = 1e-4
ATOL_DB = 3e-4
LOW_LEVEL_LINF = 3.0          # below this, RTOL * L-inf < ATOL_DB: the reference's own noise exceeds the pure gate
LOW_LEVEL_CASES = {
    "tiny": "golden clip at the amin floor, ||frame||_inf = 0.46; reference vs float64 truth = 3.1e-4 of L-inf (1.46e-4 abs)",
}

_GATE_REPORT = {}


def gate(got, ref, case, floor_case=None):
    """Assert the pure 1e-4 * L-inf gate and record the worst ratio under `case`.  floor_case: a key of LOW_LEVEL_CASES
    (or a justification string for a seeded input) -- only then do frames with ||ref||_inf < LOW_LEVEL_LINF get ATOL_DB."""
    ok, worst = frame_linf_close(got, ref, RTOL, 0.0)
    entry = {"pure_worst_rel": worst, "floor": None}
    if not ok and floor_case is not None:
        got64, ref64 = np.asarray(got, np.float64), np.asarray(ref, np.float64)
        low = np.abs(ref64).max(axis=-1) < LOW_LEVEL_LINF
        ok_hi, worst_hi = frame_linf_close(got64[~low], ref64[~low], RTOL, 0.0) if (~low).any() else (True, 0.0)
        ok_lo, worst_lo = frame_linf_close(got64[low], ref64[low], RTOL, ATOL_DB) if low.any() else (True, 0.0)
        ok = ok_hi and ok_lo
        entry = {"pure_worst_rel": worst, "floor": LOW_LEVEL_CASES.get(floor_case, floor_case), "frames_on_floor": int(low.sum()),
                 "worst_rel_other_frames": worst_hi, "worst_with_floor": worst_lo}
    prev = _GATE_REPORT.get(case)
    if prev is None or entry["pure_worst_rel"] > prev["pure_worst_rel"]:
        _GATE_REPORT[case] = entry
    assert ok, f"{case}: worst |err| / (1e-4 * max(|ref|, L-inf)) * 1e-4 = {worst:.3e} exceeds the pure 1e-4 gate" + \
               ("" if floor_case is None else " (even with the low-level floor)")
    return worst


def pytest_sessionfinish(session, exitstatus):
    if not _GATE_REPORT:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "gate_report.json"), "w") as f:
            json.dump({"rtol": RTOL, "atol_db_low_level_only": ATOL_DB, "low_level_linf": LOW_LEVEL_LINF, "cases": _GATE_REPORT}, f, indent=1, sort_keys=True)
    except OSError:
        pass
    worst = sorted(_GATE_REPORT.items(), key=lambda kv: -kv[1]["pure_worst_rel"])[:8]
    print("\nparity gate (pure 1e-4 of frame L-inf), closest cases: " + ", ".join(f"{k} {v['pure_worst_rel']:.1e}" + ("*" if v["floor"] else "") for k, v in worst))


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
