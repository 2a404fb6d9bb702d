#!/usr/bin/env python3
"""Round-2 additions to tests/golden (same rules as make_golden.py: runs only where the reference checkout and
oracle/_ref exist, executes the reference's OWN compiled code, stores inputs + outputs only).

    python tests/golden/make_golden_round2.py

Fixtures written:
  sum_intense_ref.npz   sum_intense (sync/lib/classifier.cpp:370-431) of the reference's compiled classifier.cpp on
                        dB maps with dropped (NaN) cells: real band-kept maps of two golden clips, a random map, and
                        small maps whose windows hit the clamp / swap branches (:383-412)
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def kept_map(sxx):
    """classifier.cpp:35-80 in numpy (any map with NaNs would do as an INPUT of sum_intense; this one has the real shape)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        db = np.where(sxx > 0, (10 * np.log10(sxx.astype(np.float64) / 1e-12)).astype(np.float32), np.float32(np.nan))
    mn, mx = np.nanmin(db), np.nanmax(db)
    v = ((db - mn) / (mx - mn)).astype(np.float32)
    return np.where((v > np.float32(0.65)) & (v < np.float32(0.80)), v, np.float32(np.nan)).astype(np.float32)


def main():
    L = O.ref_classifier_lib()
    g = np.load(os.path.join(HERE, "classifier_ref.npz"))
    freqs, times = g["freqs"], g["times_16000"]
    rng = np.random.default_rng(2)
    out, n = {}, 0

    def add(db, fr, tm, lower, upper, half, mid):
        nonlocal n
        db = np.ascontiguousarray(db, np.float32); fr = np.ascontiguousarray(fr, np.float32); tm = np.ascontiguousarray(tm, np.float32)
        want = L.ref_sum_intense(lower, upper, half, fr.copy(), fr.size, tm.copy(), tm.size, db.reshape(-1).copy(), mid)
        out[f"c{n}_db"] = db; out[f"c{n}_freqs"] = fr; out[f"c{n}_times"] = tm
        out[f"c{n}_params"] = np.array([lower, upper, half, mid], np.float32)
        out[f"c{n}_sum"] = np.float32(want)
        print(f"case {n}: shape {db.shape} band {lower}-{upper} +-{half} @ {mid:.4f} -> {want!r}")
        n += 1

    for name in ("scrub_a", "jay_like"):
        m = kept_map(g[f"{name}__sxx"])
        for mid in list(g[f"{name}__midpoints"]) + [0.05, 0.99]:
            for lo, hi, half in ((5000, 7000, 0.18), (2500, 5000, 0.05), (500, 2500, 0.18)):     # classifier.cpp:99-101
                add(m, freqs, times, lo, hi, half, float(mid))
    r = rng.normal(0, 3, (129, 71)).astype(np.float32)
    r[rng.random((129, 71)) < 0.6] = np.nan
    r[5, 7] = -0.0
    add(r, freqs, times, 0, 8000, 10.0, 0.5)                 # the whole map: 9159 cells in order
    add(r, freqs, times, 3100, 3200, 0.02, 0.5)              # a handful of cells
    small = np.array([[1.5, np.nan, 2.25], [np.nan, np.nan, np.nan], [0.125, 7.0, np.nan], [3.0, -1.0, 0.5], [np.nan, 1e-3, 9.0]], np.float32)
    sf, st = np.array([0, 100, 200, 300, 400], np.float32), np.array([0.1, 0.2, 0.3], np.float32)
    add(small, sf, st, 1000, 2000, 0.05, 0.2)                # lower above every bin: clamp to the last row (:383-384)
    add(small, sf, st, -50, -10, 0.05, 0.2)                  # upper below every bin: clamp to row 0 (:385-386)
    add(small, sf, st, 250, 260, 0.05, 0.2)                  # empty band: min index > max index, swapped (:389-394)
    add(small, sf, st, 0, 400, 0.01, 5.0)                    # window after the last column
    add(small, sf, st, 0, 400, 0.01, -5.0)                   # window before the first column
    add(small, sf, st, 0, 400, 0.01, 0.25)                   # window between two columns (swap)
    add(small, sf, st, 100, 300, 0.1, 0.2)
    out["n_cases"] = np.int32(n)
    np.savez_compressed(os.path.join(HERE, "sum_intense_ref.npz"), **out)


if __name__ == "__main__":
    main()
