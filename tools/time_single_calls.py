#!/usr/bin/env python3
"""Latency of the reference-shaped single-item entry points (host buffers in, result out), as the reference's
callers use them: compute_mfcc on one 1 s clip, classify on one 1 s clip."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime per process: torch first)
import dsp_amd  # noqa: E402
from tests import signals as S  # noqa: E402

x = S.classify_cases()["scrub_a"]
for name, fn in (("compute_mfcc(1 s clip, 98 frames)", lambda: dsp_amd.compute_mfcc(x, 98)),
                 ("classify(1 s clip)", lambda: dsp_amd.classify(x))):
    for _ in range(20):
        fn()
    t = []
    for _ in range(200):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print(f"{name}: median {np.median(t):.3f} ms, p10 {np.percentile(t, 10):.3f}, p90 {np.percentile(t, 90):.3f}")
