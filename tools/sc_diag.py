"""Diagnostic build (-DSC_DIAG) of the float64 classifier's screening kernel: which role ran on which SIMD, busy vs total cycles."""
import ctypes as C
import os
import sys
from collections import Counter

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from dsp_amd import lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
clips = (torch.rand((n, 16000), device="cuda", dtype=torch.float64) * 2 - 1) * 0.005
labels = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    dsp_amd.classify_device_f64(clips, labels)
torch.cuda.synchronize()
nb = (n + 63) // 64
buf = (C.c_int * (16 * nb))()
lib = L.load()
lib.dsp_classify_debug_f64.argtypes = [C.c_int, C.c_void_p, C.c_int]
L.check(lib.dsp_classify_debug_f64(0, buf, 16 * nb), "debug")
a = np.frombuffer(buf, dtype=np.int32).reshape(nb, 16)
if len(set(a[:, 0].tolist())) < 8:
    sys.exit("no per-CU records (DSP_AMD_F64_ROLES=0 or not a -DSC_DIAG build)")
print("blocks per CU:", Counter(Counter(a[:, 0].tolist()).values()))
print("arrival numbers:", Counter(a[:, 1].tolist()))
# per CU: role on each SIMD
per = {}
for row in a:
    per.setdefault(int(row[0]), []).append(row)
pat = Counter()
for cu, rows in per.items():
    simd = [[], [], [], []]
    for row in rows:
        for role in range(4):
            simd[row[4 + role]].append("BMTX"[role])
    pat[" ".join("".join(sorted(x)) for x in simd)] += 1
print("roles per SIMD (B = R_bp, M = R_mp, T, X), CUs with that pattern:")
for k, v in pat.most_common(12):
    print("  ", k, v)
busy = a[:, 8:12].astype(np.float64) * 16
tot = a[:, 12:16].astype(np.float64) * 16
for role in range(4):
    print("BMTX"[role], "busy %.0f k cycles (%.0f %% of %.0f k)" % (busy[:, role].mean() / 1e3, 100 * busy[:, role].mean() / tot[:, role].mean(), tot[:, role].mean() / 1e3))
