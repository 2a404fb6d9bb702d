"""Determinism of the ragged entry points: clips of 0.5 - 1.5 s in one buffer, LAUNCHES calls of the fused clip -> label kernel, of
classify() and of the float64 classify(), every result compared with the first call's.   python tools/soak_ragged.py [launches]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import dsp_amd
from dsp_amd import lib as L
from dsp_amd.scrubjay import ScrubJay
from tests import signals as S
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(7)
torch.manual_seed(7)
n = 20000
lens = rng.integers(8000, 24001, n)
off = np.zeros(n + 1, dtype=np.int64)
off[1:] = np.cumsum(lens)
c_off = L.c_offsets(off)
flat = (torch.rand(int(off[-1]), device="cuda") * 2 - 1) * 0.05
call = torch.from_numpy(S.classify_cases()["scrub_a"]).cuda()
for c in range(0, n, 4):
    m = int(lens[c])
    flat[int(off[c]):int(off[c]) + m] += call.repeat(2)[:m]
sj = ScrubJay(dict(np.load("tests/golden/scrubjay_svm.npz")))
runs = {"scrubjay fused": lambda: torch.cat([t.reshape(n, -1).float() for t in sj.ragged(flat, c_off, 500)], dim=1),
        "classify": lambda: dsp_amd.classify_device_ragged(flat, c_off).clone(),
        "classify_f64": lambda: dsp_amd.classify_device_ragged_f64(flat.double(), c_off).clone()}
rc = 0
for name, fn in runs.items():
    ref = fn()
    bad = sum(int(not torch.equal(fn(), ref)) for _ in range(launches))
    torch.cuda.synchronize()
    print(f"ragged {name}, {n} clips of 0.5 - 1.5 s, {launches} calls: calls that differ from the first: {bad}")
    rc |= int(bad > 0)
sys.exit(rc)
