"""Stations of the fused config-5 kernel's clip tail (pool_finish) from a -DDSP_PF_STAMPS build:
    bash tools/mkvariant.sh pfstamps -DDSP_PF_STAMPS ; DSP_AMD_LIB=variants/pfstamps.so python tools/pf_stamps.py"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import dsp_amd
from dsp_amd.scrubjay import ScrubJay
n = 125000
clips = torch.rand((n, 16000), device="cuda") * 2 - 1
sj = ScrubJay(dict(np.load("tests/golden/scrubjay_svm.npz")))
for _ in range(3):
    sj(clips, 500, fused=True)
torch.cuda.synchronize()
L = dsp_amd.load()
buf = np.zeros(8 * 1024, np.uint64)
L.dsp_debug_pf_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.dsp_debug_pf_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(1024, 8).astype(np.int64)
st = st[st[:, 0] != 0]
names = ["flush of the clip's last tile (6 -> 0)", "mean / std / Scaler (0 -> 1)", "support vectors (1 -> 2)", "reduction (2 -> 3)", "libsvm tail + stores (3 -> 4)", "sync (4 -> 5)"]
pairs = [(6, 0), (0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]
print(f"{len(st)} waves; shader cycles, median / p90")
for nm, (a, b) in zip(names, pairs):
    d = st[:, b] - st[:, a]
    print(f"  {nm:42s} {np.median(d):9.0f} {np.percentile(d, 90):9.0f}")
print(f"  {'whole tail (6 -> 5)':42s} {np.median(st[:, 5] - st[:, 6]):9.0f}")
