"""Phase timing of spec_from_ckpt_kernel from a -DDSP_RC_STAMPS build (in-kernel s_memtime stamps of the first 2048 blocks).
    DSP_AMD_LIB=variants/stamps.so DSP_AMD_EXTRA_FLAGS=-DDSP_RC_STAMPS python -m dsp_amd.build     # build container
    DSP_AMD_LIB=variants/stamps.so python tools/rc_stamps.py                                      # GPU box
The last launch of spec_from_ckpt_kernel in classify() is the 3000-7500 Hz map (every frame of the clips with midpoints)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import dsp_amd
from tests import signals as S

n = 49152
clips = (torch.rand((n, 16000), device="cuda") * 2 - 1) * 0.05
call = torch.from_numpy(S.classify_cases()["scrub_a"]).cuda()
clips[::4] = call + clips[::4] * 0.01
labels = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    dsp_amd.classify_device(clips, labels)
torch.cuda.synchronize()
L = dsp_amd.load()
buf = np.zeros(8 * 2048, np.uint64)
L.dsp_debug_rc_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.dsp_debug_rc_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(2048, 8).astype(np.int64)
st = st[st[:, 0] != 0]                    # persistent kernel: only the resident blocks exist
print(f"{len(st)} blocks stamped; whole kernel per block (stamp 6 -> 7): median {np.median(st[:, 7] - st[:, 6]):.0f}, "
      f"min {np.min(st[:, 7] - st[:, 6])}, max {np.max(st[:, 7] - st[:, 6])} cycles; span over blocks {st[:, 7].max() - st[:, 6].min()}")
d = np.diff(st[:, :6], axis=1)
names = ["L (segment loads -> LDS)", "R (recurrence, one wave)", "T (taps, all waves)", "M (means, one wave)", "F (FFT + PSD out)"]
print("ticks of s_memtime (100 MHz constant clock on gfx950: 1 tick = 10 ns), median / p90 over 2048 blocks")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 90):9.0f}")
print(f"  {'block total':28s} {np.median(st[:, 5] - st[:, 0]):9.0f}")

buf2 = np.zeros(8 * 2048, np.uint64)
L.dsp_debug_bd_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.dsp_debug_bd_stamps(buf2.ctypes.data, buf2.size) == 0
st = buf2.reshape(2048, 8).astype(np.int64)
d = np.diff(st[:, :4], axis=1)
print("classify_bands_kernel, last clip of each of the first 2048 blocks (shader cycles, median / p90)")
for i, nm in enumerate(["PSD load + min/max", "candidates + float64 log10", "band sums per midpoint + rule"]):
    print(f"  {nm:32s} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 90):9.0f}")
print(f"  {'clip total':32s} {np.median(st[:, 3] - st[:, 0]):9.0f}")
