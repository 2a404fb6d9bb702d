"""Segments the energy gate leaves to the flag transform, and clips with midpoints, in bench.py's classify workload."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import dsp_amd
from dsp_amd import lib as L
from tests import signals as S
n = 49152
torch.manual_seed(1234)
clips = (torch.rand((n, 16000), device="cuda") * 2 - 1) * 0.005
call = torch.from_numpy(S.classify_cases()["scrub_a"]).cuda()
clips[::4] = call + clips[::4] * 0.1
lab = dsp_amd.classify_device(clips)
torch.cuda.synchronize()
g, h = C.c_long(), C.c_long()
L.check(L.load().dsp_classify_stats(0, C.byref(g), C.byref(h)), "stats")
T = (16000 - 256) // 224 + 1
print(f"{n} clips x {T} segments = {n * T}: gated in {g.value} ({100.0 * g.value / (n * T):.1f} %), clips with midpoints {h.value}, label 1 on {int(lab.sum())}")
