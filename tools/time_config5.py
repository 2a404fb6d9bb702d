import torch, sys, os, numpy as np
sys.path.insert(0, '.')
import dsp_amd
from dsp_amd.scrubjay import ScrubJay
attrs = dict(np.load("tests/golden/scrubjay_svm.npz"))
n = 125000
clips = torch.rand((n, 16000), device="cuda") * 2 - 1
sj = ScrubJay(attrs)
bpc = int(os.environ.get("BPC", "0"))
if bpc: sj.plan.set_launch(bpc, 0)
for fused in (True, False):
    for _ in range(3): sj(clips, 500, fused=fused)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): sj(clips, 500, fused=fused)
    e1.record(); torch.cuda.synchronize()
    print("fused" if fused else "three", e0.elapsed_time(e1) / 10, "ms")
