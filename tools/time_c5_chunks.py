"""config 5 breakdown: the plain clips kernel at n_mfcc = 20 for several chunk sizes beside the fused clip -> label kernel"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import dsp_amd
from dsp_amd.scrubjay import ScrubJay
n = 125000
clips = torch.rand((n, 16000), device="cuda") * 2 - 1


def timeit(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for n_mfcc in (13, 20):
    plan = dsp_amd.MfccPlan(dsp_amd.default_config(n_mfcc=n_mfcc))
    out = torch.empty((n, 98, n_mfcc), device="cuda")
    for chunk in (8, 16, 32, 96):
        plan.set_launch(0, chunk)
        print(f"plain n_mfcc {n_mfcc} chunk {chunk}: {timeit(lambda: plan.clips(clips, 500, out)):.4f} ms")
    del out
sj = ScrubJay(dict(np.load("tests/golden/scrubjay_svm.npz")))
print(f"fused: {timeit(lambda: sj(clips, 500, fused=True)):.4f} ms")
