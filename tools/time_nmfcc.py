"""clips kernel time vs n_mfcc (13 -> DCT shape <4,10,3>, 20 -> <2,20,3>) on 125 000 x 1 s clips"""
import sys, torch
sys.path.insert(0, '.')
import dsp_amd
n = 125000
clips = torch.rand((n, 16000), device="cuda") * 2 - 1
for n_mfcc in (13, 16, 20):
    plan = dsp_amd.MfccPlan(dsp_amd.default_config(n_mfcc=n_mfcc))
    out = torch.empty((n, 98, n_mfcc), device="cuda")
    for _ in range(3): plan.clips(clips, 500, out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.clips(clips, 500, out)
    e1.record(); torch.cuda.synchronize()
    print("n_mfcc", n_mfcc, e0.elapsed_time(e1) / 10, "ms")
