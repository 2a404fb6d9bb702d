import torch, sys, os
sys.path.insert(0, '.')
import dsp_amd
n = 12500
clips = torch.rand((n, 16000), device="cuda") * 2 - 1
plan = dsp_amd.MfccPlan()
out = torch.empty((n, 98, 13), device="cuda")
for _ in range(5): plan.clips(clips, 500, out)
torch.cuda.synchronize()
ts = []
for r in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan.clips(clips, 500, out)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print(os.environ.get("DSP_AMD_LIB", "default"), "clips ms", sorted(ts)[2])
