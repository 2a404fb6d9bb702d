"""Development aid: the two-frames-per-wave kernel (DSP_KERNEL_PAIR = 3) against the default kernel on the same frames."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import dsp_amd

for n in (1, 2, 15, 16, 17, 31, 33, 1000, 100_003):
    x = torch.rand((n, 512), device="cuda") * 2 - 1
    if n > 8:
        x[3] = 0.0
        x[5] *= 1e-4
    a = dsp_amd.MfccPlan(dsp_amd.default_config(frame_length=512, hop_length=512))
    b = dsp_amd.MfccPlan(dsp_amd.default_config(frame_length=512, hop_length=512))
    b.set_kernel(3)
    ya, yb = a.frames(x).cpu().numpy().astype(np.float64), b.frames(x).cpu().numpy().astype(np.float64)
    den = np.maximum(np.abs(ya), np.abs(ya).max(axis=1, keepdims=True)) + 1e-30
    rel = np.abs(ya - yb) / den
    print(n, "worst rel", f"{rel.max():.2e}", "frames over 1e-4:", int((rel.max(axis=1) > 1e-4).sum()), "nan", int(np.isnan(yb).sum()))
