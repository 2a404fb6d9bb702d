import torch, time, sys
sys.path.insert(0,'/root/repo')
import dsp_amd
n=1_000_000
x=torch.rand((n,1024),device='cuda')*2-1
for pre in (0,2):
    plan=dsp_amd.MfccPlan(dsp_amd.default_config(n_fft=1024,frame_length=1024,hop_length=1024,n_mels=128,prefilter=pre))
    out=torch.empty((n,13),device='cuda')
    for _ in range(3): plan.frames(x,out)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.frames(x,out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print("prefilter",pre,"ms",ms,"GB/s",4148*n/ms/1e6)
