"""Determinism of classify() at the bench's size (49 152 clips: three blocks per CU, the checkpoint kernel's SIMD load table on by
default): LAUNCHES launches, labels compared with the first launch's every time.   python tools/soak_classify_full.py [launches]"""
import sys
import torch
sys.path.insert(0, ".")
import dsp_amd
from tests import signals as S
n = 49152
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
clips = (torch.rand((n, 16000), device="cuda") * 2 - 1) * 0.05
call = torch.from_numpy(S.classify_cases()["scrub_a"]).cuda()
clips[::4] = call + clips[::4] * 0.01
lab = torch.empty(n, dtype=torch.int32, device="cuda")
dsp_amd.classify_device(clips, lab)
ref = lab.clone()
bad = 0
for i in range(launches):
    dsp_amd.classify_device(clips, lab)
    bad += int(not torch.equal(lab, ref))
torch.cuda.synchronize()
print(f"classify, {n} clips, {launches} launches: label 1 on {int(ref.sum())} clips, launches that differ from the first: {bad}")
sys.exit(1 if bad else 0)
