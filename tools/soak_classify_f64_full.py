"""Determinism of the float64 classifier at the bench's size (49 152 clips; the work list of clips with midpoints is filled with atomics):
LAUNCHES launches, labels compared with the first launch's every time.   python tools/soak_classify_f64_full.py [launches]"""
import sys
import torch
sys.path.insert(0, ".")
import dsp_amd
from tests import signals as S
n = 49152
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
clips = (torch.rand((n, 16000), device="cuda", dtype=torch.float64) * 2 - 1) * 0.005
call = torch.from_numpy(S.classify_cases()["scrub_a"]).cuda().double()
clips[::4] = call + clips[::4] * 0.1
clips[1::8] *= 10.0                      # an eighth of the batch with a noise floor above the 45 dB threshold: midpoints without the call
lab = torch.empty(n, dtype=torch.int32, device="cuda")
dsp_amd.classify_device_f64(clips, lab)
ref = lab.clone()
bad = 0
for i in range(launches):
    dsp_amd.classify_device_f64(clips, lab)
    bad += int(not torch.equal(lab, ref))
torch.cuda.synchronize()
print(f"classify_f64, {n} clips, {launches} launches: label 1 on {int(ref.sum())} clips, launches that differ from the first: {bad}")
sys.exit(1 if bad else 0)
