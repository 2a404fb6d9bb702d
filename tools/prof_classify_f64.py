"""The float64 classifier's bench workload, a few calls, for rocprofv3 --kernel-trace --stats (argv: clips, 'pcm16' for int16 input)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from tests import signals as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(2000)
clips = (torch.rand((n, 16000), device=dev, generator=gen, dtype=torch.float64) * 2 - 1) * 0.005
call = torch.from_numpy(S.classify_cases()["scrub_a"]).to(dev).double()
clips[::4] = call + clips[::4] * 0.1
labels = torch.empty(n, dtype=torch.int32, device=dev)
if len(sys.argv) > 2 and sys.argv[2] == "pcm16":
    pcm = torch.clamp(torch.round(clips * 32768.0), -32768, 32767).to(torch.int16)
    del clips
    step = lambda: dsp_amd.classify_device_f64_pcm16(pcm, labels)  # noqa: E731
else:
    step = lambda: dsp_amd.classify_device_f64(clips, labels)  # noqa: E731
for _ in range(12):
    step()
torch.cuda.synchronize()
print("labels", int(labels.sum()), dsp_amd.classify_stats_f64(0))
