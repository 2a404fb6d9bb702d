"""Per-kernel view of the float64 classifier on the bench workload without a profiler: times dsp_classify_batch_device_f64 (default
pipeline and DSP_AMD_F64_PIPELINE=materialize) with CUDA events and prints the screening's statistics."""
import os
import sys
import time

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
pcm = torch.clamp(torch.round(clips * 32768.0), -32768, 32767).to(torch.int16)


def run(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, env in (("ckpt", {}), ("materialize", {"DSP_AMD_F64_PIPELINE": "materialize"})):
    os.environ.pop("DSP_AMD_F64_PIPELINE", None)
    os.environ.update(env)
    ms = run(lambda: dsp_amd.classify_device_f64(clips, labels))
    print(f"{name:12s} float64 input {ms:7.3f} ms per {n} clips, labels {int(labels.sum())}", dsp_amd.classify_stats_f64(0) if not env else "", flush=True)
os.environ.pop("DSP_AMD_F64_PIPELINE", None)
ms = run(lambda: dsp_amd.classify_device_f64_pcm16(pcm, labels))
print(f"{'ckpt':12s} int16 input   {ms:7.3f} ms per {n} clips, labels {int(labels.sum())}", dsp_amd.classify_stats_f64(0), flush=True)
