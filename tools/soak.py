#!/usr/bin/env python3
"""Determinism soak (development aid): many launches of every MFCC kernel form on the same inputs, every output
compared bit for bit with the first one.  The kernels order their per-wave LDS traffic with wavefront-scope fences
only (no s_barrier), so a rare ordering bug would show up here as a flipped bit.   python tools/soak.py [launches]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402
from dsp_amd.scrubjay import ScrubJay  # noqa: E402


def main():
    n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    gen = torch.Generator(device="cuda").manual_seed(99)
    cases = []
    x = torch.rand((200_000, 512), device="cuda", generator=gen) * 2 - 1
    p = dsp_amd.MfccPlan(dsp_amd.default_config(frame_length=512, hop_length=512))
    cases.append(("frames 512", lambda: p.frames(x)))
    clips = torch.rand((3000, 16000), device="cuda", generator=gen) * 2 - 1
    pc = dsp_amd.MfccPlan()
    cases.append(("clips 400/160", lambda: pc.clips(clips, 500)))
    pcm = torch.randint(-32768, 32768, (3000, 16000), dtype=torch.int16, device="cuda", generator=gen)
    cases.append(("pcm16 clips", lambda: pc.clips_pcm16(pcm, 500)))
    pf = dsp_amd.MfccPlan(dsp_amd.default_config())
    pf.set_kernel(2)
    cases.append(("per-frame epilogue", lambda: pf.clips(clips, 500)))
    attrs = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scrubjay_svm.npz")))
    sj = ScrubJay(attrs)
    cases.append(("fused config 5", lambda: torch.cat([t.float().reshape(clips.shape[0], -1) for t in sj(clips, 500, fused=True)], 1)))
    x1024 = torch.rand((50_000, 1024), device="cuda", generator=gen) * 2 - 1
    p3 = dsp_amd.MfccPlan(dsp_amd.default_config(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128, prefilter=2))
    cases.append(("config 3 (fused scan + 1024)", lambda: p3.frames(x1024)))
    from dsp_amd.scrubjay import scrubjay_infer_config
    sj2k = ScrubJay(attrs, config=scrubjay_infer_config(16000))
    cases.append(("fused 2048/1024/40/20", lambda: torch.cat([t.float().reshape(clips.shape[0], -1) for t in sj2k(clips, 500, fused=True)], 1)))
    mic = dsp_amd.classify_config(dsp_amd.CLASSIFY_MICROPHONE)
    cl = (torch.rand((4096, 16000), device="cuda", generator=gen) * 2 - 1) * 0.05
    # a third of the clips carry call-like patterns (midpoints; some fire the rule): the work list between the midpoints
    # and the band kernels is filled with atomics in any order, the results must not depend on it
    from tests import signals as S
    sig = S.classify_cases()
    for k, name in enumerate(("scrub_a", "jay_like", "scrub_b")):
        cl[k::9] = torch.from_numpy(sig[name]).cuda() + cl[k::9] * 0.01
    lab = torch.empty(4096, dtype=torch.int32, device="cuda")
    host = cl[:512].cpu().numpy()

    def cls_trace():
        labels, trace = dsp_amd.classify_batch(host, with_trace=True)
        flat = [np.concatenate([[l], m, q.ravel()]) for l, (m, q) in zip(labels, trace)]
        return torch.from_numpy(np.concatenate(flat).astype(np.float32))

    def cls():
        dsp_amd.classify_device(cl, lab)
        return lab.clone()
    def cls_mic():
        dsp_amd.classify_device(cl, lab, config=mic)
        return lab.clone()
    cases.append(("classify", cls))
    cases.append(("classify (microphone thresholds)", cls_mic))
    cases.append(("classify + trace (host)", cls_trace))
    bad = 0
    for name, fn in cases:
        ref = fn().clone()
        mism = 0
        for i in range(n_launch):
            out = fn()
            if i % 10 == 9 or i == n_launch - 1:
                if not torch.equal(out, ref):
                    mism += 1
        torch.cuda.synchronize()
        print(f"{name:34s} {n_launch} launches, mismatching checks: {mism}", flush=True)
        bad += mism
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
