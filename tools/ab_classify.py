#!/usr/bin/env python3
"""A/B timing of classify() across libdsp_amd.so builds in ONE process, interleaved rounds (as tools/ab.py does for the MFCC
kernels).   python tools/ab_classify.py [--clips N] lib1.so lib2.so ...
Workload = bench.py --workload classify: 1 s clips of low noise, every fourth with a call-like burst pattern.  Labels of every
variant are compared with the first one's."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--clips", type=int, default=49152)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    import torch
    from tests import signals as S

    n = args.clips
    clips = (torch.rand((n, 16000), device="cuda") * 2 - 1) * 0.05
    call = torch.from_numpy(S.classify_cases()["scrub_a"]).cuda()
    clips[::4] = call + clips[::4] * 0.01
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs = []
    for v in args.variants:
        L = C.CDLL(os.path.abspath(v))
        L.dsp_last_error.restype = C.c_char_p
        L.dsp_classify_batch_device.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_void_p]
        libs.append((v, L, torch.empty(n, dtype=torch.int32, device="cuda")))

    def run(i):
        v, L, lab = libs[i]
        rc = L.dsp_classify_batch_device(clips.data_ptr(), n, 16000, 16000, lab.data_ptr(), stream)
        assert rc == 0, L.dsp_last_error()

    for i in range(len(libs)):
        run(i)
        run(i)
    torch.cuda.synchronize()
    for v, _, lab in libs[1:]:
        print(f"# {v}: labels equal to the first variant's: {bool(torch.equal(lab, libs[0][2]))}, label 1: {int(lab.sum())}")
    times = [[] for _ in libs]
    for r in range(args.rounds):
        for i in range(len(libs)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / args.iters)
    for (v, *_), t in zip(libs, times):
        med = statistics.median(t)
        print(f"{v:44s} median {med:.4f} ms  min {min(t):.4f} ms  {n / med / 1e3:8.3f} Mclips/s  {n * 64004 / med / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
