#!/usr/bin/env python3
"""A/B timing of libdsp_amd.so builds and launch geometries in ONE process,
interleaved rounds (cdna_hip_programming.md 5.4 rule 24).

    python tools/ab.py [--frames N] [--rounds R] lib1.so[:bpc:chunk[:kernel]] lib2.so[:bpc:chunk[:kernel]] ...

Each variant: frames kernel on BASELINE config 2 (1 M x 512 by default); prints
median / min ms, frames/s and algorithmic GB/s, and checks every variant's
output against the first one.
"""
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
    ap.add_argument("--frames", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--frame-length", type=int, default=512)
    ap.add_argument("--n-fft", type=int, default=512)
    ap.add_argument("--n-mels", type=int, default=40)
    ap.add_argument("--prefilter", type=int, default=0, help="dsp_mfcc_config.prefilter (2 = 3000-7500 Hz Butterworth per frame: BASELINE config 3)")
    args = ap.parse_args()
    import torch
    from dsp_amd.lib import MfccConfig

    n = args.frames
    fl = args.frame_length
    x = torch.rand((n, fl), device="cuda") * 2 - 1
    plans = []
    for v in args.variants:
        parts = v.split(":")
        path = os.path.abspath(parts[0])
        bpc = int(parts[1]) if len(parts) > 1 else 0
        chunk = int(parts[2]) if len(parts) > 2 else 0
        kern = int(parts[3]) if len(parts) > 3 else 0
        L = C.CDLL(path)
        cfg = MfccConfig()
        L.dsp_mfcc_default_config(C.byref(cfg))
        cfg.frame_length = fl
        cfg.hop_length = fl
        cfg.n_fft = args.n_fft
        cfg.n_mels = args.n_mels
        cfg.prefilter = args.prefilter
        h = C.c_void_p()
        L.dsp_last_error.restype = C.c_char_p
        rc = L.dsp_mfcc_plan_create(C.byref(cfg), 0, C.byref(h))
        assert rc == 0, L.dsp_last_error()
        L.dsp_mfcc_plan_set_launch(h, bpc, chunk)
        if kern:
            assert L.dsp_mfcc_plan_set_kernel(h, kern) == 0
        L.dsp_mfcc_frames_device.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        out = torch.empty((n, 13), device="cuda")
        plans.append((v, L, h, out))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(i):
        v, L, h, out = plans[i]
        rc = L.dsp_mfcc_frames_device(h, x.data_ptr(), n, out.data_ptr(), stream)
        assert rc == 0, L.dsp_last_error()

    for i in range(len(plans)):
        run(i)
    torch.cuda.synchronize()
    ref = plans[0][3]
    for v, _, _, out in plans[1:]:
        d = (out - ref).abs().max().item()
        print(f"# {v}: max |out - first| = {d:.3e}")
    times = [[] for _ in plans]
    for r in range(args.rounds):
        for i in range(len(plans)):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / args.iters)
    bytes_per = (fl * 4 + 52) * n
    for (v, *_), t in zip(plans, times):
        med, mn = statistics.median(t), min(t)
        print(f"{v:50s} median {med:.4f} ms  min {mn:.4f} ms  {n / med / 1e6:8.3f} Gframes/s  {bytes_per / med / 1e6:7.1f} GB/s  ({bytes_per / med / 8e9 :.1%} of 8 TB/s)")


if __name__ == "__main__":
    main()
