#!/usr/bin/env python3
"""Development aid: what does a short idle gap (a synchronize between warmup and the timed region, as bench.py's contract demands) cost
the next K launches of the headline kernel?  Steady state, then sync + sleep(gap), then K launches with an event each."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import dsp_amd
    plan = dsp_amd.MfccPlan(dsp_amd.default_config(frame_length=512, hop_length=512), 0)
    x = torch.rand((1_000_000, 512), device="cuda") * 2 - 1
    out = torch.empty((1_000_000, 13), device="cuda")
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for _ in range(1500):
        plan.frames(x, out)
    torch.cuda.synchronize()
    for gap in (-1.0, 0.0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 1e-1):
        avgs, firsts, lasts = [], [], []
        for rep in range(7):
            for _ in range(300):
                plan.frames(x, out)
            if gap >= 0:
                torch.cuda.synchronize()
                if gap > 0:
                    t_end = time.perf_counter() + gap
                    while time.perf_counter() < t_end:
                        pass
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
            evs[0].record()
            for i in range(K):
                plan.frames(x, out)
                evs[i + 1].record()
            torch.cuda.synchronize()
            per = [evs[i].elapsed_time(evs[i + 1]) for i in range(K)]
            avgs.append(sum(per) / K); firsts.append(per[0]); lasts.append(statistics.median(per[K // 2:]))
        print(f"gap {'none (no sync)' if gap < 0 else f'{gap * 1e3:7.2f} ms':>15s}: avg of {K} launches median {statistics.median(avgs):.4f} ms (min {min(avgs):.4f}), "
              f"first launch {statistics.median(firsts):.4f}, second half {statistics.median(lasts):.4f}", flush=True)


if __name__ == "__main__":
    main()
