#!/usr/bin/env python3
"""Development aid: time of the headline launch (BASELINE config 2, 1 M frames) from a cold GPU over a few seconds, with the
shader clock / package power / temperatures sampled from sysfs beside every group of launches (tools/gpu_sensors.py).
Shows the ramp after idle, any boost window and the sustained state the power manager settles in.

    python tools/clock_trace.py [--seconds 4] [--group 20] [--idle 3] [--lib path.so] > gpurun_out/clock_trace.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--group", type=int, default=20)
    ap.add_argument("--idle", type=float, default=3.0)
    ap.add_argument("--frames", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch

    import dsp_amd
    from tools.gpu_sensors import Sensors

    sens = Sensors.for_device(0)
    print(json.dumps({"sensors": {"pci_dir": sens.pci_dir, "hwmon": sens.hwmon, "power_file": sens.power_file, "cap_w": sens.power_cap_w()}}), flush=True)
    plan = dsp_amd.MfccPlan(dsp_amd.default_config(frame_length=512, hop_length=512), 0)
    x = torch.rand((args.frames, 512), device="cuda") * 2 - 1
    out = torch.empty((args.frames, 13), device="cuda")
    plan.frames(x, out)
    torch.cuda.synchronize()
    time.sleep(args.idle)
    print(json.dumps({"idle": sens.read()}), flush=True)
    t_start = time.perf_counter()
    rows = []
    while time.perf_counter() - t_start < args.seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.group):
            plan.frames(x, out)
        e1.record()
        mid = sens.read()                 # sampled while the group runs
        e1.synchronize()
        rows.append({"t_s": round(time.perf_counter() - t_start, 4), "ms": e0.elapsed_time(e1) / args.group, **mid})
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
