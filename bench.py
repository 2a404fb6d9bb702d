#!/usr/bin/env python3
"""bench.py -- MFCC frames/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py                                   # 1 GPU, defaults finish in ~1-2 min
    python bench.py --gpus N                          # starts N fresh rank processes itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (window -> 512-pt FFT -> power -> 40 mel ->
log -> 13-coef DCT-II) over one batch of synthetic frames already resident in HBM:
BASELINE config 2, 1 M x 512 fp32 frames per GPU.  Frames shard embarrassingly
over ranks (weak scaling, no data-path collective) -- that is `value`.  The path's one
real exchange step, BASELINE config 4's gather of the per-clip MFCC matrices, is
measured beside it in the same line (`config4`: 12 500 x 1 s clips per rank per step,
[98][13] per clip, ONE RCCL all-gather per batch, double-buffered so that the gather of
batch k overlaps the kernels of batch k + 1; dsp_amd/dist.py GatherPipeline).  --gather
adds the same pipelined gather to the frames workload.

The line explains itself: `sensors` holds shader clock, package power and temperatures read in-process from sysfs (amdgpu
hwmon; tools/gpu_sensors.py) before, during and after the timed region -- the headline kernel runs at the package power
cap, so its time follows the clock the power manager grants, which differs from box to box -- and `roofline` carries the
minimum and median of the per-launch times beside their average.

Rank 0 prints ONE JSON line.  `value` = frames all ranks processed / max-over-
ranks wall time of the K timed steps.  `roofline.achieved` = algorithmic bytes
per launch (2100 B/frame) / average kernel time measured with HIP events on the
launch stream.  `cpu_baseline` (rank 0, N=1 only) times the reference's own
compiled compute_mfcc (oracle/_ref, kind "reference") or, where that is absent,
the CPU oracle (kind "port") on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAME = 512
N_MFCC = 13
BYTES_PER_FRAME = FRAME * 4 + N_MFCC * 4      # 2100 algorithmic bytes per frame
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8.0 TB/s spec


def host_cpu_share() -> int:
    """Host threads this process may really use: cgroup quota if set, else affinity, capped."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "16"))))


def pmc_traffic(workload: str, units_per_launch: int):
    """HBM bytes per step from the rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE in separate runs,
    gfx950 x2 read correction), as condensed by tools/traffic.py into profiles/*_traffic*.json.  The newest file for the
    workload and batch size wins (r02 after r01, v12 after v9)."""
    import glob
    import re
    best = None
    natural = lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]   # noqa: E731
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic*.json")), key=natural):
        try:
            d = json.load(open(f))
            if d.get("workload", "frames") == workload and d.get("units_per_launch", d.get("frames_per_launch")) == units_per_launch:
                best = dict(d, file=os.path.basename(f))
        except Exception:  # noqa: BLE001
            pass
    return best


def sq_fractions(workload: str):
    """VALU / LDS busy fractions of the workload's dominant kernel from the committed SQ-counter pass (profiles/*_sq_counters*.json,
    written by tools/sq_fractions.py on the GPU box: vector instructions x 4 issue cycles, and LDS-array cycles, over the dispatch's own
    cycles x 256 CUs (x 4 SIMDs); the clock the kernel ran at comes with them).
    Replayed, like the traffic figure: the line says so."""
    import glob
    import re
    best = None
    natural = lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]   # noqa: E731
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters*.json")), key=natural):
        try:
            d = json.load(open(f))
            if d.get("workload") == workload:
                best = dict(d, file=os.path.basename(f))
        except Exception:  # noqa: BLE001
            pass
    return best


def roofline_extras(workload: str, units: int):
    """The replayed parts of the roofline object: HBM traffic of one step (PMC) and the VALU / LDS fractions, each with its source."""
    tr, sq = pmc_traffic(workload, units), sq_fractions(workload)
    out = {"traffic": tr["hbm_bytes_per_launch"] if tr else None,
           "traffic_replayed": bool(tr),
           "traffic_source": ("NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the same command, profiles/" + tr["file"]) if tr else None}
    if sq:
        out.update({"valu_frac": sq.get("valu_frac"), "lds_frac": sq.get("lds_frac"), "clock_ghz": sq.get("clock_ghz"), "sq_source": "replayed: profiles/" + sq["file"]})
    return out


_REF_WORKER = r"""
import ctypes, sys, time
import numpy as np
lib, n, seconds, seed = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
L = ctypes.CDLL(lib)
F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
L.compute_mfcc.argtypes = [F, ctypes.c_int, F, ctypes.c_int]
L.compute_mfcc.restype = ctypes.c_int
sig = np.random.default_rng(seed).uniform(-1, 1, 400 + 160 * (n - 1)).astype(np.float32)
out = np.empty(n * 13, np.float32)
L.compute_mfcc(sig, sig.size, out, n)                 # page everything in
print("READY", flush=True)
sys.stdin.readline()                                  # all workers start together
got, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < seconds:
    got += L.compute_mfcc(sig, sig.size, out, n)
print(got, time.perf_counter() - t0, flush=True)
"""


def reference_all_cores(ncpu: int, seconds: float = 6.0):
    """The reference's own compute_mfcc on every host core: its static scratch (mfcc.c:21-22) forbids threads, not processes, so
    `ncpu` child PROCESSES each load the compiled reference (oracle/_ref) and run it on a private clip for `seconds`; frames of all
    of them / the longest child's time.  The children never touch the GPU."""
    import subprocess
    from oracle import oracle as O
    lib = os.path.join(os.path.dirname(os.path.abspath(O.__file__)), "_ref", "libref_mfcc.so")
    n = 20_000                                            # frames per pass: ~0.2 s, so every child stops close to `seconds`
    procs = [subprocess.Popen([sys.executable, "-c", _REF_WORKER, lib, str(n), str(seconds), str(100 + i)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True) for i in range(ncpu)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "READY":
                raise RuntimeError("reference worker did not start")
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [p.stdout.readline().split() for p in procs]
        frames = sum(int(r[0]) for r in res)
        wall = max(float(r[1]) for r in res)
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:  # noqa: BLE001
                pass
            p.wait(timeout=30)
    return {"value": frames / wall, "unit": "frames/s", "cores": ncpu, "kind": "reference",
            "sample": f"{frames} frames: {ncpu} processes x the reference's compute_mfcc (compiled -O2 from its own mfcc.c) over a "
                      f"{400 + 160 * (n - 1)}-sample clip each (frame 400 / hop 160, n_fft 512), {wall:.1f} s"}


def cpu_baseline(sample_frames: int, seconds_budget: float = 25.0):
    """Time the CPU path on this host.  Only bench.py's baseline leg touches oracle/."""
    import numpy as np
    from oracle import oracle as O

    rng = np.random.default_rng(7)
    res = {}
    ncpu = host_cpu_share()
    # (a) the reference's own compute_mfcc (400/160 framing over one long clip: the same
    #     512-point FFT / 40-mel / 13-coef work per frame), single thread: it keeps
    #     static scratch (mfcc.c:21-22) and is not re-entrant.
    if O.have_ref():
        n = min(sample_frames, 400_000)
        sig = rng.uniform(-1, 1, 400 + 160 * (n - 1)).astype(np.float32)
        import ctypes as C
        L = O.ref_mfcc_lib()
        out = np.empty((n, 13), np.float32)
        reps, got, t0 = 0, 0, time.perf_counter()
        while reps < 1 or time.perf_counter() - t0 < 0.3 * seconds_budget:      # ~8 s of single-thread work
            got += L.compute_mfcc(sig, sig.size, out.reshape(-1), n)
            reps += 1
        dt = time.perf_counter() - t0
        res = {"value": got / dt, "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": f"{got} frames ({reps} passes over one {sig.size}-sample clip, frame 400 / hop 160, n_fft 512) through "
                         f"the reference's compute_mfcc compiled -O2 from its own mfcc.c, {dt:.1f} s"}
        try:
            res["reference_all_cores"] = reference_all_cores(ncpu, 0.25 * seconds_budget)
        except Exception as exc:  # noqa: BLE001
            res["reference_all_cores"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    # (b) the oracle (restatement) on the bench's own frame shape, all host cores
    cfg = O.default_cfg(frame_length=FRAME, hop_length=FRAME)
    n = min(sample_frames, 60_000 * max(1, ncpu))
    fr = rng.uniform(-1, 1, (n, FRAME)).astype(np.float32)
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or time.perf_counter() - t0 < 0.3 * seconds_budget:
        O.mfcc_frames(fr, cfg, threads=ncpu)
        reps += 1
    dt = time.perf_counter() - t0
    port = {"value": n * reps / dt, "unit": "frames/s", "cores": ncpu, "kind": "port",
            "sample": f"{reps} passes over {n} x {FRAME}-sample frames through the CPU oracle on {ncpu} threads, {dt:.1f} s"}
    if not res:
        return port
    res["port_all_cores"] = port
    return res


def open_sensors(local: int):
    """sysfs sensors of this rank's GPU (None where the box does not expose them)."""
    try:
        from tools.gpu_sensors import Sensors
        s = Sensors.for_device(local)
        return s if s.available else None
    except Exception:  # noqa: BLE001
        return None


def sensor_block(sens, before, during, after):
    if sens is None:
        return None
    return {"source": "amdgpu hwmon (sysfs), read in-process; the SMU low-pass filters these readings over ~0.3 s",
            "power_cap_w": sens.power_cap_w(), "before": before, "during": during, "after": after}


def step_stats(events, torch):
    """Per-launch times from an event recorded after every step: average over the region, minimum, median."""
    import statistics
    per = [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]
    return {"avg": events[0].elapsed_time(events[-1]) / len(per), "min": min(per), "median": statistics.median(per), "max": max(per)}


class Watchdog:
    """A secondary leg that talks to other ranks can hang when ONE rank fails (the others wait in a collective).  The headline is
    already measured when such a leg starts: after `seconds` every rank leaves the process, rank 0 printing the stashed line
    first.  Cancelled when the leg returns."""

    def __init__(self, seconds: float, rank: int, stash_line):
        import threading
        self.rank, self.stash_line = rank, stash_line
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.seconds = seconds

    def fire(self, reason: str = None):
        if self.rank == 0:
            line = self.stash_line()
            line["config4"] = {"error": reason or f"no result after {self.seconds:.0f} s (a rank failed or a collective hung); the headline above was measured before this leg"}
            print(json.dumps(line), flush=True)
        os._exit(0 if self.rank == 0 else 3)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def side_workload(args, torch, dist, dsp_amd, dev, local, rank, world):
    """Secondary measurements (not the headline metric): same timing protocol, its own JSON line."""
    gen = torch.Generator(device=dev).manual_seed(2000 + rank)
    side_finish = None                 # drains a pipelined gather at the end of a region (config 5 at N > 1)
    settle_step = None                 # the step without its collective: the settle loop runs for a TIME, so ranks do different counts of it
    if args.workload == "pcm16":
        n = args.clips or 1_000_000
        pcm = torch.randint(-32768, 32768, (n, FRAME), dtype=torch.int16, device=dev, generator=gen)
        plan = dsp_amd.MfccPlan(dsp_amd.default_config(frame_length=FRAME, hop_length=FRAME), local)
        out = torch.empty((n, 1, 13), device=dev)
        step = lambda: plan.clips_pcm16(pcm, 1, out=out)       # noqa: E731
        units, unit, bytes_per = n, "frames/s", FRAME * 2 + 52
        what = f"{n} x 512-sample int16 mono frames (PCM16 ingestion in the kernel's load, SURVEY 8f-1), otherwise configs[1]"
        kernel = "mfcc512_wave_kernel<IN=1>"
    elif args.workload in ("config5", "config5_2048"):
        import numpy as np
        from dsp_amd.scrubjay import ScrubJay, scrubjay_infer_config
        n = args.clips or 125_000                              # 1 M clips over 8 GPUs
        clips = torch.rand((n, 16000), device=dev, generator=gen) * 2 - 1
        attrs = dict(np.load(os.path.join(ROOT, "tests", "golden", "scrubjay_svm.npz")))
        own = args.workload == "config5_2048"                  # scrubjay_infer.c:10-14's own framing: WIN_SIZE 2048, HOP_SIZE 1024
        sj = ScrubJay(attrs, local, config=scrubjay_infer_config(16000)) if own else ScrubJay(attrs, local)
        # N > 1: the per-clip results -- int32 label + fp32 probability, 8 B per clip, 1 MB per rank at 125 000 clips --
        # are all-gathered every step, pipelined like config 4's matrices (dist.GatherPipeline on a packed [n][2] int32 block)
        from dsp_amd.dist import GatherPipeline
        lab_pipe = GatherPipeline(n, (2,), torch.int32, dev) if world > 1 else None

        def step():
            if lab_pipe is None:
                sj(clips, 500, fused=True)
                return

            def compute(block):
                labels, _dec, p1, _feat = sj(clips, 500, fused=True)
                block[:, 0].copy_(labels)
                block[:, 1].copy_(p1.view(torch.int32))
            lab_pipe.submit(compute)
        side_finish = lab_pipe.drain if lab_pipe is not None else None
        settle_step = lambda: sj(clips, 500, fused=True)       # noqa: E731  (a gather in the time-based settle loop would hang: unequal counts)
        units, unit, bytes_per = n, "clips/s", 64_000 + 8      # label + probability out
        what = (f"BASELINE configs[4] per-GPU share ({n} clips): 1 s 16 kHz fp32 clip -> MFCC(20) -> mean|std -> Scaler -> RBF-SVM "
                "(scrubjay_svm.onnx attributes) fused in ONE kernel, one wavefront per clip; the MFCC matrix never reaches HBM"
                + ("; the front end cepstrum/scrubjay_infer.c itself runs, aubio semantics (dsp_mfcc_scrubjay_infer_config: streaming 2048 / 1024 "
                   "frames with zero history and the padded last hop = 16 frames per clip, magnitude spectrum, 40-filter Slaney bank, log10, "
                   "20 coefficients; restated from aubio 0.4, parity unpinned)" if own else "")
                + (f"; labels + probabilities (8 B per clip) all-gathered over {world} ranks every step, pipelined" if world > 1 else ""))
        kernel = "mfcc2048_kernel<POOL, AUB>" if own else "mfcc512_wave_kernel<POOL>"
    elif args.workload in ("config5_ragged", "classify_ragged"):
        # ragged batches (the reference's callers loop over files of different lengths): clips of 0.5 - 1.5 s back to back in
        # one buffer, ONE call; the same total samples as the uniform workload beside it, so the two lines compare directly
        import numpy as np
        fused = args.workload == "config5_ragged"
        n = args.clips or (125_000 if fused else 49152)
        rng = np.random.default_rng(1234 + rank)
        lens = rng.integers(8000, 24001, n)
        lens[-1] += 16000 * n - int(lens.sum()) if abs(16000 * n - int(lens.sum())) < 8000 else 0
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        total = int(off[-1])
        from dsp_amd import lib as L
        c_off = L.c_offsets(off)
        if fused:
            from dsp_amd.scrubjay import ScrubJay
            flat = torch.rand(total, device=dev, generator=gen) * 2 - 1
            sj = ScrubJay(dict(np.load(os.path.join(ROOT, "tests", "golden", "scrubjay_svm.npz"))), local, n_mfcc=20)
            step = lambda: sj.ragged(flat, c_off, 500)        # noqa: E731
            units, unit, bytes_per = n, "clips/s", 4.0 * total / n + 8
            what = (f"{n} fp32 clips of 0.5 - 1.5 s ({total} samples in one buffer, offsets[n + 1]) -> MFCC(20) -> mean|std -> Scaler -> RBF-SVM in ONE launch "
                    "of the fused clip kernel (dsp_scrubjay_fused_ragged_device); config5's bytes, ragged")
            kernel = "mfcc512_wave_kernel<POOL> (ClipCursor over spans)"
        else:
            from tests import signals as S
            flat = (torch.rand(total, device=dev, generator=gen) * 2 - 1) * 0.05
            call = torch.from_numpy(S.classify_cases()["scrub_a"]).to(dev)
            for c in range(0, n, 4):                           # every fourth clip carries (the head of / a repetition of) the call pattern
                m = int(lens[c])
                seg = call.repeat(2)[:m]
                flat[int(off[c]):int(off[c]) + m] = seg + flat[int(off[c]):int(off[c]) + m] * 0.01
            labels = torch.empty(n, dtype=torch.int32, device=dev)
            step = lambda: dsp_amd.classify_device_ragged(flat, c_off, labels)   # noqa: E731
            units, unit, bytes_per = n, "clips/s", 4.0 * total / n + 4
            what = (f"{n} fp32 clips of 0.5 - 1.5 s ({total} samples in one buffer, offsets[n + 1]; every fourth with the call pattern) through classify() "
                    "in ONE call (dsp_classify_batch_ragged_device); classify's bytes, ragged")
            kernel = "iir2_ckpt_kernel<RAGGED> + spec_from_ckpt_kernel<flags> + classify_midpoints_kernel + spec_from_ckpt_kernel<[time][bin]> + classify_bands_kernel"
    elif args.workload == "stop":
        import numpy as np
        n = args.clips or 125_000
        clips = torch.rand((n, 16000), device=dev, generator=gen) * 2 - 1
        net = dsp_amd.StopModel(dict(np.load(os.path.join(ROOT, "tests", "golden", "stop_model.npz"))), local)
        plan = dsp_amd.MfccPlan(dsp_amd.default_config(), local)
        step = lambda: net.classify_signal_batch(plan, clips)   # noqa: E731
        units, unit, bytes_per = n, "clips/s", 64_000 + 4      # probability out
        two = bool(os.environ.get("DSP_AMD_STOP_TWO_KERNELS"))
        what = (f"classify_signal (stop_detector.c:12-55) on {n} x 1 s 16 kHz fp32 clips: MFCC(13) -> StandardScaler -> 6500-4-2-2-1 net -> P(stop), "
                + ("two kernels (MFCC matrix written, then the net)" if two else "ONE kernel (the tile epilogue feeds layer 1; the MFCC matrix never reaches HBM)"))
        kernel = "mfcc512_wave_kernel + stop_tail_kernel" if two else "mfcc512_wave_kernel<POOL = 2>"
    elif args.workload == "config3":
        n = args.clips or 10_000_000                          # BASELINE configs[2]'s own size: 41 GB of frames in HBM (--clips 1000000: round 3's line)
        frames = torch.empty((n, 1024), device=dev)
        for i0 in range(0, n, 1_000_000):                      # filled in pieces: torch.rand's temporaries for 41 GB at once would double it
            frames[i0:i0 + 1_000_000] = torch.rand((min(1_000_000, n - i0), 1024), device=dev, generator=gen) * 2 - 1
        plan = dsp_amd.MfccPlan(dsp_amd.default_config(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128, prefilter=2), local)
        out = torch.empty((n, 13), device=dev)
        step = lambda: plan.frames(frames, out)               # noqa: E731
        units, unit, bytes_per = n, "frames/s", 4096 + 52      # 4 148 B per frame
        what = (f"BASELINE configs[2] at {n} frames: float64 Butterworth 3000-7500 Hz per 1024-sample frame from zero state "
                "-> Hann(1024) -> 1024-pt FFT -> 128 HTK mel -> dB -> 13 coeffs")
        kernel = "mfcc1024_wave_kernel<PRE> (float64 prefilter scan fused in)"
    elif args.workload == "clips":
        n = args.clips or 12_500
        clips = torch.rand((n, 16000), device=dev, generator=gen) * 2 - 1
        plan = dsp_amd.MfccPlan(dsp_amd.default_config(), local)
        out = torch.empty((n, 98, 13), device=dev)
        step = lambda: plan.clips(clips, 500, out)            # noqa: E731
        units, unit, bytes_per = n * 98, "frames/s", 64_000 + 98 * 52    # 69 096 B per clip
        what = f"BASELINE configs[3] per-GPU share: {n} x 1 s 16 kHz fp32 clips, frame 400 / hop 160 -> [98][13] per clip"
        kernel = "mfcc512_wave_kernel"
    elif args.workload in ("classify_f64", "classify_f64_pcm16"):
        # the float64 classifier of donut-classifier/classifier.c
        n = args.clips or 49152
        # this classifier's midpoint threshold is 45 dB (classifier.c:660; classifier.cpp's is 70): noise of amplitude 0.05 has
        # loud bins in every frame and a midpoint in every clip; 0.005 stays below, so that -- as in the float32 workload --
        # every fourth clip (the call) reaches the spectrogram of the second filter and the band sums
        clips = (torch.rand((n, 16000), device=dev, generator=gen, dtype=torch.float64) * 2 - 1) * 0.005
        from tests import signals as S
        call = torch.from_numpy(S.classify_cases()["scrub_a"]).to(dev).double()
        clips[::4] = call + clips[::4] * 0.1
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        kernel = ("iir2_screen_f64_kernel (both recurrences, restart states, bf16-MFMA screening of the loud bins) + spec_f64_from_ckpt_kernel<flags> "
                  "(undecided segments) + classify_f64_midpoints_kernel + spec_f64_from_ckpt_kernel<maps> + classify_f64_bands_kernel")
        if args.workload == "classify_f64_pcm16":
            pcm = torch.clamp(torch.round(clips * 32768.0), -32768, 32767).to(torch.int16)      # what the reference's reader starts from (classifier.c:55-59)
            del clips
            step = lambda: dsp_amd.classify_device_f64_pcm16(pcm, labels)   # noqa: E731
            units, unit, bytes_per = n, "clips/s", 32_000 + 4
            what = (f"{n} x 1 s 16 kHz int16 mono clips (25 % with a call-like burst pattern), s / 32768.0 in the kernels' loads, through the float64 "
                    "classify() of donut-classifier/classifier.c")
        else:
            step = lambda: dsp_amd.classify_device_f64(clips, labels)   # noqa: E731
            units, unit, bytes_per = n, "clips/s", 128_000 + 4
            what = (f"{n} x 1 s 16 kHz float64 clips (25 % with a call-like burst pattern) through the float64 classify() of "
                    "donut-classifier/classifier.c")
    elif args.workload == "classify_pcm16":
        n = args.clips or 49152
        clips = (torch.rand((n, 16000), device=dev, generator=gen) * 2 - 1) * 0.05
        from tests import signals as S
        call = torch.from_numpy(S.classify_cases()["scrub_a"]).to(dev)
        clips[::4] = call + clips[::4] * 0.01
        pcm = torch.clamp(torch.round(clips * 32768.0), -32768, 32767).to(torch.int16)          # sync/sync.cpp:237-242: int16 capture, / 32768.0
        del clips
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        step = lambda: dsp_amd.classify_device_pcm16(pcm, labels)   # noqa: E731
        units, unit, bytes_per = n, "clips/s", 32_000 + 4
        what = (f"{n} x 1 s 16 kHz int16 mono clips (25 % with a call-like burst pattern), pcmSample / 32768.0 in the kernels' loads, through classify() "
                "(2 x IIR, 2 x spectrogram, rule), bit-exact with the reference")
        kernel = "iir2_ckpt_kernel<IN=1> + spec_from_ckpt_kernel<flags, IN=1> + classify_midpoints_kernel + spec_from_ckpt_kernel<[time][bin], IN=1> + classify_bands_kernel"
    else:
        n = args.clips or 49152
        clips = (torch.rand((n, 16000), device=dev, generator=gen) * 2 - 1) * 0.05
        # every fourth clip carries a call-like burst pattern that has midpoints and fires the rule (tests/signals.py
        # scrub_a, label 1 in the reference): the band-pass spectrogram and the band sums run only for clips with
        # midpoints, so an all-noise batch would flatter it
        from tests import signals as S
        call = torch.from_numpy(S.classify_cases()["scrub_a"]).to(dev)
        clips[::4] = call + clips[::4] * 0.01
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        step = lambda: dsp_amd.classify_device(clips, labels)   # noqa: E731
        units, unit, bytes_per = n, "clips/s", 64_000 + 4
        what = (f"{n} x 1 s 16 kHz fp32 clips (25 % with a call-like burst pattern, label 1) through classify() "
                "(2 x IIR, 2 x spectrogram, rule), bit-exact with the reference")
        kernel = "iir2_ckpt_kernel + spec_from_ckpt_kernel<flags> + classify_midpoints_kernel + spec_from_ckpt_kernel<[time][bin]> + classify_bands_kernel"
    sens = open_sensors(local)
    settle(settle_step or step, torch, args.settle)
    for _ in range(max(1, args.warmup // 4)):
        step()
    if side_finish:
        side_finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    steps = max(1, args.steps // 10) if args.workload.startswith("classify") else args.steps
    s_before = sens.read() if sens else None
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    s_during = sens.read() if sens else None
    if side_finish:
        side_finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    s_after = sens.read() if sens else None
    kstats = step_stats(evs, torch)
    ms = kstats["avg"]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        achieved = bytes_per * n / (ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": f"{args.workload}: {unit}", "value": world * units * steps / elapsed, "unit": unit, "n_gpus": world,
            "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.workload.startswith("classify_f64") else "f32", "data": "synthetic",
            "config": {"workload": what, "clock_settle_s": args.settle},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         **roofline_extras(args.workload, n),
                         "kernel": kernel, "kernel_ms": ms, "kernel_ms_min": kstats["min"], "kernel_ms_median": kstats["median"],
                         "algorithmic_bytes_per_launch": bytes_per * n},
            "sensors": sensor_block(sens, s_before, s_during, s_after),
            "step_calls": (max(1, args.warmup // 4) + steps) if args.settle <= 0 else None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def config4(args, torch, dist, dsp_amd, dev, local, rank, world, sync, watchdog_line=None):
    """BASELINE configs[3] beside the headline number: every rank computes its 12 500-clip share ([98][13] per clip) and the
    per-rank feature blocks (63.7 MB each) are all-gathered -- ONE RCCL collective per batch, issued asynchronously so that it
    runs over xGMI while the next batch's MFCC kernel runs (GatherPipeline, depth 2).  Same timing protocol as the headline:
    W warmup steps, K timed steps between barrier + synchronize, max over ranks.  At N > 1 the same run also times the
    compute alone (no collective) and the gather alone (K back-to-back all-gathers of the same blocks), so the line says how much
    of the gather the pipeline hides: gather_hidden_frac = 1 - (pipelined - compute_only) / gather_only.  At N = 1 there is
    nothing to gather and the number is the compute-only baseline of the same workload.

    Failure handling (N > 1): the setup phase makes no collective call, and the ranks agree on its outcome with one all-reduce
    before the first gather, so a rank that cannot allocate does not leave the others waiting in a collective; the timed phase
    runs under a watchdog (see Watchdog) because there a failure of one rank cannot be seen by the others."""
    from dsp_amd.dist import GatherPipeline
    n, T = args.clips or 12_500, 98
    err = None
    try:      # ---- setup: local work only
        gen = torch.Generator(device=dev).manual_seed(3000 + rank)
        clips = torch.rand((n, 16000), device=dev, generator=gen) * 2 - 1
        plan = dsp_amd.MfccPlan(dsp_amd.default_config(), local)
        pipe = GatherPipeline(n, (T, N_MFCC), torch.float32, dev)
        plan.clips(clips, 500, pipe.local[0])
        torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001
        err = f"{type(exc).__name__}: {exc}"[:300]
    if world > 1:
        ok = torch.tensor([0 if err else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return {"error": err or "another rank failed in the setup of this leg"} if rank == 0 else None
    elif err:
        return {"error": err}

    def timed(step_fn, finish=None):
        for _ in range(args.warmup):
            step_fn()
        if finish:
            finish()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        if finish:
            finish()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    import contextlib
    guard = Watchdog(float(os.environ.get("BENCH_CONFIG4_TIMEOUT", "120")), rank, watchdog_line) if (world > 1 and watchdog_line) else contextlib.nullcontext()
    with guard:
        elapsed = timed(lambda: pipe.submit(lambda block: plan.clips(clips, 500, block)), pipe.drain)
        full = pipe.result((pipe.k - 1) % pipe.depth)
        assert tuple(full.shape) == (world * n, T, N_MFCC)
        compute_only = gather_only = None
        if world > 1:
            compute_only = timed(lambda: plan.clips(clips, 500, pipe.local[0]))
            pipe.drain()
            work = []

            def gather_step():
                work.append(dist.all_gather_into_tensor(pipe.gathered[len(work) % pipe.depth], pipe.local[len(work) % pipe.depth], async_op=True))

            def gather_finish():
                while work:
                    work.pop().wait()

            gather_only = timed(gather_step, gather_finish)
    del clips
    if rank != 0:
        return None
    out = {"workload": f"BASELINE configs[3]: {world} x {n} x 1 s 16 kHz fp32 clips per step, frame 400 / hop 160 -> [98][13] per clip, "
                       "per-clip MFCC matrices gathered on every rank",
           "value": world * n * args.steps / elapsed, "unit": "clips/s", "frames_per_s": world * n * T * args.steps / elapsed,
           "ms_per_step": elapsed / args.steps * 1e3, "gather": world > 1,
           "collective": "one all_gather_into_tensor (RCCL over xGMI) per batch, async on the backend's stream, double-buffered: "
                         "the gather of batch k overlaps the MFCC kernel of batch k + 1" if world > 1 else None,
           "gathered_bytes_per_rank_per_step": n * T * N_MFCC * 4, "clips_per_gpu": n, "algorithmic_bytes_per_clip": 64_000 + T * 52,
           "compute_only_ms_per_step": None, "gather_only_ms_per_step": None, "gather_hidden_frac": None}
    if world > 1:
        c_ms, g_ms, p_ms = compute_only / args.steps * 1e3, gather_only / args.steps * 1e3, elapsed / args.steps * 1e3
        out.update(compute_only_ms_per_step=c_ms, gather_only_ms_per_step=g_ms,
                   gather_hidden_frac=max(0.0, min(1.0, 1.0 - (p_ms - c_ms) / g_ms)) if g_ms > 0 else None)
    return out


def settle(step, torch, seconds):
    """Untimed: keep the GPU busy with the workload's own step until its clocks have ramped (see --settle)."""
    t_end = time.perf_counter() + max(0.0, seconds)
    while time.perf_counter() < t_end:
        for _ in range(20):
            step()
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=float, default=0.5,
                    help="seconds of untimed back-to-back steps before the W warmup steps: a step is 0.45 ms, so W = 10..20 "
                         "steps end before the GPU's clocks have ramped (first launches after idle run 0.6 ms); reported in config")
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU per step")
    ap.add_argument("--gather", action="store_true", help="all-gather the per-rank features every step (RCCL)")
    ap.add_argument("--workload", choices=["frames", "clips", "classify", "classify_pcm16", "classify_f64", "classify_f64_pcm16", "classify_ragged", "config3", "config5", "config5_2048", "config5_ragged", "pcm16", "stop"], default="frames",
                    help="frames = BASELINE configs[1] (the headline metric, default); clips = configs[3] per-GPU share "
                         "(12 500 x 1 s clips, reference framing 400/160); classify = the donut classifier on 1 s clips")
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU per step for --workload clips / classify")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 (clips + all-gather) measurement of the frames workload's line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Bare `python bench.py --gpus N`: this process has not touched the GPU (torch is not even imported yet), so it can
        # start N fresh rank processes -- one per GPU, RCCL over xGMI -- as children, relay their output and exit code.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # BENCH_REHEARSAL=1 (one-GPU boxes): every rank on cuda:0 and the gloo backend, to exercise the N > 1 control flow (self
    # launch, barriers, pipelined gather, max over ranks) where no second GPU exists.  The line says so; its numbers mean nothing.
    rehearsal = world > 1 and os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import dsp_amd

    cfg = dsp_amd.default_config(frame_length=FRAME, hop_length=FRAME)   # Hann(512), 512-pt FFT, 40 mel, 13 coef
    plan = dsp_amd.MfccPlan(cfg, local)
    if args.blocks_per_cu or args.chunk:
        plan.set_launch(args.blocks_per_cu, args.chunk)

    if args.workload != "frames":
        return side_workload(args, torch, dist, dsp_amd, dev, local, rank, world)

    n = args.frames
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    frames = torch.rand((n, FRAME), device=dev, generator=gen) * 2 - 1    # synthetic PCM in [-1, 1)
    out = torch.empty((n, N_MFCC), device=dev)
    do_gather = bool(args.gather and world > 1)
    from dsp_amd.dist import GatherPipeline
    pipe = GatherPipeline(n, (N_MFCC,), torch.float32, dev) if do_gather else None

    def step():
        if pipe is not None:     # the all-gather of this batch's features overlaps the next step's kernel (double-buffered)
            pipe.submit(lambda block: plan.frames(frames, block))
        else:
            plan.frames(frames, out)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sens = open_sensors(local)
    settle(lambda: plan.frames(frames, out), torch, args.settle)      # compute only: ranks may do different counts
    for _ in range(args.warmup):
        step()
    sync()
    s_before = sens.read() if sens else None
    # an event after every step (on torch's current stream == the stream plan.frames launches on): the region's average launch
    # time is last - first over K, the per-launch minimum / median come from the pairs (BENCH_STEP_EVENTS=0: two events only)
    every = os.environ.get("BENCH_STEP_EVENTS", "1") != "0"
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1 if every else 2)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        if every:
            evs[i + 1].record()
    if not every:
        evs[1].record()
    s_during = sens.read() if sens else None      # the K launches are queued / running
    if pipe is not None:
        pipe.drain()                   # the last gathers are part of the timed work
    sync()
    elapsed = time.perf_counter() - t0
    s_after = sens.read() if sens else None
    kernel_ms = evs[0].elapsed_time(evs[-1]) / args.steps    # back-to-back launches: avg launch duration
    kstats = step_stats(evs, torch) if every else None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def headline():
        total_frames = world * n * args.steps
        achieved = BYTES_PER_FRAME * n / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "MFCC frames/sec (512-pt FFT, 40 mel, 13 coeffs)",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 1 M synthetic 512-sample fp32 frames per GPU, "
                                   "Hann(512) -> 512-pt FFT -> 40 HTK mel -> per-frame dB -> 13 DCT-II coeffs, "
                                   "inputs resident in HBM; frames sharded over the ranks with no data-path collective "
                                   "(the exchange step of the path, config 4's gather, is the `config4` object of this line)",
                       "frames_per_gpu": n, "frame_length": FRAME, "n_fft": 512, "n_mels": 40, "n_mfcc": N_MFCC,
                       "gather": do_gather, "parallelism": f"frames sharded over {world} rank(s)",
                       "clock_settle_s": args.settle},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "mfcc512_wave_kernel", "kernel_ms": kernel_ms,
                         "kernel_ms_min": kstats["min"] if kstats else None, "kernel_ms_median": kstats["median"] if kstats else None,
                         "kernel_ms_max": kstats["max"] if kstats else None,
                         "algorithmic_bytes_per_launch": BYTES_PER_FRAME * n},
            "sensors": sensor_block(sens, s_before, s_during, s_after),
        }
        line["roofline"].update(roofline_extras("frames", n))
        line["step_calls"] = args.warmup + args.steps if args.settle <= 0 else None      # tools/traffic.py divides by this
        if rehearsal:
            line["rehearsal"] = f"gloo backend, {world} ranks sharing cuda:0 -- control-flow check only, not a measurement"
        return line

    # the headline above is measured; a failure in the secondary config-4 leg must not cost the line (nor hang the job: config4())
    c4 = None
    if not args.no_config4:
        try:
            c4 = config4(args, torch, dist, dsp_amd, dev, local, rank, world, sync, watchdog_line=headline if rank == 0 else (lambda: {}))
        except Exception as exc:  # noqa: BLE001
            c4 = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            if world > 1:
                # this rank failed inside the leg's collective phase: the others may be waiting for it.  Rank 0 still owes the
                # line; every rank then leaves (the others' watchdogs end them)
                Watchdog(0, rank, headline).fire(c4["error"])

    if rank == 0:
        line = headline()
        if c4:
            line["config4"] = c4
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(600_000)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
