"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI, against (1) goldens from the reference's compiled mfcc.c, (2) the CPU
oracle on seeded inputs, (3) size-independent properties at BASELINE's full size.

Gate (tests/conftest.py gate()): the PURE |gpu - ref| <= 1e-4 * max(|ref|, ||ref frame||_inf); the absolute floor
of 3e-4 is granted only to named low-level cases, for frames with ||ref||_inf < 3 (where the reference's own noise
exceeds the pure gate).  Every comparison's worst ratio goes to gpurun_out/gate_report.json.
"""
import os

import numpy as np
import pytest

from tests import signals as S
from tests.conftest import LOW_LEVEL_CASES, gate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def dsp():
    import dsp_amd
    return dsp_amd


def _cases(g):
    c = S.mfcc_cases()
    c["chirp"] = g["input__chirp"]
    bird = g["birdq_pcm"]
    c["birdq_ch0"] = (bird[:, 0] / np.float32(32768.0)).astype(np.float32)
    c["birdq_avg"] = (np.float32(0.5) * (bird[:, 0] / np.float32(32768.0) + bird[:, 1] / np.float32(32768.0))).astype(np.float32)
    c["stop_121417"] = (g["stop_pcm"] / np.float32(32768.0)).astype(np.float32)
    return c


ALL = ["noise0", "noise1", "noise2", "chirp", "silence", "tiny", "dc", "impulse", "half_silent",
       "len399", "len400", "len559", "len560", "long", "birdq_ch0", "birdq_avg", "stop_121417"]


@pytest.mark.parametrize("name", ALL)
def test_compute_mfcc_entry_point_vs_reference_goldens(dsp, golden, name):
    """`int compute_mfcc(signal, n, out, max_frames)` exactly as the reference's callers use it."""
    g = golden("mfcc_ref.npz")
    ref = g["mfcc__" + name]
    got = dsp.compute_mfcc(_cases(g)[name], 500)
    assert got.shape == ref.shape
    gate(got, ref, f"golden/{name}", floor_case=name if name in LOW_LEVEL_CASES else None)
    # and against exact arithmetic (float64 FFT): the pure gate, no floor for any case
    from oracle import oracle as O
    truth = O.compute_mfcc(_cases(g)[name][:16000 * 2], 500, O.default_cfg(fft_mode=O.FFT_FLOAT64))
    gate(got[: truth.shape[0]], truth, f"float64-truth/{name}")


def test_max_frames_and_degenerate_arguments(dsp, golden):
    g = golden("mfcc_ref.npz")
    x = S.mfcc_cases()["noise0"]
    got = dsp.compute_mfcc(x, 7)
    assert got.shape == (7, 13)
    gate(got, g["mfcc__noise0_max7"], "golden/noise0_max7")
    assert dsp.compute_mfcc(x, 0).shape[0] == 0
    assert dsp.compute_mfcc(x[:399], 500).shape[0] == 0


def test_silence_is_exactly_zero(dsp):
    out = dsp.compute_mfcc(np.zeros(16000, np.float32), 500)
    assert out.shape == (98, 13) and not out.any()


def test_birdq_is_config_one(dsp, golden):
    """BASELINE config 1: birdQ_stereo_16k.wav channel 0 -> 148 frames."""
    g = golden("mfcc_ref.npz")
    got = dsp.compute_mfcc(_cases(g)["birdq_ch0"], 500)
    assert got.shape == (148, 13)
    gate(got, g["mfcc__birdq_ch0"], "golden/birdq_ch0 (config 1)")


def test_clips_device_path_vs_goldens(dsp, torch_cuda, golden):
    """HBM-resident clips -> [n_clips][T][13]; strided batch incl. a silent clip."""
    torch = torch_cuda
    g = golden("mfcc_ref.npz")
    names = ["noise0", "noise1", "silence", "chirp", "stop_121417", "noise2"]
    clips = np.stack([_cases(g)[n] for n in names])
    plan = dsp.MfccPlan()
    out = plan.clips(torch.from_numpy(clips).cuda(), 500).cpu().numpy()
    assert out.shape == (6, 98, 13)
    for i, n in enumerate(names):
        gate(out[i], g["mfcc__" + n], f"clips-device/{n}")
    # strided view: clips embedded in a wider buffer
    wide = torch.zeros((6, 16000 + 64), device="cuda")
    wide[:, :16000] = torch.from_numpy(clips).cuda()
    out2 = plan.clips(wide[:, :16000], 500).cpu().numpy()
    assert np.array_equal(out, out2)
    # host convenience path gives the same bits
    assert np.array_equal(plan.clips_host(clips, 500), out)


@pytest.mark.parametrize("n_frames", [1, 3, 63, 64, 65, 1000, 4099])
def test_frames_path_vs_oracle(dsp, torch_cuda, n_frames):
    """BASELINE config 2 shape (512-sample frames, Hann(512)) at oracle-friendly sizes, ragged counts."""
    from oracle import oracle as O
    torch = torch_cuda
    cfg = dsp.default_config(frame_length=512, hop_length=512)
    plan = dsp.MfccPlan(cfg)
    fr = S.uniform_pm1(512 * n_frames, 100 + n_frames).reshape(n_frames, 512)
    if n_frames >= 64:
        fr[::37] = 0.0                           # 1 % silent frames in the correctness set
        fr[5] *= 1e-6                            # energies near amin
    out = plan.frames(torch.from_numpy(fr).cuda()).cpu().numpy()
    ref = O.mfcc_frames(fr, O.default_cfg(frame_length=512, hop_length=512), threads=4)
    gate(out, ref, f"frames-vs-oracle/{n_frames}")
    assert np.array_equal(plan.frames_host(fr), out)


def test_empty_batch(dsp, torch_cuda):
    torch = torch_cuda
    plan = dsp.MfccPlan(dsp.default_config(frame_length=512, hop_length=512))
    out = plan.frames(torch.empty((0, 512), device="cuda"))
    assert tuple(out.shape) == (0, 13)
    assert plan.clips(torch.empty((0, 16000), device="cuda"), 500).shape[0] == 0


@pytest.mark.parametrize("over", [dict(n_mels=32), dict(n_mfcc=20), dict(n_mfcc=8, n_mels=24),
                                  dict(window=1), dict(mel_norm=1), dict(fmin=100.0, fmax=6000.0),
                                  dict(frame_length=256, hop_length=128), dict(top_db=40.0)])
def test_other_configurations_vs_oracle(dsp, torch_cuda, over):
    from oracle import oracle as O
    torch = torch_cuda
    cfg = dsp.default_config(**over)
    ocfg = O.default_cfg(**over)
    # the chirp stays inside every configuration's mel band: with the tone outside
    # [fmin, fmax] all mel energies are window-sidelobe leakage at the level of the
    # reference's own fp32 rounding noise and no implementation can match it there
    x = np.stack([S.uniform_pm1(8000, 40), S.chirp(8000, 200.0, 5500.0), S.uniform_pm1(8000, 41) * np.float32(1e-3)])
    plan = dsp.MfccPlan(cfg)
    out = plan.clips(torch.from_numpy(x).cuda(), 500).cpu().numpy()
    for i in range(3):
        ref = O.compute_mfcc(x[i], 500, ocfg)
        assert out[i].shape == ref.shape
        gate(out[i], ref, f"config {over}/clip{i}")


def test_full_size_properties(dsp, torch_cuda):
    """BASELINE config 2 at full size (1 M x 512): properties that need no oracle pass,
    plus an oracle spot check on a random sample of frames."""
    from oracle import oracle as O
    torch = torch_cuda
    n = 1_000_000
    cfg = dsp.default_config(frame_length=512, hop_length=512)
    plan = dsp.MfccPlan(cfg)
    gen = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand((n, 512), device="cuda", generator=gen) * 2 - 1
    x[::101] = 0.0
    a = plan.frames(x)
    b = plan.frames(x)
    assert torch.equal(a, b)                                   # deterministic
    # power-of-two gain leaves every coefficient bit-identical (per-frame max reference)
    c = plan.frames(x * 4.0)
    assert torch.equal(a, c)
    # frames are independent: permuting the input permutes the output
    perm = torch.randperm(n, device="cuda", generator=gen)
    d = plan.frames(x[perm].contiguous())
    assert torch.equal(d, a[perm])
    # silent frames -> exact zeros, nothing non-finite anywhere
    assert not a[::101].any() and bool(torch.isfinite(a).all())
    # different launch geometry -> same bits
    plan.set_launch(2, 7)
    assert torch.equal(plan.frames(x), a)
    plan.set_launch(0, 0)
    # oracle spot check
    idx = torch.randint(0, n, (3000,), device="cuda", generator=gen)
    ref = O.mfcc_frames(x[idx].cpu().numpy(), O.default_cfg(frame_length=512, hop_length=512), threads=8)
    gate(a[idx].cpu().numpy(), ref, "config2 1M frames, 3000-frame oracle sample")


def test_config4_clips_at_scale(dsp, torch_cuda):
    """BASELINE config 4, one GPU's share: 12 500 clips x 16 000 samples -> [12500][98][13].
    Properties (no oracle pass at this size) + oracle spot check of whole clips."""
    from oracle import oracle as O
    torch = torch_cuda
    n = 12_500
    plan = dsp.MfccPlan()
    gen = torch.Generator(device="cuda").manual_seed(77)
    clips = torch.rand((n, 16000), device="cuda", generator=gen) * 2 - 1
    clips[::500] = 0.0                                            # silent clips
    out = plan.clips(clips, 500)
    assert tuple(out.shape) == (n, 98, 13) and bool(torch.isfinite(out).all())
    assert not out[::500].any()
    assert torch.equal(out, plan.clips(clips, 500))               # deterministic
    # a clip's features do not depend on its neighbours or position in the batch
    perm = torch.randperm(n, device="cuda", generator=gen)
    assert torch.equal(plan.clips(clips[perm].contiguous(), 500), out[perm])
    # frame t of a clip == the same 400 samples presented as an independent frame
    fplan = dsp.MfccPlan(dsp.default_config(frame_length=400, hop_length=400))
    idx = torch.tensor([1, 777, 12_499], device="cuda")
    frames = clips[idx].unfold(1, 400, 160).reshape(-1, 400).contiguous()
    assert torch.equal(fplan.frames(frames).reshape(3, 98, 13), out[idx])
    # max_frames clamp on a batch
    assert torch.equal(plan.clips(clips[:64], 7), out[:64, :7])
    for i in (1, 4242):
        ref = O.compute_mfcc(clips[i].cpu().numpy(), 500)
        gate(out[i].cpu().numpy(), ref, f"config4 12500 clips, clip {i}")


def test_experiment_kernels_are_not_in_the_default_library(dsp, torch_cuda):
    """The row-per-frame and two-frames-per-wave forms of the 512-point kernel are measured dead ends (profiles/r02_wave_priority_ab.txt):
    the product library does not carry them and says so; DSP_KERNEL_ROW stays valid on 1024-point plans (the Stockham fallback)."""
    if b"+experiments" in dsp.load().dsp_version():
        pytest.skip("library built with DSP_AMD_EXPERIMENTS=1")
    plan = dsp.MfccPlan()
    for kern in (1, 3):
        with pytest.raises(dsp.DspError, match="DSP_AMD_EXPERIMENTS"):
            plan.set_kernel(kern)
    plan.set_kernel(2)
    dsp.MfccPlan(dsp.default_config(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128)).set_kernel(1)


def test_config3_1024_point_128_mel_with_prefilter(dsp, torch_cuda):
    """BASELINE config 3 at oracle-friendly size: per-frame float64 Butterworth (3000-7500 Hz literals)
    from zero state -> Hann(1024) -> 1024-pt FFT -> 128 HTK mel -> per-frame dB -> 13 coeffs."""
    from oracle import oracle as O
    torch = torch_cuda
    over = dict(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128)
    for pre in (0, 2, 1):
        cfg = dsp.default_config(prefilter=pre, **over)
        ocfg = O.default_cfg(prefilter=pre, **over)
        plan = dsp.MfccPlan(cfg)
        for n in (1, 5, 64, 333):
            fr = S.uniform_pm1(1024 * n, 700 + n).reshape(n, 1024)
            if n >= 5:
                fr[1] = 0.0
                fr[3] = S.chirp(1024, 3500.0, 7000.0)
            out = plan.frames(torch.from_numpy(fr).cuda()).cpu().numpy()
            ref = O.mfcc_frames(fr, ocfg, threads=4)
            gate(out, ref, f"config3 prefilter {pre}/{n} frames")
            if n >= 5:
                assert not out[1].any()                       # silent frame stays exactly zero through the filter


def test_config3_prefilter_on_frames_that_live_in_the_stop_band(dsp, torch_cuda):
    """The fused prefilter runs its last two cascade sections in float32; frames whose energy lies wholly in the stop band are where a
    mixed-precision filter would show first (what is left after 76 dB of attenuation is the filter's own leakage): tones below and
    above the band, DC, a step, an impulse, low-passed and high-passed noise -- each against the oracle's float64 direct form under
    the same gate (the frames of tools/emulate_prefilter_cascade.py, which chose the precisions).  The yardstick is the oracle with
    its float64 transform (fft_mode FFT_FLOAT64): the question is the FILTER's precision, and on these frames the reference-order
    float32 FFT's own noise is of the gate's size (the impulse frame: 1.03e-4 against it, whatever the filter's precision)."""
    from oracle import oracle as O
    from scipy import signal as ss
    torch = torch_cuda
    over = dict(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128)
    t = np.arange(1024) / 16000.0

    def filtered_noise(order, wn, seed, scale=1.0, kind="low"):
        bb, aa = ss.butter(order, wn, kind)
        return (scale * ss.lfilter(bb, aa, np.random.default_rng(seed).standard_normal(4096))[-1024:]).astype(np.float32)
    frames = {"tone 500 Hz": 0.5 * np.sin(2 * np.pi * 500 * t), "tone 1500 Hz": 0.5 * np.sin(2 * np.pi * 1500 * t),
              "tone 7900 Hz": 0.5 * np.sin(2 * np.pi * 7900 * t), "chirp 200-7900 Hz": S.chirp(1024, 200.0, 7900.0),
              "impulse": np.eye(1, 1024, 777)[0], "dc": np.full(1024, 0.7), "step": np.concatenate([np.zeros(500), np.ones(524)]),
              "low-pass 0.1": filtered_noise(6, 0.1, 0), "low-pass 0.05": filtered_noise(8, 0.05, 1),
              "low-pass 0.2 loud": filtered_noise(6, 0.2, 2, 5.0), "high-pass 0.97": filtered_noise(6, 0.97, 4, kind="high")}
    fr = np.stack([np.asarray(v, np.float32) for v in frames.values()])
    for pre in (2, 1):
        plan = dsp.MfccPlan(dsp.default_config(prefilter=pre, **over))
        out = plan.frames(torch.from_numpy(fr).cuda()).cpu().numpy()
        ref = O.mfcc_frames(fr, O.default_cfg(prefilter=pre, fft_mode=O.FFT_FLOAT64, **over), threads=4)
        for i, name in enumerate(frames):
            gate(out[i:i + 1], ref[i:i + 1], f"config3 prefilter {pre} stop-band frame: {name}")


def test_1024_general_fallback_kernel_agrees_with_the_wave_kernel(dsp, torch_cuda):
    """n_fft = 1024 has two kernels: the register-resident wave kernel (default) and the general Stockham kernel (fallback
    for filterbanks with more than three chunks per lane; forced here through set_kernel(1)).  Both against the oracle."""
    from oracle import oracle as O
    torch = torch_cuda
    over = dict(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128)
    fr = S.uniform_pm1(1024 * 200, 4242).reshape(200, 1024)
    fr[3] = 0.0
    x = torch.from_numpy(fr).cuda()
    ref = O.mfcc_frames(fr, O.default_cfg(**over), threads=4)
    outs = []
    for kern in (0, 1):
        plan = dsp.MfccPlan(dsp.default_config(**over))
        plan.set_kernel(kern)
        out = plan.frames(x).cpu().numpy()
        gate(out, ref, f"1024 kernel {kern}")
        assert not out[3].any()
        outs.append(out)
    assert np.abs(outs[0] - outs[1]).max() <= 2e-3        # two different FFT factorisations of the same chain


def test_config3_full_size_properties(dsp, torch_cuda):
    """BASELINE config 3 at its full size: 10 M frames x 1024 fp32 (41 GB in HBM) through the float64 prefilter and the
    1024-point chain.  Size-independent properties + an oracle spot check; skipped when the card lacks the memory."""
    from oracle import oracle as O
    torch = torch_cuda
    n = 10_000_000
    free, _total = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs ~50 GB of free HBM")
    over = dict(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128, prefilter=2)
    plan = dsp.MfccPlan(dsp.default_config(**over))
    gen = torch.Generator(device="cuda").manual_seed(31)
    x = torch.empty((n, 1024), device="cuda")
    for c0 in range(0, n, 1_000_000):                              # generate in 4 GB pieces
        x[c0:c0 + 1_000_000] = torch.rand((1_000_000, 1024), device="cuda", generator=gen) * 2 - 1
    x[::100_003] = 0.0                                             # silent frames
    x[7] = x[9_999_999]                                            # the same frame at both ends of the batch
    out = plan.frames(x)
    assert tuple(out.shape) == (n, 13) and bool(torch.isfinite(out).all())
    assert not out[::100_003].any()
    assert torch.equal(out[7], out[9_999_999])                     # position in the batch does not matter
    assert torch.equal(plan.frames(x[5_000_000:5_000_512]), out[5_000_000:5_000_512])      # nor does the batch size
    idx = torch.randint(0, n, (96,), device="cuda", generator=gen)
    ref = O.mfcc_frames(x[idx].cpu().numpy(), O.default_cfg(**over), threads=8)
    gate(out[idx].cpu().numpy(), ref, "config3 10M frames, 96-frame oracle sample")
    del x, out
    torch.cuda.empty_cache()


def test_1024_point_clip_framing_and_other_shapes(dsp, torch_cuda):
    from oracle import oracle as O
    torch = torch_cuda
    for over in (dict(n_fft=1024, frame_length=800, hop_length=320, n_mels=40),
                 dict(n_fft=1024, frame_length=1024, hop_length=512, n_mels=64, n_mfcc=16)):
        cfg, ocfg = dsp.default_config(**over), O.default_cfg(**over)
        x = np.stack([S.uniform_pm1(16000, 80), S.chirp(16000, 200.0, 7000.0)])
        out = dsp.MfccPlan(cfg).clips(torch.from_numpy(x).cuda(), 500).cpu().numpy()
        for i in range(2):
            ref = O.compute_mfcc(x[i], 500, ocfg)
            assert out[i].shape == ref.shape
            gate(out[i], ref, f"1024 shapes {over}/clip{i}")


def test_pcm16_ingestion_matches_float_path(dsp, torch_cuda, golden):
    """int16 PCM converted in the kernel's load == the reference's WAV-reader conversions
    followed by compute_mfcc (goldens birdq_ch0 / birdq_avg / stop_121417 come from exactly that)."""
    torch = torch_cuda
    g = golden("mfcc_ref.npz")
    plan = dsp.MfccPlan()
    bird = torch.from_numpy(g["birdq_pcm"].copy()).cuda()                  # [24029][2] int16, interleaved
    stop = torch.from_numpy(g["stop_pcm"].copy()).cuda()
    # stereo: channel 0 (donut-classifier/classifier.c:292-297) and average (main_test.c:205-217)
    for mode, key in ((0, "birdq_ch0"), (1, "birdq_avg")):
        out = plan.clips_pcm16(bird[None].contiguous(), 500, stereo_mode=mode).cpu().numpy()[0]
        assert out.shape == (148, 13)
        gate(out, g["mfcc__" + key], f"pcm16/{key}")
        # bit-identical to the float path fed with the reference's conversion
        x = _cases(g)[key]
        assert np.array_equal(out, plan.clips(torch.from_numpy(x[None]).cuda(), 500).cpu().numpy()[0])
    out = plan.clips_pcm16(stop[None].contiguous(), 500).cpu().numpy()[0]   # mono
    gate(out, g["mfcc__stop_121417"], "pcm16/stop_121417")
    assert np.array_equal(out, plan.clips(torch.from_numpy(_cases(g)["stop_121417"][None]).cuda(), 500).cpu().numpy()[0])
    # batches, extreme samples, short clips
    pcm = torch.randint(-32768, 32768, (33, 3000), dtype=torch.int16, device="cuda")
    pcm[0, :8] = torch.tensor([-32768, 32767, 0, -1, 1, -32768, 32767, 0], dtype=torch.int16)
    a = plan.clips_pcm16(pcm, 500)
    b = plan.clips((pcm.float() / 32768.0).contiguous(), 500)
    assert torch.equal(a, b)
    st = torch.randint(-32768, 32768, (5, 3000, 2), dtype=torch.int16, device="cuda")
    avg = 0.5 * (st[..., 0].float() / 32768.0 + st[..., 1].float() / 32768.0)
    assert torch.equal(plan.clips_pcm16(st, 500, stereo_mode=1), plan.clips(avg.contiguous(), 500))
    assert plan.clips_pcm16(pcm[:, :398].contiguous(), 500).shape[1] == 0


def test_config3_fused_prefilter_equals_the_two_pass_path(dsp, torch_cuda, monkeypatch):
    """BASELINE config 3 in one pass: the Butterworth prefilter as a float64 parallel-form scan inside the 1024-point kernel
    (no filtered copy in HBM) against the two-pass path (serial direct-form-II kernel, then MFCC) and the oracle.  The scan
    equals the serial recurrence to ~1e-13 before the rounding to float, so the two paths differ by float rounding of a few
    filtered samples at most."""
    from oracle import oracle as O
    torch = torch_cuda
    over = dict(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128)
    for pre in (2, 1):
        plan = dsp.MfccPlan(dsp.default_config(prefilter=pre, **over))
        n = 777
        fr = S.uniform_pm1(1024 * n, 9000 + pre).reshape(n, 1024)
        fr[1] = 0.0
        fr[2] *= np.float32(1e-4)
        fr[3] = S.chirp(1024, 3500.0, 7000.0)
        fr[4, 1:] = 0.0                                            # an impulse: the filter's impulse response
        x = torch.from_numpy(fr).cuda()
        fused = plan.frames(x).cpu().numpy()
        monkeypatch.setenv("DSP_AMD_PREFILTER_TWO_PASS", "1")
        two = plan.frames(x).cpu().numpy()
        monkeypatch.delenv("DSP_AMD_PREFILTER_TWO_PASS")
        assert not fused[1].any() and not two[1].any()
        gate(fused, two, f"config3 fused vs two-pass, prefilter {pre}")
        assert np.abs(fused - two).max() <= 2e-3
        ref = O.mfcc_frames(fr, O.default_cfg(prefilter=pre, **over), threads=4)
        gate(fused, ref, f"config3 fused vs oracle, prefilter {pre}")
        # unaligned input falls back to the two-pass path and gives the same numbers as that path
        y = torch.empty(1024 * n + 2, device="cuda")[2:].view(n, 1024)
        y.copy_(x)
        assert np.array_equal(plan.frames(y).cpu().numpy(), two)


def test_gather_pipeline_through_rccl_on_one_rank():
    """The pipelined all-gather of BASELINE config 4 through the real backend ("nccl" = RCCL) in a one-rank group: what a
    one-GPU box can check of the N > 1 path beyond the gloo tests -- communicator creation, the collective on the backend's
    stream behind the MFCC kernel of the same batch, Work.wait() as a stream dependency, slot reuse at depth 2.  Runs in a
    child process (one process group per process)."""
    import subprocess
    import sys
    code = r"""
import os, sys, socket
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import dsp_amd
from dsp_amd.dist import GatherPipeline
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
# phase 1: the backend itself -- communicator creation and one trivial collective, before any of this library's code runs
probe = torch.ones(8, device="cuda")
dist.all_reduce(probe)
torch.cuda.synchronize()
assert float(probe.sum()) == 8.0
print("RCCL-UP", flush=True)
# phase 2: the pipeline through it.  Whatever fails from here on is a failure of the test.
plan = dsp_amd.MfccPlan(dsp_amd.default_config())
gen = torch.Generator(device="cuda").manual_seed(5)
batches = [torch.rand((64, 16000), device="cuda", generator=gen) * 2 - 1 for _ in range(5)]
want = [plan.clips(b, 500).clone() for b in batches]
pipe = GatherPipeline(64, (98, 13), torch.float32, torch.device("cuda", 0), always_collective=True)
assert pipe.collective and pipe.gathered[0] is not pipe.local[0]
slots, got = [], []
for k, b in enumerate(batches):
    slots.append(pipe.submit(lambda block, b=b: plan.clips(b, 500, block)))
    if k >= 1:                                   # consume batch k - 1 while batch k's gather is in flight
        got.append(pipe.result(slots[k - 1]).clone())
got.append(pipe.result(slots[-1]).clone())
pipe.drain()
torch.cuda.synchronize()
for g, w in zip(got, want):
    assert torch.equal(g, w)
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    # Only a box whose RCCL cannot even bring up a one-rank communicator and run a trivial all-reduce (phase 1, no code of this
    # library involved) skips; once "RCCL-UP" is printed every failure -- including NCCL errors, which are also what a misuse of
    # the communicator or a stream-ordering bug in GatherPipeline would raise -- fails the test with the child's stderr.
    if "RCCL-UP" not in r.stdout:
        pytest.skip("RCCL could not bring up a one-rank communicator on this box (before any dsp_amd code ran): "
                    + (r.stderr.strip().splitlines() or ["no stderr"])[-1][:300])
    assert "RCCL-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_fft_real_forward_entry_point_vs_the_reference():
    """fft_real_forward (2fa/audio/word/c/mfcc.c:16-95, non-static there): 400 samples -> 512 complex bins, against the compiled
    reference where oracle/_ref is present and the oracle's reference-order restatement (bit-exact with it on CPU), within the MFCC
    gate's 1e-4 of the frame's L-inf norm; the batch form for other sizes against numpy's float64 transform."""
    import ctypes as C
    from dsp_amd import lib as L
    from oracle import oracle as O
    lib = L.load()
    rng = np.random.default_rng(8)
    frames = [S.uniform_pm1(400, 3), (S.uniform_pm1(400, 4) * np.float32(1e-3)), np.zeros(400, np.float32),
              np.sin(2 * np.pi * 1000 * np.arange(400) / 16000).astype(np.float32), rng.standard_normal(400).astype(np.float32) * 5]
    for x in frames:
        x = np.ascontiguousarray(x, np.float32)
        out = np.full(1024, 7.0, np.float32)
        lib.fft_real_forward(x.ctypes.data, out.ctypes.data)
        ref = O.fft_real_forward(x)
        scale = max(np.abs(ref).max(), 1e-30)
        assert np.abs(out - ref.reshape(-1)).max() <= 1e-4 * scale or scale <= 1e-30
        try:
            rl = O.ref_mfcc_lib()
        except OSError:
            rl = None                                            # (oracle/_ref is built where /root/reference exists and travels with the tree)
        if rl is not None:
            cref = np.empty(1024, np.float32)
            rl.fft_real_forward(x, cref)
            assert np.abs(out - cref).max() <= 1e-4 * max(np.abs(cref).max(), 1e-30)
    for n_fft, flen, n in ((512, 512, 7), (1024, 800, 3), (64, 64, 5), (2048, 2048, 2)):
        x = rng.standard_normal((n, flen)).astype(np.float32)
        out = np.empty((n, n_fft, 2), np.float32)
        L.check(lib.dsp_fft_real_forward_host(x.ctypes.data, n, flen, flen, n_fft, out.ctypes.data), "fft batch")
        want = np.fft.fft(np.pad(x.astype(np.float64), ((0, 0), (0, n_fft - flen))), axis=1)
        got = out[..., 0] + 1j * out[..., 1]
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    assert lib.dsp_fft_real_forward_host(x.ctypes.data, 1, 100, 100, 48, out.ctypes.data) < 0
