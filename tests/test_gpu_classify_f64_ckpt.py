"""GPU: the float64 classifier's default pipeline (round 4) -- one pass that keeps the filters' restart states instead of the filtered
signals and settles the loud time bins with a bounded bf16 screening transform on the matrix pipe, float64 recompute + transform for
undecided segments and for the clips with midpoints (dsp_amd/csrc/classify_f64_ckpt_kernels.hip).  tests/test_gpu_classify_f64.py holds
the oracle comparisons (untouched); here the new pipeline is held against round 3's (DSP_AMD_F64_PIPELINE=materialize: both filtered
signals through HBM, the same float64 transform) BIT FOR BIT, the screening's verdicts against the float64 transform's on every
segment, and the int16 entry points against the float64 ones on the same samples."""
import os

import numpy as np
import pytest

from tests import signals as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dsp():
    import dsp_amd
    return dsp_amd


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _same(a, b):
    (la, ta), (lb, tb) = a, b
    assert np.array_equal(la, lb)
    for (m, s), (m2, s2) in zip(ta, tb):
        assert np.array_equal(m, m2) and np.array_equal(s, s2)


def _mixed_clips(n_noise=48, seed=23):
    cases = S.classify_cases()
    rng = np.random.default_rng(seed)
    call = cases["scrub_a"].astype(np.float64)
    return np.concatenate([np.stack([c.astype(np.float64) for c in cases.values()]),
                           rng.uniform(-1, 1, (n_noise, 16000)) * np.logspace(-3.5, -0.5, n_noise)[:, None],      # floors below, around and above 45 dB
                           rng.uniform(-0.01, 0.01, (16, 16000)) + call,
                           call[None, ::-1] * np.linspace(0.05, 1.0, 8)[:, None]])


def test_checkpoint_pipeline_equals_the_materialised_one_bit_for_bit(dsp):
    """A segment recomputed from its restart state sees the reference's operations on the same values in the same order as the
    whole-clip recurrence, and both pipelines run the same float64 transform: labels, midpoints and band sums identical."""
    clips = _mixed_clips()
    got = dsp.classify_batch_f64(clips, with_trace=True)
    with _env(DSP_AMD_F64_PIPELINE="materialize"):
        ref = dsp.classify_batch_f64(clips, with_trace=True)
    _same(got, ref)
    assert got[0].any() and not got[0].all() and sum(len(m) for m, _ in got[1]) >= 25
    # odd lengths, one segment, lengths whose last int16 / float64 load tile is partial
    rng = np.random.default_rng(5)
    for n in (256, 479, 480, 703, 5001, 15999, 16001):
        x = rng.uniform(-0.3, 0.3, (5, n))
        a = dsp.classify_batch_f64(x, with_trace=True)
        with _env(DSP_AMD_F64_PIPELINE="materialize"):
            b = dsp.classify_batch_f64(x, with_trace=True)
        _same(a, b)


def test_screening_never_contradicts_the_float64_transform(dsp):
    """DSP_AMD_F64_GUARD=0.9 leaves (nearly) every segment to the float64 transform: the default run, where the screening settles most
    of them, must give the same midpoints and sums -- on clips whose noise floors sweep through the threshold, where a wrong "quiet
    for sure" / "loud for sure" would move a midpoint."""
    clips = _mixed_clips(n_noise=96, seed=31)
    got = dsp.classify_batch_f64(clips, with_trace=True)
    seg, undecided, listed = dsp.classify_stats_f64(0)
    assert seg == len(clips) * 71 and 0 < listed <= len(clips)
    with _env(DSP_AMD_F64_GUARD="0.9"):
        ref = dsp.classify_batch_f64(clips, with_trace=True)
        seg2, undecided2, _ = dsp.classify_stats_f64(0)
    _same(got, ref)
    # the screening does its job: few segments undecided by default, most of them with the guard band at +-90 %
    assert undecided < 0.05 * seg, (undecided, seg)
    assert undecided2 > 0.15 * seg2 and undecided2 > 4 * undecided, (undecided, undecided2, seg2)
    print(f"screening: {undecided} of {seg} segments undecided by default, {undecided2} with the guard at +-90 %")
    counts = [len(m) for m, _ in got[1]]
    assert 0 in counts and max(counts) >= 1


def test_screening_on_tones_dc_and_steps(dsp):
    """Inputs the screening's bounds are tight or loose on: tones in and outside the 1000-3000 Hz band (energy in bins the screening
    computes / only bounds), a DC offset (the mean term), steps and impulses, silence, full scale: same results as the float64
    transform on every segment."""
    n = 16000
    t = np.arange(n) / 16000.0
    rng = np.random.default_rng(9)
    rows = []
    for f in (300.0, 1000.0, 2000.0, 2999.0, 3900.0, 4100.0, 5000.0, 7900.0):
        for a in (1e-4, 3e-3, 0.05, 0.9):
            rows.append(a * np.sin(2 * np.pi * f * t + rng.uniform(0, 6)))
    rows.append(np.full(n, 0.7))
    rows.append(np.zeros(n))
    rows.append(np.where(t > 0.5, 0.9, -0.9))
    imp = np.zeros(n); imp[::997] = 1.0
    rows.append(imp)
    rows.append(rng.uniform(-1, 1, n))
    rows.append(0.5 + 0.02 * np.sin(2 * np.pi * 2000 * t))
    burst = np.zeros(n); burst[4000:9000] = 0.02 * np.sin(2 * np.pi * 2200 * t[4000:9000])
    rows.append(burst)
    clips = np.stack(rows)
    got = dsp.classify_batch_f64(clips, with_trace=True)
    with _env(DSP_AMD_F64_GUARD="0.9"):
        ref = dsp.classify_batch_f64(clips, with_trace=True)
    _same(got, ref)
    with _env(DSP_AMD_F64_PIPELINE="materialize"):
        _same(got, dsp.classify_batch_f64(clips, with_trace=True))
    assert sum(len(m) for m, _ in got[1]) >= 5


def test_int16_input_is_bit_identical_to_float64_input(dsp, golden):
    """int16 / 32768.0 is exact in double (classifier.c:55-59): the pcm16 entry points must return the float64 entry points' bits on the
    same samples -- mono, stereo channel 0 (classifier.c:286-297), stereo average -- through host and device entries, aligned and not."""
    import torch
    rng = np.random.default_rng(41)
    call = S.classify_cases()["scrub_a"].astype(np.float64)
    n = 16000
    base = np.concatenate([rng.uniform(-0.002, 0.002, (10, n)) + call, rng.uniform(-1, 1, (10, n)) * np.logspace(-3, -0.5, 10)[:, None]])
    pcm = np.clip(np.round(base * 32768.0), -32768, 32767).astype(np.int16)
    as_f64 = pcm.astype(np.float64) / 32768.0
    ref = dsp.classify_batch_f64(as_f64, with_trace=True)
    _same(dsp.classify_batch_f64_pcm16(pcm, with_trace=True), ref)
    assert ref[0].any() and not ref[0].all()
    # stereo: channel 0 = the mono samples beside an unrelated channel; average of (L, R)
    other = rng.integers(-20000, 20000, pcm.shape).astype(np.int16)
    st = np.stack([pcm, other], axis=2)
    _same(dsp.classify_batch_f64_pcm16(st, dsp.STEREO_CHANNEL0, with_trace=True), ref)
    avg = (pcm.astype(np.float64) + other.astype(np.float64)) / 65536.0
    _same(dsp.classify_batch_f64_pcm16(st, dsp.STEREO_AVERAGE, with_trace=True), dsp.classify_batch_f64(avg, with_trace=True))
    # device entries (stream-ordered) and rows that are not 16-byte aligned (odd stride: the element-wise loader)
    d = torch.from_numpy(pcm).cuda()
    assert np.array_equal(dsp.classify_device_f64_pcm16(d).cpu().numpy(), ref[0])
    padded = torch.zeros((20, n + 3), dtype=torch.int16, device="cuda")
    padded[:, :n] = d
    assert np.array_equal(dsp.classify_device_f64_pcm16(padded[:, :n]).cpu().numpy(), ref[0])
    ds = torch.from_numpy(st).cuda()
    assert np.array_equal(dsp.classify_device_f64_pcm16(ds, stereo_mode=dsp.STEREO_CHANNEL0).cpu().numpy(), ref[0])
    # odd lengths through the int16 loader (load tiles of 64 samples, the last one partial)
    for m in (256, 479, 703, 5001, 15999):
        _same(dsp.classify_batch_f64_pcm16(pcm[:6, :m], with_trace=True), dsp.classify_batch_f64(as_f64[:6, :m], with_trace=True))
    # the classifier's own recordings as the reference reads them: int16 channel 0
    g = golden("donut16k_ref.npz")
    for name in sorted({k.split("__")[0] for k in g.files}):
        p = g[name + "__pcm"]
        _same(dsp.classify_batch_f64_pcm16(p[None, :, :] if p.shape[1] == 2 else p[None, :, 0], dsp.STEREO_CHANNEL0, with_trace=True),
              dsp.classify_batch_f64((p[:, 0].astype(np.float64) / 32768.0)[None, :], with_trace=True))


def test_float64_device_entry_with_trace_odd_stride_and_streams(dsp):
    """The device entry is stream-ordered: two calls on two streams share the device's workspace and are ordered by an event; a device
    trace buffer and an odd row stride (element-wise loader) agree with the host entry."""
    import ctypes as C
    import torch
    from dsp_amd import lib as L
    clips = _mixed_clips(n_noise=24, seed=3)
    ref = dsp.classify_batch_f64(clips, with_trace=True)
    n_clips, n = clips.shape
    padded = torch.zeros((n_clips, n + 1), dtype=torch.float64, device="cuda")       # odd stride
    padded[:, :n] = torch.from_numpy(clips).cuda()
    lab = torch.empty(n_clips, dtype=torch.int32, device="cuda")
    tr = torch.zeros(n_clips * C.sizeof(L.ClassifyTraceF64), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.check(L.load().dsp_classify_batch_device_f64(None, padded.data_ptr(), n_clips, n, n + 1, lab.data_ptr(), tr.data_ptr(), C.c_void_p(st)), "device f64")
    torch.cuda.synchronize()
    assert np.array_equal(lab.cpu().numpy(), ref[0])
    host = (L.ClassifyTraceF64 * n_clips).from_buffer_copy(tr.cpu().numpy().tobytes())
    for t, (m, s) in zip(host, ref[1]):
        assert t.n_midpoints == len(m) and np.array_equal(np.array(t.midpoints[:len(m)]), m)
        assert np.array_equal(np.array([[t.sums[i][j] for j in range(3)] for i in range(len(m))]).reshape(-1, 3), s)
    # two streams, interleaved calls, no synchronisation in between
    d = torch.from_numpy(clips).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for k in range(6):
        with torch.cuda.stream(s1 if k % 2 == 0 else s2):
            outs.append(dsp.classify_device_f64(d if k % 3 else d.flip(0)))
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), ref[0] if k % 3 else ref[0][::-1]), k
    dsp.classify_release_f64(0)
    assert np.array_equal(dsp.classify_device_f64(d).cpu().numpy(), ref[0])         # the workspace comes back after a release
