"""GPU: ragged batches -- clips of different lengths in ONE launch of the fused clip kernels (the reference's callers loop over files:
cepstrum/scrubjay_infer.c:158-176, 2fa/audio/word/c/main_test.c:254-331).  dsp_scrubjay_fused_ragged_*_device,
dsp_classify_signal_batch_ragged_*_device.  The bar: clip c of the batch gets what a one-clip call on it returns, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dsp():
    import dsp_amd
    return dsp_amd


def _lengths(rng, count, lo, hi):
    n = rng.integers(lo, hi, count)
    n[0], n[1] = lo, hi          # the shortest clip that still has a frame, and the longest
    n[2] |= 1                    # odd lengths: every later clip starts at an odd sample
    n[3] |= 1
    return n


def _pack(clips):
    offsets = np.zeros(len(clips) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([c.shape[0] for c in clips])
    return np.concatenate(clips, axis=0), offsets


def test_scrubjay_ragged_equals_one_call_per_clip(golden):
    import torch
    import dsp_amd
    from dsp_amd.scrubjay import ScrubJay, scrubjay_infer_config
    attrs = dict(golden("scrubjay_svm.npz"))
    rng = np.random.default_rng(77)
    for name, sj, lo in (("reference framing, 20 coefficients", ScrubJay(attrs), 400),
                         ("scrubjay_infer.c's front end", ScrubJay(attrs, config=scrubjay_infer_config(16000)), 1)):
        lens = _lengths(rng, 37, lo, 52000)
        # float samples
        clips = [(rng.standard_normal(n) * rng.choice([0.001, 0.1, 0.5])).astype(np.float32) for n in lens]
        flat, off = _pack(clips)
        got = sj.ragged(torch.from_numpy(flat).cuda(), off, 500)
        for c, x in enumerate(clips):
            want = sj(torch.from_numpy(x[None, :]).cuda(), 500, fused=True)
            for a, b in zip(got, want):
                assert torch.equal(a[c:c + 1], b), (name, "float", c, lens[c])
        # a max_frames cap that bites on the long clips only
        got = sj.ragged(torch.from_numpy(flat).cuda(), off, 40)
        for c in (0, 1, 5):
            want = sj(torch.from_numpy(clips[c][None, :]).cuda(), 40, fused=True)
            for a, b in zip(got, want):
                assert torch.equal(a[c:c + 1], b), (name, "capped", c)
        # int16 mono and interleaved stereo
        pcm = [rng.integers(-20000, 20000, n).astype(np.int16) for n in lens]
        flat, off = _pack(pcm)
        got = sj.ragged(torch.from_numpy(flat).cuda(), off, 500)
        for c, x in enumerate(pcm):
            want = sj.pcm16(torch.from_numpy(x[None, :]).cuda(), 500)
            for a, b in zip(got, want):
                assert torch.equal(a[c:c + 1], b), (name, "int16", c, lens[c])
        st = [rng.integers(-20000, 20000, (n, 2)).astype(np.int16) for n in lens]
        flat, off = _pack(st)
        for mode in (dsp_amd.STEREO_CHANNEL0, dsp_amd.STEREO_AVERAGE):
            got = sj.ragged(torch.from_numpy(flat).cuda(), off, 500, stereo_mode=mode)
            for c, x in enumerate(st):
                want = sj.pcm16(torch.from_numpy(x[None]).cuda(), 500, stereo_mode=mode)
                for a, b in zip(got, want):
                    assert torch.equal(a[c:c + 1], b), (name, "stereo", mode, c, lens[c])
        # clips anywhere in the buffer (gaps, order): offsets only need to be non-decreasing
        flat, off = _pack(clips)
        gap = np.concatenate([flat[:off[3]], np.full(5, 9.0, np.float32), flat[off[3]:]])
        sub = np.array([off[1], off[2], off[3]])                 # clips 1 and 2
        got = sj.ragged(torch.from_numpy(gap).cuda(), sub, 500)
        for i, c in enumerate((1, 2)):
            want = sj(torch.from_numpy(clips[c][None, :]).cuda(), 500, fused=True)
            assert torch.equal(got[3][i:i + 1], want[3])
        # an empty batch, a clip without a frame, offsets that run backwards
        assert sj.ragged(torch.from_numpy(flat).cuda(), np.array([0]), 500)[0].shape[0] == 0
        with pytest.raises(dsp_amd.DspError, match="shorter than one frame"):
            sj.ragged(torch.from_numpy(flat).cuda(), np.array([0, 16000, 16000 + lo - 1]), 500)
        with pytest.raises(dsp_amd.DspError, match="non-decreasing"):
            sj.ragged(torch.from_numpy(flat).cuda(), np.array([0, 16000, 8000]), 500)


def test_scrubjay_ragged_on_the_labelled_recordings(golden):
    """The reference's two labelled recordings of tests/golden/labelled_audio.npz (stereo int16, 96 kHz, different lengths), channel
    average, through scrubjay_infer.c's own front end: ONE call, against one call per file."""
    import torch
    import dsp_amd
    from dsp_amd.scrubjay import ScrubJay, scrubjay_infer_config
    g = golden("labelled_audio.npz")
    attrs = dict(golden("scrubjay_svm.npz"))
    files = [np.ascontiguousarray(g[f"{name}__pcm"]) for name in ("sj_short", "not_sj")]
    sj = ScrubJay(attrs, config=scrubjay_infer_config(int(g["sj_short__sr"])))
    flat, off = _pack(files)
    got = sj.ragged(torch.from_numpy(flat).cuda(), off, 1 << 20, stereo_mode=dsp_amd.STEREO_AVERAGE)
    for c, x in enumerate(files):
        want = sj.pcm16(torch.from_numpy(x[None]).cuda(), 1 << 20, stereo_mode=dsp_amd.STEREO_AVERAGE)
        for a, b in zip(got, want):
            assert torch.equal(a[c:c + 1], b), c


def test_classify_signal_ragged_equals_one_call_per_clip_and_the_reference_clips(golden):
    import torch
    import dsp_amd
    net = dsp_amd.StopModel(dict(golden("stop_model.npz")))
    plan = dsp_amd.MfccPlan(dsp_amd.default_config())
    g = golden("stop_ref.npz")
    # the reference's seven test clips in ONE call, as its harness reads them (int16), against the compiled reference's probabilities
    pcm = [np.asarray(g[f"clip{i}__pcm"]).astype(np.int16) for i in range(7)]
    flat, off = _pack(pcm)
    p = net.classify_signal_ragged(plan, torch.from_numpy(flat).cuda(), off).cpu().numpy()
    for i in range(7):
        assert abs(p[i] - float(g[f"clip{i}__prob"])) <= 5e-5, i
        one = net.classify_signal_batch_pcm16(plan, torch.from_numpy(pcm[i][None, :]).cuda()).cpu().numpy()[0]
        assert p[i] == one, i
    # different lengths: below and above the model's 500 frames (80 400 samples), odd starts
    rng = np.random.default_rng(78)
    lens = _lengths(rng, 41, 400, 120000)
    clips = [(rng.standard_normal(n) * rng.choice([0.001, 0.1, 0.5])).astype(np.float32) for n in lens]
    flat, off = _pack(clips)
    got = net.classify_signal_ragged(plan, torch.from_numpy(flat).cuda(), off)
    for c, x in enumerate(clips):
        want = net.classify_signal_batch(plan, torch.from_numpy(x[None, :]).cuda())
        assert torch.equal(got[c:c + 1], want), (c, lens[c])
    st = [rng.integers(-20000, 20000, (n, 2)).astype(np.int16) for n in lens[:12]]
    flat, off = _pack(st)
    got = net.classify_signal_ragged(plan, torch.from_numpy(flat).cuda(), off, stereo_mode=dsp_amd.STEREO_AVERAGE)
    for c, x in enumerate(st):
        want = net.classify_signal_batch_pcm16(plan, torch.from_numpy(x[None]).cuda(), stereo_mode=dsp_amd.STEREO_AVERAGE)
        assert torch.equal(got[c:c + 1], want), c
    with pytest.raises(dsp_amd.DspError):
        net.classify_signal_ragged(dsp_amd.MfccPlan(dsp_amd.default_config(n_mfcc=20)), torch.from_numpy(flat).cuda(), off)


def test_many_ragged_calls_in_flight_share_no_buffer(golden):
    """The spans ride a ring of four pinned / device buffer pairs: ten calls queued on one stream, each with its own offsets, return what
    each returns alone."""
    import torch
    from dsp_amd.scrubjay import ScrubJay
    sj = ScrubJay(dict(golden("scrubjay_svm.npz")))
    rng = np.random.default_rng(79)
    flat = torch.from_numpy((rng.standard_normal(400000) * 0.1).astype(np.float32)).cuda()
    offs = [np.sort(rng.integers(0, 400000 // 800, 60)) * 800 for _ in range(10)]
    offs = [np.unique(o) for o in offs]
    alone = []
    for o in offs:
        alone.append([t.clone() for t in sj.ragged(flat, o, 500)])
        torch.cuda.synchronize()
    queued = [sj.ragged(flat, o, 500) for o in offs]
    torch.cuda.synchronize()
    for a, q in zip(alone, queued):
        for x, y in zip(a, q):
            assert torch.equal(x, y)


def _trace_equal(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _classify_clips(rng, count):
    """clips of 0.01 s .. 5 s cut from the classifier's test signals (calls, bursts, noise) with offsets, gains and tails of noise"""
    from tests import signals as S
    cases = S.classify_cases()
    names = ("scrub_a", "scrub_b", "jay_like", "burst_2k", "noise", "silence")
    clips = []
    for i in range(count):
        base = cases[names[i % len(names)]]
        n = int(rng.integers(160, 80000))
        reps = -(-n // base.size)
        x = np.tile(base, reps)[:n] * np.float32(rng.choice([1.0, 0.5, 2.0]))
        x = x + (rng.standard_normal(n) * 1e-3).astype(np.float32)
        clips.append(x.astype(np.float32))
    clips[0] = clips[0][:255]            # no segment at all
    clips[1] = np.tile(cases["scrub_a"], 2)[:256 + 224 * 3 + 7]      # a tail past the last whole segment
    clips[2] = cases["scrub_a"][:15999]  # odd length: later clips start at odd samples
    return clips


def test_classify_ragged_equals_one_call_per_clip(dsp, golden):
    """dsp_classify_batch_ragged_host / _device: labels, midpoints and band sums of every clip as a one-clip call returns them, under the
    sync/lib thresholds and the microphone firmware's (where the test signals have midpoints and band sums)."""
    import torch
    rng = np.random.default_rng(90)
    clips = _classify_clips(rng, 150)
    flat, off = _pack(clips)
    for cfg in (None, dsp.CLASSIFY_MICROPHONE):
        labels, trace = dsp.classify_ragged(flat, off, with_trace=True, config=cfg)
        dev = dsp.classify_device_ragged(torch.from_numpy(flat).cuda(), off, config=cfg).cpu().numpy()
        assert np.array_equal(dev, labels)
        seen = 0
        for c, x in enumerate(clips):
            l1, t1 = dsp.classify_batch(x[None, :], with_trace=True, config=cfg)
            assert labels[c] == l1[0] and _trace_equal(trace[c], t1[0]), (c, x.size, cfg)
            seen += len(t1[0][0])
        assert seen > 40 and labels.sum() > 5
    # int16: mono, stereo channel 0, stereo average
    pcm = [np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16) for x in clips[:60]]
    flat, off = _pack(pcm)
    labels, trace = dsp.classify_ragged(flat, off, with_trace=True, config=dsp.CLASSIFY_MICROPHONE)
    assert np.array_equal(dsp.classify_device_ragged(torch.from_numpy(flat).cuda(), off, config=dsp.CLASSIFY_MICROPHONE).cpu().numpy(), labels)
    for c, x in enumerate(pcm):
        l1, t1 = dsp.classify_batch_pcm16(x[None, :], with_trace=True, config=dsp.CLASSIFY_MICROPHONE)
        assert labels[c] == l1[0] and _trace_equal(trace[c], t1[0]), (c, x.size)
    st = [np.stack([x, (x // 3).astype(np.int16)], axis=1) for x in pcm]
    flat, off = _pack(st)
    for mode in (dsp.STEREO_CHANNEL0, dsp.STEREO_AVERAGE):
        labels, trace = dsp.classify_ragged(flat, off, stereo_mode=mode, with_trace=True, config=dsp.CLASSIFY_MICROPHONE)
        for c, x in enumerate(st):
            l1, t1 = dsp.classify_batch_pcm16(x[None], stereo_mode=mode, with_trace=True, config=dsp.CLASSIFY_MICROPHONE)
            assert labels[c] == l1[0] and _trace_equal(trace[c], t1[0]), (mode, c)
    # nothing but clips without a segment; an empty batch; a clip of more than 13.4 s; offsets that run backwards
    assert not dsp.classify_ragged(flat[:, 0].copy(), np.array([0, 100, 355])).any()
    assert dsp.classify_ragged(flat[:, 0].copy(), np.array([0])).size == 0
    with pytest.raises(dsp.DspError, match="too long"):
        dsp.classify_ragged(np.zeros(300000, np.float32), np.array([0, 1000, 300000]))
    with pytest.raises(dsp.DspError, match="non-decreasing"):
        dsp.classify_ragged(np.zeros(30000, np.float32), np.array([0, 16000, 1000]))


def test_classify_ragged_on_the_donut_recordings(dsp, golden):
    """The donut classifier's own 16 kHz recordings (0.15 s .. 3 s, stereo int16; donut-classifier/16k/*.wav) in ONE call, as its reader
    takes them (channel 0): labels and midpoints as the compiled reference returned them, band sums under the microphone firmware's
    thresholds as the oracle gives them."""
    g = golden("donut16k_ref.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    flat, off = _pack([np.ascontiguousarray(g[n + "__pcm"]) for n in names])
    labels, trace = dsp.classify_ragged(flat, off, stereo_mode=dsp.STEREO_CHANNEL0, with_trace=True)
    mlabels, mtrace = dsp.classify_ragged(flat, off, stereo_mode=dsp.STEREO_CHANNEL0, with_trace=True, config=dsp.CLASSIFY_MICROPHONE)
    for c, n in enumerate(names):
        assert labels[c] == int(g[n + "__label"]) and np.array_equal(trace[c][0], g[n + "__midpoints"]), n
        assert mlabels[c] == int(g[n + "__mic_label"]), n
        assert np.array_equal(mtrace[c][0], g[n + "__mic_midpoints"]) and np.array_equal(mtrace[c][1], g[n + "__mic_sums"]), n


def test_classify_ragged_across_sub_batches_and_block_edges(dsp):
    """More clips than one pass through the workspace holds (65 536), lengths that differ inside every 64-clip block: labels against the
    uniform entry point on the same clips grouped by length."""
    import torch
    from tests import signals as S
    cases = S.classify_cases()
    rng = np.random.default_rng(91)
    lens = (4000, 9000, 16000)
    n_clips = 70000
    which = rng.integers(0, 3, n_clips)
    kind = rng.integers(0, 4, n_clips)
    src = [cases["scrub_a"], cases["noise"], cases["jay_like"], cases["scrub_b"]]
    off = np.zeros(n_clips + 1, dtype=np.int64)
    off[1:] = np.cumsum(np.asarray(lens)[which])
    flat = torch.empty(int(off[-1]), device="cuda")
    srcs = [[torch.from_numpy(s[:n].copy()).cuda() for n in lens] for s in src]
    want = np.zeros(n_clips, np.int32)
    for k in range(4):
        for w, n in enumerate(lens):
            one = int(dsp.classify_device(srcs[k][w][None, :]).cpu()[0])
            idx = np.nonzero((kind == k) & (which == w))[0]
            want[idx] = one
            pos = torch.from_numpy(off[idx]).cuda()[:, None] + torch.arange(n, device="cuda")[None, :]
            flat[pos.reshape(-1)] = srcs[k][w].repeat(idx.size)
    got = dsp.classify_device_ragged(flat, off).cpu().numpy()
    assert np.array_equal(got, want) and 0 < want.sum() < n_clips


def _trace_equal_f64(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_classify_f64_ragged_equals_one_call_per_clip(dsp, golden):
    """dsp_classify_batch_ragged_*_f64 (the float64 classify() of donut-classifier/classifier.c): labels, midpoints and band sums of
    every clip as a one-clip call returns them, bit for bit -- default thresholds and a 30 dB midpoint threshold (where the faint tails
    have midpoints too); float64 samples, int16 mono, stereo channel 0 / average; device and host entry points."""
    import torch
    rng = np.random.default_rng(92)
    clips = [x.astype(np.float64) for x in _classify_clips(rng, 90)]
    flat, off = _pack(clips)
    for cfg in (None, (0.70, 0.85, 30.0, 75.0, 300.0, 100.0)):
        labels, trace = dsp.classify_ragged_f64(flat, off, with_trace=True, config=cfg)
        dev = dsp.classify_device_ragged_f64(torch.from_numpy(flat).cuda(), off, config=cfg).cpu().numpy()
        assert np.array_equal(dev, labels)
        seen = 0
        for c, x in enumerate(clips):
            l1, t1 = dsp.classify_batch_f64(x[None, :], with_trace=True, config=cfg)
            assert labels[c] == l1[0] and _trace_equal_f64(trace[c], t1[0]), (c, x.size, cfg)
            seen += len(t1[0][0])
        assert seen > 20
    pcm = [np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16) for x in clips[:40]]
    flat, off = _pack(pcm)
    labels, trace = dsp.classify_ragged_f64(flat, off, with_trace=True)
    assert np.array_equal(dsp.classify_device_ragged_f64(torch.from_numpy(flat).cuda(), off).cpu().numpy(), labels)
    for c, x in enumerate(pcm):
        l1, t1 = dsp.classify_batch_f64_pcm16(x[None, :], with_trace=True)
        assert labels[c] == l1[0] and _trace_equal_f64(trace[c], t1[0]), (c, x.size)
    st = [np.stack([x, (x // 3).astype(np.int16)], axis=1) for x in pcm]
    flat, off = _pack(st)
    for mode in (dsp.STEREO_CHANNEL0, dsp.STEREO_AVERAGE):
        labels, trace = dsp.classify_ragged_f64(flat, off, stereo_mode=mode, with_trace=True)
        for c, x in enumerate(st):
            l1, t1 = dsp.classify_batch_f64_pcm16(x[None], stereo_mode=mode, with_trace=True)
            assert labels[c] == l1[0] and _trace_equal_f64(trace[c], t1[0]), (mode, c)
    assert not dsp.classify_ragged_f64(flat[:, 0].copy(), np.array([0, 100, 355])).any()
    with pytest.raises(dsp.DspError, match="too long"):
        dsp.classify_ragged_f64(np.zeros(300000), np.array([0, 1000, 300000]))


def test_classify_f64_ragged_on_the_donut_recordings(dsp, golden):
    """The classifier's own 16 kHz recordings in ONE call, int16 channel 0 as classifier.c:286-297 reads them: against one call per file."""
    g = golden("donut16k_ref.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    files = [np.ascontiguousarray(g[n + "__pcm"]) for n in names]
    flat, off = _pack(files)
    for cfg in (None, (0.70, 0.85, 30.0, 75.0, 300.0, 100.0)):
        labels, trace = dsp.classify_ragged_f64(flat, off, stereo_mode=dsp.STEREO_CHANNEL0, with_trace=True, config=cfg)
        for c, x in enumerate(files):
            l1, t1 = dsp.classify_batch_f64_pcm16(x[None], stereo_mode=dsp.STEREO_CHANNEL0, with_trace=True, config=cfg)
            assert labels[c] == l1[0] and _trace_equal_f64(trace[c], t1[0]), (names[c], cfg)


def test_classify_f64_ragged_across_passes(dsp, monkeypatch):
    """A ragged float64 batch that spans several passes through the workspace (DSP_AMD_F64_SUB_BATCH=128): the clips run in order of
    length, pass by pass, and every label and trace record still lands at the caller's index."""
    import torch
    rng = np.random.default_rng(93)
    clips = [x.astype(np.float64) for x in _classify_clips(rng, 300)]
    flat, off = _pack(clips)
    whole, trace = dsp.classify_ragged_f64(flat, off, with_trace=True)
    monkeypatch.setenv("DSP_AMD_F64_SUB_BATCH", "128")
    labels, tr2 = dsp.classify_ragged_f64(flat, off, with_trace=True)
    dev = dsp.classify_device_ragged_f64(torch.from_numpy(flat).cuda(), off).cpu().numpy()
    assert np.array_equal(labels, whole) and np.array_equal(dev, whole) and whole.sum() > 10
    for a, b in zip(trace, tr2):
        assert _trace_equal_f64(a, b)
