"""GPU: ragged batches -- clips of different lengths in ONE launch of the fused clip kernels (the reference's callers loop over files:
cepstrum/scrubjay_infer.c:158-176, 2fa/audio/word/c/main_test.c:254-331).  dsp_scrubjay_fused_ragged_*_device,
dsp_classify_signal_batch_ragged_*_device.  The bar: clip c of the batch gets what a one-clip call on it returns, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
