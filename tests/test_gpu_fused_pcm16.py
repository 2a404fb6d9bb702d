"""GPU: int16 PCM in front of the fused clip kernels (main_test.c:198-217 decodes int16 in front of classify_signal,
scrubjay_infer.c's callers int16 WAV files): dsp_scrubjay_fused_pcm16_device, dsp_classify_signal_batch_pcm16_device.  int16 / 32768
(folded into the window table, exact) gives the float path's inputs bit for bit, so the bar is the float entry points' results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pcm(n_clips, n, seed, stereo=False):
    rng = np.random.default_rng(seed)
    shape = (n_clips, n, 2) if stereo else (n_clips, n)
    pcm = rng.integers(-32768, 32768, shape).astype(np.int16)
    pcm[::4] = (pcm[::4] * 0.01).astype(np.int16)
    if n_clips > 2:
        pcm[2] = 0
    return pcm


def test_scrubjay_fused_on_int16_equals_the_float_path(golden):
    import torch
    import dsp_amd
    from dsp_amd.scrubjay import ScrubJay
    attrs = dict(golden("scrubjay_svm.npz"))
    sj = ScrubJay(attrs)
    for n_clips, n in ((200, 16000), (5, 400), (7, 560), (3, 40000)):
        pcm = _pcm(n_clips, n, 100 + n)
        ref = sj(torch.from_numpy(pcm.astype(np.float32) / np.float32(32768.0)).cuda(), 500, fused=True)
        got = sj.pcm16(torch.from_numpy(pcm).cuda(), 500)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), (n_clips, n)
        st = _pcm(n_clips, n, 200 + n, stereo=True)
        ch0 = sj(torch.from_numpy(st[:, :, 0].astype(np.float32) / np.float32(32768.0)).cuda(), 500, fused=True)
        got = sj.pcm16(torch.from_numpy(st).cuda(), 500, stereo_mode=dsp_amd.STEREO_CHANNEL0)
        for a, b in zip(got, ch0):
            assert torch.equal(a, b)
        # the average as main_test.c:205-217 forms it: 0.5 (L / 32768 + R / 32768); the kernel folds 1 / 65536 into its window
        avg = sj.pcm16(torch.from_numpy(st).cuda(), 500, stereo_mode=dsp_amd.STEREO_AVERAGE)
        want = sj(torch.from_numpy((st[:, :, 0].astype(np.float32) + st[:, :, 1].astype(np.float32)) / np.float32(65536.0)).cuda(), 500, fused=True)
        assert torch.equal(avg[0], want[0]) and torch.allclose(avg[3], want[3], rtol=0, atol=2e-4)
    assert got[0].shape[0] == 3
    # the front end scrubjay_infer.c itself runs (2048 / 1024 streaming frames, aubio semantics): fused and plain, from int16
    from dsp_amd.scrubjay import scrubjay_infer_config
    own = ScrubJay(attrs, config=scrubjay_infer_config(16000))
    for n_clips, n in ((64, 16000), (5, 1024), (6, 15998), (3, 50000)):
        pcm = _pcm(n_clips, n, 500 + n)
        as_f = torch.from_numpy(pcm.astype(np.float32) / np.float32(32768.0)).cuda()
        for a, b in zip(own.pcm16(torch.from_numpy(pcm).cuda(), 500), own(as_f, 500, fused=True)):
            assert torch.equal(a, b), (n_clips, n)
        assert torch.equal(own.plan.clips_pcm16(torch.from_numpy(pcm).cuda(), 500), own.plan.clips(as_f, 500))
        st = _pcm(n_clips, n, 600 + n, stereo=True)
        for a, b in zip(own.pcm16(torch.from_numpy(st).cuda(), 500, stereo_mode=dsp_amd.STEREO_CHANNEL0),
                        own(torch.from_numpy(st[:, :, 0].astype(np.float32) / np.float32(32768.0)).cuda(), 500, fused=True)):
            assert torch.equal(a, b)
    # plans without an int16 form refuse with a reason (the librosa-semantics 2048-point plan of train.py)
    other = ScrubJay(attrs, config=scrubjay_infer_config(16000, aubio=False))
    with pytest.raises(dsp_amd.DspError):
        other.pcm16(torch.from_numpy(_pcm(4, 16000, 1)).cuda(), 500)


def test_classify_signal_on_int16_equals_the_float_path_and_the_reference_clips(golden):
    import torch
    import dsp_amd
    m = dict(golden("stop_model.npz"))
    net = dsp_amd.StopModel(m)
    plan = dsp_amd.MfccPlan(dsp_amd.default_config())
    for n_clips, n in ((300, 16000), (5, 400), (7, 560), (3, 81000)):
        pcm = _pcm(n_clips, n, 300 + n)
        ref = net.classify_signal_batch(plan, torch.from_numpy(pcm.astype(np.float32) / np.float32(32768.0)).cuda())
        got = net.classify_signal_batch_pcm16(plan, torch.from_numpy(pcm).cuda())
        assert torch.equal(got, ref), (n_clips, n)
        st = _pcm(n_clips, n, 400 + n, stereo=True)
        got = net.classify_signal_batch_pcm16(plan, torch.from_numpy(st).cuda(), stereo_mode=dsp_amd.STEREO_CHANNEL0)
        assert torch.equal(got, net.classify_signal_batch(plan, torch.from_numpy(st[:, :, 0].astype(np.float32) / np.float32(32768.0)).cuda()))
    # the reference's own test clips, as its harness reads them (int16), against the compiled reference's probabilities
    g = golden("stop_ref.npz")
    for i in range(7):
        pcm = np.asarray(g[f"clip{i}__pcm"]).astype(np.int16)
        p = net.classify_signal_batch_pcm16(plan, torch.from_numpy(pcm[None, :]).cuda()).cpu().numpy()[0]
        assert abs(p - float(g[f"clip{i}__prob"])) <= 5e-5, i
    # a plan whose shape has no fused form (20 coefficients need another net) and argument checks of the fused default path
    with pytest.raises(dsp_amd.DspError):
        net.classify_signal_batch_pcm16(dsp_amd.MfccPlan(dsp_amd.default_config(n_mfcc=20)), torch.from_numpy(_pcm(4, 16000, 2)).cuda())
    import ctypes as C
    from dsp_amd import lib as L
    x = torch.zeros((4, 16000), device="cuda")
    prob = torch.empty(4, device="cuda")
    assert L.load().dsp_classify_signal_batch_device(plan._h, net._h, x.data_ptr(), 4, 16000, 15000, prob.data_ptr(), None) == -1      # stride < samples
