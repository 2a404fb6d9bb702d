"""HIP consumers of the MFCC matrix and the resampler against goldens from the reference's own compiled sources
(tests/golden/stop_ref.npz, speaker_gmm_ref.npz; generator make_golden.py --only consumers) and the oracle.
  stop-word net    stop_detector.c:12-55 + audio_classifier_inference.c:38-90   float: |dP| <= 2e-5 (the reference sums
                   6500 fp32 terms sequentially; the kernel sums in float64)
  speaker GMM      speaker_gmm.c:29-141                                          integer: bit-exact
  upsampleLinear   sync/particle/main.cpp:62-77                                  fp32, same operation order: bit-exact vs oracle"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
PTOL = 2e-5


def _model(golden):
    return dict(golden("stop_model.npz"))


def _gmms(golden):
    s = golden("speaker_gmm_ref.npz")
    t = {k: s[f"target_{k}"] for k in ("means", "inv_covs", "log_consts")}
    u = {k: s[f"ubm_{k}"] for k in ("means", "inv_covs", "log_consts")}
    return s, t, u


def test_stop_net_on_reference_feature_cases(golden):
    import torch
    import dsp_amd
    m, g = _model(golden), golden("stop_ref.npz")
    net = dsp_amd.StopModel(m)
    # the golden cases are full coefficient-major [13][500] feature vectors = frame-major [500][13] matrices
    mf = np.ascontiguousarray(g["feats"].reshape(-1, 13, 500).transpose(0, 2, 1))
    got = net.predict(torch.from_numpy(mf).cuda()).cpu().numpy()
    assert np.abs(got - g["feats_prob"]).max() <= PTOL
    assert 0.3 < got[1] < 0.9 and got[4] > 0.99


def test_classify_signal_on_reference_test_clips(golden):
    import torch
    import dsp_amd
    m, g = _model(golden), golden("stop_ref.npz")
    net = dsp_amd.StopModel(m)
    clips = np.stack([(g[f"clip{i}__pcm"] / np.float32(32768.0)).astype(np.float32) for i in range(7)])
    want = np.array([g[f"clip{i}__prob"] for i in range(7)], np.float32)
    plan = dsp_amd.MfccPlan(dsp_amd.default_config())
    got = net.classify_signal_batch(plan, torch.from_numpy(clips).cuda()).cpu().numpy()
    assert np.abs(got - want).max() <= 5e-5, (got, want)          # MFCC within its parity gate, then the net
    assert (got > 0.5).tolist() == (want > 0.5).tolist()
    one = net.classify_signal(clips[0])                             # host entry point, classify_signal's contract
    assert abs(one - want[0]) <= 5e-5
    assert net.classify_signal(np.zeros(100, np.float32)) == pytest.approx(O.classify_signal(m, np.zeros(100, np.float32)), abs=PTOL)


def test_classify_signal_fused_kernel(golden, monkeypatch):
    """dsp_classify_signal_batch_device runs ONE kernel for the reference's shape (mfcc512_wave_kernel<POOL = 2>: the tile epilogue
    feeds layer 1, the MFCC matrix never reaches HBM).  Against the two-kernel path (same products, another float64
    summation order), the oracle's classify_signal, and on clips longer than max_frames (frames past 500 are dropped) and of one frame."""
    import torch
    import dsp_amd
    m = _model(golden)
    net = dsp_amd.StopModel(m)
    plan = dsp_amd.MfccPlan(dsp_amd.default_config())
    gen = torch.Generator(device="cuda").manual_seed(41)
    for n_clips, n in ((300, 16000), (5, 400), (7, 559), (3, 81000), (9, 16002)):
        # rows on an even stride (the 8-byte alignment the frame loads need); 559: an odd clip length inside such rows
        clips = (torch.rand((n_clips, n + (n & 1)), device="cuda", generator=gen) * 2 - 1)[:, :n]
        clips[::4] *= 0.01
        if n_clips > 2:
            clips[2] = 0.0
        fused = net.classify_signal_batch(plan, clips).cpu().numpy()
        monkeypatch.setenv("DSP_AMD_STOP_TWO_KERNELS", "1")
        two = net.classify_signal_batch(plan, clips).cpu().numpy()
        monkeypatch.delenv("DSP_AMD_STOP_TWO_KERNELS")
        assert np.abs(fused - two).max() <= 1e-6, (n_clips, n)
        for i in (0, 2, n_clips - 1):
            assert abs(fused[i] - O.classify_signal(m, clips[i].cpu().numpy())) <= 5e-5, (n, i)
    # a strided batch (rows padded to an even stride) and a plan whose shape has no fused form (20 coefficients): two kernels, same API
    wide = torch.zeros((8, 16010), device="cuda")
    wide[:, :16000] = torch.rand((8, 16000), device="cuda", generator=gen) * 2 - 1
    a = net.classify_signal_batch(plan, wide[:, :16000]).cpu().numpy()
    b = net.classify_signal_batch(plan, wide[:, :16000].contiguous()).cpu().numpy()
    assert np.array_equal(a, b)


def test_stop_net_shapes_padding_and_truncation(golden):
    import torch
    import dsp_amd
    m = _model(golden)
    net = dsp_amd.StopModel(m)
    rng = np.random.default_rng(5)
    for t in (0, 1, 98, 499, 500, 640):
        mf = (rng.standard_normal((3, max(t, 1), 13)) * 40 - 20).astype(np.float32)[:, :t]
        got = net.predict(torch.from_numpy(np.ascontiguousarray(mf)).cuda().reshape(3, t, 13)).cpu().numpy()
        ref = np.array([O.stop_predict(m, O.stop_features(m, x)) for x in mf], np.float32)
        assert np.abs(got - ref).max() <= PTOL, t


def test_stop_net_random_models_against_oracle():
    import torch
    import dsp_amd
    rng = np.random.default_rng(9)
    for units, n_coef, max_frames in (((4, 2, 2, 1), 13, 500), ((16, 8, 3, 1), 20, 64), ((1, 1, 1, 1), 5, 7)):
        n_in = n_coef * max_frames
        m = {"n_coef": n_coef, "max_frames": max_frames,
             "scaler_mean": rng.standard_normal(n_in).astype(np.float32), "scaler_scale": (rng.random(n_in) + 0.5).astype(np.float32)}
        m["scaler_scale"][::17] = 0.0                                # the reference's divide-by-zero guard
        fan = n_in
        for i, u in enumerate(units):
            m[f"kernel{i}"] = (rng.standard_normal(fan * u) / np.sqrt(fan)).astype(np.float32)
            m[f"bias{i}"] = (rng.standard_normal(u) * 0.1).astype(np.float32)
            fan = u
        net = dsp_amd.StopModel(m)
        mf = rng.standard_normal((5, max_frames - 3, n_coef)).astype(np.float32)
        got = net.predict(torch.from_numpy(mf).cuda()).cpu().numpy()
        ref = np.array([O.stop_predict(m, O.stop_features(m, x)) for x in mf], np.float32)
        assert np.abs(got - ref).max() <= PTOL, units


def test_speaker_gmm_bit_exact(golden):
    import torch
    import dsp_amd
    s, t, u = _gmms(golden)
    spk = dsp_amd.SpeakerModel(t, u)
    for i in range(4):
        mf = s[f"clip{i}__mfcc"]
        mean, label, lt, lu = spk.llr(torch.from_numpy(mf[None]).cuda(), per_frame=True)
        assert np.array_equal(lt.cpu().numpy()[0], s[f"clip{i}__ll_target"])
        assert np.array_equal(lu.cpu().numpy()[0], s[f"clip{i}__ll_ubm"])
        assert int(mean[0]) == int(s[f"clip{i}__llr_mean"]) and int(label[0]) == int(s[f"clip{i}__label"])
    mean, label = spk.llr(torch.from_numpy(s["synth__mfcc"][None]).cuda())
    assert int(mean[0]) == int(s["synth__llr_mean"]) and int(label[0]) == int(s["synth__label"]) == 1


def test_speaker_gmm_batch_and_extreme_values(golden):
    import torch
    import dsp_amd
    s, t, u = _gmms(golden)
    spk = dsp_amd.SpeakerModel(t, u)
    rng = np.random.default_rng(2)
    mf = (rng.standard_normal((37, 130, 13)) * 2).astype(np.float32)
    mf[3, :, 0] = 600.0          # x * 64 beyond int16: the reference keeps the low 16 bits
    mf[4, 7, :] = -513.0
    mean, label = spk.llr(torch.from_numpy(mf).cuda())
    ref = np.array([O.speaker_llr_mean(t, u, x) for x in mf], np.int64)
    assert np.array_equal(mean.cpu().numpy(), ref)
    assert np.array_equal(label.cpu().numpy(), np.array([O.classify_speaker(t, u, x) for x in mf], np.int32))


def test_mfcc_to_speaker_pipeline_matches_reference_chain(golden):
    """PCM -> MFCC (HIP) -> Q6 -> GMM: the decision of the reference chain on the reference's clips."""
    import torch
    import dsp_amd
    s, t, u = _gmms(golden)
    g = golden("stop_ref.npz")
    spk = dsp_amd.SpeakerModel(t, u)
    plan = dsp_amd.MfccPlan(dsp_amd.default_config())
    clips = np.stack([(g[f"clip{i}__pcm"] / np.float32(32768.0)).astype(np.float32) for i in range(4)])
    mf = plan.clips(torch.from_numpy(clips).cuda(), 500)
    mean, label = spk.llr(mf)
    want = np.array([s[f"clip{i}__llr_mean"] for i in range(4)], np.int64)
    # the MFCC differs from the reference's within its fp32 noise floor, so a Q6 feature may flip by one count:
    # the frame-mean LLR moves by a few Q8 counts at most
    assert np.abs(mean.cpu().numpy() - want).max() <= 8, (mean, want)
    assert label.cpu().numpy().tolist() == [int(s[f"clip{i}__label"]) for i in range(4)]


def test_upsample_linear_bit_exact():
    import torch
    import dsp_amd
    rng = np.random.default_rng(4)
    for old, new in ((8000, 16000), (7, 19), (100, 100), (2, 5), (16000, 48000), (1, 4)):
        x = rng.standard_normal((3, old)).astype(np.float32)
        got = dsp_amd.upsample_linear(torch.from_numpy(x).cuda(), new).cpu().numpy()
        ref = np.stack([O.upsample_linear(r, new) for r in x])
        assert np.array_equal(got, ref), (old, new)
    one = dsp_amd.upsample_linear(x[0], 33)                           # host entry point
    assert np.array_equal(one, O.upsample_linear(x[0], 33))
