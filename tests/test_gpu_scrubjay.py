"""GPU: pooling (scrubjay_infer.c:36-66) and the decoded ONNX SVM through the C ABI against libsvm's own answers
(tests/golden/svm_libsvm_ref.npz, tools/pin_svm_libsvm.py) and the CPU oracle; floating point -> tolerance 2e-5 on
decision / probability, labels (the pairwise vote, decision > 0 -> class 0) equal away from decision = 0.  The
probability is libsvm's multiclass_probability ITERATION, whose stopping test makes it jump by up to 0.005 / 2 when an
input moves by an ulp across a stopping boundary: a comparison is "close" within 2e-5, or within that jump.
The aubio front end is unpinned: the features here come from this library's own MFCC chain with n_mfcc = 20."""
import numpy as np
import pytest

from tests import signals as S

pytestmark = pytest.mark.gpu


def _prob_close(p, q):
    """-> (ok, on_a_boundary): 2e-5, or the iteration's own tolerance when the stopping test fell the other way."""
    d = abs(float(p) - float(q))
    return d <= 2.6e-3, d > 2e-5


def _model(m):
    model = {k: m[k] for k in ("offset", "scale", "sv", "coef")}
    model.update(gamma=float(m["kernel_params"][0]), rho=float(m["rho"][0]), prob_a=float(m["prob_a"][0]), prob_b=float(m["prob_b"][0]))
    return model


def test_mfcc_stats_vs_oracle():
    import torch
    from dsp_amd import scrubjay
    from oracle import oracle as O
    for (n, t, c) in ((5, 98, 20), (3, 1, 13), (2, 148, 13), (70, 7, 20)):
        m = (S.uniform_pm1(n * t * c, 300 + n).reshape(n, t, c) * np.float32(40.0)).astype(np.float32)
        m[0, :, 0] = 3.25                               # constant column -> std exactly 0
        got = scrubjay.mfcc_stats(torch.from_numpy(m).cuda()).cpu().numpy()
        want = np.stack([O.mfcc_stats(m[i]) for i in range(n)])
        assert np.array_equal(got, want)                # same float64 accumulation order -> bit exact
        assert got[0, c] == 0.0


def test_svm_vs_oracle(golden):
    import torch
    from dsp_amd import scrubjay
    from oracle import oracle as O
    m = golden("scrubjay_svm.npz")
    svm = scrubjay.SvmModel({k: m[k] for k in m.files})
    x = np.stack([(m["offset"] + S.uniform_pm1(40, 900 + i) * (2.5 / m["scale"])).astype(np.float32) for i in range(300)])
    labels, dec, p1 = (a.cpu().numpy() for a in svm.predict(torch.from_numpy(x).cuda()))
    seen, loose = set(), 0
    for i in range(x.shape[0]):
        lab, odec, op1 = O.svm_predict(_model(m), x[i])
        assert abs(dec[i] - odec) <= 2e-5 * max(1.0, abs(odec))
        ok, edge = _prob_close(p1[i], op1)
        assert ok
        loose += edge
        if abs(odec) > 1e-5:
            assert labels[i] == lab
        seen.add(int(labels[i]))
    assert seen == {0, 1} and loose <= 3


def test_svm_vs_libsvm(golden):
    """The HIP SVM against libsvm itself (sklearn.svm._libsvm fed the decoded ONNX attributes): decision values,
    predict_proba and svm_predict's vote label on 320 vectors, 64 of them inside the zone where a sigmoid + arg-max reading
    of the attributes would give the other label / another probability; plus the reference's 13 labelled WAVs (features
    from a numpy restatement of train.py's librosa call), which libsvm -- and therefore this kernel -- labels 13 / 13."""
    import torch
    from dsp_amd import scrubjay
    m = golden("scrubjay_svm.npz")
    r = golden("svm_libsvm_ref.npz")
    svm = scrubjay.SvmModel({k: m[k] for k in m.files})
    labels, dec, p1 = (a.cpu().numpy() for a in svm.predict(torch.from_numpy(r["feat"]).cuda()))
    assert np.abs(dec - r["decision"]).max() <= 2e-5
    firm = np.abs(r["decision"]) > 1e-5
    assert np.array_equal(labels[firm], r["label_vote"][firm]) and firm.sum() >= 300
    sliver = (r["decision"] > 1e-5) & (r["decision"] < 0.0079)
    assert sliver.sum() >= 16 and np.all(labels[sliver] == 0)            # sigmoid arg max says 1 here; libsvm (and ORT's votes) 0
    d = np.abs(p1 - r["proba"][:, 1])
    assert d.max() <= 2.6e-3 and (d > 2e-5).sum() <= 3
    labels, dec, p1 = (a.cpu().numpy() for a in svm.predict(torch.from_numpy(r["labelled_feat"]).cuda()))
    assert np.array_equal(labels, r["labelled_y"]) and np.array_equal(labels, r["labelled_vote"])      # polarity: 1 = scrub jay
    assert np.abs(dec - r["labelled_decision"]).max() <= 2e-5 and np.abs(p1 - r["labelled_proba"][:, 1]).max() <= 2e-5


def test_clip_to_label_pipeline(golden):
    """scrubjay_infer main loop: clips -> MFCC(20) -> mean|std -> SVM, against the oracle chain."""
    import torch
    from dsp_amd import scrubjay
    from oracle import oracle as O
    from tests.conftest import ATOL_DB, RTOL, frame_linf_close
    m = golden("scrubjay_svm.npz")
    sj = scrubjay.ScrubJay({k: m[k] for k in m.files})
    clips = np.stack([S.uniform_pm1(16000, 70), S.chirp(16000, 300.0, 6000.0), S.uniform_pm1(16000, 71) * np.float32(0.01),
                      S.classify_cases()["jay_like"]])
    labels, dec, p1, feat = sj(torch.from_numpy(clips).cuda())
    feat = feat.cpu().numpy()
    ocfg = O.default_cfg(n_mfcc=20)
    for i in range(clips.shape[0]):
        omfcc = O.compute_mfcc(clips[i], 1 << 20, ocfg)
        ofeat = O.mfcc_stats(omfcc)
        # pooled features inherit the MFCC gate (mean / std of values that each meet it)
        assert np.all(np.abs(feat[i] - ofeat) <= RTOL * np.abs(omfcc).max() + ATOL_DB)
        lab, odec, op1 = O.svm_predict(_model(m), feat[i])          # SVM checked on the SAME features
        assert abs(dec[i].item() - odec) <= 2e-5 * max(1.0, abs(odec)) and _prob_close(p1[i].item(), op1)[0]
        if abs(odec) > 1e-5:
            assert labels[i].item() == lab


def test_config5_at_scale(golden):
    """BASELINE config 5 at 100 000 clips (6.4 GB of PCM in HBM; the full 1 M is ten such batches): properties that
    need no oracle pass + the oracle chain on a few clips."""
    import torch
    from dsp_amd import scrubjay
    from oracle import oracle as O
    free, _total = torch.cuda.mem_get_info()
    if free < 12 << 30:
        pytest.skip("needs ~10 GB of free HBM")
    m = golden("scrubjay_svm.npz")
    sj = scrubjay.ScrubJay({k: m[k] for k in m.files})
    n = 100_000
    gen = torch.Generator(device="cuda").manual_seed(5)
    clips = torch.rand((n, 16000), device="cuda", generator=gen) * 2 - 1
    clips[::7] *= 0.001                                            # quiet clips
    clips[::1009, 4000:] = 0.0                                     # clips that fall silent
    labels, dec, p1, feat = sj(clips)
    assert tuple(feat.shape) == (n, 40) and bool(torch.isfinite(feat).all()) and bool(torch.isfinite(p1).all())
    assert bool(((p1 >= 0) & (p1 <= 1)).all())
    l2, d2, q2, f2 = sj(clips)
    assert torch.equal(labels, l2) and torch.equal(dec, d2) and torch.equal(feat, f2)          # deterministic
    perm = torch.randperm(n, device="cuda", generator=gen)
    l3, d3, q3, f3 = sj(clips[perm].contiguous())
    assert torch.equal(f3, feat[perm]) and torch.equal(d3, dec[perm]) and torch.equal(l3, labels[perm])
    ocfg = O.default_cfg(n_mfcc=20)
    for i in (0, 7, 1009, 99_999):
        of = O.mfcc_stats(O.compute_mfcc(clips[i].cpu().numpy(), 1 << 20, ocfg))
        lab, odec, op1 = O.svm_predict(_model(m), feat[i].cpu().numpy())                      # SVM on the GPU's own features
        assert np.abs(feat[i].cpu().numpy() - of).max() <= 1e-4 * np.abs(of).max() + 3e-4
        assert abs(float(dec[i]) - odec) <= 2e-5 * max(1.0, abs(odec)) and _prob_close(float(p1[i]), op1)[0]
        if abs(odec) > 1e-5:
            assert int(labels[i]) == lab


def test_fused_kernel_equals_the_three_kernel_path(golden):
    """BASELINE config 5 "in one kernel": pooled features, decision values, probabilities and labels of
    dsp_scrubjay_fused_device are those of MFCC -> mfcc_stats -> SVM (same arithmetic in the same order)."""
    import torch
    from dsp_amd import scrubjay
    m = golden("scrubjay_svm.npz")
    sj = scrubjay.ScrubJay({k: m[k] for k in m.files})
    gen = torch.Generator(device="cuda").manual_seed(17)
    for n, samples in ((1, 16000), (5, 16000), (700, 16000), (33, 8000), (64, 1200), (3, 400)):
        clips = torch.rand((n, samples), device="cuda", generator=gen) * 2 - 1
        clips[::3] *= 0.01
        if n > 4:
            clips[4] = 0.0                                           # a silent clip: all coefficients 0, std 0
        a = sj(clips, fused=False)
        b = sj(clips, fused=True)
        assert torch.equal(a[3], b[3]), (n, samples)                 # features
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]), (n, samples)
    # strided rows and a max_frames clamp
    wide = torch.rand((40, 16384), device="cuda", generator=gen) * 2 - 1
    a = sj(wide[:, :16000].contiguous(), 50, fused=False)
    b = sj(wide[:, :16000], 50, fused=True)
    assert torch.equal(a[3], b[3]) and torch.equal(a[0], b[0])


def test_n_fft_2048_scrubjay_infer_framing(golden):
    """The parameterisation of cepstrum/scrubjay_infer.c itself (:10-14: WIN_SIZE 2048, HOP_SIZE 1024, 40 filters, 20
    coefficients): the 2048-point MFCC kernel against the oracle at that shape (clips, independent frames, a short frame
    length, another sample rate / 128 mels), and the fused clip -> label kernel against the oracle chain and the three-kernel
    path.  (aubio's filterbank and scaling are not vendored: what is pinned is the chain's arithmetic at this framing.)"""
    import torch
    import dsp_amd
    from dsp_amd import scrubjay
    from oracle import oracle as O
    from tests.conftest import gate
    m = golden("scrubjay_svm.npz")
    for sr, extra in ((16000, {}), (44100, {}), (16000, dict(n_mels=128, n_mfcc=32, frame_length=1600, hop_length=800))):
        cfg = scrubjay.scrubjay_infer_config(sr, aubio=False)       # the file's numbers on mfcc.c semantics; aubio's own: next test
        over = dict(sample_rate=sr, n_fft=2048, frame_length=2048, hop_length=1024, n_mels=40, n_mfcc=20, fmin=0.0, fmax=sr / 2.0)
        over.update(extra)
        for k, v in extra.items():
            setattr(cfg, k, v)
        plan = dsp_amd.MfccPlan(cfg)
        ocfg = O.default_cfg(**over)
        x = np.stack([S.uniform_pm1(24000, 60), S.chirp(24000, 300.0, 7000.0), S.uniform_pm1(24000, 61) * np.float32(0.01)])
        x[2, 6000:] = 0.0
        out = plan.clips(torch.from_numpy(x).cuda(), 500).cpu().numpy()
        for i in range(3):
            ref = O.compute_mfcc(x[i], 500, ocfg)
            assert out[i].shape == ref.shape
            gate(out[i], ref, f"n_fft 2048 sr {sr} {extra}/clip{i}")
        if not extra:
            fr = S.uniform_pm1(2048 * 70, 62).reshape(70, 2048)
            fr[3] = 0.0
            fplan = dsp_amd.MfccPlan(dsp_amd.default_config(**dict(over, hop_length=2048)))
            got = fplan.frames(torch.from_numpy(fr).cuda()).cpu().numpy()
            gate(got, O.mfcc_frames(fr, O.default_cfg(**dict(over, hop_length=2048)), threads=4), f"n_fft 2048 frames sr {sr}")
            assert not got[3].any()
    # fused: clip -> MFCC(2048/1024/40/20) -> mean | std -> SVM
    sj = scrubjay.ScrubJay({k: m[k] for k in m.files}, config=scrubjay.scrubjay_infer_config(16000, aubio=False))
    gen = torch.Generator(device="cuda").manual_seed(23)
    clips = torch.rand((300, 16000), device="cuda", generator=gen) * 2 - 1
    clips[::3] *= 0.01
    clips[4] = 0.0
    a = sj(clips, fused=False)
    b = sj(clips, fused=True)
    assert torch.equal(a[3], b[3]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0], b[0])
    ocfg = O.default_cfg(n_fft=2048, frame_length=2048, hop_length=1024, n_mels=40, n_mfcc=20)
    model = _model(m)
    for i in (0, 3, 4, 299):
        omfcc = O.compute_mfcc(clips[i].cpu().numpy(), 1 << 20, ocfg)
        assert omfcc.shape == (14, 20)
        ofeat = O.mfcc_stats(omfcc)
        feat = b[3][i].cpu().numpy()
        assert np.all(np.abs(feat - ofeat) <= 1e-4 * np.abs(omfcc).max() + (3e-4 if np.abs(omfcc).max() < 3.0 else 0.0))
        lab, odec, op1 = O.svm_predict(model, feat)
        assert abs(float(b[1][i]) - odec) <= 2e-5 * max(1.0, abs(odec)) and _prob_close(float(b[2][i]), op1)[0]
        if abs(odec) > 1e-5:
            assert int(b[0][i]) == lab


def test_aubio_semantics_front_end_of_scrubjay_infer(golden):
    """cepstrum/scrubjay_infer.c:21-53 as the file runs it: aubio_source_do -> aubio_pvoc_do -> aubio_mfcc_do per hop, pooled
    (:36-66) and classified (:105-141).  GPU (dsp_mfcc_scrubjay_infer_config on mfcc2048_kernel<AUB>, plain and fused) against
    the oracle's restatement of aubio 0.4 (oracle/aubio_oracle.c) under the 1e-4 * L-inf gate.  PARITY UNPINNED at the aubio
    boundary (the library is not in the image, the reference holds no vector there); pinned below it: pooling and SVM."""
    import torch
    import dsp_amd
    from dsp_amd import scrubjay
    from oracle import oracle as O
    from tests.conftest import gate
    m = golden("scrubjay_svm.npz")
    cfg = scrubjay.scrubjay_infer_config(16000)
    plan = dsp_amd.MfccPlan(cfg)
    # the frame count is the do/while's (:39-53): 16 frames for a 16 000-sample clip, not 14
    for n, t in ((1, 1), (1024, 1), (1025, 2), (16000, 16), (16384, 16), (16385, 17)):
        assert dsp_amd.frames_for(cfg, n, 1 << 20) == t == O.aubio_frames_for(n, 1024)
    noise = S.uniform_pm1(16000, 70)
    cases = {
        "noise": noise,
        "chirp+noise": S.chirp(16000, 200.0, 6000.0) + np.float32(1e-3) * S.uniform_pm1(16000, 71),
        "quiet": noise * np.float32(1e-3),
        "silent tail": np.concatenate([noise[:5000], np.zeros(11000, np.float32)]),
    }
    x = np.stack(list(cases.values()))
    out = plan.clips(torch.from_numpy(x).cuda(), 1 << 20).cpu().numpy()
    assert out.shape == (4, 16, 20)
    for i, name in enumerate(cases):
        gate(out[i], O.aubio_mfcc_clip(x[i]), f"aubio front end/{name}")
    # silent frames sit on SAFE_LOG10's floor: log10(2e-42) in every filter -> c0 = sqrt(40) * floor, the rest 0
    assert np.abs(out[3, 8:, 0] - np.sqrt(40.0) * np.log10(np.float64(np.float32(2e-42)))).max() < 1e-3 and np.abs(out[3, 8:, 1:]).max() < 1e-4
    # ragged lengths: zero history in front, the zero-padded short last hop, odd lengths (host entry point packs the stride)
    for n in (1, 2, 1023, 1025, 2500, 4097, 15999):
        sig = S.uniform_pm1(n, 80 + n)
        got = plan.clips_host(sig, 1 << 20)[0]
        ref = O.aubio_mfcc_clip(sig)
        assert got.shape == ref.shape == (-(-n // 1024), 20)
        gate(got, ref, f"aubio front end/len{n}")
    # max_frames caps the clip like any other framing; another sample rate (the reference's recordings are 96 kHz) moves the bank
    assert torch.equal(plan.clips(torch.from_numpy(x).cuda(), 5), torch.from_numpy(out[:, :5]).cuda())
    p96 = dsp_amd.MfccPlan(scrubjay.scrubjay_infer_config(96000))
    sig = S.uniform_pm1(30000, 90)
    gate(p96.clips_host(sig, 1 << 20)[0], O.aubio_mfcc_clip(sig, sample_rate=96000), "aubio front end/96 kHz")
    # fused clip -> label: same features as MFCC -> pooling -> SVM in three kernels, and as the oracle chain
    sj = scrubjay.ScrubJay({k: m[k] for k in m.files}, config=cfg)
    gen = torch.Generator(device="cuda").manual_seed(29)
    clips = torch.rand((200, 16000), device="cuda", generator=gen) * 2 - 1
    clips[::3] *= 0.01
    clips[4, 3000:] = 0.0
    a = sj(clips, fused=False)
    b = sj(clips, fused=True)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    model = _model(m)
    for i in (0, 3, 4, 199):
        omfcc = O.aubio_mfcc_clip(clips[i].cpu().numpy())
        assert omfcc.shape == (16, 20)
        ofeat = O.mfcc_stats(omfcc)
        feat = b[3][i].cpu().numpy()
        assert np.all(np.abs(feat - ofeat) <= 1e-4 * np.abs(omfcc).max())
        lab, odec, op1 = O.svm_predict(model, feat)
        assert abs(float(b[1][i]) - odec) <= 2e-5 * max(1.0, abs(odec)) and _prob_close(float(b[2][i]), op1)[0]
        if abs(odec) > 1e-5:
            assert int(b[0][i]) == lab


@pytest.mark.parametrize("n_mfcc,n_mels,fmax", [(13, 40, 8000.0), (16, 64, 3000.0), (8, 24, 8000.0)])
def test_fused_kernel_at_other_coefficient_counts(n_mfcc, n_mels, fmax):
    """The fused clip -> label kernel is instantiated per DCT shape; shapes with at most 16 coefficients take other
    instantiations (one coefficient tile) than config 5's 20.  A synthetic RBF model of 2 n_mfcc features: fused equals the
    three-kernel path bit for bit, and the pooled features equal the oracle's mfcc_stats of the oracle-gated MFCC."""
    import torch
    import dsp_amd
    from dsp_amd import scrubjay
    from oracle import oracle as O
    from tests.conftest import gate
    rng = np.random.default_rng(100 + n_mfcc)
    nf, nsv = 2 * n_mfcc, 37
    attrs = dict(offset=rng.normal(0, 5, nf).astype(np.float32), scale=(1.0 / rng.uniform(2, 20, nf)).astype(np.float32),
                 sv=rng.normal(0, 1, (nsv, nf)).astype(np.float32), coef=rng.normal(0, 1, nsv).astype(np.float32),
                 kernel_params=np.array([0.03, 0.0, 3.0], np.float32), rho=np.array([0.1], np.float32),
                 prob_a=np.array([-2.0], np.float32), prob_b=np.array([0.05], np.float32), vectors_per_class=np.array([20, 17], np.int64))
    cfg = dsp_amd.default_config(n_mfcc=n_mfcc, n_mels=n_mels, fmax=fmax)       # 64 filters fit the 64 chunk lanes only below ~3 kHz
    sj = scrubjay.ScrubJay(attrs, config=cfg)
    gen = torch.Generator(device="cuda").manual_seed(23 + n_mfcc)
    clips = torch.rand((70, 16000), device="cuda", generator=gen) * 2 - 1
    clips[::4] *= 0.003
    clips[5] = 0.0
    a = sj(clips, fused=False)
    b = sj(clips, fused=True)
    for i in range(4):
        assert torch.equal(a[i], b[i]), i
    ocfg = O.default_cfg(n_mfcc=n_mfcc, n_mels=n_mels, fmax=fmax)
    x = clips[:3].cpu().numpy()
    feats = b[3].cpu().numpy()
    for i in range(3):
        ref = O.mfcc_stats(O.compute_mfcc(x[i], 500, ocfg))
        gate(feats[i][None, :], ref[None, :], f"fused n_mfcc {n_mfcc} / {n_mels} mel pooled features clip{i}")


def test_librosa_default_framing_on_the_2048_kernel():
    """cepstrum/train.py:45-52 computes its features with librosa's defaults: n_fft 2048, hop 512, centred frames, Hann,
    128 mel filters on Slaney's scale with Slaney's normalisation (DSP_MELNORM_LIBROSA), power_to_db(ref = 1, top_db = 80 below the CLIP's maximum), 20 ortho DCT-II
    coefficients, then mean | std.  DSP_LOG_GLOBAL_REF1 on the 2048-point kernel: against the oracle (gate), and the pooled
    features against the float64 numpy restatement of that librosa call that pinned the SVM's polarity
    (tools/pin_svm_libsvm.py:librosa_like_features) -- an independent implementation of the same definition."""
    import importlib.util
    import os
    import torch
    import dsp_amd
    from dsp_amd.lib import MELNORM_LIBROSA, LOG_GLOBAL_REF1
    from oracle import oracle as O
    from tests.conftest import gate
    spec = importlib.util.spec_from_file_location("pin_svm", os.path.join(os.path.dirname(__file__), "..", "tools", "pin_svm_libsvm.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    for sr in (16000, 22050):
        over = dict(sample_rate=sr, n_fft=2048, frame_length=2048, hop_length=512, n_mels=128, n_mfcc=20, fmin=0.0, fmax=sr / 2.0,
                    mel_norm=MELNORM_LIBROSA, log_mode=LOG_GLOBAL_REF1)
        plan = dsp_amd.MfccPlan(dsp_amd.default_config(**over))
        ocfg = O.default_cfg(**over)
        n = (3 * sr // 2) & ~1                                   # even clip stride (8-byte aligned frame loads)
        ys = [S.uniform_pm1(n, 80 + sr % 7) * np.float32(0.3), S.chirp(n, 150.0, 0.4 * sr, fs=float(sr)).astype(np.float32),
              (S.uniform_pm1(n, 81) * np.float32(0.002)).astype(np.float32)]
        ys[2][: n // 3] += S.chirp(n // 3, 500.0, 3000.0, fs=float(sr)).astype(np.float32)       # a loud start, a near-silent rest: the clip floor bites
        padded = np.stack([np.pad(y, 1024) for y in ys]).astype(np.float32)                      # librosa center=True, constant padding
        out = plan.clips(torch.from_numpy(padded).cuda(), 1000).cpu().numpy()
        for i, y in enumerate(ys):
            ref = O.compute_mfcc(padded[i], 1000, ocfg)
            assert out[i].shape == ref.shape == (1 + (padded.shape[1] - 2048) // 512, 20)
            gate(out[i], ref, f"librosa defaults sr {sr} clip{i}")
            feat = np.concatenate([out[i].mean(axis=0), out[i].std(axis=0)])
            want = pin.librosa_like_features(y.astype(np.float64), sr)
            err = np.abs(feat - want).max() / np.abs(want).max()
            assert err < 2e-4, (sr, i, err)


def test_end_to_end_on_the_reference_recordings(golden):
    """Two of the reference's own labelled recordings (cepstrum/testing/P_1363v2-sj-short.WAV, N_1809v2-not-sj.wav; fixture
    tests/golden/labelled_audio.npz) through the GPU path end to end as cepstrum/train.py / run.py define it: stereo int16 -> mono
    average -> centred frames -> MFCC with librosa's defaults on the 2048-point kernel (96 kHz) -> mean | std -> Scaler -> RBF-SVM.
    The 40 features against what the float64 restatement of the librosa call gives for the same file, decision value and
    probability against libsvm's for that vector, and the label against the FILE's label (P_ = scrub jay)."""
    import torch
    import dsp_amd
    from dsp_amd import scrubjay
    from dsp_amd.lib import MELNORM_LIBROSA, LOG_GLOBAL_REF1
    g = golden("labelled_audio.npz")
    m = golden("scrubjay_svm.npz")
    svm = scrubjay.SvmModel({k: m[k] for k in m.files})
    for name in ("sj_short", "not_sj"):
        pcm, sr, label = g[f"{name}__pcm"], int(g[f"{name}__sr"]), int(g[f"{name}__label"])
        y = (pcm.astype(np.float64) / 32768.0).mean(axis=1).astype(np.float32)          # librosa.load(sr=None, mono=True)
        y = np.pad(y, 1024)                                                              # center=True, constant padding
        if y.size & 1:
            y = y[:-1]                                                                   # an even stride; the dropped sample lies past the last frame or is padding
        cfg = dsp_amd.default_config(sample_rate=sr, n_fft=2048, frame_length=2048, hop_length=512, n_mels=128, n_mfcc=20, fmin=0.0,
                                     fmax=sr / 2.0, mel_norm=MELNORM_LIBROSA, log_mode=LOG_GLOBAL_REF1)
        mf = dsp_amd.MfccPlan(cfg).clips(torch.from_numpy(y[None, :]).cuda(), 100000)
        assert mf.shape[1] == 1 + pcm.shape[0] // 512
        feat = scrubjay.mfcc_stats(mf)                                                   # scrubjay_infer.c:36-66 pooling, float64 sums
        want = g[f"{name}__feat"]
        err = np.abs(feat.cpu().numpy()[0] - want).max() / np.abs(want).max()
        assert err < 3e-4, (name, err)
        labels, decision, prob1 = svm.predict(feat.contiguous())
        assert int(labels[0]) == label == int(g[f"{name}__vote"])
        assert abs(float(decision[0]) - float(g[f"{name}__decision"])) < 5e-3
        assert abs(float(prob1[0]) - float(g[f"{name}__proba"][1])) < 5e-3


def test_fused_kernel_with_many_support_vectors_and_a_partial_last_block(golden):
    """A model with more support vectors than one wavefront copies into the block's LDS table (the reference's has 55), on batches
    whose last block has idle wavefronts: every thread of a block must have written its share of the table before an idle wave leaves
    (round 4: the idle waves left first; harmless only while the table was at most 64 entries long)."""
    import torch
    from dsp_amd import scrubjay
    m = {k: np.array(v) for k, v in golden("scrubjay_svm.npz").items()}
    rng = np.random.default_rng(5)
    reps = 7                                                          # 385 support vectors
    m["sv"] = np.concatenate([m["sv"] + rng.standard_normal(m["sv"].shape).astype(np.float32) * 0.05 * k for k in range(reps)]).astype(np.float32)
    m["coef"] = np.concatenate([m["coef"] * (1.0 - 0.1 * k) for k in range(reps)]).astype(np.float32)
    m["vectors_per_class"] = (np.asarray(m["vectors_per_class"]) * reps).astype(m["vectors_per_class"].dtype)
    sj = scrubjay.ScrubJay(m)
    gen = torch.Generator(device="cuda").manual_seed(18)
    for n in (1, 2, 5, 7, 1021):
        clips = torch.rand((n, 16000), device="cuda", generator=gen) * 2 - 1
        clips[::3] *= 0.01
        a = sj(clips, fused=False)
        b = sj(clips, fused=True)
        for x, y in zip(a, b):
            assert torch.equal(x, y), n
