"""The reference's only in-repo MFCC golden, TEST_MFCC (2fa/audio/word/c/test_mfcc.h:8): librosa-mode
features of data/testing/stop_121417.wav (512-sample frames, 400-tap Hann centred, hop 160, HTK mel with
Slaney norm, power_to_db(ref=1, top_db=80 over the clip), 13 ortho DCT coefficients; exporter
2fa/audio/word/python/export_test_mfcc.py, keyword_classifier.py:42-55).  It does NOT pin compute_mfcc()
 but it pins the librosa-compatible mode of the oracle and of the HIP path."""
import numpy as np
import pytest

from oracle import oracle as O

COMPAT = dict(frame_length=512, hop_length=160, win_length=400, mel_norm=1, log_mode=1)


def _clip(golden):
    k = golden("librosa_mfcc_kat.npz")
    return (k["pcm"] / np.float32(32768.0)).astype(np.float32), k["test_mfcc"]


def test_oracle_reproduces_test_mfcc(golden):
    x, g = _clip(golden)
    m = O.compute_mfcc(x, 1000, O.default_cfg(fft_mode=O.FFT_FLOAT64, **COMPAT))
    assert m.shape == (97, 13) and not g[:, 97:].any()            # zero padded to 1000 columns
    assert np.abs(m.T - g[:, :97]).max() <= 3e-4                  # librosa itself runs in float32
    m32 = O.compute_mfcc(x, 1000, O.default_cfg(**COMPAT))        # reference-order fp32 FFT
    assert np.abs(m32.T - g[:, :97]).max() <= 2e-3                # 2e-3 of values up to 591


@pytest.mark.gpu
def test_gpu_reproduces_test_mfcc(golden):
    import torch
    import dsp_amd
    x, g = _clip(golden)
    plan = dsp_amd.MfccPlan(dsp_amd.default_config(**COMPAT))
    out = plan.clips(torch.from_numpy(x[None]).cuda(), 1000).cpu().numpy()[0]
    assert out.shape == (97, 13)
    assert np.abs(out.T - g[:, :97]).max() <= 5e-4
    truth = O.compute_mfcc(x, 1000, O.default_cfg(fft_mode=O.FFT_FLOAT64, **COMPAT))
    assert np.abs(out - truth).max() <= 3e-4


@pytest.mark.gpu
def test_gpu_global_top_db_batches_and_frames(golden):
    """Clip-global clipping is per clip (a loud clip must not change a quiet neighbour), and independent
    frames behave as one-frame clips."""
    import torch
    import dsp_amd
    from tests import signals as S
    x, _ = _clip(golden)
    clips = np.stack([x, S.uniform_pm1(16000, 5) * np.float32(1e-3), S.chirp(16000, 300.0, 7000.0), np.zeros(16000, np.float32)])
    plan = dsp_amd.MfccPlan(dsp_amd.default_config(**COMPAT))
    out = plan.clips(torch.from_numpy(clips).cuda(), 1000).cpu().numpy()
    ocfg = O.default_cfg(**COMPAT)
    for i in range(4):
        ref = O.compute_mfcc(clips[i], 1000, ocfg)
        assert np.abs(out[i] - ref).max() <= 2e-3 + 1e-5 * np.abs(ref).max(), i
    fr = np.stack([clips[0][160 * t: 160 * t + 512] for t in range(40)])
    got = plan.frames(torch.from_numpy(fr).cuda()).cpu().numpy()
    ref = O.mfcc_frames(fr, ocfg)
    assert np.abs(got - ref).max() <= 2e-3
