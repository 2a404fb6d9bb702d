"""The measured dead ends of the 512-point kernel (dsp_amd/csrc/mfcc_row_kernel.hip, mfcc512_pair_kernel.hip): outside the default build
(dsp_amd/build.py EXPERIMENT_SOURCES); these tests run only against a library built with DSP_AMD_EXPERIMENTS=1."""
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
    from tests.test_gpu_mfcc import _cases as cases
    return cases(g)


@pytest.fixture(autouse=True)
def _needs_experiments(dsp):
    if b"+experiments" not in dsp.load().dsp_version():
        pytest.skip("default build: rebuild with DSP_AMD_EXPERIMENTS=1 python -m dsp_amd.build to run the experiment kernels")


def test_row_per_frame_kernel_matches_wave_kernel(dsp, torch_cuda, golden):
    """The alternative kernel form (one 16-lane row per frame, 4 frames per wave) meets the same
    gates, handles ragged frame counts, and agrees with the default form to rounding."""
    from oracle import oracle as O
    torch = torch_cuda
    g = golden("mfcc_ref.npz")
    plan = dsp.MfccPlan()
    plan.set_kernel(1)
    for name in ("noise0", "chirp", "silence", "tiny", "len400", "len560", "birdq_ch0", "stop_121417"):
        x = _cases(g)[name]
        got = plan.clips_host(x, 500)[0]
        gate(got, g["mfcc__" + name], f"row-kernel golden/{name}", floor_case=name if name in LOW_LEVEL_CASES else None)
    fcfg = dsp.default_config(frame_length=512, hop_length=512)
    a, b = dsp.MfccPlan(fcfg), dsp.MfccPlan(fcfg)
    b.set_kernel(1)
    for n in (1, 2, 3, 4, 5, 63, 4099):
        fr = S.uniform_pm1(512 * n, 500 + n).reshape(n, 512)
        if n > 4:
            fr[2] = 0.0
        x = torch.from_numpy(fr).cuda()
        ya, yb = a.frames(x).cpu().numpy(), b.frames(x).cpu().numpy()
        assert np.abs(ya - yb).max() <= 2e-4
        ref = O.mfcc_frames(fr, O.default_cfg(frame_length=512, hop_length=512), threads=4)
        gate(yb, ref, f"row-kernel frames/{n}")


def test_two_frames_per_wave_kernel_matches_the_oracle(dsp, torch_cuda):
    """DSP_KERNEL_PAIR (mfcc512_pair_kernel.hip, an experiment kept selectable): two independent frames per wavefront step on
    the radix-8 pipeline of the 1024-point kernel.  Same gate against the oracle on ragged frame counts (odd counts: the last
    frame rides alone), a silent and a quiet frame beside loud ones (the frames of a pair never mix), and agreement with the
    default form to rounding; other shapes / clip mode fall back to the default kernel."""
    from oracle import oracle as O
    torch = torch_cuda
    fcfg = dsp.default_config(frame_length=512, hop_length=512)
    a, b = dsp.MfccPlan(fcfg), dsp.MfccPlan(fcfg)
    b.set_kernel(3)
    for n in (1, 2, 3, 15, 16, 17, 33, 4099):
        fr = S.uniform_pm1(512 * n, 700 + n).reshape(n, 512)
        if n > 4:
            fr[2] = 0.0
            fr[3] *= np.float32(1e-4)
        x = torch.from_numpy(fr).cuda()
        ya, yb = a.frames(x).cpu().numpy(), b.frames(x).cpu().numpy()
        assert np.abs(ya - yb).max() <= 2e-4
        if n > 4:
            assert not yb[2].any()                                # the silent frame: exact zeros whatever its partner holds
        ref = O.mfcc_frames(fr, O.default_cfg(frame_length=512, hop_length=512), threads=4)
        gate(yb, ref, f"pair-kernel frames/{n}")
    clips = torch.from_numpy(np.stack([S.uniform_pm1(16000, 31), S.chirp(16000, 300.0, 7000.0)])).cuda()
    c = dsp.MfccPlan()
    c.set_kernel(3)                                               # clip mode: the default kernel runs
    assert torch.equal(c.clips(clips, 500), dsp.MfccPlan().clips(clips, 500))


