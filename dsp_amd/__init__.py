"""dsp_amd -- MI355X-native MFCC / Butterworth / spectrogram hot path of
cornell-c2s2/dsp behind the reference's C entry points (include/dsp_amd.h).

Python here is plumbing over the C ABI (ctypes) plus torch for HBM buffers,
streams and torch.distributed; all arithmetic runs in the HIP kernels of
dsp_amd/csrc.  Importing the package does not need a GPU; calling into it does.
"""
from .lib import DspError, MfccConfig, load  # noqa: F401
from .classify import (CLASSIFY_DONUT_C, CLASSIFY_MICROPHONE, CLASSIFY_MICROPHONE_C, CLASSIFY_SYNC_LIB, butter_bandpass, butter_bandpass_filter, classify, classify_batch,  # noqa: F401
                       classify_batch_f64, classify_batch_f64_pcm16, classify_batch_pcm16, classify_device_pcm16, classify_release, classify_ragged, classify_device_ragged, classify_ragged_f64, classify_device_ragged_f64, classify_config, classify_device, classify_device_f64, classify_device_f64_pcm16,
                       classify_release_f64, classify_stats_f64, compute_spectrogram, find_midpoints, STEREO_AVERAGE, STEREO_CHANNEL0)
from .mfcc import MfccPlan, compute_mfcc, default_config, frames_for, tables  # noqa: F401
from .consumers import SpeakerModel, StopModel, upsample_linear  # noqa: F401

__all__ = ["DspError", "MfccConfig", "MfccPlan", "compute_mfcc", "default_config", "frames_for", "tables", "load",
           "butter_bandpass", "butter_bandpass_filter", "compute_spectrogram", "find_midpoints", "classify", "classify_batch", "classify_batch_f64", "classify_device", "classify_device_f64", "classify_batch_f64_pcm16", "classify_batch_pcm16", "classify_device_pcm16", "classify_release", "classify_ragged", "classify_device_ragged", "classify_ragged_f64", "classify_device_ragged_f64", "classify_device_f64_pcm16", "classify_stats_f64", "classify_release_f64", "STEREO_CHANNEL0", "STEREO_AVERAGE", "classify_config", "CLASSIFY_SYNC_LIB", "CLASSIFY_MICROPHONE", "CLASSIFY_MICROPHONE_C", "CLASSIFY_DONUT_C",
           "StopModel", "SpeakerModel", "upsample_linear"]
