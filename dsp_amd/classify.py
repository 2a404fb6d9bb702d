"""Host-side mirror of the donut classifier interface (sync/lib/classifier.h:14-19,
donut-classifier/classifier.c:319-592) over the C ABI: same function names and
argument meaning; numpy arrays stand in for the caller-owned / malloc'd C buffers.
All arithmetic runs in the HIP kernels of dsp_amd/csrc/classify_kernels.hip.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def butter_bandpass(lowcut: float, highcut: float):
    """-> (ok, b[9], a[9]) float64; ok False for any band but (1000,3000) / (3000,7500)
    (classifier.c:402-407)."""
    b = (C.c_double * 9)()
    a = (C.c_double * 9)()
    ok = _lib.load().dsp_butter_bandpass(float(lowcut), float(highcut), b, a)
    return bool(ok), np.array(b[:], np.float64), np.array(a[:], np.float64)


def butter_bandpass_filter(data: np.ndarray, b, a) -> np.ndarray:
    """Direct form II from zero state along the last axis; float32 input runs the fp32
    firmware arithmetic (classifier.cpp:193-219), float64 the fp64 one (classifier.c:420-446)."""
    data = np.asarray(data)
    dt = np.float64 if data.dtype == np.float64 else np.float32
    x = np.ascontiguousarray(np.atleast_2d(data), dt)
    y = np.empty_like(x)
    bb = np.ascontiguousarray(b, dt)
    aa = np.ascontiguousarray(a, dt)
    fn = _lib.load().dsp_butter_bandpass_filter_f64 if dt == np.float64 else _lib.load().dsp_butter_bandpass_filter_f32
    _lib.check(fn(x.ctypes.data, x.shape[0], x.shape[1], x.shape[1], bb.ctypes.data, aa.ctypes.data, y.ctypes.data),
               "dsp_butter_bandpass_filter")
    return y.reshape(data.shape)


def compute_spectrogram(signal: np.ndarray, fs: int = 16000):
    """-> (frequencies[129], times[T], Sxx[129][T]).  float32 input: the fp32 firmware arithmetic, bit-identical to
    classifier.cpp:221-368; float64 input: the float64 pipeline of donut-classifier/classifier.c:448-592."""
    signal = np.asarray(signal)
    dt = np.float64 if signal.dtype == np.float64 else np.float32
    signal = np.ascontiguousarray(signal, dt)
    t_max = max(1, (signal.size - 256) // 224 + 1) if signal.size >= 256 else 1
    freqs = np.empty(129, dt)
    times = np.empty(t_max, dt)
    sxx = np.empty((129, t_max), dt)
    fn = _lib.load().dsp_compute_spectrogram_f64 if dt == np.float64 else _lib.load().dsp_compute_spectrogram_f32
    t = _lib.check(fn(signal.ctypes.data, signal.size, int(fs), freqs.ctypes.data, times.ctypes.data, sxx.ctypes.data),
                   "dsp_compute_spectrogram")
    return freqs, times[:t], sxx.reshape(-1)[: 129 * t].reshape(129, t)


def classify(data: np.ndarray) -> int:
    """`int classify(float *data, int data_size)` (sync/lib/classifier.h:19) through the C ABI."""
    data = np.ascontiguousarray(data, np.float32)
    return int(_lib.load().dsp_classify(data.ctypes.data, data.size))


# thresholds of the reference's variants: (keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min)
CLASSIFY_SYNC_LIB = (0.65, 0.80, 70.0, 100.0, 200.0, 80.0)        # sync/lib/classifier.cpp:67-68, :436, :109 (default)
CLASSIFY_MICROPHONE = (0.70, 0.85, 45.0, 100.0, 200.0, 150.0)     # microphone/src/classifier.cpp:79-80, :448, :123
CLASSIFY_MICROPHONE_C = (0.70, 0.85, 45.0, 50.0, 200.0, 200.0)    # microphone/src/classifier.c:120-121, :608, :164 (thresholds of the float64 file)
CLASSIFY_DONUT_C = (0.70, 0.85, 45.0, 75.0, 300.0, 100.0)         # donut-classifier/classifier.c:141-142, :660, :184 (thresholds of the float64 file)


def classify_config(values=None) -> "_lib.ClassifyConfig":
    """dsp_classify_config: the library's defaults (sync/lib thresholds) or a 6-tuple like CLASSIFY_MICROPHONE."""
    c = _lib.ClassifyConfig()
    _lib.load().dsp_classify_default_config(C.byref(c))
    if values is not None:
        c.keep_lo, c.keep_hi, c.midpoint_db, c.middle_max, c.above_min, c.below_min = (float(v) for v in values)
    return c


def classify_batch(clips: np.ndarray, with_trace: bool = False, config=None):
    """clips [n_clips][n] float32 (host) -> labels int32 [n_clips] (+ per-clip midpoints / band sums).
    config: None (sync/lib thresholds) or a 6-tuple / ClassifyConfig (dsp_classify_batch_host_cfg)."""
    clips = np.ascontiguousarray(np.atleast_2d(clips), np.float32)
    n_clips, n = clips.shape
    labels = np.zeros(n_clips, np.int32)
    tr = (_lib.ClassifyTrace * n_clips)() if with_trace else None
    cfg = None if config is None else (config if isinstance(config, _lib.ClassifyConfig) else classify_config(config))
    _lib.check(_lib.load().dsp_classify_batch_host_cfg(C.byref(cfg) if cfg is not None else None, clips.ctypes.data, n_clips, n, n,
                                                        labels.ctypes.data, C.byref(tr) if with_trace else None), "dsp_classify_batch_host_cfg")
    if not with_trace:
        return labels
    out = []
    for t in tr:
        k = t.n_midpoints
        out.append((np.array(t.midpoints[:k], np.float32),
                    np.array([[t.sums[i][j] for j in range(3)] for i in range(k)], np.float32).reshape(-1, 3)))
    return labels, out


def _pcm_shape(pcm):
    if pcm.ndim == 1:
        pcm = pcm[None, :]
    if pcm.ndim not in (2, 3) or (pcm.ndim == 3 and pcm.shape[2] != 2):
        raise ValueError("pcm must be int16 [n_clips][n] or [n_clips][n][2]")
    return pcm, (2 if pcm.ndim == 3 else 1)


def classify_batch_pcm16(pcm: np.ndarray, stereo_mode: int = 0, with_trace: bool = False, config=None):
    """dsp_classify_batch_pcm16_host: the float32 classify() on int16 PCM [n_clips][n] (mono) or [n_clips][n][2] (interleaved stereo:
    channel 0 or the channels' average), converted in the kernels' loads as sync/sync.cpp:237-242 does (pcmSample / 32768.0)."""
    pcm, channels = _pcm_shape(np.ascontiguousarray(pcm, np.int16))
    n_clips, n = pcm.shape[:2]
    labels = np.zeros(n_clips, np.int32)
    tr = (_lib.ClassifyTrace * n_clips)() if with_trace else None
    cfg = None if config is None else (config if isinstance(config, _lib.ClassifyConfig) else classify_config(config))
    _lib.check(_lib.load().dsp_classify_batch_pcm16_host(C.byref(cfg) if cfg is not None else None, pcm.ctypes.data, n_clips, n, n, channels, int(stereo_mode),
                                                          labels.ctypes.data, C.byref(tr) if with_trace else None), "dsp_classify_batch_pcm16_host")
    if not with_trace:
        return labels
    out = []
    for t in tr:
        k = t.n_midpoints
        out.append((np.array(t.midpoints[:k], np.float32),
                    np.array([[t.sums[i][j] for j in range(3)] for i in range(k)], np.float32).reshape(-1, 3)))
    return labels, out


def classify_device_pcm16(pcm, labels=None, stereo_mode: int = 0, config=None):
    """pcm: cuda int16 [n_clips][n] or [n_clips][n][2] -> cuda int32 labels (dsp_classify_batch_pcm16_device), stream-ordered."""
    import torch
    if not (pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() in (2, 3) and pcm.stride(-1) == 1):
        raise ValueError("pcm must be an int16 CUDA tensor [n_clips][n] or [n_clips][n][2] with unit inner stride")
    channels = 2 if pcm.dim() == 3 else 1
    if channels == 2 and (pcm.shape[2] != 2 or pcm.stride(1) != 2 or pcm.stride(0) % 2):
        raise ValueError("stereo pcm must be interleaved [n_clips][n][2]")
    n_clips, n = pcm.shape[:2]
    if labels is None:
        labels = torch.empty(n_clips, dtype=torch.int32, device=pcm.device)
    st = C.c_void_p(torch.cuda.current_stream(pcm.device).cuda_stream)
    cfg = None if config is None else (config if isinstance(config, _lib.ClassifyConfig) else classify_config(config))
    _lib.check(_lib.load().dsp_classify_batch_pcm16_device(C.byref(cfg) if cfg is not None else None, pcm.data_ptr(), n_clips, n, pcm.stride(0) // channels,
                                                            channels, int(stereo_mode), labels.data_ptr(), st), "dsp_classify_batch_pcm16_device")
    return labels


def _traces(tr):
    out = []
    for t in tr:
        k = t.n_midpoints
        out.append((np.array(t.midpoints[:k], np.float32),
                    np.array([[t.sums[i][j] for j in range(3)] for i in range(k)], np.float32).reshape(-1, 3)))
    return out


def classify_ragged(signal: np.ndarray, offsets, stereo_mode: int = 0, with_trace: bool = False, config=None):
    """Clips of different lengths in ONE call (dsp_classify_batch_ragged_host / _pcm16_host): `signal` is a flat host buffer -- float32
    [total], int16 [total] or interleaved stereo int16 [total][2] -- and clip c is samples [offsets[c], offsets[c + 1]).  Labels (and
    traces) as one classify() per clip gives them."""
    signal = np.ascontiguousarray(signal)
    off, n_clips = _lib.c_offsets(offsets)
    assert int(offsets[-1]) <= signal.shape[0]
    labels = np.zeros(n_clips, np.int32)
    tr = (_lib.ClassifyTrace * max(n_clips, 1))() if with_trace else None
    cfg = None if config is None else (config if isinstance(config, _lib.ClassifyConfig) else classify_config(config))
    cp = C.byref(cfg) if cfg is not None else None
    if signal.dtype == np.int16:
        assert signal.ndim in (1, 2)
        _lib.check(_lib.load().dsp_classify_batch_ragged_pcm16_host(cp, signal.ctypes.data, n_clips, off, signal.ndim, int(stereo_mode), labels.ctypes.data,
                                                                     C.byref(tr) if with_trace else None), "dsp_classify_batch_ragged_pcm16_host")
    else:
        signal = np.ascontiguousarray(signal, np.float32)
        assert signal.ndim == 1
        _lib.check(_lib.load().dsp_classify_batch_ragged_host(cp, signal.ctypes.data, n_clips, off, labels.ctypes.data, C.byref(tr) if with_trace else None),
                   "dsp_classify_batch_ragged_host")
    return (labels, _traces(tr)[:n_clips]) if with_trace else labels


def classify_device_ragged(signal, offsets, labels=None, stereo_mode: int = 0, config=None):
    """The same on a flat cuda buffer (float32 [total], int16 [total] or [total][2]) -> cuda int32 labels, stream-ordered."""
    import torch
    off, n_clips = offsets if isinstance(offsets, tuple) else _lib.c_offsets(offsets)      # (a prepared (ctypes array, n_clips) pair: no conversion per call)
    assert signal.is_cuda and signal.stride(-1) == 1 and int(off[n_clips]) <= signal.shape[0]
    if labels is None:
        labels = torch.empty(n_clips, dtype=torch.int32, device=signal.device)
    st = C.c_void_p(torch.cuda.current_stream(signal.device).cuda_stream)
    cfg = None if config is None else (config if isinstance(config, _lib.ClassifyConfig) else classify_config(config))
    cp = C.byref(cfg) if cfg is not None else None
    if signal.dtype == torch.int16:
        assert signal.dim() in (1, 2)
        _lib.check(_lib.load().dsp_classify_batch_ragged_pcm16_device(cp, signal.data_ptr(), n_clips, off, signal.dim(), int(stereo_mode), labels.data_ptr(), st),
                   "dsp_classify_batch_ragged_pcm16_device")
    else:
        assert signal.dtype == torch.float32 and signal.dim() == 1
        _lib.check(_lib.load().dsp_classify_batch_ragged_device(cp, signal.data_ptr(), n_clips, off, labels.data_ptr(), st), "dsp_classify_batch_ragged_device")
    return labels


def classify_release(device: int = -1) -> None:
    _lib.check(_lib.load().dsp_classify_release(int(device)), "dsp_classify_release")


def classify_batch_f64(clips: np.ndarray, with_trace: bool = False, config=None):
    """The float64 classifier of donut-classifier/classifier.c (:83-192) on the GPU: clips [n_clips][n] float64 (host) -> labels
    int32 (+ per-clip float64 midpoints / band sums).  config: None (the file's thresholds 0.70 / 0.85, 45 dB, 75 / 300 / 100) or a
    6-tuple (keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min) of doubles."""
    clips = np.ascontiguousarray(np.atleast_2d(clips), np.float64)
    n_clips, n = clips.shape
    labels = np.zeros(n_clips, np.int32)
    tr = (_lib.ClassifyTraceF64 * n_clips)() if with_trace else None
    cfg = None if config is None else _lib.ClassifyConfigF64(*[float(v) for v in config])
    _lib.check(_lib.load().dsp_classify_batch_host_f64(C.byref(cfg) if cfg is not None else None, clips.ctypes.data, n_clips, n, n,
                                                        labels.ctypes.data, C.byref(tr) if with_trace else None), "dsp_classify_batch_host_f64")
    return (labels, _trace_f64(tr)) if with_trace else labels


def classify_device_f64(clips, labels=None, config=None):
    """clips: cuda float64 [n_clips][n] -> cuda int32 labels (dsp_classify_batch_device_f64); runs on torch's current stream."""
    import torch
    if not (clips.is_cuda and clips.dtype == torch.float64 and clips.dim() == 2 and clips.stride(1) == 1):
        raise ValueError("clips must be a float64 CUDA tensor [n_clips][n] with unit inner stride")
    n_clips, n = clips.shape
    if labels is None:
        labels = torch.empty(n_clips, dtype=torch.int32, device=clips.device)
    st = C.c_void_p(torch.cuda.current_stream(clips.device).cuda_stream)
    cfg = None if config is None else _lib.ClassifyConfigF64(*[float(v) for v in config])
    _lib.check(_lib.load().dsp_classify_batch_device_f64(C.byref(cfg) if cfg is not None else None, clips.data_ptr(), n_clips, n,
                                                          clips.stride(0), labels.data_ptr(), None, st), "dsp_classify_batch_device_f64")
    return labels


STEREO_CHANNEL0, STEREO_AVERAGE = 0, 1       # dsp_amd.h: how interleaved stereo PCM becomes mono


def _trace_f64(tr):
    out = []
    for t in tr:
        k = t.n_midpoints
        out.append((np.array(t.midpoints[:k], np.float64),
                    np.array([[t.sums[i][j] for j in range(3)] for i in range(k)], np.float64).reshape(-1, 3)))
    return out


def classify_batch_f64_pcm16(pcm: np.ndarray, stereo_mode: int = STEREO_CHANNEL0, with_trace: bool = False, config=None):
    """dsp_classify_batch_pcm16_host_f64: pcm int16 [n_clips][n] (mono) or [n_clips][n][2] (interleaved stereo; channel 0 as
    donut-classifier/classifier.c:286-297 or the channels' average) -> labels (+ midpoints / band sums), the samples converted in the
    kernels' loads exactly like classifier.c:55-59 (s / 32768.0)."""
    pcm = np.ascontiguousarray(pcm, np.int16)
    if pcm.ndim == 1:
        pcm = pcm[None, :]
    channels = 2 if pcm.ndim == 3 else 1
    if pcm.ndim not in (2, 3) or (pcm.ndim == 3 and pcm.shape[2] != 2):
        raise ValueError("pcm must be int16 [n_clips][n] or [n_clips][n][2]")
    n_clips, n = pcm.shape[:2]
    labels = np.zeros(n_clips, np.int32)
    tr = (_lib.ClassifyTraceF64 * n_clips)() if with_trace else None
    cfg = None if config is None else _lib.ClassifyConfigF64(*[float(v) for v in config])
    _lib.check(_lib.load().dsp_classify_batch_pcm16_host_f64(C.byref(cfg) if cfg is not None else None, pcm.ctypes.data, n_clips, n, n, channels,
                                                              int(stereo_mode), labels.ctypes.data, C.byref(tr) if with_trace else None),
               "dsp_classify_batch_pcm16_host_f64")
    return (labels, _trace_f64(tr)) if with_trace else labels


def classify_device_f64_pcm16(pcm, labels=None, stereo_mode: int = STEREO_CHANNEL0, config=None):
    """pcm: cuda int16 [n_clips][n] or [n_clips][n][2] -> cuda int32 labels (dsp_classify_batch_pcm16_device_f64), stream-ordered on
    torch's current stream."""
    import torch
    if not (pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() in (2, 3) and pcm.stride(-1) == 1):
        raise ValueError("pcm must be an int16 CUDA tensor [n_clips][n] or [n_clips][n][2] with unit inner stride")
    channels = 2 if pcm.dim() == 3 else 1
    if channels == 2 and (pcm.shape[2] != 2 or pcm.stride(1) != 2 or pcm.stride(0) % 2):
        raise ValueError("stereo pcm must be interleaved [n_clips][n][2]")
    n_clips, n = pcm.shape[:2]
    if labels is None:
        labels = torch.empty(n_clips, dtype=torch.int32, device=pcm.device)
    st = C.c_void_p(torch.cuda.current_stream(pcm.device).cuda_stream)
    cfg = None if config is None else _lib.ClassifyConfigF64(*[float(v) for v in config])
    _lib.check(_lib.load().dsp_classify_batch_pcm16_device_f64(C.byref(cfg) if cfg is not None else None, pcm.data_ptr(), n_clips, n,
                                                                pcm.stride(0) // channels, channels, int(stereo_mode), labels.data_ptr(), None, st),
               "dsp_classify_batch_pcm16_device_f64")
    return labels


def classify_ragged_f64(signal: np.ndarray, offsets, stereo_mode: int = STEREO_CHANNEL0, with_trace: bool = False, config=None):
    """The float64 classifier on clips of different lengths in ONE call (dsp_classify_batch_ragged_host_f64 / _pcm16_host_f64): `signal`
    is a flat host buffer -- float64 [total], int16 [total] or interleaved stereo int16 [total][2] -- and clip c is samples
    [offsets[c], offsets[c + 1])."""
    signal = np.ascontiguousarray(signal)
    off, n_clips = _lib.c_offsets(offsets)
    assert int(offsets[-1]) <= signal.shape[0]
    labels = np.zeros(n_clips, np.int32)
    tr = (_lib.ClassifyTraceF64 * max(n_clips, 1))() if with_trace else None
    cfg = None if config is None else _lib.ClassifyConfigF64(*[float(v) for v in config])
    cp = C.byref(cfg) if cfg is not None else None
    if signal.dtype == np.int16:
        assert signal.ndim in (1, 2)
        _lib.check(_lib.load().dsp_classify_batch_ragged_pcm16_host_f64(cp, signal.ctypes.data, n_clips, off, signal.ndim, int(stereo_mode), labels.ctypes.data,
                                                                         C.byref(tr) if with_trace else None), "dsp_classify_batch_ragged_pcm16_host_f64")
    else:
        signal = np.ascontiguousarray(signal, np.float64)
        assert signal.ndim == 1
        _lib.check(_lib.load().dsp_classify_batch_ragged_host_f64(cp, signal.ctypes.data, n_clips, off, labels.ctypes.data, C.byref(tr) if with_trace else None),
                   "dsp_classify_batch_ragged_host_f64")
    return (labels, _trace_f64(tr)[:n_clips]) if with_trace else labels


def classify_device_ragged_f64(signal, offsets, labels=None, stereo_mode: int = STEREO_CHANNEL0, config=None):
    """The same on a flat cuda buffer (float64 [total], int16 [total] or [total][2]) -> cuda int32 labels, stream-ordered."""
    import torch
    off, n_clips = offsets if isinstance(offsets, tuple) else _lib.c_offsets(offsets)      # (a prepared (ctypes array, n_clips) pair: no conversion per call)
    assert signal.is_cuda and signal.stride(-1) == 1 and int(off[n_clips]) <= signal.shape[0]
    if labels is None:
        labels = torch.empty(n_clips, dtype=torch.int32, device=signal.device)
    st = C.c_void_p(torch.cuda.current_stream(signal.device).cuda_stream)
    cfg = None if config is None else _lib.ClassifyConfigF64(*[float(v) for v in config])
    cp = C.byref(cfg) if cfg is not None else None
    if signal.dtype == torch.int16:
        assert signal.dim() in (1, 2)
        _lib.check(_lib.load().dsp_classify_batch_ragged_pcm16_device_f64(cp, signal.data_ptr(), n_clips, off, signal.dim(), int(stereo_mode), labels.data_ptr(), None, st),
                   "dsp_classify_batch_ragged_pcm16_device_f64")
    else:
        assert signal.dtype == torch.float64 and signal.dim() == 1
        _lib.check(_lib.load().dsp_classify_batch_ragged_device_f64(cp, signal.data_ptr(), n_clips, off, labels.data_ptr(), None, st), "dsp_classify_batch_ragged_device_f64")
    return labels


def classify_stats_f64(device: int = 0):
    """-> (segments, undecided, listed_clips) of the last float64 classifier pass on `device` (dsp_classify_stats_f64)."""
    a, b, c = C.c_long(), C.c_long(), C.c_long()
    _lib.check(_lib.load().dsp_classify_stats_f64(int(device), C.byref(a), C.byref(b), C.byref(c)), "dsp_classify_stats_f64")
    return a.value, b.value, c.value


def classify_release_f64(device: int = -1) -> None:
    _lib.check(_lib.load().dsp_classify_release_f64(int(device)), "dsp_classify_release_f64")


def find_midpoints(data: np.ndarray, fs: int = 16000) -> np.ndarray:
    """sync/lib/classifier.h:18: midpoints (seconds) of the loud 1000-3000 Hz stretches of one clip."""
    data = np.ascontiguousarray(data, np.float32).reshape(-1)
    out = np.zeros(64, np.float32)
    n = _lib.check(_lib.load().dsp_find_midpoints(data.ctypes.data, data.size, int(fs), out.ctypes.data, out.size), "dsp_find_midpoints")
    return out[:n].copy()


def classify_device(clips, labels=None, config=None):
    """clips: cuda float32 [n_clips][n] -> cuda int32 labels; runs on torch's current stream."""
    import torch
    if not (clips.is_cuda and clips.dtype == torch.float32 and clips.dim() == 2 and clips.stride(1) == 1):
        raise ValueError("clips must be a float32 CUDA tensor [n_clips][n] with unit inner stride")
    n_clips, n = clips.shape
    if labels is None:
        labels = torch.empty(n_clips, dtype=torch.int32, device=clips.device)
    st = C.c_void_p(torch.cuda.current_stream(clips.device).cuda_stream)
    cfg = None if config is None else (config if isinstance(config, _lib.ClassifyConfig) else classify_config(config))
    _lib.check(_lib.load().dsp_classify_batch_device_cfg(C.byref(cfg) if cfg is not None else None, clips.data_ptr(), n_clips, n,
                                                          clips.stride(0), labels.data_ptr(), st), "dsp_classify_batch_device_cfg")
    return labels
