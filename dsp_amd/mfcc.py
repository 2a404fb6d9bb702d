"""Host-side mirror of the reference's MFCC interface over the C ABI.

Reference: 2fa/audio/word/c/mfcc.h:16-19 (`compute_mfcc`) and its first caller
2fa/audio/word/c/stop_detector.c:12-21.  Same names, argument meaning and
return conventions; numpy arrays stand in for the caller-owned C buffers and
torch tensors for HBM-resident buffers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .lib import DspError, MfccConfig


def default_config(**over) -> MfccConfig:
    """dsp_mfcc_default_config() with keyword overrides."""
    cfg = MfccConfig()
    _lib.load().dsp_mfcc_default_config(C.byref(cfg))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"dsp_mfcc_config has no field {k!r}")
        setattr(cfg, k, v)
    return cfg


def frames_for(cfg: MfccConfig, num_samples: int, max_frames: int) -> int:
    return _lib.load().dsp_mfcc_frames_for(C.byref(cfg), int(num_samples), int(max_frames))


def compute_mfcc(signal: np.ndarray, max_frames: int) -> np.ndarray:
    """The reference entry point itself: `int compute_mfcc(signal, n, out, max_frames)`
    called through the C ABI with host buffers.  Returns out[:T] (frame-major [T][13])."""
    signal = np.ascontiguousarray(signal, np.float32)
    out = np.zeros((max(int(max_frames), 1), 13), np.float32)
    t = _lib.load().compute_mfcc(signal.ctypes.data, signal.size, out.ctypes.data, int(max_frames))
    return out[:t]


def tables(cfg: MfccConfig):
    """Reference-layout constant tables (window, mel, dct) for `cfg` (host only)."""
    L = _lib.load()
    win = np.empty(cfg.frame_length, np.float32)
    mel = np.empty((cfg.n_mels, cfg.n_fft // 2 + 1), np.float32)
    dct = np.empty((cfg.n_mfcc, cfg.n_mels), np.float32)
    _lib.check(L.dsp_mfcc_tables(C.byref(cfg), win.ctypes.data, mel.ctypes.data, dct.ctypes.data), "dsp_mfcc_tables")
    return win, mel, dct


class MfccPlan:
    """dsp_mfcc_plan: device tables for one configuration on one GPU."""

    def __init__(self, cfg: MfccConfig | None = None, device: int = 0):
        self._L = _lib.load()
        self.cfg = cfg or default_config()
        self.device = device
        h = C.c_void_p()
        _lib.check(self._L.dsp_mfcc_plan_create(C.byref(self.cfg), device, C.byref(h)), "dsp_mfcc_plan_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsp_mfcc_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_kernel(self, kernel: int):
        """0 = one wavefront per frame (default), 1 = one 16-lane row per frame (4 frames per wavefront), 2 = wave per frame with the
        per-frame epilogue, 3 = two frames per wavefront step (reference shape, independent full frames; an experiment)."""
        _lib.check(self._L.dsp_mfcc_plan_set_kernel(self._h, int(kernel)), "dsp_mfcc_plan_set_kernel")

    def set_launch(self, blocks_per_cu: int = 0, frames_per_chunk: int = 0):
        _lib.check(self._L.dsp_mfcc_plan_set_launch(self._h, blocks_per_cu, frames_per_chunk), "dsp_mfcc_plan_set_launch")

    # ---- host buffers -----------------------------------------------------
    def frames_host(self, frames: np.ndarray) -> np.ndarray:
        frames = np.ascontiguousarray(frames, np.float32)
        if frames.ndim != 2 or frames.shape[1] != self.cfg.frame_length:
            raise ValueError(f"frames must be [n][{self.cfg.frame_length}]")
        out = np.empty((frames.shape[0], self.cfg.n_mfcc), np.float32)
        _lib.check(self._L.dsp_mfcc_frames_host(self._h, frames.ctypes.data, frames.shape[0], out.ctypes.data), "dsp_mfcc_frames_host")
        return out

    def clips_host(self, clips: np.ndarray, max_frames: int) -> np.ndarray:
        clips = np.ascontiguousarray(clips, np.float32)
        if clips.ndim == 1:
            clips = clips[None, :]
        n, s = clips.shape
        t = frames_for(self.cfg, s, max_frames)
        out = np.empty((n, t, self.cfg.n_mfcc), np.float32)
        if n == 0 or t == 0:
            return out
        got = _lib.check(self._L.dsp_mfcc_clips_host(self._h, clips.ctypes.data, n, s, s, out.ctypes.data, int(max_frames)), "dsp_mfcc_clips_host")
        assert got == t
        return out

    # ---- HBM-resident buffers (torch tensors on this plan's device) -----------
    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def frames(self, frames, out=None):
        """frames: cuda float32 [n][frame_length] -> cuda float32 [n][n_mfcc]; async on torch's current stream."""
        import torch
        if not (frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous()):
            raise ValueError("frames must be a contiguous float32 CUDA tensor")
        if frames.dim() != 2 or frames.shape[1] != self.cfg.frame_length:
            raise ValueError(f"frames must be [n][{self.cfg.frame_length}]")
        n = frames.shape[0]
        if out is None:
            out = torch.empty((n, self.cfg.n_mfcc), dtype=torch.float32, device=frames.device)
        _lib.check(self._L.dsp_mfcc_frames_device(self._h, frames.data_ptr(), n, out.data_ptr(), self._stream()), "dsp_mfcc_frames_device")
        return out

    def clips_pcm16(self, pcm, max_frames: int, stereo_mode: int = 0, out=None):
        """pcm: cuda int16 [n_clips][samples] (mono) or [n_clips][samples][2] (interleaved stereo;
        stereo_mode 0 = channel 0, 1 = channel average) -> cuda float32 [n_clips][T][n_mfcc]."""
        import torch
        if not (pcm.is_cuda and pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.dim() in (2, 3)):
            raise ValueError("pcm must be a contiguous int16 CUDA tensor [n_clips][samples] or [n_clips][samples][2]")
        channels = 1 if pcm.dim() == 2 else int(pcm.shape[2])
        n, s = int(pcm.shape[0]), int(pcm.shape[1])
        t = frames_for(self.cfg, s, max_frames)
        if out is None:
            out = torch.empty((n, t, self.cfg.n_mfcc), dtype=torch.float32, device=pcm.device)
        if n and t:
            got = _lib.check(self._L.dsp_mfcc_clips_pcm16_device(self._h, pcm.data_ptr(), n, s, s, channels, int(stereo_mode),
                                                                  out.data_ptr(), int(max_frames), self._stream()), "dsp_mfcc_clips_pcm16_device")
            assert got == t
        return out

    def clips(self, clips, max_frames: int, out=None):
        """clips: cuda float32 [n_clips][samples] -> cuda float32 [n_clips][T][n_mfcc]."""
        import torch
        if not (clips.is_cuda and clips.dtype == torch.float32 and clips.dim() == 2 and clips.stride(1) == 1):
            raise ValueError("clips must be a float32 CUDA tensor [n_clips][samples] with unit inner stride")
        n, s = clips.shape
        t = frames_for(self.cfg, s, max_frames)
        if out is None:
            out = torch.empty((n, t, self.cfg.n_mfcc), dtype=torch.float32, device=clips.device)
        if n and t:
            got = _lib.check(self._L.dsp_mfcc_clips_device(self._h, clips.data_ptr(), n, s, clips.stride(0), out.data_ptr(),
                                                            int(max_frames), self._stream()), "dsp_mfcc_clips_device")
            assert got == t
        return out
