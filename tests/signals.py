"""Seeded, numpy-version-independent test signals shared by the golden generator
and the parity tests (so a fixture's input can be rebuilt bit-for-bit anywhere).
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """Counter-based PRNG: splitmix64 finaliser of (seed * 2^32 + idx)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) << np.uint64(32)) + idx.astype(np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(n: int, seed: int, offset: int = 0) -> np.ndarray:
    """n float32 samples uniform in [-1, 1) from the top 24 bits of splitmix64."""
    z = splitmix64(np.arange(offset, offset + n, dtype=np.uint64), seed)
    u = (z >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (2.0 * u - 1.0).astype(np.float32)


def chirp(n: int, f0: float, f1: float, fs: float = 16000.0, amp: float = 0.5) -> np.ndarray:
    t = np.arange(n, dtype=np.float64) / fs
    k = (f1 - f0) / (n / fs)
    return (amp * np.sin(2 * np.pi * (f0 * t + 0.5 * k * t * t))).astype(np.float32)


def mfcc_cases() -> dict:
    """Clips for compute_mfcc parity (reference framing 400/160)."""
    c = {}
    for s in (0, 1, 2):
        c[f"noise{s}"] = uniform_pm1(16000, s)
    c["chirp"] = chirp(16000, 100.0, 7000.0)
    c["silence"] = np.zeros(16000, np.float32)                  # all-zero frames -> all-zero MFCC
    c["tiny"] = (uniform_pm1(4000, 7) * np.float32(3e-7)).astype(np.float32)  # mel energies straddle amin
    c["dc"] = np.full(2000, 0.25, np.float32)
    imp = np.zeros(1200, np.float32); imp[477] = 1.0
    c["impulse"] = imp
    half = uniform_pm1(8000, 9); half[:4000] = 0.0               # silent frames next to loud ones
    c["half_silent"] = half
    c["len399"] = uniform_pm1(399, 3)                            # too short -> 0 frames
    c["len400"] = uniform_pm1(400, 4)                            # exactly one frame
    c["len559"] = uniform_pm1(559, 5)                            # still one frame
    c["len560"] = uniform_pm1(560, 6)                            # two frames
    c["long"] = uniform_pm1(16000 * 6, 11)                       # 598 frames > max_frames 500
    return c


def tone_burst(n, fs, spans, freqs, amp):
    """Sum of sinusoids `freqs` gated to the [t0, t1) spans (seconds)."""
    t = np.arange(n, dtype=np.float64) / fs
    x = np.zeros(n)
    gate = np.zeros(n)
    for t0, t1 in spans:
        gate[(t >= t0) & (t < t1)] = 1.0
    for f, a in zip(freqs, amp):
        x += a * np.sin(2 * np.pi * f * t)
    return (x * gate).astype(np.float32)


def band_noise(n: int, seed: int, lo: float, hi: float, fs: float = 16000.0) -> np.ndarray:
    """splitmix noise band-limited to [lo, hi] Hz by an rfft mask, peak-normalised (float64)."""
    x = uniform_pm1(n, seed).astype(np.float64)
    spec = np.fft.rfft(x)
    f = np.fft.rfftfreq(n, 1.0 / fs)
    spec[(f < lo) | (f > hi)] = 0
    y = np.fft.irfft(spec, n)
    return y / np.abs(y).max()


def classify_cases() -> dict:
    """1 s / 16 kHz clips for the donut classifier (sync/lib/classifier.cpp)."""
    n, fs = 16000, 16000.0
    c = {}
    c["noise"] = (uniform_pm1(n, 21) * np.float32(0.05)).astype(np.float32)
    # one 0.3 s burst with a 2 kHz carrier: yields a blob cluster -> one midpoint
    c["burst_2k"] = (tone_burst(n, fs, [(0.30, 0.62)], [2000.0], [0.4]) + uniform_pm1(n, 22) * np.float32(1e-3)).astype(np.float32)
    # two bursts with broadband content above and below the 2.5-5 kHz gap
    c["jay_like"] = (tone_burst(n, fs, [(0.15, 0.40), (0.55, 0.85)],
                                [1500.0, 2000.0, 5600.0, 6100.0, 6600.0, 900.0, 1300.0],
                                [0.3, 0.3, 0.25, 0.25, 0.25, 0.2, 0.2])
                     + uniform_pm1(n, 23) * np.float32(2e-3)).astype(np.float32)
    # a 2 kHz call with faint 5-7 kHz and 0.5-2.5 kHz noise skirts: the rule
    # middle<100 && above>200 && below>80 (classifier.cpp:109) fires -> label 1
    t = np.arange(n, dtype=np.float64) / fs
    gate = ((t >= 0.15) & (t < 0.45)).astype(np.float64)
    for tag, a_lo in (("scrub_a", 0.05), ("scrub_b", 0.2)):
        x = gate * (0.4 * np.sin(2 * np.pi * 2000.0 * t) + 0.02 * band_noise(n, 31, 5000, 7000)
                    + a_lo * band_noise(n, 32, 500, 2500)) + 1e-3 * uniform_pm1(n, 33)
        c[tag] = x.astype(np.float32)
    c["silence"] = np.zeros(n, np.float32)
    return c
