#!/usr/bin/env python3
"""Development aid: numpy emulation of the per-frame Butterworth prefilter of BASELINE config 3 as the 1024-point kernel can run
it -- a CASCADE of four second-order sections, each a lane-parallel scan (64 lanes x 16 samples: chunk from zero state,
Kogge-Stone over the lanes with M^(16 * 2^d), chunk again from its true state), with a chosen precision per section -- against the
float64 direct form of donut-classifier/classifier.c:420-446 under the MFCC parity gate (1e-4 of the frame's L-inf norm).

What it decided (round 3): the parallel form (round 2) needs float64 throughout because its four terms cancel to -76 dB in
the stop band.  A cascade has no such cancellation; all-float32 still fails stop-band-only frames (1.7e-4), float64 for the first
two sections (poles sorted by radius) and float32 for the last two passes with a 10 x margin (1e-5), and halves the filter's
issue cycles.  Uses the CPU oracle as the reference (a tool, not product code).

    python tools/emulate_prefilter_cascade.py
"""
import os
import sys

import numpy as np
from scipy import signal as ss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # noqa: E402
from tests import signals as S                      # noqa: E402
from tests.conftest import frame_linf_close         # noqa: E402

f32, F = np.float32, np.float64


def fma(a, b, c, dt):
    if dt == F:
        return a * b + c
    return (a.astype(F) * F(b) + c.astype(F)).astype(f32)          # exact product, one rounding: a float32 fma


def build(lo, hi, order, tol32=1e-9, tol64=1e-13, dts=None):
    _, b, a = O.butter_bandpass(lo, hi)
    p = np.roots(a)
    p = p[np.imag(p) > 0]
    p = p[np.argsort(np.abs(p))][list(order)]
    secs = []
    for k, pk in enumerate(p):
        a1, a2 = -2 * pk.real, abs(pk) ** 2
        M = np.array([[-a1, -a2], [1.0, 0.0]])
        tol = tol64 if dts is None or dts[k] == F else tol32
        secs.append(dict(a1=a1, a2=a2, pw=[np.linalg.matrix_power(M, 16 * (1 << d)) for d in range(6)],
                         steps=sum(1 for d in range(6) if abs(pk) ** (16 * (1 << d)) >= tol)))
    return b[0], secs


def run_frame(x, g, secs, dts, form="df2"):
    lanes = np.arange(64)
    u = x.reshape(64, 16).astype(F)
    for s, dt in zip(secs, dts):
        u = u.astype(dt)
        a1, a2 = dt(s["a1"]), dt(s["a2"])
        prev = lambda v: np.where(lanes >= 1, np.roll(v, 1), 0).astype(dt)      # noqa: E731
        if form == "df1":                               # zeros first: v[n] = u[n] - u[n-2]
            ext = np.concatenate([prev(u[:, 14])[:, None], prev(u[:, 15])[:, None], u], axis=1)
            u = (ext[:, 2:] - ext[:, :-2]).astype(dt)
        t0, t1 = np.zeros(64, dt), np.zeros(64, dt)
        for i in range(16):
            w0 = fma(t0, -a1, fma(t1, -a2, u[:, i], dt), dt)
            t1, t0 = t0, w0
        for d in range(s["steps"]):
            sh = 1 << d
            u0 = np.where(lanes >= sh, np.roll(t0, sh), 0).astype(dt)
            u1 = np.where(lanes >= sh, np.roll(t1, sh), 0).astype(dt)
            m = s["pw"][d].astype(dt)
            n0 = fma(u1, m[0, 1], fma(u0, m[0, 0], t0, dt), dt)
            n1 = fma(u1, m[1, 1], fma(u0, m[1, 0], t1, dt), dt)
            t0, t1 = np.where(lanes >= sh, n0, t0), np.where(lanes >= sh, n1, t1)
        t0, t1 = prev(t0), prev(t1)
        out = np.empty_like(u)
        for i in range(16):
            w0 = fma(t0, -a1, fma(t1, -a2, u[:, i], dt), dt)
            out[:, i] = (w0 - t1).astype(dt) if form == "df2" else w0
            t1, t0 = t0, w0
        u = out
    return (u.reshape(-1).astype(F) * g).astype(f32)


def test_frames():
    def lp(order, wn, seed, scale=1.0, kind="low"):
        bb, aa = ss.butter(order, wn, kind)
        return (scale * ss.lfilter(bb, aa, np.random.default_rng(seed).standard_normal(4096))[-1024:]).astype(f32)
    t = np.arange(1024) / 16000.0
    return {"noise": S.uniform_pm1(1024, 1), "tone500": (0.5 * np.sin(2 * np.pi * 500 * t)).astype(f32),
            "tone1500": (0.5 * np.sin(2 * np.pi * 1500 * t)).astype(f32), "tone7900": (0.5 * np.sin(2 * np.pi * 7900 * t)).astype(f32),
            "chirp": S.chirp(1024, 200.0, 7900.0), "impulse@777": np.eye(1, 1024, 777, dtype=f32)[0], "dc": np.full(1024, 0.7, f32),
            "lp0.1": lp(6, 0.1, 0), "lp0.05": lp(8, 0.05, 1), "lp0.2 loud": lp(6, 0.2, 2, 5.0), "lp0.3": lp(8, 0.3, 3),
            "hp0.97": lp(6, 0.97, 4, kind="high"), "step": np.concatenate([np.zeros(500, f32), np.ones(524, f32)])}


def main():
    over = dict(n_fft=1024, frame_length=1024, hop_length=1024, n_mels=128)
    frames = test_frames()
    X = np.stack(list(frames.values()))
    variants = [(form, order, dts) for form in ("df2", "df1") for order in ((0, 1, 2, 3), (3, 2, 1, 0))
                for dts in ((f32,) * 4, (F, f32, f32, f32), (F, F, f32, f32), (f32, f32, F, F))]
    for pre, (lo, hi) in ((2, (3000, 7500)), (1, (1000, 3000))):
        ref = O.mfcc_frames(X, O.default_cfg(prefilter=pre, **over), threads=4)
        cfg0 = O.default_cfg(prefilter=0, **over)
        for form, order, dts in variants:
            g, secs = build(lo, hi, order, dts=dts)
            Y = np.stack([run_frame(x, g, secs, dts, form) for x in X])
            M = O.mfcc_frames(Y, cfg0, threads=4)
            ws = [frame_linf_close(M[i:i + 1], ref[i:i + 1], 1e-4)[1] for i in range(len(X))]
            k = int(np.argmax(ws))
            print(f"prefilter {pre} {form} order {order} precision {''.join('d' if d == F else 's' for d in dts)}: worst {max(ws):.1e} "
                  f"({list(frames)[k]}), median {np.median(ws):.1e}, scan steps {[s['steps'] for s in secs]}")


if __name__ == "__main__":
    main()
