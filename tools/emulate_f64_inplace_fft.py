"""Emulates fft_frame_inplace (dsp_amd/csrc/classify_f64_device.hpp): the 128-point complex Stockham FFT of fft_frame (radix 4, 4, 4, 2)
run IN PLACE in one 128-slot buffer -- every stage's lane writes its four outputs to the four physical slots it read, so no second
buffer is needed -- and checks the address maps against numpy.  Run on the CPU; the kernel's formulas are copied from here."""
import numpy as np

N = 128


def PA(a):                      # swizzle of the physical slot (bank spread; tools/f64_inplace_banks.py)
    return a ^ ((5 * ((a >> 5) & 3)) & 31) ^ ((4 * ((a >> 3) & 3)) & 7)


def physZ(m):                   # slot of logical element m after stages 3 / 4
    return (m >> 6) + 2 * ((m >> 4) & 3) + 8 * ((m >> 2) & 3) + 32 * (m & 3)


def fft4(u):
    v0, v1, v2, d = u[0] + u[2], u[0] - u[2], u[1] + u[3], u[1] - u[3]
    v3 = d * (-1j)
    return [v0 + v2, v1 + v3, v0 - v2, v1 - v3]


def run(z):
    buf = np.zeros(max(PA(a) for a in range(N)) + 1, complex)
    W = lambda num, den: np.exp(-2j * np.pi * num / den)
    # stage p = 1: inputs from registers, outputs r -> slot i + 32 r
    for i in range(32):
        out = fft4([z[i + 32 * r] for r in range(4)])
        for r in range(4):
            buf[PA(i + 32 * r)] = out[r]
    # stage p = 4
    for i in range(32):
        ad = [PA((i >> 2) + 8 * r + 32 * (i & 3)) for r in range(4)]
        k = i & 3
        out = fft4([buf[ad[r]] * W(r * k, 16) for r in range(4)])
        for r in range(4):
            buf[ad[r]] = out[r]
    # stage p = 16
    for i in range(32):
        ad = [PA((i >> 4) + 2 * r + 8 * ((i >> 2) & 3) + 32 * (i & 3)) for r in range(4)]
        k = i & 15
        out = fft4([buf[ad[r]] * W(r * k, 64) for r in range(4)])
        for r in range(4):
            buf[ad[r]] = out[r]
    # stage p = 64, radix 2: butterflies b = i and i + 32 on (y[b], y[b + 64])
    for i in range(32):
        ad = [PA((q >> 1) + 2 * (i >> 4) + 4 * (q & 1) + 8 * ((i >> 2) & 3) + 32 * (i & 3)) for q in range(4)]
        u = [buf[a] for a in ad]
        a0, a1 = u[2] * W(i, 128), u[3] * W(i + 32, 128)
        buf[ad[0]], buf[ad[2]] = u[0] + a0, u[0] - a0
        buf[ad[1]], buf[ad[3]] = u[1] + a1, u[1] - a1
    return np.array([buf[PA(physZ(k))] for k in range(N)])


rng = np.random.default_rng(1)
z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
got = run(z)
ref = np.fft.fft(z)
print("max error vs numpy:", np.abs(got - ref).max())
assert np.abs(got - ref).max() < 1e-12
# the slots of stage p = 64 equal physZ of the elements they hold
for i in range(32):
    for q in range(4):
        assert (q >> 1) + 2 * (i >> 4) + 4 * (q & 1) + 8 * ((i >> 2) & 3) + 32 * (i & 3) == physZ(i + 32 * q)
assert sorted(PA(physZ(k)) for k in range(N)) == sorted(PA(a) for a in range(N))
print("address maps consistent; buffer slots:", max(PA(a) for a in range(N)) + 1)
