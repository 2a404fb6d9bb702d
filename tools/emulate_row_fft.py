#!/usr/bin/env python3
"""Lane-level numpy model of the ROW-PER-FRAME dataflow (4 frames per wave, 16 lanes x 16
complex per frame) planned for the second MFCC kernel: 256-point complex FFT as 16 x 16 with
ONE exchange through LDS, then the packed-real untangling through a natural-order LDS image.
Checks the index maps and the LDS bank behaviour (ds_write_b64: 4 groups of 16 consecutive
lanes over 32 banks; ds_read_b128: the 4 x 16 lane groups of MI355X_MICROARCH.md over 64 banks).
"""
import numpy as np

LANE = np.arange(64)
G, J = LANE >> 4, LANE & 15

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
B64_WRITE_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
B64_READ_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def worst_conflict(byte_addr, width, groups, n_banks):
    worst = 1
    for grp in groups:
        use = {}
        for l in grp:
            for w in range(width // 4):
                bank = ((byte_addr[l] // 4) + w) % n_banks
                use.setdefault(bank, set()).add(byte_addr[l] // width)
        worst = max(worst, max(len(v) for v in use.values()))
    return worst


def fft16(v):
    """v: [16][lanes] complex -> natural-order 16-point DFT along axis 0, as 4 x 4."""
    W = lambda n, k: np.exp(-2j * np.pi * k / n)
    t = np.zeros((4, 4) + v.shape[1:], complex)          # t[a][b]
    for a in range(4):
        x = [v[a + 4 * k1] for k1 in range(4)]
        for b in range(4):
            t[a][b] = sum(x[k1] * W(4, k1 * b) for k1 in range(4)) * W(16, a * b)
    out = np.zeros_like(v)
    for b in range(4):
        for c in range(4):
            out[b + 4 * c] = sum(t[a][b] * W(4, a * c) for a in range(4))
    return out


TILE = 2048 + 256       # bytes per row tile (exchange tile, later Z image, later P)


def tile_base(g):
    """odd rows start 128 B (32 banks) later: rows 0/1 (and 2/3) share a 32-lane ds_read_b64 pass"""
    return g * TILE + (g & 1) * 128


def xchg_addr(g, q, j):
    """byte address of element (slot q of lane j) in row g's exchange tile"""
    return tile_base(g) + q * 128 + (((j >> 1) ^ (q >> 1)) << 4) + ((j & 1) << 3)


def wave_fft256_rows(z4):
    """z4: [4 frames][256] complex -> regs[p][lane]: lane (g, q') holds Z_g[q' + 16 p]."""
    W = lambda n, k: np.exp(-2j * np.pi * k / n)
    v = np.stack([z4[G, J + 16 * k] for k in range(16)])            # load: reg k = z[j + 16 k]
    v = fft16(v)                                                      # stage 1 over k -> reg q
    v = v * np.stack([W(256, J * q) for q in range(16)])             # twiddle W256^(j q)
    lds = {}
    for q in range(16):                                               # 16 x ds_write_b64
        addr = xchg_addr(G, q, J)
        assert worst_conflict(addr, 8, B64_WRITE_GROUPS, 32) == 1
        for l in range(64):
            lds[addr[l]] = v[q][l]
    out = np.zeros((16, 64), complex)
    for jj in range(8):                                               # 8 x ds_read_b128 (elements 2jj, 2jj+1)
        addr = tile_base(G) + J * 128 + ((jj ^ (J >> 1)) << 4)
        assert worst_conflict(addr, 16, B128_GROUPS, 64) == 1, jj
        for l in range(64):
            out[2 * jj][l] = lds[addr[l]]
            out[2 * jj + 1][l] = lds[addr[l] + 8]
    # check: lane q' now holds B_j[q'] for j = 0..15
    return fft16(out)                                                 # stage 2 over j -> reg p = Z[q' + 16 p]


def untangle_rows(S):
    """S[p][lane (g,q')] = Z_g[q'+16p] / 2 -> P[g][0..256] via a natural-order LDS image."""
    P = np.zeros((4, 257))
    img = np.zeros((4, 257), complex)
    for p in range(16):                                               # ds_write_b64, lanes contiguous
        addr = tile_base(G) + (J + 16 * p) * 8
        assert worst_conflict(addr, 8, B64_WRITE_GROUPS, 32) == 1
        img[G, J + 16 * p] = S[p]
    img[:, 256] = img[:, 0]                                           # Z[256] == Z[0]
    for m in range(8):
        k = J + 16 * m
        addr_u = tile_base(G) + k * 8
        addr_v = tile_base(G) + (256 - k) * 8
        assert worst_conflict(addr_u, 8, B64_READ_GROUPS, 64) == 1
        assert worst_conflict(addr_v, 8, B64_READ_GROUPS, 64) == 1
        u, v = img[G, k], img[G, 256 - k]
        w = np.exp(-2j * np.pi * k / 512)
        E, O = u + np.conj(v), u - np.conj(v)
        T = w * O
        P[G, k] = np.abs(E - 1j * T) ** 2
        P[G, 256 - k] = np.abs(np.conj(E) - 1j * np.conj(T)) ** 2
    P[:, 128] = np.abs(2 * img[:, 128]) ** 2
    return P


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (4, 512))
    z = x[:, 0::2] + 1j * x[:, 1::2]
    S = wave_fft256_rows(z)
    Z = np.fft.fft(z, axis=1)
    got = np.zeros((4, 256), complex)
    for p in range(16):
        got[G, J + 16 * p] = S[p]
    print("row fft256 max err", np.abs(got - Z).max())
    P = untangle_rows(S * 0.5)
    Pref = np.abs(np.fft.rfft(x, axis=1)) ** 2
    print("power max rel err", (np.abs(P - Pref) / Pref.max()).max())
