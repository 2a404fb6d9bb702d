#!/usr/bin/env python3
"""Lane-level numpy model of the wave-per-frame 512-point real FFT dataflow used by
dsp_amd/csrc/mfcc_kernels.hip (development aid: validates index maps, swizzles and
the conjugate-pair untangling before they are transcribed to HIP).

64 lanes x 4 complex slots.  Exchange 1 is a register transpose (permlane swaps on
the GPU); exchanges 2 and 3 go through an LDS image with XOR-swizzled addresses.
"""
import numpy as np

L = np.arange(64)


def radix4(s):
    """s: [4][64] complex; forward radix-4 butterfly over axis 0 (W4 = -i)."""
    t0, t1 = s[0] + s[2], s[0] - s[2]
    t2, t3 = s[1] + s[3], (s[1] - s[3]) * (-1j)
    return np.stack([t0 + t2, t1 + t3, t0 - t2, t1 - t3])


def bank_conflicts_b64(addr8, groups):
    """addr8: [64] addresses in 8-byte units; groups: list of lane-index arrays;
    modulus = number of 8-byte bank pairs the group spans."""
    worst = 1
    for lanes, mod in groups:
        a = addr8[lanes] % mod
        worst = max(worst, np.bincount(a, minlength=mod).max())
    return worst


READ_GROUPS = [(np.arange(0, 32), 32), (np.arange(32, 64), 32)]        # ds_read_b64: 2 x 32 lanes, 64 banks
WRITE_GROUPS = [(np.arange(16 * g, 16 * g + 16), 16) for g in range(4)]  # ds_write_b64: 4 x 16 lanes, 32 banks


def wave_fft256(z):
    """z: [256] complex -> slots[t][lane] = Z[lane + 64 t]."""
    W = lambda n, k: np.exp(-2j * np.pi * k / n)
    s = np.stack([z[L + 64 * a] for a in range(4)])                      # load: lane L, slot a
    # stage 1 (digit a), twiddle W256^(L q)
    s = radix4(s) * np.stack([W(256, L * q) for q in range(4)])
    # exchange 1: slot q <-> lane bits 5:4.  new[b][16 beta + r] = old[beta][16 b + r]
    beta, r = L >> 4, L & 15
    s = np.stack([s[beta, 16 * b + r] for b in range(4)])
    # stage 2 (digit b), twiddle W64^(r p)
    s = radix4(s) * np.stack([W(64, r * p) for p in range(4)])
    # exchange 2 through LDS: element (beta, p, r=4c+d) at A2 = 64 p + 16 beta + 4 (c ^ p) + d
    c_w, d = (L >> 2) & 3, L & 3
    lds = np.zeros(256, complex)
    for p in range(4):
        addr = 64 * p + 16 * beta + 4 * (c_w ^ p) + d                    # == (L ^ (4 p)) + 64 p
        assert np.array_equal(addr, (L ^ (4 * p)) + 64 * p)
        assert bank_conflicts_b64(addr, WRITE_GROUPS) == 1
        lds[addr] = s[p]
    p_r = (L >> 2) & 3                                                   # reader lane (beta, p, d)
    out = []
    for c in range(4):
        addr = 64 * p_r + 16 * beta + 4 * (c ^ p_r) + d
        assert bank_conflicts_b64(addr, READ_GROUPS) == 1
        out.append(lds[addr])
    s = np.stack(out)
    # stage 3 (digit c), twiddle W16^(d o)
    s = radix4(s) * np.stack([W(16, d * o) for o in range(4)])
    # exchange 3 through LDS: element (beta, p, o, d) at A3 = 64 beta + 16 o + 4 (d ^ beta) + p
    lds = np.zeros(256, complex)
    for o in range(4):
        addr = 64 * beta + 16 * o + 4 * (d ^ beta) + p_r
        assert bank_conflicts_b64(addr, WRITE_GROUPS) == 1
        lds[addr] = s[o]
    o_r, p3, b3 = L >> 4, (L >> 2) & 3, L & 3                            # reader lane 16 o + 4 p + beta
    out = []
    for dd in range(4):
        addr = 64 * b3 + 16 * o_r + 4 * (dd ^ b3) + p3
        assert bank_conflicts_b64(addr, READ_GROUPS) == 1
        out.append(lds[addr])
    s = np.stack(out)
    # stage 4 (digit d), no twiddle: slot t of lane L is Z[L + 64 t]
    return radix4(s)


def untangle_power(S):
    """S[t][lane] = Z[lane+64t] (already scaled by 1/2 via the window) -> P[0..256]."""
    P = np.zeros(257)
    partner = (64 - L) % 64
    a, c = S[0], S[1]                                                    # Z[l], Z[l+64]
    b = np.where(L == 0, S[0], S[3][partner])                            # Z[256-l]  (Z[256] == Z[0])
    dd = np.where(L == 0, S[3], S[2][partner])                           # Z[192-l]
    for (u, v, k) in ((a, b, L), (c, dd, L + 64)):
        w = np.exp(-2j * np.pi * k / 512)
        E, O = u + np.conj(v), u - np.conj(v)
        T = w * O
        Xk = E - 1j * T
        Xm = np.conj(E) - 1j * np.conj(T)
        P[k] = np.abs(Xk) ** 2
        P[256 - k] = np.abs(Xm) ** 2
    P[128] = np.abs(2 * S[2][0]) ** 2                                    # X[128] = conj(Z[128]) (x2: un-halved)
    return P


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, 512)
    z = x[0::2] + 1j * x[1::2]
    S = wave_fft256(z)
    Z = np.fft.fft(z)
    got = np.concatenate([S[t] for t in range(4)])
    print("fft256 max err", np.abs(got - Z).max())
    P = untangle_power(S * 0.5)
    Pref = np.abs(np.fft.rfft(x)) ** 2
    print("power max rel err", (np.abs(P - Pref) / Pref.max()).max())


# ----------------------------------------------------------------------------
# Register-only variant: exchanges 2 and 3 as DPP moves (no LDS at all in the FFT).
# update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) semantics, gfx9:
# a lane is written when its row / bank is enabled and the source lane exists.

def update_dpp(old, src, ctrl, row_mask=0xF, bank_mask=0xF, bound_ctrl=False):
    out = old.copy()
    for lane in range(64):
        row, within = lane >> 4, lane & 15
        bank = within >> 2
        if not (row_mask >> row) & 1 or not (bank_mask >> bank) & 1:
            continue
        if ctrl < 0x100:                                   # quad_perm
            sel = (ctrl >> (2 * (lane & 3))) & 3
            s = (lane & ~3) | sel
        elif 0x101 <= ctrl <= 0x10F:                       # row_shl:n  dst[i] = src[i+n]
            w = within + (ctrl - 0x100)
            s = None if w > 15 else (row << 4) | w
        elif 0x111 <= ctrl <= 0x11F:                       # row_shr:n  dst[i] = src[i-n]
            w = within - (ctrl - 0x110)
            s = None if w < 0 else (row << 4) | w
        elif 0x121 <= ctrl <= 0x12F:                       # row_ror:n  dst[i] = src[(i-n) mod 16]
            s = (row << 4) | ((within - (ctrl - 0x120)) & 15)
        else:
            raise ValueError(hex(ctrl))
        if s is None:
            if bound_ctrl:
                out[lane] = 0
            continue
        out[lane] = src[s]
    return out


def swap_block(A, B, bit):
    """2x2 block transpose between register pair (A: slot bit 0, B: slot bit 1) and lane bit `bit`."""
    lane_bit = (L >> bit) & 1
    if bit == 3:       # partner lane^8: row_ror:8 either way
        Bn = update_dpp(B, A, 0x128, 0xF, 0x3)             # lanes with bit3 = 0 (banks 0,1)
        An = update_dpp(A, B, 0x128, 0xF, 0xC)             # lanes with bit3 = 1 (banks 2,3)
    elif bit == 2:     # partner lane^4
        Bn = update_dpp(B, A, 0x104, 0xF, 0x5)             # bit2 = 0 (banks 0,2) read lane+4
        An = update_dpp(A, B, 0x114, 0xF, 0xA)             # bit2 = 1 (banks 1,3) read lane-4
    else:              # bits 1, 0: quad_perm + v_cndmask on the lane bit
        ctrl = 0x4E if bit == 1 else 0xB1                  # [2,3,0,1] / [1,0,3,2]
        Ap = update_dpp(A, A, ctrl)
        Bp = update_dpp(B, B, ctrl)
        Bn = np.where(lane_bit == 0, Ap, B)
        An = np.where(lane_bit == 1, Bp, A)
    return An, Bn


def wave_fft256_regs(z):
    """All-register dataflow; returns slots[t][lane] with lane (beta,p,o) holding
    Z[64 t + kappa(lane)], kappa = 16 o + 4 p + beta."""
    W = lambda n, k: np.exp(-2j * np.pi * k / n)
    s = np.stack([z[L + 64 * a] for a in range(4)])
    s = radix4(s) * np.stack([W(256, L * q) for q in range(4)])
    beta, r = L >> 4, L & 15
    s = np.stack([s[beta, 16 * b + r] for b in range(4)])          # exchange 1 (permlane swaps)
    s = radix4(s) * np.stack([W(64, r * p) for p in range(4)])
    s = list(s)
    s[0], s[2] = swap_block(s[0], s[2], 3)                          # exchange 2: slot bit1 <-> lane bit3
    s[1], s[3] = swap_block(s[1], s[3], 3)
    s[0], s[1] = swap_block(s[0], s[1], 2)                          #             slot bit0 <-> lane bit2
    s[2], s[3] = swap_block(s[2], s[3], 2)
    d = L & 3
    s = radix4(np.stack(s)) * np.stack([W(16, d * o) for o in range(4)])
    s = list(s)
    s[0], s[2] = swap_block(s[0], s[2], 1)                          # exchange 3: slot bit1 <-> lane bit1
    s[1], s[3] = swap_block(s[1], s[3], 1)
    s[0], s[1] = swap_block(s[0], s[1], 0)                          #             slot bit0 <-> lane bit0
    s[2], s[3] = swap_block(s[2], s[3], 0)
    return radix4(np.stack(s))


def kappa(lane):
    return 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (lane >> 4)


def untangle_power_regs(S):
    """S[t][lane] = Z[64 t + kappa(lane)] / 2 -> P[0..256]; partner lane holds kappa' = (64 - kappa) % 64."""
    P = np.zeros(257)
    kap = kappa(L)
    inv = np.zeros(64, int); inv[kap] = L
    partner = inv[(64 - kap) % 64]
    a, c = S[0], S[1]
    b = np.where(kap == 0, S[0], S[3][partner])
    dd = np.where(kap == 0, S[3], S[2][partner])
    for (u, v, k) in ((a, b, kap), (c, dd, kap + 64)):
        w = np.exp(-2j * np.pi * k / 512)
        E, O = u + np.conj(v), u - np.conj(v)
        T = w * O
        P[k] = np.abs(E - 1j * T) ** 2
        P[256 - k] = np.abs(np.conj(E) - 1j * np.conj(T)) ** 2
    P[128] = np.abs(2 * S[2][0]) ** 2
    return P, partner


if __name__ == "__main__":
    S2 = wave_fft256_regs(z)
    got = np.zeros(256, complex)
    for t in range(4):
        got[64 * t + kappa(L)] = S2[t]
    print("register-only fft256 max err", np.abs(got - Z).max())
    P2, partner = untangle_power_regs(S2 * 0.5)
    print("register-only power max rel err", (np.abs(P2 - Pref) / Pref.max()).max())
    print("partner lanes:", partner[:8], "...")
