#!/usr/bin/env python3
"""LDS bank model of mfcc2048_kernel's per-frame accesses (MI355X_MICROARCH.md, LDS: ds_read_b64 = 2 groups of 32 lanes, bank
(a/4) mod 64; ds_write_b64 = 4 groups of 16, bank (a/4) mod 32; ds_read_b32 / ds_write_b32 = 2 x 32, mod 32; a group costs one cycle
per distinct address on its busiest bank).  Prints cycles per access pattern against the conflict-free count."""
import sys


def cost(addrs_bytes, width, write):
    """addrs_bytes: 64 byte addresses (None = lane inactive) -> (cycles, ideal)"""
    if width == 8:
        groups = [range(g * 16, g * 16 + 16) for g in range(4)] if write else [range(0, 32), range(32, 64)]
        nb = 32 if write else 64
    else:
        groups = [range(0, 32), range(32, 64)]
        nb = 32
    total = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs_bytes[l]
            if a is None:
                continue
            for d in range(width // 4):
                banks.setdefault(((a // 4) + d) % nb, set()).add((a // 4) + d)
        total += max((len(v) for v in banks.values()), default=0)
    return total, len(groups)


def main():
    ZI = (lambda i: i + (i >> 4)) if len(sys.argv) < 2 else eval(sys.argv[1])
    zoff = 0
    r16 = lambda p: (p >> 2) + 4 * (p & 3)
    rows = []
    def acc(name, fn, n, width, write, base=0):
        c = i = 0
        for t in n:
            cc, ii = cost([base + width * fn(l, t) if fn(l, t) is not None else None for l in range(64)], width, write)
            c += cc; i += ii
        rows.append((name, c, i))
    acc("pass0 write zbuf[ZI(16 l + r16(p))]", lambda l, p: ZI(16 * l + r16(p)), range(16), 8, True)
    acc("pass1 read  zbuf[ZI(l + 64 t)]", lambda l, t: ZI(l + 64 * t), range(16), 8, False)
    acc("pass1 twiddle w1024[4 t k] (before round 4)", lambda l, t: 4 * t * (l & 15), range(1, 16), 8, False)
    rows.pop()
    acc("pass1 twiddle tw1[16 (t - 1) + k]", lambda l, t: 16 * (t - 1) + (l & 15), range(1, 16), 8, False)
    acc("pass1 write zbuf[ZI(16 (l - k) + k + 16 r16(p))]", lambda l, p: ZI(16 * (l - (l & 15)) + (l & 15) + 16 * r16(p)), range(16), 8, True)
    acc("pass2 read  zbuf[ZI(l + 64 m + 256 t)]", lambda l, mt: ZI(l + 64 * (mt & 3) + 256 * (mt >> 2)), range(16), 8, False)
    acc("pass2 twiddle w1024[t j & 1023]", lambda l, mt: ((1 + mt % 3) * (l + 64 * (mt // 3))) & 1023, range(12), 8, False)
    acc("pass2 write zbuf[ZI(j + 256 q)]", lambda l, mq: ZI(l + 64 * (mq & 3) + 256 * (mq >> 2)), range(16), 8, True)
    acc("untangle read zbuf[ZI(k)]", lambda l, t: ZI(l + 64 * t), range(8), 8, False)
    acc("untangle read zbuf[ZI(1024 - k)]", lambda l, t: ZI(1024 - l - 64 * t), range(8), 8, False)
    acc("untangle read w2048[k]", lambda l, t: l + 64 * t, range(8), 8, False)
    acc("pbuf write [l + 64 t]", lambda l, t: l + 64 * t, range(8), 4, True)
    acc("pbuf write [1024 - l - 64 t]", lambda l, t: 1024 - l - 64 * t, range(8), 4, True)
    tot = sum(r[1] for r in rows); ideal = sum(r[2] for r in rows)
    for name, c, i in rows:
        print(f"{name:55s} {c:5d} cycles (conflict-free {i})")
    print(f"{'total (FFT + untangle + power spectrum)':55s} {tot:5d} cycles (conflict-free {ideal})")


if __name__ == "__main__":
    main()
