// classify_f64_device.hpp -- device pieces shared by the float64 classifier's kernels (classify_f64_kernels.hip: transform of
// materialised filter outputs, tail kernels; classify_f64_ckpt_kernels.hip: checkpoint pass, screening, recompute + transform).
#pragma once

#include <hip/hip_runtime.h>

#include "classify_kernels.hpp"

namespace dsp {
namespace f64dev {

__device__ __forceinline__ double to_db64(double s) { return 10 * log10(s / 1e-12); }      // classifier.c:113, :688

__device__ __forceinline__ void wave_sync_lds()
{   // a wave's DS instructions complete in order: ordering the compiler is all a write -> read of another lane's data needs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// find_midpoints' "10 log10(s / 1e-12) > midpoint_db" (:688-745).  A cell well above / below the threshold's power mid_power =
// 1e-12 * 10^(midpoint_db / 10) is decided by a comparison; within 1e-9 relative of it (4e-9 dB, against the ~1e-14 dB the
// expression's roundings can move) the reference's expression decides
__device__ __forceinline__ bool is_loud(double s, double mid_power, double midpoint_db)
{
    return s > mid_power * (1.0 + 1e-9) || (s >= mid_power * (1.0 - 1e-9) && s > 0 && to_db64(s) > midpoint_db);
}

// a double from another lane by a DPP move of its two words (VALU, not the LDS crossbar a __shfl_xor takes)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// the sum of v over the 32 lanes of this lane's half-wave, in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
// (after each step the lanes of a group hold the group's sum, so the mirrored lane's value is the other group's), then the
// neighbouring row through v_permlane16_swap
__device__ __forceinline__ double half_wave_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}

struct cd { double re, im; };
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd cmul(cd a, double wr, double wi) { return {a.re * wr - a.im * wi, a.re * wi + a.im * wr}; }

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kPingCd = kSpecSeg / 2 + 32;     // the image stage p = 1 writes is padded by one element in four (P1 below)
constexpr int kPongCd = kSpecSeg / 2;

// block-shared twiddle tables of the transform (filled once per block by fill_twiddles, then a __syncthreads):
//   w16[r - 1][k] = W64^(k r), k < 16 (stage p = 16; stage p = 4 reads its W16^(k r) = W64^(4 k r) at [r - 1][4 k])     w256[k] = W256^k, k < 128
struct FftTwiddles { cd w16[3][16], w256[kSpecSeg / 2]; };

__device__ __forceinline__ void fill_twiddles(FftTwiddles &tw, const SpecTablesD *__restrict__ tab, int tid)
{
    auto w = [&](int m) {                                               // W256^m from the half-circle table
        m &= 255;
        const double sg = (m & 128) ? -1.0 : 1.0;
        return cd{sg * tab->w_re[m & 127], sg * tab->w_im[m & 127]};
    };
    if (tid < 48) tw.w16[tid >> 4][tid & 15] = w(4 * (tid & 15) * ((tid >> 4) + 1));
    if (tid < 128) tw.w256[tid] = w(tid);
}

// what lane i (0 .. 31) of a half-wave keeps in registers across frames: the twiddles of the three butterfly stages (32 VGPRs; the
// bins' W256^k stay in the LDS table), its eight window values, its output slots of stages p = 4 and p = 16
struct FftLane {
    cd t4[3], t16[3], t2[2];
    double win[8];
    int j4, j16;
};

__device__ __forceinline__ void fft_lane_init(FftLane &L, const FftTwiddles &tw, const SpecTablesD *__restrict__ tab, int i)
{
#pragma unroll
    for (int r = 1; r < 4; ++r) { L.t4[r - 1] = tw.w16[r - 1][4 * (i & 3)]; L.t16[r - 1] = tw.w16[r - 1][i & 15]; }
    L.t2[0] = tw.w256[2 * i]; L.t2[1] = tw.w256[2 * i + 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) { L.win[2 * r] = tab->win[2 * i + 64 * r]; L.win[2 * r + 1] = tab->win[2 * i + 64 * r + 1]; }
    L.j4 = ((i - (i & 3)) << 2) + (i & 3);
    L.j16 = ((i - (i & 15)) << 2) + (i & 15);
}

__device__ __forceinline__ void fft4(cd &u0, cd &u1, cd &u2, cd &u3)
{
    const cd v0 = u0 + u2, v1 = u0 - u2, v2 = u1 + u3, d = u1 - u3;
    const cd v3 = {d.im, -d.re};                                     // (u1 - u3) (-i)
    u0 = v0 + v2; u1 = v1 + v3; u2 = v0 - v2; u3 = v1 - v3;
}

// compute_spectrogram (classifier.c:448-592) of ONE frame by the 32 lanes of a half-wave (the other half transforms another frame
// in the same instructions): lane i holds samples 2 i + 64 r, 2 i + 64 r + 1 (r < 4) in x[r].  The mean is a reduction over the 32
// lanes (:551-561), the detrended windowed samples are packed as z[n] = x[2 n] + i x[2 n + 1], so that lane i holds z[i + 32 r].
// 128 = 4 x 4 x 4 x 2: three radix-4 stages (p = 1, 4, 16) and one radix-2 stage (p = 64), ping-pong through b0 (kPingCd) and b1
// (kPongCd).  A stage of radix R with p = the product of the radices before it, thread i of N / R: k = i mod p, inputs x[i + r N / R]
// times exp(-2 pi i r k / (p R)), an R-point DFT, outputs y[(i - k) R + k + r p].  The real spectrum X[k] = E[k] + W256^k O[k] is
// taken from Z[k] and conj(Z[128 - k]): m[r] = |X[k]|^2 of bin k = i + 32 r, doubled for 0 < k < 128 (:574-592), m128 = |X[128]|^2:
// U * PSD -- the division by U is left to whoever needs the cell's value (x / U is monotonic and commutes with the doubling).
// The caller orders its own accesses to b0 / b1 around the call (wave_sync_lds() after it before either is reused).
__device__ __forceinline__ void fft_frame(const d2 (&x)[4], const FftLane &L, const FftTwiddles &tw, cd *__restrict__ b0, cd *__restrict__ b1, int i,
                                          double (&m)[4], double &m128)
{
    const double sum = half_wave_sum(((x[0].x + x[0].y) + (x[1].x + x[1].y)) + ((x[2].x + x[2].y) + (x[3].x + x[3].y)));
    const double mean = sum / (double)kSpecSeg;                      // classifier.c:551-561 detrend
    cd u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = {(x[r].x - mean) * L.win[2 * r], (x[r].y - mean) * L.win[2 * r + 1]};
    // stage p = 1
    fft4(u[0], u[1], u[2], u[3]);
    // (written at P1(n) = n + (n >> 2): lane i's four outputs start 80 bytes after lane i - 1's, so that the eight lanes a
    // ds_write_b128 serves together fall on all 32 banks -- at 64 bytes they shared them four ways: 32 LDS cycles per store
    // instead of 13, and the stores of this stage and the next were half of the kernel's LDS time)
#pragma unroll
    for (int r = 0; r < 4; ++r) b0[5 * i + r] = u[r];
    wave_sync_lds();
    // stage p = 4
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = b0[i + (i >> 2) + 40 * r];     // P1(i + 32 r)
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cmul(u[r], L.t4[r - 1].re, L.t4[r - 1].im);
    fft4(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[L.j4 + 4 * r] = u[r];
    wave_sync_lds();
    // stage p = 16
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = b1[i + 32 * r];
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cmul(u[r], L.t16[r - 1].re, L.t16[r - 1].im);
    fft4(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) b0[L.j16 + 16 * r] = u[r];
    wave_sync_lds();
    // stage p = 64, radix 2: butterflies b = i and i + 32 on (x[b], x[b + 64])
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = b0[i + 32 * r];
    {
        const cd a0 = cmul(u[2], L.t2[0].re, L.t2[0].im), a1 = cmul(u[3], L.t2[1].re, L.t2[1].im);
        b1[i] = u[0] + a0; b1[i + 64] = u[0] - a0;
        b1[i + 32] = u[1] + a1; b1[i + 96] = u[1] - a1;
    }
    wave_sync_lds();
    // Z in natural order in b1.  X[k] = (A + B) / 2 + W256^k (A - B) / (2 i), A = Z[k], B = conj(Z[128 - k])
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 32 * r;
        const cd A = b1[k], Zb = b1[(128 - k) & 127];
        const cd e2 = {A.re + Zb.re, A.im - Zb.im}, d = {A.re - Zb.re, A.im + Zb.im};
        const cd o2 = {d.im, -d.re};
        const cd w = tw.w256[k];
        const cd x2 = e2 + cmul(o2, w.re, w.im);
        const double re = 0.5 * x2.re, im = 0.5 * x2.im;
        m[r] = (re * re + im * im) * ((r == 0 && i == 0) ? 1.0 : 2.0);
    }
    const cd Z0 = b1[0];
    const double r128 = Z0.re - Z0.im;                               // X[128] = E[0] - O[0]
    m128 = r128 * r128;
}

// fft_frame with ONE buffer: every stage's lane writes its four outputs to the four slots it read, so the transform runs in place in a
// 134-slot image (the frame's own LDS row, once its samples are in registers) and needs no second buffer -- which is what lets the
// recompute kernel hold two waves per SIMD.  The butterflies, twiddles and their order are fft_frame's (the data-flow graph of a
// Stockham autosort FFT and of the in-place form are the same; only where an element lives differs), so the results are fft_frame's
// bit for bit.  Slot of logical element m after stage p = 1: (m >> 2) + 32 (m & 3); after p = 4: (m >> 4) + 8 ((m >> 2) & 3) + 32 (m & 3);
// after p = 16 and p = 64: Zslot(m).  PA swizzles a slot's low bits by its higher digits (XOR with 5 x digit 6:5 and 4 x digit 4:3, found by
// the bank model of tools/f64_inplace_banks.py: 240 LDS cycles per transform against 208 conflict-free; a plain shift of two slots per
// 32-slot block: 348 -- measured with it, half of the kernel's LDS cycles were conflicts).  Checked against numpy by
// tools/emulate_f64_inplace_fft.py.
constexpr int kInplaceCd = kSpecSeg / 2;
__device__ __forceinline__ int fft_pa(int a) { return a ^ ((5 * ((a >> 5) & 3)) & 31) ^ ((4 * ((a >> 3) & 3)) & 7); }
__device__ __forceinline__ int fft_zslot(int m) { return (m >> 6) + 2 * ((m >> 4) & 3) + 8 * ((m >> 2) & 3) + 32 * (m & 3); }
__device__ __forceinline__ void fft_frame_inplace(const d2 (&x)[4], const FftLane &L, const FftTwiddles &tw, cd *__restrict__ b, int i,
                                                  double (&m)[4], double &m128)
{
    const double sum = half_wave_sum(((x[0].x + x[0].y) + (x[1].x + x[1].y)) + ((x[2].x + x[2].y) + (x[3].x + x[3].y)));
    const double mean = sum / (double)kSpecSeg;                      // classifier.c:551-561 detrend
    cd u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = {(x[r].x - mean) * L.win[2 * r], (x[r].y - mean) * L.win[2 * r + 1]};
    // stage p = 1
    fft4(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[fft_pa(i + 32 * r)] = u[r];
    wave_sync_lds();
    // stage p = 4
    const int a4 = (i >> 2) + 32 * (i & 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = b[fft_pa(a4 + 8 * r)];
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cmul(u[r], L.t4[r - 1].re, L.t4[r - 1].im);
    fft4(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[fft_pa(a4 + 8 * r)] = u[r];
    wave_sync_lds();
    // stage p = 16
    const int a16 = (i >> 4) + 8 * ((i >> 2) & 3) + 32 * (i & 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = b[fft_pa(a16 + 2 * r)];
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cmul(u[r], L.t16[r - 1].re, L.t16[r - 1].im);
    fft4(u[0], u[1], u[2], u[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[fft_pa(a16 + 2 * r)] = u[r];
    wave_sync_lds();
    // stage p = 64, radix 2: butterflies b = i and i + 32 on (x[b], x[b + 64]); element i + 32 q lives in slot a64 + (q >> 1) + 4 (q & 1)
    const int a64 = 2 * (i >> 4) + 8 * ((i >> 2) & 3) + 32 * (i & 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = b[fft_pa(a64 + (q >> 1) + 4 * (q & 1))];
    {
        const cd a0 = cmul(u[2], L.t2[0].re, L.t2[0].im), a1 = cmul(u[3], L.t2[1].re, L.t2[1].im);
        b[fft_pa(a64)] = u[0] + a0; b[fft_pa(a64 + 1)] = u[0] - a0;              // Z[i], Z[i + 64]
        b[fft_pa(a64 + 4)] = u[1] + a1; b[fft_pa(a64 + 5)] = u[1] - a1;          // Z[i + 32], Z[i + 96]
    }
    wave_sync_lds();
    // X[k] = (A + B) / 2 + W256^k (A - B) / (2 i), A = Z[k], B = conj(Z[128 - k])
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 32 * r;
        const cd A = b[fft_pa(fft_zslot(k))], Zb = b[fft_pa(fft_zslot((128 - k) & 127))];
        const cd e2 = {A.re + Zb.re, A.im - Zb.im}, d = {A.re - Zb.re, A.im + Zb.im};
        const cd o2 = {d.im, -d.re};
        const cd w = tw.w256[k];
        const cd x2 = e2 + cmul(o2, w.re, w.im);
        const double re = 0.5 * x2.re, im = 0.5 * x2.im;
        m[r] = (re * re + im * im) * ((r == 0 && i == 0) ? 1.0 : 2.0);
    }
    const cd Z0 = b[fft_pa(fft_zslot(0))];
    const double r128 = Z0.re - Z0.im;                               // X[128] = E[0] - O[0]
    m128 = r128 * r128;
}

// "one of this frame's 129 cells is above the midpoint threshold", from the lane's m[] / m128 (U * PSD): above / below the band
// around U x threshold: decided; inside it (rare): the reference's expression, evaluated in ONE rolled loop (inlined per cell, the
// float64 log10 cost the kernel 40 VGPRs).  Returns the verdict of this lane's half-wave (the same in its 32 lanes).
__device__ __forceinline__ bool frame_is_loud(const double (&m)[4], double m128, int i, int half, double U, double mid_power, double midpoint_db, double guard)
{
    const double mid_power_u = mid_power * U;
    const double thr_hi = mid_power_u * (1.0 + guard), thr_lo = mid_power_u * (1.0 - guard);
    const double c4 = i == 0 ? m128 : 0.0;
    bool any = m[0] > thr_hi || m[1] > thr_hi || m[2] > thr_hi || m[3] > thr_hi || c4 > thr_hi;
    const bool near = (m[0] >= thr_lo && m[0] <= thr_hi) || (m[1] >= thr_lo && m[1] <= thr_hi) || (m[2] >= thr_lo && m[2] <= thr_hi) ||
                      (m[3] >= thr_lo && m[3] <= thr_hi) || (c4 >= thr_lo && c4 <= thr_hi);
    if (__ballot(near) != 0) {
#pragma unroll 1
        for (int c = 0; c < 5; ++c) {
            const double v = c == 0 ? m[0] : c == 1 ? m[1] : c == 2 ? m[2] : c == 3 ? m[3] : c4;
            if (v >= thr_lo && v <= thr_hi && is_loud(v / U, mid_power, midpoint_db)) any = true;
        }
    }
    const unsigned long long bal = __ballot(any);
    return ((half ? (bal >> 32) : bal) & 0xffffffffull) != 0;
}

}  // namespace f64dev
}  // namespace dsp
