// mfcc1024_wave_kernel.hip -- the n_fft = 1024 MFCC chain (BASELINE config 3: 1024-point FFT, up to 128 mel
// bins) built like the 512-point kernel of mfcc_kernels.hip: one 64-lane wavefront owns one frame, every
// butterfly stage runs in registers, constants live in registers, log + DCT run once per 16-frame tile.
//
//   load      8 x global_load_dwordx2 per lane: z[l + 64 a] = x[2n] + i x[2n+1], one frame ahead
//   FFT       1024-point real FFT as a 512-point complex radix-8 DIF, 8 points per lane:
//               stage A  radix-8 over a,        twiddle W512^(l q)
//               exchange 1 (LDS): slot q <-> lane bits 5:3
//               stage B  radix-8 over l_hi,     twiddle W64^(l_lo p)
//               exchange 2 (LDS): slot p <-> lane bits 2:0, readers in natural order (lane = k mod 64)
//               stage C  radix-8 over l_lo  ->  lane l holds Z[l + 64 r], r = 0..7
//             both exchanges XOR-swizzled: ds_write_b64 (16-lane groups) and ds_read_b64 (32-lane groups) conflict-free
//   untangle  lane l pairs Z[l + 64 t] with Z[512 - l - 64 t] (lane 64 - l, slot 7 - t; 8 ds_bpermute), t = 0..3
//   power     513 bins -> LDS
//   mel       sparse: <= 3 chunks of 12 bins per lane (weights in registers), partial sums -> LDS -> 6-way gather, 2 filters per lane
//   tile      E[mel][16 frames] in LDS; every 16 frames per-frame max + log + DCT on v_mfma_f32_16x16x4_f32
//
// The general Stockham kernel (mfcc1024_kernel.hip) stays as the fallback for filterbanks that need more than three
// chunk slots per lane.  Reference chain: 2fa/audio/word/c/mfcc.c:142-221 with the constants as parameters.
#include "diag_guard.hpp"
#include <hip/hip_runtime.h>

#include "mfcc_device.hpp"
#include "tables.hpp"

// Wave priority per phase of a frame (s_setprio, rising as the frame progresses; see mfcc_kernels.hip).  Schemes: 0 = none,
// 1 = stages A, B at 0, C at 1, untangling at 2, mel + tile at 3,  2 = B at 1, C at 2,  3 = A, B at 1, C at 2, untangling at 3.
// The float64 prefilter scan of the PRE variant always runs at 0.  Measured (tools/ab.py, 1 M x 1024 frames, interleaved):
// plain kernel 1.132 -> 1.109 ms with scheme 1 (2, 3: no gain); PRE kernel (config 3) 2.99 -> 2.93 / 2.91 / 2.85 ms with 1 / 2 / 3.
#ifndef DSP_P1024
#define DSP_P1024 1
#endif
#ifndef DSP_P1024_PRE
#define DSP_P1024_PRE 3
#endif

namespace dsp {

namespace {

constexpr int W_ZBUF = 0;                           // 512 x float2 exchange image; later P[0..512]
constexpr int W_PART = 4096;                        // 192 partial sums + zero slot (kGenZeroSlot = 256 is remapped to 192)
constexpr int W_ETILE = W_PART + 208 * 4;           // mel energies of 16 frames: E[mel][frame ^ ((mel >> 1) & 15)]
constexpr int W_WAVE_BYTES = W_ETILE + 128 * 16 * 4;
static_assert(W_WAVE_BYTES % 16 == 0, "keep the carve 16-byte aligned");
constexpr int kWaveSlots = 3;                       // chunk slots per lane this kernel holds in registers (128 HTK filters on 513 bins: 155 chunks)
constexpr int kWaveZero = 192;                      // partial slot that always reads 0
// block-shared LDS copies behind the four per-wave carves (round 3):
//   the MFMA A operand of the tile epilogue (32 k-steps x 64 lanes): a global re-read in the rolled k loop sat latency-exposed
//   in front of every MFMA (the 512-point kernel's larger shapes had the same cure);
//   PRE only: the mel weights (3 slots x 12 taps x 64 lanes) as float4 rows [slot][quad][lane] -- the PRE variant has no registers
//   for them, and 36 dword re-reads per frame from L1 made the kernel's speed hang on how many of them hipcc kept in flight
//   (2.58 .. 3.64 ms per 1 M frames between builds that differed in nothing else).
// DSP_PRE_W3 = 1 (PRE variant, an experiment): THREE waves per SIMD -- <= 168 VGPRs (mel gather slots and window starts packed
// into three registers, the window folded into the prefilter's output gain from a chunk-order table instead of living in 16
// registers: still 14 spilled) and three blocks' LDS per CU (8-frame tiles: the MFMA epilogue then runs half empty, twice as
// often).  Measured 2.55 ms against 2.44 at two waves (profiles/r03_config3_ab.txt): not adopted.
#ifndef DSP_PRE_W3
#define DSP_PRE_W3 0
#endif
constexpr int W_WAVE_BYTES_T8 = W_ETILE + 128 * 8 * 4;
constexpr int wave_bytes(bool t8) { return t8 ? W_WAVE_BYTES_T8 : W_WAVE_BYTES; }
constexpr int b_dcta(bool t8) { return 4 * wave_bytes(t8); }
constexpr int b_melw(bool t8) { return b_dcta(t8) + kGenDctSteps * 64 * 4; }
constexpr int B_BYTES_PLAIN = b_melw(false);
constexpr int B_BYTES_PRE = b_melw(DSP_PRE_W3 != 0) + kWaveSlots * kMelChunk * 64 * 4;

// forward radix-8 butterfly: u[q] = sum_a v[a] W8^(a q)
__device__ __forceinline__ void radix8w(c32 (&v)[8])
{
    c32 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    radix4(e);
    radix4(o);
    constexpr float R2 = 0.70710678118654752f;
    const c32 t1 = {(o[1].x + o[1].y) * R2, (o[1].y - o[1].x) * R2};        // W8^1 = (1 - i)/sqrt2
    const c32 t2 = {o[2].y, -o[2].x};                                        // W8^2 = -i
    const c32 t3 = {(o[3].y - o[3].x) * R2, -(o[3].x + o[3].y) * R2};       // W8^3 = (-1 - i)/sqrt2
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
    v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
}

__device__ __forceinline__ double shfl_up_f64(double v, int byte_addr)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)bits), hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(bits >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// Per-frame Butterworth prefilter (BASELINE config 3) on the wave that owns the frame: lane l holds samples [16 l, 16 l + 16).
// Parallel form + scan over the lanes, float64 (PrefilterScan, tables.hpp); y = the filtered samples rounded to float, as
// the two-pass path stores them.  ~500 v_fma_f64 per lane and frame, no HBM traffic.
__device__ __forceinline__ void prefilter_scan(const float (&x)[kScanChunk], float (&y)[kScanChunk], const PrefilterScan *__restrict__ S, int lane)
{
    double t0[4], t1[4];
    // pass 1: each section over the chunk from zero state -> the chunk's own contribution to the state at its end
#pragma unroll
    for (int sc = 0; sc < 4; ++sc) { t0[sc] = 0.0; t1[sc] = 0.0; }
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) {
        const double xi = (double)x[i];
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) {
            const double w0 = xi - S->a1[sc] * t0[sc] - S->a2[sc] * t1[sc];
            t1[sc] = t0[sc];
            t0[sc] = w0;
        }
    }
    // inclusive scan: after step d lane l holds the contribution of chunks (l - 2^(d+1), l] to the state at the end of chunk l
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        const int from = ((lane - (1 << d)) & 63) << 2;
        const bool on = lane >= (1 << d);
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) {
            if (d >= S->steps[sc]) continue;          // wave-uniform: this section's older chunks are damped below 1e-14
            const double u0 = shfl_up_f64(t0[sc], from), u1 = shfl_up_f64(t1[sc], from);
            const double *m = S->pw[d][sc];
            const double n0 = t0[sc] + m[0] * u0 + m[1] * u1, n1 = t1[sc] + m[2] * u0 + m[3] * u1;
            t0[sc] = on ? n0 : t0[sc];
            t1[sc] = on ? n1 : t1[sc];
        }
    }
    // the state a chunk starts from is the scan value of the lane before it (zero for lane 0)
    {
        const int from = ((lane - 1) & 63) << 2;
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) {
            const double u0 = shfl_up_f64(t0[sc], from), u1 = shfl_up_f64(t1[sc], from);
            t0[sc] = lane ? u0 : 0.0;
            t1[sc] = lane ? u1 : 0.0;
        }
    }
    // pass 2: the chunk again from its true state, with the output taps
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) {
        const double xi = (double)x[i];
        double acc = S->k0 * xi;
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) {
            const double w0 = xi - S->a1[sc] * t0[sc] - S->a2[sc] * t1[sc];
            acc += S->b0[sc] * w0 + S->b1[sc] * t0[sc];
            t1[sc] = t0[sc];
            t0[sc] = w0;
        }
        y[i] = (float)acc;
    }
}

// The same filter as a CASCADE of four second-order sections (PrefilterScan::c_*, tables.hpp), the form this kernel runs since
// round 3.  Section s maps the lane's 16 samples u -> y in place: w[n] = u[n] - a1 w[n-1] - a2 w[n-2], y[n] = w[n] - w[n-2]
// (the literal numerators are g (1 - z^-2)^4).  Lane-parallel like the parallel form: the chunk from zero state, a Kogge-Stone
// scan over the lanes with M^(16 * 2^d), the chunk again from its true state.  T = double for the first two sections (they see
// the unattenuated stop-band energy), float for the last two: no cancellation between sections of a cascade, so float32 there
// costs 1e-5 of the parity gate's 1e-4 on stop-band-only frames (tools/emulate_prefilter_cascade.py) and half the issue cycles
// of the float64 parallel form (552 float64 instructions per lane and frame -> ~210 float64 + ~210 float32).
template <typename T>
__device__ __forceinline__ T shfl_lane(T v, int byte_addr);
template <>
__device__ __forceinline__ double shfl_lane<double>(double v, int byte_addr) { return shfl_up_f64(v, byte_addr); }
template <>
__device__ __forceinline__ float shfl_lane<float>(float v, int byte_addr)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// lane l <- lane l - 1 (lane 0 <- 0) without the LDS crossbar: v_mov_b32_dpp wave_shr:1.  DSP_PRE_DPP1 = 1 uses it for the scan's
// first step and for the final one-lane shift of every section (24 of the 44 ds_bpermute_b32 per frame).
#ifndef DSP_PRE_DPP1
#define DSP_PRE_DPP1 1
#endif
// (bound_ctrl: lane 0, which has no source lane, reads 0 -- no v_mov of an `old` value in front of every move)
__device__ __forceinline__ int dpp_up1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
template <typename T>
__device__ __forceinline__ T shfl_up1(T v);
template <>
__device__ __forceinline__ float shfl_up1<float>(float v) { return __int_as_float(dpp_up1(__float_as_int(v))); }
template <>
__device__ __forceinline__ double shfl_up1<double>(double v)
{
    const long long bits = __double_as_longlong(v);
    const int lo = dpp_up1((int)bits), hi = dpp_up1((int)(bits >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// DPP moves of a float / double: lanes without a source lane read 0 (bound_ctrl)
template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov0(double v)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
constexpr int kDppRowShr = 0x110;
// lane 15 of every row to all lanes of the next row; row 0 reads 0 (row_mask 0xE over an `old` of 0: with all rows enabled, row 0
// passes its own values through, bound_ctrl or not -- tools/micro/dpp_probe.hip)
__device__ __forceinline__ float dpp_bcast15(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xE, 0xF, false)); }
__device__ __forceinline__ double dpp_bcast15(double v)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, 0x142, 0xE, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), 0x142, 0xE, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// fused multiply-add in T (__builtin_fma alone is the float64 builtin: on float operands it converts, runs v_fma_f64 and converts back)
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// STEPS: scan steps of this section, a compile-time count (the two literal filters need 1 / 3 / 3 / 5 and 2 / 3 / 3 / 4: one kernel
// instantiation each; more steps than the poles need multiply by matrices below the tolerance and are harmless): the section is
// straight-line code without a branch, so that it can share a scheduling region with the transform of the previous frame.
template <typename T, int STEPS>
__device__ __forceinline__ void cascade_section(T (&u)[kScanChunk], T a1, T a2, const T (*pw)[4][4], int sc, int lane, const T (&rowm)[4])
{
    // pass 1: the chunk from zero state -> its own contribution to the state (w[n-1], w[n-2]) at its end
    T t0 = 0, t1 = 0;
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) {
        const T w0 = fma_t(-a1, t0, fma_t(-a2, t1, u[i]));      // the older state first: one dependent fma per sample
        t1 = t0;
        t0 = w0;
    }
#if DSP_PRE_ROWSCAN
    // Row form of the inclusive scan (tables.hpp): Kogge-Stone inside each 16-lane row with DPP row_shr moves -- no LDS crossbar,
    // no select (a lane without a source reads 0: n = t + 0 exactly) --, then lane (r, j) adds M^(16 (j + 1)) (rowm, per lane) times
    // lane 15 of row r - 1 (DPP row_bcast:15; row 0 reads 0).  A second such step reaches two rows back (sections of 5 steps).
    {
        constexpr int RS = scan_row_steps(STEPS), RR = scan_row_rounds(STEPS);
        auto step = [&](T u0, T u1, const T *m) {
            t0 = fma_t(m[1], u1, fma_t(m[0], u0, t0));
            t1 = fma_t(m[3], u1, fma_t(m[2], u0, t1));
        };
        if (RS > 0) step(dpp_mov0<kDppRowShr + 1>(t0), dpp_mov0<kDppRowShr + 1>(t1), pw[0][sc]);
        if (RS > 1) step(dpp_mov0<kDppRowShr + 2>(t0), dpp_mov0<kDppRowShr + 2>(t1), pw[1][sc]);
        if (RS > 2) step(dpp_mov0<kDppRowShr + 4>(t0), dpp_mov0<kDppRowShr + 4>(t1), pw[2][sc]);
        if (RS > 3) step(dpp_mov0<kDppRowShr + 8>(t0), dpp_mov0<kDppRowShr + 8>(t1), pw[3][sc]);
        const T w0 = t0, w1 = t1;                    // the lane's own row
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const T c0 = dpp_bcast15(t0), c1 = dpp_bcast15(t1);
            t0 = fma_t(rowm[1], c1, fma_t(rowm[0], c0, w0));
            t1 = fma_t(rowm[3], c1, fma_t(rowm[2], c0, w1));
        }
    }
#else
    // inclusive scan over the lanes
#pragma unroll
    for (int d = 0; d < STEPS; ++d) {
        const int from = ((lane - (1 << d)) & 63) << 2;
        const bool on = lane >= (1 << d);
        const T u0 = (DSP_PRE_DPP1 && d == 0) ? shfl_up1<T>(t0) : shfl_lane<T>(t0, from), u1 = (DSP_PRE_DPP1 && d == 0) ? shfl_up1<T>(t1) : shfl_lane<T>(t1, from);
        const T *m = pw[d][sc];
        const T n0 = fma_t(m[1], u1, fma_t(m[0], u0, t0)), n1 = fma_t(m[3], u1, fma_t(m[2], u0, t1));
        if (DSP_PRE_DPP1 && d == 0) {      // lane 0 received zeros (the DPP move's `old`): n0 = t0 + 0, n1 = t1 + 0 exactly, no select
            t0 = n0; t1 = n1;
        } else {
            t0 = on ? n0 : t0;
            t1 = on ? n1 : t1;
        }
    }
#endif
    // a chunk starts from the scan value of the lane before it (zero for lane 0)
    {
        const int from = ((lane - 1) & 63) << 2;
        const T u0 = DSP_PRE_DPP1 ? shfl_up1<T>(t0) : shfl_lane<T>(t0, from), u1 = DSP_PRE_DPP1 ? shfl_up1<T>(t1) : shfl_lane<T>(t1, from);
        t0 = (DSP_PRE_DPP1 || lane) ? u0 : (T)0;      // DPP: lane 0 already holds the move's `old` = 0
        t1 = (DSP_PRE_DPP1 || lane) ? u1 : (T)0;
    }
    // pass 2: the chunk again from its true state, output y = w[n] - w[n-2]
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) {
        const T w0 = fma_t(-a1, t0, fma_t(-a2, t1, u[i]));
        u[i] = w0 - t1;
        t1 = t0;
        t0 = w0;
    }
}

// timing-only diagnostic build (never shipped; outputs are wrong): 1 = no filter at all
#ifndef DSP_PRE_DIAG
#define DSP_PRE_DIAG 0
#endif
// wc != nullptr: the lane's 16 window values in chunk order (GenTables1024::win_chunk), multiplied into the output here
// the per-lane matrices of the row form's cross-row step (PrefilterScan::c_rowm / c_rowmf at j = lane % 16): 24 VGPRs, loaded once
// DSP_PRE_F64_SECTIONS: cascade sections that run in float64 (2, the default: the two that see the unattenuated stop-band energy;
// 1 is an A/B build: section 1 in float32 as well -- measured faster and closer to the gate, see profiles/r03_config3_ab.txt)
#ifndef DSP_PRE_F64_SECTIONS
#define DSP_PRE_F64_SECTIONS 2
#endif
struct RowMats { double d[2][4]; float f[2][4]; float f1[4]; };
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ void prefilter_cascade(const float (&x)[kScanChunk], float (&y)[kScanChunk], const PrefilterScan *__restrict__ S, int lane,
                                                  const RowMats &rm, const float *__restrict__ wc = nullptr)
{
#if DSP_PRE_DIAG == 1
    for (int i = 0; i < kScanChunk; ++i) y[i] = x[i];
    return;
#endif
    double ud[kScanChunk];
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) ud[i] = (double)x[i];
    cascade_section<double, S0>(ud, S->c_a1[0], S->c_a2[0], S->c_pw, 0, lane, rm.d[0]);
    if (DSP_PRE_F64_SECTIONS >= 2) cascade_section<double, S1>(ud, S->c_a1[1], S->c_a2[1], S->c_pw, 1, lane, rm.d[1]);
    float uf[kScanChunk];
#pragma unroll
    for (int i = 0; i < kScanChunk; ++i) uf[i] = (float)ud[i];
    if (DSP_PRE_F64_SECTIONS < 2) cascade_section<float, S1>(uf, S->c_a1f[1], S->c_a2f[1], S->c_pwf, 1, lane, rm.f1);
    cascade_section<float, S2>(uf, S->c_a1f[2], S->c_a2f[2], S->c_pwf, 2, lane, rm.f[0]);
    cascade_section<float, S3>(uf, S->c_a1f[3], S->c_a2f[3], S->c_pwf, 3, lane, rm.f[1]);
    const float g = (float)S->c_gain;
    if (wc) {
        f4v w4[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) w4[a] = reinterpret_cast<const f4v *>(wc)[a];
#pragma unroll
        for (int i = 0; i < kScanChunk; ++i) y[i] = (uf[i] * g) * w4[i >> 2][i & 3];
    } else {
        (void)g;                                     // the cascade's gain rides in the window the kernel multiplies with anyway (win[] below)
#pragma unroll
        for (int i = 0; i < kScanChunk; ++i) y[i] = uf[i];
    }
}

}  // namespace

// Prefilter form of the PRE variant: 1 = cascade (default where the tables allow it), 0 = the float64 parallel form of round 2
#ifndef DSP_PRE_CASCADE
#define DSP_PRE_CASCADE 1
#endif

// PRE: independent full frames run through the Butterworth prefilter in this kernel (lane-contiguous loads, scan, one LDS
// transpose into the FFT's sample order) instead of a separate pass that writes a filtered copy to HBM.
// Two waves per SIMD for every instantiation, stated: VGPRs + AGPRs share one 512-entry file per lane, and left alone hipcc
// parks spills in AGPRs (a build with 256 VGPRs + 25 AGPRs ran at ONE wave per SIMD: 3.9 ms instead of 2.5 per 1 M config-3
// frames; SQ_WAVES 1024 instead of 2048 was what gave it away).
#define DSP_PRE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PRE && DSP_PRE_W3 ? 3 : 2, PRE && DSP_PRE_W3 ? 3 : 2)))
// PS0..PS3 (PRE): scan steps of the four cascade sections (cascade_section)
// PRE: mel weights in registers (1) like the plain kernel, or re-read per frame from the block-shared LDS copy (0)
#ifndef DSP_PRE_MELW_REGS
#define DSP_PRE_MELW_REGS 1
#endif
template <bool FULL, bool CLIPS, bool PRE = false, int PS0 = 6, int PS1 = 6, int PS2 = 6, int PS3 = 6>
__global__ __launch_bounds__(256) DSP_PRE_WAVES_ATTR void mfcc1024_wave_kernel(const Mfcc512Args args, const GenTables1024 *__restrict__ G,
                                                            const PrefilterScan *__restrict__ S = nullptr)
{
    static_assert(!PRE || (FULL && !CLIPS), "the fused prefilter runs on independent 1024-sample frames");
    constexpr int SCH = PRE ? DSP_P1024_PRE : DSP_P1024;       // wave-priority scheme
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool W3 = PRE && DSP_PRE_W3 != 0;
    constexpr int TF = W3 ? 8 : 16;                   // frames per tile
    char *wl = smem + wib * wave_bytes(W3);
    float2 *zbuf = reinterpret_cast<float2 *>(wl + W_ZBUF);
    float *pbuf = reinterpret_cast<float *>(wl + W_ZBUF);
    float *part = reinterpret_cast<float *>(wl + W_PART);
    float *etile = reinterpret_cast<float *>(wl + W_ETILE);

    float *dcta_lds = reinterpret_cast<float *>(smem + b_dcta(W3));
    f4v *melw_lds = reinterpret_cast<f4v *>(smem + b_melw(W3));
    (void)melw_lds;
    for (int i = threadIdx.x; i < kGenDctSteps * 64; i += 256) dcta_lds[i] = (&G->dct_a[0][0])[i];
    if (PRE)
        for (int i = threadIdx.x; i < kWaveSlots * 3 * 64; i += 256) {      // row (slot c, quad qd), lane l: taps 4 qd .. 4 qd + 3
            const int l = i & 63, qd = (i >> 6) % 3, c = i / 192;
            melw_lds[i] = f4v{G->mel_w[c][4 * qd][l], G->mel_w[c][4 * qd + 1][l], G->mel_w[c][4 * qd + 2][l], G->mel_w[c][4 * qd + 3][l]};
        }
    __syncthreads();

    // ---- per-lane constants ------------------------------------------------------------------------------------
    float win[16];
    if (!W3) {
        // PRE (cascade form): the filter's output gain g = b0 is folded into the window here, once per kernel, instead of 16 multiplies per frame
        const float wg = (PRE && DSP_PRE_CASCADE) ? (float)S->c_gain : 1.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) win[i] = G->win[i][lane] * wg;
    }
    c32 tw1[7], tw2[7], twp[4];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        tw1[q] = {G->tw1[2 * q][lane], G->tw1[2 * q + 1][lane]};
        tw2[q] = {G->tw2[2 * q][lane], G->tw2[2 * q + 1][lane]};
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) twp[t] = {G->twp[2 * t][lane], G->twp[2 * t + 1][lane]};
    // mel weights: in registers (36 VGPRs), except in the PRE variant, whose float64 scan needs the room: there they are
    // re-read per frame (coalesced dwords, L1-resident 9 KB table)
    float melw[kWaveSlots][kMelChunk];
    int mel_k0[kWaveSlots];
#pragma unroll
    for (int c = 0; c < kWaveSlots; ++c) {
        mel_k0[c] = G->mel_k0[c][lane];
        if (!PRE || DSP_PRE_MELW_REGS) {
#pragma unroll
            for (int i = 0; i < kMelChunk; ++i) melw[c][i] = G->mel_w[c][i][lane];
        }
    }
    int gat[kGenMelsPerLane][kGenGather];           // partial slots of filters lane and lane + 64 (slots >= 128 read the zero slot)
    unsigned gatp[kGenMelsPerLane][2] = {};          // W3: the same six slots (< 256 each) in two registers per filter
    static_assert(kGenGather == 6 && kWaveZero < 256, "packed gather slots");
#pragma unroll
    for (int i = 0; i < kGenMelsPerLane; ++i)
#pragma unroll
        for (int g = 0; g < kGenGather; ++g) {
            const int sidx = G->mel_src[i][g][lane];
            const int v = sidx < kWaveSlots * 64 ? sidx : kWaveZero;
            if (W3) gatp[i][g >> 2] |= (unsigned)v << (8 * (g & 3));
            else gat[i][g] = v;
        }
    unsigned k0p = 0;                                  // W3: the three window starts (< 1024 each) in one register
    if (W3) {
#pragma unroll
        for (int c = 0; c < kWaveSlots; ++c) k0p |= (unsigned)mel_k0[c] << (10 * c);
    }
    (void)gatp; (void)k0p;
    const int n_mels = args.n_mels, n_mfcc = args.n_mfcc;
    const int l_hi = lane >> 3, l_lo = lane & 7;
    const int partner = ((64 - lane) & 63) << 2;     // byte index for ds_bpermute
    const bool self_paired = lane == 0;
    if (lane == 0) part[kWaveZero] = 0.0f;
    wave_lds_sync();

    const long wave = (long)blockIdx.x * 4 + wib;
    const long n_waves = (long)gridDim.x * 4;
    const unsigned amin_u = __float_as_uint(args.amin);
    const float neg_top_db = -args.top_db;
    const int frame_len = args.frame_len;
    const long n_frames = args.n_frames;

    auto load_frame8 = [&](long off, c32 (&z)[8]) {
        const float *src = static_cast<const float *>(args.in) + off;
        if (PRE) {                       // lane l: samples [16 l, 16 l + 16), four 16-byte loads (the wave reads 4 KB contiguously)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src + 16 * lane + 4 * a));
                z[2 * a] = {x.x, x.y};
                z[2 * a + 1] = {x.z, x.w};
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int i = 2 * (lane + 64 * a);
            if (FULL || i + 1 < frame_len) {
                const f2v x = CLIPS ? *reinterpret_cast<const f2v *>(src + i) : __builtin_nontemporal_load(reinterpret_cast<const f2v *>(src + i));   // clips re-read samples: cacheable
                z[a] = {x.x, x.y};
            } else if (i < frame_len) {
                z[a] = {src[i], 0.0f};
            } else {
                z[a] = {0.0f, 0.0f};
            }
        }
    };

    // one cursor, one frame of look-ahead (a second one was measured in the PRE variant and changes nothing: 2.58 / 2.60 ms)
    WaveCursor<CLIPS> pre;
    pre.init(wave, n_waves, args.chunk, n_frames, args.frames_per_clip, CLIPS ? args.hop : frame_len, args.clip_stride);
    if (!pre.valid()) return;
    c32 nxt[8];
    long f_next = pre.f;
    load_frame8(pre.off, nxt);
    pre.next();

    // ---- 16-frame tile epilogue --------------------------------------------------------------------------------
    int slot = 0;
    long fb0 = 0, fb1 = 0;
    auto flush = [&](int count) {
        wave_lds_sync();
        const int n = lane & 15, q = lane >> 4;
        unsigned mx = amin_u;
#pragma unroll 4
        for (int s = 0; s < kGenDctSteps; ++s) mx = max(mx, __float_as_uint(etile[TF * (4 * s + q) + ((n ^ (2 * s + (q >> 1))) & (TF - 1))]));
        {
            auto r = __builtin_amdgcn_permlane16_swap(mx, mx, false, false);
            mx = max(r[0], r[1]);
            r = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mx = max(r[0], r[1]);
        }
        const float rinv = __builtin_amdgcn_rcpf(__uint_as_float(mx));
        f4v acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        auto kstep = [&](int s, f4v &acc) {
            const float e_s = etile[TF * (4 * s + q) + ((n ^ (2 * s + (q >> 1))) & (TF - 1))];      // TF = 8: columns 8 .. 15 repeat 0 .. 7 and are not stored
            const float ec = __uint_as_float(max(__float_as_uint(e_s), amin_u));
            float db = 3.01029995663981195f * __builtin_amdgcn_logf(ec * rinv);
            db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f);
            if (4 * s + q >= n_mels) db = 0.0f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dcta_lds[s * 64 + lane], db, acc, 0, 0, 0);
        };
#pragma unroll 1
        for (int s = 0; s < kGenDctSteps; s += 2) { kstep(s, acc0); kstep(s + 1, acc1); }
        const f4v d = acc0 + acc1;
        const long fl = (n < 8 ? fb0 : fb1 - 8) + n;
        const bool ok = n < count && n < TF && fl < n_frames;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            if (ok && c < n_mfcc) args.out[fl * n_mfcc + c] = d[j];
        }
        wave_lds_sync();
    };

    // PRE: an iteration transforms frame k from its filtered samples `ys` (in the lanes' chunk order) and filters frame k + 1 --
    // after the transform (default), or (DSP_PRE_PIPE = 1, an experiment) in front of it in the same branch-free scheduling region,
    // in the hope that the scheduler overlaps the two latency-bound chains.  The last iteration filters a frame nobody uses.
// DSP_PRE_PIPE = 1 was built and measured: hipcc does not interleave the two chains, it only stretches the live ranges (256 VGPRs,
// 16 spilled): 3.52 ms against 2.43 for the plain order (profiles/r03_config3_ab.txt).  The plain order is the default.
#ifndef DSP_PRE_PIPE
#define DSP_PRE_PIPE 0
#endif
    float ys[kScanChunk];
    (void)ys;
    RowMats rowm = {};
    if (PRE && DSP_PRE_ROWSCAN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rowm.d[0][k] = S->c_rowm[0][lane & 15][k]; rowm.d[1][k] = S->c_rowm[1][lane & 15][k];
            rowm.f[0][k] = S->c_rowmf[2][lane & 15][k]; rowm.f[1][k] = S->c_rowmf[3][lane & 15][k];
            rowm.f1[k] = DSP_PRE_F64_SECTIONS < 2 ? S->c_rowmf[1][lane & 15][k] : 0.0f;
        }
    }
// DSP_PRE_EARLY_LOAD = 1 (an experiment): the loads of the frame after next are issued as soon as the filter has taken `nxt`, not
// after the filter: in flight during the filter AND the next transform (sixteen more live VGPRs across the filter, still 238 in all).
// Measured 1 % slower (1.82 vs 1.80 ms, profiles/r03_config3_ab.txt): this kernel does not wait for its loads.
#ifndef DSP_PRE_EARLY_LOAD
#define DSP_PRE_EARLY_LOAD 0
#endif
    auto filter_next = [&](float (&out)[kScanChunk]) {
        float xs[kScanChunk];
#pragma unroll
        for (int a = 0; a < 8; ++a) { xs[2 * a] = nxt[a].x; xs[2 * a + 1] = nxt[a].y; }
        if (DSP_PRE_EARLY_LOAD) {
#pragma unroll
            for (int a = 0; a < kScanChunk; ++a) asm volatile("" : "+v"(xs[a]));       // the copy is real: nxt is free from here
            if (pre.valid()) { f_next = pre.f; load_frame8(pre.off, nxt); pre.next(); } else f_next = -1;
        }
#if DSP_PRE_CASCADE
        prefilter_cascade<PS0, PS1, PS2, PS3>(xs, out, S, lane, rowm, W3 ? &G->win_chunk[lane][0] : nullptr);
#else
        prefilter_scan(xs, out, S, lane);
#endif
    };
    long f_cur = f_next;
    if (PRE) {                      // prologue: frame 0 filtered, frame 1 in flight
        filter_next(ys);
        if (!DSP_PRE_EARLY_LOAD) { if (pre.valid()) { f_next = pre.f; load_frame8(pre.off, nxt); pre.next(); } else f_next = -1; }
    }

    while (true) {
        const long f = PRE ? f_cur : f_next;
        c32 v[8];
        if (SCH) __builtin_amdgcn_s_setprio(0);
        bool more;
        float ys_next[kScanChunk];
        (void)ys_next;
        if (PRE) {
            // filtered samples into the FFT's order: lane l takes the pairs (2 (l + 64 a), 2 (l + 64 a) + 1)
            float *yb = reinterpret_cast<float *>(zbuf);
#pragma unroll
            for (int a = 0; a < 4; ++a) *reinterpret_cast<float4 *>(yb + 16 * lane + 4 * a) = make_float4(ys[4 * a], ys[4 * a + 1], ys[4 * a + 2], ys[4 * a + 3]);
            wave_lds_sync();
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float2 q = zbuf[lane + 64 * a];
                if (W3) v[a] = {q.x, q.y};                      // the window went in with the prefilter's gain
                else v[a] = {q.x * win[2 * a], q.y * win[2 * a + 1]};
            }
            wave_lds_sync();
            // the next frame: its samples are in `nxt` (or nothing is left: then what `nxt` still holds is filtered for nobody);
            // the frame after it starts loading behind the filter's reads of `nxt`
            more = f_next >= 0;
            f_cur = f_next;
            if (DSP_PRE_PIPE) filter_next(ys_next);
            if (DSP_PRE_PIPE) { if (pre.valid()) { f_next = pre.f; load_frame8(pre.off, nxt); pre.next(); } else f_next = -1; }
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a) v[a] = {nxt[a].x * win[2 * a], nxt[a].y * win[2 * a + 1]};
            more = pre.valid();
            if (more) { f_next = pre.f; load_frame8(pre.off, nxt); pre.next(); }
        }

        // ---- stage A: radix-8 over a, twiddle W512^(l q) -----------------------------------------------------------
        if (SCH == 3) __builtin_amdgcn_s_setprio(1);
        radix8w(v);
#pragma unroll
        for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw1[q - 1]);
        // exchange 1: element (q, l_hi, l_lo) at (64 q + 8 l_hi + l_lo) ^ 8 (q & 3): writer lane (l_hi, l_lo) slot q,
        // reader lane (q, l_lo) slot l_hi
#pragma unroll
        for (int q = 0; q < 8; ++q) zbuf[(64 * q + lane) ^ (8 * (q & 3))] = make_float2(v[q].x, v[q].y);
        wave_lds_sync();
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float2 x = zbuf[(64 * l_hi + 8 * h + l_lo) ^ (8 * (l_hi & 3))];
            v[h] = {x.x, x.y};
        }
        wave_lds_sync();
        // ---- stage B: radix-8 over l_hi, twiddle W64^(l_lo p) -----------------------------------------------------
        if (SCH == 2) __builtin_amdgcn_s_setprio(1);
        radix8w(v);
#pragma unroll
        for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], tw2[p - 1]);
        // exchange 2: element (q, p, l_lo) at (64 p + 8 q + l_lo) ^ ((p & 3) << 1 | q >> 2): writer lane (q, l_lo) slot p,
        // reader lane (p, q) slot l_lo -> after stage C lane l holds bins l + 64 r
        // (here this lane's l_hi is its q)
#pragma unroll
        for (int p = 0; p < 8; ++p) zbuf[(64 * p + lane) ^ (((p & 3) << 1) | (l_hi >> 2))] = make_float2(v[p].x, v[p].y);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            // reader lane = 8 p' + q' with p' = l_hi, q' = l_lo; slot m = l_lo of the element
            const float2 x = zbuf[(64 * l_hi + 8 * l_lo + m) ^ (((l_hi & 3) << 1) | (l_lo >> 2))];
            v[m] = {x.x, x.y};
        }
        wave_lds_sync();
        // ---- stage C: radix-8 over l_lo: v[r] = Z[lane + 64 r] / 2 ------------------------------------------------
        if (SCH == 1) __builtin_amdgcn_s_setprio(1);
        if (SCH >= 2) __builtin_amdgcn_s_setprio(2);
        radix8w(v);
        if (SCH == 1) __builtin_amdgcn_s_setprio(2);
        if (SCH == 3) __builtin_amdgcn_s_setprio(3);

        // ---- untangle + power: pairs (k, 512 - k), k = lane + 64 t, t = 0..3 ----------------------------------------
        float P[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            c32 b;
            b.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[7 - t].x)));
            b.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[7 - t].y)));
            // lane 0: Z[512 - 64 t] = Z[64 (8 - t)] is its own slot (8 - t) % 8
            const c32 own = v[(8 - t) & 7];
            b.x = self_paired ? own.x : b.x;
            b.y = self_paired ? own.y : b.y;
            const c32 a = v[t];
            const c32 E = {a.x + b.x, a.y - b.y};
            const c32 O = {a.x - b.x, a.y + b.y};
            const c32 Tw = cmul(O, twp[t]);
            const float xr = E.x + Tw.y, xi = E.y - Tw.x;
            const float mr = E.x - Tw.y, mi = E.y + Tw.x;
            P[2 * t] = xr * xr + xi * xi;
            P[2 * t + 1] = mr * mr + mi * mi;
        }
        const float p256 = 4.0f * (v[4].x * v[4].x + v[4].y * v[4].y);     // lane 0: bin 256 pairs with itself
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pbuf[lane + 64 * t] = P[2 * t];
            pbuf[512 - lane - 64 * t] = P[2 * t + 1];
        }
        if (self_paired) pbuf[256] = p256;
        if (SCH) __builtin_amdgcn_s_setprio(3);
        wave_lds_sync();

        // ---- sparse mel: two chunk slots per lane, weights in registers ---------------------------------------------
        int wl_lane = lane;
        if (PRE) asm volatile("" : "+v"(wl_lane));       // opaque per frame: the weight reads stay in the loop (not hoisted back into registers)
#pragma unroll
        for (int c = 0; c < kWaveSlots; ++c) {
            const float *rd = pbuf + (W3 ? (int)((k0p >> (10 * c)) & 1023u) : mel_k0[c]);
            float acc = 0.0f;
            if (PRE && !DSP_PRE_MELW_REGS) {
#pragma unroll
                for (int qd = 0; qd < 3; ++qd) {
                    const f4v w = melw_lds[(c * 3 + qd) * 64 + wl_lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = fmaf(w[j], rd[4 * qd + j], acc);
                }
            } else {
#pragma unroll
                for (int i = 0; i < kMelChunk; ++i) acc = fmaf(melw[c][i], rd[i], acc);
            }
            part[c * 64 + lane] = acc;
        }
        wave_lds_sync();
        if ((slot & 7) == 0) { if (slot == 0) fb0 = f; else fb1 = f; }
#pragma unroll
        for (int i = 0; i < kGenMelsPerLane; ++i) {
            float s = 0.0f;
#pragma unroll
            for (int g = 0; g < kGenGather; ++g) s += part[W3 ? (int)((gatp[i][g >> 2] >> (8 * (g & 3))) & 255u) : gat[i][g]];
            const int m = lane + 64 * i;
            if (m >= n_mels) s = 0.0f;
            etile[TF * m + ((slot ^ (m >> 1)) & (TF - 1))] = s;
        }
        if (++slot == TF || !more) { flush(slot); slot = 0; }
        else wave_lds_sync();
        if (!more) return;
        if (PRE) {
            if (!DSP_PRE_PIPE) {        // not pipelined: the next frame is filtered here, after this one's transform
                filter_next(ys_next);
                if (!DSP_PRE_EARLY_LOAD) { if (pre.valid()) { f_next = pre.f; load_frame8(pre.off, nxt); pre.next(); } else f_next = -1; }
            }
#pragma unroll
            for (int i = 0; i < kScanChunk; ++i) ys[i] = ys_next[i];
        }
    }
}

hipError_t launch_mfcc1024_wave(const Mfcc512Args &args, const GenTables1024 *tables, int blocks, hipStream_t stream, const PrefilterScan *scan,
                                const int *scan_steps)
{
    const bool full = args.frame_len == 1024, clips = args.frames_per_clip > 0;
    if (args.chunk % 8 != 0) return hipErrorInvalidConfiguration;
    const size_t lds = B_BYTES_PLAIN;
    const dim3 g(blocks), b(256);
    if (scan) {
        if (!full || clips || (reinterpret_cast<uintptr_t>(args.in) & 15)) return hipErrorInvalidConfiguration;
        // the scan-step counts are compile-time: one instantiation per literal filter (donut-classifier/classifier.c:342-401), a
        // generic one (six steps everywhere) for anything else
        const int *st = scan_steps;
        if (st && st[0] <= 1 && st[1] <= 3 && st[2] <= 3 && st[3] <= 5)
            hipLaunchKernelGGL((mfcc1024_wave_kernel<true, false, true, 1, 3, 3, 5>), g, b, (size_t)B_BYTES_PRE, stream, args, tables, scan);
        else if (st && st[0] <= 2 && st[1] <= 3 && st[2] <= 3 && st[3] <= 4)
            hipLaunchKernelGGL((mfcc1024_wave_kernel<true, false, true, 2, 3, 3, 4>), g, b, (size_t)B_BYTES_PRE, stream, args, tables, scan);
        else
            hipLaunchKernelGGL((mfcc1024_wave_kernel<true, false, true>), g, b, (size_t)B_BYTES_PRE, stream, args, tables, scan);
        return hipGetLastError();
    }
    const PrefilterScan *none = nullptr;
    if (full && !clips) hipLaunchKernelGGL((mfcc1024_wave_kernel<true, false>), g, b, lds, stream, args, tables, none);
    else if (full) hipLaunchKernelGGL((mfcc1024_wave_kernel<true, true>), g, b, lds, stream, args, tables, none);
    else if (!clips) hipLaunchKernelGGL((mfcc1024_wave_kernel<false, false>), g, b, lds, stream, args, tables, none);
    else hipLaunchKernelGGL((mfcc1024_wave_kernel<false, true>), g, b, lds, stream, args, tables, none);
    return hipGetLastError();
}

int mfcc1024_wave_blocks_per_cu(bool full, bool prefilter)
{
    int n = 0;
    const size_t lds = B_BYTES_PLAIN;
    if (prefilter) {
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc1024_wave_kernel<true, false, true>, 256, (size_t)B_BYTES_PRE);
        return e == hipSuccess && n > 0 ? n : 1;
    }
    hipError_t e = full ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc1024_wave_kernel<true, false>, 256, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc1024_wave_kernel<false, false>, 256, lds);
    return e == hipSuccess && n > 0 ? n : 2;
}

}  // namespace dsp
