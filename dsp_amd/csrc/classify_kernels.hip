// classify_kernels.hip -- gfx950 kernels for the donut classifier path:
//   a9  butter_bandpass_filter   sync/lib/classifier.cpp:193-219 (fp32), donut-classifier/classifier.c:420-446 (fp64)
//   a10 compute_spectrogram      sync/lib/classifier.cpp:221-368 + PlainFFT.cpp:29-94
//   a11 classify / find_midpoints / sum_intense   sync/lib/classifier.cpp:9-136, 370-598
//
// These kernels reproduce the reference's fp32 results BIT FOR BIT, because the label
// hangs on float thresholds: every multiply / add is issued in the reference's order
// with FMA contraction off, the sequential sums stay sequential, and the twiddle /
// window tables come from the reference's own recurrences evaluated on the host.
// Parallelism is across clips and frames (embarrassing), not inside a recurrence.
// Round 1: correct and batched, not yet tuned.
#include "diag_guard.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "classify_kernels.hpp"

#pragma clang fp contract(off)

namespace dsp {

// ---------------------------------------------------------------------------------
// a9: direct form II, one lane per clip, time tiles transposed through LDS so HBM sees
// coalesced 128-byte rows although each lane walks its own clip.
// ---------------------------------------------------------------------------------
typedef float f4nt __attribute__((ext_vector_type(4)));     // nontemporal 16-byte stores
constexpr int IIR_TS = 32;       // samples per tile = one full 128-byte line per float row and tile (16-sample tiles
                                 // fetch every line twice: measured memory-bound on BASELINE config 3)
#ifndef DSP_IIR_BURST
#define DSP_IIR_BURST 16
#endif
constexpr int IIR_BURST = DSP_IIR_BURST;    // samples a lane carries in registers between its LDS reads and writes
constexpr int IIR_LD = IIR_TS + 1;   // +1 word: lane l reads column i of row l -> banks (17 l + i) % 32 distinct

template <typename T>
struct IirState { T d[8]; };

template <typename T, typename C, bool EVEN_B = false>
__device__ __forceinline__ T iir_step(IirState<T> &s, const C &c, T x)
{
    // classifier.cpp:199-216: v = x - sum_{j=1..8} a[j] d[j-1] (left to right), y = b0 v + sum b[j] d[j-1]
    // EVEN_B: the odd numerator taps are exactly 0 and are skipped (see iir2_ckpt_kernel: at most the sign of an exact zero changes)
    T v = x;
#pragma unroll
    for (int j = 1; j <= 8; ++j) v = v - c.a[j] * s.d[j - 1];
    T y = c.b[0] * v;
#pragma unroll
    for (int j = 1; j <= 8; ++j)
        if (!EVEN_B || j % 2 == 0) y = y + c.b[j] * s.d[j - 1];
#pragma unroll
    for (int j = 7; j > 0; --j) s.d[j] = s.d[j - 1];
    s.d[0] = v;
    return y;
}

// T: arithmetic type of the recurrence; TIO: element type in HBM (float rows filtered in double
// are converted on load and rounded once on store: BASELINE config 3's per-frame prefilter).
// TWO: two filters over the same input, one wavefront each (128-thread block, shared input tile): the serial
// recurrences of the two filters run side by side instead of back to back in one lane.
// MEANS (float): also emit the spectrogram's segment means of the outputs (see classify_kernels.hpp).
template <typename T, typename C, bool TWO, typename TIO = T, bool MEANS = false, bool EVEN_B = false>
__global__ __launch_bounds__(TWO ? 128 : 64) void iir_kernel(const TIO *__restrict__ x, long n_clips, int n, long stride, long ystride,
                                                             const C c1, TIO *__restrict__ y1, const C c2, TIO *__restrict__ y2,
                                                             float *__restrict__ means1 = nullptr, float *__restrict__ means2 = nullptr)
{
    constexpr int NTHR = TWO ? 128 : 64;
    __shared__ TIO tin[64 * IIR_LD];
    __shared__ TIO tout[TWO ? 2 : 1][64 * IIR_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = TWO ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;       // which filter this wavefront runs
    const C c = wv ? c2 : c1;
    TIO *__restrict__ y = wv ? y2 : y1;
    float *__restrict__ means = wv ? means2 : means1;
    TIO *to = tout[wv];
    const long clip0 = (long)blockIdx.x * 64;
    const int rows = (int)((n_clips - clip0) < 64 ? (n_clips - clip0) : 64);
    IirState<T> st;
#pragma unroll
    for (int j = 0; j < 8; ++j) st.d[j] = T(0);
    // 16-byte vector path needs every row start and every tile start 16-byte aligned
    const bool vec_ok = (stride * sizeof(TIO)) % 16 == 0 && (ystride * sizeof(TIO)) % 16 == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0 &&
                        (reinterpret_cast<uintptr_t>(y1) % 16) == 0 && (!TWO || (reinterpret_cast<uintptr_t>(y2) % 16) == 0);
    float cur = 0.0f, prev = 0.0f;                    // MEANS: running sums of the current and the previous segment
    const int n_seg = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    (void)cur; (void)prev; (void)n_seg; (void)means;
    // Full tiles travel as 16-byte vectors (4 lanes cover one 64-byte row segment) and are software-pipelined:
    // the next tile's global loads are issued before this tile's recurrence runs, so the serial arithmetic
    // hides the HBM latency even with one wave per SIMD.
    constexpr int PER = 16 / sizeof(TIO), CH = IIR_TS / PER;      // elements per vector, vectors per row
    constexpr int NV = 64 * CH / NTHR;                            // vectors per thread in the tile load
    // (a vector of TIO, its elements taken by constant index: a float4 copied into double[2] made the compiler keep the prefetch
    // in scratch memory, and a load parked in scratch is waited for on the spot -- the float64 kernel ran unpipelined)
    typedef TIO VIO __attribute__((ext_vector_type(PER)));
    VIO pre[NV];
    auto issue = [&](int t0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + NTHR * k, r = e / CH, cc = (e % CH) * PER;
            pre[k] = r < rows ? *reinterpret_cast<const VIO *>(x + (clip0 + r) * stride + t0 + cc) : VIO(TIO(0));
        }
    };
    if (vec_ok && n >= IIR_TS) issue(0);
    for (int t0 = 0; t0 < n; t0 += IIR_TS) {
        const int cols = n - t0 < IIR_TS ? n - t0 : IIR_TS;
        const bool vec = vec_ok && cols == IIR_TS;
        if (vec) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e = tid + NTHR * k, r = e / CH, cc = (e % CH) * PER;
#pragma unroll
                for (int i = 0; i < PER; ++i) tin[r * IIR_LD + cc + i] = pre[k][i];
            }
        } else
        for (int e = tid; e < 64 * IIR_TS; e += NTHR) {
            const int r = e / IIR_TS, cidx = e % IIR_TS;
            TIO v = TIO(0);
            if (r < rows && cidx < cols) v = x[(clip0 + r) * stride + t0 + cidx];
            tin[r * IIR_LD + cidx] = v;
        }
        __syncthreads();
        if (vec_ok && t0 + 2 * IIR_TS <= n) issue(t0 + IIR_TS);
        // one sample of this lane's recurrence (+ the segment sums when MEANS)
        // (the segment bookkeeping is derived from the wave-uniform sample index s, so it stays in scalar registers)
        auto sample = [&](T xv, int s) -> TIO {
            const TIO o = (TIO)iir_step<T, C, EVEN_B>(st, c, xv);
            if (MEANS) {
                // sample s = 224 k + pos belongs to segment k and, for pos < 32, still to segment k - 1
                const int k = s / kSpecHop, pos = s - k * kSpecHop;
                if (pos == 0) { prev = cur; cur = 0.0f; }
                cur = cur + (float)o;
                if (pos < kSpecSeg - kSpecHop && k >= 1) {
                    prev = prev + (float)o;
                    if (pos == kSpecSeg - kSpecHop - 1 && k - 1 < n_seg) means[(clip0 + lane) * n_seg + k - 1] = prev / (float)kSpecSeg;
                }
            }
            return o;
        };
        if (lane < rows && cols == IIR_TS) {
            // full tile: the row goes LDS -> registers -> LDS in bursts, so the LDS latency is paid once per
            // 16 samples and not once per sample on top of the recurrence's own dependency chain
#pragma unroll
            for (int h = 0; h < IIR_TS; h += IIR_BURST) {
                T xr[IIR_BURST];
                TIO orr[IIR_BURST];
#pragma unroll
                for (int i = 0; i < IIR_BURST; ++i) xr[i] = (T)tin[lane * IIR_LD + h + i];
#pragma unroll
                for (int i = 0; i < IIR_BURST; ++i) orr[i] = sample(xr[i], t0 + h + i);
#pragma unroll
                for (int i = 0; i < IIR_BURST; ++i) to[lane * IIR_LD + h + i] = orr[i];
            }
        } else if (lane < rows) {
            for (int i = 0; i < cols; ++i) {
                to[lane * IIR_LD + i] = sample((T)tin[lane * IIR_LD + i], t0 + i);
            }
        }
        __syncthreads();
        // each wavefront stores its own filter's tile
        if (vec) {
            for (int e = lane; e < 64 * CH; e += 64) {
                const int r = e / CH, cc = (e % CH) * PER;
                if (r < rows) {
                    VIO q;
#pragma unroll
                    for (int i = 0; i < PER; ++i) q[i] = to[r * IIR_LD + cc + i];
#ifdef DSP_IIR_DIAG_NO_STORE        // timing-only probe: the tiles are computed and staged, not stored
                    if (q[0] == (TIO)123.456)
#endif
                    __builtin_nontemporal_store(q, reinterpret_cast<VIO *>(y + (clip0 + r) * ystride + t0 + cc));
                }
            }
        } else
        for (int e = lane; e < 64 * IIR_TS; e += 64) {
            const int r = e / IIR_TS, cidx = e % IIR_TS;
            if (r < rows && cidx < cols) y[(clip0 + r) * ystride + t0 + cidx] = to[r * IIR_LD + cidx];
        }
        // no barrier here: the next iteration refills tin (read before the barrier above) and the barrier
        // after that refill orders this tile's reads of tout before the next tile's writes
    }
}

// ---------------------------------------------------------------------------------
// a9 for classify(): both band-pass filters of a clip batch with the recurrence and the output
// taps on DIFFERENT wavefronts.  Direct form II is v[n] = x[n] - sum a[j] v[n-j] (serial) followed
// by y[n] = b0 v[n] + sum b[j] v[n-j] (an FIR over v, no feedback).  With a few thousand clips the
// lane-per-clip kernel is bound by one wavefront's issue latency (~7 cycles per dependent VALU
// instruction), so halving the instructions each wavefront executes per sample nearly halves the
// time: waves 0/1 run the recurrences of filter 1/2 and hand v tiles over through LDS, waves 2/3
// run the taps one tile behind, keep the spectrogram's segment sums and store y.  Every sample
// sees exactly the reference's operations in the reference's order (classifier.cpp:199-216).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iir2_split_kernel(const float *__restrict__ x, long n_clips, int n, long stride, long ystride,
                                                         const IirCoef c1, float *__restrict__ y1, const IirCoef c2, float *__restrict__ y2,
                                                         float *__restrict__ means1, float *__restrict__ means2,
                                                         const SpecTables *__restrict__ tab, int *__restrict__ gate2)
{
    __shared__ float tin[2][64 * IIR_LD];
    __shared__ float vbuf[2][2][64 * IIR_LD];          // [filter][tile parity]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = wv & 1;                               // filter
    const bool is_r = wv < 2;                           // recurrence wave (else: taps wave)
    const IirCoef c = f ? c2 : c1;
    float *__restrict__ y = f ? y2 : y1;
    float *__restrict__ means = f ? means2 : means1;
    const long clip0 = (long)blockIdx.x * 64;
    const int rows = (int)((n_clips - clip0) < 64 ? (n_clips - clip0) : 64);
    const int n_tiles = (n + IIR_TS - 1) / IIR_TS;
    const int n_seg = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    float d[8];                                         // v[n-1] .. v[n-8] of this lane's clip
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = 0.0f;
    float cur = 0.0f, prev = 0.0f;                      // taps waves: running sums of the current / previous segment
    // gate2 (filter 2, needs means2 and tab): an upper bound of each segment's windowed, mean-removed energy
    // E = sum w^2 (y - m)^2 = A - 2 m B + m^2 C <= A + 2 |m| |B| + m^2 C with A = sum w^2 y^2, B = sum w^2 y, C = sum w^2.
    // gate2[clip][k] = 0 when 512 E / U (Parseval bound of every PSD cell, 1 % margin for the float sums) stays below
    // the 70 dB threshold: spectrogram_kernel<SPEC_FLAGS> then never reads that frame.
    const bool gating = gate2 != nullptr && f == 1 && means != nullptr && tab != nullptr && tab->gate_ok != 0;
    float ea_cur = 0.0f, eb_cur = 0.0f, ea_prev = 0.0f, eb_prev = 0.0f;

    // x tiles: the 128 recurrence threads load them (4 float4 each per full tile), one tile ahead in registers
    constexpr int CH = IIR_TS / 4, NV = 64 * CH / 128;
    float4 pre[NV];
    auto issue = [&](int t0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + 128 * k, r = e / CH, cc = (e % CH) * 4;
            pre[k] = r < rows ? *reinterpret_cast<const float4 *>(x + (clip0 + r) * stride + t0 + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](float *dst) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + 128 * k, r = e / CH, cc = (e % CH) * 4;
            dst[r * IIR_LD + cc] = pre[k].x; dst[r * IIR_LD + cc + 1] = pre[k].y;
            dst[r * IIR_LD + cc + 2] = pre[k].z; dst[r * IIR_LD + cc + 3] = pre[k].w;
        }
    };
    auto load_partial = [&](int t0, int cols, float *dst) {       // last, short tile: element-wise
        for (int e = tid; e < 64 * IIR_TS; e += 128) {
            const int r = e / IIR_TS, ci = e % IIR_TS;
            dst[r * IIR_LD + ci] = (r < rows && ci < cols) ? x[(clip0 + r) * stride + t0 + ci] : 0.0f;
        }
    };
    auto tile_cols = [&](int s) { const int t0 = s * IIR_TS; return n - t0 < IIR_TS ? n - t0 : IIR_TS; };
    if (is_r) {                                                    // tile 0 into tin[0], tile 1 in flight
        if (tile_cols(0) == IIR_TS) { issue(0); commit(tin[0]); } else load_partial(0, tile_cols(0), tin[0]);
        if (n_tiles > 1 && tile_cols(1) == IIR_TS) issue(IIR_TS);
    }
    __syncthreads();

    for (int s = 0; s <= n_tiles; ++s) {
        if (is_r) {
            if (s < n_tiles) {
                // stage x tile s+1 for the next step, start the loads of tile s+2
                if (s + 1 < n_tiles) {
                    if (tile_cols(s + 1) == IIR_TS) commit(tin[(s + 1) & 1]); else load_partial((s + 1) * IIR_TS, tile_cols(s + 1), tin[(s + 1) & 1]);
                    if (s + 2 < n_tiles && tile_cols(s + 2) == IIR_TS) issue((s + 2) * IIR_TS);
                }
                const float *xin = tin[s & 1];
                float *vo = vbuf[f][s & 1];
                const int cols = tile_cols(s);
                auto rec = [&](float xv) {                           // classifier.cpp:199-205
                    float v = xv;
#pragma unroll
                    for (int j = 1; j <= 8; ++j) v = v - c.a[j] * d[j - 1];
#pragma unroll
                    for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                    d[0] = v;
                    return v;
                };
                if (lane < rows && cols == IIR_TS) {
#pragma unroll
                    for (int h = 0; h < IIR_TS; h += IIR_BURST) {
                        float xr[IIR_BURST], vr[IIR_BURST];
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) xr[i] = xin[lane * IIR_LD + h + i];
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) vr[i] = rec(xr[i]);
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) vo[lane * IIR_LD + h + i] = vr[i];
                    }
                } else if (lane < rows) {
                    for (int i = 0; i < cols; ++i) vo[lane * IIR_LD + i] = rec(xin[lane * IIR_LD + i]);
                }
            }
        } else if (s >= 1) {
            const int t0 = (s - 1) * IIR_TS, cols = tile_cols(s - 1);
            // y replaces v in place (each lane rewrites the row it has just read), so the taps wave needs no tile of its own:
            // 6 tiles = 50 KB per block, three blocks per CU
            float *yo = vbuf[f][(s - 1) & 1];
            const float *vin = yo;
            auto taps = [&](float v) {                               // classifier.cpp:207-216
                float o = c.b[0] * v;
#pragma unroll
                for (int j = 1; j <= 8; ++j) o = o + c.b[j] * d[j - 1];
#pragma unroll
                for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                d[0] = v;
                return o;
            };
            // The spectrogram's segments (256 samples every 224) start on tile boundaries and overlap by exactly one
            // tile: tile 7k is the first tile of segment k and the last one of segment k-1.  Their sequential sums
            // (classifier.cpp:329-333) are carried per tile: one add per sample, two in the shared tile.
            static_assert(kSpecHop % IIR_TS == 0 && kSpecSeg - kSpecHop == IIR_TS, "segment sums are kept per IIR tile");
            constexpr int kTilesPerHop = kSpecHop / IIR_TS;
            const int ti = s - 1, seg_k = ti / kTilesPerHop;
            const bool seg_start = means != nullptr && ti % kTilesPerHop == 0;
            const bool seg_both = seg_start && seg_k >= 1;           // the tile also closes segment seg_k - 1
            if (seg_start) { prev = cur; cur = 0.0f; ea_prev = ea_cur; eb_prev = eb_cur; ea_cur = eb_cur = 0.0f; }
            if (lane < rows && cols == IIR_TS) {
#pragma unroll
                for (int h = 0; h < IIR_TS; h += IIR_BURST) {
                    float vr[IIR_BURST], orr[IIR_BURST];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) vr[i] = vin[lane * IIR_LD + h + i];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) orr[i] = taps(vr[i]);
                    if (seg_both) {
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) { cur = cur + orr[i]; prev = prev + orr[i]; }
                    } else if (means) {
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) cur = cur + orr[i];
                    }
                    if (gating) {
                        if (seg_start) {                             // tapered tile: in for the new segment, out for the old one
#pragma unroll
                            for (int i = 0; i < IIR_BURST; ++i) {
                                const float wi = tab->win2_in[h + i] * orr[i], wo = tab->win2_out[h + i] * orr[i];
                                ea_cur = fmaf(wi, orr[i], ea_cur); eb_cur = eb_cur + wi;
                                ea_prev = fmaf(wo, orr[i], ea_prev); eb_prev = eb_prev + wo;
                            }
                        } else {                                     // the window is 1 here
#pragma unroll
                            for (int i = 0; i < IIR_BURST; ++i) { ea_cur = fmaf(orr[i], orr[i], ea_cur); eb_cur = eb_cur + orr[i]; }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) yo[lane * IIR_LD + h + i] = orr[i];
                }
                if (seg_both && seg_k - 1 < n_seg) {
                    const float m = prev / (float)kSpecSeg;
                    means[(clip0 + lane) * n_seg + seg_k - 1] = m;
                    if (gating) {
                        const float e = ea_prev + 2.0f * fabsf(m) * fabsf(eb_prev) + m * m * tab->win2_sum;
                        gate2[(clip0 + lane) * n_seg + seg_k - 1] = e * tab->gate_scale >= tab->mp_keep_min ? 1 : 0;
                    }
                }
            } else if (lane < rows) {
                // a short last tile lies past every whole segment: no sums to keep
                for (int i = 0; i < cols; ++i) yo[lane * IIR_LD + i] = taps(vin[lane * IIR_LD + i]);
            }
            // this wave's own tile: wave-level ordering is enough before the coalesced store
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (cols == IIR_TS) {
                for (int e = lane; e < 64 * CH; e += 64) {
                    const int r = e / CH, cc = (e % CH) * 4;
                    if (r < rows) {                                  // streamed out: nothing in this kernel reads it back (-10 %)
                        const f4nt q = {yo[r * IIR_LD + cc], yo[r * IIR_LD + cc + 1], yo[r * IIR_LD + cc + 2], yo[r * IIR_LD + cc + 3]};
                        __builtin_nontemporal_store(q, reinterpret_cast<f4nt *>(y + (clip0 + r) * ystride + t0 + cc));
                    }
                }
            } else {
                for (int e = lane; e < 64 * IIR_TS; e += 64) {
                    const int r = e / IIR_TS, ci = e % IIR_TS;
                    if (r < rows && ci < cols) y[(clip0 + r) * ystride + t0 + ci] = yo[r * IIR_LD + ci];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
    }
}

hipError_t launch_iir_f32(const float *x, long n_clips, int n, long stride, const IirCoef &c1, float *y1,
                          const IirCoef &c2, float *y2, hipStream_t stream, float *means1, float *means2,
                          const SpecTables *tables, int *gate2, long ystride, bool gate_tables_ok)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    if (ystride <= 0) ystride = stride;
    const int blocks = (int)((n_clips + 63) / 64);
    const bool aligned = stride % 4 == 0 && ystride % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(y1) % 16 == 0 && reinterpret_cast<uintptr_t>(y2) % 16 == 0;
    // only the split kernel computes the gate, and only when the window tables allow it (SpecTables::gate_ok): otherwise
    // every frame is "maybe" (non-zero ints)
    if (gate2 && !(y2 && aligned && means2 && tables && gate_tables_ok)) {
        const int n_seg = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
        hipError_t e = hipMemsetAsync(gate2, 1, (size_t)n_clips * n_seg * sizeof(int), stream);
        if (e != hipSuccess) return e;
    }
    if (y2 && aligned)       // classify(): recurrence / taps split over four wavefronts (means1 / means2 may be nullptr)
        hipLaunchKernelGGL(iir2_split_kernel, dim3(blocks), dim3(256), 0, stream, x, n_clips, n, stride, ystride, c1, y1, c2, y2, means1, means2, tables, gate2);
    else if (y2 && means1 && means2)
        hipLaunchKernelGGL((iir_kernel<float, IirCoef, true, float, true>), dim3(blocks), dim3(128), 0, stream, x, n_clips, n, stride, ystride, c1, y1, c2, y2,
                           means1, means2);
    else if (y2) hipLaunchKernelGGL((iir_kernel<float, IirCoef, true>), dim3(blocks), dim3(128), 0, stream, x, n_clips, n, stride, ystride, c1, y1, c2, y2, nullptr, nullptr);
    else hipLaunchKernelGGL((iir_kernel<float, IirCoef, false>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, ystride, c1, y1, c1, y1, nullptr, nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// a9 for classify(), checkpoint form: ONE pass over x runs both band-pass recurrences and writes back NEITHER filtered
// signal.  What classify() needs from the filtered signals are spectrogram segments (256 samples every 224), and a direct
// form II filter can be restarted anywhere from its delay line v[n-1..n-8]: the kernel stores that state at every segment
// start and (round 3) at every segment's middle (32 bytes each; 9 KB per 1 s clip and filter at most, instead of 128 KB of
// filtered samples), and the spectrogram kernels recompute exactly the segments they transform -- same operations on the same
// values in the same order, so the same bits (spec_from_ckpt_kernel).  For the 1000-3000 Hz filter the output taps run here as well, because its segments'
// sequential sums (classifier.cpp:329-333) and the energy gate (see iir2_split_kernel) come from every sample; the
// 3000-7500 Hz filter needs its taps only for clips that turn out to have midpoints, so they wait for the recompute.
//   part 0: recurrence 3000-7500 Hz (checkpoints to HBM)   part 1: recurrence 1000-3000 Hz (checkpoints parked in LDS, v tiles to LDS)
//   part 2: taps 1000-3000 Hz one tile behind (segment means, energy gate; writes the parked checkpoints of gated-in segments)
// Which wave of the block runs which part follows the SIMD loads the launch measures itself (simd_load, below).
// HBM traffic: x once + 71 x (64 + 4) B per 1 s clip + 64 + 4 B per gated-in segment.
// ---------------------------------------------------------------------------------
// DUAL (round 3, an experiment; default off): ONE wave runs both recurrences of its 64 clips as PACKED fp32 operations -- lane l
// holds the pair (1000-3000 Hz, 3000-7500 Hz) of its clip's delay lines, and v = v - a[j] d[j-1] is one v_pk_mul_f32 + one
// v_pk_add_f32 for both filters (component-wise IEEE multiply and add, no contraction: the same bits as two scalar chains; the 50
// classify tests stay array_equal).  The block then has two waves (recurrences, taps) instead of three.  Measured on 49 152 clips
// (tools/ab_classify.py, interleaved): three waves 2.58 ms, packed pair 2.79 ms, two SCALAR chains in one wave 2.85 ms.  What limits
// a recurrence wave is how often ONE wave gets to issue (16 operations per ~110-cycle sample step, ~7 cycles apiece whatever their
// dependencies); a packed operation issues for twice as long, so one wave doing both filters is slower than two waves doing one each.
// ---- int16 PCM in the classifier kernels' loads (SURVEY 8f-1's second reader): IN = 0 float samples, 1 int16 mono (s / 32768,
// sync/sync.cpp:237-242, donut-classifier/classifier.c:55-59), 2 interleaved int16 stereo, channel 0 (classifier.c:286-297), 3 stereo,
// (L + R) / 65536 (the average of the channels' s / 32768, main_test.c:205-217: exact in float for |L + R| < 2^24).  The conversions are
// exact, so the kernels see the float path's values bit for bit.  A 16-byte piece holds 4 / 8 / 4 / 4 samples.
typedef unsigned cls_u4 __attribute__((ext_vector_type(4)));
template <int IN> struct ClsIn { static constexpr int kBytes = IN == 0 ? 4 : (IN == 1 ? 2 : 4), kPerPiece = 16 / kBytes; };
template <int IN>
__device__ __forceinline__ void cls_piece_to_float(const cls_u4 &q, float (&o)[ClsIn<IN>::kPerPiece])
{
    if constexpr (IN == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = __uint_as_float(q[k]);
    } else if constexpr (IN == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = (float)(int)(short)(q[k] & 0xffffu) * (1.0f / 32768.0f);
            o[2 * k + 1] = (float)((int)q[k] >> 16) * (1.0f / 32768.0f);
        }
    } else if constexpr (IN == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (float)(int)(short)(q[k] & 0xffffu) * (1.0f / 32768.0f);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (float)((int)(short)(q[k] & 0xffffu) + ((int)q[k] >> 16)) * (1.0f / 65536.0f);
    }
}
template <int IN>
__device__ __forceinline__ float cls_sample_at(const void *__restrict__ x, long idx)
{
    if constexpr (IN == 0) return reinterpret_cast<const float *>(x)[idx];
    else if constexpr (IN == 1) return (float)reinterpret_cast<const short *>(x)[idx] * (1.0f / 32768.0f);
    else if constexpr (IN == 2) return (float)reinterpret_cast<const short *>(x)[2 * idx] * (1.0f / 32768.0f);
    else return (float)((int)reinterpret_cast<const short *>(x)[2 * idx] + (int)reinterpret_cast<const short *>(x)[2 * idx + 1]) * (1.0f / 65536.0f);
}

#ifndef DSP_CKPT_DUAL
#define DSP_CKPT_DUAL 0
#endif
typedef float ck_f2 __attribute__((ext_vector_type(2)));
// RAGGED: clips of different lengths (spans[clip]: start, segments).  The workspaces keep the uniform [clip][n_seg(n)] layout, n = the
// longest clip; a block walks as many tiles as its own longest clip's WHOLE segments cover (a multiple of the tile: a tail past the last
// whole segment feeds nothing, see the taps wave) and a lane simply stops keeping states / means / gates at its clip's last segment --
// what it computes past that point (the next clip's samples, zeros past the buffer's end) is never looked at.
template <bool EVEN_B, bool DUAL, int IN = 0, bool RAGGED = false>
__global__ __launch_bounds__(DUAL ? 128 : 192) void iir2_ckpt_kernel(const void *__restrict__ xv, long n_clips, int n_arg, long stride,
                                                        const IirCoef c_bp, const IirCoef c_mp, float *__restrict__ ck_bp,
                                                        float *__restrict__ ck_mp, float *__restrict__ means_mp,
                                                        int *__restrict__ want_mp, const SpecTables *__restrict__ tab, int vec_ok, int *__restrict__ simd_load,
                                                        int blocks_per_cu, const ClipSpan *__restrict__ spans = nullptr, long total = 0)
{
    static_assert(!(RAGGED && DUAL), "the ragged form exists for the three-wave kernel");
    __shared__ float tin[2][64 * IIR_LD];
    __shared__ float vbuf[2][64 * IIR_LD];             // v tiles of the 1000-3000 Hz filter, [tile parity]
    // The 1000-3000 Hz restart states wait in LDS until the taps wave has the segment's energy gate: only the gated-in segments
    // (8 % in the bench workload) are ever recomputed, so only their states go to HBM (4.1 of the 4.5 KB per clip and filter
    // stay on chip).  Three entries are live at most: start of segment k (by parity), its middle, start of segment k + 1.
    __shared__ float ckbuf[DUAL ? 1 : 3][8][64];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Which wave takes which part (wv: 0 = 3000-7500 Hz recurrence, 1 = 1000-3000 Hz recurrence, 2 = taps) follows the load of the
    // SIMD it landed on.  49 152 clips are three 3-wave blocks per CU: nine waves on four SIMDs, 3/2/2/2, every block with one wave
    // on the crowded SIMD -- and every block runs at that SIMD's pace (two recurrences and a taps wave there: 153 cycles per sample
    // against ~115 for a recurrence wave with one neighbour).  The dispatcher places each wave on the least-loaded SIMD, so WHICH
    // SIMD is crowded differs from CU to CU (tools/micro/hwid.hip): the waves of a launch count themselves into simd_load[CU][SIMD]
    // (HW_ID / XCC_ID), wait until the CU's other blocks have done the same, and each block gives the TAPS -- a tile behind, the
    // shorter dependent chains -- to its wave on the most loaded SIMD (ties go round by the block's arrival number), so a crowded
    // SIMD holds taps waves and no recurrence wave has more than one neighbour.  Scheduling only: what a part computes does not
    // depend on the wave that runs it, and a block that reads the table too early merely keeps a poorer choice.
    int wv = wib;
    if (!DUAL && simd_load != nullptr) {
        __shared__ int s_simd[3], s_taps;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;     // HW_ID, XCC_ID
        int *row = simd_load + (((xcc << 8) | ((hw >> 8) & 0xFFu)) & (kSimdLoadCus - 1)) * kSimdLoadStride;       // (XCC, SE, SH, CU)
        if (lane == 0) {
            const int before = atomicAdd(row + ((hw >> 4) & 3u), 1);                // returning form: complete before the barrier below
            s_simd[wib] = (int)((hw >> 4) & 3u) + (before & 0);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int arrival = atomicAdd(row + 4, 1);                               // every wave of this block is in the table
            // wait until the CU's share of the launch has registered (blocks of a launch are dispatched within microseconds of each
            // other and all fit: no block waits for one that cannot start), at most ~60 us; then decide
            for (int i = 0; i < 64 && atomicAdd(row + 4, 0) < blocks_per_cu; ++i) __builtin_amdgcn_s_sleep(32);
            int cnt[3], mx = -1, n_tied = 0;
            for (int w = 0; w < 3; ++w) { cnt[w] = atomicAdd(row + s_simd[w], 0); mx = cnt[w] > mx ? cnt[w] : mx; }
            for (int w = 0; w < 3; ++w) n_tied += cnt[w] == mx;
            int pick = arrival % n_tied, tw = 2;
            for (int w = 0; w < 3; ++w) if (cnt[w] == mx && pick-- == 0) tw = w;
            s_taps = tw;
        }
        __syncthreads();
        const int tw = s_taps;
        wv = __builtin_amdgcn_readfirstlane(wib == tw ? 2 : (wib < tw ? wib : wib - 1));
    }
    // the recurrence waves go ahead of the taps waves in the SIMD's arbiter (the taps lag a tile behind anyway): with the
    // SIMD-aware parts above -1.2 % (2.359 -> 2.331 ms per 49 152 clips); before them the same priorities were neutral to +3 %
#ifndef DSP_CKPT_PRIO
#define DSP_CKPT_PRIO 3
#endif
    if (DSP_CKPT_PRIO && wv < 2) __builtin_amdgcn_s_setprio(DSP_CKPT_PRIO);
    const int tid = wv * 64 + lane;                     // thread number by part: the 128 recurrence threads stage the x tiles
    const bool is_r = DUAL ? wv == 0 : wv < 2;          // runs a recurrence (DUAL: both)
    const bool is_t = DUAL ? wv == 1 : wv == 2;         // runs the 1000-3000 Hz taps
    const bool loader = DUAL ? true : wv < 2;           // the 128 threads that stage the x tiles
    const IirCoef c = (DUAL || wv != 0) ? c_mp : c_bp;   // the wave's (second) filter: 1000-3000 Hz except for !DUAL wave 0
    float *__restrict__ ck = (DUAL || wv != 0) ? ck_mp : ck_bp;
    const long clip0 = (long)blockIdx.x * 64;
    const int rows = (int)((n_clips - clip0) < 64 ? (n_clips - clip0) : 64);
    const int n_seg = n_arg < kSpecSeg ? 0 : (n_arg - kSpecSeg) / kSpecHop + 1;      // the workspaces' row length (RAGGED: the longest clip's)
    __shared__ long s_base[RAGGED ? 64 : 1];
    __shared__ int s_nseg[RAGGED ? 64 : 1], s_cover;
    int n = n_arg, my_nseg = n_seg;                     // RAGGED: samples this block walks, segments of this lane's clip
    if (RAGGED) {
        if (threadIdx.x == 0) s_cover = 0;
        __syncthreads();
        if (threadIdx.x < 64) {
            ClipSpan sp{0, 0, 0};
            if ((int)threadIdx.x < rows) sp = spans[clip0 + threadIdx.x];
            s_base[threadIdx.x] = sp.off; s_nseg[threadIdx.x] = sp.frames;
            if (sp.frames > 0) atomicMax(&s_cover, (sp.frames - 1) * kSpecHop + kSpecSeg);
        }
        __syncthreads();
        n = s_cover; my_nseg = s_nseg[lane];
    }
    const int n_tiles = (n + IIR_TS - 1) / IIR_TS;
    static_assert(kSpecHop % IIR_TS == 0 && kSpecSeg - kSpecHop == IIR_TS, "segments start on tile boundaries and overlap by one tile");
    constexpr int kTilesPerHop = kSpecHop / IIR_TS;
    float d[8];                                         // v[n-1] .. v[n-8] of this lane's clip
    ck_f2 dp[8], ap[9];                                 // DUAL: (1000-3000 Hz, 3000-7500 Hz) pairs of delay line and a[j]
#pragma unroll
    for (int j = 0; j < 8; ++j) { d[j] = 0.0f; dp[j] = ck_f2{0.0f, 0.0f}; }
#pragma unroll
    for (int j = 0; j < 9; ++j) ap[j] = ck_f2{c_mp.a[j], c_bp.a[j]};
    (void)dp; (void)ap;
    float cur = 0.0f, prev = 0.0f;
    const bool gating = tab->gate_ok != 0;
    float ea_cur = 0.0f, eb_cur = 0.0f, ea_prev = 0.0f, eb_prev = 0.0f;
    // The gate's B = sum w^2 y of a segment: in the six tiles where the window is 1 it is the same sum as `cur`, so only the tapered
    // first tile is accumulated (eb_cur) and the flat part is taken as cur - c_first when the segment closes (c_first = `cur` after
    // the first tile): one instruction per sample less on the taps wave.  B only enters an upper bound with 1 % margin; the
    // difference of two running sums is within a few ulp of sum |y| of the separately accumulated value.
    float c_first = 0.0f;

    // x tiles: the 128 recurrence threads load them (4 float4 each per full tile), one tile ahead in registers.  (Two tiles
    // ahead in a second register set measured slower, 1.50 vs 1.34 ms per 49 152 clips: the kernel is bound by VALU issue --
    // 17 + 17 + ~26 instructions per sample and clip --, not by the load latency.)
    // (a tile row is IIR_TS samples = 8 / 4 / 8 / 8 16-byte pieces for IN = 0 / 1 / 2 / 3)
    constexpr int PP = ClsIn<IN>::kPerPiece, CH = IIR_TS / PP, NV = 64 * CH / 128;
    cls_u4 pre[NV];
    auto issue = [&](int t0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + 128 * k, r = e / CH, cc = (e % CH) * PP;
            if (RAGGED) {      // any alignment (the hardware's unaligned loads); a piece that would cross the buffer's end lies past every whole segment
                const long idx = s_base[r] + t0 + cc;
                pre[k] = (r < rows && idx + PP <= total) ? *reinterpret_cast<const cls_u4 *>(reinterpret_cast<const unsigned char *>(xv) + idx * ClsIn<IN>::kBytes)
                                                         : cls_u4{0u, 0u, 0u, 0u};
            } else
            pre[k] = r < rows ? *reinterpret_cast<const cls_u4 *>(reinterpret_cast<const unsigned char *>(xv) + ((clip0 + r) * stride + t0 + cc) * ClsIn<IN>::kBytes)
                              : cls_u4{0u, 0u, 0u, 0u};
        }
    };
    auto commit = [&](float *dst) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + 128 * k, r = e / CH, cc = (e % CH) * PP;
            float o[PP];
            cls_piece_to_float<IN>(pre[k], o);
#pragma unroll
            for (int i = 0; i < PP; ++i) dst[r * IIR_LD + cc + i] = o[i];
        }
    };
    auto load_scalar = [&](int t0, int cols, float *dst) {        // short last tile, or rows that are not 16-byte aligned
        for (int e = tid; e < 64 * IIR_TS; e += 128) {
            const int r = e / IIR_TS, ci = e % IIR_TS;
            dst[r * IIR_LD + ci] = (r < rows && ci < cols) ? cls_sample_at<IN>(reinterpret_cast<const unsigned char *>(xv) + (clip0 + r) * stride * ClsIn<IN>::kBytes, t0 + ci) : 0.0f;
        }
    };
    auto tile_cols = [&](int s) { const int t0 = s * IIR_TS; return n - t0 < IIR_TS ? n - t0 : IIR_TS; };
    auto fast = [&](int s) { return RAGGED || (vec_ok && tile_cols(s) == IIR_TS); };      // (RAGGED: every tile is whole)
    if (loader) {                                                  // tile 0 into tin[0], tile 1 in flight
        if (fast(0)) { issue(0); commit(tin[0]); } else load_scalar(0, tile_cols(0), tin[0]);
        if (n_tiles > 1 && fast(1)) issue(IIR_TS);
    }
    __syncthreads();

    for (int s = 0; s <= n_tiles; ++s) {
        if (loader && s + 1 < n_tiles) {
            // stage x tile s+1 for the next step, start the loads of tile s+2
            if (fast(s + 1)) commit(tin[(s + 1) & 1]); else load_scalar((s + 1) * IIR_TS, tile_cols(s + 1), tin[(s + 1) & 1]);
            if (s + 2 < n_tiles && fast(s + 2)) issue((s + 2) * IIR_TS);
        }
        if (is_r) {
            if (s < n_tiles) {
                // segment seg starts with this tile (or reaches its middle, where the wave's filter keeps a second state): its restart
                // state is the delay line as it stands
                const int seg = s / kTilesPerHop, seg_tile = s % kTilesPerHop;
                constexpr int kMidTile = kSpecSeg / 2 / IIR_TS;
                static_assert(kSpecSeg / 2 % IIR_TS == 0 && kMidTile < kTilesPerHop, "the segment's middle lies on a tile boundary");
                const int ck_n = (DUAL || wv != 0) ? kCkPerSegMp : kCkPerSegBp;      // states per segment of `ck`
                const bool at_start = seg_tile == 0, at_mid = seg_tile == kMidTile && (DUAL ? (kCkPerSegBp == 2 || kCkPerSegMp == 2) : ck_n == 2);
                if ((at_start || at_mid) && seg < my_nseg && lane < rows) {
                    const long slot = (((clip0 + lane) * n_seg + seg) * ck_n + (at_mid ? 1 : 0)) * 8;
                    float4 *dst = reinterpret_cast<float4 *>(ck + slot);
                    if (DUAL) {
                        float4 *dst2 = reinterpret_cast<float4 *>(ck_bp + (((clip0 + lane) * n_seg + seg) * kCkPerSegBp + (at_mid ? 1 : 0)) * 8);
                        if (at_start || kCkPerSegMp == 2) {
                            dst[0] = make_float4(dp[0].x, dp[1].x, dp[2].x, dp[3].x);
                            dst[1] = make_float4(dp[4].x, dp[5].x, dp[6].x, dp[7].x);
                        }
                        if (at_start || kCkPerSegBp == 2) {
                            dst2[0] = make_float4(dp[0].y, dp[1].y, dp[2].y, dp[3].y);
                            dst2[1] = make_float4(dp[4].y, dp[5].y, dp[6].y, dp[7].y);
                        }
                    } else if (wv != 0) {                            // 1000-3000 Hz: parked in LDS for the taps wave's gate
                        float (*e)[64] = ckbuf[at_mid ? 1 : (seg & 1) * 2];
#pragma unroll
                        for (int j = 0; j < 8; ++j) e[j][lane] = d[j];
                    } else {
                        dst[0] = make_float4(d[0], d[1], d[2], d[3]);
                        dst[1] = make_float4(d[4], d[5], d[6], d[7]);
                    }
                }
                const float *xin = tin[s & 1];
                float *vo = vbuf[s & 1];
                const int cols = tile_cols(s);
                auto rec = [&](float xv) {                           // classifier.cpp:199-205
                    float v = xv;
#pragma unroll
                    for (int j = 1; j <= 8; ++j) v = v - c.a[j] * d[j - 1];
#pragma unroll
                    for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                    d[0] = v;
                    return v;
                };
                auto recp = [&](float xv) {                          // DUAL: both filters on the sample, packed (x: 1000-3000 Hz, y: 3000-7500 Hz)
#pragma clang fp contract(off)
                    ck_f2 v = ck_f2{xv, xv};
#pragma unroll
                    for (int j = 1; j <= 8; ++j) v = v - ap[j] * dp[j - 1];
#pragma unroll
                    for (int j = 7; j > 0; --j) dp[j] = dp[j - 1];
                    dp[0] = v;
                    return v.x;
                };
                const bool keep_v = DUAL || wv == 1;                 // the 1000-3000 Hz v tiles go to the taps wave
                if (lane < rows && cols == IIR_TS) {
#pragma unroll
                    for (int h = 0; h < IIR_TS; h += IIR_BURST) {
                        float xr[IIR_BURST], vr[IIR_BURST];
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) xr[i] = xin[lane * IIR_LD + h + i];
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) vr[i] = DUAL ? recp(xr[i]) : rec(xr[i]);
                        if (keep_v) {
#pragma unroll
                            for (int i = 0; i < IIR_BURST; ++i) vo[lane * IIR_LD + h + i] = vr[i];
                        }
                    }
                } else if (lane < rows) {
                    for (int i = 0; i < cols; ++i) {
                        const float xv = xin[lane * IIR_LD + i];
                        const float v = DUAL ? recp(xv) : rec(xv);
                        if (keep_v) vo[lane * IIR_LD + i] = v;
                    }
                }
            }
        } else if (is_t && s >= 1) {
            const int cols = tile_cols(s - 1);
            const float *vin = vbuf[(s - 1) & 1];
            auto taps = [&](float v) {                               // classifier.cpp:207-216
                float o = c.b[0] * v;
                // EVEN_B: b[1] = b[3] = b[5] = b[7] = 0 exactly (a Butterworth band-pass numerator is (1 - z^-2)^4 scaled: both
                // literal tables).  Their products are +-0 for finite data and o + (+-0) == o, so skipping them changes at most
                // the sign of an exact zero -- which no PSD value, hence no output of classify(), can see.
#pragma unroll
                for (int j = 1; j <= 8; ++j)
                    if (!EVEN_B || j % 2 == 0) o = o + c.b[j] * d[j - 1];
#pragma unroll
                for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                d[0] = v;
                return o;
            };
            const int ti = s - 1, seg_k = ti / kTilesPerHop;
            const bool seg_start = ti % kTilesPerHop == 0;
            const bool seg_both = seg_start && seg_k >= 1;           // the tile also closes segment seg_k - 1
            if (seg_start) { prev = cur; ea_prev = ea_cur; eb_prev = eb_cur + (cur - c_first); cur = 0.0f; ea_cur = eb_cur = 0.0f; }
            if (lane < rows && cols == IIR_TS) {
#pragma unroll
                for (int h = 0; h < IIR_TS; h += IIR_BURST) {
                    float vr[IIR_BURST], orr[IIR_BURST];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) vr[i] = vin[lane * IIR_LD + h + i];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) orr[i] = taps(vr[i]);
                    if (seg_both) {
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) { cur = cur + orr[i]; prev = prev + orr[i]; }
                    } else {
#pragma unroll
                        for (int i = 0; i < IIR_BURST; ++i) cur = cur + orr[i];
                    }
                    if (gating) {
                        if (seg_start) {                             // tapered tile: in for the new segment, out for the old one
#pragma unroll
                            for (int i = 0; i < IIR_BURST; ++i) {
                                const float wi = tab->win2_in[h + i] * orr[i], wo = tab->win2_out[h + i] * orr[i];
                                ea_cur = fmaf(wi, orr[i], ea_cur); eb_cur = eb_cur + wi;
                                ea_prev = fmaf(wo, orr[i], ea_prev); eb_prev = eb_prev + wo;
                            }
                        } else {                                     // the window is 1 here
#pragma unroll
                            for (int i = 0; i < IIR_BURST; ++i) ea_cur = fmaf(orr[i], orr[i], ea_cur);
                        }
                    }
                }
                if (seg_start) c_first = cur;
                if (seg_both && seg_k - 1 < my_nseg) {
                    const float m = prev / (float)kSpecSeg;
                    means_mp[(clip0 + lane) * n_seg + seg_k - 1] = m;
                    int g = 1;
                    if (gating) {
                        const float e = ea_prev + 2.0f * fabsf(m) * fabsf(eb_prev) + m * m * tab->win2_sum;
                        g = e * tab->gate_scale >= tab->mp_keep_min ? 1 : 0;
                    }
                    // work list of the flag spectrogram: want_mp[0] = count, then frame numbers clip * T + t, in any order
                    // (hipcc turns the per-lane add into one atomic per wave)
                    if (g) {
                        want_mp[1 + atomicAdd(want_mp, 1)] = (int)((clip0 + lane) * n_seg + seg_k - 1);
                        if (!DUAL) {                                 // the segment will be recomputed: its restart states leave LDS
                            const int sg = seg_k - 1;
#pragma unroll
                            for (int h = 0; h < kCkPerSegMp; ++h) {
                                const float (*e)[64] = ckbuf[h ? 1 : (sg & 1) * 2];
                                float4 *dst = reinterpret_cast<float4 *>(ck + (((clip0 + lane) * n_seg + sg) * kCkPerSegMp + h) * 8);
                                dst[0] = make_float4(e[0][lane], e[1][lane], e[2][lane], e[3][lane]);
                                dst[1] = make_float4(e[4][lane], e[5][lane], e[6][lane], e[7][lane]);
                            }
                        }
                    }
                }
            } else if (lane < rows) {
                for (int i = 0; i < cols; ++i) (void)taps(vin[lane * IIR_LD + i]);    // a short last tile lies past every whole segment
            }
        }
        __syncthreads();
    }
}

static bool even_taps_only(const IirCoef &c) { return c.b[1] == 0.0f && c.b[3] == 0.0f && c.b[5] == 0.0f && c.b[7] == 0.0f; }

hipError_t launch_iir2_ckpt(const void *x, long n_clips, int n, long stride, const IirCoef &c_bp, const IirCoef &c_mp,
                            float *ck_bp, float *ck_mp, float *means_mp, int *want_mp, const SpecTables *tables, hipStream_t stream,
                            int *simd_load, int in_kind, const ClipSpan *spans, long total)
{
    // DSP_AMD_CKPT_SIMD_AWARE: 0 = fixed parts (A/B runs), 2 = the table for every launch whatever its size (tests)
    static const int simd_mode = [] { const char *e = std::getenv("DSP_AMD_CKPT_SIMD_AWARE"); return e ? std::atoi(e) : 1; }();
    const bool simd_aware = simd_mode != 0;
    const int blocks = (int)((n_clips + 63) / 64);
    // worth its ~10 us (table reset, the blocks' wait) only when CUs hold three or more blocks: up to two blocks per CU every
    // recurrence wave has at most one neighbour anyway (measured: 12 288 clips +0.03 ms, 24 576 +-0, 49 152 -0.13 ms, 131 072 -0.33 ms)
    static int n_cu_of[64] = {0};                    // per device
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= 64) cur_dev = 0;
    if (n_cu_of[cur_dev] <= 0) { int n = 0; n_cu_of[cur_dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, cur_dev) == hipSuccess && n > 0 ? n : 256; }
    const int n_cu = n_cu_of[cur_dev];
    if (!simd_aware || (blocks <= 2 * n_cu && simd_mode != 2)) simd_load = nullptr;
    if (simd_load) {
        hipError_t e = hipMemsetAsync(simd_load, 0, sizeof(int) * kSimdLoadCus * kSimdLoadStride, stream);
        if (e != hipSuccess) return e;
    }
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    {
        const long n_seg = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
        if (n_clips * n_seg >= (1L << 31)) return hipErrorInvalidValue;          // frame numbers are ints
        hipError_t e = hipMemsetAsync(want_mp, 0, sizeof(int), stream);
        if (e != hipSuccess) return e;
    }
    const int bytes = in_kind == 0 ? 4 : (in_kind == 1 ? 2 : 4);
    const int vec_ok = (stride * bytes) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
    constexpr bool dual = DSP_CKPT_DUAL != 0;
#define DSP_CKPT_LAUNCH(E, D, I)                                                                                                          \
    hipLaunchKernelGGL((iir2_ckpt_kernel<E, D, I>), dim3(blocks), dim3(D ? 128 : 192), 0, stream, x, n_clips, n, stride, c_bp, c_mp, ck_bp, ck_mp, \
                       means_mp, want_mp, tables, vec_ok, simd_load, blocks / n_cu)
    const bool even = even_taps_only(c_mp);
    if (spans) {                                                     // ragged batches: the literal tables' even numerators, three-wave form
#define DSP_CKPT_LAUNCH_RAGGED(I)                                                                                                         \
    hipLaunchKernelGGL((iir2_ckpt_kernel<true, false, I, true>), dim3(blocks), dim3(192), 0, stream, x, n_clips, n, stride, c_bp, c_mp, ck_bp, ck_mp, \
                       means_mp, want_mp, tables, 1, simd_load, blocks / n_cu, spans, total)
        if (!even || in_kind < 0 || in_kind > 3) return hipErrorInvalidValue;
        if (in_kind == 0) DSP_CKPT_LAUNCH_RAGGED(0);
        else if (in_kind == 1) DSP_CKPT_LAUNCH_RAGGED(1);
        else if (in_kind == 2) DSP_CKPT_LAUNCH_RAGGED(2);
        else DSP_CKPT_LAUNCH_RAGGED(3);
#undef DSP_CKPT_LAUNCH_RAGGED
        return hipGetLastError();
    }
    if (in_kind == 0) { if (even) DSP_CKPT_LAUNCH(true, dual, 0); else DSP_CKPT_LAUNCH(false, dual, 0); }
    else if (!even) return hipErrorInvalidValue;                     // int16 input: the literal tables' even numerators only
    else if (in_kind == 1) DSP_CKPT_LAUNCH(true, false, 1);
    else if (in_kind == 2) DSP_CKPT_LAUNCH(true, false, 2);
    else if (in_kind == 3) DSP_CKPT_LAUNCH(true, false, 3);
    else return hipErrorInvalidValue;
#undef DSP_CKPT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_iir_f64_on_f32(const float *x, long n_clips, int n, long stride, const IirCoefD &c, float *y,
                                 hipStream_t stream)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    const int blocks = (int)((n_clips + 63) / 64);
    hipLaunchKernelGGL((iir_kernel<double, IirCoefD, false, float>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, stride, c, y, c, y, nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_iir_f64(const double *x, long n_clips, int n, long stride, const IirCoefD &c, double *y,
                          hipStream_t stream)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    const int blocks = (int)((n_clips + 63) / 64);
    hipLaunchKernelGGL((iir_kernel<double, IirCoefD, false>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, stride, c, y, c, y, nullptr, nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// a10: spectrogram.  A wavefront takes 64 consecutive (clip, time bin) frames.
//   phase 1  lane f runs the reference's SEQUENTIAL fp32 sum of frame f (classifier.cpp:329-333):
//            the order of the 256 additions is part of the result, so it stays serial per frame
//            and the parallelism is across the 64 frames;
//   phase 2  frame by frame the whole wave runs the 256-point FFT: lane g owns 4 points and does
//            two radix-2 levels of PlainFFT.cpp:50-84 per LDS round trip (4 round trips).  Every
//            butterfly is the reference's expression with the reference's twiddle value, and the
//            butterflies of one level are independent, so the results are bit-identical to the
//            serial loop whatever the lane split.
// ---------------------------------------------------------------------------------
#ifndef DSP_SPEC_TILE
#define DSP_SPEC_TILE 16
#endif
// frames whose PSD columns are collected in LDS before they are stored as row segments (SPEC_TILE floats each)
constexpr int SPEC_TILE = DSP_SPEC_TILE;
__device__ __forceinline__ unsigned bitrev6(unsigned v) { return __brev(v) >> 26; }

struct cpx { float x, y; };

// LDS position of FFT point i (float2 units).  The four level pairs touch i = base + j 4^p with the
// lane number spread over the other base-4 digits; XOR-ing 5 d2 and 16 d3 (d2, d3 = base-4 digits 2 and 3
// of i) into the low five bits makes every ds_write_b64 (16-lane groups) and ds_read_b64 (32-lane
// groups) of all four patterns bank-conflict free (exhaustive search, tools/emulate_wave_fft.py style).
__device__ __forceinline__ int spec_swz(int i)
{
    const int d2 = (i >> 4) & 3, d3 = (i >> 6) & 3;
    return (i & ~31) | ((i ^ (5 * d2) ^ (16 * d3)) & 31);
}

// PlainFFT.cpp:66-76: t = u * b; b = a - t; a = a + t
__device__ __forceinline__ void butterfly(cpx &a, cpx &b, const cpx u)
{
    const float t1 = u.x * b.x - u.y * b.y;
    const float t2 = u.x * b.y + u.y * b.x;
    b.x = a.x - t1;
    b.y = a.y - t2;
    a.x = a.x + t1;
    a.y = a.y + t2;
}

// hits (optional): work list of the clips whose map is wanted (hits[0] = count, then clip numbers, as written by
// classify_midpoints_kernel): frame slot s of the launch is time bin s % T of clip hits[1 + s / T], so the wanted clips
// are packed into the first wavefronts whatever their position in the batch, and the others exit at once.
// OUT selects what leaves the kernel:
//   SPEC_BIN_MAJOR    the reference's [bin][time] map (the spectrogram entry point)
//   SPEC_FRAME_MAJOR  [time][bin]: one frame = 129 consecutive floats, stored straight from the registers (the band-pass
//                     map inside classify(); classify_bands_kernel reads that layout)
//   SPEC_FLAGS        no map at all: one int per frame, 1 when any of its 129 cells is >= SpecTables::mp_keep_min.  That
//                     is all find_midpoints takes from the 1000-3000 Hz map (classifier.cpp:457-518), so classify() never
//                     writes that map to HBM (36 KB per clip written and read back otherwise)
enum { SPEC_BIN_MAJOR = 0, SPEC_FRAME_MAJOR = 1, SPEC_FLAGS = 2 };
template <int OUT>
__global__ __launch_bounds__(256) void spectrogram_kernel(const float *__restrict__ y, long n_clips, int n, long stride,
                                                          const SpecTables *__restrict__ tab, float *__restrict__ sxx, int T,
                                                          const float *__restrict__ means, const int *__restrict__ hits,
                                                          const int *__restrict__ gate)
{
    __shared__ float2 lds[4][kSpecSeg];
    // PSD columns of SPEC_TILE consecutive frames are collected here and stored as row segments: the output is
    // [bin][time], one frame is a COLUMN of it (129 scattered dwords if stored directly)
    constexpr bool TILED = OUT == SPEC_BIN_MAJOR;
    __shared__ float psd_tile[TILED ? 4 : 1][TILED ? kSpecBins * (SPEC_TILE + 1) : 1];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *buf = lds[wib];
    float *tile = psd_tile[TILED ? wib : 0];
    const long total = (hits ? (long)hits[0] : n_clips) * T;
    const long gid0 = ((long)blockIdx.x * 4 + wib) * 64;
    if (gid0 >= total) return;
    auto clip_of = [&](long slot) { return hits ? (long)hits[1 + slot] : slot; };
    // SPEC_FLAGS with a gate (written by the IIR kernel, [clip][T]): 0 = the frame's energy proves that no cell can reach
    // the threshold (max_k PSD[k] <= 2 |X[k]|^2 / U <= 512 sum v_n^2 / U, Parseval), so its flag is 0 without a transform
    unsigned long long todo = ~0ull;
    if (OUT == SPEC_FLAGS && gate) {
        const bool maybe = gid0 + lane < total && gate[gid0 + lane] != 0;
        todo = __ballot(maybe);
        if (todo == 0) {
            if (gid0 + lane < total) reinterpret_cast<int *>(sxx)[gid0 + lane] = 0;
            return;
        }
    }

    // ---- phase 1: sequential mean of this lane's frame
    float mean = 0.0f;
    {
        const long gid = gid0 + lane;
        if (gid < total) {
            const long slot = gid / T;
            const int t = (int)(gid - slot * T);
            const long clip = clip_of(slot);
            if (means) {              // already summed, in the same order, by the IIR kernel's lanes
                mean = means[clip * T + t];
            } else {
                const float *seg = y + clip * stride + (long)t * kSpecHop;
                float sum = 0.0f;
                for (int i = 0; i < kSpecSeg; ++i) sum = sum + seg[i];
                mean = sum / (float)kSpecSeg;
            }
        }
    }

    // ---- per-lane constants of phase 2
    // level pair (l, l+1), l1 = 2^l: lane g owns i_j = base + j l1, base = ((g >> l) << (l + 2)) | (g & (l1 - 1));
    // level l uses tw[l1 - 1 + m] for both its butterflies, level l+1 tw[2 l1 - 1 + m] and tw[2 l1 - 1 + m + l1], m = g & (l1 - 1)
    int pos[4][4];
    cpx ua[4], ub0[4], ub1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int l = 2 * p, l1 = 1 << l, m = lane & (l1 - 1);
        const int base = ((lane >> l) << (l + 2)) | m;
#pragma unroll
        for (int j = 0; j < 4; ++j) pos[p][j] = spec_swz(base + (j << l));
        ua[p] = {tab->tw_re[l1 - 1 + m], tab->tw_im[l1 - 1 + m]};
        ub0[p] = {tab->tw_re[2 * l1 - 1 + m], tab->tw_im[2 * l1 - 1 + m]};
        ub1[p] = {tab->tw_re[2 * l1 - 1 + m + l1], tab->tw_im[2 * l1 - 1 + m + l1]};
    }
    // first pair works on bit-reversed positions 4g + j, i.e. on samples 64 bitrev2(j) + bitrev6(g) (PlainFFT.cpp:33-47)
    int src[4];
    float win[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        src[j] = 64 * (((j & 1) << 1) | (j >> 1)) + (int)bitrev6(lane);
        win[j] = tab->window[src[j]];
    }
    const float U = tab->U;
    const bool trivial01 = tab->trivial_first_levels != 0;
    const float keep_min = tab->mp_keep_min;
    int flag = 0;

    // ---- phase 2
    const int n_here = (int)(total - gid0 < 64 ? total - gid0 : 64);
    long slot = gid0 / T;
    int t = (int)(gid0 - slot * T);
    long clip = clip_of(slot);
    // the four samples of the NEXT frame are requested before the current one is transformed: a frame is ~0.4 us of
    // arithmetic behind ~1.5 us of load latency otherwise (three wavefronts per SIMD do not cover that)
    float raw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool have_raw = false;                                  // raw holds the samples of the frame about to be transformed
    for (int f = 0; f < n_here; ++f) {
        if (OUT == SPEC_FLAGS && !((todo >> f) & 1)) {      // gated out: flag 0
            have_raw = false;
            if (++t == T) { t = 0; ++slot; if (f + 1 < n_here) clip = clip_of(slot); }
            continue;
        }
        const float mean_f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), f));
        if (!have_raw) {
            const float *seg = y + clip * stride + (long)t * kSpecHop;
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = seg[src[j]];
        }
        float cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = raw[j];
        have_raw = f + 1 < n_here && ((todo >> (f + 1)) & 1);
        if (have_raw) {
            const bool wrap = t + 1 == T;
            const long nclip = wrap ? clip_of(slot + 1) : clip;
            const float *seg = y + nclip * stride + (long)(wrap ? 0 : t + 1) * kSpecHop;
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = seg[src[j]];
        }
        cpx v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = {(cur[j] - mean_f) * win[j], 0.0f};          // classifier.cpp:336-346
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float2 q = buf[pos[p][j]]; v[j] = {q.x, q.y}; }
            }
            if (p == 0 && trivial01) {
                // Levels 0 and 1 on real input with twiddles (1,0), (1,0), (0,-1): t = u * b is (b.x, 0) resp. (0, -b.x),
                // so the reference's ten operations per butterfly leave these sums and differences and zeros.  Every
                // non-zero value is the reference's bit for bit (finite input); a zero may differ in sign, which
                // re^2 + im^2 cannot see.
                const float a0 = v[0].x + v[1].x, a1 = v[0].x - v[1].x, a2 = v[2].x + v[3].x, a3 = v[2].x - v[3].x;
                v[0] = {a0 + a2, 0.0f};
                v[2] = {a0 - a2, 0.0f};
                v[1] = {a1, -a3};
                v[3] = {a1, a3};
            } else {
                butterfly(v[0], v[1], ua[p]);
                butterfly(v[2], v[3], ua[p]);
                butterfly(v[0], v[2], ub0[p]);
                butterfly(v[1], v[3], ub1[p]);
            }
            if (p < 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) buf[pos[p][j]] = make_float2(v[j].x, v[j].y);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        // lane g now holds X[g], X[g + 64], X[g + 128], X[g + 192]; classifier.cpp:350-365
        float p0 = (v[0].x * v[0].x + v[0].y * v[0].y) / U;
        if (lane >= 1) p0 = p0 * 2.0f;
        float p1 = (v[1].x * v[1].x + v[1].y * v[1].y) / U;
        p1 = p1 * 2.0f;
        const float p2 = (v[2].x * v[2].x + v[2].y * v[2].y) / U;      // bin 128, lane 0 only
        if (OUT == SPEC_FLAGS) {
            const bool loud = p0 >= keep_min || p1 >= keep_min || (lane == 0 && p2 >= keep_min);
            if (__ballot(loud) != 0 && lane == f) flag = 1;             // lane f keeps the flag of frame f
        } else if (OUT == SPEC_FRAME_MAJOR) {
            float *out = sxx + (clip * T + t) * (long)kSpecBins;
            out[lane] = p0;
            out[lane + 64] = p1;
            if (lane == 0) out[128] = p2;
        } else {
            const int col = f & (SPEC_TILE - 1);
            tile[lane * (SPEC_TILE + 1) + col] = p0;
            tile[(lane + 64) * (SPEC_TILE + 1) + col] = p1;
            if (lane == 0) tile[128 * (SPEC_TILE + 1) + col] = p2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((f & (SPEC_TILE - 1)) == SPEC_TILE - 1 || f == n_here - 1) {
                // flush frames [f & ~(SPEC_TILE - 1), f]: lane -> column lane % SPEC_TILE (one frame, one division), rows lane / SPEC_TILE + k 64 / SPEC_TILE
                const int col2 = lane & (SPEC_TILE - 1), ff = (f & ~(SPEC_TILE - 1)) + col2;
                if (ff <= f) {
                    const long g = gid0 + ff;
                    const long sl = g / T;
                    const int tt = (int)(g - sl * T);
                    float *out = sxx + clip_of(sl) * (long)kSpecBins * T + tt;
                    for (int row = lane / SPEC_TILE; row < kSpecBins; row += 64 / SPEC_TILE) out[(long)row * T] = tile[row * (SPEC_TILE + 1) + col2];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (++t == T) { t = 0; ++slot; if (f + 1 < n_here) clip = clip_of(slot); }
    }
    if (OUT == SPEC_FLAGS && lane < n_here) reinterpret_cast<int *>(sxx)[gid0 + lane] = flag;      // [clip][T] (no work list here)
}

// ---------------------------------------------------------------------------------
// a10 from checkpoints: the spectrogram of segments that were never written to HBM.  A 512-thread block takes 60 frame
// slots ((clip, time bin) pairs); for the wanted ones it
//   L  loads the segments' 256 input samples into its LDS rows (all waves, coalesced 1 KB rows);
//   R  recomputes the recurrence v over the segment's 256 samples, lane per frame (wave 0), from the delay line
//      iir2_ckpt_kernel stored at the segment start: classifier.cpp:199-205 on the same x with the same state;
//   T  applies the output taps y[n] = b0 v[n] + sum b[j] v[n-j] (classifier.cpp:207-216): no feedback, so the eight waves
//      take an eighth of every segment each, and y overwrites v in place;
//   M  (3000-7500 Hz map only) sums each segment in order for its mean (classifier.cpp:329-333), lane per frame;
//   F  runs the 256-point PlainFFT of each wanted frame, a wave per frame, 7 or 8 frames per wave, exactly as
//      spectrogram_kernel does, reading the samples from the block's LDS rows.
// OUT = SPEC_FLAGS: slot = clip * T + t over all clips, wanted = gate != 0, output one flag per slot (0 for the others);
// OUT = SPEC_FRAME_MAJOR: slots walk the work list `hits`, every frame is wanted, output [time][bin] PSD rows.
// ---------------------------------------------------------------------------------
// 512 threads and 60 frame slots per block: 60 rows + 8 FFT buffers = 80.5 KB, so TWO blocks of eight waves fit a CU's 160 KB
// (64 slots would be 84.7 KB: one block per CU).  The recompute is lane-per-frame on one wave (~9 us whatever the lane
// count), so what a CU delivers is frames in flight / block latency: 2 x 60 frames over (1 + 9 + 1 + 1 + 3.4) us instead of
// 2 x 64 over (2 + 9 + 2 + 1 + 7) us with four waves per block.
constexpr int RC_THREADS = 512, RC_WAVES = RC_THREADS / 64, RC_FRAMES = 60;
static_assert((kSpecSeg / RC_WAVES) % IIR_BURST == 0 && RC_FRAMES <= 64, "taps split evenly over the waves; a frame per lane");
constexpr int RC_ROW = 8 + kSpecSeg + 1;      // v[-8..-1] | 256 samples | pad: odd stride, lane l <-> row l is conflict free

// p / U in three instructions (SpecTables::rU, div_fast): q = p rU, then one Newton step on the exact residual
__device__ __forceinline__ float div_by_u_fast(float p, float U, float rU)
{
    const float q = p * rU;
    return fmaf(fmaf(-q, U, p), rU, q);
}

__global__ __launch_bounds__(256) void spec_div_verify_kernel(const SpecTables *__restrict__ tab, unsigned long long *__restrict__ mismatches)
{
    const float U = tab->U, rU = tab->rU;
    const unsigned lo = __float_as_uint(kDivFastLo), hi = __float_as_uint(kDivFastHi);
    unsigned long long bad = 0;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float p = __uint_as_float((unsigned)b);
        const float exact = p / U;
        bad += __float_as_uint(div_by_u_fast(p, U, rU)) != __float_as_uint(exact);
    }
    if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_spec_div_verify(const SpecTables *tables, unsigned long long *mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(spec_div_verify_kernel, dim3(4096), dim3(256), 0, stream, tables, mismatches);
    return hipGetLastError();
}

struct SpecLane {                             // per-lane constants of the 256-point FFT (see spectrogram_kernel)
    int pos[4][4];
    cpx ua[4], ub0[4], ub1[4];
    int src[4];
    float win[4];
    __device__ __forceinline__ void init(int lane, const SpecTables *__restrict__ tab)
    {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int l = 2 * p, l1 = 1 << l, m = lane & (l1 - 1);
            const int base = ((lane >> l) << (l + 2)) | m;
#pragma unroll
            for (int j = 0; j < 4; ++j) pos[p][j] = spec_swz(base + (j << l));
            ua[p] = {tab->tw_re[l1 - 1 + m], tab->tw_im[l1 - 1 + m]};
            ub0[p] = {tab->tw_re[2 * l1 - 1 + m], tab->tw_im[2 * l1 - 1 + m]};
            ub1[p] = {tab->tw_re[2 * l1 - 1 + m + l1], tab->tw_im[2 * l1 - 1 + m + l1]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            src[j] = 64 * (((j & 1) << 1) | (j >> 1)) + (int)bitrev6(lane);
            win[j] = tab->window[src[j]];
        }
    }
    // cur[j] = sample src[j] of the frame; returns the PSD cells lane, lane + 64 and (lane 0) 128
    __device__ __forceinline__ void psd(const float (&cur)[4], float mean_f, float U, bool trivial01, float2 *buf, int lane,
                                        float &p0, float &p1, float &p2, float rU = 0.0f, bool div_fast = false) const
    {
        cpx v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = {(cur[j] - mean_f) * win[j], 0.0f};          // classifier.cpp:336-346
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float2 q = buf[pos[p][j]]; v[j] = {q.x, q.y}; }
            }
            if (p == 0 && trivial01) {
                const float a0 = v[0].x + v[1].x, a1 = v[0].x - v[1].x, a2 = v[2].x + v[3].x, a3 = v[2].x - v[3].x;
                v[0] = {a0 + a2, 0.0f};
                v[2] = {a0 - a2, 0.0f};
                v[1] = {a1, -a3};
                v[3] = {a1, a3};
            } else {
                butterfly(v[0], v[1], ua[p]);
                butterfly(v[2], v[3], ua[p]);
                butterfly(v[0], v[2], ub0[p]);
                butterfly(v[1], v[3], ub1[p]);
            }
            if (p < 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) buf[pos[p][j]] = make_float2(v[j].x, v[j].y);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        const float s0 = v[0].x * v[0].x + v[0].y * v[0].y, s1 = v[1].x * v[1].x + v[1].y * v[1].y, s2 = v[2].x * v[2].x + v[2].y * v[2].y;
        // classifier.cpp:350-365.  A frame whose every cell lies in the verified range takes the three-instruction division (the
        // same bits, SpecTables::div_fast); silent, denormal, huge or non-finite cells send the whole frame through the real one.
        const bool in_range = fminf(fminf(s0, s1), s2) >= kDivFastLo && fmaxf(fmaxf(s0, s1), s2) <= kDivFastHi;
        if (div_fast && __ballot(!in_range) == 0) {
            p0 = div_by_u_fast(s0, U, rU);
            p1 = div_by_u_fast(s1, U, rU);
            p2 = div_by_u_fast(s2, U, rU);
        } else {
            p0 = s0 / U;
            p1 = s1 / U;
            p2 = s2 / U;
        }
        if (lane >= 1) p0 = p0 * 2.0f;
        p1 = p1 * 2.0f;
        // the next frame's first butterfly level writes buf: order it behind this frame's last reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// Diagnostic build only (-DDSP_RC_STAMPS, never shipped): s_memtime at the phase boundaries of the first 2048 blocks, written to a
// buffer of their own that no kernel reads (cdna_hip_programming.md 7, in-kernel stamps); tools/rc_stamps.py prints the medians.
#ifdef DSP_RC_STAMPS
__device__ unsigned long long g_rc_stamps[8 * 2048];
#define RC_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_rc_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_bd_stamps[8 * 2048];
#define BD_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_bd_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
hipError_t read_bd_stamps(unsigned long long *host, int count)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bd_stamps), sizeof(unsigned long long) * (size_t)count);
}
hipError_t read_rc_stamps(unsigned long long *host, int count)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rc_stamps), sizeof(unsigned long long) * (size_t)count);
}
#else
#define RC_STAMP(k) do { } while (0)
#define BD_STAMP(k) do { } while (0)
#endif

// (wave-wide minimum / maximum of a float, every lane gets it)
__device__ __forceinline__ float wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_maxf(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Round 3: the block is PERSISTENT (grid = the blocks the chip holds; a block walks groups g, g + gridDim.x, ... of 60 frame slots) and
// software-pipelined over its groups: while group g runs R .. F, the 60 KB of x rows of group g + 1, their restart states and (flags)
// their segment means are already in flight into registers (8 float4 per thread), and the (clip, time bin) description of group g + 2
// is being read from the work list -- phase L of a group is then only the LDS stores (in-kernel stamps of round 2: L took 9.6 k of a
// block's 56 k cycles, plus the exposed latency of the restart-state loads inside R).  With two restart states per segment (kCkPerSegBp) phase R runs as two
// 128-sample chains on waves 0 and 1 (restart states at the segment start and at its middle) instead of one 256-sample chain.
template <int OUT, bool EVEN_B, int IN = 0>
__global__ __launch_bounds__(RC_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void spec_from_ckpt_kernel(const void *__restrict__ xv, long n_clips, int n, long stride, const IirCoef c,
                                                             const float *__restrict__ ck, const float *__restrict__ means,
                                                             const int *__restrict__ wantlist, const int *__restrict__ hits,
                                                             const SpecTables *__restrict__ tab, float *__restrict__ out, int T, int vec_ok,
                                                             const int *__restrict__ need, unsigned *__restrict__ minmax,
                                                             const ClipSpan *__restrict__ spans = nullptr)
{
    // spans (ragged batches): clip c starts at spans[c].off instead of c * stride; T stays the workspaces' row length.
    // SPEC_FRAME_MAJOR with need / minmax (classify): a row is stored only when need[clip][t] says a band window of one of the clip's
    // midpoints covers time bin t (classify_midpoints_kernel), and the smallest / largest positive cell of every transformed frame
    // goes into minmax[clip][2] (float bits, atomicMin / atomicMax; reset by classify_midpoints_kernel) -- what the band kernel's dB
    // normalisation needs of the rows it no longer reads.
    static_assert(OUT == SPEC_FLAGS || OUT == SPEC_FRAME_MAJOR, "flags or [time][bin]");
    constexpr int kCkPerSeg = OUT == SPEC_FLAGS ? kCkPerSegMp : kCkPerSegBp;      // restart states per segment in `ck`
    __shared__ float rows[RC_FRAMES * RC_ROW];
    __shared__ float2 fftbuf[RC_WAVES][kSpecSeg];
    __shared__ float smean[64];
    __shared__ int sflag[64];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // frame slots of this launch: SPEC_FLAGS walks the work list of gated-in frames (wantlist[0] = count, then clip * T + t),
    // SPEC_FRAME_MAJOR every time bin of the clips on `hits` (hits[0] = count, then clip numbers)
    const long total = OUT == SPEC_FLAGS ? (long)wantlist[0] : (long)hits[0] * T;
    const long n_groups = (total + RC_FRAMES - 1) / RC_FRAMES;
    long g = blockIdx.x;
    if (g >= n_groups) return;
    RC_STAMP(6);

    // every wave's lane l describes frame slot 60 g + l of group g: (clip, time bin); clip < 0 = no such slot
    auto describe = [&](long grp, int &dclip, int &dt) {
        const long gid = grp * RC_FRAMES + lane;
        dclip = -1; dt = 0;
        if (grp < n_groups && lane < RC_FRAMES && gid < total) {
            if (OUT == SPEC_FLAGS) {
                const int fr = wantlist[1 + gid];
                dclip = fr / T;
                dt = fr - dclip * T;
            } else {
                const long slotc = gid / T;
                dt = (int)(gid - slotc * T);
                dclip = hits[1 + slotc];
                if (spans && dt >= spans[dclip].frames) dclip = -1;      // ragged batches: past the clip's last segment
            }
        }
    };
    // the loads of a group: its x rows (one wave-instruction = one row = 1 KB contiguous; every load of the block in flight
    // together), the restart states (waves 0 .. kCkPerSeg - 1, lane per frame) and the segment means (flags; wave 0)
    // (int16 input: a lane's four samples are 8 bytes of a mono row, 16 of a stereo one; converted when they reach the LDS rows)
    constexpr int PER_ROW = kSpecSeg / 4, NL = (RC_FRAMES * PER_ROW + RC_THREADS - 1) / RC_THREADS;
    static_assert(PER_ROW == 64, "one wave-instruction loads one row");
    cls_u4 v4[NL];
    float4 ckv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float mean_pre = 0.0f;
    int need_pre = 1;
    // role: this wave's part of the serial phases (0 .. kCkPerSeg - 1: that part of R; 0 also M).  (Rotating the parts over the
    // waves from group to group, so that both resident blocks of a CU do not keep R on the same SIMDs, was measured: no change.)
    const int role = wib;
    auto issue = [&](int iclip, int it) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = threadIdx.x + RC_THREADS * i, r = e / PER_ROW, c4 = (e % PER_ROW) * 4;      // r is wave-uniform
            if (r >= RC_FRAMES) { v4[i] = cls_u4{0u, 0u, 0u, 0u}; continue; }
            const int rclip = __builtin_amdgcn_readlane(iclip, r);
            const long rc0 = rclip < 0 ? 0 : rclip;                                                                          // unwanted slots: clip 0, t 0
            const long s0 = (spans ? (rclip < 0 ? 0 : spans[rc0].off) : rc0 * stride) + (long)__builtin_amdgcn_readlane(it, r) * kSpecHop + c4;
            const unsigned char *xs = reinterpret_cast<const unsigned char *>(xv) + s0 * ClsIn<IN>::kBytes;
            if (IN == 1) {
                if (vec_ok) { const uint2 q = *reinterpret_cast<const uint2 *>(xs); v4[i] = cls_u4{q.x, q.y, 0u, 0u}; }
                else { const unsigned short *h = reinterpret_cast<const unsigned short *>(xs); v4[i] = cls_u4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), 0u, 0u}; }
            } else if (vec_ok) v4[i] = *reinterpret_cast<const cls_u4 *>(xs);
            else { const unsigned *w = reinterpret_cast<const unsigned *>(xs); v4[i] = cls_u4{w[0], w[1], w[2], w[3]}; }
        }
        if (role < kCkPerSeg && iclip >= 0) {
            const float4 *src = reinterpret_cast<const float4 *>(ck + (((long)iclip * T + it) * kCkPerSeg + role) * 8);
            ckv[0] = src[0]; ckv[1] = src[1];
        }
        if (means && role == 0 && iclip >= 0) mean_pre = means[(long)iclip * T + it];
        if (OUT == SPEC_FRAME_MAJOR && need && iclip >= 0) need_pre = need[(long)iclip * T + it];
    };

    int clip1, t1, clip2, t2;                       // descriptions of the next group and the one after it
    describe(g, clip1, t1);
    issue(clip1, t1);
    describe(g + gridDim.x, clip2, t2);

    const float U = tab->U, keep_min = tab->mp_keep_min, rU = tab->rU;
    const bool trivial01 = tab->trivial_first_levels != 0, div_fast = tab->div_fast != 0;
    float2 *buf = fftbuf[wib];
    float *row = rows + (lane < RC_FRAMES ? lane : 0) * RC_ROW;

    for (; g < n_groups; g += gridDim.x) {
        const int clip = clip1, t = t1;
        const bool want = clip >= 0;
        const unsigned long long todo = __ballot(want);
        RC_STAMP(0);
        // ---- L: the group's x rows from the registers into the LDS rows; its restart state and mean out of the prefetch registers
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = threadIdx.x + RC_THREADS * i, r = e / PER_ROW, c4 = (e % PER_ROW) * 4;
            if (r >= RC_FRAMES) continue;
            float *dst = rows + r * RC_ROW + 8 + c4;
            if (IN == 1) {
                dst[0] = (float)(int)(short)(v4[i][0] & 0xffffu) * (1.0f / 32768.0f); dst[1] = (float)((int)v4[i][0] >> 16) * (1.0f / 32768.0f);
                dst[2] = (float)(int)(short)(v4[i][1] & 0xffffu) * (1.0f / 32768.0f); dst[3] = (float)((int)v4[i][1] >> 16) * (1.0f / 32768.0f);
            } else {
                float o[4];
                cls_piece_to_float<IN == 1 ? 0 : IN>(v4[i], o);
                dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
            }
        }
        float d[8] = {ckv[0].x, ckv[0].y, ckv[0].z, ckv[0].w, ckv[1].x, ckv[1].y, ckv[1].z, ckv[1].w};
        const float mean_ck = mean_pre;
        (void)mean_ck;
        const unsigned long long store_mask = __ballot(want && need_pre != 0);      // frames whose rows the band sums will read
        (void)store_mask;
        if (wib == 0) sflag[lane] = 0;
        __syncthreads();
        // the next group's loads go out now and land while this group computes; the description after that is requested too
        clip1 = clip2; t1 = t2;
        if (g + gridDim.x < n_groups) issue(clip1, t1);
        describe(g + 2 * (long)gridDim.x, clip2, t2);
        RC_STAMP(1);

        // ---- R: v over the segment, from the stored delay line(s); v replaces x in the row
        if (role < kCkPerSeg && want) {
            constexpr int RC_RQ = kSpecSeg / kCkPerSeg;
            const int h0 = RC_RQ * role;
            if (role == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) row[7 - j] = d[j];                   // row[8 + m] = v[m], m = -8 .. 255
            }
#pragma unroll 1
            for (int h = h0; h < h0 + RC_RQ; h += IIR_BURST) {
                float xr[IIR_BURST], vr[IIR_BURST];
#pragma unroll
                for (int i = 0; i < IIR_BURST; ++i) xr[i] = row[8 + h + i];
#pragma unroll
                for (int i = 0; i < IIR_BURST; ++i) {                       // classifier.cpp:199-205
                    float v = xr[i];
#pragma unroll
                    for (int j = 1; j <= 8; ++j) v = v - c.a[j] * d[j - 1];
#pragma unroll
                    for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                    d[0] = v;
                    vr[i] = v;
                }
#pragma unroll
                for (int i = 0; i < IIR_BURST; ++i) row[8 + h + i] = vr[i];
            }
        }
        __syncthreads();
        RC_STAMP(2);

        // ---- T: output taps, wave w takes samples [RC_TQ w, RC_TQ (w + 1)) of every wanted frame; y replaces v
        {
            constexpr int RC_TQ = kSpecSeg / RC_WAVES;
            const int n0 = RC_TQ * wib;
            float dt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dt[j] = want ? row[8 + n0 - 1 - j] : 0.0f;       // v[n0-1] .. v[n0-8]
            __syncthreads();                                                         // every history is read before any y lands
            if (want) {
#pragma unroll 1
                for (int h = 0; h < RC_TQ; h += IIR_BURST) {
                    float vr[IIR_BURST], orr[IIR_BURST];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) vr[i] = row[8 + n0 + h + i];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) {                           // classifier.cpp:207-216
                        float o = c.b[0] * vr[i];
#pragma unroll
                        for (int j = 1; j <= 8; ++j)
                            if (!EVEN_B || j % 2 == 0) o = o + c.b[j] * dt[j - 1];      // EVEN_B: see iir2_ckpt_kernel
#pragma unroll
                        for (int j = 7; j > 0; --j) dt[j] = dt[j - 1];
                        dt[0] = vr[i];
                        orr[i] = o;
                    }
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) row[8 + n0 + h + i] = orr[i];
                }
            }
        }
        __syncthreads();
        RC_STAMP(3);
        // the FFT's 48 per-lane constants are formed per group from an opaque copy of the lane number: held across the loop they
        // (and the 32 prefetch registers) spilled 19 VGPRs; their L2-resident loads are in flight during M
        SpecLane K;
        {
            int lane_k = lane;
            asm volatile("" : "+v"(lane_k));
            K.init(lane_k, tab);
        }

        // ---- M: segment means (sequential sums, classifier.cpp:329-333)
        if (role == 0 && want) {
            float mean;
            if (means) {
                mean = mean_ck;                                  // summed in the same order by iir2_ckpt_kernel's taps wave
            } else {
                float sum = 0.0f;
#pragma unroll 1
                for (int h = 0; h < kSpecSeg; h += IIR_BURST) {
                    float yr[IIR_BURST];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) yr[i] = row[8 + h + i];
#pragma unroll
                    for (int i = 0; i < IIR_BURST; ++i) sum = sum + yr[i];
                }
                mean = sum / (float)kSpecSeg;
            }
            smean[lane] = mean;
        }
        __syncthreads();
        RC_STAMP(4);

        // ---- F: one frame at a time per wave
        // (minmax: a lane keeps the running minimum / maximum of the positive cells it has seen of the current clip; a wave reduction
        // and two atomics when the clip changes or the group ends)
        unsigned run_mn = 0xFFFFFFFFu;                      // (bits of the smallest positive cell) - 1
        float run_mx = 0.0f;
        int run_clip = -1;
        auto flush_minmax = [&] {
            if (run_clip < 0) return;
            unsigned a = run_mn;
            for (int o = 32; o > 0; o >>= 1) a = min(a, (unsigned)__shfl_xor((int)a, o));
            const float b = wave_maxf(run_mx);
            if (lane == 0) {
                if (a != 0xFFFFFFFFu) atomicMin(minmax + 2 * (long)run_clip, a + 1u);
                atomicMax(minmax + 2 * (long)run_clip + 1, __float_as_uint(b));
            }
        };
        (void)run_mn; (void)run_mx; (void)run_clip; (void)flush_minmax;
#pragma unroll 1
        for (int f = wib; f < RC_FRAMES; f += RC_WAVES) {
            if (!((todo >> f) & 1)) continue;
            const float *fr = rows + f * RC_ROW + 8;
            float cur[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = fr[K.src[j]];
            float p0, p1, p2;
            K.psd(cur, smean[f], U, trivial01, buf, lane, p0, p1, p2, rU, div_fast);
            if (OUT == SPEC_FLAGS) {
                const bool loud = p0 >= keep_min || p1 >= keep_min || (lane == 0 && p2 >= keep_min);
                if (__ballot(loud) != 0 && lane == 0) sflag[f] = 1;
            } else {
                const long fclip = (long)__builtin_amdgcn_readlane(clip, f);
                const int ft = __builtin_amdgcn_readlane(t, f);
                if (minmax) {
                    if ((int)fclip != run_clip) { flush_minmax(); run_clip = (int)fclip; run_mn = 0xFFFFFFFFu; run_mx = 0.0f; }
                    // cells are >= 0: as unsigned words minus one, the positive ones keep their order and a zero becomes the largest
                    const float q2 = lane == 0 ? p2 : p0;          // bin 128 exists on lane 0 only
                    run_mn = min(run_mn, min(min(__float_as_uint(p0) - 1u, __float_as_uint(p1) - 1u), __float_as_uint(q2) - 1u));
                    run_mx = fmaxf(run_mx, fmaxf(fmaxf(p0, p1), q2));
                }
                if ((store_mask >> f) & 1) {
                    float *o = out + (fclip * T + ft) * (long)kSpecBins;
                    o[lane] = p0;
                    o[lane + 64] = p1;
                    if (lane == 0) o[128] = p2;
                }
            }
        }
        if (OUT == SPEC_FRAME_MAJOR && minmax) { flush_minmax(); run_clip = -1; }
#ifdef DSP_RC_STAMPS
        __syncthreads();
        RC_STAMP(5);
#endif
        if (OUT == SPEC_FLAGS) {
            __syncthreads();
            if (wib == 0 && want) reinterpret_cast<int *>(out)[(long)clip * T + t] = sflag[lane];      // the rest of loud[] was zeroed by the launcher
        }
        __syncthreads();                                     // rows, smean and sflag are reused by the next group
    }
    RC_STAMP(7);
}

// resident 512-thread blocks of the chip for one instantiation (queried once per process AND device: a process may drive several GPUs)
template <int OUT, bool EVEN_B>
static int rc_resident_blocks()
{
    static int cache[64] = {0};
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= 64) cur = 0;
    int &cached = cache[cur];
    if (cached > 0) return cached;
    int per_cu = 0, dev = 0, n_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spec_from_ckpt_kernel<OUT, EVEN_B>, RC_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
    if (const char *e = std::getenv("DSP_AMD_RC_BLOCKS_PER_CU")) { if (std::atoi(e) > 0) per_cu = std::atoi(e); }      // A/B runs
    if (std::getenv("DSP_AMD_DEBUG")) fprintf(stderr, "[dsp_amd] spec_from_ckpt_kernel<%d,%d>: %d resident blocks per CU x %d CUs\n", OUT, (int)EVEN_B, per_cu, n_cu);
    cached = per_cu * n_cu;
    return cached;
}

hipError_t launch_spec_from_ckpt(const void *x, long n_clips, int n, long stride, const IirCoef &c, const float *ck, const float *means,
                                 const int *wantlist, const int *hits, const SpecTables *tables, float *out, bool flags, hipStream_t stream,
                                 const int *need, unsigned *minmax, int in_kind, const ClipSpan *spans)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    if (n_clips >= (1L << 31)) return hipErrorInvalidValue;      // clip numbers travel as ints
    const long total = n_clips * T;                       // the bound: the work lists' counts are read on the device
    // (ragged batches: rows start anywhere; the 16-byte loads go out unaligned, which the hardware serves)
    const int vec_ok = spans ? 1 : (stride % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0);
    const long groups = (total + RC_FRAMES - 1) / RC_FRAMES;
    auto launch = [&](auto kernel, int resident, const int *wl, const int *hl) {
        const dim3 grid((unsigned)std::min<long>(groups, resident));
        hipLaunchKernelGGL(kernel, grid, dim3(RC_THREADS), 0, stream, x, n_clips, n, stride, c, ck, means, wl, hl, tables, out, T, vec_ok, need, minmax, spans);
    };
    const bool even = even_taps_only(c);
    if (in_kind != 0 && (!even || in_kind < 0 || in_kind > 3)) return hipErrorInvalidValue;      // int16 input: the literal tables' even numerators only
    if (flags) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)total * sizeof(int), stream);      // frames not on the list are not loud
        if (e != hipSuccess) return e;
        const int res = even ? rc_resident_blocks<SPEC_FLAGS, true>() : rc_resident_blocks<SPEC_FLAGS, false>();
        if (in_kind == 1) launch(spec_from_ckpt_kernel<SPEC_FLAGS, true, 1>, res, wantlist, (const int *)nullptr);
        else if (in_kind == 2) launch(spec_from_ckpt_kernel<SPEC_FLAGS, true, 2>, res, wantlist, (const int *)nullptr);
        else if (in_kind == 3) launch(spec_from_ckpt_kernel<SPEC_FLAGS, true, 3>, res, wantlist, (const int *)nullptr);
        else if (even) launch(spec_from_ckpt_kernel<SPEC_FLAGS, true>, res, wantlist, (const int *)nullptr);
        else launch(spec_from_ckpt_kernel<SPEC_FLAGS, false>, res, wantlist, (const int *)nullptr);
    } else {
        const int res = even ? rc_resident_blocks<SPEC_FRAME_MAJOR, true>() : rc_resident_blocks<SPEC_FRAME_MAJOR, false>();
        if (in_kind == 1) launch(spec_from_ckpt_kernel<SPEC_FRAME_MAJOR, true, 1>, res, (const int *)nullptr, hits);
        else if (in_kind == 2) launch(spec_from_ckpt_kernel<SPEC_FRAME_MAJOR, true, 2>, res, (const int *)nullptr, hits);
        else if (in_kind == 3) launch(spec_from_ckpt_kernel<SPEC_FRAME_MAJOR, true, 3>, res, (const int *)nullptr, hits);
        else if (even) launch(spec_from_ckpt_kernel<SPEC_FRAME_MAJOR, true>, res, (const int *)nullptr, hits);
        else launch(spec_from_ckpt_kernel<SPEC_FRAME_MAJOR, false>, res, (const int *)nullptr, hits);
    }
    return hipGetLastError();
}

hipError_t launch_iir2_f64(const double *x, long n_clips, int n, long stride, long ystride, const IirCoefD &c1, double *y1,
                           const IirCoefD &c2, double *y2, hipStream_t stream)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    const int blocks = (int)((n_clips + 63) / 64);
    auto even = [](const IirCoefD &c) { return c.b[1] == 0.0 && c.b[3] == 0.0 && c.b[5] == 0.0 && c.b[7] == 0.0; };
    if (even(c1) && even(c2))                 // a Butterworth band-pass numerator is (1 - z^-2)^4 scaled: the designs of dsp_butter_bandpass have exact zeros there
        hipLaunchKernelGGL((iir_kernel<double, IirCoefD, true, double, false, true>), dim3(blocks), dim3(128), 0, stream, x, n_clips, n, stride, ystride, c1, y1, c2, y2, nullptr, nullptr);
    else
        hipLaunchKernelGGL((iir_kernel<double, IirCoefD, true>), dim3(blocks), dim3(128), 0, stream, x, n_clips, n, stride, ystride, c1, y1, c2, y2, nullptr, nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// a10 in float64: compute_spectrogram of donut-classifier/classifier.c:448-592 (the file north_star names for the IIR keeps
// its whole pipeline in double and transforms with FFTW's r2c).  FFTW is unvendored, so there is no operation order to replay:
// a block per frame evaluates the 129 bins of the 256-point DFT directly in float64 (129 x 256 multiply-adds, twiddles from
// sincospi) -- the same mathematical transform, checked by tolerance against scipy.signal.spectrogram and the reference-held dump _blobtimes.txt (tests).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spectrogram_f64_kernel(const double *__restrict__ y, long n_clips, int n, long stride, int fs,
                                                              double *__restrict__ sxx, int T)
{
    __shared__ double seg[kSpecSeg], cs[kSpecSeg], sn[kSpecSeg], red[kSpecSeg];
    const int tid = threadIdx.x;
    const long clip = blockIdx.x / T;
    const int t = (int)(blockIdx.x - clip * T);
    if (clip >= n_clips) return;
    const double x = y[clip * stride + (long)t * kSpecHop + tid];
    double s_, c_;
    sincospi(2.0 * (double)tid / (double)kSpecSeg, &s_, &c_);
    cs[tid] = c_; sn[tid] = s_;
    // periodic Tukey(0.25): the 257-point symmetric formula without its last point (classifier.c:484-521)
    const double alpha = 0.25, M = (double)(kSpecSeg + 1), pi = 3.14159265358979323846;
    const int width = (int)floor(alpha * (M - 1.0) / 2.0);
    double w;
    if (tid <= width) w = 0.5 * (1.0 + cos(pi * (-1.0 + 2.0 * tid / (alpha * (M - 1.0)))));
    else if (tid <= (int)(M - width - 2)) w = 1.0;
    else w = 0.5 * (1.0 + cos(pi * (-2.0 / alpha + 1.0 + 2.0 * tid / (alpha * (M - 1.0)))));
    auto block_sum = [&](double v) {
        red[tid] = v;
        __syncthreads();
        for (int o = kSpecSeg / 2; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    const double mean = block_sum(x) / (double)kSpecSeg;               // classifier.c:551-561 detrend
    const double U = block_sum(w * w) * (double)fs;                     // :524-530
    seg[tid] = (x - mean) * w;
    __syncthreads();
    if (tid < kSpecBins) {
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < kSpecSeg; ++i) {
            const int ph = (tid * i) & (kSpecSeg - 1);
            sr += seg[i] * cs[ph];
            si -= seg[i] * sn[ph];
        }
        double p = (sr * sr + si * si) / U;                             // :574-586
        if (tid >= 1 && tid < kSpecBins - 1) p *= 2.0;
        sxx[(clip * kSpecBins + tid) * (long)T + t] = p;
    }
}

hipError_t launch_spectrogram_f64(const double *y, long n_clips, int n, long stride, int fs, double *sxx, hipStream_t stream)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    if (n_clips * T >= (1L << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(spectrogram_f64_kernel, dim3((unsigned)(n_clips * T)), dim3(256), 0, stream, y, n_clips, n, stride, fs, sxx, T);
    return hipGetLastError();
}

hipError_t launch_spectrogram_f32(const float *y, long n_clips, int n, long stride, const SpecTables *tables,
                                  float *sxx, hipStream_t stream, const float *means, const int *hits, bool frame_major)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    const long total = n_clips * T;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (frame_major)
        hipLaunchKernelGGL(spectrogram_kernel<SPEC_FRAME_MAJOR>, grid, dim3(256), 0, stream, y, n_clips, n, stride, tables, sxx, T, means, hits, (const int *)nullptr);
    else
        hipLaunchKernelGGL(spectrogram_kernel<SPEC_BIN_MAJOR>, grid, dim3(256), 0, stream, y, n_clips, n, stride, tables, sxx, T, means, hits, (const int *)nullptr);
    return hipGetLastError();
}

hipError_t launch_spectrogram_flags(const float *y, long n_clips, int n, long stride, const SpecTables *tables, int *flags,
                                    hipStream_t stream, const float *means, const int *gate)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    const long total = n_clips * T;
    hipLaunchKernelGGL(spectrogram_kernel<SPEC_FLAGS>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y, n_clips, n, stride, tables,
                       reinterpret_cast<float *>(flags), T, means, (const int *)nullptr, gate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// a11: midpoints, dB map, normalisation band, three band sums, rule (classify_midpoints_kernel, classify_bands_kernel).
// Element-wise steps and min/max (order-independent, exact) run across lanes; the order-dependent float sums keep the
// reference's order: cluster means on one lane, sum_intense as a chain of v_readlane adds over the packed kept cells.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float to_db(float s)
{
    // classifier.cpp:43: 10 * log10(s / 1e-12) evaluated in double, stored as float
    return (float)(10 * log10((double)s / 1e-12));
}

// sum_intense for a whole wavefront: the cells are fetched 64 at a time in the reference's (row, column) order, then
// added ONE BY ONE in that order (v_readlane + add: the float sum's order is part of the result).  NaN cells add 0.0f,
// which leaves a float sum unchanged bit for bit, exactly like the reference's skip.  Returns the same value in every lane.
// The reference's index searches ("while (i < n && key(i) < lo) ++i" and its mirror image) for a whole wavefront: 64
// candidates per step, the exit index is the first (last) lane whose test fails -- the same comparisons on the same
// floats, without a float division per loop trip on a single lane.  All 64 lanes must be active.
template <class Pred>
__device__ __forceinline__ int wave_first_false(int count, Pred pred)
{
    const int lane = threadIdx.x & 63;
    for (int b = 0; b < count; b += 64) {
        const int i = b + lane;
        const unsigned long long m = __ballot(i < count && !pred(i));
        if (m) return b + __ffsll((long long)m) - 1;
    }
    return count;
}
template <class Pred>
__device__ __forceinline__ int wave_last_false(int count, Pred pred)
{
    const int lane = threadIdx.x & 63;
    for (int b = ((count - 1) / 64) * 64; b >= 0; b -= 64) {
        const int i = b + lane;
        const unsigned long long m = __ballot(i < count && !pred(i));
        if (m) return b + 63 - __clzll((long long)m);
    }
    return -1;
}

// sum_intense's time-bin search (classifier.cpp:395-413) for a whole wavefront: the columns [t0, t1] of the window midpoint +-
// half_range.  Used by the band sums AND by the midpoints kernel, which marks the columns whose rows have to be stored at all.
__device__ __forceinline__ void intense_time_window(float half_range, int fs, int T, float midpoint, int &t0, int &t1)
{
    auto time = [&](int t) { return ((float)(t * kSpecHop + kSpecSeg / 2)) / (float)fs; };
    const float t_lo = midpoint - half_range, t_hi = midpoint + half_range;
    t0 = wave_first_false(T, [&](int t) { return time(t) < t_lo; });
    t1 = wave_last_false(T, [&](int t) { return time(t) > t_hi; });
    if (t0 >= T) t0 = T - 1;
    if (t1 < 0) t1 = 0;
    if (t0 > t1) { int x = t0; t0 = t1; t1 = x; }
}

__device__ float sum_intense_wave(float lower, float upper, float half_range, int fs, int T, const float *db, float midpoint,
                                  float *scratch /* LDS, 128 floats of this wavefront */)
{
    auto freq = [&](int k) { return (float)k * (float)fs / (float)kSpecSeg; };
    int f0 = wave_first_false(kSpecBins, [&](int k) { return freq(k) < lower; });
    int f1 = wave_last_false(kSpecBins, [&](int k) { return freq(k) > upper; });
    if (f0 >= kSpecBins) f0 = kSpecBins - 1;
    if (f1 < 0) f1 = 0;
    if (f0 > f1) { int x = f0; f0 = f1; f1 = x; }
    int t0, t1;
    intense_time_window(half_range, fs, T, midpoint, t0, t1);
    const int lane = threadIdx.x & 63;
    const int W = t1 - t0 + 1, N = (f1 - f0 + 1) * W;
    float total = 0.0f;
    // 64 adds in lane order, total = ((total + a_0) + a_1) + ... + a_63, as a systolic chain over the lanes: lane 0 starts with
    // total + a_0, and 63 times every lane adds its own a to the value of the lane before it (ONE instruction per add:
    // v_add_f32_dpp wave_shr:1) -- after step k lane k holds the partial sum through a_k, the same additions in the same order as
    // the reference's loop.  (64 x v_readlane + v_add took three to five instructions per add: the unrolled readlanes were hoisted
    // into 64 SGPRs, 134 of which hipcc spilled back into VGPR lanes.)
    auto chain = [&](float a) {
        float s = lane == 0 ? total + a : a;
#pragma unroll
        for (int k = 1; k < 64; ++k)
            s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x138, 0xF, 0xF, true)) + a;      // wave_shr:1; lane 0 reads 0
        total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
    };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Only 7-18 % of the cells of a map are kept; the others are NaN and the reference skips them.  The kept cells are
    // packed, still in (row, column) order, into the wavefront's scratch (ballot + mbcnt ranks) and the add chain runs
    // once per 64 KEPT cells instead of once per 64 cells.  Padding lanes hold 0.0f, and x + 0.0f == x bit for bit.
    int pend = 0;                                            // wave-uniform: cells waiting in scratch
    for (int e0 = 0; e0 < N; e0 += 64) {
        const int e = e0 + lane;
        float v = 0.0f;
        if (e < N) {
            const int r = e / W, cidx = e - r * W;
            v = db[(long)(t0 + cidx) * kSpecBins + f0 + r];       // the map is [time][bin]
            if (isnan(v)) v = 0.0f;
        }
        const bool kept = v != 0.0f;                         // kept cells lie in (0.65, 0.80)
        const unsigned long long m = __ballot(kept);
        if (m == 0) continue;
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kept) scratch[pend + rank] = v;
        pend += __popcll(m);
        if (pend >= 64) {
            wave_sync();
            const float a = scratch[lane];
            const float rest = lane < pend - 64 ? scratch[64 + lane] : 0.0f;
            wave_sync();
            scratch[lane] = rest;
            chain(a);
            pend -= 64;
        }
    }
    if (pend > 0) {
        wave_sync();
        chain(lane < pend ? scratch[lane] : 0.0f);
    }
    return total;
}

// One 256-thread block per clip.  The band-pass dB map lives in LDS when it fits (129 x T <= kTailLdsCells,
// always for 1 s clips), so the order-dependent sums of thread 0 read LDS instead of HBM.  The midpoint map
// is never formed: "10 log10(s / 1e-12) > 70 dB" is monotone in s, so its pass compares s with the smallest
// float that passes (SpecTables::mp_keep_min, found once on the device with the same to_db) and raises one
// flag per time bin -- half of the path's float64 log10 evaluations disappear.
constexpr int kTailLdsCells = 129 * 72;

__global__ void spec_threshold_kernel(SpecTables *tab, float thr_db)
{
    // smallest positive float s with to_db(s) > thr_db: bisection on the bit pattern (positive floats order like ints;
    // to_db is monotone).  Thresholds outside (to_db(FLT_MIN), to_db(FLT_MAX)) = (-259, 505) dB saturate.
    unsigned lo = 0x00800000u, hi = 0x7F7FFFFFu;      // to_db(lo) <= thr < to_db(hi)
    if (to_db(__uint_as_float(lo)) > thr_db) hi = lo;
    while (hi - lo > 1) {
        const unsigned mid = lo + (hi - lo) / 2;
        if (to_db(__uint_as_float(mid)) > thr_db) hi = mid; else lo = mid;
    }
    tab->mp_keep_min = to_db(__uint_as_float(hi)) > thr_db ? __uint_as_float(hi) : INFINITY;
}

hipError_t launch_spec_threshold(SpecTables *tables, float threshold_db, hipStream_t stream)
{
    hipLaunchKernelGGL(spec_threshold_kernel, dim3(1), dim3(1), 0, stream, tables, threshold_db);
    return hipGetLastError();
}

// classify() after the spectrograms, as two kernels so that the band-pass spectrogram can be skipped for clips
// without midpoints:
//   classify_midpoints_kernel  (midpoint spectrogram only)  time bins above 70 dB -> blob times -> greedy clusters ->
//                              midpoints, written to the clip's ClassifyTrace record; label 0 when there are none
//   classify_bands_kernel      (band-pass spectrogram, clips with midpoints only)  dB map, normalisation, the three
//                              band sums per midpoint in the reference's order, the rule
constexpr int kMidClipsPerWave = 4;       // clips one wavefront walks: one atomic on the work-list counter per 16 clips, not per clip
__global__ __launch_bounds__(256) void classify_midpoints_kernel(int *loud, long n_clips, const int T, int fs,
                                                                 int *__restrict__ labels, ClassifyTrace *__restrict__ trace,
                                                                 int *__restrict__ hits, int full_records, unsigned *__restrict__ minmax,
                                                                 const ClipSpan *__restrict__ spans = nullptr)
{
    // spans (ragged batches): clip c has spans[c].frames time bins (the head of its row of T_row); everything below sees that count.
    const int T_row = T;
    // minmax != nullptr (classify): once a clip's flags are read, its row of loud[] is rewritten as need[t] = "time bin t lies in a
    // band window (+- 0.18 s or +- 0.05 s, sum_intense's own search) of one of the clip's midpoints" -- the rows the band sums will
    // read and the only ones spec_from_ckpt_kernel<[time][bin]> stores -- and minmax[clip] is reset for that kernel's atomics.
    __shared__ float mids_lds[4][kMaxMidpoints];
    __shared__ int win_lo[4][2 * kMaxMidpoints], win_hi[4][2 * kMaxMidpoints];
    // a wavefront per clip, kMidClipsPerWave clips in turn; loud[clip][T] are the time bins with a cell above the threshold
    // (spec_from_ckpt_kernel<SPEC_FLAGS>).  Clips with midpoints go on the band kernels' work list: collected per block and
    // appended with ONE atomic (12 288 single-word atomics from as many wavefronts took 0.14 ms: the word saturates at
    // ~88 adds per microsecond).
    __shared__ float blob_all[4][1024];
    __shared__ int hit_list[4][kMidClipsPerWave];
    __shared__ int hit_count[4], hit_base;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *blob = blob_all[wib];
    int n_hits = 0;                                         // wave-uniform
    const long first = ((long)blockIdx.x * 4 + wib) * kMidClipsPerWave;
    for (int ci = 0; ci < kMidClipsPerWave; ++ci) {
        const long clip = first + ci;
        if (clip >= n_clips) break;
        const int T = spans ? spans[clip].frames : T_row;
        // blob times of the flagged bins, in order (ballot ranks), then the greedy clustering on one lane
        int nb = 0;
        for (int j0 = 0; j0 < T; j0 += 64) {
            const int j = j0 + lane;
            const bool flag = j < T && loud[clip * T_row + j] != 0;
            const unsigned long long m = __ballot(flag);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (flag) blob[nb + rank] = ((float)(j * kSpecHop + kSpecSeg / 2)) / (float)fs;
            nb += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int count = 0;
        if (lane == 0) {
            // greedy clustering, classifier.cpp:522-574
            const float tol = 0.05f, min_dur = 0.15f;
            int i0 = 0;
            while (i0 < nb) {
                int i1 = i0;
                while (i1 + 1 < nb && (blob[i1 + 1] - blob[i1]) <= tol) ++i1;
                const float dur = blob[i1] - blob[i0];
                if (dur >= min_dur) {
                    float sm = 0.0f;
                    for (int k = i0; k <= i1; ++k) sm = sm + blob[k];
                    if (count < kMaxMidpoints) { trace[clip].midpoints[count] = sm / (float)(i1 - i0 + 1); mids_lds[wib][count] = sm / (float)(i1 - i0 + 1); }
                    ++count;
                }
                i0 = i1 + 1;
            }
            if (count > kMaxMidpoints) count = kMaxMidpoints;
            trace[clip].n_midpoints = count;
            if (count == 0) labels[clip] = 0;               // classifier.cpp:93-114: no midpoint can fire the rule
            else hit_list[wib][n_hits] = (int)clip;
        }
        count = __builtin_amdgcn_readfirstlane(count);
        n_hits += count > 0 ? 1 : 0;
        if (minmax && count > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // mids_lds written by lane 0
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k = 0; k < count; ++k) {
                const float mid = mids_lds[wib][k];
                int a0, a1, b0, b1;
                intense_time_window(0.18f, fs, T, mid, a0, a1);        // the 5000-7000 Hz and 500-2500 Hz bands
                intense_time_window(0.05f, fs, T, mid, b0, b1);        // the 2500-5000 Hz band
                if (lane == 0) { win_lo[wib][2 * k] = a0; win_hi[wib][2 * k] = a1; win_lo[wib][2 * k + 1] = b0; win_hi[wib][2 * k + 1] = b1; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int j0 = 0; j0 < T; j0 += 64) {
                const int j = j0 + lane;
                int nd = 0;
                for (int w = 0; w < 2 * count; ++w) nd |= (j >= win_lo[wib][w] && j <= win_hi[wib][w]) ? 1 : 0;
                if (j < T) loud[clip * T_row + j] = nd;
            }
            if (lane == 0) { minmax[2 * clip] = 0x7F800000u; minmax[2 * clip + 1] = 0u; }      // + inf, 0: nothing seen yet
        }
        if (full_records) {
            // a whole record per clip: unused midpoints and the band sums the rule never reaches (no midpoints, or after the
            // first hit) read as 0.  midpoints[64] and sums[64][3] are 256 consecutive floats.
            static_assert(kMaxMidpoints == 64 && sizeof(ClassifyTrace) == 4 + 4 * 256, "record layout");
            float *rec = trace[clip].midpoints;
            for (int i = lane; i < 4 * kMaxMidpoints; i += 64)
                if (i >= count) rec[i] = 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next clip reuses blob
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) hit_count[wib] = n_hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = hit_count[0] + hit_count[1] + hit_count[2] + hit_count[3];
        hit_base = total ? atomicAdd(hits, total) : 0;      // work list of the band kernels, any order
    }
    __syncthreads();
    int off = hit_base;
    for (int w = 0; w < wib; ++w) off += hit_count[w];
    if (lane < n_hits) hits[1 + off + lane] = hit_list[wib][lane];
}

// USE_LDS: the map fits the LDS budget (129 x T <= kTailLdsCells): the PSD cells of the clip are read ONCE into
// registers (kTailPerThread per thread) and the map is written to LDS; otherwise the map is rebuilt in place in HBM.
constexpr int kTailPerThread = (kTailLdsCells + 255) / 256;

// need / minmax (both or neither): only the rows with need[clip][t] != 0 exist in sxx_bp (the others were never stored and are never
// read: no band window covers them), and the smallest / largest positive cell of the WHOLE map comes from minmax[clip] (float bits).
template <bool USE_LDS>
__device__ __forceinline__ void classify_bands_clip(float *__restrict__ sxx_bp, long clip, int T, int fs, int *__restrict__ labels,
                                                    ClassifyTrace *__restrict__ trace, float *map_lds, const ClassifyRule &rule,
                                                    bool partial, const int (&need_regs)[USE_LDS ? 1 : 4], const unsigned (&minmax_regs)[2], int T_row)
{
    // T_row: time bins per clip in sxx_bp's layout (ragged batches: the longest clip's; this clip's map is the first T of them)
    // partial: need_regs = this thread's words of need[clip][] (time bins tid + 256 k) and minmax_regs = minmax[clip][], fetched by
    // the caller while the previous clip was being worked on (read here they cost each clip two exposed round trips to memory)
    __shared__ float red_mn[4], red_mx[4];
    __shared__ float mids[kMaxMidpoints];
    __shared__ unsigned char need_lds[1024];
    if (partial) {
#pragma unroll
        for (int k = 0; k < (USE_LDS ? 1 : 4); ++k)          // USE_LDS: 129 T <= kTailLdsCells, T < 256
            if (threadIdx.x + 256 * k < T) need_lds[threadIdx.x + 256 * k] = need_regs[k] != 0;
        __syncthreads();
    }
    auto row_exists = [&](int idx) { return !partial || need_lds[idx / kSpecBins] != 0; };
    const int n_mids = trace[clip].n_midpoints;
    const int tid = threadIdx.x;
    const int cells = kSpecBins * T;
    float *bp_g = sxx_bp + clip * (long)kSpecBins * T_row;
    if (tid < n_mids) mids[tid] = trace[clip].midpoints[tid];
    BD_STAMP(0);

    // ---- band-pass map: dB, clip min/max, normalise, keep (0.65, 0.80)  classifier.cpp:35-80
    // to_db is monotone in the PSD value s, so the clip's dB minimum / maximum are to_db of the smallest / largest
    // positive s (two float64 log10 instead of one per cell), and only cells whose s lies in the (0.65, 0.80) band
    // widened by a safety margin can survive the reference's test: those get the exact dB value and the reference's
    // own float comparison, all others are NaN in the reference as well.
    float smn = INFINITY, smx = -INFINITY;
    if (USE_LDS && partial) {
        // only the rows a band window covers exist: a wave takes rows wave, wave + 4, ... (whether a row exists is wave-uniform: a
        // scalar branch, no per-cell predicate), eight rows' loads in flight; rows that do not exist read as 0 = "no candidate"
        const int wv = tid >> 6, ln = tid & 63;
        constexpr int RB = 8;
        for (int r0 = wv; r0 < T; r0 += 4 * RB) {
            float c0[RB], c1[RB], c2[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = r0 + 4 * u;
                const bool ex = r < T && need_lds[r < T ? r : 0] != 0;
                c0[u] = c1[u] = c2[u] = 0.0f;
                if (ex) {
                    const float *row = bp_g + (long)r * kSpecBins;
                    c0[u] = row[ln]; c1[u] = row[64 + ln];
                    if (ln == 0) c2[u] = row[128];
                }
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = r0 + 4 * u;
                if (r < T) {
                    float *row = map_lds + r * kSpecBins;
                    row[ln] = c0[u]; row[64 + ln] = c1[u];
                    if (ln == 0) row[128] = c2[u];
                }
            }
        }
        // (smn / smx come from minmax below)
    } else if (USE_LDS) {
        // all loads of the thread in flight together, then the raw PSD values wait in the LDS map for their second pass
        float cell[kTailPerThread];
#pragma unroll
        for (int k = 0; k < kTailPerThread; ++k) cell[k] = tid + 256 * k < cells ? bp_g[tid + 256 * k] : 0.0f;
#pragma unroll
        for (int k = 0; k < kTailPerThread; ++k) {
            if (cell[k] > 0) { smn = fminf(smn, cell[k]); smx = fmaxf(smx, cell[k]); }
            if (tid + 256 * k < cells) map_lds[tid + 256 * k] = cell[k];
        }
    } else {
        for (int i = tid; i < cells; i += 256) {
            const float sv = row_exists(i) ? bp_g[i] : 0.0f;
            if (sv > 0) { smn = fminf(smn, sv); smx = fmaxf(smx, sv); }
        }
    }
    smn = wave_min(smn);
    smx = wave_maxf(smx);
    if ((tid & 63) == 0) { red_mn[tid >> 6] = smn; red_mx[tid >> 6] = smx; }
    __syncthreads();
    BD_STAMP(1);
    smn = fminf(fminf(red_mn[0], red_mn[1]), fminf(red_mn[2], red_mn[3]));      // min / max are order-independent
    smx = fmaxf(fmaxf(red_mx[0], red_mx[1]), fmaxf(red_mx[2], red_mx[3]));
    if (partial) {                                                              // of the whole map, rows that were not stored included
        smn = __uint_as_float(minmax_regs[0]);
        const float whole_mx = __uint_as_float(minmax_regs[1]);
        smx = whole_mx > 0 ? whole_mx : -INFINITY;
    }
    // the reference starts its running min / max from +-DBL_MAX stored in floats = +-inf
    const float mn = smn <= smx ? to_db(smn) : INFINITY, mx = smn <= smx ? to_db(smx) : -INFINITY;
    const float lo_thr = rule.keep_lo, hi_thr = rule.keep_hi;
    const double range = (double)mx - (double)mn, slack = 1e-4 * range + 1e-4;       // dB; float rounding of v is ~1e-6 range
    const float s_lo = (float)(1e-12 * pow(10.0, ((double)mn + (double)lo_thr * range - slack) / 10.0) * (1.0 - 1e-6));
    const float s_hi = (float)(1e-12 * pow(10.0, ((double)mn + (double)hi_thr * range + slack) / 10.0) * (1.0 + 1e-6));
    auto keep = [&](float sv) {
        float v = NAN;
        if (sv >= s_lo && sv <= s_hi) {                 // (false for s <= 0 and NaN)
            v = (to_db(sv) - mn) / (mx - mn);
            v = (v > lo_thr && v < hi_thr) ? v : NAN;
        }
        return v;
    };
    __shared__ float2 pend_buf[4][64];                       // per wavefront: (cell index, s) waiting for its log10
    if (USE_LDS) {
        // Cells outside the widened band are NaN at once.  The others (7-18 % of a map) are queued per wavefront and
        // evaluated up to 64 at a time, so a float64 log10 is paid per ~64 candidates and not per 64 cells.
        float2 *pb = pend_buf[tid >> 6];
        const int lane = tid & 63;
        auto wave_sync = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        auto drain = [&](int count) {
            wave_sync();
            if (lane < count) { const float2 e = pb[lane]; map_lds[__float_as_int(e.x)] = keep(e.y); }
            wave_sync();
        };
        int pend = 0;                                        // wave-uniform
        // a rolled loop: the float64 log10 of keep() is inlined once, not once per cell of the thread (the unrolled form
        // was 9 000 instructions, more than the instruction cache holds)
#pragma unroll 1
        for (int k = 0; k < kTailPerThread; ++k) {
            const int idx = tid + 256 * k;
            const float sv = idx < cells ? map_lds[idx] : 0.0f;       // this thread's own store above
            const bool cand = idx < cells && sv >= s_lo && sv <= s_hi;
            if (idx < cells && !cand) map_lds[idx] = NAN;
            const unsigned long long m = __ballot(cand);
            if (m == 0) continue;
            const int add = __popcll(m);
            if (pend + add > 64) { drain(pend); pend = 0; }
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (cand) pb[pend + rank] = make_float2(__int_as_float(idx), sv);
            pend += add;
        }
        drain(pend);
    } else {
        for (int i = tid; i < cells; i += 256)
            if (row_exists(i)) bp_g[i] = keep(bp_g[i]);
    }
    const float *bp = USE_LDS ? map_lds : bp_g;
    __syncthreads();
    BD_STAMP(2);
    // classifier.cpp:93-114: per midpoint the three band sums, one wavefront each (the fourth idles), then the rule
    __shared__ float band[3];
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *scratch = reinterpret_cast<float *>(pend_buf[wv]);       // free again after the barrier above
    int hit = 0;
    for (int k = 0; k < n_mids; ++k) {
        const float mid = mids[k];
        if (wv == 0) { const float v = sum_intense_wave(5000, 7000, 0.18f, fs, T, bp, mid, scratch); if ((tid & 63) == 0) band[0] = v; }
        if (wv == 1) { const float v = sum_intense_wave(2500, 5000, 0.05f, fs, T, bp, mid, scratch); if ((tid & 63) == 0) band[1] = v; }
        if (wv == 2) { const float v = sum_intense_wave(500, 2500, 0.18f, fs, T, bp, mid, scratch); if ((tid & 63) == 0) band[2] = v; }
        __syncthreads();
        const float above = band[0], middle = band[1], below = band[2];
        if (tid == 0) { trace[clip].sums[k][0] = above; trace[clip].sums[k][1] = middle; trace[clip].sums[k][2] = below; }
        hit = (middle < rule.middle_max && above > rule.above_min && below > rule.below_min) ? 1 : 0;
        __syncthreads();
        if (hit) break;                                      // uniform: every thread read the same three sums
    }
    BD_STAMP(3);
    if (tid == 0) labels[clip] = hit;
}

// Clips with midpoints come from the work list the midpoints kernel filled (hits[0] = count, then clip numbers): the
// blocks of a fixed grid walk it, so the load is even over the XCDs whatever the positions of those clips in the batch
// (workgroups go round-robin to the XCDs by block number: "one block per clip" left 6 of 8 XCDs idle for a batch with
// every fourth clip positive).
// (four waves per SIMD = four blocks per CU beside the 36 KB map: hipcc found 127 VGPRs by itself until the kernel grew, then 182 -- two
// blocks per CU and 625 us instead of 396)
template <bool USE_LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void classify_bands_kernel(float *__restrict__ sxx_bp, int T, int fs, int *__restrict__ labels,
                                                             ClassifyTrace *__restrict__ trace, const int *__restrict__ hits,
                                                             const ClassifyRule rule, const int *__restrict__ need, const unsigned *__restrict__ minmax,
                                                             const ClipSpan *__restrict__ spans = nullptr)
{
    extern __shared__ float map_lds[];
    const int count = hits[0];
    const bool partial = need != nullptr;
    constexpr int NK = USE_LDS ? 1 : 4;                      // words of need[clip][] per thread (T <= 1024; USE_LDS: T < 256)
    int nd[NK], nd_next[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { nd[k] = 1; nd_next[k] = 1; }
    unsigned mm2[2] = {0u, 0u}, mm2_next[2] = {0u, 0u};
    auto fetch = [&](int it, int (&n4)[NK], unsigned (&m2)[2]) {
        if (!partial || it >= count) return;
        const long c = hits[1 + it];
#pragma unroll
        for (int k = 0; k < NK; ++k) n4[k] = (int)threadIdx.x + 256 * k < T ? need[c * T + threadIdx.x + 256 * k] : 0;
        m2[0] = minmax[2 * c]; m2[1] = minmax[2 * c + 1];
    };
    fetch(blockIdx.x, nd_next, mm2_next);
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NK; ++k) nd[k] = nd_next[k];
        mm2[0] = mm2_next[0]; mm2[1] = mm2_next[1];
        fetch(it + gridDim.x, nd_next, mm2_next);            // the next clip's, in flight during this one
        const long bclip = hits[1 + it];
        classify_bands_clip<USE_LDS>(sxx_bp, bclip, spans ? spans[bclip].frames : T, fs, labels, trace, map_lds, rule, partial, nd, mm2, T);
        __syncthreads();                                     // the block's LDS is reused by the next clip
    }
}

hipError_t launch_classify_midpoints(int *loud, long n_clips, int n, int fs, int *labels, ClassifyTrace *trace, int *hits,
                                     hipStream_t stream, bool full_records, unsigned *minmax, const ClipSpan *spans)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > 1024 || !trace || !hits) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(hits, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(classify_midpoints_kernel, dim3((unsigned)((n_clips + 4 * kMidClipsPerWave - 1) / (4 * kMidClipsPerWave))), dim3(256), 0, stream, loud, n_clips, T, fs, labels, trace, hits, full_records ? 1 : 0, minmax, spans);
    return hipGetLastError();
}

hipError_t launch_classify_bands(float *sxx_bp, long n_clips, int n, int fs, int *labels, ClassifyTrace *trace, const int *hits,
                                 hipStream_t stream, const ClassifyRule &rule, const int *need, const unsigned *minmax, const ClipSpan *spans)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > 1024 || !trace || !hits || (need == nullptr) != (minmax == nullptr)) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)(n_clips < 2048 ? n_clips : 2048);      // 256 CUs x 4 resident blocks x 2
    if (kSpecBins * T <= kTailLdsCells)
        hipLaunchKernelGGL(classify_bands_kernel<true>, dim3(blocks), dim3(256), (size_t)kSpecBins * T * sizeof(float), stream, sxx_bp, T, fs,
                           labels, trace, hits, rule, need, minmax, spans);
    else
        hipLaunchKernelGGL(classify_bands_kernel<false>, dim3(blocks), dim3(256), 0, stream, sxx_bp, T, fs, labels, trace, hits, rule, need, minmax, spans);
    return hipGetLastError();
}

// Ragged batches run in order of length (a block's 64 clips alike); their per-clip results go home through the permutation:
// dst record perm[i] = src record i (records of rec_words 32-bit words).
__global__ __launch_bounds__(256) void scatter_records_kernel(const unsigned *__restrict__ src, const int *__restrict__ perm, long n, int rec_words,
                                                              unsigned *__restrict__ dst)
{
    const long total = n * rec_words;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / rec_words;
        dst[(long)perm[r] * rec_words + (i - r * rec_words)] = src[i];
    }
}

hipError_t launch_scatter_records(const void *src, const int *perm, long n, size_t rec_bytes, void *dst, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (rec_bytes % 4 != 0 || rec_bytes == 0) return hipErrorInvalidValue;
    const long total = n * (long)(rec_bytes / 4);
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(scatter_records_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const unsigned *>(src), perm, n, (int)(rec_bytes / 4),
                       static_cast<unsigned *>(dst));
    return hipGetLastError();
}

// sum_intense as its own entry point (classifier.h:17): arbitrary frequency / time axes and a [bin][time] map, as the
// reference's row pointers describe it.  One wavefront; cells are fetched 64 at a time in (row, column) order and added one by
// one in that order.  A NaN cell adds +0.0f instead of being skipped: the running sum starts at +0 and can never become -0, so
// x + 0.0f == x bit for bit.
__global__ __launch_bounds__(64) void sum_intense_kernel(float lower, float upper, float half_range, const float *__restrict__ freqs, int nf,
                                                         const float *__restrict__ times, int nt, const float *__restrict__ db,
                                                         float midpoint, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const float t_lo = midpoint - half_range, t_hi = midpoint + half_range;
    int f0 = wave_first_false(nf, [&](int k) { return freqs[k] < lower; });
    int f1 = wave_last_false(nf, [&](int k) { return freqs[k] > upper; });
    if (f0 >= nf) f0 = nf - 1;
    if (f1 < 0) f1 = 0;
    if (f0 > f1) { const int x = f0; f0 = f1; f1 = x; }
    int t0 = wave_first_false(nt, [&](int t) { return times[t] < t_lo; });
    int t1 = wave_last_false(nt, [&](int t) { return times[t] > t_hi; });
    if (t0 >= nt) t0 = nt - 1;
    if (t1 < 0) t1 = 0;
    if (t0 > t1) { const int x = t0; t0 = t1; t1 = x; }
    const long W = t1 - t0 + 1, N = (long)(f1 - f0 + 1) * W;
    float total = 0.0f;
    for (long e0 = 0; e0 < N; e0 += 64) {
        const long e = e0 + lane;
        float v = 0.0f;
        if (e < N) {
            const long r = e / W, c = e - r * W;
            v = db[(f0 + r) * (long)nt + t0 + c];
            if (isnan(v)) v = 0.0f;
        }
#pragma unroll
        for (int l = 0; l < 64; ++l) total = total + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    }
    if (lane == 0) *out = total;
}

hipError_t launch_sum_intense(float lower, float upper, float half_range, const float *freqs, int nf, const float *times, int nt,
                              const float *db, float midpoint, float *out, hipStream_t stream)
{
    if (nf <= 0 || nt <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sum_intense_kernel, dim3(1), dim3(64), 0, stream, lower, upper, half_range, freqs, nf, times, nt, db, midpoint, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// host tables
// ---------------------------------------------------------------------------------
void build_spec_tables(int fs, SpecTables &t)
{
    // periodic Tukey(0.25) through the reference's 257-point symmetric formula, float32
    // arithmetic and cosf as in classifier.cpp:259-293 (the one-past-the-end store is dropped)
    const float alpha = 0.25f, M = (float)(kSpecSeg + 1), pi_f = (float)3.14159265358979323846;
    const int width = (int)floorf(alpha * (M - 1.0f) / 2.0f);
    for (int n = 0; n < kSpecSeg; ++n) {
        if (n <= width)
            t.window[n] = 0.5f * (1.0f + cosf(pi_f * (-1.0f + 2.0f * (float)n / (alpha * (M - 1.0f)))));
        else if (n <= (int)(M - (float)width - 2.0f))
            t.window[n] = 1.0f;
        else
            t.window[n] = 0.5f * (1.0f + cosf(pi_f * (-2.0f / alpha + 1.0f + 2.0f * (float)n / (alpha * (M - 1.0f)))));
    }
    float U = 0.0f;
    for (int i = 0; i < kSpecSeg; ++i) U = U + t.window[i] * t.window[i];
    t.U = U * (float)fs;
    {   // the float nearest to 1 / U (long double: 64 mantissa bits, then the neighbours compared)
        const long double inv = 1.0L / (long double)t.U;
        float r = (float)inv;
        const float cand[3] = {std::nextafterf(r, 0.0f), r, std::nextafterf(r, INFINITY)};
        for (float cnd : cand) if (fabsl((long double)cnd - inv) < fabsl((long double)r - inv)) r = cnd;
        t.rU = r;
        t.div_fast = 0;       // set on the device table by the context after launch_spec_div_verify
    }
    // PlainFFT.cpp:52-84: per level the running (u1,u2) starts at (1,0) and is advanced by
    // (c1,c2); between levels c2 = -sqrt((1-c1)/2), c1 = sqrt((1+c1)/2) in double, stored float
    float c1 = -1.0f, c2 = 0.0f;
    int l1 = 1;
    for (int l = 0; l < 8; ++l) {
        float u1 = 1.0f, u2 = 0.0f;
        for (int m = 0; m < l1; ++m) {
            t.tw_re[l1 - 1 + m] = u1;
            t.tw_im[l1 - 1 + m] = u2;
            const float z = u1 * c1 - u2 * c2;
            u2 = u1 * c2 + u2 * c1;
            u1 = z;
        }
        c2 = (float)std::sqrt((1.0 - (double)c1) / 2.0);
        c2 = -c2;
        c1 = (float)std::sqrt((1.0 + (double)c1) / 2.0);
        l1 <<= 1;
    }
    // energy gate of the IIR kernel: the taper covers exactly the first and the last 32 samples of a segment
    t.gate_ok = 1;
    for (int i = kSpecSeg - kSpecHop; i < kSpecHop; ++i) t.gate_ok &= t.window[i] == 1.0f;
    float c = 0.0f;
    for (int i = 0; i < kSpecSeg; ++i) c += t.window[i] * t.window[i];
    for (int i = 0; i < kSpecSeg - kSpecHop; ++i) {
        t.win2_in[i] = t.window[i] * t.window[i];
        t.win2_out[i] = t.window[kSpecHop + i] * t.window[kSpecHop + i];
    }
    t.win2_sum = c * 1.0001f;
    t.gate_scale = 2.0f * (float)kSpecSeg / t.U * 1.01f;
    // the spectrogram kernel's short form of levels 0 and 1 (real input) is valid for exactly these twiddles
    t.trivial_first_levels = t.tw_re[0] == 1.0f && t.tw_im[0] == 0.0f && t.tw_re[1] == 1.0f && t.tw_im[1] == 0.0f &&
                             t.tw_re[2] == 0.0f && t.tw_im[2] == -1.0f;
}

}  // namespace dsp
