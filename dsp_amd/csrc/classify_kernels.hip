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
#include <hip/hip_runtime.h>

#include <cmath>

#include "classify_kernels.hpp"

#pragma clang fp contract(off)

namespace dsp {

// ---------------------------------------------------------------------------------
// a9: direct form II, one lane per clip, time tiles transposed through LDS so HBM sees
// coalesced 128-byte rows although each lane walks its own clip.
// ---------------------------------------------------------------------------------
constexpr int IIR_TS = 16;       // samples per tile: small tiles keep LDS per block low -> more resident waves to
                                 // hide the serial recurrence's latency (measured 3x over 32-sample tiles at fp64)
constexpr int IIR_LD = IIR_TS + 1;   // +1 word: lane l reads column i of row l -> banks (17 l + i) % 32 distinct

template <typename T>
struct IirState { T d[8]; };

template <typename T, typename C>
__device__ __forceinline__ T iir_step(IirState<T> &s, const C &c, T x)
{
    // classifier.cpp:199-216: v = x - sum_{j=1..8} a[j] d[j-1] (left to right), y = b0 v + sum b[j] d[j-1]
    T v = x;
#pragma unroll
    for (int j = 1; j <= 8; ++j) v = v - c.a[j] * s.d[j - 1];
    T y = c.b[0] * v;
#pragma unroll
    for (int j = 1; j <= 8; ++j) y = y + c.b[j] * s.d[j - 1];
#pragma unroll
    for (int j = 7; j > 0; --j) s.d[j] = s.d[j - 1];
    s.d[0] = v;
    return y;
}

// T: arithmetic type of the recurrence; TIO: element type in HBM (float rows filtered in double
// are converted on load and rounded once on store: BASELINE config 3's per-frame prefilter).
template <typename T, typename C, bool TWO, typename TIO = T>
__global__ __launch_bounds__(64) void iir_kernel(const TIO *__restrict__ x, long n_clips, int n, long stride,
                                                 const C c1, TIO *__restrict__ y1, const C c2, TIO *__restrict__ y2)
{
    __shared__ TIO tin[64 * IIR_LD];
    __shared__ TIO tout1[64 * IIR_LD];
    __shared__ TIO tout2[TWO ? 64 * IIR_LD : 1];
    const int lane = threadIdx.x;
    const long clip0 = (long)blockIdx.x * 64;
    const int rows = (int)((n_clips - clip0) < 64 ? (n_clips - clip0) : 64);
    IirState<T> s1, s2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1.d[j] = T(0); s2.d[j] = T(0); }
    // 16-byte vector path needs every row start and every tile start 16-byte aligned
    const bool vec_ok = (stride * sizeof(TIO)) % 16 == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0 &&
                        (reinterpret_cast<uintptr_t>(y1) % 16) == 0 && (!TWO || (reinterpret_cast<uintptr_t>(y2) % 16) == 0);
    for (int t0 = 0; t0 < n; t0 += IIR_TS) {
        const int cols = n - t0 < IIR_TS ? n - t0 : IIR_TS;
        // tile load: 64 rows x IIR_TS columns.  Fast path: 16-byte vectors (4 lanes cover one
        // 64-byte row segment); otherwise element-wise, lane -> (row = e / TS, col = e % TS)
        const bool vec = vec_ok && cols == IIR_TS;
        if (vec) {
            constexpr int PER = 16 / sizeof(TIO), CH = IIR_TS / PER;      // elements per vector, vectors per row
            for (int e = lane; e < 64 * CH; e += 64) {
                const int r = e / CH, c = (e % CH) * PER;
                TIO tmp[PER];
                if (r < rows) {
                    const float4 q = *reinterpret_cast<const float4 *>(x + (clip0 + r) * stride + t0 + c);
                    __builtin_memcpy(tmp, &q, 16);
                } else {
#pragma unroll
                    for (int i = 0; i < PER; ++i) tmp[i] = TIO(0);
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) tin[r * IIR_LD + c + i] = tmp[i];
            }
        } else
        for (int e = lane; e < 64 * IIR_TS; e += 64) {
            const int r = e / IIR_TS, cidx = e % IIR_TS;
            TIO v = TIO(0);
            if (r < rows && cidx < cols) v = x[(clip0 + r) * stride + t0 + cidx];
            tin[r * IIR_LD + cidx] = v;
        }
        __syncthreads();
        if (lane < rows) {
            for (int i = 0; i < cols; ++i) {
                const T xv = (T)tin[lane * IIR_LD + i];
                tout1[lane * IIR_LD + i] = (TIO)iir_step<T, C>(s1, c1, xv);
                if (TWO) tout2[lane * IIR_LD + i] = (TIO)iir_step<T, C>(s2, c2, xv);
            }
        }
        __syncthreads();
        if (vec) {
            constexpr int PER = 16 / sizeof(TIO), CH = IIR_TS / PER;
            for (int e = lane; e < 64 * CH; e += 64) {
                const int r = e / CH, c = (e % CH) * PER;
                if (r < rows) {
                    TIO tmp[PER];
                    float4 q;
#pragma unroll
                    for (int i = 0; i < PER; ++i) tmp[i] = tout1[r * IIR_LD + c + i];
                    __builtin_memcpy(&q, tmp, 16);
                    *reinterpret_cast<float4 *>(y1 + (clip0 + r) * stride + t0 + c) = q;
                    if (TWO) {
#pragma unroll
                        for (int i = 0; i < PER; ++i) tmp[i] = tout2[r * IIR_LD + c + i];
                        __builtin_memcpy(&q, tmp, 16);
                        *reinterpret_cast<float4 *>(y2 + (clip0 + r) * stride + t0 + c) = q;
                    }
                }
            }
        } else
        for (int e = lane; e < 64 * IIR_TS; e += 64) {
            const int r = e / IIR_TS, cidx = e % IIR_TS;
            if (r < rows && cidx < cols) {
                y1[(clip0 + r) * stride + t0 + cidx] = tout1[r * IIR_LD + cidx];
                if (TWO) y2[(clip0 + r) * stride + t0 + cidx] = tout2[r * IIR_LD + cidx];
            }
        }
        __syncthreads();
    }
}

hipError_t launch_iir_f32(const float *x, long n_clips, int n, long stride, const IirCoef &c1, float *y1,
                          const IirCoef &c2, float *y2, hipStream_t stream)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    const int blocks = (int)((n_clips + 63) / 64);
    if (y2) hipLaunchKernelGGL((iir_kernel<float, IirCoef, true>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, c1, y1, c2, y2);
    else hipLaunchKernelGGL((iir_kernel<float, IirCoef, false>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, c1, y1, c1, y1);
    return hipGetLastError();
}

hipError_t launch_iir_f64_on_f32(const float *x, long n_clips, int n, long stride, const IirCoefD &c, float *y,
                                 hipStream_t stream)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    const int blocks = (int)((n_clips + 63) / 64);
    hipLaunchKernelGGL((iir_kernel<double, IirCoefD, false, float>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, c, y, c, y);
    return hipGetLastError();
}

hipError_t launch_iir_f64(const double *x, long n_clips, int n, long stride, const IirCoefD &c, double *y,
                          hipStream_t stream)
{
    if (n_clips <= 0 || n <= 0) return hipSuccess;
    const int blocks = (int)((n_clips + 63) / 64);
    hipLaunchKernelGGL((iir_kernel<double, IirCoefD, false>), dim3(blocks), dim3(64), 0, stream, x, n_clips, n, stride, c, y, c, y);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// a10: spectrogram.  One lane per (clip, time bin): the lane runs the reference's per-frame
// algorithm serially (sequential mean, window, in-place radix-2 FFT, PSD) on a private
// 256-point work array.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bitrev8(unsigned v) { return __brev(v) >> 24; }

__global__ __launch_bounds__(64) void spectrogram_kernel(const float *__restrict__ y, long n_clips, int n, long stride,
                                                         const SpecTables *__restrict__ tab, float *__restrict__ sxx, int T)
{
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= n_clips * T) return;
    const long clip = gid / T;
    const int t = (int)(gid - clip * T);
    const float *seg = y + clip * stride + (long)t * kSpecHop;
    float re[kSpecSeg], im[kSpecSeg];
    // classifier.cpp:329-346: sequential sum, mean, detrend, window
    float sum = 0.0f;
    for (int i = 0; i < kSpecSeg; ++i) sum = sum + seg[i];
    const float mean = sum / (float)kSpecSeg;
    // bit-reversal permutation (PlainFFT.cpp:33-47) applied while loading
    for (int i = 0; i < kSpecSeg; ++i) {
        const float v = (seg[i] - mean) * tab->window[i];
        re[bitrev8(i)] = v;
        im[bitrev8(i)] = 0.0f;
    }
    // PlainFFT.cpp:50-84: levels l = 0..7, column m uses (u1,u2) = tw[l][m]
    int l1 = 1;
    for (int l = 0; l < 8; ++l) {
        const int l2 = l1 << 1;
        const float *ur = tab->tw_re + (l1 - 1), *ui = tab->tw_im + (l1 - 1);
        for (int m = 0; m < l1; ++m) {
            const float u1 = ur[m], u2 = ui[m];
            for (int i = m; i < kSpecSeg; i += l2) {
                const int i1 = i + l1;
                const float t1 = u1 * re[i1] - u2 * im[i1];
                const float t2 = u1 * im[i1] + u2 * re[i1];
                re[i1] = re[i] - t1;
                im[i1] = im[i] - t2;
                re[i] = re[i] + t1;
                im[i] = im[i] + t2;
            }
        }
        l1 = l2;
    }
    // classifier.cpp:350-365: PSD, one-sided doubling of bins 1..127
    const float U = tab->U;
    float *out = sxx + clip * (long)kSpecBins * T + t;
    for (int k = 0; k < kSpecBins; ++k) {
        float p = (re[k] * re[k] + im[k] * im[k]) / U;
        if (k >= 1 && k < kSpecBins - 1) p = p * 2.0f;
        out[(long)k * T] = p;
    }
}

hipError_t launch_spectrogram_f32(const float *y, long n_clips, int n, long stride, const SpecTables *tables,
                                  float *sxx, hipStream_t stream)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    const long total = n_clips * T;
    hipLaunchKernelGGL(spectrogram_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, stream, y, n_clips, n, stride, tables, sxx, T);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// a11: dB maps, normalisation band, midpoints, three band sums, rule.  One wave per clip:
// element-wise steps and min/max (order-independent, exact) run across lanes; the
// order-dependent float sums (cluster means, sum_intense) run on lane 0 in the reference order.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float to_db(float s)
{
    // classifier.cpp:43: 10 * log10(s / 1e-12) evaluated in double, stored as float
    return (float)(10 * log10((double)s / 1e-12));
}

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

__device__ float sum_intense_dev(float lower, float upper, float half_range, int fs, const float *times_unused,
                                 int T, const float *db, float midpoint)
{
    // classifier.cpp:370-431 with freqs[k] = k*fs/256 and times[t] = (224 t + 128)/fs recomputed in place
    (void)times_unused;
    auto freq = [&](int k) { return (float)k * (float)fs / (float)kSpecSeg; };
    auto time = [&](int t) { return ((float)(t * kSpecHop + kSpecSeg / 2)) / (float)fs; };
    int f0 = 0;
    while (f0 < kSpecBins && freq(f0) < lower) ++f0;
    int f1 = kSpecBins - 1;
    while (f1 >= 0 && freq(f1) > upper) --f1;
    if (f0 >= kSpecBins) f0 = kSpecBins - 1;
    if (f1 < 0) f1 = 0;
    if (f0 > f1) { int x = f0; f0 = f1; f1 = x; }
    int t0 = 0;
    while (t0 < T && time(t0) < midpoint - half_range) ++t0;
    int t1 = T - 1;
    while (t1 >= 0 && time(t1) > midpoint + half_range) --t1;
    if (t0 >= T) t0 = T - 1;
    if (t1 < 0) t1 = 0;
    if (t0 > t1) { int x = t0; t0 = t1; t1 = x; }
    float total = 0.0f;
    for (int i = f0; i <= f1; ++i)
        for (int j = t0; j <= t1; ++j) {
            const float v = db[(long)i * T + j];
            if (!isnan(v)) total = total + v;
        }
    return total;
}

__global__ __launch_bounds__(64) void classify_tail_kernel(float *__restrict__ sxx_bp, float *__restrict__ sxx_mp,
                                                           long n_clips, int T, int fs, int *__restrict__ labels,
                                                           ClassifyTrace *__restrict__ trace)
{
    const long clip = blockIdx.x;
    if (clip >= n_clips) return;
    const int lane = threadIdx.x;
    const int cells = kSpecBins * T;
    float *bp = sxx_bp + clip * (long)cells;
    float *mp = sxx_mp + clip * (long)cells;
    __shared__ float blob[1024];
    __shared__ int n_blob;

    // ---- band-pass map: dB, clip min/max, normalise, keep (0.65, 0.80)  classifier.cpp:35-80
    float mn = INFINITY, mx = -INFINITY;   // the reference starts from +-DBL_MAX stored in floats = +-inf
    for (int i = lane; i < cells; i += 64) {
        float v = bp[i];
        if (v > 0) {
            v = to_db(v);
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        } else {
            v = NAN;
        }
        bp[i] = v;
    }
    mn = wave_min(mn);
    mx = wave_maxf(mx);
    const float lo_thr = 0.65f, hi_thr = 0.80f;
    for (int i = lane; i < cells; i += 64) {
        float v = bp[i];
        if (!isnan(v)) {
            v = (v - mn) / (mx - mn);
            v = (v > lo_thr && v < hi_thr) ? v : NAN;
            bp[i] = v;
        }
    }
    // ---- midpoint map: dB, keep > 70 dB, time bins with any cell   classifier.cpp:457-518
    for (int i = lane; i < cells; i += 64) {
        float v = mp[i];
        v = (v > 0) ? to_db(v) : NAN;
        mp[i] = (v > 70.0f) ? v : NAN;
    }
    __syncthreads();
    if (lane == 0) {
        int nb = 0;
        for (int j = 0; j < T && nb < 1024; ++j) {
            bool any = false;
            for (int i = 0; i < kSpecBins && !any; ++i) any = !isnan(mp[(long)i * T + j]);
            if (any) blob[nb++] = ((float)(j * kSpecHop + kSpecSeg / 2)) / (float)fs;
        }
        n_blob = nb;
        // greedy clustering, classifier.cpp:522-574
        const float tol = 0.05f, min_dur = 0.15f;
        float mids[kMaxMidpoints];
        int count = 0, i0 = 0;
        while (i0 < nb) {
            int i1 = i0;
            while (i1 + 1 < nb && (blob[i1 + 1] - blob[i1]) <= tol) ++i1;
            const float dur = blob[i1] - blob[i0];
            if (dur >= min_dur) {
                float s = 0.0f;
                for (int k = i0; k <= i1; ++k) s = s + blob[k];
                if (count < kMaxMidpoints) mids[count] = s / (float)(i1 - i0 + 1);
                ++count;
            }
            i0 = i1 + 1;
        }
        if (count > kMaxMidpoints) count = kMaxMidpoints;
        // classifier.cpp:93-114
        int hit = 0;
        if (trace) trace[clip].n_midpoints = count;
        for (int k = 0; k < count; ++k) {
            const float above = sum_intense_dev(5000, 7000, 0.18f, fs, nullptr, T, bp, mids[k]);
            const float middle = sum_intense_dev(2500, 5000, 0.05f, fs, nullptr, T, bp, mids[k]);
            const float below = sum_intense_dev(500, 2500, 0.18f, fs, nullptr, T, bp, mids[k]);
            if (trace) {
                trace[clip].midpoints[k] = mids[k];
                trace[clip].sums[k][0] = above; trace[clip].sums[k][1] = middle; trace[clip].sums[k][2] = below;
            }
            if (middle < 100 && above > 200 && below > 80) { hit = 1; break; }
        }
        labels[clip] = hit;
    }
}

hipError_t launch_classify_tail(float *sxx_bp, float *sxx_mp, long n_clips, int n, int fs, int *labels,
                                ClassifyTrace *trace, hipStream_t stream)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(classify_tail_kernel, dim3((unsigned)n_clips), dim3(64), 0, stream, sxx_bp, sxx_mp, n_clips, T, fs, labels, trace);
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
}

}  // namespace dsp
