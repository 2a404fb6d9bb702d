// classify_f64_kernels.hip -- the float64 scrub-jay classifier of donut-classifier/classifier.c (main's per-file body :83-192,
// sum_intense :594-653, find_midpoints :655-830) after its two band-pass filters and spectrograms (iir_kernel<double>,
// spectrogram_f64_kernel in classify_kernels.hip): dB maps, 45 dB midpoints, clip-global normalisation, keep band, three
// band sums per midpoint, rule.  Correctness first: one 256-thread block per clip, everything in double.
//
//   phase A  1000-3000 Hz map: time bins with a cell above the threshold (any order: a flag per column)
//   phase B  one thread clusters the blob times and averages them in the reference's order (:747-800)
//   phase C  3000-7500 Hz map: minimum / maximum of the dB values over the clip (order-independent, exact)
//   phase D  per midpoint, until the rule fires: the reference's index searches (:597-639) on one thread, the band window's
//            kept cells staged into LDS by all threads, then ONE thread adds them row by row, column by column, NaN cells
//            skipped -- the order of a float64 sum is part of its value (:643-651)
//
// FFTW (the reference's transform) is unvendored, so the spectrogram is a float64 DFT checked by tolerance; from there on every
// operation is the reference's, and what can differ is the last bit of log10 (ocml vs glibc).
#include <hip/hip_runtime.h>

#include "classify_kernels.hpp"

namespace dsp {

namespace {

constexpr int kMaxColsF64 = 960;            // >= capi.cpp's kMaxSpecColumns (957): a flag per spectrogram column
constexpr int kWinCols = 32;                // band window staged in LDS: 129 rows x <= 32 columns (0.36 s at 14 ms per column = 27)

__device__ __forceinline__ double to_db64(double s) { return 10 * log10(s / 1e-12); }      // classifier.c:113, :688

}  // namespace

__global__ __launch_bounds__(256) void classify_f64_tail_kernel(const double *__restrict__ sxx_bp, const double *__restrict__ sxx_mp, long n_clips,
                                                                int T, int fs, ClassifyRuleD rule, int *__restrict__ labels,
                                                                ClassifyTraceD *__restrict__ trace)
{
    __shared__ int flags[kMaxColsF64];
    __shared__ double red_lo[256], red_hi[256];
    __shared__ double mids[kMaxMidpoints];
    __shared__ double win[kSpecBins * kWinCols];
    __shared__ int sh_i[8];                 // n_mid, f0, f1, t0, t1, staged, hit
    const int tid = threadIdx.x;
    const long clip = blockIdx.x;
    if (clip >= n_clips) return;
    const double *mp = sxx_mp + clip * (long)kSpecBins * T;
    const double *bp = sxx_bp + clip * (long)kSpecBins * T;
    const int cells = kSpecBins * T;
    auto time_of = [&](int j) { return (double)(j * kSpecHop + kSpecSeg / 2) / (double)fs; };      // classifier.c compute_spectrogram: segment centres
    auto freq_of = [&](int i) { return (double)i * (double)fs / (double)kSpecSeg; };

    // ---- phase A: find_midpoints' mask (:679-745) ------------------------------------------------------------------------
    for (int j = tid; j < T; j += 256) flags[j] = 0;
    __syncthreads();
    for (int idx = tid; idx < cells; idx += 256) {
        const double s = mp[idx];
        if (s > 0 && to_db64(s) > rule.midpoint_db) flags[idx % T] = 1;
    }
    __syncthreads();
    // ---- phase B: clusters -> midpoints (:747-800), one thread, the reference's order of additions ------------------------------
    if (tid == 0) {
        const double tol = 0.05, min_dur = 0.15;
        int count = 0, j = 0;
        while (j < T) {
            while (j < T && !flags[j]) ++j;
            if (j >= T) break;
            // a cluster: consecutive BLOB times (flagged columns) whose gaps stay <= tol
            int first = j, last = j, members = 1;
            double sum = time_of(j);
            int k = j + 1;
            while (true) {
                while (k < T && !flags[k]) ++k;
                if (k >= T || !((time_of(k) - time_of(last)) <= tol)) break;
                sum += time_of(k);
                last = k; ++members; ++k;
            }
            if (time_of(last) - time_of(first) >= min_dur) {
                if (count < kMaxMidpoints) mids[count] = sum / (double)members;
                ++count;
            }
            j = k;
        }
        sh_i[0] = count < kMaxMidpoints ? count : kMaxMidpoints;
        sh_i[6] = 0;
    }
    __syncthreads();
    const int n_mid = sh_i[0];
    ClassifyTraceD *tr = trace ? trace + clip : nullptr;
    if (tr) {
        for (int i = tid; i < kMaxMidpoints; i += 256) {
            tr->midpoints[i] = i < n_mid ? mids[i] : 0.0;
            tr->sums[i][0] = tr->sums[i][1] = tr->sums[i][2] = 0.0;
        }
        if (tid == 0) tr->n_midpoints = n_mid;
    }
    if (n_mid == 0) {
        if (tid == 0) labels[clip] = 0;
        return;
    }
    // ---- phase C: clip-global minimum / maximum of the dB map (:105-125) -----------------------------------------------------
    double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
    for (int idx = tid; idx < cells; idx += 256) {
        const double s = bp[idx];
        if (s > 0) {
            const double d = to_db64(s);
            lo = d < lo ? d : lo;
            hi = d > hi ? d : hi;
        }
    }
    red_lo[tid] = lo; red_hi[tid] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red_lo[tid] = red_lo[tid + o] < red_lo[tid] ? red_lo[tid + o] : red_lo[tid];
            red_hi[tid] = red_hi[tid + o] > red_hi[tid] ? red_hi[tid + o] : red_hi[tid];
        }
        __syncthreads();
    }
    const double mn = red_lo[0], mx = red_hi[0];
    // the kept, normalised value of a cell or NaN (:130-157)
    auto kept = [&](double s) {
        if (!(s > 0)) return (double)NAN;
        const double v = (to_db64(s) - mn) / (mx - mn);
        return (v > rule.keep_lo && v < rule.keep_hi) ? v : (double)NAN;
    };
    // ---- phase D: the three band sums per midpoint and the rule (:170-190) --------------------------------------------------------
    const double band_lo[3] = {5000, 2500, 500}, band_hi[3] = {7000, 5000, 2500}, band_half[3] = {0.18, 0.05, 0.18};
    for (int k = 0; k < n_mid; ++k) {
        double sums[3] = {0, 0, 0};
        for (int bnd = 0; bnd < 3; ++bnd) {
            if (tid == 0) {
                const double midpoint = mids[k];
                int f0 = 0;
                while (f0 < kSpecBins && freq_of(f0) < band_lo[bnd]) ++f0;
                int f1 = kSpecBins - 1;
                while (f1 >= 0 && freq_of(f1) > band_hi[bnd]) --f1;
                if (f0 >= kSpecBins) f0 = kSpecBins - 1;
                if (f1 < 0) f1 = 0;
                if (f0 > f1) { const int t = f0; f0 = f1; f1 = t; }
                int t0 = 0;
                while (t0 < T && time_of(t0) < midpoint - band_half[bnd]) ++t0;
                int t1 = T - 1;
                while (t1 >= 0 && time_of(t1) > midpoint + band_half[bnd]) --t1;
                if (t0 >= T) t0 = T - 1;
                if (t1 < 0) t1 = 0;
                if (t0 > t1) { const int t = t0; t0 = t1; t1 = t; }
                sh_i[1] = f0; sh_i[2] = f1; sh_i[3] = t0; sh_i[4] = t1;
                sh_i[5] = (t1 - t0 + 1) <= kWinCols;
            }
            __syncthreads();
            const int f0 = sh_i[1], f1 = sh_i[2], t0 = sh_i[3], t1 = sh_i[4], cols = t1 - t0 + 1, rows = f1 - f0 + 1;
            const bool staged = sh_i[5] != 0;
            if (staged)
                for (int idx = tid; idx < rows * cols; idx += 256) win[idx] = kept(bp[(long)(f0 + idx / cols) * T + t0 + idx % cols]);
            __syncthreads();
            if (tid == 0) {
                double total = 0.0;
                for (int i = 0; i < rows; ++i)
                    for (int j = 0; j < cols; ++j) {
                        const double v = staged ? win[i * cols + j] : kept(bp[(long)(f0 + i) * T + t0 + j]);
                        if (!(v != v)) total += v;
                    }
                sums[bnd] = total;
            }
            __syncthreads();
        }
        if (tid == 0) {
            if (tr) { tr->sums[k][0] = sums[0]; tr->sums[k][1] = sums[1]; tr->sums[k][2] = sums[2]; }
            if (sums[1] < rule.middle_max && sums[0] > rule.above_min && sums[2] > rule.below_min) sh_i[6] = 1;
        }
        __syncthreads();
        if (sh_i[6]) break;
    }
    if (tid == 0) labels[clip] = sh_i[6];
}

hipError_t launch_classify_f64_tail(const double *sxx_bp, const double *sxx_mp, long n_clips, int n, int fs, const ClassifyRuleD &rule,
                                    int *labels, ClassifyTraceD *trace, hipStream_t stream)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > kMaxColsF64 || n_clips >= (1L << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(classify_f64_tail_kernel, dim3((unsigned)n_clips), dim3(256), 0, stream, sxx_bp, sxx_mp, n_clips, T, fs, rule, labels, trace);
    return hipGetLastError();
}

}  // namespace dsp
