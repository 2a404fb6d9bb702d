// classify_f64_kernels.hip -- the float64 scrub-jay classifier of donut-classifier/classifier.c (main's per-file body :83-192,
// sum_intense :594-653, find_midpoints :655-830) after its two band-pass filters and spectrograms (iir_kernel<double>,
// spectrogram_f64_kernel in classify_kernels.hip): dB maps, 45 dB midpoints, clip-global normalisation, keep band, three
// band sums per midpoint, rule.  One 256-thread block per clip, everything in double.
//
//   phase A  1000-3000 Hz map: time bins with a cell above the threshold (any order: a flag per column)
//   phase B  one thread clusters the blob times and averages them in the reference's order (:747-800)
//   phase C  3000-7500 Hz map: minimum / maximum of the dB values over the clip (order-independent, exact)
//   phase D  per midpoint, until the rule fires: the band window = the reference's index searches (:597-639; the frequency rows are
//            the same for every clip and come from the host, the time columns are counted in parallel over the monotonic bin
//            times: the first bin not below t - half = the number of bins below it), its kept cells staged into LDS by all
//            threads, then ONE thread adds them row by row, column by column, NaN cells skipped -- the order of a float64 sum is
//            part of its value (:643-651)
//
// FFTW (the reference's transform) is unvendored, so the spectrogram is a float64 transform checked by tolerance -- for batches
// spectrogram_f64_fft_kernel below (a wavefront per frame, 128-point complex Stockham FFT through LDS, maps frame-major), for
// dsp_compute_spectrogram_f64 the direct DFT of classify_kernels.hip; from there on every operation is the reference's, and what can
// differ is the last bit of log10 (ocml vs glibc).  Two shortcuts that cannot change a decision: a cell's "dB above the midpoint
// threshold" is decided by comparing the cell with the threshold's power unless it lies within 1e-9 relative of it (then the
// reference's expression is evaluated), and the clip's minimum / maximum dB are the dB of its smallest / largest positive cell.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "classify_kernels.hpp"

namespace dsp {

namespace {

constexpr int kMaxColsF64 = 960;            // >= capi.cpp's kMaxSpecColumns (957): a flag per spectrogram column
constexpr int kWinCells = 48 * 32;          // band window staged in LDS: the widest band has 41 rows (2500-5000 Hz) x 27 columns (0.36 s at 14 ms per column)
constexpr int kTailLoads = 8;               // map cells a thread has in flight while it scans a map (one at a time left the scan bound by the load latency)

__device__ __forceinline__ double to_db64(double s) { return 10 * log10(s / 1e-12); }      // classifier.c:113, :688

__device__ __forceinline__ void wave_sync_lds()
{   // a wave's DS instructions complete in order: ordering the compiler is all a write -> read of another lane's data needs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct cd { double re, im; };
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd cmul(cd a, double wr, double wi) { return {a.re * wr - a.im * wi, a.re * wi + a.im * wr}; }

}  // namespace

// compute_spectrogram (classifier.c:448-592) for the batch path.  Wave w transforms frames w, w + n_waves, ...: lane l loads samples
// 2 l, 2 l + 1 and 128 + 2 l, 129 + 2 l (two 16-byte loads, the next frame's in flight during this one's transform), the mean comes
// from a wave reduction, the detrended windowed samples are packed as z[n] = x[2 n] + i x[2 n + 1] and go through a radix-2 Stockham
// FFT of 128 points (seven stages, two points per lane, ping-pong through 4 KB of LDS per wave, the window and the per-stage twiddles
// in registers), then the real spectrum X[k] = E[k] + W256^k O[k] is taken from Z[k] and conj(Z[128 - k]), and
// |X|^2 / U (doubled for 0 < k < 128) is stored at sxx[frame][k].
__global__ __launch_bounds__(256) void spectrogram_f64_fft_kernel(const double *__restrict__ y, long total, int T, long stride,
                                                                  const SpecTablesD *__restrict__ tab, double *__restrict__ sxx)
{
    __shared__ __attribute__((aligned(16))) cd buf[4][2][kSpecSeg / 2];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    cd *cur = buf[wib][0], *nxt = buf[wib][1];
    const double w0 = tab->win[2 * lane], w1 = tab->win[2 * lane + 1], w2 = tab->win[128 + 2 * lane], w3 = tab->win[129 + 2 * lane];
    double tr[7], ti[7];
    int jout[7];
#pragma unroll
    for (int s = 1; s < 7; ++s) {
        const int p = 1 << s, k = lane & (p - 1);
        tr[s] = tab->w_re[k << (7 - s)];
        ti[s] = tab->w_im[k << (7 - s)];
        jout[s] = ((lane - k) << 1) + k;
    }
    const double pr0 = tab->w_re[lane], pi0 = tab->w_im[lane], pr1 = tab->w_re[lane + 64], pi1 = tab->w_im[lane + 64];
    const double U = tab->U;
    const long wave = (long)blockIdx.x * 4 + wib, n_waves = (long)gridDim.x * 4;
    typedef double d2 __attribute__((ext_vector_type(2)));
    auto src_of = [&](long f) {
        const long clip = f / T;
        return y + clip * stride + (f - clip * T) * (long)kSpecHop + 2 * lane;
    };
    d2 na = {0, 0}, nb = {0, 0};
    if (wave < total) {
        const double *src = src_of(wave);
        na = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src));
        nb = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src + 128));
    }
    for (long f = wave; f < total; f += n_waves) {
        const d2 a = na, b = nb;
        if (f + n_waves < total) {
            const double *src = src_of(f + n_waves);
            na = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src));
            nb = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src + 128));
        }
        double sum = (a.x + a.y) + (b.x + b.y);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const double mean = sum / (double)kSpecSeg;                      // classifier.c:551-561 detrend
        const cd z0 = {(a.x - mean) * w0, (a.y - mean) * w1}, z1 = {(b.x - mean) * w2, (b.y - mean) * w3};
        // stage 0 (p = 1, twiddle 1)
        cur[2 * lane] = z0 + z1;
        cur[2 * lane + 1] = z0 - z1;
        wave_sync_lds();
#pragma unroll
        for (int s = 1; s < 7; ++s) {
            const cd u0 = cur[lane], u1 = cmul(cur[lane + 64], tr[s], ti[s]);
            nxt[jout[s]] = u0 + u1;
            nxt[jout[s] + (1 << s)] = u0 - u1;
            wave_sync_lds();
            cd *t = cur; cur = nxt; nxt = t;
        }
        // Z in natural order in cur.  X[k] = (A + B) / 2 + W256^k (A - B) / (2 i), A = Z[k], B = conj(Z[128 - k])
        auto bin = [&](int k, double wr, double wi) {
            const cd A = cur[k & 127], Zb = cur[(128 - k) & 127];
            const cd e2 = {A.re + Zb.re, A.im - Zb.im}, d = {A.re - Zb.re, A.im + Zb.im};
            const cd o2 = {d.im, -d.re};
            const cd x2 = e2 + cmul(o2, wr, wi);
            const double re = 0.5 * x2.re, im = 0.5 * x2.im;
            return (re * re + im * im) / U;                              // :574-586
        };
        double p0 = bin(lane, pr0, pi0);
        const double p1 = bin(lane + 64, pr1, pi1) * 2.0;
        if (lane > 0) p0 *= 2.0;
        double *out = sxx + f * (long)kSpecBins;
        out[lane] = p0;
        out[lane + 64] = p1;
        if (lane == 0) {
            const cd Z0 = cur[0];
            const double r = Z0.re - Z0.im;                              // X[128] = E[0] - O[0]
            out[128] = (r * r) / U;
        }
        wave_sync_lds();                                                 // the next frame's stage 0 overwrites what bin() read (six swaps: cur is buf[wib][0] again)
    }
}

void build_spec_tables_f64(int fs, SpecTablesD &t)
{
    // classifier.c:484-521 with alpha = 0.25, window_size 256 (M = 257): the periodic Tukey window, in the reference's expressions
    const double alpha = 0.25, PI = 3.14159265358979323846;
    const int M = kSpecSeg + 1;
    const int width = (int)std::floor(alpha * (M - 1) / 2.0);
    for (int n = 0; n < kSpecSeg; ++n) {
        if (n <= width) t.win[n] = 0.5 * (1 + std::cos(PI * (-1 + 2.0 * n / (alpha * (M - 1)))));
        else if (n <= M - width - 2) t.win[n] = 1.0;
        else t.win[n] = 0.5 * (1 + std::cos(PI * (-2.0 / alpha + 1 + 2.0 * n / (alpha * (M - 1)))));
    }
    double U = 0.0;                                                      // :524-530
    for (int i = 0; i < kSpecSeg; ++i) U += t.win[i] * t.win[i];
    t.U = U * fs;
    for (int k = 0; k < kSpecSeg / 2; ++k) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)kSpecSeg;
        t.w_re[k] = (double)cosl(a);
        t.w_im[k] = (double)sinl(a);
    }
}

hipError_t launch_spectrogram_f64_fft(const double *y, long n_clips, int n, long stride, const SpecTablesD *tables, double *sxx, hipStream_t stream)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    if (stride % 2 != 0 || reinterpret_cast<uintptr_t>(y) % 16 != 0) return hipErrorInvalidValue;
    const long total = n_clips * T;
    static int resident = 0;                  // blocks that fit the GPU at once: the waves walk the frames from there
    if (resident == 0) {
        int dev = 0, cus = 0, per = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, spectrogram_f64_fft_kernel, 256, 0) != hipSuccess || cus <= 0 || per <= 0) {
            (void)hipGetLastError();
            cus = 256; per = 2;
        }
        resident = cus * per;
    }
    const long blocks = std::min<long>((total + 3) / 4, resident);
    hipLaunchKernelGGL(spectrogram_f64_fft_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, total, T, stride, tables, sxx);
    return hipGetLastError();
}

struct BandRows { int f0[3], f1[3]; };       // sum_intense's frequency rows of the three bands (classifier.c:597-617), the same for every clip

template <bool FM>
__global__ __launch_bounds__(256) void classify_f64_tail_kernel(const double *__restrict__ sxx_bp, const double *__restrict__ sxx_mp, long n_clips,
                                                                int T, int fs, ClassifyRuleD rule, double mid_power, BandRows bands,
                                                                int *__restrict__ labels, ClassifyTraceD *__restrict__ trace)
{
    __shared__ double times[kMaxColsF64];
    __shared__ int flags[kMaxColsF64];
    __shared__ double red_lo[256], red_hi[256];
    __shared__ double mids[kMaxMidpoints];
    __shared__ double win[kWinCells];
    __shared__ int sh_i[8];                 // n_mid, f0, f1, t0, t1, staged, hit
    const int tid = threadIdx.x;
    const long clip = blockIdx.x;
    if (clip >= n_clips) return;
    const double *mp = sxx_mp + clip * (long)kSpecBins * T;
    const double *bp = sxx_bp + clip * (long)kSpecBins * T;
    const int cells = kSpecBins * T;
    auto time_of = [&](int j) { return times[j]; };      // classifier.c compute_spectrogram (:478-481): segment centres

    // ---- phase A: find_midpoints' mask (:679-745) ------------------------------------------------------------------------
    for (int j = tid; j < T; j += 256) { flags[j] = 0; times[j] = (double)(j * kSpecHop + kSpecSeg / 2) / (double)fs; }
    __syncthreads();
    // a cell well above / below the threshold's power mid_power = 1e-12 * 10^(midpoint_db / 10) is decided by a comparison; within
    // 1e-9 relative of it (4e-9 dB, against the ~1e-14 dB the expression's roundings can move) the reference's expression decides
    const double mid_hi = mid_power * (1.0 + 1e-9), mid_lo = mid_power * (1.0 - 1e-9);
    for (int base = tid; base < cells; base += 256 * kTailLoads) {
        double v[kTailLoads];
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) v[u] = base + 256 * u < cells ? __builtin_nontemporal_load(mp + base + 256 * u) : 0.0;
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) {
            const int idx = base + 256 * u;
            const double s = v[u];
            const bool loud = s > mid_hi || (s >= mid_lo && s > 0 && to_db64(s) > rule.midpoint_db);
            if (loud) flags[FM ? idx / kSpecBins : idx % T] = 1;
        }
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
    // (the dB of the smallest / largest positive cell: to_db64 is monotonic, and one log10 per clip instead of one per cell)
    double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
    for (int base = tid; base < cells; base += 256 * kTailLoads) {
        double v[kTailLoads];
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) v[u] = base + 256 * u < cells ? bp[base + 256 * u] : 0.0;      // (cacheable: phase D reads the windows again)
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) {
            const double s = v[u];
            if (s > 0) {
                lo = s < lo ? s : lo;
                hi = s > hi ? s : hi;
            }
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
    const bool any_cell = red_hi[0] > 0;
    const double mn = any_cell ? to_db64(red_lo[0]) : red_lo[0], mx = any_cell ? to_db64(red_hi[0]) : red_hi[0];
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
            if (tid == 0) { sh_i[3] = 0; sh_i[4] = 0; }
            __syncthreads();
            {
                const double midpoint = mids[k], lo_t = midpoint - band_half[bnd], hi_t = midpoint + band_half[bnd];
                int below = 0, above = 0;
                for (int j = tid; j < T; j += 256) { below += times[j] < lo_t; above += times[j] > hi_t; }
                for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o); above += __shfl_xor(above, o); }
                if ((tid & 63) == 0) { atomicAdd(&sh_i[3], below); atomicAdd(&sh_i[4], above); }
            }
            __syncthreads();
            const int f0 = bands.f0[bnd], f1 = bands.f1[bnd];
            int t0 = sh_i[3], t1 = T - 1 - sh_i[4];                          // :619-639 on monotonic times
            if (t0 >= T) t0 = T - 1;
            if (t1 < 0) t1 = 0;
            if (t0 > t1) { const int t = t0; t0 = t1; t1 = t; }
            const int cols = t1 - t0 + 1, rows = f1 - f0 + 1;
            const bool staged = rows * cols <= kWinCells;
            if (staged) {
                if (FM) for (int idx = tid; idx < rows * cols; idx += 256) win[(idx % rows) * cols + idx / rows] = kept(bp[(long)(t0 + idx / rows) * kSpecBins + f0 + idx % rows]);
                else for (int idx = tid; idx < rows * cols; idx += 256) win[idx] = kept(bp[(long)(f0 + idx / cols) * T + t0 + idx % cols]);
            }
            __syncthreads();
            if (staged) {
                // the ordered sum on wave 0: 64 staged cells per step into the lanes' registers, then one addition per cell in the
                // reference's order, the cell read out of its lane (v_readlane) -- a NaN cell (skipped by the reference) counts as
                // + 0.0, which leaves a sum of non-negative terms that started at + 0.0 unchanged, bit for bit
                if (tid < 64) {
                    const int n_cells = rows * cols;
                    double total = 0.0;
                    for (int c0 = 0; c0 < n_cells; c0 += 64) {
                        double v = c0 + tid < n_cells ? win[c0 + tid] : 0.0;
                        v = v != v ? 0.0 : v;
                        const int lo = __double2loint(v), hi = __double2hiint(v);
#pragma unroll
                        for (int l = 0; l < 64; ++l)
                            total = total + __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
                    }
                    sums[bnd] = total;
                }
            } else if (tid == 0) {
                double total = 0.0;
                for (int i = 0; i < rows; ++i)
                    for (int j = 0; j < cols; ++j) {
                        const double v = kept(FM ? bp[(long)(t0 + j) * kSpecBins + f0 + i] : bp[(long)(f0 + i) * T + t0 + j]);
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
                                    int *labels, ClassifyTraceD *trace, hipStream_t stream, bool frame_major)
{
    const int T = n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1;
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > kMaxColsF64 || n_clips >= (1L << 31)) return hipErrorInvalidValue;
    const double mid_power = 1e-12 * std::pow(10.0, rule.midpoint_db / 10.0);
    // sum_intense's frequency searches (classifier.c:597-617), once for all clips: the bins' frequencies in the reference's expression
    const double band_lo[3] = {5000, 2500, 500}, band_hi[3] = {7000, 5000, 2500};
    auto freq_of = [&](int i) { return (double)i * (double)fs / (double)kSpecSeg; };
    BandRows bands;
    for (int bnd = 0; bnd < 3; ++bnd) {
        int f0 = 0;
        while (f0 < kSpecBins && freq_of(f0) < band_lo[bnd]) ++f0;
        int f1 = kSpecBins - 1;
        while (f1 >= 0 && freq_of(f1) > band_hi[bnd]) --f1;
        if (f0 >= kSpecBins) f0 = kSpecBins - 1;
        if (f1 < 0) f1 = 0;
        if (f0 > f1) std::swap(f0, f1);
        bands.f0[bnd] = f0; bands.f1[bnd] = f1;
    }
    if (frame_major) hipLaunchKernelGGL(classify_f64_tail_kernel<true>, dim3((unsigned)n_clips), dim3(256), 0, stream, sxx_bp, sxx_mp, n_clips, T, fs, rule, mid_power, bands, labels, trace);
    else hipLaunchKernelGGL(classify_f64_tail_kernel<false>, dim3((unsigned)n_clips), dim3(256), 0, stream, sxx_bp, sxx_mp, n_clips, T, fs, rule, mid_power, bands, labels, trace);
    return hipGetLastError();
}

}  // namespace dsp
