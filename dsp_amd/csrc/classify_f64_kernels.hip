// classify_f64_kernels.hip -- the float64 scrub-jay classifier of donut-classifier/classifier.c (main's per-file body :83-192,
// sum_intense :594-653, find_midpoints :655-830) after its two band-pass filters (iir_kernel<double>, classify_kernels.hip).
// A batch runs as
//   spectrogram_f64_fft_kernel<flags>   1000-3000 Hz output: two frames per wavefront (128-point complex Stockham FFT through LDS); only
//                                       "a cell of this time bin is above the midpoint threshold" leaves the kernel (:679-745)
//   classify_f64_midpoints_kernel       a thread per clip clusters the flagged bins and averages them in the reference's order
//                                       (:747-800); clips with midpoints go on a work list, the others are label 0
//   spectrogram_f64_fft_kernel<maps>    3000-7500 Hz output of the listed clips only: PSD maps, frame-major [entry][t][129]
//   classify_f64_bands_kernel           a block per listed clip: minimum / maximum dB of the map (:105-125), then per midpoint, until
//                                       the rule fires, the three band sums (:170-190): the band window = the reference's index
//                                       searches (:597-639; the frequency rows are the same for every clip and come from the host,
//                                       the time columns are counted in parallel over the monotonic bin times), its kept cells
//                                       staged into LDS by all threads, then ONE wave adds them row by row, column by column, NaN
//                                       cells skipped -- the order of a float64 sum is part of its value (:643-651)
// and, as the yardstick of that pipeline in the tests (DSP_AMD_F64_DFT=1), over the [129][T] maps of the direct DFT
// (spectrogram_f64_kernel, classify_kernels.hip, the transform behind dsp_compute_spectrogram_f64) as ONE kernel per clip,
// classify_f64_tail_kernel: flags, midpoints, minimum / maximum, band sums, rule.
//
// FFTW (the reference's transform) is unvendored, so the spectrogram is a float64 transform checked by tolerance; from there on every
// operation is the reference's, and what can differ is the last bit of log10 (ocml vs glibc).  Two shortcuts that cannot change a
// decision: a cell's "dB above the midpoint threshold" is decided by comparing the cell with the threshold's power unless it lies
// within 1e-9 relative of it (then the reference's expression is evaluated), and the clip's minimum / maximum dB are the dB of its
// smallest / largest positive cell.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "classify_f64_device.hpp"
#include "classify_kernels.hpp"

namespace dsp {

using namespace f64dev;

namespace {

constexpr int kMaxColsF64 = 960;            // >= capi.cpp's kMaxSpecColumns (957): a flag per spectrogram column
constexpr int kWinCells = 48 * 32;          // band window staged in LDS: the widest band has 41 rows (2500-5000 Hz) x 27 columns (0.36 s at 14 ms per column)
constexpr int kTailLoads = 8;               // map cells a thread has in flight while it scans a map (one at a time left the scan bound by the load latency)

}  // namespace

// compute_spectrogram (classifier.c:448-592) of MATERIALISED filter outputs (the yardstick pipeline, DSP_AMD_F64_PIPELINE=materialize;
// the default pipeline transforms segments recomputed from checkpoints, classify_f64_ckpt_kernels.hip, with the same fft_frame):
// TWO frames per wavefront, 32 lanes own a frame; lane i loads samples 2 i + 64 r, 2 i + 64 r + 1, r < 4 (four 16-byte loads per
// lane, 512 contiguous bytes per load and frame, the next turn's loads in flight during this one's transform).  Measured history of
// this kernel (radix 2 -> radix 4, twiddle placement, padded stage image, DPP mean): profiles/r03_classify_f64_session2.txt.
//   MAPS = false  every frame of every clip: loud[frame] = one of its 129 cells is above the midpoint threshold (no map leaves the kernel)
//   MAPS = true   the frames of the clips on the work list hits (hits[0] entries, clip numbers from hits[1]): sxx[entry][t][k] = U * PSD
template <bool MAPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void spectrogram_f64_fft_kernel(const double *__restrict__ y, long n_clips, int T, long stride,
                                                                   const SpecTablesD *__restrict__ tab, const int *__restrict__ hits,
                                                                   double *__restrict__ sxx, int *__restrict__ loud, double mid_power, double midpoint_db, double guard)
{
    const long total = (MAPS ? (long)hits[0] : n_clips) * T;
    __shared__ __attribute__((aligned(16))) cd buf0[4][2][kPingCd], buf1[4][2][kPongCd];       // [wave][half]
    __shared__ __attribute__((aligned(16))) FftTwiddles tw;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, half = lane >> 5, i = lane & 31;
    cd *b0 = buf0[wib][half], *b1 = buf1[wib][half];
    fill_twiddles(tw, tab, threadIdx.x);
    __syncthreads();
    FftLane L;
    fft_lane_init(L, tw, tab, i);
    const double U = tab->U;
    const long wave = (long)blockIdx.x * 4 + wib, n_waves = (long)gridDim.x * 4;
    // this half's frame f = 2 wave + half, then + 2 n_waves per turn; (entry, column) kept incrementally: one division per kernel
    long f = 2 * wave + half;
    long e = f / T;
    int t = (int)(f - e * T);
    const long step = 2 * n_waves, step_e = step / T;
    const int step_t = (int)(step - step_e * T);
    auto src_of = [&](long ee, int tt) {
        const long clip = MAPS ? (long)hits[1 + ee] : ee;
        return y + clip * stride + (long)tt * kSpecHop + 2 * i;
    };
    d2 q[4];
    auto advance = [&](long &ee, int &tt) {
        ee += step_e; tt += step_t;
        if (tt >= T) { tt -= T; ++ee; }
    };
    long ea = e;                                                         // (entry, column) of the next frame to request
    int ta = t;
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = d2{0.0, 0.0};
    if (f < total) {
        const double *src = src_of(ea, ta);
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src + 64 * r));
    }
    advance(ea, ta);
    for (long f0 = 2 * wave; f0 < total; f0 += step) {                    // wave-uniform: the wave runs while its first frame exists
        const bool live = f < total;
        d2 x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = q[r];
        if (f + step < total) {
            const double *src = src_of(ea, ta);
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(src + 64 * r));
        }
        double m[4], m128;
        fft_frame(x, L, tw, b0, b1, i, m, m128);
        if (MAPS) {
            if (live) {
                double *out = sxx + f * (long)kSpecBins;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[i + 32 * r] = m[r];
                if (i == 0) out[128] = m128;
            }
        } else {
            const bool hit = frame_is_loud(m, m128, i, half, U, mid_power, midpoint_db, guard);
            if (i == 0 && live) loud[f] = hit;
        }
        wave_sync_lds();                                                 // the next turn's stages overwrite b0 / b1
        f += step;
        advance(ea, ta);
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

double f64_threshold_guard()
{
    const char *ge = std::getenv("DSP_AMD_F64_GUARD");
    return ge && std::atof(ge) >= 2e-9 && std::atof(ge) < 1.0 ? std::atof(ge) : 2e-9;
}

namespace {

// blocks that fit the GPU at once (the waves walk the frames from there), per device: a process may drive several GPUs
template <typename K>
int resident_blocks_of(K kernel, int (&cache)[64])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 512; }
    if (cache[dev] == 0) {
        int cus = 0, per = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, 0) != hipSuccess || cus <= 0 || per <= 0) {
            (void)hipGetLastError();
            cus = 256; per = 2;
        }
        cache[dev] = cus * per;
    }
    return cache[dev];
}
template <bool MAPS>
int fft_resident_blocks()
{
    static int cache[64] = {0};
    return resident_blocks_of(spectrogram_f64_fft_kernel<MAPS>, cache);
}

int columns_of(int n) { return n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1; }

}  // namespace

hipError_t launch_spectrogram_f64_flags(const double *y, long n_clips, int n, long stride, const SpecTablesD *tables, double midpoint_db,
                                        int *loud, hipStream_t stream)
{
    const int T = columns_of(n);
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    if (stride % 2 != 0 || reinterpret_cast<uintptr_t>(y) % 16 != 0) return hipErrorInvalidValue;
    const long blocks = std::min<long>((n_clips * T + 7) / 8, fft_resident_blocks<false>());
    const double mid_power = 1e-12 * std::pow(10.0, midpoint_db / 10.0);
    const double guard = f64_threshold_guard();
    hipLaunchKernelGGL(spectrogram_f64_fft_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, y, n_clips, T, stride, tables,
                       (const int *)nullptr, (double *)nullptr, loud, mid_power, midpoint_db, guard);
    return hipGetLastError();
}

hipError_t launch_spectrogram_f64_listed(const double *y, long n_clips, int n, long stride, const SpecTablesD *tables, const int *hits,
                                         double *sxx, hipStream_t stream)
{
    const int T = columns_of(n);
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    if (stride % 2 != 0 || reinterpret_cast<uintptr_t>(y) % 16 != 0) return hipErrorInvalidValue;
    const long blocks = std::min<long>((n_clips * T + 7) / 8, fft_resident_blocks<true>());      // the bound: the list's count is read on the device
    hipLaunchKernelGGL(spectrogram_f64_fft_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, y, n_clips, T, stride, tables, hits, sxx,
                       (int *)nullptr, 0.0, 0.0, 0.0);
    return hipGetLastError();
}

struct BandRows { int f0[3], f1[3]; };       // sum_intense's frequency rows of the three bands (classifier.c:597-617), the same for every clip

namespace {

// the centre time of spectrogram column j (classifier.c:478-481)
__device__ __forceinline__ double column_time(int j, int fs) { return (double)(j * kSpecHop + kSpecSeg / 2) / (double)fs; }

// find_midpoints' clusters (:747-800) on ONE thread, the reference's order of additions: consecutive blob times (flagged columns)
// whose gaps stay <= 0.05 s, kept when they span >= 0.15 s; mids[] takes the first kMaxMidpoints cluster means.  Returns their number.
template <typename Flag>
__device__ __forceinline__ int cluster_midpoints(Flag flag, const double *times, int T, double *mids)
{
    const double tol = 0.05, min_dur = 0.15;
    int count = 0, j = 0;
    while (j < T) {
        while (j < T && !flag(j)) ++j;
        if (j >= T) break;
        int first = j, last = j, members = 1;
        double sum = times[j];
        int k = j + 1;
        while (true) {
            while (k < T && !flag(k)) ++k;
            if (k >= T || !((times[k] - times[last]) <= tol)) break;
            sum += times[k];
            last = k; ++members; ++k;
        }
        if (times[last] - times[first] >= min_dur) {
            if (count < kMaxMidpoints) mids[count] = sum / (double)members;
            ++count;
        }
        j = k;
    }
    return count < kMaxMidpoints ? count : kMaxMidpoints;
}

// LDS of the band sums: a 256-thread block works on one clip
struct BandShared {
    double red_lo[256], red_hi[256];
    double win[kWinCells];
    int cnt[2], hit;
};

// main's per-file body after find_midpoints (:105-190) for one clip, by a whole block: bp = the clip's 3000-7500 Hz PSD map,
// [129][T] or (FM) [T][129]; times[] / mids[] in LDS.  Returns the label (every thread).
// FM: the map holds U * PSD (spectrogram_f64_fft_kernel<maps>): a cell is divided by U where its value is used, the minimum and
// the maximum after they are found (x / U is monotonic)
// mm (optional): the smallest / largest positive cell of the map as double bits, found by the kernel that wrote it -- the scan is skipped
template <bool FM>
__device__ __forceinline__ int band_sums_and_rule(const double *__restrict__ bp, int T, const ClassifyRuleD &rule, const BandRows &bands,
                                                  const double *times, const double *mids, int n_mid, ClassifyTraceD *tr, BandShared &sh, double U,
                                                  const unsigned long long *__restrict__ mm = nullptr)
{
    const int tid = threadIdx.x;
    const int cells = kSpecBins * T;
    // ---- clip-global minimum / maximum of the dB map (:105-125): the dB of the smallest / largest positive cell (to_db64 is
    // monotonic: one log10 per clip instead of one per cell) ----
    double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
    if (mm) {
        if (tid == 0) {
            const unsigned long long lo_b = mm[0], hi_b = mm[1];
            sh.red_lo[0] = hi_b ? __longlong_as_double((long long)lo_b) : lo;        // (no positive cell: the scan's own initial values)
            sh.red_hi[0] = hi_b ? __longlong_as_double((long long)hi_b) : hi;
            sh.hit = 0;
        }
        __syncthreads();
    } else {
    for (int base = tid; base < cells; base += 256 * kTailLoads) {
        double v[kTailLoads];
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) v[u] = base + 256 * u < cells ? bp[base + 256 * u] : 0.0;      // (cacheable: the windows are read again)
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) {
            const double s = v[u];
            if (s > 0) {
                lo = s < lo ? s : lo;
                hi = s > hi ? s : hi;
            }
        }
    }
    sh.red_lo[tid] = lo; sh.red_hi[tid] = hi;
    if (tid == 0) sh.hit = 0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            sh.red_lo[tid] = sh.red_lo[tid + o] < sh.red_lo[tid] ? sh.red_lo[tid + o] : sh.red_lo[tid];
            sh.red_hi[tid] = sh.red_hi[tid + o] > sh.red_hi[tid] ? sh.red_hi[tid + o] : sh.red_hi[tid];
        }
        __syncthreads();
    }
    }
    const bool any_cell = sh.red_hi[0] > 0;
    const double mn = any_cell ? to_db64(FM ? sh.red_lo[0] / U : sh.red_lo[0]) : sh.red_lo[0];
    const double mx = any_cell ? to_db64(FM ? sh.red_hi[0] / U : sh.red_hi[0]) : sh.red_hi[0];
    // the kept, normalised value of a cell or NaN (:130-157)
    auto kept = [&](double cell) {
        const double s = FM ? cell / U : cell;
        if (!(s > 0)) return (double)NAN;
        const double v = (to_db64(s) - mn) / (mx - mn);
        return (v > rule.keep_lo && v < rule.keep_hi) ? v : (double)NAN;
    };
    // ---- the three band sums per midpoint and the rule (:170-190) ----
    const double band_half[3] = {0.18, 0.05, 0.18};
    for (int k = 0; k < n_mid; ++k) {
        double sums[3] = {0, 0, 0};
        for (int bnd = 0; bnd < 3; ++bnd) {
            if (tid == 0) { sh.cnt[0] = 0; sh.cnt[1] = 0; }
            __syncthreads();
            {
                const double midpoint = mids[k], lo_t = midpoint - band_half[bnd], hi_t = midpoint + band_half[bnd];
                int below = 0, above = 0;
                for (int j = tid; j < T; j += 256) { below += times[j] < lo_t; above += times[j] > hi_t; }
                for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o); above += __shfl_xor(above, o); }
                if ((tid & 63) == 0) { atomicAdd(&sh.cnt[0], below); atomicAdd(&sh.cnt[1], above); }
            }
            __syncthreads();
            const int f0 = bands.f0[bnd], f1 = bands.f1[bnd];
            int t0 = sh.cnt[0], t1 = T - 1 - sh.cnt[1];                      // :619-639 on monotonic times
            if (t0 >= T) t0 = T - 1;
            if (t1 < 0) t1 = 0;
            if (t0 > t1) { const int t = t0; t0 = t1; t1 = t; }
            const int cols = t1 - t0 + 1, rows = f1 - f0 + 1;
            const bool staged = rows * cols <= kWinCells;
            if (staged) {
                if (FM) for (int idx = tid; idx < rows * cols; idx += 256) sh.win[(idx % rows) * cols + idx / rows] = kept(bp[(long)(t0 + idx / rows) * kSpecBins + f0 + idx % rows]);
                else for (int idx = tid; idx < rows * cols; idx += 256) sh.win[idx] = kept(bp[(long)(f0 + idx / cols) * T + t0 + idx % cols]);
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
                        double v = c0 + tid < n_cells ? sh.win[c0 + tid] : 0.0;
                        v = v != v ? 0.0 : v;
                        const int vlo = __double2loint(v), vhi = __double2hiint(v);
#pragma unroll
                        for (int l = 0; l < 64; ++l)
                            total = total + __hiloint2double(__builtin_amdgcn_readlane(vhi, l), __builtin_amdgcn_readlane(vlo, l));
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
            if (sums[1] < rule.middle_max && sums[0] > rule.above_min && sums[2] > rule.below_min) sh.hit = 1;
        }
        __syncthreads();
        if (sh.hit) break;
    }
    const int label = sh.hit;
    __syncthreads();                                                         // (a caller that loops over clips resets sh.hit next)
    return label;
}

BandRows band_rows(int fs)
{
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
    return bands;
}

}  // namespace

// The whole tail of one clip over [129][T] maps (the direct-DFT yardstick path): flags, midpoints, band sums, rule.
__global__ __launch_bounds__(256) void classify_f64_tail_kernel(const double *__restrict__ sxx_bp, const double *__restrict__ sxx_mp, long n_clips,
                                                                int T, int fs, ClassifyRuleD rule, double mid_power, BandRows bands,
                                                                int *__restrict__ labels, ClassifyTraceD *__restrict__ trace)
{
    __shared__ double times[kMaxColsF64];
    __shared__ int flags[kMaxColsF64];
    __shared__ double mids[kMaxMidpoints];
    __shared__ BandShared sh;
    __shared__ int n_mid_sh;
    const int tid = threadIdx.x;
    const long clip = blockIdx.x;
    if (clip >= n_clips) return;
    const double *mp = sxx_mp + clip * (long)kSpecBins * T;
    const double *bp = sxx_bp + clip * (long)kSpecBins * T;
    const int cells = kSpecBins * T;
    for (int j = tid; j < T; j += 256) { flags[j] = 0; times[j] = column_time(j, fs); }
    __syncthreads();
    for (int base = tid; base < cells; base += 256 * kTailLoads) {           // find_midpoints' mask (:679-745)
        double v[kTailLoads];
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u) v[u] = base + 256 * u < cells ? __builtin_nontemporal_load(mp + base + 256 * u) : 0.0;
#pragma unroll
        for (int u = 0; u < kTailLoads; ++u)
            if (is_loud(v[u], mid_power, rule.midpoint_db)) flags[(base + 256 * u) % T] = 1;
    }
    __syncthreads();
    if (tid == 0) n_mid_sh = cluster_midpoints([&](int j) { return flags[j] != 0; }, times, T, mids);
    __syncthreads();
    const int n_mid = n_mid_sh;
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
    const int label = band_sums_and_rule<false>(bp, T, rule, bands, times, mids, n_mid, tr, sh, 1.0);
    if (tid == 0) labels[clip] = label;
}

// find_midpoints' clusters for a batch: a thread per clip reads its row of loud[] (spectrogram_f64_fft_kernel<flags>).  Clips with
// midpoints go on the work list hits (hits[0] = count, then clip numbers; the order is whatever the atomics give, each clip's
// results do not depend on it), mids[clip][] / n_mids[clip] carry them to classify_f64_bands_kernel; the others are label 0.
__global__ __launch_bounds__(256) void classify_f64_midpoints_kernel(const int *__restrict__ loud, long n_clips, int T, int fs, double *__restrict__ mids,
                                                                     int *__restrict__ n_mids, int *__restrict__ hits, int *__restrict__ labels,
                                                                     ClassifyTraceD *__restrict__ trace, unsigned long long *__restrict__ minmax,
                                                                     const ClipSpan *__restrict__ spans = nullptr)
{
    // spans (ragged batches): clip c has spans[c].frames columns, the head of its row of T
    __shared__ double times[kMaxColsF64];
    for (int j = threadIdx.x; j < T; j += 256) times[j] = column_time(j, fs);
    __syncthreads();
    const long clip = (long)blockIdx.x * 256 + threadIdx.x;
    if (clip >= n_clips) return;
    const int *fl = loud + clip * T;
    double *m = mids + clip * kMaxMidpoints;
    const int n_mid = cluster_midpoints([&](int j) { return fl[j] != 0; }, times, spans ? spans[clip].frames : T, m);
    n_mids[clip] = n_mid;
    if (trace) {
        ClassifyTraceD *tr = trace + clip;
        tr->n_midpoints = n_mid;
        for (int i = 0; i < kMaxMidpoints; ++i) {
            tr->midpoints[i] = i < n_mid ? m[i] : 0.0;
            tr->sums[i][0] = tr->sums[i][1] = tr->sums[i][2] = 0.0;
        }
    }
    if (n_mid == 0) labels[clip] = 0;
    else {
        hits[1 + atomicAdd(hits, 1)] = (int)clip;
        if (minmax) { minmax[2 * clip] = 0x7FF0000000000000ull; minmax[2 * clip + 1] = 0ull; }      // +inf, 0: the map kernel's atomicMin / atomicMax
    }
}

// Band sums and rule of the listed clips: block b takes entries b, b + gridDim.x, ... of the work list; entry e's map is
// sxx[e][T][129] = U * PSD (spectrogram_f64_fft_kernel<maps> walked the same list).
__global__ __launch_bounds__(256) void classify_f64_bands_kernel(const double *__restrict__ sxx, const int *__restrict__ hits, int T, int fs, double U,
                                                                 ClassifyRuleD rule, BandRows bands, const double *__restrict__ mids_all,
                                                                 const int *__restrict__ n_mids, int *__restrict__ labels, ClassifyTraceD *__restrict__ trace,
                                                                 const unsigned long long *__restrict__ minmax, const ClipSpan *__restrict__ spans = nullptr)
{
    __shared__ double times[kMaxColsF64];
    __shared__ double mids[kMaxMidpoints];
    __shared__ BandShared sh;
    const int tid = threadIdx.x;
    for (int j = tid; j < T; j += 256) times[j] = column_time(j, fs);
    const int n_hits = hits[0];
    for (int e = blockIdx.x; e < n_hits; e += gridDim.x) {
        const long clip = hits[1 + e];
        const int n_mid = n_mids[clip];
        __syncthreads();                                                     // times[] written; the previous entry's mids[] read
        if (tid < n_mid) mids[tid] = mids_all[clip * kMaxMidpoints + tid];
        __syncthreads();
        // (ragged batches: the clip's map is the first spans[clip].frames rows of its T)
        const int label = band_sums_and_rule<true>(sxx + (long)e * T * kSpecBins, spans ? spans[clip].frames : T, rule, bands, times, mids, n_mid, trace ? trace + clip : nullptr, sh, U,
                                                       minmax ? minmax + 2 * clip : nullptr);
        if (tid == 0) labels[clip] = label;
    }
}

hipError_t launch_classify_f64_tail(const double *sxx_bp, const double *sxx_mp, long n_clips, int n, int fs, const ClassifyRuleD &rule,
                                    int *labels, ClassifyTraceD *trace, hipStream_t stream)
{
    const int T = columns_of(n);
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > kMaxColsF64 || n_clips >= (1L << 31)) return hipErrorInvalidValue;
    const double mid_power = 1e-12 * std::pow(10.0, rule.midpoint_db / 10.0);
    hipLaunchKernelGGL(classify_f64_tail_kernel, dim3((unsigned)n_clips), dim3(256), 0, stream, sxx_bp, sxx_mp, n_clips, T, fs, rule, mid_power,
                       band_rows(fs), labels, trace);
    return hipGetLastError();
}

hipError_t launch_classify_f64_midpoints(const int *loud, long n_clips, int n, int fs, double *mids, int *n_mids, int *hits, int *labels,
                                         ClassifyTraceD *trace, hipStream_t stream, unsigned long long *minmax, const ClipSpan *spans)
{
    const int T = columns_of(n);
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > kMaxColsF64 || n_clips >= (1L << 31)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(hits, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(classify_f64_midpoints_kernel, dim3((unsigned)((n_clips + 255) / 256)), dim3(256), 0, stream, loud, n_clips, T, fs, mids, n_mids,
                       hits, labels, trace, minmax, spans);
    return hipGetLastError();
}

hipError_t launch_classify_f64_bands(const double *sxx, const int *hits, long n_clips, int n, int fs, double U, const ClassifyRuleD &rule, const double *mids,
                                     const int *n_mids, int *labels, ClassifyTraceD *trace, hipStream_t stream, const unsigned long long *minmax, const ClipSpan *spans)
{
    const int T = columns_of(n);
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || T > kMaxColsF64) return hipErrorInvalidValue;
    static int cache[64] = {0};
    const int resident = resident_blocks_of(classify_f64_bands_kernel, cache);
    const long blocks = std::min<long>(n_clips, 4L * resident);              // the bound: the list's count is read on the device
    hipLaunchKernelGGL(classify_f64_bands_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, sxx, hits, T, fs, U, rule, band_rows(fs), mids, n_mids, labels, trace, minmax, spans);
    return hipGetLastError();
}

}  // namespace dsp
