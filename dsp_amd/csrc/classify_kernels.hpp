// classify_kernels.hpp -- launch interface of classify_kernels.hip (Butterworth IIR,
// scipy-default spectrogram, scrub-jay rule; reference sync/lib/classifier.cpp and
// donut-classifier/classifier.c).
#pragma once

#include <hip/hip_runtime_api.h>

#include "clip_span.hpp"

namespace dsp {

constexpr int kSpecSeg = 256;    // nperseg            classifier.cpp:223
constexpr int kSpecHop = 224;    // nperseg - nperseg/8 classifier.cpp:224-225
constexpr int kSpecBins = 129;   // nfft/2 + 1         classifier.cpp:235
constexpr int kMaxMidpoints = 64;
// Restart states (delay lines) iir2_ckpt_kernel stores per spectrogram segment and filter: at the segment start AND at its middle
// (sample 128), so the recompute of a segment runs as TWO 128-sample chains on two wavefronts instead of one 256-sample chain on
// one (spec_from_ckpt_kernel, phase R).  Per filter, measured on 49 152 clips (profiles/r03_classify_session2.txt): both filters
// 2.485 ms, the 3000-7500 Hz filter only 2.521 ms, neither 2.586 ms; each second state is 2.3 KB more traffic per clip.
// DSP_CK_HALF = 0 / DSP_CK_HALF_MP = 0: segment starts only (A/B builds).
#ifndef DSP_CK_HALF
#define DSP_CK_HALF 1
#endif
#ifndef DSP_CK_HALF_MP
#define DSP_CK_HALF_MP DSP_CK_HALF
#endif
constexpr int kCkPerSegBp = DSP_CK_HALF ? 2 : 1;       // 3000-7500 Hz (the map of the clips with midpoints)
constexpr int kCkPerSegMp = DSP_CK_HALF_MP ? 2 : 1;    // 1000-3000 Hz (the flags of the gated-in frames)

struct SpecTables {
    float window[kSpecSeg];      // periodic Tukey(0.25), evaluated on the host like classifier.cpp:259-293
    float U;                     // fs * sum(window^2), classifier.cpp:296-301
    // PlainFFT twiddles: (u1,u2) for level l (8 levels) and column m < 2^l, produced by the
    // reference's own recurrence (PlainFFT.cpp:52-84) so every butterfly sees the same bits
    float tw_re[255], tw_im[255];   // level l starts at (1<<l) - 1
    float mp_keep_min;              // smallest float s with (float)(10 log10(s / 1e-12)) > 70 (filled on the device)
    int trivial_first_levels;       // 1 when the twiddles of FFT levels 0 and 1 are exactly (1,0), (1,0), (0,-1) (they are)
    // energy gate (iir2_split_kernel -> spectrogram_kernel<SPEC_FLAGS>): squared taper of the first / last 32 samples of
    // a segment, sum of the squared window, 2 N / U with 1 % margin, 1 when the window is exactly 1 in between
    float win2_in[kSpecSeg - kSpecHop], win2_out[kSpecSeg - kSpecHop];
    float win2_sum, gate_scale;
    int gate_ok;
    // PSD = |X|^2 / U is an IEEE division in the reference (classifier.cpp:350-365), ten instructions on this hardware, three per
    // transformed frame.  rU = the float nearest to 1 / U; div_fast = 1 when launch_spec_div_verify found q + fma(-q, U, p) rU with
    // q = p rU equal to p / U, bit for bit, for EVERY float p in [kDivFastLo, kDivFastHi] (exhaustive: ~1e9 values, on the device,
    // once per context) -- then the recompute kernel takes those three instructions for frames whose cells all lie in that range and
    // the division for the others.  0 (tables built on the host, any table that was not verified): always the division.
    float rU;
    int div_fast;
};
constexpr float kDivFastLo = 8.673617379884035e-19f, kDivFastHi = 1.152921504606847e18f;      // 2^-60, 2^60

struct IirCoef { float b[9], a[9]; };
struct IirCoefD { double b[9], a[9]; };

// y[c][i] for n_clips clips of n samples (row stride `stride` floats), direct form II from
// zero state per clip.  Two filters are run in one pass over x when y2 != nullptr.
// With y2 and means1 / means2 != nullptr the lanes also keep the spectrogram's per-segment sequential sums of
// their own outputs (classifier.cpp:329-333: 256 samples from 224 t, added in order) and store the segment
// means means[c][t], so the spectrogram kernel need not walk the samples serially again.
// ystride: row stride of y1 / y2 (0 = the input's); gate_tables_ok: the host's SpecTables::gate_ok (when false, or when the
// launch falls back to a kernel that does not compute the gate, gate2 is filled with "maybe").
hipError_t launch_iir_f32(const float *x, long n_clips, int n, long stride, const IirCoef &c1, float *y1,
                          const IirCoef &c2, float *y2, hipStream_t stream, float *means1 = nullptr, float *means2 = nullptr,
                          const SpecTables *tables = nullptr, int *gate2 = nullptr, long ystride = 0, bool gate_tables_ok = true);
hipError_t launch_iir_f64(const double *x, long n_clips, int n, long stride, const IirCoefD &c, double *y,
                          hipStream_t stream);
// float rows, recurrence in double, one rounding on store (per-frame prefilter of BASELINE config 3)
// both band-pass filters of the float64 classifier over one read of x: one wavefront per filter and 64 clips (rows of y1 / y2: ystride doubles)
hipError_t launch_iir2_f64(const double *x, long n_clips, int n, long stride, long ystride, const IirCoefD &c1, double *y1,
                           const IirCoefD &c2, double *y2, hipStream_t stream);
hipError_t launch_iir_f64_on_f32(const float *x, long n_clips, int n, long stride, const IirCoefD &c, float *y,
                                 hipStream_t stream);

// classify()'s form of a9: both recurrences in one pass over x, no filtered signal written.  Per clip and spectrogram segment
// k (n_seg = (n-256)/224+1): ck_bp [c][k][kCkPerSegBp][8], ck_mp [c][k][kCkPerSegMp][8] = the filter's delay line v[224k-1 .. 224k-8] (restart state of
// the segment) and, as the second entry, v[224k+127 .. 224k+120] (restart state of its second half), means_mp[c][k] = mean of the 1000-3000 Hz output over the segment (classifier.cpp:329-333), want_mp = work list
// (want_mp[0] = count, then frame numbers c * n_seg + k, 1 + n_clips * n_seg ints) of the segments whose energy does NOT prove
// that every PSD cell stays below SpecTables::mp_keep_min -- the only ones the flag spectrogram has to transform.
// simd_load (optional, kSimdLoadCus * kSimdLoadStride ints, zeroed by the launch): per (XCC, SE, SH, CU) the waves of this launch on
// each SIMD + a block counter; lets every block put its taps wave on its CU's most loaded SIMD (scheduling only, see the kernel).
constexpr int kSimdLoadCus = 4096, kSimdLoadStride = 8;
// in_kind: 0 = float samples, 1 = int16 mono (s / 32768), 2 = interleaved int16 stereo channel 0, 3 = stereo (L + R) / 65536; stride
// counts samples per channel; converted in the kernels' loads (exact: the float path's bits)
hipError_t launch_iir2_ckpt(const void *x, long n_clips, int n, long stride, const IirCoef &c_bp, const IirCoef &c_mp,
                            float *ck_bp, float *ck_mp, float *means_mp, int *want_mp, const SpecTables *tables, hipStream_t stream,
                            int *simd_load = nullptr, int in_kind = 0, const ClipSpan *spans = nullptr, long total = 0);
// RAGGED batches (spans != nullptr; every launcher of the classify() pipeline takes them): clip c starts at spans[c].off samples from x
// and has spans[c].frames whole segments; n is then the LONGEST clip's length (the workspaces keep the uniform [clip][T(n)] layout, a
// shorter clip uses the head of its rows), stride is unused, and `total` = samples in the buffer (loads past it read as zero).
// Spectrogram of segments recomputed from those checkpoints (filter c, checkpoints ck: ck_mp's layout for flags = true, ck_bp's
// otherwise).  flags = true: out = int loud[c][T]
// (1 = some cell >= mp_keep_min; 0 for every segment not on `wantlist`), means = means_mp; flags = false: out = PSD
// [c][T][129] of every segment of the clips on the work list `hits` (means computed here) -- with need / minmax (optional, flags =
// false): only the rows with need[c][t] != 0 are stored, and minmax[c][2] receives the float bits of the smallest / largest positive
// cell over ALL transformed rows of clip c (atomicMin / atomicMax: the caller resets them to +inf / 0; launch_classify_midpoints does).
hipError_t launch_spec_from_ckpt(const void *x, long n_clips, int n, long stride, const IirCoef &c, const float *ck, const float *means,
                                 const int *wantlist, const int *hits, const SpecTables *tables, float *out, bool flags, hipStream_t stream,
                                 const int *need = nullptr, unsigned *minmax = nullptr, int in_kind = 0, const ClipSpan *spans = nullptr);

// sxx[c][129][T] (T = (n-256)/224+1) of clip rows y[c][0..n)
struct ClassifyTrace {           // per clip, for parity tests
    int n_midpoints;
    float midpoints[kMaxMidpoints];
    float sums[kMaxMidpoints][3];
};

// means (optional): segment means [c][T] already computed by launch_iir_f32.  hits (optional, device): work list
// (hits[0] = count, then clip numbers) of the clips whose map is wanted; the maps of the other clips are not written.
// frame_major: sxx[c][T][129] instead of the reference's [c][129][T] -- the layout launch_classify_midpoints / _bands read.
hipError_t launch_spectrogram_f32(const float *y, long n_clips, int n, long stride, const SpecTables *tables,
                                  float *sxx, hipStream_t stream, const float *means = nullptr, const int *hits = nullptr,
                                  bool frame_major = false);


// float64 twin (donut-classifier/classifier.c:448-592): sxx[c][129][T] in double, direct DFT per frame (tolerance parity: FFTW
// is unvendored)
hipError_t launch_spectrogram_f64(const double *y, long n_clips, int n, long stride, int fs, double *sxx, hipStream_t stream);

// The 1000-3000 Hz spectrogram reduced to what find_midpoints reads from it (classifier.cpp:457-518): loud[c][T] = 1 for
// the time bins with a cell above 70 dB.  The map itself is not written.
hipError_t launch_spectrogram_flags(const float *y, long n_clips, int n, long stride, const SpecTables *tables, int *loud,
                                    hipStream_t stream, const float *means = nullptr, const int *gate = nullptr);

// classify() after the spectrograms (classifier.cpp:35-135), two kernels: midpoints from the loud time bins (records in
// `trace`, label 0 when there are none), then the band sums + rule from the frame_major 3000-7500 Hz map for the clips that
// have midpoints (sxx_bp is overwritten with its dB map when it does not fit LDS).  `trace` is required (it carries the
// midpoints); `hits` (1 + n_clips ints) is the work list between the two: hits[0] = clips with midpoints, then their numbers.
// full_records = false: only what the band kernel reads (count + midpoints) is written, not the zero-filled remainder of the
// 1 KB record (callers that return labels only).
// minmax (optional, 2 x n_clips words): the rows of loud[] of the clips with midpoints are REWRITTEN as need[c][t] = "a band window of
// one of the clip's midpoints covers time bin t", and minmax[c] is reset (+inf, 0) for launch_spec_from_ckpt's atomics; pass both to
// launch_spec_from_ckpt and launch_classify_bands: only the needed rows of the map are then stored and read.
hipError_t launch_classify_midpoints(int *loud, long n_clips, int n, int fs, int *labels, ClassifyTrace *trace, int *hits,
                                     hipStream_t stream, bool full_records = true, unsigned *minmax = nullptr, const ClipSpan *spans = nullptr);
// the thresholds of classify() the reference's variants differ in (dsp_classify_config): keep band of the normalised dB
// map (classifier.cpp:67-68) and the rule middle < . && above > . && below > . (classifier.cpp:109)
struct ClassifyRule { float keep_lo, keep_hi, middle_max, above_min, below_min; };
hipError_t launch_classify_bands(float *sxx_bp, long n_clips, int n, int fs, int *labels, ClassifyTrace *trace, const int *hits,
                                 hipStream_t stream, const ClassifyRule &rule, const int *need = nullptr, const unsigned *minmax = nullptr,
                                 const ClipSpan *spans = nullptr);
// dst record perm[i] = src record i, i < n (records of rec_bytes, a multiple of 4): how a ragged batch's results, computed in order of
// length, return to the caller's order
hipError_t launch_scatter_records(const void *src, const int *perm, long n, size_t rec_bytes, void *dst, hipStream_t stream);
// Counts, into *mismatches (device, zeroed by the caller), the floats p in [kDivFastLo, kDivFastHi] for which the three-instruction
// form of p / tables->U (see SpecTables::rU) differs from the division: every bit pattern in the range is tried.
hipError_t launch_spec_div_verify(const SpecTables *tables, unsigned long long *mismatches, hipStream_t stream);
// fills tables->mp_keep_min for find_midpoints' lower_threshold_dB (classifier.cpp:436; once per context and threshold)
hipError_t launch_spec_threshold(SpecTables *tables, float threshold_db, hipStream_t stream);

// sum_intense (classifier.cpp:370-431) on a flat db[nf][nt] map in HBM: one wavefront, the reference's index searches and
// its (row, column) order of additions; *out = the sum.
hipError_t launch_sum_intense(float lower, float upper, float half_range, const float *freqs, int nf, const float *times, int nt,
                              const float *db, float midpoint, float *out, hipStream_t stream);

// ---- the float64 classifier of donut-classifier/classifier.c (classify_f64_kernels.hip) ----
struct ClassifyRuleD { double keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min; };
struct ClassifyTraceD {
    int n_midpoints;
    double midpoints[kMaxMidpoints];
    double sums[kMaxMidpoints][3];
};
// Host-built constants of the float64 spectrogram (classifier.c:484-530): the periodic Tukey(0.25) window and U = fs sum win^2,
// evaluated on the host in double in the reference's order; exp(-2 pi i k / 256) for the transform.
struct SpecTablesD {
    double win[kSpecSeg];
    double w_re[kSpecSeg / 2], w_im[kSpecSeg / 2];
    double U;
};
void build_spec_tables_f64(int fs, SpecTablesD &t);
// compute_spectrogram (classifier.c:448-592) of a batch for the classifier: one wavefront per frame, 256-point real transform as a
// 128-point complex Stockham FFT through LDS (launch_spectrogram_f64 is the [129][T] direct DFT behind dsp_compute_spectrogram_f64
// and the yardstick of this one in the tests).  y rows must be 16-byte aligned (stride even).
//   _flags   every frame of every clip; loud[c][t] = a cell of the frame is above midpoint_db (find_midpoints :679-745); no map
//   _listed  the clips on the work list hits (hits[0] entries, then clip numbers): sxx[entry][t][129] = U * PSD (SpecTablesD::U)
hipError_t launch_spectrogram_f64_flags(const double *y, long n_clips, int n, long stride, const SpecTablesD *tables, double midpoint_db,
                                        int *loud, hipStream_t stream);
hipError_t launch_spectrogram_f64_listed(const double *y, long n_clips, int n, long stride, const SpecTablesD *tables, const int *hits,
                                         double *sxx, hipStream_t stream);
// find_midpoints' clusters (:747-800) from loud[c][t]: mids[c][kMaxMidpoints], n_mids[c]; clips with midpoints on the work list
// hits[1 + n_clips] (reset here), label 0 for the others; trace (optional): midpoints, sums zeroed
// minmax (optional, 2 x n_clips words): reset to (+inf, 0) for the clips with midpoints; launch_spec_f64_listed_from_ckpt then keeps the
// smallest / largest positive cell of each listed clip's map there (double bits, atomicMin / atomicMax) and launch_classify_f64_bands reads
// them instead of scanning the map
hipError_t launch_classify_f64_midpoints(const int *loud, long n_clips, int n, int fs, double *mids, int *n_mids, int *hits, int *labels,
                                         ClassifyTraceD *trace, hipStream_t stream, unsigned long long *minmax = nullptr, const ClipSpan *spans = nullptr);
// band sums and rule (:105-190) of the listed clips over their frame-major U * PSD maps: labels[c], trace sums
hipError_t launch_classify_f64_bands(const double *sxx, const int *hits, long n_clips, int n, int fs, double U, const ClassifyRuleD &rule, const double *mids,
                                     const int *n_mids, int *labels, ClassifyTraceD *trace, hipStream_t stream, const unsigned long long *minmax = nullptr,
                                     const ClipSpan *spans = nullptr);
// the whole tail in one kernel per clip over [c][129][T] float64 PSD maps of the 3000-7500 Hz / 1000-3000 Hz filtered clips
// (launch_spectrogram_f64): labels[c] = the rule's verdict (classifier.c:184), trace (optional): midpoints and band sums per clip
hipError_t launch_classify_f64_tail(const double *sxx_bp, const double *sxx_mp, long n_clips, int n, int fs, const ClassifyRuleD &rule,
                                    int *labels, ClassifyTraceD *trace, hipStream_t stream);

// ---- the same classifier without materialised filter outputs (classify_f64_ckpt_kernels.hip; the default pipeline) ----
// Restart states of a Butterworth filter: its delay line v[n-1 .. n-8] at offsets 0, 64, 128, 192 of every spectrogram segment,
// ck[t][quarter][clip][8] doubles (the clip index innermost among the rows: a wavefront of 64 clips stores 4 KB contiguous).
constexpr int kCkStrideF64 = 64, kCkPerSegF64 = 4;
// Device-resident constants of the screening pass: the windowed DFT matrix of bins 0 .. 63 and 128 as two bf16 tables in MFMA A-fragment
// order [k-step][32-row block][hi / lo][lane][8] (rows 2 q, 2 q + 1 = real, imaginary part of bin q; row 1 = bin 128), the window's own
// transform at those bins, the squared taper and sum w^2 (rounded up: they enter upper bounds).
struct ScreenTablesD {
    unsigned short a_tab[16][4][2][64][8];
    float what_re[64], what_im[64], what128;
    float win2_in[kSpecSeg - kSpecHop], win2_out[kSpecSeg - kSpecHop], win2_sum;
};
bool build_screen_tables_f64(const SpecTablesD &spec, int fs, ScreenTablesD &t);      // false: the window is not 1 between its tapers
// Input kinds of the float64 classifier's kernels: 0 = float64 samples, 1 = int16 mono (s / 32768), 2 = interleaved int16 stereo, channel 0,
// 3 = interleaved int16 stereo, (L + R) / 65536.  stride counts samples (per channel).
// One pass over x: restart states of both filters (ck_bp, ck_mp: [T][4][n_clips][8]) and loud[c][t] = 0 quiet / 1 loud / 2 undecided for every
// segment of the 1000-3000 Hz output; the undecided ones also on the work list want (want[0] = count, then frame numbers c * T + t;
// 1 + n_clips * T ints).  guard: the relative half-width (in power) of the band around the threshold that is always left to the float64
// transform (>= 2e-9).  At most f64_screen_blocks_per_pass() * 64 clips per launch (every block must be resident).
// cu_table (optional, kSimdLoadCus ints, zeroed by the launch): per CU the blocks that have arrived -- spreads the roles over the SIMDs.
int f64_screen_blocks_per_pass();
hipError_t launch_iir2_screen_f64(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c_bp, const IirCoefD &c_mp,
                                  double *ck_bp, double *ck_mp, const ScreenTablesD *tables, double U, double midpoint_db, double guard,
                                  int *loud, int *want, int *cu_table, hipStream_t stream, const ClipSpan *spans = nullptr, long total = 0);
// (spans != nullptr, here and below: a ragged batch -- clip c at spans[c].off samples from x with spans[c].frames segments; n = the longest
// clip, whose segment count is the row length of loud / the restart states / the maps; total = samples in the buffer)
// the float64 verdict on the listed (undecided) segments, recomputed from ck_mp: loud[frame] = 0 / 1
hipError_t launch_spec_f64_recheck(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c_mp, const double *ck_mp,
                                   const SpecTablesD *tables, const int *want, double midpoint_db, double guard, int *loud, hipStream_t stream,
                                   const ClipSpan *spans = nullptr);
// sxx[entry][t][129] = U * PSD of the 3000-7500 Hz output of the clips on the work list hits, recomputed from ck_bp
hipError_t launch_spec_f64_listed_from_ckpt(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c_bp, const double *ck_bp,
                                            const SpecTablesD *tables, const int *hits, double *sxx, hipStream_t stream, unsigned long long *minmax = nullptr,
                                            const ClipSpan *spans = nullptr);
// DSP_AMD_F64_GUARD (read per call; tests): the half-width of the band around the threshold inside which the reference's own expression
// decides, default 2e-9
double f64_threshold_guard();

void build_spec_tables(int fs, SpecTables &t);

}  // namespace dsp
