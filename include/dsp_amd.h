/*
 * dsp_amd.h -- C ABI of libdsp_amd.so: the MI355X (gfx950) MFCC / Butterworth /
 * spectrogram hot path of cornell-c2s2/dsp behind the reference's own C entry
 * points.  Plain C: pointers and sizes only, no torch / HIP types (streams are
 * passed as void*).
 *
 * Section 1 are the reference's symbols, bit-for-bit the same signatures, so the
 * library links in place of the reference's mfcc.c / classifier.cpp object
 * files.  Section 2 are batch / device-resident extensions the reference does
 * not have; the Section-1 symbols are thin wrappers over them.
 *
 * Reference paths are relative to the upstream repository root.
 * Error convention: Section-1 functions keep the reference's return values
 * (frame count / 0-1 label, 0 on failure) and never change them to signal GPU
 * trouble; the cause is readable through dsp_last_error().  Section-2 functions
 * return >= 0 on success and a negative DSP_E* code on failure.
 */
#ifndef DSP_AMD_H
#define DSP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the functions declared between this push and
 * the matching pop (plus the C++-linkage reference names of dsp_amd_classifier.h) are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* ===================================================================== */
/* 1. Reference entry points (drop-in)                                    */
/* ===================================================================== */

/* Replaces 2fa/audio/word/c/mfcc.h:16-19 (definition mfcc.c:108-232; identical
 * copy 2fa/audio/pico-audio/src/mfcc.c).  Mono float PCM in [-1,1] at 16 kHz
 * -> frame-major out_mfcc[T][13], T = min(max_frames, 1 + (n-400)/160);
 * returns T, 0 when num_samples < 400 or max_frames <= 0 (mfcc.c:117-119).
 * Caller owns both buffers (host memory); out_mfcc holds max_frames*13 floats. */
int compute_mfcc(const float *signal, int num_samples, float *out_mfcc, int max_frames);

/* Replaces sync/lib/classifier.h:19 (definition classifier.cpp:9-136; fp32 firmware
 * twin of donut-classifier/classifier.c:30-212).  16 kHz mono clip -> 1 if the scrub-jay
 * rule fires, else 0 (also 0 on internal failure, classifier.cpp:87-91).  The reference
 * header is C++ without extern "C": this is the C symbol; the C++-linkage `int classify(float*, int)`
 * that sync.cpp links against is exported too and declared in dsp_amd_classifier.h, next to
 * butter_bandpass, butter_bandpass_filter, compute_spectrogram, sum_intense and find_midpoints
 * (classifier.h:14-18).  Caller owns `data` (host memory); nothing is printed unless the
 * environment variable DSP_AMD_VERBOSE is set.                                      */
int dsp_classify(float *data, int data_size);

/* ===================================================================== */
/* 2. Extensions                                                          */
/* ===================================================================== */

enum {
    DSP_OK = 0,
    DSP_EINVAL = -1,      /* bad argument / unsupported configuration      */
    DSP_ENODEV = -2,      /* no usable HIP device                           */
    DSP_EHIP = -3,        /* a HIP runtime call failed (see dsp_last_error) */
    DSP_ENOMEM = -4
};

enum { DSP_WINDOW_HANN = 0, DSP_WINDOW_HAMMING = 1, DSP_WINDOW_RECT = 2 };
enum { DSP_MELNORM_NONE = 0, DSP_MELNORM_SLANEY = 1,       /* HTK mel scale, triangles of peak 1 / of unit area          */
       DSP_MELNORM_LIBROSA = 2,                             /* Slaney mel scale + unit area: librosa.filters.mel defaults */
       DSP_MELNORM_AUBIO_SLANEY = 3 };                      /* aubio_filterbank_set_mel_coeffs_slaney (what new_aubio_mfcc picks for 40
                                                               filters, cepstrum/scrubjay_infer.c:30): Slaney's Auditory-Toolbox bank, 13
                                                               linear + 27 log-spaced unit-area triangles, 133 Hz .. 6.85 kHz; n_mels must
                                                               be 40, fmin / fmax are not used; n_fft 2048                              */
enum { DSP_LOG_PER_FRAME_MAX = 0,    /* mfcc.c:169-206: reference = the frame's own maximum                                    */
       DSP_LOG_GLOBAL_REF1 = 1,      /* librosa power_to_db(ref = 1, top_db below the CLIP's maximum); n_fft 512 and 2048      */
       DSP_LOG_LOG10_FLOOR = 2 };    /* aubio fvec_log10 (aubio_mfcc_do): plain log10 of each filter output, inputs below 2e-42
                                        count as 2e-42; no dB factor, no reference, no top_db; n_fft 2048                      */
enum { DSP_SPECTRUM_POWER = 0,       /* mfcc.c:151-155: |X[k]|^2 into the filterbank                                             */
       DSP_SPECTRUM_MAGNITUDE = 1 }; /* aubio: the phase vocoder's norm |X[k]| (aubio_fft_get_norm), filterbank power 1; n_fft 2048 */
enum { DSP_FRAMING_COMPLETE = 0,     /* mfcc.c:132-139: frames that lie completely inside the clip, T = 1 + (n - frame) / hop    */
       DSP_FRAMING_STREAM = 1 };     /* aubio_source_do + aubio_pvoc_do as cepstrum/scrubjay_infer.c:39-53 drives them: one frame per
                                        hop of NEW samples, T = ceil(n / hop); frame t ends with hop t and starts frame_length - hop
                                        samples earlier (zeros before the clip), the last partial hop is zero padded; clips, n_fft 2048 */
enum { DSP_PREFILTER_NONE = 0, DSP_PREFILTER_BUTTER_1000_3000 = 1, DSP_PREFILTER_BUTTER_3000_7500 = 2 };

/* Compile-time constants of the reference (mfcc_params.h:6-12, mfcc.c:172-173)
 * turned into a POD; dsp_mfcc_default_config() fills in the reference values. */
typedef struct dsp_mfcc_config {
    int sample_rate;  /* 16000 */
    int n_fft;        /* 512   (supported: 512, 1024, 2048; one wavefront per frame)  */
    int frame_length; /* 400   (<= n_fft) */
    int hop_length;   /* 160 */
    int n_mels;       /* 40 */
    int n_mfcc;       /* 13 */
    int window;       /* DSP_WINDOW_*  (reference: periodic Hann, export_mfcc_params.py:46) */
    int mel_norm;     /* DSP_MELNORM_* (reference: none, export_mfcc_params.py:56) */
    int log_mode;     /* DSP_LOG_*     (reference: per-frame max, mfcc.c:169-206) */
    int prefilter;    /* DSP_PREFILTER_* fp64 Butterworth per frame from zero state */
    int win_length;   /* 0 = frame_length; else window taps centred in the frame (librosa win_length < n_fft) */
    float fmin, fmax; /* 0, 8000 */
    float amin;       /* 1e-10 */
    float top_db;     /* 80 */
    int spectrum;     /* DSP_SPECTRUM_* (reference: power) */
    int framing;      /* DSP_FRAMING_*  (reference: complete frames only) */
} dsp_mfcc_config;

void dsp_mfcc_default_config(dsp_mfcc_config *cfg);
/* The parameterisation of cepstrum/scrubjay_infer.c:9-13,28-30 with aubio 0.4's semantics for the calls it makes:
 * WIN_SIZE 2048 / HOP_SIZE 1024 streaming frames (DSP_FRAMING_STREAM), "hanningz" = periodic Hann, magnitude spectrum,
 * the 40-filter Slaney bank, log10, orthonormal DCT-II, 20 coefficients.  sample_rate: the file's own (aubio_source with
 * samplerate 0).  aubio is an unvendored dependency of the reference: restated from its published algorithm, parity unpinned. */
void dsp_mfcc_scrubjay_infer_config(dsp_mfcc_config *cfg, int sample_rate);

typedef struct dsp_mfcc_plan dsp_mfcc_plan; /* opaque: device tables for one config on one GPU */

/* Builds the constant tables (window, FFT twiddles, sparse HTK-mel chunks,
 * DCT-II basis; formulas of 2fa/audio/word/python/export_mfcc_params.py:27-60)
 * on the host and uploads them to `device`.                                   */
int dsp_mfcc_plan_create(const dsp_mfcc_config *cfg, int device, dsp_mfcc_plan **out);
void dsp_mfcc_plan_destroy(dsp_mfcc_plan *plan);
int dsp_mfcc_plan_config(const dsp_mfcc_plan *plan, dsp_mfcc_config *cfg);

/* Number of frames compute_mfcc produces for an n-sample clip (mfcc.c:132-139). */
int dsp_mfcc_frames_for(const dsp_mfcc_config *cfg, int num_samples, int max_frames);

/* --- device-resident entry points: pointers are HBM addresses on the plan's
 * device, work is enqueued on `stream` (a hipStream_t, NULL = default stream)
 * and the call returns without synchronising.                                 */

/* n_frames independent frames, d_frames[n_frames][frame_length] back to back
 * -> d_out[n_frames][n_mfcc]  (BASELINE configs 2/3).                          */
int dsp_mfcc_frames_device(dsp_mfcc_plan *plan, const float *d_frames, long n_frames,
                           float *d_out, void *stream);

/* n_clips clips of samples_per_clip floats, clip c starting at
 * d_signal + c*clip_stride; each framed like compute_mfcc (frame_length/hop)
 * and capped at max_frames -> d_out[n_clips][T][n_mfcc]; returns T.           */
int dsp_mfcc_clips_device(dsp_mfcc_plan *plan, const float *d_signal, long n_clips,
                          int samples_per_clip, long clip_stride, float *d_out,
                          int max_frames, void *stream);

/* fft_real_forward (2fa/audio/word/c/mfcc.c:16-95; non-static there, so callers may link it): 400 real samples, zero-padded to 512,
 * forward transform, all 512 complex bins interleaved [re0, im0, re1, im1, ...].  Host pointers; void like the reference (a failure:
 * zeros out, reason in dsp_last_error()).  dsp_fft_real_forward_host: the same for a batch and any power-of-two n_fft <= 4096.    */
void fft_real_forward(const float *in_time, float *out_freq);
int dsp_fft_real_forward_host(const float *in_time, long n_frames, int frame_length, long in_stride, int n_fft, float *out_freq);

/* PCM16 ingestion on the device: the same framing on interleaved int16 PCM,
 * converted in the load exactly like the reference's WAV readers: mono s/32768
 * (2fa/audio/word/c/main_test.c:198-203); stereo channel 0 (donut-classifier/classifier.c:292-297)
 * or the channel average 0.5(L/32768 + R/32768) (main_test.c:205-217).  samples_per_clip and
 * clip_stride count samples PER CHANNEL.  Halves (mono) the HBM bytes of the float path.
 * Plans: n_fft 512 (per-frame log mode) and dsp_mfcc_scrubjay_infer_config (n_fft 2048).     */
enum { DSP_STEREO_CHANNEL0 = 0, DSP_STEREO_AVERAGE = 1 };
int dsp_mfcc_clips_pcm16_device(dsp_mfcc_plan *plan, const int16_t *d_pcm, long n_clips, int samples_per_clip,
                                long clip_stride, int channels, int stereo_mode, float *d_out, int max_frames,
                                void *stream);

/* --- host-pointer conveniences: copy in, run, copy out, synchronise. -------- */
int dsp_mfcc_frames_host(dsp_mfcc_plan *plan, const float *frames, long n_frames, float *out);
int dsp_mfcc_clips_host(dsp_mfcc_plan *plan, const float *signal, long n_clips,
                        int samples_per_clip, long clip_stride, float *out, int max_frames);

/* Launch geometry knobs for tuning / profiling (0 = library default). */
int dsp_mfcc_plan_set_launch(dsp_mfcc_plan *plan, int blocks_per_cu, int frames_per_chunk);
/* Kernel form: DSP_KERNEL_WAVE = one wavefront per frame, log + DCT once per 16-frame tile
 * (default); DSP_KERNEL_ROW = one 16-lane row per frame (4 frames per wavefront);
 * DSP_KERNEL_WAVE_FRAME = one wavefront per frame with the per-frame log + DCT epilogue (what
 * DSP_LOG_GLOBAL_REF1 plans always run); DSP_KERNEL_PAIR = two frames per wavefront step for independent
 * 512-sample frames of the reference shape (an experiment, slower than the default: DESIGN.md 3; other
 * shapes and clip mode run the default form).  Same results to rounding; DESIGN.md has the numbers. */
enum { DSP_KERNEL_WAVE = 0, DSP_KERNEL_ROW = 1, DSP_KERNEL_WAVE_FRAME = 2, DSP_KERNEL_PAIR = 3 };
int dsp_mfcc_plan_set_kernel(dsp_mfcc_plan *plan, int kernel);

/* --- Butterworth band-pass (donut-classifier/classifier.c:319-446) ---------- */

/* Literal 9-tap tables of classifier.c:342-360 / 383-401; returns 1, or 0 for
 * any other band (the reference prints "invalid bandpass range").              */
int dsp_butter_bandpass(double lowcut, double highcut, double *b, double *a);

/* butter_bandpass_filter (sync/lib/classifier.cpp:193-219 fp32, donut-classifier/
 * classifier.c:420-446 fp64): direct form II from zero state over n_clips rows of n
 * samples (row stride in elements), bit-identical to the reference's operation order.
 * Host pointers; b and a hold 9 taps each.                                        */
int dsp_butter_bandpass_filter_f32(const float *data, long n_clips, int n, long stride,
                                   const float *b, const float *a, float *output);
int dsp_butter_bandpass_filter_f64(const double *data, long n_clips, int n, long stride,
                                   const double *b, const double *a, double *output);

/* compute_spectrogram (sync/lib/classifier.cpp:221-368): nperseg 256, hop 224, detrend,
 * periodic Tukey(0.25), PSD.  Flat outputs instead of the reference's malloc'd rows:
 * frequencies[129], times[T], sxx[129][T]; returns T = (n-256)/224+1 (0 if n < 256).
 * Host pointers; frequencies / times may be NULL.                                 */
int dsp_compute_spectrogram_f32(const float *signal, int signal_length, int fs,
                                float *frequencies, float *times, float *sxx);
/* The float64 twin, compute_spectrogram of donut-classifier/classifier.c:448-592 (FFTW r2c there; here a float64 DFT per
 * frame, the same transform -- parity by tolerance, FFTW is unvendored).  Same flat layout in double.                  */
int dsp_compute_spectrogram_f64(const double *signal, int signal_length, int fs,
                                double *frequencies, double *times, double *sxx);

/* Per-clip trace of classify() for parity tests: midpoints (classifier.cpp:433-598) and
 * the three band sums per midpoint (classifier.cpp:99-101).                       */
typedef struct dsp_classify_trace {
    int n_midpoints;
    float midpoints[64];
    float sums[64][3];   /* above (5-7 kHz), middle (2.5-5 kHz), below (0.5-2.5 kHz); rows after the
                          * first midpoint that fires the rule are 0 (the reference stops there)  */
} dsp_classify_trace;

/* classify() over n_clips clips of n samples (row stride in floats).  labels[n_clips];
 * trace may be NULL.  _host: host pointers, blocking.  _device: HBM pointers, STREAM-ORDERED: the call returns once
 * its work is enqueued on `stream`; the labels are valid when the stream reaches that point.  The library keeps one
 * grow-only workspace per device (42 KB per one-second clip of the largest pass; dsp_classify_release frees it): calls
 * on one device are ordered one behind the other whatever their streams, calls on different devices share nothing.  */
int dsp_classify_batch_host(const float *signal, long n_clips, int n, long stride, int *labels,
                            dsp_classify_trace *trace);
int dsp_classify_batch_device(const float *d_signal, long n_clips, int n, long stride, int *d_labels,
                              void *stream);

/* The thresholds the reference's variants of classify() hard-code, as a POD (everything else in
 * those files is identical, `diff sync/lib/classifier.cpp microphone/src/classifier.cpp`):
 *                                   keep band      midpoint dB   rule  middle <  above >  below >
 *   sync/lib/classifier.cpp         0.65 / 0.80    70            100 / 200 / 80    (:67-68, :436, :109)  default
 *   microphone/src/classifier.cpp   0.70 / 0.85    45            100 / 200 / 150   (:79-80, :448, :123)
 *   microphone/src/classifier.c     0.70 / 0.85    45            50 / 200 / 200    (:120-121, :608, :164)   float64 file
 *   donut-classifier/classifier.c   0.70 / 0.85    45            75 / 300 / 100    (:141-142, :660, :184)   float64 file
 * (the two float64 files: their THRESHOLDS on this library's fp32 arithmetic, which is sync/lib's)
 * The _cfg entry points take a NULL cfg for the default set.                          */
typedef struct dsp_classify_config {
    float keep_lo, keep_hi;        /* normalised-dB band kept in the 3000-7500 Hz map            */
    float midpoint_db;             /* lower_threshold_dB of find_midpoints                        */
    float middle_max, above_min, below_min;   /* sum_middle < . && sum_above > . && sum_below > . */
} dsp_classify_config;
void dsp_classify_default_config(dsp_classify_config *cfg);
int dsp_classify_batch_host_cfg(const dsp_classify_config *cfg, const float *signal, long n_clips, int n, long stride,
                                int *labels, dsp_classify_trace *trace);
int dsp_classify_batch_device_cfg(const dsp_classify_config *cfg, const float *d_signal, long n_clips, int n, long stride,
                                  int *d_labels, void *stream);
/* The same on int16 PCM, mono or interleaved stereo (DSP_STEREO_CHANNEL0 / DSP_STEREO_AVERAGE as dsp_mfcc_clips_pcm16_device),
 * converted in the kernels' loads exactly like the reference's capture loop and readers (sync/sync.cpp:237-242 pcmSample / 32768.0,
 * donut-classifier/classifier.c:55-59, :286-297; the average as main_test.c:205-217): bit-identical to the float entry points on the
 * same samples at half the input bytes.  n and stride count samples PER CHANNEL.                                              */
int dsp_classify_batch_pcm16_host(const dsp_classify_config *cfg, const int16_t *pcm, long n_clips, int n, long stride, int channels,
                                  int stereo_mode, int *labels, dsp_classify_trace *trace);
int dsp_classify_batch_pcm16_device(const dsp_classify_config *cfg, const int16_t *d_pcm, long n_clips, int n, long stride, int channels,
                                    int stereo_mode, int *d_labels, void *stream);
/* RAGGED batches: clips of different lengths in one call (donut-classifier/classifier.c:286-297 reads a file of any length; its callers
 * loop over files).  offsets is a HOST array of n_clips + 1 sample positions (per channel) into the buffer: clip c is
 * [offsets[c], offsets[c + 1]), non-decreasing, any parity; a clip shorter than 256 samples holds no spectrogram segment and gets label
 * 0, one of more than 13.4 s is refused like in the uniform entry points.  Every clip gets the label (and trace record) of a one-clip
 * call on it, bit for bit.  The array is read before the call returns.  _device: the whole buffer in HBM, stream-ordered; _host: the
 * buffer signal[offsets[n_clips]] is copied to the GPU once, trace may be NULL.                                                       */
int dsp_classify_batch_ragged_device(const dsp_classify_config *cfg, const float *d_signal, long n_clips, const long *offsets,
                                     int *d_labels, void *stream);
int dsp_classify_batch_ragged_pcm16_device(const dsp_classify_config *cfg, const int16_t *d_pcm, long n_clips, const long *offsets,
                                           int channels, int stereo_mode, int *d_labels, void *stream);
int dsp_classify_batch_ragged_host(const dsp_classify_config *cfg, const float *signal, long n_clips, const long *offsets, int *labels,
                                   dsp_classify_trace *trace);
int dsp_classify_batch_ragged_pcm16_host(const dsp_classify_config *cfg, const int16_t *pcm, long n_clips, const long *offsets,
                                         int channels, int stereo_mode, int *labels, dsp_classify_trace *trace);
/* What the LAST pass (sub-batch) of the shared context on `device` did (blocks until it has finished): segments of the 1000-3000 Hz
 * output the energy gate left to the flag transform, clips that had midpoints.  Either pointer may be NULL.                           */
int dsp_classify_stats(int device, long *gated_segments, long *listed_clips);
/* A classifier context of the caller's own (tables + workspace on `device`).  The entry points above share one context per device, so
 * two calls on one device run one behind the other; calls through different contexts, on different streams, may overlap.            */
typedef struct dsp_classify_ctx dsp_classify_ctx;
int dsp_classify_ctx_create(int device, dsp_classify_ctx **out);
void dsp_classify_ctx_destroy(dsp_classify_ctx *ctx);
int dsp_classify_batch_device_ctx(dsp_classify_ctx *ctx, const dsp_classify_config *cfg, const float *d_signal, long n_clips, int n,
                                  long stride, int *d_labels, void *stream);
/* test hook, no GPU call: holds the default context of `device` for hold_ms milliseconds (two devices overlap, one device queues) */
int dsp_debug_hold_classify_ctx(int device, int hold_ms);
/* frees the float32 classifier's tables and workspace on `device` (-1: every device) after its pending work has finished */
int dsp_classify_release(int device);

/* The float64 classifier, donut-classifier/classifier.c (the file's per-clip body :83-192, sum_intense :594-653,
 * find_midpoints :655-830): both Butterworth filters, both spectrograms, dB maps, 45 dB midpoints, normalisation, keep band,
 * band sums and rule in double on the GPU.  Thresholds are doubles as in the file; cfg NULL = its own
 * (0.70 / 0.85 :141-142, 45 dB :660, middle < 75 && above > 300 && below > 100 :184).  fs is 16000 (the rate butter_bandpass has
 * coefficients for).  The reference transforms with FFTW (unvendored): the spectrogram is a float64 transform of its own (a
 * 128-point complex FFT per half-wavefront; DSP_AMD_F64_DFT=1: the direct DFT of dsp_compute_spectrogram_f64), so parity is by
 * tolerance (labels and midpoints equal, band sums to ~1e-9 relative).  Neither filtered signal is written to HBM: one pass keeps the
 * filters' restart states (36 KB per one-second clip) and settles the loud time bins with a bounded screening transform; only
 * undecided segments and the clips with midpoints are recomputed and transformed in float64.  The library keeps one grow-only workspace
 * PER DEVICE (112 KB per one-second clip of the largest pass, passes of at most 49 152 clips on 256 CUs; DSP_AMD_F64_SUB_BATCH lowers
 * that; dsp_classify_release_f64 frees it).
 * _host: host pointers, blocking.  _device: HBM pointers (d_trace may be NULL), STREAM-ORDERED: the call returns once its work is
 * enqueued on `stream`; results are valid when the stream reaches that point.  Calls on one device share its workspace and are
 * ordered one behind the other (whatever their streams); calls on different devices share nothing.
 * _pcm16_: int16 PCM, mono or interleaved stereo (channel 0, or the average of the two channels), converted in the kernels' loads
 * exactly like the reference's readers: s / 32768.0 (classifier.c:55-59, channel 0 of a stereo file :286-297; sync/sync.cpp:237-242;
 * the average as main_test.c:205-217) -- bit-identical to the float64 entry points on the same samples, at a quarter of the input bytes.
 * n and stride count samples PER CHANNEL.                                                                                  */
typedef struct dsp_classify_config_f64 {
    double keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min;
} dsp_classify_config_f64;
typedef struct dsp_classify_trace_f64 {
    int n_midpoints;
    double midpoints[64];
    double sums[64][3];
} dsp_classify_trace_f64;
void dsp_classify_default_config_f64(dsp_classify_config_f64 *cfg);
int dsp_classify_batch_host_f64(const dsp_classify_config_f64 *cfg, const double *signal, long n_clips, int n, long stride,
                                int *labels, dsp_classify_trace_f64 *trace);
int dsp_classify_batch_device_f64(const dsp_classify_config_f64 *cfg, const double *d_signal, long n_clips, int n, long stride,
                                  int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream);
int dsp_classify_batch_pcm16_host_f64(const dsp_classify_config_f64 *cfg, const int16_t *pcm, long n_clips, int n, long stride, int channels,
                                      int stereo_mode, int *labels, dsp_classify_trace_f64 *trace);
int dsp_classify_batch_pcm16_device_f64(const dsp_classify_config_f64 *cfg, const int16_t *d_pcm, long n_clips, int n, long stride, int channels,
                                        int stereo_mode, int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream);
/* RAGGED batches (offsets as dsp_classify_batch_ragged_device: a HOST array of n_clips + 1 sample positions per channel): the float64
 * classify() of donut-classifier/classifier.c on clips of different lengths in one call -- the files its reader (:286-297) takes one per
 * run.  Labels and trace records of a one-clip call on every clip, bit for bit.  d_trace / trace may be NULL.                          */
int dsp_classify_batch_ragged_device_f64(const dsp_classify_config_f64 *cfg, const double *d_signal, long n_clips, const long *offsets,
                                         int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream);
int dsp_classify_batch_ragged_pcm16_device_f64(const dsp_classify_config_f64 *cfg, const int16_t *d_pcm, long n_clips, const long *offsets,
                                               int channels, int stereo_mode, int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream);
int dsp_classify_batch_ragged_host_f64(const dsp_classify_config_f64 *cfg, const double *signal, long n_clips, const long *offsets, int *labels,
                                       dsp_classify_trace_f64 *trace);
int dsp_classify_batch_ragged_pcm16_host_f64(const dsp_classify_config_f64 *cfg, const int16_t *pcm, long n_clips, const long *offsets,
                                             int channels, int stereo_mode, int *labels, dsp_classify_trace_f64 *trace);
/* What the LAST pass of the float64 classifier on `device` did (blocks until it has finished): spectrogram segments of the
 * 1000-3000 Hz output, how many of them the screening left to the float64 transform, clips that had midpoints.  Any pointer may be NULL. */
int dsp_classify_stats_f64(int device, long *segments, long *undecided, long *listed_clips);
/* Diagnostic builds of the library only (-DSC_DIAG): the screening kernel's per-block records of the last pass (16 ints per block).  */
int dsp_classify_debug_f64(int device, int *out, int n_ints);
/* frees the float64 classifier's workspace on `device` (-1: on every device) after its pending work has finished */
int dsp_classify_release_f64(int device);

/* sum_intense (sync/lib/classifier.h:17, classifier.cpp:370-431) on a flat matrix: db[freq_bins][time_bins] (NaN = dropped
 * cell), the reference's index searches and its (row, column) summation order.  Host pointers; *out receives the sum.   */
int dsp_sum_intense_f32(float lower, float upper, float half_range, const float *frequencies, int freq_bins,
                        const float *times, int time_bins, const float *db, float midpoint, float *out);

/* find_midpoints (sync/lib/classifier.h:18, classifier.cpp:433-598): the 1000-3000 Hz filter, its spectrogram,
 * time bins above 70 dB, greedy clusters of at least 0.15 s -> their mean times, in seconds.  Host pointers.
 * Returns the number of midpoints (the first max_midpoints are written) or a negative error; fs must be 16000
 * (the only rate butter_bandpass has coefficients for).  Clips are limited to 957 spectrogram columns (13.4 s):
 * a cluster needs 12 columns and a 4-column gap, so no such clip has more than the 64 midpoints a trace record
 * holds; longer clips are rejected (DSP_EINVAL) rather than truncated.                 */
int dsp_find_midpoints(const float *data, int num_frames, int fs, float *midpoints, int max_midpoints);

/* --- several GPUs from one C process ---------------------------------------
 * Clips shard without any exchange: give device d the clips [d * ceil(N / D), ...) through the entry points above (plans, models and
 * classifier contexts are per device; calls on different devices share no lock).  The path's one exchange step is the gather of the
 * per-clip results: one RCCL communicator per listed device (ncclCommInitAll) and ONE grouped all-gather per call over xGMI -- rank r
 * (= devices[r]) contributes bytes_per_rank bytes at d_send[r] and receives every rank's block, in rank order, at d_recv[r]
 * (n_devices * bytes_per_rank bytes; pad the last shard so the counts are equal).  Stream-ordered on streams[r] (NULL: the device's
 * null stream), so the gather of batch k can cross xGMI while batch k + 1 is computed on another stream.  RCCL is loaded at the first
 * dsp_gather_create (dlopen of librccl.so.1): hosts that never gather do not need it.  INTEGRATION.md, "8 GPUs from C".       */
typedef struct dsp_gather dsp_gather;
int dsp_gather_create(const int *devices, int n_devices, dsp_gather **out);
void dsp_gather_destroy(dsp_gather *g);
int dsp_gather_n_devices(const dsp_gather *g);
int dsp_gather_all(dsp_gather *g, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank, void *const *streams);

/* --- pooling + SVM (cepstrum/scrubjay_infer.c:36-66, 105-141; scrubjay_svm.onnx) ------ */

/* mfcc_stats pooling: feat[c][2*n_coef] = per-coefficient mean | population std over the T
 * frames of clip c, float64 accumulators in frame order.  HBM pointers.            */
int dsp_mfcc_stats_device(const float *d_mfcc, long n_clips, int n_frames, int n_coef, float *d_feat, void *stream);

typedef struct dsp_svm dsp_svm;   /* opaque: Scaler + RBF SVMClassifier + Platt on one GPU */
/* Attributes as stored in the ONNX graph (Scaler.offset/scale, SVMClassifier.support_vectors
 * [n_sv][n_features], coefficients[n_sv], kernel_params[0] = gamma, rho[0], prob_a[0], prob_b[0]). */
int dsp_svm_create(int device, int n_features, int n_sv, const float *offset, const float *scale,
                   const float *support_vectors, const float *coefficients, float gamma, float rho,
                   float prob_a, float prob_b, dsp_svm **out);
void dsp_svm_destroy(dsp_svm *svm);
/* labels, decision values and P(label 1) for n_clips feature rows; HBM pointers, d_decision / d_prob1 may be NULL.
 * libsvm's rules (sklearn's SVC is libsvm; pinned by tests/golden/svm_libsvm_ref.npz): decision = sum_i coef_i K(x, sv_i) +
 * rho; label = the pairwise vote, decision > 0 -> class 0 else class 1 (what .predict returns in cepstrum/run.py, and ONNX
 * Runtime's output_label in SVC mode); P(label 1) from svm_predict_probability (Platt pair -> multiclass_probability's
 * iteration, which is NOT the plain sigmoid: exactly 0.5 in a dead zone around decision = 0).                      */
int dsp_svm_predict_device(dsp_svm *svm, const float *d_feat, long n_clips, int *d_labels, float *d_decision,
                           float *d_prob1, void *stream);

/* BASELINE config 5 in ONE kernel: clip -> MFCC(n_mfcc) -> mean | std -> Scaler -> RBF-SVM -> label, the MFCC
 * matrix never leaves the chip (one wavefront walks one clip; pooling in the kernel's tile epilogue).  The plan's
 * 2 * n_mfcc must equal the SVM's n_features (<= 64) and, at n_fft = 512, the SVM may have at most 2048 support vectors (its
 * coefficients ride in the kernel's LDS; larger models take the three calls); equal results to dsp_mfcc_clips_device +
 * dsp_mfcc_stats_device + dsp_svm_predict_device.  Plans with n_fft = 512 (BASELINE config 5) or n_fft = 2048 (the framing
 * of scrubjay_infer.c:10-14 itself: 2048 / 1024 / 40 filters / 20 coefficients).  d_decision, d_prob1, d_feat ([n_clips][2 n_mfcc]) may be NULL. */
int dsp_scrubjay_fused_device(dsp_mfcc_plan *plan, dsp_svm *svm, const float *d_signal, long n_clips,
                              int samples_per_clip, long clip_stride, int max_frames, int *d_labels,
                              float *d_decision, float *d_prob1, float *d_feat, void *stream);
/* The same from int16 PCM (mono / interleaved stereo, as dsp_mfcc_clips_pcm16_device), converted in the kernel's load: bit-identical
 * to the float entry point on the same samples.  Plans of the reference framing (n_fft 512, frame 400, 40 mel filters) and of
 * dsp_mfcc_scrubjay_infer_config (the 2048-point front end cepstrum/scrubjay_infer.c itself runs).                                 */
int dsp_scrubjay_fused_pcm16_device(dsp_mfcc_plan *plan, dsp_svm *svm, const int16_t *d_pcm, long n_clips, int samples_per_clip,
                                    long clip_stride, int channels, int stereo_mode, int max_frames, int *d_labels,
                                    float *d_decision, float *d_prob1, float *d_feat, void *stream);
/* RAGGED batches -- clips of different lengths in ONE launch (the reference's callers loop over files: cepstrum/scrubjay_infer.c:158-176,
 * 2fa/audio/word/c/main_test.c:254-331).  offsets is a HOST array of n_clips + 1 sample positions (per channel) into the device buffer:
 * clip c is [offsets[c], offsets[c + 1]), non-decreasing, any parity (the buffer itself 8-byte aligned, 4 for mono int16); every clip
 * must hold at least one frame (DSP_EINVAL names the first that does not).  Clip c gets the frames ITS length gives
 * (dsp_mfcc_frames_for(cfg, offsets[c + 1] - offsets[c], max_frames)) and the results of a one-clip call on it, bit for bit.  The array
 * is read before the call returns.  Returns the frame count of the longest clip.                                                        */
int dsp_scrubjay_fused_ragged_device(dsp_mfcc_plan *plan, dsp_svm *svm, const float *d_signal, long n_clips, const long *offsets,
                                     int max_frames, int *d_labels, float *d_decision, float *d_prob1, float *d_feat, void *stream);
int dsp_scrubjay_fused_ragged_pcm16_device(dsp_mfcc_plan *plan, dsp_svm *svm, const int16_t *d_pcm, long n_clips, const long *offsets,
                                           int channels, int stereo_mode, int max_frames, int *d_labels, float *d_decision,
                                           float *d_prob1, float *d_feat, void *stream);

/* Test hook, host only (no GPU call): the order a ragged batch of the fused clip kernels runs in.  out4[4 * pos .. + 3] = start, samples,
 * frames, caller's index of the clip at position pos; wavefront w of n_waves walks positions w, w + n_waves, ...  Returns the longest
 * clip's frame count.                                                                                                                 */
int dsp_debug_fused_spans(const dsp_mfcc_config *cfg, const long *offsets, long n_clips, int max_frames, long n_waves, long *out4);

/* --- consumers of the MFCC matrix and the resampler (8f-4) ------- */

/* The stop-word net behind classify_signal (2fa/audio/word/c/stop_detector.h:10,
 * stop_detector.c:12-55; audio_classifier_inference.c:38-90).  The trained parameters are the
 * arrays of the reference's model_params.h, handed over by the caller (never compiled in here). */
typedef struct dsp_stop_model dsp_stop_model;
typedef struct dsp_stop_model_params {
    int n_coef;                  /* 13   MFCC_N_MFCC                              */
    int max_frames;              /* 500  MAX_FRAMES (stop_detector.c:9)           */
    int units[4];                /* 4, 2, 2, 1  DENSE1..4_UNITS (model_params.h:7-10), each <= 16, last = 1 */
    const float *scaler_mean;    /* SCALER_MEAN  [n_coef * max_frames], coefficient-major */
    const float *scaler_scale;   /* SCALER_SCALE [n_coef * max_frames]            */
    const float *kernel[4];      /* DENSEn_KERNEL, (in, out) row-major             */
    const float *bias[4];        /* DENSEn_BIAS                                    */
} dsp_stop_model_params;
int dsp_stop_model_create(const dsp_stop_model_params *params, int device, dsp_stop_model **out);
void dsp_stop_model_destroy(dsp_stop_model *model);
/* audio_classifier_predict over a batch: d_mfcc[n_clips][frames_per_clip][n_coef] frame-major (what
 * compute_mfcc writes), viewed coefficient-major and zero-padded / truncated at max_frames as
 * stop_detector.c:36-50 does; d_prob[n_clips] = P("stop").  HBM pointers.                       */
int dsp_stop_predict_device(dsp_stop_model *model, const float *d_mfcc, long n_clips, int frames_per_clip,
                            float *d_prob, void *stream);
/* classify_signal over a batch of equally long clips resident in HBM: plan (reference defaults,
 * n_mfcc = n_coef) -> MFCC matrices in a workspace -> the net.                                  */
int dsp_classify_signal_batch_device(dsp_mfcc_plan *plan, dsp_stop_model *model, const float *d_signal, long n_clips,
                                     int samples_per_clip, long clip_stride, float *d_prob, void *stream);
/* ... and from the int16 PCM main_test.c:198-217 decodes in front of classify_signal (mono s / 32768, stereo average or channel 0). */
int dsp_classify_signal_batch_pcm16_device(dsp_mfcc_plan *plan, dsp_stop_model *model, const int16_t *d_pcm, long n_clips,
                                           int samples_per_clip, long clip_stride, int channels, int stereo_mode, float *d_prob,
                                           void *stream);
/* Ragged batches (offsets as dsp_scrubjay_fused_ragged_device): one launch of the fused clip -> probability kernel, every clip with its
 * own frame count (capped at the model's max_frames, stop_detector.c:26-30).  Plans of the reference's shape only (the fused kernel's).   */
int dsp_classify_signal_batch_ragged_device(dsp_mfcc_plan *plan, dsp_stop_model *model, const float *d_signal, long n_clips,
                                            const long *offsets, float *d_prob, void *stream);
int dsp_classify_signal_batch_ragged_pcm16_device(dsp_mfcc_plan *plan, dsp_stop_model *model, const int16_t *d_pcm, long n_clips,
                                                  const long *offsets, int channels, int stereo_mode, float *d_prob, void *stream);
/* classify_signal's own contract (stop_detector.h:10) with host buffers: probability in [0, 1];
 * a failure returns 0 with the reason in dsp_last_error().                                       */
float dsp_classify_signal(dsp_stop_model *model, const float *signal, int num_samples);

/* Speaker verification: max-component log-likelihood ratio of a target GMM against a UBM in the
 * reference's fixed point (2fa/audio/pico-audio/src/speaker_gmm.h:11-38, speaker_gmm.c:29-141;
 * parameters gmm_params.inc: means Q6 int8, inverse covariances Q11 int32, log constants Q8 int16). */
typedef struct dsp_gmm_params {
    int k, d;                    /* 32 mixtures, 13 dimensions (k <= 64, d <= 16) */
    const int8_t *means;         /* [k][d] */
    const int32_t *inv_covs;     /* [k][d] */
    const int16_t *log_consts;   /* [k]    */
} dsp_gmm_params;
typedef struct dsp_speaker_model dsp_speaker_model;
int dsp_speaker_model_create(const dsp_gmm_params *target, const dsp_gmm_params *ubm, int device, dsp_speaker_model **out);
void dsp_speaker_model_destroy(dsp_speaker_model *model);
/* mfcc_target_speaker_llr_mean (:127-136) and classify_speaker (:138-141) per clip of
 * d_mfcc[n_clips][frames_per_clip][d]: d_llr_mean[n_clips] (Q8, integer mean over the frames),
 * d_labels[n_clips] = llr_mean > (int64)(-0.7 * 256); optional per-frame log-likelihoods
 * d_ll_target / d_ll_ubm [n_clips][frames_per_clip] (target_gmm_log_likelihood / ubm_..., :84-102).
 * Bit-exact integer results.  HBM pointers; d_labels, d_ll_* may be NULL.                        */
int dsp_speaker_llr_device(dsp_speaker_model *model, const float *d_mfcc, long n_clips, int frames_per_clip,
                           int64_t *d_llr_mean, int *d_labels, int64_t *d_ll_target, int64_t *d_ll_ubm, void *stream);

/* upsampleLinear (sync/particle/main.cpp:62-77) over a batch: d_out[c][i] for i < new_size from
 * d_in[c][0..old_size), the reference's fp32 operation order (bit-identical).  new_size >= 2.     */
int dsp_upsample_linear_device(const float *d_in, long n_clips, int old_size, long in_stride, float *d_out,
                               int new_size, long out_stride, void *stream);
int dsp_upsample_linear_host(const float *in, int old_size, float *out, int new_size);

/* Reference-layout constant tables for a configuration (what mfcc_params.h holds
 * for the reference config): window[frame_length], mel[n_mels][n_fft/2+1],
 * dct[n_mfcc][n_mels].  Host-only, no GPU needed; any pointer may be NULL.      */
int dsp_mfcc_tables(const dsp_mfcc_config *cfg, float *window, float *mel, float *dct);

/* Per-lane kernel layout of the same tables (struct dsp::LaneTables512 of
 * dsp_amd/csrc/tables.hpp, `size` must equal its sizeof; returns that size when
 * out is NULL).  Host-only introspection used by the CPU tests of the planner. */
int dsp_mfcc_lane_tables(const dsp_mfcc_config *cfg, void *out, int size);

/* classify()'s spectrogram divides every PSD cell by U = fs * sum(window^2) (classifier.cpp:350-365).  The recompute kernel takes a
 * three-instruction form of that division for cells in [2^-60, 2^60] -- but only after the classifier context has compared it with
 * the real division on EVERY float of that range, on the device, for its U.  This call repeats the comparison: *mismatches = the
 * floats on which the two differ (0 expected); returns 1 when the fast form is in use, 0 when not (a mismatch, or
 * DSP_AMD_SPEC_EXACT_DIV set), < 0 on error.                                                                              */
int dsp_classify_division_check(long long *mismatches);

/* The host planner's self-check of BASELINE config 3's fused prefilter (dsp_mfcc_config.prefilter, tables.hpp PrefilterScan): the
 * literal band-pass as a cascade of four second-order sections run as lane scans.  Returns bit 0 = the cascade reproduces the
 * direct-form recurrence (donut-classifier/classifier.c:420-446) on 1024 samples, bit 1 = so does the row form of its scan (the
 * one the kernel runs); steps4 (may be NULL) receives the Kogge-Stone steps each section needs.  < 0: DSP_EINVAL.  Host only. */
int dsp_prefilter_scan_check(int prefilter, int *steps4);

/* --- misc -------------------------------------------------------------------- */
const char *dsp_last_error(void);   /* thread-local, "" when none */
int dsp_device_count(void);
const char *dsp_version(void);
/* ABI check for bindings that mirror the structs (ctypes, cgo, JNI ...): sizeof of the library's own dsp_mfcc_config,
 * dsp_classify_trace and dsp_classify_trace_f64 for which = 0, 1, 2 (-1 otherwise).  dsp_mfcc_config carries no size field of its own:
 * compare once after loading -- a binding built against an older header (fewer fields) must not pass its struct to this library.      */
int dsp_abi_sizeof(int which);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* DSP_AMD_H */
