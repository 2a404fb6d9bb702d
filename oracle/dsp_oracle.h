/*
 * dsp_oracle.h -- CPU oracle for the MFCC / Butterworth / spectrogram hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it, and only as the checker.  The product (dsp_amd/) never links or
 * imports this library and fails loudly when its HIP library is missing.
 *
 * Every function restates the algorithm of one reference function; the
 * reference file:line it follows is cited on each declaration.  Paths are
 * relative to the upstream repository root (cornell-c2s2/dsp).
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against (1) goldens produced by the reference's own C sources compiled
 * unmodified into oracle/_ref/ (recipe: oracle/Makefile, generator:
 * tests/golden/make_golden.py) and (2) the reference's in-repo known-answer
 * dumps donut-classifier/_postbutter.txt and _blobtimes.txt (excerpts committed
 * as fixtures).  Exception: the aubio front end of cepstrum/scrubjay_infer.c is
 * a third-party library absent from the reference tree and from this image
 * (aubio, unpinned, cepstrum/CMakeLists.txt:10) -> orc_mfcc_stats()/orc_svm_*()
 * are "parity unpinned" at the aubio boundary; see DESIGN.md.  The consumers added for SURVEY 8f are
 * pinned the same way (stop-word net, speaker GMM: goldens from the reference's own stop_detector.c /
 * audio_classifier_inference.c / speaker_gmm.c in oracle/_ref); orc_upsample_linear() restates a firmware
 * file that cannot be built here and has no fixture: "parity unpinned".
 */
#ifndef DSP_ORACLE_H
#define DSP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- configuration ------------------------------------------------------ */

enum { ORC_WINDOW_HANN = 0, ORC_WINDOW_HAMMING = 1, ORC_WINDOW_RECT = 2 };
enum { ORC_MELNORM_NONE = 0, ORC_MELNORM_SLANEY = 1, ORC_MELNORM_LIBROSA = 2 };   /* 2: Slaney mel SCALE (htk=False) + Slaney norm: librosa.filters.mel's defaults */
/* log mode 0: per-frame ref=max, amin, top_db  (2fa/audio/word/c/mfcc.c:169-206)
 * log mode 1: librosa power_to_db(ref=1.0) with clip-global top_db
 *             (2fa/audio/keyword_classifier.py:37-68; golden test_mfcc.h)      */
enum { ORC_LOG_PER_FRAME_MAX = 0, ORC_LOG_GLOBAL_REF1 = 1 };
/* fft mode 0: radix-2 DIT, fp32, twiddle by running product -- the reference's
 *             operation order (mfcc.c:34-88), bit-faithful when built without
 *             FMA contraction.
 * fft mode 1: float64 direct evaluation (exact up to 1e-15), "truth" for
 *             measuring the reference's own fp32 noise floor.                  */
enum { ORC_FFT_REFERENCE_ORDER = 0, ORC_FFT_FLOAT64 = 1 };
enum { ORC_PREFILTER_NONE = 0, ORC_PREFILTER_BUTTER_1000_3000 = 1,
       ORC_PREFILTER_BUTTER_3000_7500 = 2 };

typedef struct orc_mfcc_cfg {
    int sample_rate;   /* 16000  (mfcc_params.h:6)  */
    int n_fft;         /* 512    (mfcc_params.h:7)  */
    int frame_length;  /* 400    (mfcc_params.h:8)  */
    int hop_length;    /* 160    (mfcc_params.h:9)  */
    int n_mels;        /* 40     (mfcc_params.h:10) */
    int n_mfcc;        /* 13     (mfcc_params.h:11) */
    int window;        /* ORC_WINDOW_*   */
    int mel_norm;      /* ORC_MELNORM_*  */
    int log_mode;      /* ORC_LOG_*      */
    int fft_mode;      /* ORC_FFT_*      */
    int prefilter;     /* ORC_PREFILTER_* : fp64 IIR applied per frame from zero
                          state before the window (BASELINE config 3)          */
    int win_length;    /* 0 = frame_length; else the window has win_length taps
                          centred in the frame (librosa win_length < n_fft,
                          2fa/audio/word/python/keyword_classifier.py:42-55)    */
    float fmin, fmax;  /* 0, sample_rate/2 */
    float amin;        /* 1e-10 (mfcc.c:172) */
    float top_db;      /* 80    (mfcc.c:173) */
} orc_mfcc_cfg;

/* Reference defaults (mfcc_params.h:6-12, export_mfcc_params.py:44-57). */
void orc_mfcc_default_cfg(orc_mfcc_cfg *cfg);

/* ---- constant tables (2fa/audio/word/python/export_mfcc_params.py) ------ */

/* :46  periodic window of `n` samples (scipy get_window(..., fftbins=True)). */
void orc_window(int kind, int n, float *out);
/* :49-57  librosa.filters.mel(htk=True): out[n_mels][n_fft/2+1], row-major.  */
void orc_mel_filterbank(int sample_rate, int n_fft, int n_mels, float fmin,
                        float fmax, int mel_norm, float *out);
/* :27-41  orthonormal DCT-II basis out[n_mfcc][n_mels], evaluated in float32
 * in the exporter's operation order.                                          */
void orc_dct_ortho(int n_mfcc, int n_mels, float *out);

/* ---- MFCC chain (2fa/audio/word/c/mfcc.c) -------------------------------- */

/* mfcc.c:16-95  zero-padded real input -> n_fft complex bins interleaved.     */
void orc_fft_real_forward(const float *in_time, int frame_length, int n_fft,
                          int fft_mode, float *out_freq);

/* mfcc.c:108-232  frames a clip with hop_length and writes frame-major
 * out[T][n_mfcc]; returns T (0 when the clip is shorter than one frame or
 * max_frames <= 0, mfcc.c:117-119).                                            */
int orc_compute_mfcc(const orc_mfcc_cfg *cfg, const float *signal,
                     int num_samples, float *out_mfcc, int max_frames);

/* Same per-frame chain applied to `n_frames` independent frames laid out
 * back to back, frames[n_frames][frame_length] (BASELINE configs 2 and 3).    */
void orc_mfcc_frames(const orc_mfcc_cfg *cfg, const float *frames,
                     long n_frames, float *out_mfcc);

/* As orc_mfcc_frames but split over `n_threads` pthreads (bench cpu_baseline
 * "all cores" leg).                                                            */
void orc_mfcc_frames_mt(const orc_mfcc_cfg *cfg, const float *frames,
                        long n_frames, float *out_mfcc, int n_threads);

/* ---- Butterworth band-pass (donut-classifier/classifier.c) -------------- */

/* classifier.c:319-409 (fp32 twin sync/lib/classifier.cpp:138-191): literal
 * 9-tap tables for (1000,3000) and (3000,7500) Hz @ 16 kHz; returns 0 and
 * leaves b/a untouched for any other band.                                     */
int orc_butter_bandpass(double lowcut, double highcut, double *b, double *a);
/* classifier.c:420-446  direct-form-II, zero initial state, float64.          */
void orc_iir_df2_f64(const double *x, int n, const double *b, const double *a,
                     double *y);
/* sync/lib/classifier.cpp:193-219  the same recurrence in float32.            */
void orc_iir_df2_f32(const float *x, int n, const float *b, const float *a,
                     float *y);

/* ---- spectrogram + classify (sync/lib/classifier.cpp, fp32 firmware twin) */

/* classifier.cpp:221-368 (+ PlainFFT.cpp:29-94): nperseg 256, hop 224, mean
 * removal, periodic Tukey(0.25), radix-2 FFT with sqrt-recurrence twiddles,
 * PSD scaling.  sxx is [129][T] row-major, T = (n-256)/224+1 (returned).      */
int orc_spectrogram_f32(const float *signal, int n, int fs, float *freqs,
                        float *times, float *sxx);
/* donut-classifier/classifier.c:448-592  float64 twin (exact DFT instead of
 * FFTW; FFTW is unvendored).  Same layout.                                     */
int orc_spectrogram_f64(const double *signal, int n, int fs, double *freqs,
                        double *times, double *sxx);
/* Number of spectrogram columns for an n-sample clip (classifier.cpp:236).    */
int orc_spectrogram_bins(int n);

/* classifier.cpp:370-431 */
float orc_sum_intense(float lower, float upper, float half_range,
                      const float *freqs, int n_freq, const float *times,
                      int n_time, const float *db /* [n_freq][n_time] */,
                      float midpoint);
/* classifier.cpp:433-598  returns the number of midpoints written (<= cap).   */
int orc_find_midpoints(const float *data, int n, int fs, float *midpoints,
                       int cap);

typedef struct orc_classify_trace {
    int n_midpoints;
    float midpoints[64];
    float sums[64][3]; /* above, middle, below per midpoint (classifier.cpp:99-101) */
} orc_classify_trace;

/* classifier.cpp:9-136  returns 0/1; `trace` may be NULL.                     */
int orc_classify(const float *data, int n, orc_classify_trace *trace);
typedef struct orc_classify_cfg {
    float keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min;
} orc_classify_cfg;
int orc_classify_with(const float *data, int n, const orc_classify_cfg *cfg, orc_classify_trace *trace);
int orc_find_midpoints_thr(const float *data, int n, int fs, float threshold_db, float *midpoints, int cap);

/* ---- the float64 classifier, donut-classifier/classifier.c (classify_f64_oracle.c) ---- */
typedef struct orc_classify_cfg_f64 {
    double keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min;
} orc_classify_cfg_f64;
typedef struct orc_classify_trace_f64 {
    int n_midpoints;
    double midpoints[64];
    double sums[64][3];
} orc_classify_trace_f64;
/* classifier.c:594-653 */
double orc_sum_intense_f64(double lower, double upper, double half_range, const double *freqs, int n_freq, const double *times,
                           int n_time, const double *db, double midpoint);
/* classifier.c:655-830 (threshold_db: 45 at :660) */
int orc_find_midpoints_f64(const double *data, int n, int fs, double threshold_db, double *midpoints, int cap);
/* classifier.c:83-192 per clip; cfg NULL = the file's thresholds (0.70 / 0.85 :141-142, 45 dB :660, 75 / 300 / 100 :184) */
int orc_classify_f64(const double *data, int n, const orc_classify_cfg_f64 *cfg, orc_classify_trace_f64 *trace);

/* ---- pooling + SVM (cepstrum/scrubjay_infer.c, scrubjay_svm.onnx) ------- */

/* scrubjay_infer.c:36-66  mean | population-std over T frames of n_coef,
 * float64 accumulators, out[2*n_coef].                                         */
void orc_mfcc_stats(const float *mfcc, int n_frames, int n_coef, float *out);

typedef struct orc_svm_model {
    int n_features;      /* 40 */
    int n_sv;            /* 55 */
    float gamma;         /* 0.025 */
    float rho;           /* ONNX rho[0] */
    float prob_a, prob_b;
    const float *offset; /* Scaler offset[n_features] */
    const float *scale;  /* Scaler scale[n_features]  */
    const float *sv;     /* [n_sv][n_features] */
    const float *coef;   /* [n_sv] dual coefficients */
} orc_svm_model;

/* ONNX Scaler -> SVMClassifier(RBF) -> Platt, as decoded from
 * cepstrum/scrubjay_svm.onnx (call site scrubjay_infer.c:105-141).
 * Returns the label (0/1); decision and P(label 1) are written if non-NULL.  */
int orc_svm_predict(const orc_svm_model *m, const float *x, float *decision,
                    float *prob1);

/* ---- consumers of the MFCC matrix and the resampler (8f-4) ---- */

/* 2fa/audio/word/c/model_params.h:6-10 (sizes) + :12-4925 (trained parameters; passed in,
 * never compiled in here).                                                          */
typedef struct orc_stop_model {
    int n_coef;        /* 13  MFCC_N_MFCC */
    int max_frames;    /* 500 MAX_FRAMES (stop_detector.c:9) */
    int units[4];      /* 4, 2, 2, 1 */
    const float *scaler_mean, *scaler_scale; /* [n_coef * max_frames], coefficient-major */
    const float *kernel[4];                  /* (in, out) row-major */
    const float *bias[4];
} orc_stop_model;

/* stop_detector.c:36-50  frame-major [n_frames][n_coef] -> coefficient-major
 * feats[n_coef][max_frames], zero-padded / truncated at max_frames.                 */
void orc_stop_features(const orc_stop_model *m, const float *mfcc, int n_frames, float *feats);
/* audio_classifier_inference.c:38-90  StandardScaler -> 3 x (dense + ReLU) -> dense ->
 * sigmoid, sums in the reference's order (fp32, input index ascending).             */
float orc_stop_predict(const orc_stop_model *m, const float *feats);
/* stop_detector.c:12-55  compute_mfcc (reference defaults) -> features -> net.      */
float orc_classify_signal(const orc_stop_model *m, const float *signal, int num_samples);

/* 2fa/audio/pico-audio/src/speaker_gmm.c (parameters: gmm_params.inc:8-13 Q formats). */
typedef struct orc_gmm {
    int k, d;                 /* 32 mixtures, 13 dimensions */
    const int8_t *means;      /* [k][d]  Q6  */
    const int32_t *inv_covs;  /* [k][d]  Q11 */
    const int16_t *log_consts;/* [k]     Q8  */
} orc_gmm;
/* speaker_gmm.c:29-50  max-component approximation of the log-likelihood, Q8.       */
int64_t orc_gmm_log_likelihood(const orc_gmm *g, const int16_t *x);
/* speaker_gmm.c:118-122  x * 64 truncated to int16 (out of range: low 16 bits of the
 * int32 truncation, what gcc on x86-64 does for the reference).                      */
void orc_float_to_q6(const float *in, int16_t *out, int n);
/* speaker_gmm.c:104-108,127-136  sum over frames of (target - ubm), integer mean.   */
int64_t orc_speaker_llr_mean(const orc_gmm *target, const orc_gmm *ubm, const float *mfcc, int n_frames);
/* speaker_gmm.c:124-125,138-141  mean LLR > (int64)(-0.7 * 256).                     */
int orc_classify_speaker(const orc_gmm *target, const orc_gmm *ubm, const float *mfcc, int n_frames);

/* sync/particle/main.cpp:62-77  linear-interpolation resampler (fp32, no contraction). */
void orc_upsample_linear(const float *in, int old_size, float *out, int new_size);

/* ---- aubio front end of cepstrum/scrubjay_infer.c:21-53 (aubio_oracle.c) -- PARITY UNPINNED ------------------
 * aubio is an unvendored, unpinned third-party dependency (cepstrum/CMakeLists.txt:10) absent from this image;
 * these restate the published algorithm of aubio 0.4.9 (routine names in aubio_oracle.c).                      */
/* new_aubio_window("hanningz", n) */
void orc_aubio_window_hanningz(int n, float *w);
/* aubio_filterbank_set_mel_coeffs_slaney on a (40, win_s / 2 + 1) bank, unit-area triangles: filters[40][win_s/2+1] */
void orc_aubio_filterbank_slaney(int sample_rate, int win_s, float *filters);
/* scrubjay_infer.c:39-53  frames the do/while yields for an n-sample file: ceil(n / hop_s) */
int orc_aubio_frames_for(int num_samples, int hop_s);
/* scrubjay_infer.c:41-45  aubio_source_do -> aubio_pvoc_do -> aubio_mfcc_do per hop: out[T][n_coefs], returns T
 * (-1: unsupported arguments; n_filters must be 40 as in scrubjay_infer.c:13)                                   */
int orc_aubio_mfcc_clip(const float *signal, int num_samples, int sample_rate, int win_s, int hop_s, int n_filters,
                        int n_coefs, float *out);

#ifdef __cplusplus
}
#endif
#endif /* DSP_ORACLE_H */
