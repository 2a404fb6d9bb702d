/*
 * dsp_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see dsp_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  FMA
 * contraction is disabled so the fp32 "reference order" paths round exactly
 * like the reference built with plain `gcc -O2` on x86-64.
 *
 * Citations are reference-repo-relative paths (cornell-c2s2/dsp).
 */
#include "dsp_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.14159265358979323846

/* ------------------------------------------------------------------------ */
/* configuration                                                             */
/* ------------------------------------------------------------------------ */

void orc_mfcc_default_cfg(orc_mfcc_cfg *c)
{
    /* 2fa/audio/word/c/mfcc_params.h:6-12 */
    c->sample_rate = 16000;
    c->n_fft = 512;
    c->frame_length = 400;
    c->hop_length = 160;
    c->n_mels = 40;
    c->n_mfcc = 13;
    c->window = ORC_WINDOW_HANN;          /* export_mfcc_params.py:46 */
    c->mel_norm = ORC_MELNORM_NONE;       /* export_mfcc_params.py:56 */
    c->log_mode = ORC_LOG_PER_FRAME_MAX;  /* mfcc.c:169-206 */
    c->fft_mode = ORC_FFT_REFERENCE_ORDER;
    c->prefilter = ORC_PREFILTER_NONE;
    c->win_length = 0;
    c->fmin = 0.0f;
    c->fmax = 8000.0f;
    c->amin = 1e-10f;                     /* mfcc.c:172 */
    c->top_db = 80.0f;                    /* mfcc.c:173 */
}

/* ------------------------------------------------------------------------ */
/* tables                                                                    */
/* ------------------------------------------------------------------------ */

/* export_mfcc_params.py:46 -- get_window(name, n, fftbins=True): the periodic
 * form divides by n, not n-1; evaluated in float64 and rounded once.          */
void orc_window(int kind, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        double ph = 2.0 * ORC_PI * (double)i / (double)n;
        double w;
        switch (kind) {
        case ORC_WINDOW_HANN:    w = 0.5 - 0.5 * cos(ph);   break;
        case ORC_WINDOW_HAMMING: w = 0.54 - 0.46 * cos(ph); break;
        default:                 w = 1.0;                   break;
        }
        out[i] = (float)w;
    }
}

static double hz_to_mel_htk(double hz) { return 2595.0 * log10(1.0 + hz / 700.0); }
static double mel_to_hz_htk(double mel) { return 700.0 * (pow(10.0, mel / 2595.0) - 1.0); }
/* librosa.hz_to_mel / mel_to_hz with htk=False (Slaney's Auditory Toolbox scale): linear below 1 kHz (200/3 Hz per mel),
 * logarithmic above (log(6.4) / 27 per mel) -- what librosa.feature.mfcc uses by default (cepstrum/train.py:45-52) */
static double hz_to_mel_slaney(double hz)
{
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 1000.0 / (200.0 / 3.0), logstep = log(6.4) / 27.0;
    return hz >= min_log_hz ? min_log_mel + log(hz / min_log_hz) / logstep : hz / f_sp;
}
static double mel_to_hz_slaney(double mel)
{
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 1000.0 / (200.0 / 3.0), logstep = log(6.4) / 27.0;
    return mel >= min_log_mel ? min_log_hz * exp(logstep * (mel - min_log_mel)) : f_sp * mel;
}

/* export_mfcc_params.py:49-57 -- librosa.filters.mel(htk=True, norm=None|slaney):
 * n_mels+2 band edges equally spaced on the HTK mel axis, triangles built from
 * the two ramps min(lower, upper) clipped at 0, float64 then one rounding.     */
void orc_mel_filterbank(int sample_rate, int n_fft, int n_mels, float fmin,
                        float fmax, int mel_norm, float *out)
{
    const int n_bins = n_fft / 2 + 1;
    double *edge = (double *)malloc(sizeof(double) * (size_t)(n_mels + 2));
    const int slaney_scale = mel_norm == ORC_MELNORM_LIBROSA;
    if (slaney_scale) mel_norm = ORC_MELNORM_SLANEY;
    const double m_lo = slaney_scale ? hz_to_mel_slaney((double)fmin) : hz_to_mel_htk((double)fmin);
    const double m_hi = slaney_scale ? hz_to_mel_slaney((double)fmax) : hz_to_mel_htk((double)fmax);
    for (int i = 0; i < n_mels + 2; ++i) {
        /* np.linspace: start + i*step, last point pinned to stop */
        double m = (i == n_mels + 1) ? m_hi
                                     : m_lo + (double)i * ((m_hi - m_lo) / (double)(n_mels + 1));
        edge[i] = slaney_scale ? mel_to_hz_slaney(m) : mel_to_hz_htk(m);
    }
    for (int m = 0; m < n_mels; ++m) {
        const double d_lo = edge[m + 1] - edge[m];
        const double d_hi = edge[m + 2] - edge[m + 1];
        const double enorm = (mel_norm == ORC_MELNORM_SLANEY)
                                 ? 2.0 / (edge[m + 2] - edge[m]) : 1.0;
        for (int k = 0; k < n_bins; ++k) {
            /* fftfreqs = linspace(0, sr/2, n_bins) */
            double f = (k == n_bins - 1)
                           ? 0.5 * (double)sample_rate
                           : (double)k * (0.5 * (double)sample_rate / (double)(n_bins - 1));
            double lower = -(edge[m] - f) / d_lo;
            double upper = (edge[m + 2] - f) / d_hi;
            double w = lower < upper ? lower : upper;
            if (w <= 0.0) w = 0.0; /* np.maximum(0, .) also turns -0.0 into +0.0 */
            float wf = (float)w;
            if (mel_norm == ORC_MELNORM_SLANEY) wf = (float)((double)wf * enorm);
            out[(size_t)m * n_bins + k] = wf;
        }
    }
    free(edge);
}

/* export_mfcc_params.py:27-41 -- the exporter builds the basis in float32:
 * n is a float32 arange, pi*(n+0.5)*k/n_mels is evaluated left to right in
 * float32, cos in float32, scaled by a float64 sqrt rounded into float32.      */
void orc_dct_ortho(int n_mfcc, int n_mels, float *out)
{
    const float pi_f = (float)ORC_PI;
    for (int m = 0; m < n_mels; ++m)
        out[m] = (float)sqrt(1.0 / (double)n_mels);
    for (int k = 1; k < n_mfcc; ++k) {
        for (int m = 0; m < n_mels; ++m) {
            float arg = pi_f * ((float)m + 0.5f);
            arg = arg * (float)k;
            arg = arg / (float)n_mels;
            float c = cosf(arg);
            out[(size_t)k * n_mels + m] = (float)(sqrt(2.0 / (double)n_mels) * (double)c);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* FFT                                                                       */
/* ------------------------------------------------------------------------ */

static unsigned bit_reverse(unsigned v, int bits)
{
    unsigned r = 0;
    for (int b = 0; b < bits; ++b) {
        r = (r << 1) | (v & 1u);
        v >>= 1;
    }
    return r;
}

static int ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

/* mfcc.c:34-88 -- in-place radix-2 decimation-in-time on split re/im arrays.
 * Per span `len` the unit twiddle is cosf/sinf(-2*pi_f/len) and the running
 * twiddle restarts at 1 for every block and is advanced by one fp32 complex
 * product per butterfly (mfcc.c:55-57, 83-85): that recurrence, not a table,
 * is what fixes the reference's rounding, so it is kept as is.                 */
static void fft_radix2_reference_order(float *re, float *im, int n)
{
    const int bits = ilog2(n);
    for (int i = 0; i < n; ++i) {
        int j = (int)bit_reverse((unsigned)i, bits);
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    const float pi_f = 3.14159265358979323846f;
    for (int span = 2; span <= n; span <<= 1) {
        const int half = span >> 1;
        const float theta = -2.0f * pi_f / (float)span;
        const float step_c = cosf(theta);
        const float step_s = sinf(theta);
        for (int base = 0; base < n; base += span) {
            float wc = 1.0f, ws = 0.0f;
            for (int k = 0; k < half; ++k) {
                const int lo = base + k, hi = lo + half;
                const float pr = wc * re[hi] - ws * im[hi];
                const float pi = wc * im[hi] + ws * re[hi];
                const float ar = re[lo], ai = im[lo];
                re[hi] = ar - pr;
                im[hi] = ai - pi;
                re[lo] = ar + pr;
                im[lo] = ai + pi;
                const float nc = wc * step_c - ws * step_s;
                ws = wc * step_s + ws * step_c;
                wc = nc;
            }
        }
    }
}

static void dft_float64(const float *x, int frame_length, int n, float *out)
{
    /* exact-ish reference: O(n^2) in float64 over a table of the n-th roots of
     * unity (phase reduced mod n so it stays accurate); only used at test sizes */
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *sn = cs + n;
    for (int i = 0; i < n; ++i) {
        cs[i] = cos(-2.0 * ORC_PI * (double)i / (double)n);
        sn[i] = sin(-2.0 * ORC_PI * (double)i / (double)n);
    }
    for (int k = 0; k < n; ++k) {
        double sr = 0.0, si = 0.0;
        int ph = 0;
        for (int t = 0; t < frame_length; ++t) {
            sr += (double)x[t] * cs[ph];
            si += (double)x[t] * sn[ph];
            ph += k;
            if (ph >= n) ph -= n;
        }
        out[2 * k] = (float)sr;
        out[2 * k + 1] = (float)si;
    }
    free(cs);
}

void orc_fft_real_forward(const float *in_time, int frame_length, int n_fft,
                          int fft_mode, float *out_freq)
{
    if (fft_mode == ORC_FFT_FLOAT64) {
        dft_float64(in_time, frame_length, n_fft, out_freq);
        return;
    }
    float *re = (float *)malloc(sizeof(float) * 2 * (size_t)n_fft);
    float *im = re + n_fft;
    /* mfcc.c:25-32 zero-pad to n_fft, imaginary part zero */
    for (int i = 0; i < n_fft; ++i) {
        re[i] = i < frame_length ? in_time[i] : 0.0f;
        im[i] = 0.0f;
    }
    fft_radix2_reference_order(re, im, n_fft);
    for (int k = 0; k < n_fft; ++k) { /* mfcc.c:91-94 */
        out_freq[2 * k] = re[k];
        out_freq[2 * k + 1] = im[k];
    }
    free(re);
}

/* ------------------------------------------------------------------------ */
/* Butterworth                                                               */
/* ------------------------------------------------------------------------ */

/* donut-classifier/classifier.c:342-360, 383-401 -- the live 16 kHz literal
 * tables (scipy.signal.butter(4, [lo,hi], btype="band", fs=16000) printed to 8
 * decimals).  These are data, reproduced digit for digit because the filter's
 * response is defined by them, not by a redesign.                              */
static const double BUTTER_1000_3000_B[9] = {
    0.01020948, 0.0, -0.04083792, 0.0, 0.06125688, 0.0, -0.04083792, 0.0, 0.01020948};
static const double BUTTER_1000_3000_A[9] = {
    1.0, -4.56803686, 9.95922498, -13.49912589, 12.43979269,
    -7.94997696, 3.43760562, -0.92305481, 0.1203896};
static const double BUTTER_3000_7500_B[9] = {
    0.1362017, 0.0, -0.5448068, 0.0, 0.8172102, 0.0, -0.5448068, 0.0, 0.1362017};
static const double BUTTER_3000_7500_A[9] = {
    1.0, 2.60935592, 2.32553038, 1.20262614, 1.11690211,
    0.76154474, 0.10005124, -0.0129829, 0.02236815};

int orc_butter_bandpass(double lowcut, double highcut, double *b, double *a)
{
    const double *sb, *sa;
    if (lowcut == 1000 && highcut == 3000) {
        sb = BUTTER_1000_3000_B; sa = BUTTER_1000_3000_A;
    } else if (lowcut == 3000 && highcut == 7500) {
        sb = BUTTER_3000_7500_B; sa = BUTTER_3000_7500_A;
    } else {
        return 0; /* classifier.c:402-407: unknown band -> false */
    }
    memcpy(b, sb, sizeof(double) * 9);
    memcpy(a, sa, sizeof(double) * 9);
    return 1;
}

/* classifier.c:420-446 -- direct form II with an 8-deep delay line d[]:
 *   v = x - sum_{j=1..8} a[j] d[j-1];  y = b[0] v + sum_{j=1..8} b[j] d[j-1]
 * subtraction / accumulation order j = 1..8 as in the reference.               */
void orc_iir_df2_f64(const double *x, int n, const double *b, const double *a,
                     double *y)
{
    double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        for (int j = 1; j <= 8; ++j) v -= a[j] * d[j - 1];
        double acc = b[0] * v;
        for (int j = 1; j <= 8; ++j) acc += b[j] * d[j - 1];
        memmove(d + 1, d, sizeof(double) * 7);
        d[0] = v;
        y[i] = acc;
    }
}

/* sync/lib/classifier.cpp:193-219 */
void orc_iir_df2_f32(const float *x, int n, const float *b, const float *a,
                     float *y)
{
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        float v = x[i];
        for (int j = 1; j <= 8; ++j) v -= a[j] * d[j - 1];
        float acc = b[0] * v;
        for (int j = 1; j <= 8; ++j) acc += b[j] * d[j - 1];
        memmove(d + 1, d, sizeof(float) * 7);
        d[0] = v;
        y[i] = acc;
    }
}

/* ------------------------------------------------------------------------ */
/* MFCC chain                                                                */
/* ------------------------------------------------------------------------ */

typedef struct mfcc_plan {
    orc_mfcc_cfg cfg;
    int n_bins;
    float *window;  /* [frame_length] */
    float *mel;     /* [n_mels][n_bins] */
    float *dct;     /* [n_mfcc][n_mels] */
    double fb[9], fa[9];
} mfcc_plan;

static void plan_init(mfcc_plan *p, const orc_mfcc_cfg *cfg)
{
    p->cfg = *cfg;
    p->n_bins = cfg->n_fft / 2 + 1;
    p->window = (float *)malloc(sizeof(float) * (size_t)cfg->frame_length);
    p->mel = (float *)malloc(sizeof(float) * (size_t)cfg->n_mels * (size_t)p->n_bins);
    p->dct = (float *)malloc(sizeof(float) * (size_t)cfg->n_mfcc * (size_t)cfg->n_mels);
    if (cfg->win_length > 0 && cfg->win_length < cfg->frame_length) {
        /* librosa.util.pad_center(get_window(win_length), size=n_fft): zeros either side */
        const int lpad = (cfg->frame_length - cfg->win_length) / 2;
        memset(p->window, 0, sizeof(float) * (size_t)cfg->frame_length);
        orc_window(cfg->window, cfg->win_length, p->window + lpad);
    } else {
        orc_window(cfg->window, cfg->frame_length, p->window);
    }
    orc_mel_filterbank(cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin,
                       cfg->fmax, cfg->mel_norm, p->mel);
    orc_dct_ortho(cfg->n_mfcc, cfg->n_mels, p->dct);
    if (cfg->prefilter == ORC_PREFILTER_BUTTER_1000_3000)
        orc_butter_bandpass(1000, 3000, p->fb, p->fa);
    else if (cfg->prefilter == ORC_PREFILTER_BUTTER_3000_7500)
        orc_butter_bandpass(3000, 7500, p->fb, p->fa);
}

static void plan_free(mfcc_plan *p)
{
    free(p->window); free(p->mel); free(p->dct);
}

/* Log-mel of one frame, mfcc.c:169-206 (per-frame mode).  Returns nothing;
 * the clip-global mode only converts to dB here (ref = 1) and is clipped by
 * the caller once the clip maximum is known.                                   */
static void frame_log_mel(const mfcc_plan *p, const float *mel_e, float *log_mel)
{
    const int n_mels = p->cfg.n_mels;
    const float amin = p->cfg.amin;
    if (p->cfg.log_mode == ORC_LOG_GLOBAL_REF1) {
        for (int m = 0; m < n_mels; ++m) {
            float e = mel_e[m] < amin ? amin : mel_e[m];
            log_mel[m] = 10.0f * log10f(e); /* ref = 1.0 -> -10*log10(1) = 0 */
        }
        return;
    }
    float ref = 0.0f;                               /* mfcc.c:176-184 */
    for (int m = 0; m < n_mels; ++m)
        if (mel_e[m] > ref) ref = mel_e[m];
    if (ref < amin) ref = amin;
    const float log_ref = 10.0f * log10f(ref);
    for (int m = 0; m < n_mels; ++m) {              /* mfcc.c:187-194 */
        float e = mel_e[m] < amin ? amin : mel_e[m];
        log_mel[m] = 10.0f * log10f(e) - log_ref;
    }
    float peak = log_mel[0];                        /* mfcc.c:197-206 */
    for (int m = 1; m < n_mels; ++m)
        if (log_mel[m] > peak) peak = log_mel[m];
    const float floor_db = peak - p->cfg.top_db;
    for (int m = 0; m < n_mels; ++m)
        if (log_mel[m] < floor_db) log_mel[m] = floor_db;
}

/* One frame: window -> FFT -> power -> mel -> (log) ; mfcc.c:142-164.
 * `scratch` holds frame_length + 2*n_fft + n_bins + n_mels floats.             */
static void frame_to_log_mel(const mfcc_plan *p, const float *x, float *scratch,
                             float *log_mel)
{
    const orc_mfcc_cfg *c = &p->cfg;
    float *frame = scratch;
    float *spec = frame + c->frame_length;
    float *power = spec + 2 * c->n_fft;
    float *mel_e = power + p->n_bins;

    if (c->prefilter != ORC_PREFILTER_NONE) {
        /* BASELINE config 3: fp64 DF-II from zero state over this frame
         * (donut-classifier/classifier.c:420-446), rounded to fp32 before the
         * window like the fp32 MFCC chain expects.                             */
        double *xd = (double *)malloc(sizeof(double) * 2 * (size_t)c->frame_length);
        double *yd = xd + c->frame_length;
        for (int i = 0; i < c->frame_length; ++i) xd[i] = (double)x[i];
        orc_iir_df2_f64(xd, c->frame_length, p->fb, p->fa, yd);
        for (int i = 0; i < c->frame_length; ++i)
            frame[i] = (float)yd[i] * p->window[i];
        free(xd);
    } else {
        for (int i = 0; i < c->frame_length; ++i)   /* mfcc.c:142-144 */
            frame[i] = x[i] * p->window[i];
    }
    orc_fft_real_forward(frame, c->frame_length, c->n_fft, c->fft_mode, spec);
    for (int k = 0; k < p->n_bins; ++k) {           /* mfcc.c:151-155 */
        float re = spec[2 * k], im = spec[2 * k + 1];
        power[k] = re * re + im * im;
    }
    for (int m = 0; m < c->n_mels; ++m) {           /* mfcc.c:158-164, dense, ascending k */
        const float *w = p->mel + (size_t)m * p->n_bins;
        float acc = 0.0f;
        for (int k = 0; k < p->n_bins; ++k) acc += w[k] * power[k];
        mel_e[m] = acc;
    }
    frame_log_mel(p, mel_e, log_mel);
}

static void log_mel_to_mfcc(const mfcc_plan *p, const float *log_mel, float *out)
{
    const int n_mels = p->cfg.n_mels;
    for (int c = 0; c < p->cfg.n_mfcc; ++c) {       /* mfcc.c:210-216 */
        const float *d = p->dct + (size_t)c * n_mels;
        float acc = 0.0f;
        for (int m = 0; m < n_mels; ++m) acc += d[m] * log_mel[m];
        out[c] = acc;
    }
}

static size_t scratch_floats(const orc_mfcc_cfg *c)
{
    return (size_t)c->frame_length + 2 * (size_t)c->n_fft +
           (size_t)(c->n_fft / 2 + 1) + (size_t)c->n_mels;
}

int orc_compute_mfcc(const orc_mfcc_cfg *cfg, const float *signal,
                     int num_samples, float *out_mfcc, int max_frames)
{
    if (num_samples < cfg->frame_length || max_frames <= 0) return 0; /* mfcc.c:117-119 */
    int n_frames = 1 + (num_samples - cfg->frame_length) / cfg->hop_length;
    if (n_frames > max_frames) n_frames = max_frames;                 /* mfcc.c:137-139 */

    mfcc_plan plan;
    plan_init(&plan, cfg);
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(cfg));
    float *log_mel = (float *)malloc(sizeof(float) * (size_t)cfg->n_mels * (size_t)n_frames);

    for (int t = 0; t < n_frames; ++t)
        frame_to_log_mel(&plan, signal + (size_t)t * cfg->hop_length, scratch,
                         log_mel + (size_t)t * cfg->n_mels);

    if (cfg->log_mode == ORC_LOG_GLOBAL_REF1) {
        /* librosa power_to_db(top_db=80) over the whole clip
         * (2fa/audio/keyword_classifier.py:59) */
        float peak = -FLT_MAX;
        for (size_t i = 0; i < (size_t)n_frames * cfg->n_mels; ++i)
            if (log_mel[i] > peak) peak = log_mel[i];
        const float floor_db = peak - cfg->top_db;
        for (size_t i = 0; i < (size_t)n_frames * cfg->n_mels; ++i)
            if (log_mel[i] < floor_db) log_mel[i] = floor_db;
    }
    for (int t = 0; t < n_frames; ++t)              /* mfcc.c:219-221 frame-major */
        log_mel_to_mfcc(&plan, log_mel + (size_t)t * cfg->n_mels,
                        out_mfcc + (size_t)t * cfg->n_mfcc);

    free(log_mel); free(scratch);
    plan_free(&plan);
    return n_frames;
}

static void frames_range(const mfcc_plan *p, const float *frames, long lo, long hi,
                         float *out)
{
    const orc_mfcc_cfg *c = &p->cfg;
    float *scratch = (float *)malloc(sizeof(float) * scratch_floats(c));
    float *log_mel = (float *)malloc(sizeof(float) * (size_t)c->n_mels);
    for (long f = lo; f < hi; ++f) {
        frame_to_log_mel(p, frames + (size_t)f * c->frame_length, scratch, log_mel);
        if (c->log_mode == ORC_LOG_GLOBAL_REF1) {
            /* independent frames: each frame is its own "clip" */
            float peak = log_mel[0];
            for (int m = 1; m < c->n_mels; ++m) if (log_mel[m] > peak) peak = log_mel[m];
            for (int m = 0; m < c->n_mels; ++m)
                if (log_mel[m] < peak - c->top_db) log_mel[m] = peak - c->top_db;
        }
        log_mel_to_mfcc(p, log_mel, out + (size_t)f * c->n_mfcc);
    }
    free(log_mel); free(scratch);
}

void orc_mfcc_frames(const orc_mfcc_cfg *cfg, const float *frames, long n_frames,
                     float *out_mfcc)
{
    mfcc_plan plan;
    plan_init(&plan, cfg);
    frames_range(&plan, frames, 0, n_frames, out_mfcc);
    plan_free(&plan);
}

typedef struct mt_job { const mfcc_plan *p; const float *frames; long lo, hi; float *out; } mt_job;
static void *mt_entry(void *arg)
{
    mt_job *j = (mt_job *)arg;
    frames_range(j->p, j->frames, j->lo, j->hi, j->out);
    return NULL;
}

void orc_mfcc_frames_mt(const orc_mfcc_cfg *cfg, const float *frames, long n_frames,
                        float *out_mfcc, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    mfcc_plan plan;
    plan_init(&plan, cfg);
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    mt_job *job = (mt_job *)malloc(sizeof(mt_job) * (size_t)n_threads);
    const long per = (n_frames + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        long lo = per * t, hi = lo + per;
        if (lo > n_frames) lo = n_frames;
        if (hi > n_frames) hi = n_frames;
        job[t] = (mt_job){&plan, frames, lo, hi, out_mfcc};
        pthread_create(&tid[t], NULL, mt_entry, &job[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(tid[t], NULL);
    free(job); free(tid);
    plan_free(&plan);
}

/* ------------------------------------------------------------------------ */
/* spectrogram (scipy.signal.spectrogram defaults), fp32 firmware twin       */
/* ------------------------------------------------------------------------ */

#define SPEC_NPERSEG 256
#define SPEC_HOP 224 /* nperseg - nperseg/8, classifier.cpp:225-227 */
#define SPEC_BINS 129

int orc_spectrogram_bins(int n)
{
    return n < SPEC_NPERSEG ? 0 : (n - SPEC_NPERSEG) / SPEC_HOP + 1; /* classifier.cpp:236 */
}

/* classifier.cpp:259-293 -- periodic Tukey(alpha=0.25) written through a
 * (nperseg+1)-point symmetric formula; only the first nperseg points exist
 * (the reference's one-past-the-end store at index 256 is not restated).
 * All arithmetic float32, PI rounded from the double literal.                  */
static void tukey_f32(float *w)
{
    const float alpha = 0.25f;
    const float M = (float)(SPEC_NPERSEG + 1);
    const float pi_f = (float)ORC_PI;
    const int width = (int)floorf(alpha * (M - 1.0f) / 2.0f);
    for (int n = 0; n < SPEC_NPERSEG; ++n) {
        if (n <= width)
            w[n] = 0.5f * (1.0f + cosf(pi_f * (-1.0f + 2.0f * (float)n / (alpha * (M - 1.0f)))));
        else if (n <= (int)(M - (float)width - 2.0f))
            w[n] = 1.0f;
        else
            w[n] = 0.5f * (1.0f + cosf(pi_f * (-2.0f / alpha + 1.0f +
                                               2.0f * (float)n / (alpha * (M - 1.0f)))));
    }
}

/* sync/lib/PlainFFT.cpp:29-94 -- radix-2 DIT.  Twiddles: within a level the
 * running (u1,u2) is advanced once per butterfly column by the level's unit
 * rotation (c1,c2); between levels the unit rotation is halved with the
 * half-angle square roots evaluated in double and rounded to float.            */
static void fft_radix2_halfangle_f32(float *re, float *im, int n)
{
    const int bits = ilog2(n);
    for (int i = 0; i < n; ++i) {
        int j = (int)bit_reverse((unsigned)i, bits);
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    float c1 = -1.0f, c2 = 0.0f;
    int half = 1;
    for (int level = 0; level < bits; ++level) {
        const int span = half << 1;
        float u1 = 1.0f, u2 = 0.0f;
        for (int col = 0; col < half; ++col) {
            for (int lo = col; lo < n; lo += span) {
                const int hi = lo + half;
                const float t1 = u1 * re[hi] - u2 * im[hi];
                const float t2 = u1 * im[hi] + u2 * re[hi];
                re[hi] = re[lo] - t1;
                im[hi] = im[lo] - t2;
                re[lo] += t1;
                im[lo] += t2;
            }
            const float z = u1 * c1 - u2 * c2;
            u2 = u1 * c2 + u2 * c1;
            u1 = z;
        }
        c2 = (float)sqrt((1.0 - (double)c1) / 2.0);
        c2 = -c2; /* forward transform */
        c1 = (float)sqrt((1.0 + (double)c1) / 2.0);
        half = span;
    }
}

int orc_spectrogram_f32(const float *signal, int n, int fs, float *freqs,
                        float *times, float *sxx)
{
    const int T = orc_spectrogram_bins(n);
    float win[SPEC_NPERSEG], re[SPEC_NPERSEG], im[SPEC_NPERSEG];
    tukey_f32(win);
    float U = 0.0f;                                  /* classifier.cpp:296-301 */
    for (int i = 0; i < SPEC_NPERSEG; ++i) U += win[i] * win[i];
    U *= (float)fs;
    for (int k = 0; k < SPEC_BINS; ++k)              /* classifier.cpp:248-251 */
        freqs[k] = (float)k * (float)fs / (float)SPEC_NPERSEG;
    for (int t = 0; t < T; ++t) {
        const int start = t * SPEC_HOP;
        times[t] = ((float)(start + SPEC_NPERSEG / 2)) / (float)fs; /* :254-258 */
        float sum = 0.0f;                            /* :329-338 detrend */
        for (int i = 0; i < SPEC_NPERSEG; ++i) {
            re[i] = signal[start + i];
            im[i] = 0.0f;
            sum += re[i];
        }
        const float mean = sum / (float)SPEC_NPERSEG;
        for (int i = 0; i < SPEC_NPERSEG; ++i) re[i] = (re[i] - mean) * win[i];
        fft_radix2_halfangle_f32(re, im, SPEC_NPERSEG);
        for (int k = 0; k < SPEC_BINS; ++k) {        /* :350-365 */
            float p = (re[k] * re[k] + im[k] * im[k]) / U;
            if (k >= 1 && k < SPEC_BINS - 1) p *= 2.0f;
            sxx[(size_t)k * T + t] = p;
        }
    }
    return T;
}

/* donut-classifier/classifier.c:448-592 in float64; FFTW's r2c is replaced by
 * a float64 DFT (same mathematical transform; FFTW itself is unvendored).      */
int orc_spectrogram_f64(const double *signal, int n, int fs, double *freqs,
                        double *times, double *sxx)
{
    const int T = orc_spectrogram_bins(n);
    double win[SPEC_NPERSEG], seg[SPEC_NPERSEG];
    {   /* classifier.c:484-521: same Tukey construction in double */
        const double alpha = 0.25, M = SPEC_NPERSEG + 1;
        const int width = (int)floor(alpha * (M - 1.0) / 2.0);
        for (int i = 0; i < SPEC_NPERSEG; ++i) {
            if (i <= width)
                win[i] = 0.5 * (1.0 + cos(ORC_PI * (-1.0 + 2.0 * i / (alpha * (M - 1.0)))));
            else if (i <= (int)(M - width - 2))
                win[i] = 1.0;
            else
                win[i] = 0.5 * (1.0 + cos(ORC_PI * (-2.0 / alpha + 1.0 + 2.0 * i / (alpha * (M - 1.0)))));
        }
    }
    double U = 0.0;
    for (int i = 0; i < SPEC_NPERSEG; ++i) U += win[i] * win[i];
    U *= (double)fs;
    static double cs[SPEC_NPERSEG], sn[SPEC_NPERSEG];
    for (int i = 0; i < SPEC_NPERSEG; ++i) {
        cs[i] = cos(2.0 * ORC_PI * i / SPEC_NPERSEG);
        sn[i] = sin(2.0 * ORC_PI * i / SPEC_NPERSEG);
    }
    for (int k = 0; k < SPEC_BINS; ++k) freqs[k] = (double)k * fs / SPEC_NPERSEG;
    for (int t = 0; t < T; ++t) {
        const int start = t * SPEC_HOP;
        times[t] = (double)(start + SPEC_NPERSEG / 2) / fs;
        double sum = 0.0;
        for (int i = 0; i < SPEC_NPERSEG; ++i) sum += signal[start + i];
        const double mean = sum / SPEC_NPERSEG;
        for (int i = 0; i < SPEC_NPERSEG; ++i) seg[i] = (signal[start + i] - mean) * win[i];
        for (int k = 0; k < SPEC_BINS; ++k) {
            double sr = 0.0, si = 0.0;
            for (int i = 0; i < SPEC_NPERSEG; ++i) {
                int ph = (k * i) & (SPEC_NPERSEG - 1);
                sr += seg[i] * cs[ph];
                si -= seg[i] * sn[ph];
            }
            double p = (sr * sr + si * si) / U;
            if (k >= 1 && k < SPEC_BINS - 1) p *= 2.0;
            sxx[(size_t)k * T + t] = p;
        }
    }
    return T;
}

/* ------------------------------------------------------------------------ */
/* classify() tail                                                           */
/* ------------------------------------------------------------------------ */

/* classifier.cpp:35-65 -- dB relative to 1e-12 (log10 in double, stored float),
 * non-positive cells become NaN; returns clip min/max of the finite cells.
 * min/max start at +-inf: the reference initialises floats from +-DBL_MAX.     */
static void to_db_inplace(float *s, int count, float *lo, float *hi)
{
    float mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < count; ++i) {
        if (s[i] > 0) {
            s[i] = (float)(10 * log10((double)s[i] / 1e-12));
            if (s[i] < mn) mn = s[i];
            if (s[i] > mx) mx = s[i];
        } else {
            s[i] = NAN;
        }
    }
    *lo = mn; *hi = mx;
}

float orc_sum_intense(float lower, float upper, float half_range,
                      const float *freqs, int n_freq, const float *times,
                      int n_time, const float *db, float midpoint)
{
    /* classifier.cpp:373-414: first bin >= lower, last bin <= upper, clamped
     * and swapped if inverted; same for the time window.                       */
    int f0 = 0;
    while (f0 < n_freq && freqs[f0] < lower) ++f0;
    int f1 = n_freq - 1;
    while (f1 >= 0 && freqs[f1] > upper) --f1;
    if (f0 >= n_freq) f0 = n_freq - 1;
    if (f1 < 0) f1 = 0;
    if (f0 > f1) { int t = f0; f0 = f1; f1 = t; }

    int t0 = 0;
    while (t0 < n_time && times[t0] < midpoint - half_range) ++t0;
    int t1 = n_time - 1;
    while (t1 >= 0 && times[t1] > midpoint + half_range) --t1;
    if (t0 >= n_time) t0 = n_time - 1;
    if (t1 < 0) t1 = 0;
    if (t0 > t1) { int t = t0; t0 = t1; t1 = t; }

    float total = 0.0f;                              /* classifier.cpp:418-429 */
    for (int i = f0; i <= f1; ++i)
        for (int j = t0; j <= t1; ++j) {
            float v = db[(size_t)i * n_time + j];
            if (!isnan(v)) total += v;
        }
    return total;
}

int orc_find_midpoints(const float *data, int n, int fs, float *midpoints, int cap)
{
    return orc_find_midpoints_thr(data, n, fs, 70.0f, midpoints, cap);   /* classifier.cpp:436 */
}

/* lower_threshold_dB as a parameter: 70 in sync/lib/classifier.cpp:436, 45 in microphone/src/classifier.cpp:448 */
int orc_find_midpoints_thr(const float *data, int n, int fs, float threshold_db, float *midpoints, int cap)
{
    double bd[9], ad[9];
    float b[9], a[9];
    orc_butter_bandpass(1000, 3000, bd, ad);         /* classifier.cpp:438-442 */
    for (int i = 0; i < 9; ++i) { b[i] = (float)bd[i]; a[i] = (float)ad[i]; }

    const int T = orc_spectrogram_bins(n);
    if (T <= 0) return 0;
    float *filt = (float *)malloc(sizeof(float) * (size_t)n);
    float *sxx = (float *)malloc(sizeof(float) * (size_t)SPEC_BINS * (size_t)T);
    float *times = (float *)malloc(sizeof(float) * (size_t)T);
    float freqs[SPEC_BINS];
    orc_iir_df2_f32(data, n, b, a, filt);
    orc_spectrogram_f32(filt, n, fs, freqs, times, sxx);
    float lo, hi;
    to_db_inplace(sxx, SPEC_BINS * T, &lo, &hi);     /* classifier.cpp:457-476 */

    /* classifier.cpp:479-518: time bins with any cell above the threshold */
    float *blob = (float *)malloc(sizeof(float) * (size_t)T);
    int n_blob = 0;
    for (int j = 0; j < T; ++j) {
        int any = 0;
        for (int i = 0; i < SPEC_BINS && !any; ++i) {
            float v = sxx[(size_t)i * T + j];
            any = (v > threshold_db); /* NaN compares false */
        }
        if (any) blob[n_blob++] = times[j];
    }

    /* classifier.cpp:522-574: greedy clustering of consecutive blob times with
     * gap <= 0.05 s; clusters lasting >= 0.15 s emit the mean of their times.  */
    const float tol = 0.05f, min_dur = 0.15f;
    int count = 0;
    int i0 = 0;
    while (i0 < n_blob) {
        int i1 = i0;
        while (i1 + 1 < n_blob && (blob[i1 + 1] - blob[i1]) <= tol) ++i1;
        float dur = blob[i1] - blob[i0];
        if (dur >= min_dur) {
            float s = 0.0f;
            for (int k = i0; k <= i1; ++k) s += blob[k];
            if (count < cap) midpoints[count] = s / (float)(i1 - i0 + 1);
            ++count;
        }
        i0 = i1 + 1;
    }
    free(blob); free(times); free(sxx); free(filt);
    return count < cap ? count : cap;
}

int orc_classify(const float *data, int n, orc_classify_trace *trace)
{
    /* sync/lib/classifier.cpp:67-68, :436, :109 */
    const orc_classify_cfg cfg = {0.65f, 0.80f, 70.0f, 100.0f, 200.0f, 80.0f};
    return orc_classify_with(data, n, &cfg, trace);
}

/* The variants of classify() differ in these thresholds only (diff sync/lib/classifier.cpp
 * microphone/src/classifier.cpp): keep band, midpoint dB threshold, rule.          */
int orc_classify_with(const float *data, int n, const orc_classify_cfg *cfg, orc_classify_trace *trace)
{
    const int fs = 16000;                            /* classifier.cpp:12 */
    double bd[9], ad[9];
    float b[9], a[9];
    orc_butter_bandpass(3000, 7500, bd, ad);         /* classifier.cpp:14-19 */
    for (int i = 0; i < 9; ++i) { b[i] = (float)bd[i]; a[i] = (float)ad[i]; }
    const int T = orc_spectrogram_bins(n);
    if (trace) memset(trace, 0, sizeof(*trace));
    if (T <= 0) return 0;

    float *filt = (float *)malloc(sizeof(float) * (size_t)n);
    float *sxx = (float *)malloc(sizeof(float) * (size_t)SPEC_BINS * (size_t)T);
    float *times = (float *)malloc(sizeof(float) * (size_t)T);
    float freqs[SPEC_BINS];
    orc_iir_df2_f32(data, n, b, a, filt);            /* classifier.cpp:22-23 */
    orc_spectrogram_f32(filt, n, fs, freqs, times, sxx);
    float lo, hi;
    to_db_inplace(sxx, SPEC_BINS * T, &lo, &hi);     /* classifier.cpp:35-54 */
    const float lo_thr = cfg->keep_lo, hi_thr = cfg->keep_hi;      /* classifier.cpp:67-68 */
    for (int i = 0; i < SPEC_BINS * T; ++i) {
        if (!isnan(sxx[i])) {
            float v = (sxx[i] - lo) / (hi - lo);     /* classifier.cpp:57-65 */
            sxx[i] = (v > lo_thr && v < hi_thr) ? v : NAN; /* :71-80 */
        }
    }

    float mids[64];
    int n_mid = orc_find_midpoints_thr(data, n, fs, cfg->midpoint_db, mids, 64); /* classifier.cpp:84 */
    int hit = 0;
    if (trace) {
        trace->n_midpoints = n_mid;
        for (int k = 0; k < n_mid; ++k) trace->midpoints[k] = mids[k];   /* all of find_midpoints' results */
    }
    for (int k = 0; k < n_mid; ++k) {                /* classifier.cpp:93-114 */
        float above = orc_sum_intense(5000, 7000, 0.18f, freqs, SPEC_BINS, times, T, sxx, mids[k]);
        float middle = orc_sum_intense(2500, 5000, 0.05f, freqs, SPEC_BINS, times, T, sxx, mids[k]);
        float below = orc_sum_intense(500, 2500, 0.18f, freqs, SPEC_BINS, times, T, sxx, mids[k]);
        if (trace) {
            trace->sums[k][0] = above; trace->sums[k][1] = middle; trace->sums[k][2] = below;
        }
        if (middle < cfg->middle_max && above > cfg->above_min && below > cfg->below_min) { hit = 1; break; }
    }
    free(times); free(sxx); free(filt);
    return hit;
}

/* ------------------------------------------------------------------------ */
/* pooling + SVM                                                             */
/* ------------------------------------------------------------------------ */

void orc_mfcc_stats(const float *mfcc, int n_frames, int n_coef, float *out)
{
    /* cepstrum/scrubjay_infer.c:36-66 */
    for (int c = 0; c < n_coef; ++c) {
        double s = 0.0, q = 0.0;
        for (int t = 0; t < n_frames; ++t) {
            double v = (double)mfcc[(size_t)t * n_coef + c];
            s += v;
            q += v * v;
        }
        double mean = s / (double)n_frames;
        double var = q / (double)n_frames - mean * mean;
        out[c] = (float)mean;
        out[c + n_coef] = sqrtf((float)(var > 0 ? var : 0));
    }
}

int orc_svm_predict(const orc_svm_model *m, const float *x, float *decision,
                    float *prob1)
{
    /* ONNX Scaler: (x - offset) * scale; SVMClassifier RBF, two classes:
     * score = sum_i coef_i * exp(-gamma * |z - sv_i|^2) + rho.  This is libsvm's
     * decision value sum - model.rho with model.rho = -intercept (sklearn
     * libsvm_helper.c set_model), intercept = the ONNX rho; pinned against libsvm
     * by tools/pin_svm_libsvm.py -> tests/golden/svm_libsvm_ref.npz.
     * Label (svm.cpp svm_predict, sklearn .predict in cepstrum/run.py; ONNX
     * Runtime's SVC mode counts the same votes): score > 0 votes class 0, else 1.
     * Probability (svm.cpp svm_predict_probability): r01 = Platt sigmoid clamped to
     * [1e-7, 1-1e-7], then multiclass_probability -- an iteration from (1/2, 1/2)
     * with tolerance 0.005 / k, NOT the sigmoid itself.                          */
    float z[256];
    for (int j = 0; j < m->n_features; ++j)
        z[j] = (x[j] - m->offset[j]) * m->scale[j];
    float score = 0.0f;
    for (int i = 0; i < m->n_sv; ++i) {
        const float *sv = m->sv + (size_t)i * m->n_features;
        float d2 = 0.0f;
        for (int j = 0; j < m->n_features; ++j) {
            float d = z[j] - sv[j];
            d2 += d * d;
        }
        score += m->coef[i] * expf(-m->gamma * d2);
    }
    score += m->rho;
    float fApB = score * m->prob_a + m->prob_b;
    float r01f = (fApB >= 0.0f) ? expf(-fApB) / (1.0f + expf(-fApB))
                                : 1.0f / (1.0f + expf(fApB));
    if (r01f < 1e-7f) r01f = 1e-7f;
    if (r01f > 1.0f - 1e-7f) r01f = 1.0f - 1e-7f;
    /* multiclass_probability(k = 2), svm.cpp */
    double r[2][2] = {{0.0, (double)r01f}, {1.0 - (double)r01f, 0.0}};
    double Q[2][2], Qp[2], p[2] = {0.5, 0.5}, pQp;
    const double eps = 0.005 / 2;
    Q[0][0] = r[1][0] * r[1][0]; Q[0][1] = -r[1][0] * r[0][1];
    Q[1][1] = r[0][1] * r[0][1]; Q[1][0] = Q[0][1];
    for (int iter = 0; iter < 100; ++iter) {
        pQp = 0;
        for (int t = 0; t < 2; ++t) {
            Qp[t] = Q[t][0] * p[0] + Q[t][1] * p[1];
            pQp += p[t] * Qp[t];
        }
        double max_error = 0;
        for (int t = 0; t < 2; ++t) {
            double error = fabs(Qp[t] - pQp);
            if (error > max_error) max_error = error;
        }
        if (max_error < eps) break;
        for (int t = 0; t < 2; ++t) {
            double diff = (-Qp[t] + pQp) / Q[t][t];
            p[t] += diff;
            pQp = (pQp + diff * (diff * Q[t][t] + 2 * Qp[t])) / (1 + diff) / (1 + diff);
            for (int j = 0; j < 2; ++j) {
                Qp[j] = (Qp[j] + diff * Q[t][j]) / (1 + diff);
                p[j] /= (1 + diff);
            }
        }
    }
    if (decision) *decision = score;
    if (prob1) *prob1 = (float)p[1];
    return score > 0.0f ? 0 : 1;
}

/* ---- classify_signal tail (stop_detector.c, audio_classifier_inference.c) ---------- */

void orc_stop_features(const orc_stop_model *m, const float *mfcc, int n_frames, float *feats)
{
    /* stop_detector.c:26-50: T clamped to max_frames; feats[c * max_T + t] */
    int T = n_frames > m->max_frames ? m->max_frames : n_frames;
    for (int c = 0; c < m->n_coef; ++c)
        for (int t = 0; t < m->max_frames; ++t)
            feats[(size_t)c * m->max_frames + t] = t < T ? mfcc[(size_t)t * m->n_coef + c] : 0.0f;
}

static void orc_dense(const float *x, const float *w, const float *b, int n_in, int n_out, float *y, int relu)
{
    /* audio_classifier_inference.c:18-35: y[j] = b[j] + sum_i w[i * n_out + j] * x[i], i ascending */
    for (int j = 0; j < n_out; ++j) {
        float sum = b[j];
        for (int i = 0; i < n_in; ++i) sum += w[(size_t)i * n_out + j] * x[i];
        y[j] = (relu && !(sum > 0.0f)) ? 0.0f : sum;
    }
}

float orc_stop_predict(const orc_stop_model *m, const float *feats)
{
    const int n_in = m->n_coef * m->max_frames;
    float *x = (float *)malloc((size_t)n_in * sizeof(float));
    for (int i = 0; i < n_in; ++i) {            /* :41-47 */
        float scale = m->scaler_scale[i];
        if (scale == 0.0f) scale = 1.0f;
        x[i] = (feats[i] - m->scaler_mean[i]) / scale;
    }
    float h[2][64];
    orc_dense(x, m->kernel[0], m->bias[0], n_in, m->units[0], h[0], 1);
    orc_dense(h[0], m->kernel[1], m->bias[1], m->units[0], m->units[1], h[1], 1);
    orc_dense(h[1], m->kernel[2], m->bias[2], m->units[1], m->units[2], h[0], 1);
    orc_dense(h[0], m->kernel[3], m->bias[3], m->units[2], m->units[3], h[1], 0);
    free(x);
    return 1.0f / (1.0f + expf(-h[1][0]));      /* :13-15 */
}

float orc_classify_signal(const orc_stop_model *m, const float *signal, int num_samples)
{
    orc_mfcc_cfg cfg;
    orc_mfcc_default_cfg(&cfg);
    cfg.n_mfcc = m->n_coef;
    float *mf = (float *)calloc((size_t)m->max_frames * m->n_coef, sizeof(float));
    float *feats = (float *)malloc((size_t)m->max_frames * m->n_coef * sizeof(float));
    int T = orc_compute_mfcc(&cfg, signal, num_samples, mf, m->max_frames);
    orc_stop_features(m, mf, T, feats);
    float p = orc_stop_predict(m, feats);
    free(mf);
    free(feats);
    return p;
}

/* ---- speaker GMM (pico-audio/src/speaker_gmm.c) -------------------------------------- */

int64_t orc_gmm_log_likelihood(const orc_gmm *g, const int16_t *x)
{
    int64_t best = INT64_MIN;
    for (int k = 0; k < g->k; ++k) {
        int64_t sum_sq = 0;
        for (int d = 0; d < g->d; ++d) {
            int64_t diff = (int64_t)x[d] - (int64_t)g->means[k * g->d + d];
            sum_sq += diff * diff * (int64_t)g->inv_covs[k * g->d + d];   /* Q6 * Q6 * Q11 = Q23 */
        }
        sum_sq >>= 15;      /* Q23 -> Q8 (arithmetic shift, as gcc does for int64_t) */
        sum_sq /= 2;        /* toward zero */
        int64_t term = (int64_t)g->log_consts[k] - sum_sq;
        if (term > best) best = term;
    }
    return best;
}

void orc_float_to_q6(const float *in, int16_t *out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = (int16_t)(int32_t)(in[i] * 64);
}

int64_t orc_speaker_llr_mean(const orc_gmm *target, const orc_gmm *ubm, const float *mfcc, int n_frames)
{
    int16_t x[64];
    int64_t sum = 0;
    for (int t = 0; t < n_frames; ++t) {
        orc_float_to_q6(mfcc + (size_t)t * target->d, x, target->d);
        sum += orc_gmm_log_likelihood(target, x) - orc_gmm_log_likelihood(ubm, x);
    }
    return sum / n_frames;
}

int orc_classify_speaker(const orc_gmm *target, const orc_gmm *ubm, const float *mfcc, int n_frames)
{
    const int64_t threshold = (int64_t)(-0.7 * (1 << 8));
    return orc_speaker_llr_mean(target, ubm, mfcc, n_frames) > threshold ? 1 : 0;
}

/* ---- linear resampler (sync/particle/main.cpp:62-77) ------------------------------------ */

void orc_upsample_linear(const float *in, int old_size, float *out, int new_size)
{
    for (int i = 0; i < new_size; ++i) {
        float old_index = i * ((float)(old_size - 1) / (float)(new_size - 1));
        int lo = (int)floor(old_index);
        int hi = lo == old_size - 1 ? old_size - 1 : lo + 1;
        float frac = old_index - lo;
        out[i] = in[lo] + (in[hi] - in[lo]) * frac;
    }
}
