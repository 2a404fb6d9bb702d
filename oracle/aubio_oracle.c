/*
 * aubio_oracle.c -- CPU restatement of the aubio front end that cepstrum/scrubjay_infer.c:21-53 drives:
 * aubio_source_do -> aubio_pvoc_do -> aubio_mfcc_do, one frame per hop.
 *
 * TEST INFRASTRUCTURE ONLY (see dsp_oracle.h).
 *
 * PARITY UNPINNED.  aubio is a third-party dependency of the reference (cepstrum/CMakeLists.txt:10,
 * `pkg_check_modules(AUBIO REQUIRED aubio)`, no version pinned); it is neither vendored under the reference tree nor
 * installed in this image, and the reference holds no golden vector at the aubio boundary (scrubjay_infer.c only prints
 * labels).  What follows restates the published algorithm of aubio 0.4 (0.4.9, the last 0.4 release; file and function
 * names below are aubio's), in aubio's default single-precision build (smpl_t = float).  It cannot be checked against a
 * run of the library here; it is checked against an independent float64 numpy restatement of the same published
 * algorithm (tests/test_oracle_aubio.py) and against the reference's own call sequence (frame count, history, pooling).
 *
 * The chain, with the call site in cepstrum/scrubjay_infer.c and the aubio routine restated:
 *   :21,41  aubio_source_do          hop_s new samples per call, the last partial block zero padded (src/io/source_*.c);
 *                                    the loop `do { ... } while (read == HOP_SIZE)` (:39-53) yields T = ceil(n / hop_s) frames
 *   :29,44  aubio_pvoc_do            src/spectral/phasevoc.c: slide (win_s - hop_s samples of history, zeros at the start),
 *                                    weight by new_aubio_window("hanningz") (src/mathutils.c), fvec_shift (swap halves: zero
 *                                    phase), aubio_fft_do -> cvec norm = |X[k]| (src/spectral/fft.c aubio_fft_get_norm)
 *   :30,45  aubio_mfcc_do            src/spectral/mfcc.c: aubio_filterbank_do on the MAGNITUDE spectrum (filterbank power 1)
 *                                    with aubio_filterbank_set_mel_coeffs_slaney (n_filters == 40, src/spectral/filterbank_mel.c),
 *                                    fvec_log10 (SAFE_LOG10: values below 2e-42 count as 2e-42), orthonormal DCT-II
 *                                    (src/spectral/dct_plain.c), first n_coefs outputs
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_oracle.h"

#define AUBIO_VERY_SMALL_NUMBER 2.e-42 /* aubio_priv.h */

/* src/mathutils.c fvec_set_window, case aubio_win_hanningz: w[i] = 0.5 (1 - cos(2 pi i / size)) in smpl_t */
void orc_aubio_window_hanningz(int n, float *w)
{
    for (int i = 0; i < n; ++i)
        w[i] = (float)(0.5 * (1.0 - cosf((float)(2.0 * M_PI) * (float)i / (float)n)));
}

/* src/musicutils/... aubio_bintofreq(bin, samplerate, fftsize) = samplerate / fftsize * max(bin, 0) */
static float aubio_bintofreq(float bin, float samplerate, float fftsize)
{
    const float freq = samplerate / fftsize;
    return freq * (bin > 0.f ? bin : 0.f);
}

/* src/spectral/filterbank_mel.c aubio_filterbank_set_triangle_bands: filters[n_filters][n_bins], n_bins = win_s / 2 + 1,
 * freqs[n_filters + 2] band edges in Hz; filterbank norm 1 (default): triangles of unit area. */
static void aubio_triangle_bands(const float *freqs, int n_filters, float samplerate, int n_bins, float *filters)
{
    float *fft_freqs = (float *)malloc(sizeof(float) * (size_t)n_bins);
    for (int bin = 0; bin < n_bins; ++bin) fft_freqs[bin] = aubio_bintofreq((float)bin, samplerate, (float)((n_bins - 1) * 2));
    memset(filters, 0, sizeof(float) * (size_t)n_filters * (size_t)n_bins);
    for (int fn = 0; fn < n_filters; ++fn) {
        const float lower = freqs[fn], center = freqs[fn + 1], upper = freqs[fn + 2];
        const float height = 2.f / (upper - lower);
        float *row = filters + (size_t)fn * n_bins;
        int bin;
        /* skip first elements */
        for (bin = 0; bin < n_bins - 1; ++bin) {
            if (fft_freqs[bin] <= lower && fft_freqs[bin + 1] > lower) { ++bin; break; }
        }
        /* positive slope */
        const float rise = height / (center - lower);
        for (; bin < n_bins - 1; ++bin) {
            row[bin] = (fft_freqs[bin] - lower) * rise;
            if (fft_freqs[bin + 1] >= center) { ++bin; break; }
        }
        /* negative slope */
        const float down = height / (upper - center);
        for (; bin < n_bins - 1; ++bin) {
            row[bin] += (upper - fft_freqs[bin]) * down;
            if (row[bin] < 0.f) row[bin] = 0.f;
            if (fft_freqs[bin + 1] >= upper) break;
        }
    }
    free(fft_freqs);
}

/* src/spectral/filterbank_mel.c aubio_filterbank_set_mel_coeffs_slaney: Malcolm Slaney's Auditory Toolbox bank, 13 linearly
 * spaced filters from 133.3333 Hz every 66.66666666 Hz, then 27 log-spaced ones (factor 1.0711703): 42 band edges. */
void orc_aubio_filterbank_slaney(int sample_rate, int win_s, float *filters)
{
    const float lowestFrequency = 133.3333f, linearSpacing = 66.66666666f, logSpacing = 1.0711703f;
    const int linearFilters = 13, logFilters = 27, n_filters = linearFilters + logFilters;
    float freqs[42];
    int fn;
    for (fn = 0; fn < linearFilters; ++fn) freqs[fn] = lowestFrequency + (float)fn * linearSpacing;
    const float lastlinearCF = freqs[fn - 1];
    for (fn = 0; fn < logFilters + 2; ++fn) freqs[fn + linearFilters] = lastlinearCF * powf(logSpacing, (float)(fn + 1));
    aubio_triangle_bands(freqs, n_filters, (float)sample_rate, win_s / 2 + 1, filters);
}

/* scrubjay_infer.c:39-53: a frame per aubio_source_do that returned samples; the loop ends after the first short read */
int orc_aubio_frames_for(int num_samples, int hop_s)
{
    return num_samples <= 0 ? 0 : (num_samples + hop_s - 1) / hop_s;
}

/* in-place iterative radix-2 transform in float64 (aubio's default FFT backend, ooura, computes in double and rounds the
 * result to smpl_t: src/spectral/fft.c, `fft_data_t` is double without fftw3f) */
static void fft_f64(double *re, double *im, int n)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i + k, b = a + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

/* The whole front end for one clip: out[T][n_coefs], returns T = orc_aubio_frames_for(n, hop_s).
 * n_filters must be 40 (new_aubio_mfcc picks the Slaney bank only then; scrubjay_infer.c:13 N_FILTERS 40). */
int orc_aubio_mfcc_clip(const float *signal, int num_samples, int sample_rate, int win_s, int hop_s, int n_filters,
                        int n_coefs, float *out)
{
    if (n_filters != 40 || win_s <= 0 || (win_s & (win_s - 1)) || hop_s <= 0 || hop_s > win_s || n_coefs > n_filters) return -1;
    const int T = orc_aubio_frames_for(num_samples, hop_s);
    const int n_bins = win_s / 2 + 1;
    float *w = (float *)malloc(sizeof(float) * (size_t)win_s);
    float *fb = (float *)malloc(sizeof(float) * (size_t)n_filters * (size_t)n_bins);
    float *dct = (float *)malloc(sizeof(float) * (size_t)n_filters * (size_t)n_filters);
    float *data = (float *)calloc((size_t)win_s, sizeof(float));
    float *dataold = (float *)calloc((size_t)win_s, sizeof(float));   /* win_s - hop_s samples of history, zeros (new_fvec) */
    float *in = (float *)malloc(sizeof(float) * (size_t)hop_s);
    float *norm = (float *)malloc(sizeof(float) * (size_t)n_bins);
    double *re = (double *)malloc(sizeof(double) * 2 * (size_t)win_s), *im = re + win_s;
    float in_dct[40], output[40];
    orc_aubio_window_hanningz(win_s, w);
    orc_aubio_filterbank_slaney(sample_rate, win_s, fb);
    /* src/spectral/dct_plain.c new_aubio_dct_plain: dct[j][i] = sqrt(2 / size) cos(j (i + .5) pi / size), row 0 = 1 / sqrt(size) */
    {
        const float scaling = sqrtf(2.f / (float)n_filters);
        for (int i = 0; i < n_filters; ++i) {
            for (int j = 1; j < n_filters; ++j)
                dct[(size_t)j * n_filters + i] = scaling * cosf((float)j * ((float)i + 0.5f) * (float)M_PI / (float)n_filters);
            dct[i] = 1.f / sqrtf((float)n_filters);
        }
    }
    const int end = win_s - hop_s;
    for (int t = 0; t < T; ++t) {
        /* aubio_source_do: hop_s samples, zero padded past the end of the file */
        for (int i = 0; i < hop_s; ++i) {
            const long g = (long)t * hop_s + i;
            in[i] = g < num_samples ? signal[g] : 0.f;
        }
        /* phasevoc.c aubio_pvoc_swapbuffers */
        memcpy(data, dataold, sizeof(float) * (size_t)end);
        memcpy(data + end, in, sizeof(float) * (size_t)hop_s);
        memcpy(dataold, data + hop_s, sizeof(float) * (size_t)end);
        /* fvec_weight, fvec_shift (swap halves), aubio_fft_do */
        for (int i = 0; i < win_s; ++i) {
            const float v = data[i] * w[i];
            const int j = i < win_s / 2 ? i + win_s / 2 : i - win_s / 2;
            re[j] = (double)v;
            im[j] = 0.0;
        }
        fft_f64(re, im, win_s);
        /* aubio_fft_get_norm on the half-complex spectrum rounded to smpl_t */
        for (int k = 0; k < n_bins; ++k) {
            const float a = (float)re[k], b = (float)im[k];
            norm[k] = (k == 0 || k == n_bins - 1) ? fabsf(a) : sqrtf(a * a + b * b);
        }
        /* aubio_filterbank_do: fmat_vecmul, ascending bins */
        for (int j = 0; j < n_filters; ++j) {
            float acc = 0.f;
            const float *row = fb + (size_t)j * n_bins;
            for (int k = 0; k < n_bins; ++k) acc += norm[k] * row[k];
            in_dct[j] = acc;
        }
        /* fvec_log10: SAFE_LOG10 */
        for (int j = 0; j < n_filters; ++j)
            in_dct[j] = log10f(fabsf(in_dct[j]) < (float)AUBIO_VERY_SMALL_NUMBER ? (float)AUBIO_VERY_SMALL_NUMBER : in_dct[j]);
        /* aubio_dct_do (fmat_vecmul), then the first n_coefs */
        for (int j = 0; j < n_filters; ++j) {
            float acc = 0.f;
            for (int i = 0; i < n_filters; ++i) acc += in_dct[i] * dct[(size_t)j * n_filters + i];
            output[j] = acc;
        }
        memcpy(out + (size_t)t * n_coefs, output, sizeof(float) * (size_t)n_coefs);
    }
    free(re); free(norm); free(in); free(dataold); free(data); free(dct); free(fb); free(w);
    return T;
}
