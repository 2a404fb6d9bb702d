/*
 * classify_f64_oracle.c -- CPU restatement of the float64 scrub-jay classifier, donut-classifier/classifier.c (the file
 * north_star names for the Butterworth pre-filter): main's per-file body :83-192, sum_intense :594-653, find_midpoints
 * :655-830, on top of the float64 butter_bandpass_filter / compute_spectrogram restated in dsp_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (see dsp_oracle.h).
 *
 * Pinning: donut-classifier/classifier.c itself cannot be built here (it includes sndfile.h and fftw3.h, neither in the image),
 * so the CHAIN has no output of the compiled file to compare with.  Its stages are pinned separately by the reference's own
 * known-answer dumps: _postbutter.txt (the filter, tests/golden/iir_kat.npz) and _blobtimes.txt (spectrogram + 45 dB mask =
 * exactly the blob times find_midpoints clusters, tests/golden/blobtimes_kat.npz), and the spectrogram against
 * scipy.signal.spectrogram (the Python prototype classifier16k.py:41-50 calls it).  The tail restated here (dB map, clip-global
 * normalisation, keep band, clustering, band sums, rule) is plain sequential arithmetic that follows the file line by line.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_oracle.h"

#define F64_BINS 129

/* classifier.c:594-653: first bin >= lower, last bin <= upper (clamped, swapped if inverted), the same for the time window,
 * then rows outer / columns inner, NaN cells skipped */
double orc_sum_intense_f64(double lower, double upper, double half_range, const double *freqs, int n_freq, const double *times,
                           int n_time, const double *db /* [n_freq][n_time] */, double midpoint)
{
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
    double total = 0.0;
    for (int i = f0; i <= f1; ++i)
        for (int j = t0; j <= t1; ++j) {
            const double v = db[(size_t)i * n_time + j];
            if (!isnan(v)) total += v;
        }
    return total;
}

/* classifier.c:655-830 */
int orc_find_midpoints_f64(const double *data, int n, int fs, double threshold_db, double *midpoints, int cap)
{
    double b[9], a[9];
    orc_butter_bandpass(1000.0, 3000.0, b, a);                       /* :659-664 */
    const int T = orc_spectrogram_bins(n);
    if (T <= 0) return 0;
    double *filt = (double *)malloc(sizeof(double) * (size_t)n);
    double *sxx = (double *)malloc(sizeof(double) * (size_t)F64_BINS * (size_t)T);
    double *times = (double *)malloc(sizeof(double) * (size_t)T);
    double freqs[F64_BINS];
    orc_iir_df2_f64(data, n, b, a, filt);                            /* :667-668 */
    orc_spectrogram_f64(filt, n, fs, freqs, times, sxx);             /* :676 */
    /* :679-713  dB where the PSD is positive (NaN otherwise), then keep cells above the threshold */
    double *blob = (double *)malloc(sizeof(double) * (size_t)T);
    int n_blob = 0;
    for (int j = 0; j < T; ++j) {                                    /* :715-745 */
        int any = 0;
        for (int i = 0; i < F64_BINS && !any; ++i) {
            const double s = sxx[(size_t)i * T + j];
            if (s > 0) any = 10 * log10(s / 1e-12) > threshold_db;
        }
        if (any) blob[n_blob++] = times[j];
    }
    /* :747-800  consecutive blob times with a gap <= 0.05 s form a cluster; clusters of >= 0.15 s give the mean of their times */
    const double tol = 0.05, min_dur = 0.15;
    int count = 0, i0 = 0;
    while (i0 < n_blob) {
        int i1 = i0;
        while (i1 + 1 < n_blob && (blob[i1 + 1] - blob[i1]) <= tol) ++i1;
        if (blob[i1] - blob[i0] >= min_dur) {
            double s = 0.0;
            for (int k = i0; k <= i1; ++k) s += blob[k];
            if (count < cap) midpoints[count] = s / (double)(i1 - i0 + 1);
            ++count;
        }
        i0 = i1 + 1;
    }
    free(blob); free(times); free(sxx); free(filt);
    return count < cap ? count : cap;
}

/* classifier.c:83-192 for one clip already converted to double (:55-59: wav / 32768.0).  cfg NULL: the file's own thresholds
 * (keep band 0.70 / 0.85 :141-142, 45 dB :660, rule middle < 75 && above > 300 && below > 100 :184). */
int orc_classify_f64(const double *data, int n, const orc_classify_cfg_f64 *cfg, orc_classify_trace_f64 *trace)
{
    static const orc_classify_cfg_f64 donut = {0.70, 0.85, 45.0, 75.0, 300.0, 100.0};
    if (!cfg) cfg = &donut;
    const int fs = 16000;
    double b[9], a[9];
    orc_butter_bandpass(3000.0, 7500.0, b, a);                       /* :86-91 */
    const int T = orc_spectrogram_bins(n);
    if (trace) memset(trace, 0, sizeof(*trace));
    if (T <= 0) return 0;
    double *filt = (double *)malloc(sizeof(double) * (size_t)n);
    double *sxx = (double *)malloc(sizeof(double) * (size_t)F64_BINS * (size_t)T);
    double *times = (double *)malloc(sizeof(double) * (size_t)T);
    double freqs[F64_BINS];
    orc_iir_df2_f64(data, n, b, a, filt);                            /* :94-95 */
    orc_spectrogram_f64(filt, n, fs, freqs, times, sxx);             /* :102 */
    double mn = DBL_MAX, mx = -DBL_MAX;                              /* :105-125 */
    for (int i = 0; i < F64_BINS * T; ++i) {
        if (sxx[i] > 0) {
            sxx[i] = 10 * log10(sxx[i] / 1e-12);
            if (sxx[i] < mn) mn = sxx[i];
            if (sxx[i] > mx) mx = sxx[i];
        } else {
            sxx[i] = NAN;
        }
    }
    const double lo_thr = cfg->keep_lo, hi_thr = cfg->keep_hi;
    for (int i = 0; i < F64_BINS * T; ++i) {                         /* :130-157 */
        if (!isnan(sxx[i])) {
            const double v = (sxx[i] - mn) / (mx - mn);
            sxx[i] = (v > lo_thr && v < hi_thr) ? v : NAN;
        }
    }
    double mids[64];
    const int n_mid = orc_find_midpoints_f64(data, n, fs, cfg->midpoint_db, mids, 64);   /* :161 */
    int hit = 0;
    if (trace) {
        trace->n_midpoints = n_mid;
        for (int k = 0; k < n_mid; ++k) trace->midpoints[k] = mids[k];
    }
    for (int k = 0; k < n_mid; ++k) {                                /* :170-190 */
        const double above = orc_sum_intense_f64(5000, 7000, 0.18, freqs, F64_BINS, times, T, sxx, mids[k]);
        const double middle = orc_sum_intense_f64(2500, 5000, 0.05, freqs, F64_BINS, times, T, sxx, mids[k]);
        const double below = orc_sum_intense_f64(500, 2500, 0.18, freqs, F64_BINS, times, T, sxx, mids[k]);
        if (trace) { trace->sums[k][0] = above; trace->sums[k][1] = middle; trace->sums[k][2] = below; }
        if (middle < cfg->middle_max && above > cfg->above_min && below > cfg->below_min) { hit = 1; break; }
    }
    free(times); free(sxx); free(filt);
    return hit;
}
