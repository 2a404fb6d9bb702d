/*
 * ref_classifier_shim.cpp -- C-linkage re-exports compiled TOGETHER WITH the
 * reference's own sync/lib/classifier.cpp + PlainFFT.cpp (from /root/reference,
 * never copied) into oracle/_ref/libref_classifier.so, so ctypes can call the
 * C++-mangled reference functions.  TEST INFRASTRUCTURE ONLY.
 */
#include <cstdlib>
#include "classifier.h" /* resolved via -I$(REF)/sync/lib */

extern "C" {
int ref_classify(float *data, int n) { return classify(data, n); }
int ref_butter_bandpass(float lo, float hi, float *b, float *a)
{
    return butter_bandpass(lo, hi, b, a) ? 1 : 0;
}
void ref_butter_bandpass_filter(float *x, int n, float *b, float *a, float *y)
{
    butter_bandpass_filter(x, n, b, a, y);
}
/* Flattens the reference's malloc'd float*[129] rows into sxx[129][T]. */
int ref_compute_spectrogram(float *signal, int n, int fs, float *freqs,
                            float *times, float *sxx)
{
    float *f = nullptr, *t = nullptr, **s = nullptr;
    int nf = 0, nt = 0;
    compute_spectrogram(signal, n, fs, &f, &t, &s, &nf, &nt);
    for (int i = 0; i < nf; ++i) {
        freqs[i] = f[i];
        for (int j = 0; j < nt; ++j) sxx[(size_t)i * nt + j] = s[i][j];
        free(s[i]);
    }
    for (int j = 0; j < nt; ++j) times[j] = t[j];
    free(s); free(f); free(t);
    return nt;
}
int ref_find_midpoints(float *data, int n, int fs, float *out, int cap)
{
    int count = 0;
    float *m = find_midpoints(data, n, fs, &count);
    for (int i = 0; i < count && i < cap; ++i) out[i] = m[i];
    free(m);
    return count;
}
float ref_sum_intense(float lower, float upper, float half_range, float *freqs,
                      int nf, float *times, int nt, float *db, float midpoint)
{
    float **rows = (float **)malloc(sizeof(float *) * nf);
    for (int i = 0; i < nf; ++i) rows[i] = db + (size_t)i * nt;
    float r = sum_intense(lower, upper, half_range, freqs, nf, times, nt, rows, midpoint);
    free(rows);
    return r;
}
}
