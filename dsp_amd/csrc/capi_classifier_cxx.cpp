// capi_classifier_cxx.cpp -- the reference's own C++-linkage names (sync/lib/classifier.h:14-19), exported by
// libdsp_amd.so so that sync/sync.cpp:202 links against the library in place of classifier.cpp + PlainFFT.cpp.
// Signatures, ownership (malloc'd results the caller frees) and return conventions are the reference's; the
// arithmetic is the gfx950 path behind the C entry points of include/dsp_amd.h.  Declared in
// include/dsp_amd_classifier.h.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/dsp_amd.h"
#include "../../include/dsp_amd_classifier.h"

namespace {

constexpr int kBins = 129;     // nfft / 2 + 1, classifier.cpp:235

bool verbose()
{
    // the reference prints its findings (classifier.cpp:85,104-106,118,123); the library only does so on request
    static const bool on = std::getenv("DSP_AMD_VERBOSE") != nullptr;
    return on;
}

void complain(const char *who)
{
    std::fprintf(stderr, "libdsp_amd: %s: %s\n", who, dsp_last_error());
}

}  // namespace

bool butter_bandpass(float lowcut, float highcut, float *b, float *a)
{
    // classifier.cpp:138-191: float stores of the literal tables
    double bd[9], ad[9];
    if (!dsp_butter_bandpass((double)lowcut, (double)highcut, bd, ad)) {
        std::printf("invalid bandpass range");        // classifier.cpp:186
        return false;
    }
    for (int i = 0; i < 9; ++i) { b[i] = (float)bd[i]; a[i] = (float)ad[i]; }
    return true;
}

void butter_bandpass_filter(float *data, int n, float *b, float *a, float *output)
{
    if (n <= 0) return;
    if (dsp_butter_bandpass_filter_f32(data, 1, n, n, b, a, output) < 0) {
        complain("butter_bandpass_filter");
        std::memset(output, 0, sizeof(float) * (size_t)n);
    }
}

void compute_spectrogram(float *signal, int signal_length, int fs, float **frequencies, float **times, float ***Sxx,
                         int *freq_bins, int *time_bins)
{
    // classifier.cpp:235-245: the caller receives malloc'd axes and one malloc'd row per frequency bin
    const int T = signal_length < 256 ? 0 : (signal_length - 256) / 224 + 1;
    *freq_bins = kBins;
    *time_bins = T;
    *frequencies = (float *)std::malloc(sizeof(float) * kBins);
    *times = (float *)std::malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
    *Sxx = (float **)std::malloc(sizeof(float *) * kBins);
    float *flat = (float *)std::calloc((size_t)kBins * (size_t)(T > 0 ? T : 1), sizeof(float));
    const int rc = dsp_compute_spectrogram_f32(signal, signal_length, fs, *frequencies, *times, flat);
    if (rc < 0) complain("compute_spectrogram");
    for (int i = 0; i < kBins; ++i) {
        (*Sxx)[i] = (float *)std::malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
        if (T > 0) std::memcpy((*Sxx)[i], flat + (size_t)i * T, sizeof(float) * (size_t)T);
    }
    std::free(flat);
}

float sum_intense(float lower, float upper, float half_range, float *frequencies, int freq_bins, float *times, int time_bins,
                  float **intensity_dB_filtered, float midpoint)
{
    if (freq_bins <= 0 || time_bins <= 0) return 0.0f;
    float *flat = (float *)std::malloc(sizeof(float) * (size_t)freq_bins * (size_t)time_bins);
    for (int i = 0; i < freq_bins; ++i) std::memcpy(flat + (size_t)i * time_bins, intensity_dB_filtered[i], sizeof(float) * (size_t)time_bins);
    float out = 0.0f;
    if (dsp_sum_intense_f32(lower, upper, half_range, frequencies, freq_bins, times, time_bins, flat, midpoint, &out) < 0) {
        complain("sum_intense");
        out = 0.0f;
    }
    std::free(flat);
    return out;
}

float *find_midpoints(float *data, int num_frames, int samplingFreq, int *num_midpoints)
{
    float tmp[64];
    int n = dsp_find_midpoints(data, num_frames, samplingFreq, tmp, 64);
    if (n < 0) { complain("find_midpoints"); n = 0; }
    float *m = (float *)std::malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));      // classifier.cpp:557: the caller frees it
    if (m && n > 0) std::memcpy(m, tmp, sizeof(float) * (size_t)n);
    *num_midpoints = n;
    return m;
}

int classify(float *data, int data_size)
{
    if (!data || data_size <= 0) return 0;
    int label = 0;
    dsp_classify_trace tr;
    if (dsp_classify_batch_host(data, 1, data_size, data_size, &label, &tr) < 0) {
        complain("classify");
        return 0;                                  // the reference's failure value (classifier.cpp:87-91)
    }
    if (verbose()) {                               // the reference's report, classifier.cpp:85-123
        std::printf("Number of Midpoints: %d\n", tr.n_midpoints);
        for (int k = 0; k < tr.n_midpoints; ++k) {
            std::printf("Above intensities: %f\n", tr.sums[k][0]);
            std::printf("Middle intensities: %f\n", tr.sums[k][1]);
            std::printf("Below intensities: %f\n", tr.sums[k][2]);
            if (label && tr.sums[k][1] < 100 && tr.sums[k][0] > 200 && tr.sums[k][2] > 80) break;
        }
        std::printf(label ? "We have a Scrub Jay! :)\n" : "We have no Scrub Jay! :(\n");
    }
    return label;
}
