/*
 * dsp_amd_classifier.h -- the C++-linkage names of the reference's donut classifier, exported by
 * libdsp_amd.so with the signatures of sync/lib/classifier.h:14-19, so that sync/sync.cpp:202 (and any
 * other caller of that header) links against the library INSTEAD of classifier.cpp + PlainFFT.cpp with
 * no source change: the caller keeps including the reference's own classifier.h.  This header repeats
 * those declarations for callers that do not have the reference tree; it is C++ only (the reference
 * header has no extern "C").  Plain pointers and sizes, host memory, the reference's ownership rules.
 *
 * Results are bit-identical to the compiled reference (tests/test_boundary_cxx.py); all arithmetic
 * runs in the gfx950 kernels of dsp_amd/csrc/classify_kernels.hip, and without a GPU every function
 * fails loudly (reason in dsp_last_error()) while keeping the reference's return convention.
 */
#ifndef DSP_AMD_CLASSIFIER_H
#define DSP_AMD_CLASSIFIER_H
#ifdef __cplusplus

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* classifier.h:14, classifier.cpp:138-191: the two literal 16 kHz tables as floats; false (and
 * "invalid bandpass range" on stdout, as the reference prints) for any other band.              */
bool butter_bandpass(float lowcut, float highcut, float *b, float *a);

/* classifier.h:15, classifier.cpp:193-219: direct form II from zero state, fp32, the reference's
 * operation order.  Caller owns data[n] and output[n].                                          */
void butter_bandpass_filter(float *data, int n, float *b, float *a, float *output);

/* classifier.h:16, classifier.cpp:221-368: nperseg 256, hop 224, detrend, periodic Tukey(0.25), PSD.
 * *frequencies[129], *times[T], (*Sxx)[129] rows of T floats are malloc'd here and freed by the
 * CALLER (classifier.cpp:239-245, 126-133); *freq_bins = 129, *time_bins = T = (n-256)/224+1.
 * On failure (no GPU) the arrays are still allocated, Sxx rows zero-filled.                      */
void compute_spectrogram(float *signal, int signal_length, int fs, float **frequencies, float **times,
                         float ***Sxx, int *freq_bins, int *time_bins);

/* classifier.h:17, classifier.cpp:370-431: sum of the non-NaN cells of rows [lower, upper] Hz and
 * columns [midpoint - half_range, midpoint + half_range] s, added in (row, column) order.        */
float sum_intense(float lower, float upper, float half_range, float *frequencies, int freq_bins, float *times,
                  int time_bins, float **intensity_dB_filtered, float midpoint);

/* classifier.h:18, classifier.cpp:433-598: malloc'd array of *num_midpoints cluster mean times (the
 * caller frees it; never NULL unless malloc fails, like the reference).                          */
float *find_midpoints(float *data, int num_frames, int samplingFreq, int *num_midpoints);

/* classifier.h:19, classifier.cpp:9-136: 1 if the scrub-jay rule fires for the 16 kHz clip, else 0. */
int classify(float *data, int data_size);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#endif /* __cplusplus */
#endif /* DSP_AMD_CLASSIFIER_H */
