// consumer_kernels.hpp -- consumers of the MFCC matrix and the linear resampler (8f-4):
//   stop-word net     2fa/audio/word/c/stop_detector.c:36-50, audio_classifier_inference.c:18-90
//   speaker GMM LLR   2fa/audio/pico-audio/src/speaker_gmm.c:29-141
//   upsampleLinear    sync/particle/main.cpp:62-77
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace dsp {

constexpr int kStopMaxUnits = 16;    // widest hidden layer the tail kernel is built for
constexpr int kStopFusedUnits = 4;   // first-layer width the fused MFCC epilogue is built for (model_params.h: 4 units)

struct StopModelDev {
    int n_coef, max_frames;          // 13, 500
    int units[4];                    // 4, 2, 2, 1
    const float *mean, *div;         // [n_coef * max_frames] coefficient-major; div = scaler scale with 0 replaced by 1
    const float *kernel[4];          // (in, out) row-major
    const float *bias[4];
    const double *pad;               // [max_frames + 1][units[0]]: layer-1 contribution of the zero-padded frames t >= T
    // The fused MFCC epilogue (mfcc512_wave_kernel<POOL = 2>, units[0] <= kStopFusedUnits; nullptr otherwise): layer 1 with the
    // scaler folded in, w (x - mean) / div = x A + B per input and unit:
    const float *fold_a;             // [n_coef * max_frames][kStopFusedUnits]: A[i][j] = w[i][j] / div[i] (0 past units[0])
    const double *pad_b;             // [max_frames + 1][units[0]]: pad[T][j] + sum over the live inputs (t < T) of B[i][j] = -mean[i] A[i][j]
};

// prob[c] = net(coefficient-major, zero-padded view of mfcc[c][T][n_coef]); one wave per clip
hipError_t launch_stop_tail(const StopModelDev &m, const float *mfcc, long n_clips, int T, float *prob, hipStream_t stream);

struct GmmDev {
    int k, d;                        // 32, 13
    const int8_t *means;             // [k][d]  Q6
    const int32_t *inv_covs;         // [k][d]  Q11
    const int16_t *log_consts;       // [k]     Q8
};

// llr_mean[c] = (sum_t (LL_target - LL_ubm)(frame t)) / T in int64, label[c] = llr_mean > threshold; optional
// per-frame log-likelihoods ll_target / ll_ubm [n_clips][T].  One wave per clip, one lane per frame.
hipError_t launch_speaker_llr(const GmmDev &target, const GmmDev &ubm, const float *mfcc, long n_clips, int T,
                              long long threshold, long long *llr_mean, int *labels, long long *ll_target,
                              long long *ll_ubm, hipStream_t stream);

// out[c][i] = in[c][lo] + (in[c][hi] - in[c][lo]) * frac, the reference's fp32 operation order
hipError_t launch_upsample_linear(const float *in, long n_clips, int old_size, long in_stride, float *out, int new_size,
                                  long out_stride, hipStream_t stream);

// fft_real_forward (mfcc.c:16-95): out[f][n_fft] complex interleaved of the frame_length real samples at in + f * in_stride, zero-padded
hipError_t launch_fft_real_forward(const float *in, long n_frames, int frame_length, long in_stride, int n_fft, float *out, hipStream_t stream);

}  // namespace dsp
