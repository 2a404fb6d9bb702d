// svm_kernels.hpp -- pooling (cepstrum/scrubjay_infer.c:36-66) and the ONNX Scaler ->
// SVMClassifier(RBF) -> Platt tail of cepstrum/scrubjay_svm.onnx (scrubjay_infer.c:105-141).
#pragma once

#include <hip/hip_runtime_api.h>

namespace dsp {

struct SvmModelDev {
    int n_features;      // 2 * n_coef
    int n_sv;
    float gamma, rho, prob_a, prob_b;
    const float *offset; // [n_features]   Scaler: (x - offset) * scale
    const float *scale;  // [n_features]
    const float *sv;     // [n_sv][n_features]
    const float *coef;   // [n_sv]
};

// feat[c][2*n_coef] = mean | population std over the T frames of clip c (float64 accumulators)
hipError_t launch_mfcc_stats(const float *mfcc, long n_clips, int T, int n_coef, float *feat, hipStream_t stream);
// label (0/1), decision value and P(label 1) per clip
hipError_t launch_svm_predict(const SvmModelDev &m, const float *feat, long n_clips, int *labels, float *decision,
                              float *prob1, hipStream_t stream);

}  // namespace dsp
