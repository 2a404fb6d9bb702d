// svm_kernels.hip -- a12 pooling + a13 RBF-SVM, one wave per clip.
#include <hip/hip_runtime.h>

#include "svm_kernels.hpp"

#pragma clang fp contract(off)

namespace dsp {

// cepstrum/scrubjay_infer.c:36-66: per coefficient sum and sum of squares in double over the
// frames in order, mean = s/T, var = q/T - mean^2, std = sqrtf(max(var, 0)).
__global__ __launch_bounds__(64) void mfcc_stats_kernel(const float *__restrict__ mfcc, long n_clips, int T, int n_coef,
                                                        float *__restrict__ feat)
{
    const long clip = blockIdx.x;
    const int c = threadIdx.x;
    if (clip >= n_clips || c >= n_coef) return;
    const float *p = mfcc + clip * (long)T * n_coef + c;
    double s = 0.0, q = 0.0;
    for (int t = 0; t < T; ++t) {
        const double v = (double)p[(long)t * n_coef];
        s = s + v;
        q = q + v * v;
    }
    const double mean = s / (double)T;
    const double var = q / (double)T - mean * mean;
    feat[clip * 2L * n_coef + c] = (float)mean;
    feat[clip * 2L * n_coef + n_coef + c] = sqrtf((float)(var > 0 ? var : 0));
}

hipError_t launch_mfcc_stats(const float *mfcc, long n_clips, int T, int n_coef, float *feat, hipStream_t stream)
{
    if (n_clips <= 0) return hipSuccess;
    if (n_coef > 64 || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mfcc_stats_kernel, dim3((unsigned)n_clips), dim3(64), 0, stream, mfcc, n_clips, T, n_coef, feat);
    return hipGetLastError();
}

// ONNX Scaler + SVMClassifier (two classes, RBF, Platt): lane s owns support vector s.
__global__ __launch_bounds__(64) void svm_kernel(const SvmModelDev m, const float *__restrict__ feat, long n_clips,
                                                 int *__restrict__ labels, float *__restrict__ decision, float *__restrict__ prob1)
{
    __shared__ float z[256];
    const long clip = blockIdx.x;
    if (clip >= n_clips) return;
    const int lane = threadIdx.x;
    for (int j = lane; j < m.n_features; j += 64) z[j] = (feat[clip * (long)m.n_features + j] - m.offset[j]) * m.scale[j];
    __syncthreads();
    float term = 0.0f;
    for (int s = lane; s < m.n_sv; s += 64) {
        const float *sv = m.sv + (long)s * m.n_features;
        float d2 = 0.0f;
        for (int j = 0; j < m.n_features; ++j) {
            const float d = z[j] - sv[j];
            d2 = d2 + d * d;
        }
        term = term + m.coef[s] * expf(-m.gamma * d2);
    }
    for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
    if (lane == 0) {
        const float score = term + m.rho;                    // ONNX "sum + rho" = libsvm's sum - model.rho (rho = intercept)
        int label;
        float p1;
        svm_binary_tail(score, m.prob_a, m.prob_b, label, p1);
        labels[clip] = label;
        if (decision) decision[clip] = score;
        if (prob1) prob1[clip] = p1;
    }
}

hipError_t launch_svm_predict(const SvmModelDev &m, const float *feat, long n_clips, int *labels, float *decision,
                              float *prob1, hipStream_t stream)
{
    if (n_clips <= 0) return hipSuccess;
    if (m.n_features > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(svm_kernel, dim3((unsigned)n_clips), dim3(64), 0, stream, m, feat, n_clips, labels, decision, prob1);
    return hipGetLastError();
}

}  // namespace dsp
