// svm_kernels.hpp -- pooling (cepstrum/scrubjay_infer.c:36-66) and the ONNX Scaler ->
// SVMClassifier(RBF) -> Platt tail of cepstrum/scrubjay_svm.onnx (scrubjay_infer.c:105-141).
#pragma once

#include <hip/hip_runtime.h>

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

// The two-class tail of libsvm's svm_predict / svm_predict_probability (svm.cpp; sklearn's SVC is libsvm, ONNX Runtime's
// SVMClassifier ports the same routines), pinned by tests/golden/svm_libsvm_ref.npz:
//   label  the pairwise VOTE: a decision value > 0 votes for the first class (0), anything else for class 1 -- what
//          sklearn's .predict (cepstrum/run.py) returns, and what ORT's output_label is in SVC mode (votes, not probabilities);
//   p1     P(class 1) = 1 - p[0] where p comes from multiclass_probability on the Platt pair r01 = 1 / (1 + exp(A d + B))
//          clamped to [1e-7, 1 - 1e-7]: that routine is an ITERATION from (1/2, 1/2) with tolerance 0.005 / k, so it
//          returns exactly (0.5, 0.5) for |r01 - 0.5| < ~0.0025 and the sigmoid to ~1e-3..1e-7 elsewhere -- not the sigmoid.
// The iteration runs in double like libsvm's (a handful of scalar steps per clip).
__host__ __device__ inline void svm_binary_tail(float score, float prob_a, float prob_b, int &label, float &p1)
{
    label = score > 0.0f ? 0 : 1;
    const float fApB = score * prob_a + prob_b;
    float r01f = fApB >= 0.0f ? expf(-fApB) / (1.0f + expf(-fApB)) : 1.0f / (1.0f + expf(fApB));     // sigmoid_predict
    r01f = fminf(fmaxf(r01f, 1e-7f), 1.0f - 1e-7f);
    const double r01 = (double)r01f, r10 = 1.0 - r01;
    // multiclass_probability(k = 2, r, p)
    const double Q00 = r10 * r10, Q11 = r01 * r01, Q01 = -r10 * r01;
    double p0 = 0.5, p1d = 0.5;
    const double eps = 0.005 / 2;
    for (int iter = 0; iter < 100; ++iter) {
        double Qp0 = Q00 * p0 + Q01 * p1d, Qp1 = Q01 * p0 + Q11 * p1d;
        double pQp = p0 * Qp0 + p1d * Qp1;
        const double e0 = Qp0 - pQp, e1 = Qp1 - pQp;
        const double err = (e0 < 0 ? -e0 : e0) > (e1 < 0 ? -e1 : e1) ? (e0 < 0 ? -e0 : e0) : (e1 < 0 ? -e1 : e1);
        if (err < eps) break;
        {   // t = 0
            const double diff = (-Qp0 + pQp) / Q00;
            p0 += diff;
            pQp = (pQp + diff * (diff * Q00 + 2 * Qp0)) / (1 + diff) / (1 + diff);
            Qp0 = (Qp0 + diff * Q00) / (1 + diff); p0 /= (1 + diff);
            Qp1 = (Qp1 + diff * Q01) / (1 + diff); p1d /= (1 + diff);
        }
        {   // t = 1
            const double diff = (-Qp1 + pQp) / Q11;
            p1d += diff;
            pQp = (pQp + diff * (diff * Q11 + 2 * Qp1)) / (1 + diff) / (1 + diff);
            Qp0 = (Qp0 + diff * Q01) / (1 + diff); p0 /= (1 + diff);
            Qp1 = (Qp1 + diff * Q11) / (1 + diff); p1d /= (1 + diff);
        }
    }
    p1 = (float)p1d;
}

// feat[c][2*n_coef] = mean | population std over the T frames of clip c (float64 accumulators)
hipError_t launch_mfcc_stats(const float *mfcc, long n_clips, int T, int n_coef, float *feat, hipStream_t stream);
// label (0/1), decision value and P(label 1) per clip
hipError_t launch_svm_predict(const SvmModelDev &m, const float *feat, long n_clips, int *labels, float *decision,
                              float *prob1, hipStream_t stream);

}  // namespace dsp
