// consumer_kernels.hip -- gfx950 kernels for the consumers of the MFCC matrix and the resampler.
// All three are tiny next to the MFCC chain (a 98 x 13 matrix per clip); they exist so that a clip
// goes from PCM to its decision without leaving HBM.
#include <hip/hip_runtime.h>

#include "consumer_kernels.hpp"

#pragma clang fp contract(off)

namespace dsp {

// ---- stop-word net ---------------------------------------------------------------------------
// The reference standardises all n_coef * max_frames inputs and sums layer 1 sequentially in fp32
// (audio_classifier_inference.c:25-33).  Here a wave sums only the T x n_coef live inputs (float64
// partial sums, so the result does not depend on the lane split); the zero-padded inputs t >= T
// contribute a constant per T that the host precomputed in float64 (StopModelDev::pad).  Layers 2-4
// (<= 16 units) run on lane 0 in the reference's order.
__global__ __launch_bounds__(256) void stop_tail_kernel(const StopModelDev m, const float *__restrict__ mfcc, long n_clips, int T,
                                                        float *__restrict__ prob)
{
    const int lane = threadIdx.x & 63;
    const long clip = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (clip >= n_clips) return;
    const int u1 = m.units[0];
    const int Tc = T < m.max_frames ? T : m.max_frames;          // stop_detector.c:26-30
    const float *x = mfcc + clip * (long)T * m.n_coef;
    double acc[kStopMaxUnits];
#pragma unroll
    for (int j = 0; j < kStopMaxUnits; ++j) acc[j] = 0.0;
    const int live = Tc * m.n_coef;
    for (int p = lane; p < live; p += 64) {
        const int t = p / m.n_coef, c = p - t * m.n_coef;
        const int i = c * m.max_frames + t;                     // stop_detector.c:48: coefficient-major index
        const float xs = (x[p] - m.mean[i]) / m.div[i];     // audio_classifier_inference.c:46
        const float *w = m.kernel[0] + (long)i * u1;
#pragma unroll
        for (int j = 0; j < kStopMaxUnits; ++j)
            if (j < u1) acc[j] += (double)w[j] * (double)xs;
    }
#pragma unroll
    for (int j = 0; j < kStopMaxUnits; ++j)
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    if (lane != 0) return;
    float h[2][kStopMaxUnits];
    for (int j = 0; j < u1; ++j) {
        const float s = (float)((double)m.bias[0][j] + m.pad[(long)Tc * u1 + j] + acc[j]);
        h[0][j] = s > 0.0f ? s : 0.0f;
    }
    int n_in = u1;
    for (int l = 1; l < 4; ++l) {                                // dense_forward, :18-35
        const int n_out = m.units[l];
        const float *src = h[(l - 1) & 1];
        float *dst = h[l & 1];
        for (int j = 0; j < n_out; ++j) {
            float s = m.bias[l][j];
            for (int i = 0; i < n_in; ++i) s = s + m.kernel[l][i * n_out + j] * src[i];
            dst[j] = (l < 3 && !(s > 0.0f)) ? 0.0f : s;
        }
        n_in = n_out;
    }
    prob[clip] = 1.0f / (1.0f + expf(-h[1][0]));                 // :13-15
}

hipError_t launch_stop_tail(const StopModelDev &m, const float *mfcc, long n_clips, int T, float *prob, hipStream_t stream)
{
    if (n_clips <= 0) return hipSuccess;
    for (int l = 0; l < 4; ++l)
        if (m.units[l] <= 0 || m.units[l] > kStopMaxUnits) return hipErrorInvalidValue;
    if (T < 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stop_tail_kernel, dim3((unsigned)((n_clips + 3) / 4)), dim3(256), 0, stream, m, mfcc, n_clips, T, prob);
    return hipGetLastError();
}

// ---- speaker GMM -------------------------------------------------------------------------------
// Integer path, bit-exact: x Q6 = (int16)(x * 64) (speaker_gmm.c:118-122), per mixture
// sum_d (x - mean)^2 * inv_cov in int64 (Q23), >> 15, / 2, log_const - that, max over mixtures (:29-50).
constexpr int kGmmMaxD = 16, kGmmMaxK = 64;

struct GmmLds {
    int8_t means[kGmmMaxK * kGmmMaxD];
    int32_t inv_covs[kGmmMaxK * kGmmMaxD];
    int16_t log_consts[kGmmMaxK];
};

__device__ __forceinline__ long long gmm_ll(const GmmLds &g, int k_n, int d_n, const int (&x)[kGmmMaxD])
{
    long long best = LLONG_MIN;
    for (int k = 0; k < k_n; ++k) {
        long long sum_sq = 0;
#pragma unroll
        for (int d = 0; d < kGmmMaxD; ++d) {
            if (d < d_n) {
                const int diff = x[d] - (int)g.means[k * d_n + d];            // |diff| < 2^16
                sum_sq += (long long)diff * (long long)diff * (long long)g.inv_covs[k * d_n + d];
            }
        }
        sum_sq >>= 15;
        sum_sq /= 2;
        const long long term = (long long)g.log_consts[k] - sum_sq;
        best = term > best ? term : best;
    }
    return best;
}

__global__ __launch_bounds__(256) void speaker_llr_kernel(const GmmDev target, const GmmDev ubm, const float *__restrict__ mfcc,
                                                          long n_clips, int T, long long threshold, long long *__restrict__ llr_mean,
                                                          int *__restrict__ labels, long long *__restrict__ ll_target,
                                                          long long *__restrict__ ll_ubm)
{
    __shared__ GmmLds gt, gu;
    for (int i = threadIdx.x; i < target.k * target.d; i += 256) {
        gt.means[i] = target.means[i]; gt.inv_covs[i] = target.inv_covs[i];
        gu.means[i] = ubm.means[i]; gu.inv_covs[i] = ubm.inv_covs[i];
    }
    for (int i = threadIdx.x; i < target.k; i += 256) { gt.log_consts[i] = target.log_consts[i]; gu.log_consts[i] = ubm.log_consts[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long clip = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (clip >= n_clips) return;
    const int d_n = target.d;
    long long sum = 0;
    for (int t = lane; t < T; t += 64) {
        const float *f = mfcc + (clip * (long)T + t) * d_n;
        int x[kGmmMaxD];
#pragma unroll
        for (int d = 0; d < kGmmMaxD; ++d) x[d] = d < d_n ? (int)(short)(int)(f[d] * 64.0f) : 0;   // low 16 bits of the int32 truncation
        const long long lt = gmm_ll(gt, target.k, d_n, x), lu = gmm_ll(gu, ubm.k, d_n, x);
        if (ll_target) ll_target[clip * (long)T + t] = lt;
        if (ll_ubm) ll_ubm[clip * (long)T + t] = lu;
        sum += lt - lu;                                                                           // :104-108
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) {
        const long long mean = sum / (long long)T;                                                // :135
        llr_mean[clip] = mean;
        if (labels) labels[clip] = mean > threshold ? 1 : 0;                                      // :138-141
    }
}

hipError_t launch_speaker_llr(const GmmDev &target, const GmmDev &ubm, const float *mfcc, long n_clips, int T,
                              long long threshold, long long *llr_mean, int *labels, long long *ll_target,
                              long long *ll_ubm, hipStream_t stream)
{
    if (n_clips <= 0) return hipSuccess;
    if (T <= 0 || target.d != ubm.d || target.k != ubm.k || target.d > kGmmMaxD || target.k > kGmmMaxK) return hipErrorInvalidValue;
    hipLaunchKernelGGL(speaker_llr_kernel, dim3((unsigned)((n_clips + 3) / 4)), dim3(256), 0, stream, target, ubm, mfcc, n_clips, T,
                       threshold, llr_mean, labels, ll_target, ll_ubm);
    return hipGetLastError();
}

// ---- linear resampler ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_linear_kernel(const float *__restrict__ in, long n_clips, int old_size, long in_stride,
                                                              float *__restrict__ out, int new_size, long out_stride)
{
    const long clip = blockIdx.y;
    const float step = (float)(old_size - 1) / (float)(new_size - 1);       // main.cpp:66
    const float *src = in + clip * in_stride;
    float *dst = out + clip * out_stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < new_size; i += gridDim.x * 256) {
        const float old_index = (float)i * step;
        const int lo = (int)floorf(old_index);
        const int hi = lo == old_size - 1 ? old_size - 1 : lo + 1;
        const float frac = old_index - (float)lo;
        const float a = src[lo], b = src[hi];
        dst[i] = a + (b - a) * frac;
    }
}

hipError_t launch_upsample_linear(const float *in, long n_clips, int old_size, long in_stride, float *out, int new_size,
                                  long out_stride, hipStream_t stream)
{
    if (n_clips <= 0 || new_size <= 0) return hipSuccess;
    if (old_size < 1 || new_size < 2 || n_clips > 65535) return hipErrorInvalidValue;
    const unsigned gx = (unsigned)((new_size + 255) / 256 < 64 ? (new_size + 255) / 256 : 64);
    hipLaunchKernelGGL(upsample_linear_kernel, dim3(gx, (unsigned)n_clips), dim3(256), 0, stream, in, n_clips, old_size, in_stride, out,
                       new_size, out_stride);
    return hipGetLastError();
}

// ---- fft_real_forward (2fa/audio/word/c/mfcc.c:16-95) as its own entry point -------------------------------------------------
// The reference's non-static helper: frame_length real samples, zero-padded to n_fft, forward transform, ALL n_fft complex bins
// interleaved [re, im].  Inside compute_mfcc the transform lives in the MFCC kernels' registers (mfcc_kernels.hip); this batch kernel
// serves callers that link the symbol itself: one 256-thread block per frame, radix-2 Stockham through LDS (9 passes at n_fft = 512),
// twiddles from sincospif (the reference runs a float32 recurrence; the gate is 1e-4 of the frame's L-inf norm, as for the MFCCs).
__global__ __launch_bounds__(256) void fft_real_forward_kernel(const float *__restrict__ in, long n_frames, int frame_length, long in_stride, int n_fft,
                                                               float *__restrict__ out)
{
    extern __shared__ float2 fbuf[];                       // [2][n_fft]
    const long fr = blockIdx.x;
    if (fr >= n_frames) return;
    float2 *a = fbuf, *b = fbuf + n_fft;
    for (int i = threadIdx.x; i < n_fft; i += 256) a[i] = make_float2(i < frame_length ? in[fr * in_stride + i] : 0.0f, 0.0f);
    __syncthreads();
    for (int ns = 1; ns < n_fft; ns <<= 1) {               // butterfly j: k = j % ns, inputs x[j], x[j + n/2] W_{2 ns}^k, outputs y[(j - k) 2 + k], + ns
        for (int j = threadIdx.x; j < n_fft / 2; j += 256) {
            const int k = j % ns;
            float sn, cs;
            sincospif(-(float)k / (float)ns, &sn, &cs);
            const float2 u = a[j], v = a[j + n_fft / 2];
            const float2 t = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
            const int o = (j - k) * 2 + k;
            b[o] = make_float2(u.x + t.x, u.y + t.y);
            b[o + ns] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
        float2 *sw = a; a = b; b = sw;
    }
    for (int i = threadIdx.x; i < n_fft; i += 256) reinterpret_cast<float2 *>(out)[fr * n_fft + i] = a[i];
}

hipError_t launch_fft_real_forward(const float *in, long n_frames, int frame_length, long in_stride, int n_fft, float *out, hipStream_t stream)
{
    if (n_frames <= 0) return hipSuccess;
    if (n_fft < 2 || (n_fft & (n_fft - 1)) || n_fft > 4096 || frame_length < 0 || frame_length > n_fft || n_frames >= (1L << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fft_real_forward_kernel, dim3((unsigned)n_frames), dim3(256), (size_t)2 * n_fft * sizeof(float2), stream, in, n_frames, frame_length,
                       in_stride, n_fft, out);
    return hipGetLastError();
}

}  // namespace dsp
