// capi_consumers.cpp -- C ABI of the consumers of the MFCC matrix (stop-word net, speaker GMM) and of the
// linear resampler (include/dsp_amd.h).  Same rules as capi.cpp: models own
// device copies of their parameters, there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "capi_util.hpp"
#include "consumer_kernels.hpp"

using dsp::capi_fail;

namespace {

int check_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return capi_fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback");
    if (device < 0 || device >= n) return capi_fail(DSP_EINVAL, "device index out of range");
    return DSP_OK;
}

}  // namespace

struct dsp_stop_model {
    int device = 0;
    dsp::StopModelDev m{};
    void *d_blob = nullptr;
    // workspace of dsp_classify_signal_batch_device / dsp_classify_signal
    float *d_mfcc = nullptr, *d_sig = nullptr, *d_prob = nullptr;
    size_t mfcc_cap = 0, sig_cap = 0;
    dsp_mfcc_plan *plan = nullptr;     // default plan of dsp_classify_signal
    std::mutex mu;
};

struct dsp_speaker_model {
    int device = 0;
    dsp::GmmDev target{}, ubm{};
    void *d_blob = nullptr;
};

extern "C" {

int dsp_stop_model_create(const dsp_stop_model_params *p, int device, dsp_stop_model **out)
{
    if (!out) return capi_fail(DSP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!p || p->n_coef <= 0 || p->max_frames <= 0 || !p->scaler_mean || !p->scaler_scale) return capi_fail(DSP_EINVAL, "bad stop-model parameters");
    for (int l = 0; l < 4; ++l)
        if (p->units[l] <= 0 || p->units[l] > dsp::kStopMaxUnits || !p->kernel[l] || !p->bias[l])
            return capi_fail(DSP_EINVAL, "stop-model layers must have 1..16 units and non-NULL parameters");
    if (p->units[3] != 1) return capi_fail(DSP_EINVAL, "the last layer must have one unit (sigmoid output)");
    int rc = check_device(device);
    if (rc < 0) return rc;
    DSP_ON_DEVICE(device);
    const size_t n_in = (size_t)p->n_coef * p->max_frames, u1 = p->units[0];
    // divisor with the reference's zero guard (audio_classifier_inference.c:44-45)
    std::vector<float> div(n_in);
    for (size_t i = 0; i < n_in; ++i) div[i] = p->scaler_scale[i] == 0.0f ? 1.0f : p->scaler_scale[i];
    // layer-1 contribution of the zero-padded frames: pad[T][j] = sum_{c, t >= T} W[c*max+t][j] * fl((0 - mean) / div)
    std::vector<double> pad((size_t)(p->max_frames + 1) * u1, 0.0);
    for (int t = p->max_frames - 1; t >= 0; --t)
        for (size_t j = 0; j < u1; ++j) {
            double s = pad[(size_t)(t + 1) * u1 + j];
            for (int c = 0; c < p->n_coef; ++c) {
                const size_t i = (size_t)c * p->max_frames + t;
                const float xs = (0.0f - p->scaler_mean[i]) / div[i];
                s += (double)p->kernel[0][i * u1 + j] * (double)xs;
            }
            pad[(size_t)t * u1 + j] = s;
        }
    // the fused epilogue's form of layer 1 (consumer_kernels.hpp): A = w / div per input, and per T the constant the live
    // inputs' B = -mean A add up to, on top of the padded frames' contribution
    const bool foldable = u1 <= (size_t)dsp::kStopFusedUnits;
    std::vector<float> fold_a;
    std::vector<double> pad_b;
    if (foldable) {
        fold_a.assign(n_in * dsp::kStopFusedUnits, 0.0f);
        pad_b.assign(pad.size(), 0.0);
        std::vector<double> live(u1, 0.0);                          // sum_{t' < t, c} B
        for (int t = 0; t <= p->max_frames; ++t) {
            for (size_t j = 0; j < u1; ++j) pad_b[(size_t)t * u1 + j] = pad[(size_t)t * u1 + j] + live[j];
            if (t == p->max_frames) break;
            for (int c = 0; c < p->n_coef; ++c) {
                const size_t i = (size_t)c * p->max_frames + t;
                for (size_t j = 0; j < u1; ++j) {
                    const double a = (double)p->kernel[0][i * u1 + j] / (double)div[i];
                    fold_a[i * dsp::kStopFusedUnits + j] = (float)a;
                    live[j] += -(double)p->scaler_mean[i] * (double)(float)a;      // B pairs with the ROUNDED A the kernel multiplies by
                }
            }
        }
    }
    // one device blob: doubles first (alignment), then floats
    size_t n_f = 2 * n_in + fold_a.size();
    size_t fan_in = n_in;
    for (int l = 0; l < 4; ++l) { n_f += fan_in * p->units[l] + p->units[l]; fan_in = p->units[l]; }
    const size_t bytes = (pad.size() + pad_b.size() + 1) * sizeof(double) + n_f * sizeof(float);
    auto *m = new dsp_stop_model;
    m->device = device;
    if (hipMalloc(&m->d_blob, bytes) != hipSuccess) { delete m; return capi_fail(DSP_ENOMEM, "hipMalloc"); }
    std::vector<char> host(bytes);
    char *h = host.data();
    char *d = static_cast<char *>(m->d_blob);
    size_t off = 0;
    auto put = [&](const void *src, size_t n) { std::memcpy(h + off, src, n); const void *dev = d + off; off += n; return dev; };
    m->m.n_coef = p->n_coef;
    m->m.max_frames = p->max_frames;
    m->m.pad = static_cast<const double *>(put(pad.data(), pad.size() * sizeof(double)));
    m->m.pad_b = foldable ? static_cast<const double *>(put(pad_b.data(), pad_b.size() * sizeof(double))) : nullptr;
    if ((pad.size() + pad_b.size()) % 2) { static const double zero = 0.0; (void)put(&zero, sizeof(double)); }      // keep fold_a 16-byte aligned
    m->m.fold_a = foldable ? static_cast<const float *>(put(fold_a.data(), fold_a.size() * sizeof(float))) : nullptr;
    m->m.mean = static_cast<const float *>(put(p->scaler_mean, n_in * 4));
    m->m.div = static_cast<const float *>(put(div.data(), n_in * 4));
    fan_in = n_in;
    for (int l = 0; l < 4; ++l) {
        m->m.units[l] = p->units[l];
        m->m.kernel[l] = static_cast<const float *>(put(p->kernel[l], fan_in * p->units[l] * 4));
        m->m.bias[l] = static_cast<const float *>(put(p->bias[l], (size_t)p->units[l] * 4));
        fan_in = p->units[l];
    }
    if (hipMemcpy(m->d_blob, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(m->d_blob); delete m; return capi_fail(DSP_EHIP, "hipMemcpy"); }
    *out = m;
    return DSP_OK;
}

void dsp_stop_model_destroy(dsp_stop_model *m)
{
    if (!m) return;
    dsp::DeviceScope dsp_device_scope_(m->device);
    if (m->plan) dsp_mfcc_plan_destroy(m->plan);
    for (void *p : {(void *)m->d_blob, (void *)m->d_mfcc, (void *)m->d_sig, (void *)m->d_prob})
        if (p) hipFree(p);
    delete m;
}

int dsp_stop_predict_device(dsp_stop_model *m, const float *d_mfcc, long n_clips, int frames_per_clip, float *d_prob, void *stream)
{
    if (!m || n_clips < 0 || frames_per_clip < 0 || (n_clips > 0 && (!d_prob || (frames_per_clip > 0 && !d_mfcc))))
        return capi_fail(DSP_EINVAL, "bad argument");
    DSP_CAPI_HIP(dsp::launch_stop_tail(m->m, d_mfcc, n_clips, frames_per_clip, d_prob, (hipStream_t)stream));
    return DSP_OK;
}

}  // extern "C"

// classify_signal over a batch; in_kind 0 = float samples, 1 / 2 / 3 = int16 mono / stereo channel 0 / stereo average (what
// main_test.c:198-217 decodes in front of classify_signal, converted in the kernel's load)
static int classify_signal_batch(dsp_mfcc_plan *plan, dsp_stop_model *m, const void *d_signal, int in_kind, int channels, int stereo_mode, long n_clips,
                                 int samples_per_clip, long clip_stride, float *d_prob, void *stream)
{
    if (!plan || !m || n_clips < 0 || (n_clips > 0 && (!d_signal || !d_prob))) return capi_fail(DSP_EINVAL, "bad argument");
    dsp_mfcc_config cfg;
    dsp_mfcc_plan_config(plan, &cfg);
    if (cfg.n_mfcc != m->m.n_coef) return capi_fail(DSP_EINVAL, "plan n_mfcc differs from the model's n_coef");
    // (the fused kernel is the default path: it must refuse what the two-kernel path refuses)
    if (n_clips > 1 && clip_stride < samples_per_clip) return capi_fail(DSP_EINVAL, "clip_stride < samples_per_clip");
    if (dsp::plan_device(plan) != m->device) return capi_fail(DSP_EINVAL, "plan and stop model live on different devices");
    if (n_clips == 0) return DSP_OK;
    const int t = dsp_mfcc_frames_for(&cfg, samples_per_clip, m->m.max_frames);          // stop_detector.c:18-21
    {   // one kernel from PCM to probability when the plan is the reference's shape: the MFCC matrix is never written
        const int fused = dsp::stop_fused_device(plan, m->m, d_signal, n_clips, samples_per_clip, clip_stride, t, d_prob, stream, in_kind);
        if (fused != 0) return fused < 0 ? fused : DSP_OK;
    }
    std::lock_guard<std::mutex> lock(m->mu);
    DSP_ON_DEVICE(m->device);
    const size_t need = (size_t)n_clips * (t > 0 ? t : 1) * cfg.n_mfcc * sizeof(float);
    if (m->mfcc_cap < need) {
        if (m->d_mfcc) { hipFree(m->d_mfcc); m->d_mfcc = nullptr; m->mfcc_cap = 0; }
        DSP_CAPI_HIP(hipMalloc(&m->d_mfcc, need));
        m->mfcc_cap = need;
    }
    if (t > 0) {
        const int rc = in_kind == 0 ? dsp_mfcc_clips_device(plan, static_cast<const float *>(d_signal), n_clips, samples_per_clip, clip_stride, m->d_mfcc, m->m.max_frames, stream)
                                    : dsp_mfcc_clips_pcm16_device(plan, static_cast<const int16_t *>(d_signal), n_clips, samples_per_clip, clip_stride, channels,
                                                                  stereo_mode, m->d_mfcc, m->m.max_frames, stream);
        if (rc < 0) return rc;
    }
    DSP_CAPI_HIP(dsp::launch_stop_tail(m->m, m->d_mfcc, n_clips, t, d_prob, (hipStream_t)stream));
    return DSP_OK;
}

extern "C" {

int dsp_classify_signal_batch_device(dsp_mfcc_plan *plan, dsp_stop_model *m, const float *d_signal, long n_clips,
                                     int samples_per_clip, long clip_stride, float *d_prob, void *stream)
{
    return classify_signal_batch(plan, m, d_signal, 0, 1, 0, n_clips, samples_per_clip, clip_stride, d_prob, stream);
}

int dsp_classify_signal_batch_pcm16_device(dsp_mfcc_plan *plan, dsp_stop_model *m, const int16_t *d_pcm, long n_clips, int samples_per_clip,
                                           long clip_stride, int channels, int stereo_mode, float *d_prob, void *stream)
{
    if (channels != 1 && channels != 2) return capi_fail(DSP_EINVAL, "channels must be 1 or 2");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return capi_fail(DSP_EINVAL, "bad stereo_mode");
    const int kind = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
    return classify_signal_batch(plan, m, d_pcm, kind, channels, stereo_mode, n_clips, samples_per_clip, clip_stride, d_prob, stream);
}

// Ragged batches (main_test.c:254-331 loops over files of different lengths): one launch of the fused kernel, every clip with the
// frames its own length gives (capped at the model's max_frames).  The fused kernel only: plans outside its shape are refused.
static int classify_signal_batch_ragged(dsp_mfcc_plan *plan, dsp_stop_model *m, const void *d_signal, int in_kind, long n_clips, const long *offsets,
                                        float *d_prob, void *stream)
{
    if (!plan || !m || n_clips < 0 || !offsets || (n_clips > 0 && (!d_signal || !d_prob))) return capi_fail(DSP_EINVAL, "bad argument");
    dsp_mfcc_config cfg;
    dsp_mfcc_plan_config(plan, &cfg);
    if (cfg.n_mfcc != m->m.n_coef) return capi_fail(DSP_EINVAL, "plan n_mfcc differs from the model's n_coef");
    if (dsp::plan_device(plan) != m->device) return capi_fail(DSP_EINVAL, "plan and stop model live on different devices");
    if (n_clips == 0) return DSP_OK;
    const int fused = dsp::stop_fused_device(plan, m->m, d_signal, n_clips, 0, 0, 1, d_prob, stream, in_kind, offsets);
    if (fused == 0) return capi_fail(DSP_EINVAL, "ragged batches run on the fused clip -> probability kernel: the reference's MFCC shape (dsp_mfcc_default_config), "
                                                 "a model with at most 4 first-layer units, an 8-byte aligned buffer (4 for mono int16)");
    return fused < 0 ? fused : DSP_OK;
}

int dsp_classify_signal_batch_ragged_device(dsp_mfcc_plan *plan, dsp_stop_model *m, const float *d_signal, long n_clips, const long *offsets,
                                            float *d_prob, void *stream)
{
    return classify_signal_batch_ragged(plan, m, d_signal, 0, n_clips, offsets, d_prob, stream);
}

int dsp_classify_signal_batch_ragged_pcm16_device(dsp_mfcc_plan *plan, dsp_stop_model *m, const int16_t *d_pcm, long n_clips, const long *offsets,
                                                  int channels, int stereo_mode, float *d_prob, void *stream)
{
    if (channels != 1 && channels != 2) return capi_fail(DSP_EINVAL, "channels must be 1 or 2");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return capi_fail(DSP_EINVAL, "bad stereo_mode");
    const int kind = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
    return classify_signal_batch_ragged(plan, m, d_pcm, kind, n_clips, offsets, d_prob, stream);
}

float dsp_classify_signal(dsp_stop_model *m, const float *signal, int num_samples)
{
    if (!m || !signal || num_samples < 0) { capi_fail(DSP_EINVAL, "bad argument"); return 0.0f; }
    auto bail = [](const char *what) { std::fprintf(stderr, "libdsp_amd: classify_signal: %s: %s\n", what, dsp_last_error()); return 0.0f; };
    {
        std::lock_guard<std::mutex> lock(m->mu);
        dsp::DeviceScope dsp_device_scope_(m->device);
        if (dsp_device_scope_.err != hipSuccess) { capi_fail(DSP_EHIP, "hipSetDevice"); return bail("device"); }
        if (!m->plan) {
            dsp_mfcc_config cfg;
            dsp_mfcc_default_config(&cfg);
            cfg.n_mfcc = m->m.n_coef;
            if (dsp_mfcc_plan_create(&cfg, m->device, &m->plan) < 0) return bail("plan");
        }
        const size_t need = ((size_t)num_samples + 2) * sizeof(float);
        if (m->sig_cap < need) {
            if (m->d_sig) { hipFree(m->d_sig); m->d_sig = nullptr; m->sig_cap = 0; }
            if (hipMalloc(&m->d_sig, need) != hipSuccess) { capi_fail(DSP_ENOMEM, "hipMalloc"); return bail("workspace"); }
            m->sig_cap = need;
        }
        if (!m->d_prob && hipMalloc(&m->d_prob, sizeof(float)) != hipSuccess) { capi_fail(DSP_ENOMEM, "hipMalloc"); return bail("workspace"); }
        if (num_samples > 0 && hipMemcpy(m->d_sig, signal, (size_t)num_samples * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            capi_fail(DSP_EHIP, "hipMemcpy");
            return bail("copy in");
        }
    }
    if (dsp_classify_signal_batch_device(m->plan, m, m->d_sig, 1, num_samples, num_samples, m->d_prob, nullptr) < 0) return bail("run");
    float p = 0.0f;
    if (hipMemcpy(&p, m->d_prob, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { capi_fail(DSP_EHIP, "hipMemcpy"); return bail("copy out"); }
    return p;
}

int dsp_speaker_model_create(const dsp_gmm_params *target, const dsp_gmm_params *ubm, int device, dsp_speaker_model **out)
{
    if (!out) return capi_fail(DSP_EINVAL, "out is NULL");
    *out = nullptr;
    for (const dsp_gmm_params *g : {target, ubm})
        if (!g || g->k <= 0 || g->k > 64 || g->d <= 0 || g->d > 16 || !g->means || !g->inv_covs || !g->log_consts)
            return capi_fail(DSP_EINVAL, "bad GMM parameters (k <= 64, d <= 16)");
    if (target->k != ubm->k || target->d != ubm->d) return capi_fail(DSP_EINVAL, "target and UBM must have the same shape");
    int rc = check_device(device);
    if (rc < 0) return rc;
    DSP_ON_DEVICE(device);
    const size_t kd = (size_t)target->k * target->d, k = target->k;
    // layout: int32 inv_covs (t, u), int16 log_consts (t, u), int8 means (t, u)
    const size_t bytes = 2 * kd * 4 + 2 * k * 2 + 2 * kd;
    std::vector<char> host(bytes);
    auto *m = new dsp_speaker_model;
    m->device = device;
    if (hipMalloc(&m->d_blob, bytes) != hipSuccess) { delete m; return capi_fail(DSP_ENOMEM, "hipMalloc"); }
    char *d = static_cast<char *>(m->d_blob);
    size_t off = 0;
    auto put = [&](const void *src, size_t n) { std::memcpy(host.data() + off, src, n); const void *dev = d + off; off += n; return dev; };
    m->target.k = m->ubm.k = target->k;
    m->target.d = m->ubm.d = target->d;
    m->target.inv_covs = static_cast<const int32_t *>(put(target->inv_covs, kd * 4));
    m->ubm.inv_covs = static_cast<const int32_t *>(put(ubm->inv_covs, kd * 4));
    m->target.log_consts = static_cast<const int16_t *>(put(target->log_consts, k * 2));
    m->ubm.log_consts = static_cast<const int16_t *>(put(ubm->log_consts, k * 2));
    m->target.means = static_cast<const int8_t *>(put(target->means, kd));
    m->ubm.means = static_cast<const int8_t *>(put(ubm->means, kd));
    if (hipMemcpy(m->d_blob, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(m->d_blob); delete m; return capi_fail(DSP_EHIP, "hipMemcpy"); }
    *out = m;
    return DSP_OK;
}

void dsp_speaker_model_destroy(dsp_speaker_model *m)
{
    if (!m) return;
    dsp::DeviceScope dsp_device_scope_(m->device);
    if (m->d_blob) hipFree(m->d_blob);
    delete m;
}

int dsp_speaker_llr_device(dsp_speaker_model *m, const float *d_mfcc, long n_clips, int frames_per_clip, int64_t *d_llr_mean,
                           int *d_labels, int64_t *d_ll_target, int64_t *d_ll_ubm, void *stream)
{
    if (!m || n_clips < 0 || frames_per_clip <= 0 || (n_clips > 0 && (!d_mfcc || !d_llr_mean))) return capi_fail(DSP_EINVAL, "bad argument");
    const long long threshold = (long long)(-0.7 * (1 << 8));                              // speaker_gmm.c:124-125
    DSP_CAPI_HIP(dsp::launch_speaker_llr(m->target, m->ubm, d_mfcc, n_clips, frames_per_clip, threshold,
                                         reinterpret_cast<long long *>(d_llr_mean), d_labels, reinterpret_cast<long long *>(d_ll_target),
                                         reinterpret_cast<long long *>(d_ll_ubm), (hipStream_t)stream));
    return DSP_OK;
}

int dsp_upsample_linear_device(const float *d_in, long n_clips, int old_size, long in_stride, float *d_out, int new_size,
                               long out_stride, void *stream)
{
    if (n_clips < 0 || old_size < 1 || new_size < 2 || (n_clips > 0 && (!d_in || !d_out)) || (n_clips > 1 && (in_stride < old_size || out_stride < new_size)))
        return capi_fail(DSP_EINVAL, "bad argument (old_size >= 1, new_size >= 2)");
    for (long c0 = 0; c0 < n_clips; c0 += 65535) {
        const long cnt = n_clips - c0 < 65535 ? n_clips - c0 : 65535;
        DSP_CAPI_HIP(dsp::launch_upsample_linear(d_in + c0 * in_stride, cnt, old_size, in_stride, d_out + c0 * out_stride, new_size, out_stride,
                                                 (hipStream_t)stream));
    }
    return DSP_OK;
}

int dsp_upsample_linear_host(const float *in, int old_size, float *out, int new_size)
{
    if (!in || !out || old_size < 1 || new_size < 2) return capi_fail(DSP_EINVAL, "bad argument (old_size >= 1, new_size >= 2)");
    int rc = check_device(0);
    if (rc < 0) return rc;
    float *d_in = nullptr, *d_out = nullptr;
    DSP_CAPI_HIP(hipMalloc(&d_in, (size_t)old_size * 4));
    if (hipMalloc(&d_out, (size_t)new_size * 4) != hipSuccess) { hipFree(d_in); return capi_fail(DSP_ENOMEM, "hipMalloc"); }
    hipError_t e = hipMemcpy(d_in, in, (size_t)old_size * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_upsample_linear(d_in, 1, old_size, old_size, d_out, new_size, new_size, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)new_size * 4, hipMemcpyDeviceToHost);
    hipFree(d_in);
    hipFree(d_out);
    if (e != hipSuccess) return capi_fail(DSP_EHIP, hipGetErrorString(e));
    return DSP_OK;
}

int dsp_fft_real_forward_host(const float *in_time, long n_frames, int frame_length, long in_stride, int n_fft, float *out_freq)
{
    if (!in_time || !out_freq || n_frames < 0 || (n_frames > 1 && in_stride < frame_length)) return capi_fail(DSP_EINVAL, "bad argument");
    if (n_fft < 2 || (n_fft & (n_fft - 1)) || n_fft > 4096 || frame_length < 1 || frame_length > n_fft) return capi_fail(DSP_EINVAL, "n_fft: a power of two <= 4096, 1 <= frame_length <= n_fft");
    if (n_frames == 0) return DSP_OK;
    int rc = check_device(0);
    if (rc < 0) return rc;
    float *d_in = nullptr, *d_out = nullptr;
    DSP_CAPI_HIP(hipMalloc(&d_in, (size_t)n_frames * frame_length * 4));
    if (hipMalloc(&d_out, (size_t)n_frames * n_fft * 8) != hipSuccess) { hipFree(d_in); return capi_fail(DSP_ENOMEM, "hipMalloc"); }
    hipError_t e = hipMemcpy2D(d_in, (size_t)frame_length * 4, in_time, (size_t)(n_frames > 1 ? in_stride : frame_length) * 4, (size_t)frame_length * 4, (size_t)n_frames, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_fft_real_forward(d_in, n_frames, frame_length, frame_length, n_fft, d_out, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_freq, d_out, (size_t)n_frames * n_fft * 8, hipMemcpyDeviceToHost);
    hipFree(d_in);
    hipFree(d_out);
    if (e != hipSuccess) return capi_fail(DSP_EHIP, hipGetErrorString(e));
    return DSP_OK;
}

/* 2fa/audio/word/c/mfcc.c:16 (non-static there; SURVEY 8b lists it as an optional same-layer symbol): FRAME_LENGTH = 400 samples in,
 * MFCC_N_FFT = 512 complex bins out.  Same contract: void; a failure leaves the reason in dsp_last_error() and zeros in out_freq. */
void fft_real_forward(const float *in_time, float *out_freq)
{
    if (out_freq && dsp_fft_real_forward_host(in_time, 1, 400, 400, 512, out_freq) < 0) {
        std::fprintf(stderr, "libdsp_amd: fft_real_forward: %s\n", dsp_last_error());
        for (int i = 0; i < 1024; ++i) out_freq[i] = 0.0f;
    }
}

}  // extern "C"
