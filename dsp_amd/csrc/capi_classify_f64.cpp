// capi_classify_f64.cpp -- C ABI of the float64 classifier (include/dsp_amd.h: dsp_classify_batch_*_f64), the whole
// per-clip chain of donut-classifier/classifier.c:83-192 on the GPU in double:
//     butter_bandpass_filter (3000-7500 Hz and, inside find_midpoints, 1000-3000 Hz; :420-446)  iir_kernel<double>
//     compute_spectrogram of both (:448-592)                                                     spectrogram_f64_kernel
//     dB maps, 45 dB midpoints, normalisation, keep band, band sums, rule (:105-190, :594-830)   classify_f64_tail_kernel
// Correctness first: sub-batches through a scratch workspace that lives for the call; not tuned.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "capi_util.hpp"
#include "classify_kernels.hpp"

static_assert(sizeof(dsp::ClassifyTraceD) == sizeof(dsp_classify_trace_f64), "trace layouts must match");

namespace {

constexpr long kSubBatch = 2048;           // clips per pass: ~0.9 GB of float64 scratch for 1 s clips
constexpr int kMaxColumns = 957;           // as the float32 path (capi.cpp kMaxSpecColumns): at most 64 midpoints fit such a clip

int columns(int n) { return n < dsp::kSpecSeg ? 0 : (n - dsp::kSpecSeg) / dsp::kSpecHop + 1; }

struct Scratch {
    double *x = nullptr, *y_bp = nullptr, *y_mp = nullptr, *s_bp = nullptr, *s_mp = nullptr;
    int *labels = nullptr;
    dsp::ClassifyTraceD *trace = nullptr;
    ~Scratch()
    {
        for (void *p : {(void *)x, (void *)y_bp, (void *)y_mp, (void *)s_bp, (void *)s_mp, (void *)labels, (void *)trace})
            if (p) (void)hipFree(p);
    }
};

bool valid(const dsp_classify_config_f64 &c)
{
    auto fin = [](double v) { return v == v && v - v == 0.0; };
    return fin(c.keep_lo) && fin(c.keep_hi) && fin(c.midpoint_db) && fin(c.middle_max) && fin(c.above_min) && fin(c.below_min) && c.keep_lo < c.keep_hi;
}

// one sub-batch resident at d_x (row stride `stride`): labels (+ trace) into the scratch arrays
int run(const dsp_classify_config_f64 &cfg, Scratch &w, const double *d_x, long cnt, int n, long stride, bool want_trace, hipStream_t st)
{
    double b[9], a[9];
    dsp::IirCoefD c_bp, c_mp;
    dsp_butter_bandpass(3000.0, 7500.0, b, a);                       // classifier.c:86-91
    for (int i = 0; i < 9; ++i) { c_bp.b[i] = b[i]; c_bp.a[i] = a[i]; }
    dsp_butter_bandpass(1000.0, 3000.0, b, a);                       // :659-664
    for (int i = 0; i < 9; ++i) { c_mp.b[i] = b[i]; c_mp.a[i] = a[i]; }
    DSP_CAPI_HIP(dsp::launch_iir_f64(d_x, cnt, n, stride, c_bp, w.y_bp, st));     // launch_iir_f64 writes y with the input's stride
    DSP_CAPI_HIP(dsp::launch_iir_f64(d_x, cnt, n, stride, c_mp, w.y_mp, st));
    DSP_CAPI_HIP(dsp::launch_spectrogram_f64(w.y_bp, cnt, n, stride, 16000, w.s_bp, st));
    DSP_CAPI_HIP(dsp::launch_spectrogram_f64(w.y_mp, cnt, n, stride, 16000, w.s_mp, st));
    const dsp::ClassifyRuleD rule{cfg.keep_lo, cfg.keep_hi, cfg.midpoint_db, cfg.middle_max, cfg.above_min, cfg.below_min};
    DSP_CAPI_HIP(dsp::launch_classify_f64_tail(w.s_bp, w.s_mp, cnt, n, 16000, rule, w.labels, want_trace ? w.trace : nullptr, st));
    return DSP_OK;
}

int reserve(Scratch &w, long clips, int n, long stride, bool need_x)
{
    const size_t T = (size_t)columns(n), row = (size_t)stride;
    if (need_x) DSP_CAPI_HIP(hipMalloc(&w.x, (size_t)clips * row * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.y_bp, (size_t)clips * row * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.y_mp, (size_t)clips * row * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.s_bp, (size_t)clips * dsp::kSpecBins * T * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.s_mp, (size_t)clips * dsp::kSpecBins * T * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.labels, (size_t)clips * sizeof(int)));
    DSP_CAPI_HIP(hipMalloc(&w.trace, (size_t)clips * sizeof(dsp::ClassifyTraceD)));
    return DSP_OK;
}

int check_args(const dsp_classify_config_f64 *cfgp, const void *signal, long n_clips, int n, long stride, const int *labels, dsp_classify_config_f64 &cfg)
{
    if (!signal || !labels || n_clips < 0 || n < 0 || (n_clips > 1 && stride < n)) return dsp::capi_fail(DSP_EINVAL, "bad argument");
    if (cfgp) cfg = *cfgp; else dsp_classify_default_config_f64(&cfg);
    if (!valid(cfg)) return dsp::capi_fail(DSP_EINVAL, "classify config: thresholds must be finite with keep_lo < keep_hi");
    if (columns(n) > kMaxColumns) return dsp::capi_fail(DSP_EINVAL, "clip too long (more than 957 spectrogram columns = 13.4 s at 16 kHz)");
    return DSP_OK;
}

}  // namespace

extern "C" {

void dsp_classify_default_config_f64(dsp_classify_config_f64 *c)
{
    // donut-classifier/classifier.c:141-142 (0.70 / 0.85), :660 (45 dB), :184 (75 / 300 / 100)
    *c = dsp_classify_config_f64{0.70, 0.85, 45.0, 75.0, 300.0, 100.0};
}

int dsp_classify_batch_device_f64(const dsp_classify_config_f64 *cfgp, const double *d_signal, long n_clips, int n, long stride,
                                  int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream)
{
    dsp_classify_config_f64 cfg;
    int rc = check_args(cfgp, d_signal, n_clips, n, stride, d_labels, cfg);
    if (rc < 0) return rc;
    if (n_clips == 0) return DSP_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_signal) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return dsp::capi_fail(DSP_EINVAL, "signal is not a device pointer");
    }
    DSP_ON_DEVICE(attr.device);
    hipStream_t st = (hipStream_t)stream;
    if (columns(n) == 0) {                   // shorter than one spectrogram segment: no midpoints, label 0
        DSP_CAPI_HIP(hipMemsetAsync(d_labels, 0, (size_t)n_clips * sizeof(int), st));
        if (d_trace) DSP_CAPI_HIP(hipMemsetAsync(d_trace, 0, (size_t)n_clips * sizeof(dsp_classify_trace_f64), st));
        DSP_CAPI_HIP(hipStreamSynchronize(st));
        return DSP_OK;
    }
    if (n_clips == 1) stride = n;
    Scratch w;
    if ((rc = reserve(w, std::min(kSubBatch, n_clips), n, stride, false)) < 0) return rc;
    for (long c0 = 0; c0 < n_clips; c0 += kSubBatch) {
        const long cnt = std::min(kSubBatch, n_clips - c0);
        if ((rc = run(cfg, w, d_signal + c0 * stride, cnt, n, stride, d_trace != nullptr, st)) < 0) return rc;
        DSP_CAPI_HIP(hipMemcpyAsync(d_labels + c0, w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToDevice, st));
        if (d_trace) DSP_CAPI_HIP(hipMemcpyAsync(d_trace + c0, w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToDevice, st));
    }
    DSP_CAPI_HIP(hipStreamSynchronize(st));   // the scratch dies with this call
    return DSP_OK;
}

int dsp_classify_batch_host_f64(const dsp_classify_config_f64 *cfgp, const double *signal, long n_clips, int n, long stride,
                                int *labels, dsp_classify_trace_f64 *trace)
{
    dsp_classify_config_f64 cfg;
    int rc = check_args(cfgp, signal, n_clips, n, stride, labels, cfg);
    if (rc < 0) return rc;
    if (n_clips == 0) return DSP_OK;
    if (columns(n) == 0) {
        for (long c = 0; c < n_clips; ++c) labels[c] = 0;
        if (trace) for (long c = 0; c < n_clips; ++c) trace[c] = dsp_classify_trace_f64{};
        return DSP_OK;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return dsp::capi_fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback");
    const char *dev = std::getenv("DSP_AMD_DEVICE");
    const int device = dev ? std::atoi(dev) : 0;
    if (device < 0 || device >= count) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
    DSP_ON_DEVICE(device);
    if (n_clips == 1) stride = n;
    Scratch w;
    if ((rc = reserve(w, std::min(kSubBatch, n_clips), n, n, true)) < 0) return rc;
    for (long c0 = 0; c0 < n_clips; c0 += kSubBatch) {
        const long cnt = std::min(kSubBatch, n_clips - c0);
        DSP_CAPI_HIP(hipMemcpy2D(w.x, (size_t)n * sizeof(double), signal + c0 * stride, (size_t)stride * sizeof(double), (size_t)n * sizeof(double),
                                 (size_t)cnt, hipMemcpyHostToDevice));
        if ((rc = run(cfg, w, w.x, cnt, n, n, trace != nullptr, nullptr)) < 0) return rc;
        DSP_CAPI_HIP(hipMemcpy(labels + c0, w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost));
        if (trace) DSP_CAPI_HIP(hipMemcpy(trace + c0, w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToHost));
    }
    return DSP_OK;
}

}  // extern "C"
