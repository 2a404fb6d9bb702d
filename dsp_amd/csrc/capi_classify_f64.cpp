// capi_classify_f64.cpp -- C ABI of the float64 classifier (include/dsp_amd.h: dsp_classify_batch_*_f64), the whole
// per-clip chain of donut-classifier/classifier.c:83-192 on the GPU in double:
//     butter_bandpass_filter (3000-7500 Hz and, inside find_midpoints, 1000-3000 Hz; :420-446)  iir_kernel<double>, both filters in one launch
//     compute_spectrogram of the 1000-3000 Hz output (:448-592) -> loud time bins (:679-745)     spectrogram_f64_fft_kernel<flags>
//     clusters -> midpoints (:747-800), work list of the clips that have any                     classify_f64_midpoints_kernel
//     compute_spectrogram of the 3000-7500 Hz output, listed clips only                          spectrogram_f64_fft_kernel<maps>
//     dB map, normalisation, keep band, band sums, rule (:105-190, :594-653)                     classify_f64_bands_kernel
// Sub-batches of up to 65 536 clips (one recurrence wavefront per SIMD and filter) through a grow-only scratch workspace that
// stays with the library, like the float32 classifier's (capi.cpp, ClassifyCtx).  DSP_AMD_F64_DFT=1 routes the batch through the
// direct-DFT spectrogram ([129][T] maps) instead of the FFT kernel: the yardstick of tests/test_gpu_classify_f64.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>

#include "capi_util.hpp"
#include "classify_kernels.hpp"

static_assert(sizeof(dsp::ClassifyTraceD) == sizeof(dsp_classify_trace_f64), "trace layouts must match");

namespace {

constexpr long kSubBatchDefault = 65536;   // clips per pass: 330 KB of float64 scratch per 1 s clip (22 GB at the full sub-batch)
constexpr int kMaxColumns = 957;           // as the float32 path (capi.cpp kMaxSpecColumns): at most 64 midpoints fit such a clip

int columns(int n) { return n < dsp::kSpecSeg ? 0 : (n - dsp::kSpecSeg) / dsp::kSpecHop + 1; }

struct Scratch {
    int device = -1;
    dsp::SpecTablesD *tab = nullptr;
    double U = 0.0;                            // SpecTablesD::U of tab
    double *x = nullptr, *y_bp = nullptr, *y_mp = nullptr, *s_bp = nullptr, *s_mp = nullptr, *mids = nullptr;      // s_mp: DSP_AMD_F64_DFT only
    int *labels = nullptr, *loud = nullptr, *n_mids = nullptr, *hits = nullptr;
    dsp::ClassifyTraceD *trace = nullptr;
    long cap_clips = 0, cap_row = 0;           // what the workspace holds: clips x row doubles (x only when cap_x)
    int cap_T = 0;
    bool cap_x = false;
    std::mutex mu;
    void drop()
    {
        for (void *p : {(void *)x, (void *)y_bp, (void *)y_mp, (void *)s_bp, (void *)s_mp, (void *)mids, (void *)labels, (void *)loud, (void *)n_mids,
                        (void *)hits, (void *)trace})
            if (p) (void)hipFree(p);
        x = y_bp = y_mp = s_bp = s_mp = mids = nullptr; labels = loud = n_mids = hits = nullptr; trace = nullptr;
        cap_clips = cap_row = 0; cap_T = 0; cap_x = false;
    }
};
Scratch g_w;

long sub_batch()                            // DSP_AMD_F64_SUB_BATCH: a smaller pass (tests: a batch that spans passes)
{
    const char *e = std::getenv("DSP_AMD_F64_SUB_BATCH");
    const long v = e ? std::atol(e) : 0;
    return v >= 64 ? std::min(v, kSubBatchDefault) : kSubBatchDefault;
}
long row_of(int n) { return ((long)n + 1) & ~1L; }      // workspace row: n doubles rounded up to 16 bytes
bool use_dft() { const char *e = std::getenv("DSP_AMD_F64_DFT"); return e && std::atoi(e) != 0; }

bool valid(const dsp_classify_config_f64 &c)
{
    auto fin = [](double v) { return v == v && v - v == 0.0; };
    return fin(c.keep_lo) && fin(c.keep_hi) && fin(c.midpoint_db) && fin(c.middle_max) && fin(c.above_min) && fin(c.below_min) && c.keep_lo < c.keep_hi;
}

// one sub-batch resident at d_x (row stride `stride`): labels (+ trace) into the scratch arrays
int run(const dsp_classify_config_f64 &cfg, Scratch &w, const double *d_x, long cnt, int n, long stride, bool want_trace, hipStream_t st)
{
    const long row = row_of(n);
    double b[9], a[9];
    dsp::IirCoefD c_bp, c_mp;
    dsp_butter_bandpass(3000.0, 7500.0, b, a);                       // classifier.c:86-91
    for (int i = 0; i < 9; ++i) { c_bp.b[i] = b[i]; c_bp.a[i] = a[i]; }
    dsp_butter_bandpass(1000.0, 3000.0, b, a);                       // :659-664
    for (int i = 0; i < 9; ++i) { c_mp.b[i] = b[i]; c_mp.a[i] = a[i]; }
    DSP_CAPI_HIP(dsp::launch_iir2_f64(d_x, cnt, n, stride, row, c_bp, w.y_bp, c_mp, w.y_mp, st));
    const dsp::ClassifyRuleD rule{cfg.keep_lo, cfg.keep_hi, cfg.midpoint_db, cfg.middle_max, cfg.above_min, cfg.below_min};
    dsp::ClassifyTraceD *tr = want_trace ? w.trace : nullptr;
    if (use_dft()) {                          // the yardstick: direct DFT, [129][T] maps of both outputs, one tail kernel per clip
        if (!w.s_mp) DSP_CAPI_HIP(hipMalloc(&w.s_mp, (size_t)w.cap_clips * dsp::kSpecBins * (size_t)w.cap_T * sizeof(double)));
        DSP_CAPI_HIP(dsp::launch_spectrogram_f64(w.y_bp, cnt, n, row, 16000, w.s_bp, st));
        DSP_CAPI_HIP(dsp::launch_spectrogram_f64(w.y_mp, cnt, n, row, 16000, w.s_mp, st));
        DSP_CAPI_HIP(dsp::launch_classify_f64_tail(w.s_bp, w.s_mp, cnt, n, 16000, rule, w.labels, tr, st));
        return DSP_OK;
    }
    DSP_CAPI_HIP(dsp::launch_spectrogram_f64_flags(w.y_mp, cnt, n, row, w.tab, cfg.midpoint_db, w.loud, st));
    DSP_CAPI_HIP(dsp::launch_classify_f64_midpoints(w.loud, cnt, n, 16000, w.mids, w.n_mids, w.hits, w.labels, tr, st));
    DSP_CAPI_HIP(dsp::launch_spectrogram_f64_listed(w.y_bp, cnt, n, row, w.tab, w.hits, w.s_bp, st));
    DSP_CAPI_HIP(dsp::launch_classify_f64_bands(w.s_bp, w.hits, cnt, n, 16000, w.U, rule, w.mids, w.n_mids, w.labels, tr, st));
    return DSP_OK;
}

// the workspace on `device` for sub-batches of `clips` clips of n samples; grows, never shrinks; moves with the device
int reserve(Scratch &w, int device, long clips, int n, bool need_x)
{
    const size_t T = (size_t)columns(n), row = (size_t)row_of(n);
    if (w.device != device) {
        if (w.device >= 0) {
            dsp::DeviceScope on_old(w.device);
            (void)hipDeviceSynchronize();
            w.drop();
            if (w.tab) (void)hipFree(w.tab);
            w.tab = nullptr;
        }
        w.device = device;
    }
    if (!w.tab) {
        dsp::SpecTablesD t;
        dsp::build_spec_tables_f64(16000, t);
        DSP_CAPI_HIP(hipMalloc(&w.tab, sizeof(t)));
        DSP_CAPI_HIP(hipMemcpy(w.tab, &t, sizeof(t), hipMemcpyHostToDevice));
        w.U = t.U;
    }
    if (clips <= w.cap_clips && (long)row <= w.cap_row && (int)T <= w.cap_T && (!need_x || w.cap_x)) return DSP_OK;
    (void)hipDeviceSynchronize();
    need_x = need_x || w.cap_x;
    clips = std::max(clips, w.cap_clips);
    w.drop();
    if (need_x) DSP_CAPI_HIP(hipMalloc(&w.x, (size_t)clips * row * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.y_bp, (size_t)clips * row * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.y_mp, (size_t)clips * row * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.s_bp, (size_t)clips * dsp::kSpecBins * T * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.mids, (size_t)clips * dsp::kMaxMidpoints * sizeof(double)));
    DSP_CAPI_HIP(hipMalloc(&w.loud, (size_t)clips * T * sizeof(int)));
    DSP_CAPI_HIP(hipMalloc(&w.n_mids, (size_t)clips * sizeof(int)));
    DSP_CAPI_HIP(hipMalloc(&w.hits, ((size_t)clips + 1) * sizeof(int)));
    DSP_CAPI_HIP(hipMalloc(&w.labels, (size_t)clips * sizeof(int)));
    DSP_CAPI_HIP(hipMalloc(&w.trace, (size_t)clips * sizeof(dsp::ClassifyTraceD)));
    w.cap_clips = clips; w.cap_row = (long)row; w.cap_T = (int)T; w.cap_x = need_x;
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
    std::lock_guard<std::mutex> lock(g_w.mu);
    Scratch &w = g_w;
    const long kSubBatch = sub_batch();
    if ((rc = reserve(w, attr.device, std::min(kSubBatch, n_clips), n, false)) < 0) return rc;
    for (long c0 = 0; c0 < n_clips; c0 += kSubBatch) {
        const long cnt = std::min(kSubBatch, n_clips - c0);
        if ((rc = run(cfg, w, d_signal + c0 * stride, cnt, n, stride, d_trace != nullptr, st)) < 0) return rc;
        DSP_CAPI_HIP(hipMemcpyAsync(d_labels + c0, w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToDevice, st));
        if (d_trace) DSP_CAPI_HIP(hipMemcpyAsync(d_trace + c0, w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToDevice, st));
    }
    DSP_CAPI_HIP(hipStreamSynchronize(st));   // the workspace is free for the next call
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
    std::lock_guard<std::mutex> lock(g_w.mu);
    Scratch &w = g_w;
    const long kSubBatch = sub_batch();
    if ((rc = reserve(w, device, std::min(kSubBatch, n_clips), n, true)) < 0) return rc;
    const long row = row_of(n);
    for (long c0 = 0; c0 < n_clips; c0 += kSubBatch) {
        const long cnt = std::min(kSubBatch, n_clips - c0);
        DSP_CAPI_HIP(hipMemcpy2D(w.x, (size_t)row * sizeof(double), signal + c0 * stride, (size_t)stride * sizeof(double), (size_t)n * sizeof(double),
                                 (size_t)cnt, hipMemcpyHostToDevice));
        if ((rc = run(cfg, w, w.x, cnt, n, row, trace != nullptr, nullptr)) < 0) return rc;
        DSP_CAPI_HIP(hipMemcpy(labels + c0, w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost));
        if (trace) DSP_CAPI_HIP(hipMemcpy(trace + c0, w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToHost));
    }
    return DSP_OK;
}

}  // extern "C"
