// capi_classify_f64.cpp -- C ABI of the float64 classifier (include/dsp_amd.h: dsp_classify_batch_*_f64), the whole
// per-clip chain of donut-classifier/classifier.c:83-192 on the GPU in double:
//     butter_bandpass_filter (3000-7500 Hz and, inside find_midpoints, 1000-3000 Hz; :420-446)    iir2_screen_f64_kernel: both recurrences in one
//     compute_spectrogram of the 1000-3000 Hz output (:448-592) -> loud time bins (:679-745)       pass, restart states instead of filtered signals,
//                                                                                                  the loud bins by a bounded screening transform;
//                                                                                                  spec_f64_from_ckpt_kernel<flags> for the undecided
//     clusters -> midpoints (:747-800), work list of the clips that have any                       classify_f64_midpoints_kernel
//     compute_spectrogram of the 3000-7500 Hz output, listed clips only                            spec_f64_from_ckpt_kernel<maps>
//     dB map, normalisation, keep band, band sums, rule (:105-190, :594-653)                       classify_f64_bands_kernel
// Input: float64 samples or int16 PCM (mono / interleaved stereo, converted in the kernels' loads exactly like classifier.c:55-59, :286-297).
// One workspace PER DEVICE (grow-only until dsp_classify_release_f64), its own mutex; a call enqueues on the caller's stream and
// returns: the next call on that device first makes its stream wait for the event the previous one left behind.
// Two yardstick pipelines for the tests (float64 input only; they materialise both filtered signals like round 3):
//     DSP_AMD_F64_PIPELINE=materialize   iir_kernel<double> x 2 -> spectrogram_f64_fft_kernel<flags / maps> (the same fft_frame)
//     DSP_AMD_F64_DFT=1                  ... -> the direct 256-point DFT ([129][T] maps) and the one-kernel tail
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "capi_util.hpp"
#include "classify_kernels.hpp"

static_assert(sizeof(dsp::ClassifyTraceD) == sizeof(dsp_classify_trace_f64), "trace layouts must match");

namespace {

constexpr int kMaxDevices = 64;
constexpr int kMaxColumns = 957;           // as the float32 path (capi.cpp kMaxSpecColumns): at most 64 midpoints fit such a clip

int columns(int n) { return n < dsp::kSpecSeg ? 0 : (n - dsp::kSpecSeg) / dsp::kSpecHop + 1; }
long row_of(int n) { return ((long)n + 1) & ~1L; }      // staging row: n doubles rounded up to 16 bytes

enum Pipeline { kCkpt = 0, kMaterialize = 1, kDft = 2 };
Pipeline pipeline()
{
    const char *d = std::getenv("DSP_AMD_F64_DFT");
    if (d && std::atoi(d) != 0) return kDft;
    const char *p = std::getenv("DSP_AMD_F64_PIPELINE");
    return p && std::strcmp(p, "materialize") == 0 ? kMaterialize : kCkpt;
}

struct Scratch {
    std::mutex mu;
    int device = -1;
    dsp::SpecTablesD *tab = nullptr;
    dsp::ScreenTablesD *scr = nullptr;
    unsigned long long *minmax = nullptr;  // [clip][2]: smallest / largest positive cell of a listed clip's map (double bits)
    int *cu_table = nullptr;               // launch_iir2_screen_f64's per-CU arrival counters
    double U = 0.0;
    // per pass: `cap_clips` clips, `cap_cells` = clips x columns of the largest pass
    double *ck_bp = nullptr, *ck_mp = nullptr, *s_bp = nullptr, *mids = nullptr;
    int *labels = nullptr, *loud = nullptr, *want = nullptr, *n_mids = nullptr, *hits = nullptr;
    dsp::ClassifyTraceD *trace = nullptr;
    long cap_clips = 0;
    size_t cap_cells = 0;
    // staging of host input (bytes) and the yardstick pipelines' filtered signals (doubles per row)
    void *x = nullptr;
    size_t cap_x = 0;
    double *y_bp = nullptr, *y_mp = nullptr, *s_mp = nullptr;
    long cap_y_clips = 0, cap_y_row = 0;
    int cap_y_T = 0;
    long last_segments = 0;                // clips x columns of the last pass (dsp_classify_stats_f64)
    hipEvent_t done = nullptr;             // recorded behind the last call's work: the workspace is free once it has fired
    bool pending = false;
    dsp::SpanRing spans;                   // ragged batches: the clips' spans on their way to the GPU (capi_util.hpp)

    void free_pass()
    {
        for (void *p : {(void *)ck_bp, (void *)ck_mp, (void *)s_bp, (void *)mids, (void *)labels, (void *)loud, (void *)want, (void *)n_mids, (void *)hits, (void *)trace, (void *)minmax})
            if (p) (void)hipFree(p);
        minmax = nullptr;
        ck_bp = ck_mp = s_bp = mids = nullptr; labels = loud = want = n_mids = hits = nullptr; trace = nullptr;
        cap_clips = 0; cap_cells = 0;
    }
    void free_yardstick()
    {
        for (void *p : {(void *)y_bp, (void *)y_mp, (void *)s_mp})
            if (p) (void)hipFree(p);
        y_bp = y_mp = s_mp = nullptr; cap_y_clips = cap_y_row = 0; cap_y_T = 0;
    }
    void wait_idle()
    {
        if (pending && done) (void)hipEventSynchronize(done);
        pending = false;
    }
    void release_all()                      // on this->device, which the caller has made current
    {
        wait_idle();
        free_pass();
        free_yardstick();
        if (x) (void)hipFree(x);
        x = nullptr; cap_x = 0;
        if (tab) (void)hipFree(tab);
        if (scr) (void)hipFree(scr);
        if (cu_table) (void)hipFree(cu_table);
        tab = nullptr; scr = nullptr; cu_table = nullptr;
        if (done) (void)hipEventDestroy(done);
        done = nullptr;
        spans.release();
    }
};
Scratch g_w[kMaxDevices];                  // one per device: threads on different GPUs share nothing

// DSP_AMD_F64_SUB_BATCH: a smaller pass (tests: a batch that spans passes).  A pass of the default pipeline is bounded by the
// blocks of the screening kernel that are resident at once (3 per CU: 49 152 clips on 256 CUs).
long sub_batch(Pipeline pl)
{
    const long cap = pl == kCkpt ? (long)dsp::f64_screen_blocks_per_pass() * 64 : 65536;
    const char *e = std::getenv("DSP_AMD_F64_SUB_BATCH");
    const long v = e ? std::atol(e) : 0;
    return v >= 64 ? std::min(v, cap) : cap;
}

bool valid(const dsp_classify_config_f64 &c)
{
    auto fin = [](double v) { return v == v && v - v == 0.0; };
    return fin(c.keep_lo) && fin(c.keep_hi) && fin(c.midpoint_db) && fin(c.middle_max) && fin(c.above_min) && fin(c.below_min) && c.keep_lo < c.keep_hi;
}

void coefficients(dsp::IirCoefD &c_bp, dsp::IirCoefD &c_mp)
{
    double b[9], a[9];
    dsp_butter_bandpass(3000.0, 7500.0, b, a);                       // classifier.c:86-91
    for (int i = 0; i < 9; ++i) { c_bp.b[i] = b[i]; c_bp.a[i] = a[i]; }
    dsp_butter_bandpass(1000.0, 3000.0, b, a);                       // :659-664
    for (int i = 0; i < 9; ++i) { c_mp.b[i] = b[i]; c_mp.a[i] = a[i]; }
}

// one sub-batch resident at d_x (kind `in`, row stride `stride` samples): labels (+ trace) into the scratch arrays
// spans != nullptr: a ragged sub-batch (clip c at spans[c].off samples from d_x; n = the longest clip of the BATCH, total = samples in the buffer)
int run(const dsp_classify_config_f64 &cfg, Scratch &w, Pipeline pl, const void *d_x, int in, long cnt, int n, long stride, bool want_trace, hipStream_t st,
        const dsp::ClipSpan *spans = nullptr, long total = 0)
{
    dsp::IirCoefD c_bp, c_mp;
    coefficients(c_bp, c_mp);
    const dsp::ClassifyRuleD rule{cfg.keep_lo, cfg.keep_hi, cfg.midpoint_db, cfg.middle_max, cfg.above_min, cfg.below_min};
    dsp::ClassifyTraceD *tr = want_trace ? w.trace : nullptr;
    w.last_segments = cnt * (long)columns(n);
    if (spans && pl != kCkpt) return dsp::capi_fail(DSP_EINVAL, "ragged batches run on the default pipeline");
    if (pl == kCkpt) {
        const double guard = dsp::f64_threshold_guard();
        DSP_CAPI_HIP(dsp::launch_iir2_screen_f64(d_x, in, cnt, n, stride, c_bp, c_mp, w.ck_bp, w.ck_mp, w.scr, w.U, cfg.midpoint_db, guard, w.loud, w.want, w.cu_table, st, spans, total));
        DSP_CAPI_HIP(dsp::launch_spec_f64_recheck(d_x, in, cnt, n, stride, c_mp, w.ck_mp, w.tab, w.want, cfg.midpoint_db, guard, w.loud, st, spans));
        DSP_CAPI_HIP(dsp::launch_classify_f64_midpoints(w.loud, cnt, n, 16000, w.mids, w.n_mids, w.hits, w.labels, tr, st, w.minmax, spans));
        DSP_CAPI_HIP(dsp::launch_spec_f64_listed_from_ckpt(d_x, in, cnt, n, stride, c_bp, w.ck_bp, w.tab, w.hits, w.s_bp, st, w.minmax, spans));
        DSP_CAPI_HIP(dsp::launch_classify_f64_bands(w.s_bp, w.hits, cnt, n, 16000, w.U, rule, w.mids, w.n_mids, w.labels, tr, st, w.minmax, spans));
        return DSP_OK;
    }
    const double *xd = static_cast<const double *>(d_x);
    const long row = w.cap_y_row;
    DSP_CAPI_HIP(dsp::launch_iir2_f64(xd, cnt, n, stride, row, c_bp, w.y_bp, c_mp, w.y_mp, st));
    if (pl == kDft) {                         // direct DFT, [129][T] maps of both outputs, one tail kernel per clip
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

// the workspace of `device` (made current by the caller) for passes of `clips` clips of n samples; grows, never shrinks
int reserve(Scratch &w, int device, Pipeline pl, long clips, int n, size_t x_bytes)
{
    const size_t T = (size_t)columns(n);
    w.device = device;
    if (!w.done) DSP_CAPI_HIP(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    if (!w.tab) {
        auto t = std::make_unique<dsp::SpecTablesD>();
        auto s = std::make_unique<dsp::ScreenTablesD>();
        dsp::build_spec_tables_f64(16000, *t);
        if (!dsp::build_screen_tables_f64(*t, 16000, *s)) return dsp::capi_fail(DSP_EINVAL, "screening tables: the window is not flat between its tapers");
        DSP_CAPI_HIP(hipMalloc(&w.tab, sizeof(*t)));
        DSP_CAPI_HIP(hipMemcpy(w.tab, t.get(), sizeof(*t), hipMemcpyHostToDevice));
        DSP_CAPI_HIP(hipMalloc(&w.cu_table, sizeof(int) * (dsp::kSimdLoadCus + 16 * 4096)));      // (+ the diagnostic build's per-block records)
        DSP_CAPI_HIP(hipMalloc(&w.scr, sizeof(*s)));
        DSP_CAPI_HIP(hipMemcpy(w.scr, s.get(), sizeof(*s), hipMemcpyHostToDevice));
        w.U = t->U;
    }
    // (the per-segment arrays are sized by the PRODUCT clips x T of the largest pass: a ragged batch's pass of few long clips and its
    // pass of many short ones share them; by each dimension's maximum a single 13 s clip among 49 152 would ask for 72 GB)
    const size_t cells_needed = (size_t)clips * std::max<size_t>(T, 1);
    if (clips > w.cap_clips || cells_needed > w.cap_cells) {
        w.wait_idle();
        clips = std::max(clips, w.cap_clips);
        const size_t cells = std::max(cells_needed, w.cap_cells);
        w.free_pass();
        const size_t ck = cells * dsp::kCkPerSegF64 * 8 * sizeof(double);
        DSP_CAPI_HIP(hipMalloc(&w.ck_bp, ck));
        DSP_CAPI_HIP(hipMalloc(&w.ck_mp, ck));
        DSP_CAPI_HIP(hipMalloc(&w.s_bp, cells * dsp::kSpecBins * sizeof(double)));
        DSP_CAPI_HIP(hipMalloc(&w.mids, (size_t)clips * dsp::kMaxMidpoints * sizeof(double)));
        DSP_CAPI_HIP(hipMalloc(&w.loud, cells * sizeof(int)));
        DSP_CAPI_HIP(hipMalloc(&w.want, (cells + 1) * sizeof(int)));
        DSP_CAPI_HIP(hipMalloc(&w.n_mids, (size_t)clips * sizeof(int)));
        DSP_CAPI_HIP(hipMalloc(&w.hits, ((size_t)clips + 1) * sizeof(int)));
        DSP_CAPI_HIP(hipMalloc(&w.labels, (size_t)clips * sizeof(int)));
        DSP_CAPI_HIP(hipMalloc(&w.trace, (size_t)clips * sizeof(dsp::ClassifyTraceD)));
        DSP_CAPI_HIP(hipMalloc(&w.minmax, (size_t)clips * 2 * sizeof(unsigned long long)));
        w.cap_clips = clips; w.cap_cells = cells;
    }
    if (x_bytes > w.cap_x) {
        w.wait_idle();
        if (w.x) (void)hipFree(w.x);
        w.x = nullptr; w.cap_x = 0;
        DSP_CAPI_HIP(hipMalloc(&w.x, x_bytes));
        w.cap_x = x_bytes;
    }
    if (pl != kCkpt && (clips > w.cap_y_clips || row_of(n) > w.cap_y_row || (int)T > w.cap_y_T)) {
        w.wait_idle();
        const long yc = std::max(clips, w.cap_y_clips), yr = std::max(row_of(n), w.cap_y_row);
        const size_t yT = std::max(T, (size_t)w.cap_y_T);
        w.free_yardstick();
        DSP_CAPI_HIP(hipMalloc(&w.y_bp, (size_t)yc * yr * sizeof(double)));
        DSP_CAPI_HIP(hipMalloc(&w.y_mp, (size_t)yc * yr * sizeof(double)));
        DSP_CAPI_HIP(hipMalloc(&w.s_mp, (size_t)yc * dsp::kSpecBins * yT * sizeof(double)));
        w.cap_y_clips = yc; w.cap_y_row = yr; w.cap_y_T = (int)yT;
    }
    return DSP_OK;
}

// bytes per sample (all channels) of an input kind
int in_bytes(int in) { return in == 0 ? 8 : in == 1 ? 2 : 4; }

int input_kind(int channels, int stereo_mode, int &in)
{
    if (channels != 1 && channels != 2) return dsp::capi_fail(DSP_EINVAL, "channels must be 1 or 2");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return dsp::capi_fail(DSP_EINVAL, "bad stereo_mode");
    in = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
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

// leaves the "workspace busy until here" event behind the call's work on EVERY exit, so that a failed call cannot hand a workspace
// with kernels still running on it to the next one
struct BusyMark {
    Scratch &w;
    hipStream_t st;
    ~BusyMark()
    {
        if (w.done && hipEventRecord(w.done, st) == hipSuccess) w.pending = true;
        else { (void)hipGetLastError(); (void)hipStreamSynchronize(st); w.pending = false; }
    }
};

int device_entry(const dsp_classify_config_f64 *cfgp, const void *d_signal, int in, long n_clips, int n, long stride, int *d_labels,
                 dsp_classify_trace_f64 *d_trace, void *stream)
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
    if (attr.device < 0 || attr.device >= kMaxDevices) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
    DSP_ON_DEVICE(attr.device);
    hipStream_t st = (hipStream_t)stream;
    if (columns(n) == 0) {                   // shorter than one spectrogram segment: no midpoints, label 0
        DSP_CAPI_HIP(hipMemsetAsync(d_labels, 0, (size_t)n_clips * sizeof(int), st));
        if (d_trace) DSP_CAPI_HIP(hipMemsetAsync(d_trace, 0, (size_t)n_clips * sizeof(dsp_classify_trace_f64), st));
        return DSP_OK;
    }
    if (n_clips == 1) stride = n;
    const Pipeline pl = in == 0 ? pipeline() : kCkpt;
    Scratch &w = g_w[attr.device];
    std::lock_guard<std::mutex> lock(w.mu);
    const long kSubBatch = sub_batch(pl);
    if ((rc = reserve(w, attr.device, pl, std::min(kSubBatch, n_clips), n, 0)) < 0) return rc;
    if (w.pending) DSP_CAPI_HIP(hipStreamWaitEvent(st, w.done, 0));        // the previous call's work on this workspace (any stream)
    BusyMark mark{w, st};
    for (long c0 = 0; c0 < n_clips; c0 += kSubBatch) {
        const long cnt = std::min(kSubBatch, n_clips - c0);
        const void *src = static_cast<const unsigned char *>(d_signal) + (size_t)c0 * stride * in_bytes(in);
        if ((rc = run(cfg, w, pl, src, in, cnt, n, stride, d_trace != nullptr, st)) < 0) return rc;
        DSP_CAPI_HIP(hipMemcpyAsync(d_labels + c0, w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToDevice, st));
        if (d_trace) DSP_CAPI_HIP(hipMemcpyAsync(d_trace + c0, w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToDevice, st));
    }
    return DSP_OK;
}

int host_entry(const dsp_classify_config_f64 *cfgp, const void *signal, int in, long n_clips, int n, long stride, int *labels, dsp_classify_trace_f64 *trace)
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
    if (device < 0 || device >= count || device >= kMaxDevices) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
    DSP_ON_DEVICE(device);
    if (n_clips == 1) stride = n;
    const Pipeline pl = in == 0 ? pipeline() : kCkpt;
    Scratch &w = g_w[device];
    std::lock_guard<std::mutex> lock(w.mu);
    const long kSubBatch = sub_batch(pl);
    const int bps = in_bytes(in);
    const long row = in == 0 ? row_of(n) : (((long)n * bps + 15) & ~15L) / bps;      // staged rows start on 16 bytes
    const long pass = std::min(kSubBatch, n_clips);
    if ((rc = reserve(w, device, pl, pass, n, (size_t)pass * row * bps)) < 0) return rc;
    w.wait_idle();                                                              // the staging buffer is written by blocking copies
    BusyMark mark{w, nullptr};
    for (long c0 = 0; c0 < n_clips; c0 += kSubBatch) {
        const long cnt = std::min(kSubBatch, n_clips - c0);
        DSP_CAPI_HIP(hipMemcpy2D(w.x, (size_t)row * bps, static_cast<const unsigned char *>(signal) + (size_t)c0 * stride * bps, (size_t)stride * bps,
                                 (size_t)n * bps, (size_t)cnt, hipMemcpyHostToDevice));
        if ((rc = run(cfg, w, pl, w.x, in, cnt, n, row, trace != nullptr, nullptr)) < 0) return rc;
        DSP_CAPI_HIP(hipMemcpy(labels + c0, w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost));
        if (trace) DSP_CAPI_HIP(hipMemcpy(trace + c0, w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToHost));
    }
    return DSP_OK;
}

// Ragged batches (classifier.c:286-297 reads a file of any length; its callers loop over files): offsets[n_clips + 1] (host) -> spans;
// d_signal = the whole buffer on `device`.  Results to d_labels / d_trace (device) or labels / trace (host), whichever are given.
int ragged(const dsp_classify_config_f64 *cfgp, const void *d_signal, int device, int in, long n_clips, const long *offsets, int *d_labels,
           dsp_classify_trace_f64 *d_trace, int *labels, dsp_classify_trace_f64 *trace, void *stream)
{
    dsp_classify_config_f64 cfg;
    if (cfgp) cfg = *cfgp; else dsp_classify_default_config_f64(&cfg);
    if (!valid(cfg)) return dsp::capi_fail(DSP_EINVAL, "classify config: thresholds must be finite with keep_lo < keep_hi");
    int n_max = 0;
    for (long c = 0; c < n_clips; ++c) {
        const long n = offsets[c + 1] - offsets[c];
        if (offsets[c] < 0 || n < 0 || n > INT32_MAX) return dsp::capi_fail(DSP_EINVAL, "offsets must be non-negative and non-decreasing, clips shorter than 2^31 samples");
        if (columns((int)n) > kMaxColumns) return dsp::capi_fail(DSP_EINVAL, "clip " + std::to_string(c) + " too long (more than 957 spectrogram columns = 13.4 s at 16 kHz)");
        n_max = std::max(n_max, (int)n);
    }
    DSP_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    if (columns(n_max) == 0) {                 // no clip holds a segment: no midpoints, label 0
        if (d_labels) DSP_CAPI_HIP(hipMemsetAsync(d_labels, 0, (size_t)n_clips * sizeof(int), st));
        if (d_trace) DSP_CAPI_HIP(hipMemsetAsync(d_trace, 0, (size_t)n_clips * sizeof(dsp_classify_trace_f64), st));
        if (labels) for (long c = 0; c < n_clips; ++c) labels[c] = 0;
        if (trace) for (long c = 0; c < n_clips; ++c) trace[c] = dsp_classify_trace_f64{};
        return DSP_OK;
    }
    Scratch &w = g_w[device];
    std::lock_guard<std::mutex> lock(w.mu);
    const long kSubBatch = sub_batch(kCkpt);
    int rc = DSP_OK;
    // in order of length, longest first (a block's 64 clips alike: it walks to its longest); order[i] = the caller's index of the i-th
    // clip as run, the results go home through it
    if (n_clips >= (1L << 31)) return dsp::capi_fail(DSP_EINVAL, "too many clips");
    std::vector<int> order((size_t)n_clips), segs((size_t)n_clips);
    for (long c = 0; c < n_clips; ++c) segs[c] = columns((int)(offsets[c + 1] - offsets[c]));
    dsp::order_by_key_desc(segs.data(), n_clips, columns(n_max), order.data());
    const size_t span_bytes = (size_t)n_clips * sizeof(dsp::ClipSpan), perm_bytes = (size_t)n_clips * sizeof(int);
    dsp::SpanRing::Slot *slot = nullptr;
    DSP_CAPI_HIP(w.spans.acquire(span_bytes + perm_bytes, &slot));
    dsp::ClipSpan *h = static_cast<dsp::ClipSpan *>(slot->h);
    for (long i = 0; i < n_clips; ++i) {
        const long c = order[i];
        h[i] = dsp::ClipSpan{offsets[c], (int)(offsets[c + 1] - offsets[c]), columns((int)(offsets[c + 1] - offsets[c])), c, 0};
    }
    std::memcpy(static_cast<char *>(slot->h) + span_bytes, order.data(), perm_bytes);
    if (w.pending) DSP_CAPI_HIP(hipStreamWaitEvent(st, w.done, 0));
    DSP_CAPI_HIP(dsp::SpanRing::upload(slot, span_bytes + perm_bytes, st));
    const int *d_perm = reinterpret_cast<const int *>(static_cast<const char *>(slot->d) + span_bytes);
    std::vector<int> h_labels;
    std::vector<dsp_classify_trace_f64> h_trace;
    struct SlotMark { dsp::SpanRing::Slot *s; hipStream_t st; ~SlotMark() { dsp::SpanRing::mark(s, st); } } slot_mark{slot, st};
    BusyMark mark{w, st};
    const dsp::ClipSpan *d_spans = static_cast<const dsp::ClipSpan *>(slot->d);
    const bool want_trace = d_trace != nullptr || trace != nullptr;
    // passes: as many clips as a pass of equal 1 s clips has segments for (few while the clips are long, the full pass once they are short)
    // (four times that before a pass is cut short: a small remainder pass costs a whole clip's sequential chain for few clips)
    const long pass_cells = 4 * kSubBatch * 71;
    struct Pass { long c0, cnt; int n_row; };
    std::vector<Pass> passes;
    for (long c0 = 0; c0 < n_clips;) {
        const int t_row = std::max(1, h[c0].frames);                            // sorted: the pass's longest clip comes first
        const long cnt = std::min({kSubBatch, n_clips - c0, std::max(64L, pass_cells / t_row)});
        passes.push_back(Pass{c0, cnt, (t_row - 1) * dsp::kSpecHop + dsp::kSpecSeg});
        c0 += cnt;
    }
    for (const Pass &ps : passes)
        if ((rc = reserve(w, device, kCkpt, ps.cnt, ps.n_row, 0)) < 0) return rc;
    for (const Pass &ps : passes) {
        const long c0 = ps.c0, cnt = ps.cnt;
        if ((rc = run(cfg, w, kCkpt, d_signal, in, cnt, ps.n_row, 0, want_trace, st, d_spans + c0, offsets[n_clips])) < 0) return rc;
        if (d_labels) DSP_CAPI_HIP(dsp::launch_scatter_records(w.labels, d_perm + c0, cnt, sizeof(int), d_labels, st));
        if (d_trace) DSP_CAPI_HIP(dsp::launch_scatter_records(w.trace, d_perm + c0, cnt, sizeof(dsp_classify_trace_f64), d_trace, st));
        if (labels) {
            h_labels.resize((size_t)cnt);
            DSP_CAPI_HIP(hipMemcpyAsync(h_labels.data(), w.labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, st));
        }
        if (trace) {
            h_trace.resize((size_t)cnt);
            DSP_CAPI_HIP(hipMemcpyAsync(h_trace.data(), w.trace, (size_t)cnt * sizeof(dsp_classify_trace_f64), hipMemcpyDeviceToHost, st));
        }
        if (labels || trace) {
            DSP_CAPI_HIP(hipStreamSynchronize(st));
            for (long i = 0; i < cnt; ++i) {
                if (labels) labels[order[c0 + i]] = h_labels[i];
                if (trace) trace[order[c0 + i]] = h_trace[i];
            }
        }
    }
    return DSP_OK;
}

int ragged_device_entry(const dsp_classify_config_f64 *cfgp, const void *d_signal, int in, long n_clips, const long *offsets, int *d_labels,
                        dsp_classify_trace_f64 *d_trace, void *stream)
{
    if (!d_signal || !d_labels || !offsets || n_clips < 0) return dsp::capi_fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0) return DSP_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_signal) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return dsp::capi_fail(DSP_EINVAL, "signal is not a device pointer");
    }
    if (attr.device < 0 || attr.device >= kMaxDevices) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
    return ragged(cfgp, d_signal, attr.device, in, n_clips, offsets, d_labels, d_trace, nullptr, nullptr, stream);
}

int ragged_host_entry(const dsp_classify_config_f64 *cfgp, const void *signal, int in, long n_clips, const long *offsets, int *labels, dsp_classify_trace_f64 *trace)
{
    if (!signal || !labels || !offsets || n_clips < 0) return dsp::capi_fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0) return DSP_OK;
    if (offsets[n_clips] < offsets[0] || offsets[0] < 0) return dsp::capi_fail(DSP_EINVAL, "offsets must be non-negative and non-decreasing, clips shorter than 2^31 samples");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return dsp::capi_fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback");
    const char *dev = std::getenv("DSP_AMD_DEVICE");
    const int device = dev ? std::atoi(dev) : 0;
    if (device < 0 || device >= count || device >= kMaxDevices) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
    void *d_flat = nullptr;
    const size_t bytes = (size_t)offsets[n_clips] * in_bytes(in);
    {
        DSP_ON_DEVICE(device);
        DSP_CAPI_HIP(hipMalloc(&d_flat, bytes + 16));
        const hipError_t e = hipMemcpy(d_flat, signal, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d_flat); DSP_CAPI_HIP(e); }
    }
    const int rc = ragged(cfgp, d_flat, device, in, n_clips, offsets, nullptr, nullptr, labels, trace, nullptr);
    {
        dsp::DeviceScope on(device);
        (void)hipStreamSynchronize(nullptr);
        (void)hipFree(d_flat);
    }
    return rc;
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
    return device_entry(cfgp, d_signal, 0, n_clips, n, stride, d_labels, d_trace, stream);
}

int dsp_classify_batch_host_f64(const dsp_classify_config_f64 *cfgp, const double *signal, long n_clips, int n, long stride,
                                int *labels, dsp_classify_trace_f64 *trace)
{
    return host_entry(cfgp, signal, 0, n_clips, n, stride, labels, trace);
}

int dsp_classify_batch_pcm16_device_f64(const dsp_classify_config_f64 *cfgp, const int16_t *d_pcm, long n_clips, int n, long stride, int channels,
                                        int stereo_mode, int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream)
{
    int in = 0;
    const int rc = input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : device_entry(cfgp, d_pcm, in, n_clips, n, stride, d_labels, d_trace, stream);
}

int dsp_classify_batch_pcm16_host_f64(const dsp_classify_config_f64 *cfgp, const int16_t *pcm, long n_clips, int n, long stride, int channels,
                                      int stereo_mode, int *labels, dsp_classify_trace_f64 *trace)
{
    int in = 0;
    const int rc = input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : host_entry(cfgp, pcm, in, n_clips, n, stride, labels, trace);
}

int dsp_classify_batch_ragged_device_f64(const dsp_classify_config_f64 *cfgp, const double *d_signal, long n_clips, const long *offsets, int *d_labels,
                                         dsp_classify_trace_f64 *d_trace, void *stream)
{
    return ragged_device_entry(cfgp, d_signal, 0, n_clips, offsets, d_labels, d_trace, stream);
}

int dsp_classify_batch_ragged_pcm16_device_f64(const dsp_classify_config_f64 *cfgp, const int16_t *d_pcm, long n_clips, const long *offsets, int channels,
                                               int stereo_mode, int *d_labels, dsp_classify_trace_f64 *d_trace, void *stream)
{
    int in = 0;
    const int rc = input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : ragged_device_entry(cfgp, d_pcm, in, n_clips, offsets, d_labels, d_trace, stream);
}

int dsp_classify_batch_ragged_host_f64(const dsp_classify_config_f64 *cfgp, const double *signal, long n_clips, const long *offsets, int *labels,
                                       dsp_classify_trace_f64 *trace)
{
    return ragged_host_entry(cfgp, signal, 0, n_clips, offsets, labels, trace);
}

int dsp_classify_batch_ragged_pcm16_host_f64(const dsp_classify_config_f64 *cfgp, const int16_t *pcm, long n_clips, const long *offsets, int channels,
                                             int stereo_mode, int *labels, dsp_classify_trace_f64 *trace)
{
    int in = 0;
    const int rc = input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : ragged_host_entry(cfgp, pcm, in, n_clips, offsets, labels, trace);
}

int dsp_classify_stats_f64(int device, long *segments, long *undecided, long *listed_clips)
{
    if (device < 0 || device >= kMaxDevices) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
    Scratch &w = g_w[device];
    std::lock_guard<std::mutex> lock(w.mu);
    if (w.device < 0 || !w.want) return dsp::capi_fail(DSP_EINVAL, "no float64 classifier pass has run on this device");
    DSP_ON_DEVICE(device);
    w.wait_idle();
    int nw = 0, nh = 0;
    DSP_CAPI_HIP(hipMemcpy(&nw, w.want, sizeof(int), hipMemcpyDeviceToHost));
    DSP_CAPI_HIP(hipMemcpy(&nh, w.hits, sizeof(int), hipMemcpyDeviceToHost));
    if (segments) *segments = w.last_segments;
    if (undecided) *undecided = nw;
    if (listed_clips) *listed_clips = nh;
    return DSP_OK;
}

int dsp_classify_debug_f64(int device, int *out, int n_ints)      /* diagnostic builds: the screening kernel's per-block records */
{
    if (device < 0 || device >= kMaxDevices || !out || n_ints < 0 || n_ints > 16 * 4096) return dsp::capi_fail(DSP_EINVAL, "bad argument");
    Scratch &w = g_w[device];
    std::lock_guard<std::mutex> lock(w.mu);
    if (!w.cu_table) return dsp::capi_fail(DSP_EINVAL, "no pass has run");
    DSP_ON_DEVICE(device);
    w.wait_idle();
    DSP_CAPI_HIP(hipMemcpy(out, w.cu_table + dsp::kSimdLoadCus, sizeof(int) * (size_t)n_ints, hipMemcpyDeviceToHost));
    return DSP_OK;
}

int dsp_classify_release_f64(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    for (int d = 0; d < kMaxDevices && d < count; ++d) {
        if (device >= 0 && d != device) continue;
        Scratch &w = g_w[d];
        std::lock_guard<std::mutex> lock(w.mu);
        if (w.device < 0) continue;
        DSP_ON_DEVICE(d);
        w.release_all();
        w.device = -1;
    }
    return DSP_OK;
}

}  // extern "C"
