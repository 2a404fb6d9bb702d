// capi.cpp -- the C ABI of libdsp_amd.so (include/dsp_amd.h).
//
// Host side of the drop-in boundary: owns plans (device tables + staging
// buffers), validates arguments the way the reference does, and enqueues the
// gfx950 kernels.  No CPU fallback exists: without a HIP device every compute
// entry point fails and says why through dsp_last_error().
#include "diag_guard.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "../../include/dsp_amd.h"
#include "capi_util.hpp"
#include "classify_kernels.hpp"
#include "mfcc_kernels.hpp"
#include "svm_kernels.hpp"
#include "tables.hpp"

#ifdef DSP_PF_STAMPS
namespace dsp { hipError_t read_pf_stamps(unsigned long long *host, int count); }      // mfcc_kernels.hip, diagnostic builds
#endif
#ifdef DSP_RC_STAMPS
namespace dsp { hipError_t read_rc_stamps(unsigned long long *host, int count); hipError_t read_bd_stamps(unsigned long long *host, int count); }      // classify_kernels.hip, diagnostic builds
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

}  // namespace

// error sink shared with the other translation units of the C ABI (capi_util.hpp)
namespace dsp { int capi_fail(int code, const std::string &msg) { return fail(code, msg); } }

namespace {

#define DSP_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(DSP_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

bool valid_cfg(const dsp_mfcc_config &c, std::string &why)
{
    if (c.sample_rate <= 0) { why = "sample_rate must be positive"; return false; }
    if (c.hop_length <= 0) { why = "hop_length must be positive"; return false; }
    if (c.frame_length < 2) { why = "frame_length must be at least 2"; return false; }
    if (c.n_mels < 1 || c.n_mfcc < 1) { why = "n_mels and n_mfcc must be positive"; return false; }
    if (c.frame_length & 1) { why = "frame_length must be even (8-byte aligned frame loads)"; return false; }
    if (c.hop_length & 1) { why = "hop_length must be even (8-byte aligned frame loads)"; return false; }
    if (!(c.fmax > c.fmin) || c.fmin < 0) { why = "need 0 <= fmin < fmax"; return false; }
    if (!(c.amin > 0)) { why = "amin must be positive"; return false; }
    if (c.log_mode != DSP_LOG_PER_FRAME_MAX && c.log_mode != DSP_LOG_GLOBAL_REF1 && c.log_mode != DSP_LOG_LOG10_FLOOR) { why = "unknown log_mode"; return false; }
    if (c.mel_norm != DSP_MELNORM_NONE && c.mel_norm != DSP_MELNORM_SLANEY && c.mel_norm != DSP_MELNORM_LIBROSA && c.mel_norm != DSP_MELNORM_AUBIO_SLANEY) { why = "unknown mel_norm"; return false; }
    if (c.spectrum != DSP_SPECTRUM_POWER && c.spectrum != DSP_SPECTRUM_MAGNITUDE) { why = "unknown spectrum"; return false; }
    if (c.framing != DSP_FRAMING_COMPLETE && c.framing != DSP_FRAMING_STREAM) { why = "unknown framing"; return false; }
    // the aubio-semantics options of cepstrum/scrubjay_infer.c's front end live on the 2048-point kernel
    if (c.n_fft != 2048 && (c.mel_norm == DSP_MELNORM_AUBIO_SLANEY || c.log_mode == DSP_LOG_LOG10_FLOOR || c.spectrum != DSP_SPECTRUM_POWER ||
                            c.framing != DSP_FRAMING_COMPLETE)) {
        why = "DSP_MELNORM_AUBIO_SLANEY, DSP_LOG_LOG10_FLOOR, DSP_SPECTRUM_MAGNITUDE and DSP_FRAMING_STREAM are implemented for n_fft = 2048";
        return false;
    }
    if (c.mel_norm == DSP_MELNORM_AUBIO_SLANEY && c.n_mels != 40) { why = "DSP_MELNORM_AUBIO_SLANEY is aubio's 40-filter bank: n_mels must be 40"; return false; }
    if (c.framing == DSP_FRAMING_STREAM && c.hop_length > c.frame_length) { why = "DSP_FRAMING_STREAM needs hop_length <= frame_length"; return false; }
    if (c.log_mode == DSP_LOG_GLOBAL_REF1 && c.n_fft == 1024) { why = "DSP_LOG_GLOBAL_REF1 is implemented for n_fft = 512 and 2048"; return false; }
    if (c.prefilter != DSP_PREFILTER_NONE && c.n_fft == 2048) { why = "the per-frame prefilter is implemented for n_fft = 512 and 1024"; return false; }
    if (c.frame_length > c.n_fft) { why = "frame_length must not exceed n_fft"; return false; }
    if (c.win_length < 0 || c.win_length > c.frame_length) { why = "win_length must be in [0, frame_length]"; return false; }
    if (c.prefilter != DSP_PREFILTER_NONE && c.prefilter != DSP_PREFILTER_BUTTER_1000_3000 &&
        c.prefilter != DSP_PREFILTER_BUTTER_3000_7500) { why = "unknown prefilter"; return false; }
    if (c.n_fft != 512 && c.n_fft != 1024 && c.n_fft != 2048) { why = "n_fft must be 512, 1024 or 2048"; return false; }
    return true;
}

}  // namespace

struct dsp_mfcc_plan {
    dsp_mfcc_config cfg;
    int device = 0;
    int n_cu = 0;
    int resident_blocks = 4; // 256-thread blocks one CU holds (occupancy query), tile epilogue kernel
    int resident_blocks_frame = 4;   // same, per-frame epilogue kernel
    int blocks_per_cu = 0;   // 0 = default (= resident_blocks)
    int chunk = 0;           // 0 = default
    dsp::LaneTables512 host;
    dsp::LaneTables512 *d_tables = nullptr;
    dsp::RowTables512 *d_row_tables = nullptr;
    dsp::GenTables1024 *d_gen_tables = nullptr;   // n_fft = 1024
    dsp::GenTables2048 *d_tables2048 = nullptr;   // n_fft = 2048
    dsp::PairExtra512 *d_pair = nullptr;          // n_fft = 512: extra constants of the two-frames-per-wave kernel (DSP_KERNEL_PAIR)
    int resident_blocks_pair = 3;
    int resident_blocks_2048 = 2, resident_blocks_2048_pool = 2;
    int resident_blocks_gen = 3;
    int gen_slots = 0;                            // mel chunk slots per lane the 1024-point tables use (<= 3: wave kernel)
    int resident_blocks_gen_wave = 2;
    dsp::PrefilterScan *d_scan = nullptr;         // prefilter fused into the 1024-point wave kernel (full frames): its tables
    int scan_steps[4] = {6, 6, 6, 6};             // host copy of PrefilterScan::c_steps (picks the kernel instantiation)
    int resident_blocks_gen_pre = 2;
    float *d_filtered = nullptr;                  // per-frame prefilter output (sub-batch)
    size_t filtered_cap = 0;
    float *d_frame_max = nullptr, *d_clip_floor = nullptr;   // DSP_LOG_GLOBAL_REF1 two-pass workspace
    size_t frame_max_cap = 0, clip_floor_cap = 0;
    int kernel = DSP_KERNEL_WAVE;
    int resident_blocks_row = 3;
    // staging for the host-pointer entry points
    float *d_in = nullptr, *d_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    // Guards the plan's workspaces (d_filtered, d_frame_max / d_clip_floor, d_in / d_out) while a call reserves them and
    // enqueues the kernels that use them.  The kernels themselves run after the lock is released: a plan whose path uses
    // a workspace (prefilter, DSP_LOG_GLOBAL_REF1 over clips, the *_host entry points) serves ONE stream at a time;
    // the workspace-free paths (frames / clips / pcm16 / fused, per-frame log mode) may be driven from several streams.
    std::recursive_mutex mu;
    dsp::SpanRing spans;      // ragged batches of the fused clip kernels: the clips' spans on their way to the GPU (capi_util.hpp)
};

extern "C" {

const char *dsp_last_error(void) { return g_err.c_str(); }
#ifndef DSP_AMD_SRC_HASH
#define DSP_AMD_SRC_HASH "unknown"
#endif
#ifdef DSP_AMD_EXPERIMENTS
#define DSP_AMD_EXPERIMENTS_TAG " +experiments"
#else
#define DSP_AMD_EXPERIMENTS_TAG ""
#endif
const char *dsp_version(void) { return "dsp_amd 0.3 (gfx950)" DSP_AMD_EXPERIMENTS_TAG " src:" DSP_AMD_SRC_HASH; }

int dsp_abi_sizeof(int which)
{
    switch (which) {
    case 0: return (int)sizeof(dsp_mfcc_config);
    case 1: return (int)sizeof(dsp_classify_trace);
    case 2: return (int)sizeof(dsp_classify_trace_f64);
    default: return -1;
    }
}

int dsp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void dsp_mfcc_default_config(dsp_mfcc_config *c)
{
    // 2fa/audio/word/c/mfcc_params.h:6-12, mfcc.c:172-173, export_mfcc_params.py:44-57
    c->sample_rate = 16000;
    c->n_fft = 512;
    c->frame_length = 400;
    c->hop_length = 160;
    c->n_mels = 40;
    c->n_mfcc = 13;
    c->window = DSP_WINDOW_HANN;
    c->mel_norm = DSP_MELNORM_NONE;
    c->log_mode = DSP_LOG_PER_FRAME_MAX;
    c->prefilter = DSP_PREFILTER_NONE;
    c->spectrum = DSP_SPECTRUM_POWER;
    c->framing = DSP_FRAMING_COMPLETE;
    c->win_length = 0;
    c->fmin = 0.0f;
    c->fmax = 8000.0f;
    c->amin = 1e-10f;
    c->top_db = 80.0f;
}

void dsp_mfcc_scrubjay_infer_config(dsp_mfcc_config *c, int sample_rate)
{
    // cepstrum/scrubjay_infer.c:9-13 (N_MFCC 20, WIN_SIZE 2048, HOP_SIZE 1024, N_FILTERS 40), :28-30 (new_aubio_pvoc, new_aubio_mfcc)
    dsp_mfcc_default_config(c);
    c->sample_rate = sample_rate;
    c->n_fft = 2048;
    c->frame_length = 2048;
    c->hop_length = 1024;
    c->n_mels = 40;
    c->n_mfcc = 20;
    c->window = DSP_WINDOW_HANN;                 // new_aubio_window("hanningz"): the periodic Hann
    c->mel_norm = DSP_MELNORM_AUBIO_SLANEY;
    c->log_mode = DSP_LOG_LOG10_FLOOR;
    c->spectrum = DSP_SPECTRUM_MAGNITUDE;
    c->framing = DSP_FRAMING_STREAM;
    c->fmin = 0.0f;
    c->fmax = 0.5f * (float)sample_rate;         // not used by the aubio bank
}

int dsp_mfcc_frames_for(const dsp_mfcc_config *cfg, int num_samples, int max_frames)
{
    if (!cfg || max_frames <= 0) return 0;
    if (cfg->framing == DSP_FRAMING_STREAM) {
        // cepstrum/scrubjay_infer.c:39-53: a frame per aubio_source_do that returned samples
        if (num_samples <= 0) return 0;
        const long t = ((long)num_samples + cfg->hop_length - 1) / cfg->hop_length;
        return (int)std::min<long>(t, max_frames);
    }
    // mfcc.c:117-119, 132-139
    if (num_samples < cfg->frame_length) return 0;
    const int t = 1 + (num_samples - cfg->frame_length) / cfg->hop_length;
    return std::min(t, max_frames);
}

int dsp_mfcc_tables(const dsp_mfcc_config *cfg, float *window, float *mel, float *dct)
{
    if (!cfg) return fail(DSP_EINVAL, "cfg is NULL");
    {   // (the sanitizer tier's sweep found this entry point building tables for configurations dsp_mfcc_plan_create refuses --
        // sample_rate 0, fmin > fmax, n_fft 333: NaN tables rather than an error)
        std::string why;
        if (!valid_cfg(*cfg, why)) return fail(DSP_EINVAL, why);
    }
    if (window) {
        auto w = dsp::make_frame_window(*cfg);
        std::memcpy(window, w.data(), w.size() * sizeof(float));
    }
    if (mel) {
        auto m = dsp::make_mel_filterbank(cfg->sample_rate, cfg->n_fft, cfg->n_mels, cfg->fmin, cfg->fmax, cfg->mel_norm);
        std::memcpy(mel, m.data(), m.size() * sizeof(float));
    }
    if (dct) {
        auto d = dsp::make_dct_ortho(cfg->n_mfcc, cfg->n_mels);
        std::memcpy(dct, d.data(), d.size() * sizeof(float));
    }
    return DSP_OK;
}

int dsp_prefilter_scan_check(int prefilter, int *steps4)
{
    if (prefilter != DSP_PREFILTER_BUTTER_1000_3000 && prefilter != DSP_PREFILTER_BUTTER_3000_7500) return fail(DSP_EINVAL, "prefilter must name one of the two literal band-passes");
    double b[9], a[9];
    dsp_butter_bandpass(prefilter == DSP_PREFILTER_BUTTER_1000_3000 ? 1000 : 3000, prefilter == DSP_PREFILTER_BUTTER_1000_3000 ? 3000 : 7500, b, a);
    dsp::PrefilterScan sc;
    std::string why;
    if (!dsp::build_prefilter_scan(b, a, sc, why)) return fail(DSP_EINVAL, why);
    if (steps4) for (int k = 0; k < 4; ++k) steps4[k] = sc.c_steps[k];
    return (sc.c_ok ? 1 : 0) | (sc.c_row_ok ? 2 : 0);
}

int dsp_mfcc_lane_tables(const dsp_mfcc_config *cfg, void *out, int size)
{
    if (!cfg) return fail(DSP_EINVAL, "cfg is NULL");
    if (!out) return (int)sizeof(dsp::LaneTables512);
    if (size != (int)sizeof(dsp::LaneTables512)) return fail(DSP_EINVAL, "size != sizeof(LaneTables512)");
    std::string why;
    auto *t = new dsp::LaneTables512;
    const bool ok = valid_cfg(*cfg, why) && dsp::build_lane_tables_512(*cfg, *t, why);
    if (ok) std::memcpy(out, t, sizeof(*t));
    delete t;
    return ok ? DSP_OK : fail(DSP_EINVAL, why);
}

int dsp_butter_bandpass(double lowcut, double highcut, double *b, double *a)
{
    // donut-classifier/classifier.c:342-360, 383-401: the 16 kHz literal tables
    static const double B1[9] = {0.01020948, 0., -0.04083792, 0., 0.06125688, 0., -0.04083792, 0., 0.01020948};
    static const double A1[9] = {1., -4.56803686, 9.95922498, -13.49912589, 12.43979269, -7.94997696, 3.43760562, -0.92305481, 0.1203896};
    static const double B2[9] = {0.1362017, 0., -0.5448068, 0., 0.8172102, 0., -0.5448068, 0., 0.1362017};
    static const double A2[9] = {1., 2.60935592, 2.32553038, 1.20262614, 1.11690211, 0.76154474, 0.10005124, -0.0129829, 0.02236815};
    const double *sb, *sa;
    if (lowcut == 1000 && highcut == 3000) { sb = B1; sa = A1; }
    else if (lowcut == 3000 && highcut == 7500) { sb = B2; sa = A2; }
    else { fail(DSP_EINVAL, "invalid bandpass range"); return 0; }   // classifier.c:402-407
    std::memcpy(b, sb, sizeof(B1));
    std::memcpy(a, sa, sizeof(A1));
    return 1;
}

int dsp_mfcc_plan_create(const dsp_mfcc_config *cfg, int device, dsp_mfcc_plan **out)
{
    if (!cfg || !out) return fail(DSP_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    std::string why;
    if (!valid_cfg(*cfg, why)) return fail(DSP_EINVAL, why);
    auto *p = new dsp_mfcc_plan;
    p->cfg = *cfg;
    dsp::GenTables1024 *gen = nullptr;
    dsp::GenTables2048 *g2k = nullptr;
    if (cfg->n_fft == 2048) {
        g2k = new dsp::GenTables2048;
        if (!dsp::build_gen_tables_2048(*cfg, *g2k, why)) { delete g2k; delete p; return fail(DSP_EINVAL, why); }
    } else if (cfg->n_fft == 1024) {
        gen = new dsp::GenTables1024;
        if (!dsp::build_gen_tables_1024(*cfg, *gen, why)) { delete gen; delete p; return fail(DSP_EINVAL, why); }
    } else if (!dsp::build_lane_tables_512(*cfg, p->host, why)) { delete p; return fail(DSP_EINVAL, why); }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { delete gen; delete g2k; delete p; return fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback"); }
    if (device < 0 || device >= n) { delete gen; delete g2k; delete p; return fail(DSP_EINVAL, "device index out of range"); }
    p->device = device;
    dsp::DeviceScope dsp_device_scope_(device);      // the caller's current device is put back on return
    hipError_t e = dsp_device_scope_.err;
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipMalloc(&p->d_tables, sizeof(dsp::LaneTables512));
    if (e == hipSuccess) e = hipMemcpy(p->d_tables, &p->host, sizeof(dsp::LaneTables512), hipMemcpyHostToDevice);
    if (e == hipSuccess && gen) {
        e = hipMalloc(&p->d_gen_tables, sizeof(dsp::GenTables1024));
        if (e == hipSuccess) e = hipMemcpy(p->d_gen_tables, gen, sizeof(*gen), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && g2k) {
        e = hipMalloc(&p->d_tables2048, sizeof(dsp::GenTables2048));
        if (e == hipSuccess) e = hipMemcpy(p->d_tables2048, g2k, sizeof(*g2k), hipMemcpyHostToDevice);
    }
    delete g2k;
    if (gen) p->gen_slots = gen->n_chunk_slots;
    delete gen;
    if (e == hipSuccess && cfg->n_fft == 1024 && cfg->prefilter != DSP_PREFILTER_NONE && cfg->frame_length == 1024 && p->gen_slots <= 3) {
        // BASELINE config 3 in ONE pass: the per-frame Butterworth as a scan inside the MFCC kernel (tables.hpp PrefilterScan)
        double b[9], a[9];
        dsp_butter_bandpass(cfg->prefilter == DSP_PREFILTER_BUTTER_1000_3000 ? 1000 : 3000,
                            cfg->prefilter == DSP_PREFILTER_BUTTER_1000_3000 ? 3000 : 7500, b, a);
        dsp::PrefilterScan sc;
        if (!dsp::build_prefilter_scan(b, a, sc, why)) { dsp_mfcc_plan_destroy(p); return fail(DSP_EINVAL, why); }
        if (sc.c_ok && (!DSP_PRE_ROWSCAN || sc.c_row_ok)) {      // the kernel runs the cascade form (its scan in row form); coefficients without it (none of the two literal sets) take the two-pass path
            e = hipMalloc(&p->d_scan, sizeof(sc));
            if (e == hipSuccess) e = hipMemcpy(p->d_scan, &sc, sizeof(sc), hipMemcpyHostToDevice);
            for (int k = 0; k < 4; ++k) p->scan_steps[k] = sc.c_steps[k];
        }
    }
#ifdef DSP_AMD_EXPERIMENTS
    // measured dead ends kept buildable (python -m dsp_amd.build with DSP_AMD_EXPERIMENTS=1): the row-per-frame kernel and
    // the two-frames-per-wave kernel; the default library does not carry them
    if (e == hipSuccess && cfg->n_fft == 512) {
        e = hipMalloc(&p->d_row_tables, sizeof(dsp::RowTables512));
        if (e == hipSuccess) {
            auto *rt = new dsp::RowTables512;
            dsp::build_row_tables_512(*cfg, *rt);
            e = hipMemcpy(p->d_row_tables, rt, sizeof(*rt), hipMemcpyHostToDevice);
            delete rt;
        }
    }
    if (e == hipSuccess && cfg->n_fft == 512) {
        dsp::PairExtra512 px;
        dsp::build_pair_extra_512(px);
        e = hipMalloc(&p->d_pair, sizeof(px));
        if (e == hipSuccess) e = hipMemcpy(p->d_pair, &px, sizeof(px), hipMemcpyHostToDevice);
    }
#endif
    if (e != hipSuccess) {
        dsp_mfcc_plan_destroy(p);          // frees every member that was allocated
        return fail(DSP_EHIP, std::string("plan_create: ") + hipGetErrorString(e));
    }
    p->n_cu = prop.multiProcessorCount;
    if (cfg->n_fft == 2048) {
        const bool aub = cfg->spectrum != DSP_SPECTRUM_POWER || cfg->log_mode == DSP_LOG_LOG10_FLOOR || cfg->framing == DSP_FRAMING_STREAM;
        p->resident_blocks_2048 = dsp::mfcc2048_blocks_per_cu(cfg->n_mels, false, aub);
        p->resident_blocks_2048_pool = dsp::mfcc2048_blocks_per_cu(cfg->n_mels, true, aub);
    } else if (cfg->n_fft == 512) {
        p->resident_blocks_frame = dsp::mfcc512_blocks_per_cu(p->host.dct_split, p->host.dct_len, p->host.mel_gather,
                                                              cfg->frame_length == 512, false);
        p->resident_blocks = dsp::mfcc512_blocks_per_cu(p->host.dct_split, p->host.dct_len, p->host.mel_gather,
                                                        cfg->frame_length == 512, true);
#ifdef DSP_AMD_EXPERIMENTS
        p->resident_blocks_row = dsp::mfcc512_row_blocks_per_cu(p->host.dct_split, p->host.dct_len, p->host.mel_gather,
                                                                cfg->frame_length == 512);
        p->resident_blocks_pair = dsp::mfcc512_pair_blocks_per_cu();
#endif
    } else {
        p->resident_blocks_gen = dsp::mfcc1024_blocks_per_cu(cfg->frame_length == 1024);
        p->resident_blocks_gen_wave = dsp::mfcc1024_wave_blocks_per_cu(cfg->frame_length == 1024);
        if (p->d_scan) p->resident_blocks_gen_pre = dsp::mfcc1024_wave_blocks_per_cu(true, true);
    }
    if (const char *k = std::getenv("DSP_AMD_KERNEL")) {
        // an A/B switch, not a requirement: a value this build (or this plan's n_fft) has no kernel for is reported and ignored --
        // an environment left over from an experiments build must not make every plan_create fail
        if (dsp_mfcc_plan_set_kernel(p, std::atoi(k)) != DSP_OK)
            std::fprintf(stderr, "libdsp_amd: DSP_AMD_KERNEL=%s ignored (%s); using the default kernel\n", k, dsp_last_error());
    }
    *out = p;
    return DSP_OK;
}

void dsp_mfcc_plan_destroy(dsp_mfcc_plan *p)
{
    if (!p) return;
    dsp::DeviceScope dsp_device_scope_(p->device);
    if (p->d_tables) hipFree(p->d_tables);
    if (p->d_row_tables) hipFree(p->d_row_tables);
    if (p->d_gen_tables) hipFree(p->d_gen_tables);
    if (p->d_tables2048) hipFree(p->d_tables2048);
    if (p->d_pair) hipFree(p->d_pair);
    if (p->d_scan) hipFree(p->d_scan);
    if (p->d_filtered) hipFree(p->d_filtered);
    if (p->d_frame_max) hipFree(p->d_frame_max);
    if (p->d_clip_floor) hipFree(p->d_clip_floor);
    if (p->d_in) hipFree(p->d_in);
    if (p->d_out) hipFree(p->d_out);
    p->spans.release();
    delete p;
}

int dsp_mfcc_plan_config(const dsp_mfcc_plan *p, dsp_mfcc_config *cfg)
{
    if (!p || !cfg) return fail(DSP_EINVAL, "plan/cfg is NULL");
    *cfg = p->cfg;
    return DSP_OK;
}

int dsp_mfcc_plan_set_kernel(dsp_mfcc_plan *p, int kernel)
{
    if (!p || (kernel != DSP_KERNEL_WAVE && kernel != DSP_KERNEL_ROW && kernel != DSP_KERNEL_WAVE_FRAME && kernel != DSP_KERNEL_PAIR)) return fail(DSP_EINVAL, "bad kernel id");
#ifndef DSP_AMD_EXPERIMENTS
    // DSP_KERNEL_ROW on a 1024-point plan selects the general Stockham kernel (a product path: the fallback for filterbanks
    // the wave kernel's tables do not hold); the 512-point row / pair kernels are experiments outside the default build
    if (kernel == DSP_KERNEL_PAIR || (kernel == DSP_KERNEL_ROW && p->cfg.n_fft != 1024))
        return fail(DSP_EINVAL, "DSP_KERNEL_ROW / DSP_KERNEL_PAIR (512-point experiments) are not in this build: rebuild with DSP_AMD_EXPERIMENTS=1");
#endif
    p->kernel = kernel;
    return DSP_OK;
}

int dsp_mfcc_plan_set_launch(dsp_mfcc_plan *p, int blocks_per_cu, int frames_per_chunk)
{
    if (!p || blocks_per_cu < 0 || frames_per_chunk < 0) return fail(DSP_EINVAL, "bad launch knobs");
    p->blocks_per_cu = blocks_per_cu;
    p->chunk = frames_per_chunk;
    return DSP_OK;
}

static int reserve(float **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return DSP_OK;
    if (*buf) { hipFree(*buf); *buf = nullptr; *cap = 0; }
    DSP_HIP(hipMalloc(buf, need));
    *cap = need;
    return DSP_OK;
}

static int run(dsp_mfcc_plan *p, const void *d_in, float *d_out, long n_frames, int frames_per_clip,
               long clip_stride, void *stream, int in_kind = 0, bool fused_prefilter = false, int samples_per_clip = 0)
{
    if (n_frames == 0) return DSP_OK;
    DSP_ON_DEVICE(p->device);       // the caller's current device may be another GPU: tables and workspaces live on the plan's
    const bool single_clip = frames_per_clip > 0 && n_frames == frames_per_clip;   // stride unused
    if ((reinterpret_cast<uintptr_t>(d_in) & (in_kind == 1 ? 3 : 7)) || (!single_clip && (clip_stride & 1)))
        return fail(DSP_EINVAL, "input must be 8-byte aligned (4 for mono int16) with an even clip stride");
    const bool aub2048 = p->cfg.n_fft == 2048 && (p->cfg.spectrum != DSP_SPECTRUM_POWER || p->cfg.log_mode == DSP_LOG_LOG10_FLOOR || p->cfg.framing == DSP_FRAMING_STREAM);
    if (in_kind != 0 && !(aub2048 && frames_per_clip > 0) && (p->cfg.n_fft != 512 || p->kernel != DSP_KERNEL_WAVE || p->cfg.log_mode != DSP_LOG_PER_FRAME_MAX))
        return fail(DSP_EINVAL, "PCM16 ingestion runs on the 512-point wave-per-frame kernel (per-frame log mode) and on the 2048-point scrubjay_infer.c front end");
    if (p->cfg.n_fft == 2048) {
        dsp::Mfcc512Args a{};
        a.in = d_in; a.in_kind = in_kind; a.out = d_out; a.n_frames = n_frames; a.clip_stride = clip_stride; a.frames_per_clip = frames_per_clip;
        a.hop = p->cfg.hop_length; a.frame_len = p->cfg.frame_length; a.chunk = p->chunk > 0 ? p->chunk : 8;
        a.n_mels = p->cfg.n_mels; a.n_mfcc = p->cfg.n_mfcc; a.amin = p->cfg.amin; a.top_db = p->cfg.top_db;
        a.spectrum = p->cfg.spectrum;
        a.stream_framing = frames_per_clip > 0 && p->cfg.framing == DSP_FRAMING_STREAM;
        a.samples_per_clip = samples_per_clip;
        if (a.stream_framing && samples_per_clip <= 0) return fail(DSP_EINVAL, "internal: stream framing without the clip length");
        const int per_cu = p->blocks_per_cu > 0 ? p->blocks_per_cu : p->resident_blocks_2048;
        const long chunks = (n_frames + a.chunk - 1) / a.chunk;
        const long blocks = std::max(1L, std::min((long)p->n_cu * per_cu, (chunks + 3) / 4));
        a.log_mode = p->cfg.log_mode;
        if (a.log_mode == DSP_LOG_GLOBAL_REF1 && frames_per_clip > 0) {
            // clip-global top_db, as for n_fft = 512 below: pass 1 writes each frame's maximum, a tiny kernel turns them into one
            // floor per clip, pass 2 is the normal kernel clipping at that floor
            const long n_clips = n_frames / frames_per_clip;
            int rc;
            std::lock_guard<std::recursive_mutex> lock(p->mu);
            if ((rc = reserve(&p->d_frame_max, &p->frame_max_cap, (size_t)n_frames * sizeof(float))) < 0) return rc;
            if ((rc = reserve(&p->d_clip_floor, &p->clip_floor_cap, (size_t)n_clips * sizeof(float))) < 0) return rc;
            a.frame_max = p->d_frame_max;
            DSP_HIP(dsp::launch_mfcc2048(a, p->d_tables2048, (int)blocks, (hipStream_t)stream, false));
            DSP_HIP(dsp::launch_clip_floor(p->d_frame_max, n_clips, frames_per_clip, a.top_db, p->d_clip_floor, (hipStream_t)stream));
            a.frame_max = nullptr;
            a.clip_floor = p->d_clip_floor;
        }
        DSP_HIP(dsp::launch_mfcc2048(a, p->d_tables2048, (int)blocks, (hipStream_t)stream, false));
        return DSP_OK;
    }
    dsp::Mfcc512Args a;
    a.in = d_in;
    a.in_kind = in_kind;
    a.out = d_out;
    a.tables = p->d_tables;
    a.n_frames = n_frames;
    a.clip_stride = clip_stride;
    a.frames_per_clip = frames_per_clip;
    a.hop = p->cfg.hop_length;
    a.frame_len = p->cfg.frame_length;
    const bool gen = p->cfg.n_fft == 1024;
#ifdef DSP_AMD_EXPERIMENTS
    const bool row = !gen && p->kernel == DSP_KERNEL_ROW && p->cfg.n_fft == 512;
    // two frames per wavefront step (experiment): the reference shape on independent full frames only, else the default form
    const bool pair = !gen && p->kernel == DSP_KERNEL_PAIR && p->d_pair && in_kind == 0 && frames_per_clip == 0 && p->cfg.frame_length == 512 &&
                      p->cfg.log_mode == DSP_LOG_PER_FRAME_MAX && p->host.dct_split == 4 && p->host.dct_len == 10 && p->host.mel_gather == 3;
#else
    const bool row = false, pair = false;
#endif
    // 16-frame tile epilogue: per-frame log mode on the wave-per-frame kernel
    const bool tile = !gen && (p->kernel == DSP_KERNEL_WAVE || p->kernel == DSP_KERNEL_PAIR) && p->cfg.log_mode == DSP_LOG_PER_FRAME_MAX;
    // 1024-point: the register-resident wave kernel when the filterbank fits two chunk slots per lane (DSP_KERNEL_ROW selects
    // the general Stockham kernel for A/B)
    const bool gen_wave = gen && p->gen_slots <= 3 && p->kernel != DSP_KERNEL_ROW;
    const int nf = gen ? (gen_wave ? 8 : 1) : (pair ? 16 : (row ? 4 : (tile ? 8 : 1)));
    a.chunk = p->chunk > 0 ? p->chunk : (pair ? 16 : 8);
    a.chunk = ((a.chunk + nf - 1) / nf) * nf;   // whole items (tile: half-tiles of 8 frames) per chunk
    a.n_mels = p->cfg.n_mels;
    a.n_mfcc = p->cfg.n_mfcc;
    a.amin = p->cfg.amin;
    a.top_db = p->cfg.top_db;
    a.log_mode = p->cfg.log_mode;
    a.frame_max = nullptr;
    a.clip_floor = nullptr;
    // persistent-style grid: exactly the 4-wave blocks the chip holds at once (one
    // extra block per CU would run as a second, mostly idle round: measured +14 %),
    // never more blocks than there are chunks of work
    const int per_cu = p->blocks_per_cu > 0 ? p->blocks_per_cu
                       : (gen ? (gen_wave ? (fused_prefilter ? p->resident_blocks_gen_pre : p->resident_blocks_gen_wave) : p->resident_blocks_gen)
                              : (pair ? p->resident_blocks_pair : (row ? p->resident_blocks_row : (tile ? p->resident_blocks : p->resident_blocks_frame))));
    long blocks = (long)p->n_cu * per_cu;
    const long chunks = (n_frames + a.chunk - 1) / a.chunk;
    blocks = std::max(1L, std::min(blocks, (chunks + 3) / 4));
    if (a.log_mode == DSP_LOG_GLOBAL_REF1 && frames_per_clip > 0) {
        // clip-global top_db: pass 1 writes each frame's maximum, a tiny kernel turns them into one
        // floor per clip, pass 2 is the normal kernel clipping at that floor
        const long n_clips = n_frames / frames_per_clip;
        int rc;
        std::lock_guard<std::recursive_mutex> lock(p->mu);
        if ((rc = reserve(&p->d_frame_max, &p->frame_max_cap, (size_t)n_frames * sizeof(float))) < 0) return rc;
        if ((rc = reserve(&p->d_clip_floor, &p->clip_floor_cap, (size_t)n_clips * sizeof(float))) < 0) return rc;
        a.frame_max = p->d_frame_max;
        DSP_HIP(dsp::launch_mfcc512(a, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks, (hipStream_t)stream, false));
        DSP_HIP(dsp::launch_clip_floor(p->d_frame_max, n_clips, frames_per_clip, a.top_db, p->d_clip_floor, (hipStream_t)stream));
        a.frame_max = nullptr;
        a.clip_floor = p->d_clip_floor;
        DSP_HIP(dsp::launch_mfcc512(a, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks, (hipStream_t)stream, false));
        return DSP_OK;
    }
    if (a.log_mode == DSP_LOG_GLOBAL_REF1) {      // independent frames: one pass, wave-per-frame kernel only
        DSP_HIP(dsp::launch_mfcc512(a, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks, (hipStream_t)stream, false));
        return DSP_OK;
    }
    if (fused_prefilter && !(gen_wave && p->d_scan)) return fail(DSP_EINVAL, "internal: fused prefilter without its tables");
    if (gen_wave)
        DSP_HIP(dsp::launch_mfcc1024_wave(a, p->d_gen_tables, (int)blocks, (hipStream_t)stream, fused_prefilter ? p->d_scan : nullptr, p->scan_steps));
    else if (gen)
        DSP_HIP(dsp::launch_mfcc1024(a, p->d_gen_tables, (int)blocks, (hipStream_t)stream));
#ifdef DSP_AMD_EXPERIMENTS
    else if (pair)
        DSP_HIP(dsp::launch_mfcc512_pair(a, p->d_pair, (int)blocks, (hipStream_t)stream));
    else if (row)
        DSP_HIP(dsp::launch_mfcc512_row(a, p->d_row_tables, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks,
                                        (hipStream_t)stream));
#endif
    else
        DSP_HIP(dsp::launch_mfcc512(a, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks, (hipStream_t)stream, tile));
    return DSP_OK;
}

int dsp_mfcc_frames_device(dsp_mfcc_plan *p, const float *d_frames, long n_frames, float *d_out, void *stream)
{
    if (!p || n_frames < 0 || (n_frames > 0 && (!d_frames || !d_out))) return fail(DSP_EINVAL, "bad argument");
    if (p->cfg.prefilter == DSP_PREFILTER_NONE) return run(p, d_frames, d_out, n_frames, 0, 0, stream);
    // BASELINE config 3: 8th-order Butterworth (donut-classifier/classifier.c:420-446, float64) over each
    // frame from zero state, rounded to float, then the MFCC chain.
    // One pass (1024-sample frames on the wave kernel, 16-byte aligned input): the filter runs inside the MFCC kernel as a
    // float64 parallel-form scan over the wave's lanes -- the frame is read once, nothing filtered is written.  This entry
    // point is tolerance-gated (1e-4 of the frame's L-inf norm); the scan equals the serial recurrence to ~1e-13 before the
    // rounding to float.  dsp_butter_bandpass_filter_* keep the bit-exact serial recurrence.
    if (p->d_scan && p->kernel != DSP_KERNEL_ROW && (reinterpret_cast<uintptr_t>(d_frames) & 15) == 0 && !std::getenv("DSP_AMD_PREFILTER_TWO_PASS"))
        return run(p, d_frames, d_out, n_frames, 0, 0, stream, 0, true);
    // Otherwise two passes: filtered frames go through a bounded workspace (sub-batches of <= 1 Mi frames).
    std::lock_guard<std::recursive_mutex> lock(p->mu);
    DSP_ON_DEVICE(p->device);
    const int fl = p->cfg.frame_length;
    const long sub = std::min<long>(n_frames, 1L << 20);
    int rc;
    if ((rc = reserve(&p->d_filtered, &p->filtered_cap, (size_t)sub * fl * sizeof(float))) < 0) return rc;
    dsp::IirCoefD c;
    dsp_butter_bandpass(p->cfg.prefilter == DSP_PREFILTER_BUTTER_1000_3000 ? 1000 : 3000,
                        p->cfg.prefilter == DSP_PREFILTER_BUTTER_1000_3000 ? 3000 : 7500, c.b, c.a);
    for (long f0 = 0; f0 < n_frames; f0 += sub) {
        const long cnt = std::min(sub, n_frames - f0);
        DSP_HIP(dsp::launch_iir_f64_on_f32(d_frames + f0 * fl, cnt, fl, fl, c, p->d_filtered, (hipStream_t)stream));
        if ((rc = run(p, p->d_filtered, d_out + f0 * p->cfg.n_mfcc, cnt, 0, 0, stream)) < 0) return rc;
    }
    return DSP_OK;
}

int dsp_mfcc_clips_device(dsp_mfcc_plan *p, const float *d_signal, long n_clips, int samples_per_clip,
                          long clip_stride, float *d_out, int max_frames, void *stream)
{
    if (!p || n_clips < 0) return fail(DSP_EINVAL, "bad argument");
    if (p->cfg.prefilter != DSP_PREFILTER_NONE) return fail(DSP_EINVAL, "the per-frame prefilter applies to independent frames only");
    const int t = dsp_mfcc_frames_for(&p->cfg, samples_per_clip, max_frames);
    if (t == 0 || n_clips == 0) return 0;
    if (!d_signal || !d_out) return fail(DSP_EINVAL, "NULL buffer");
    if (n_clips > 1 && clip_stride < samples_per_clip) return fail(DSP_EINVAL, "clip_stride < samples_per_clip");
    const int rc = run(p, d_signal, d_out, n_clips * (long)t, t, clip_stride, stream, 0, false, samples_per_clip);
    return rc < 0 ? rc : t;
}

int dsp_mfcc_clips_pcm16_device(dsp_mfcc_plan *p, const int16_t *d_pcm, long n_clips, int samples_per_clip,
                                long clip_stride, int channels, int stereo_mode, float *d_out, int max_frames, void *stream)
{
    if (!p || n_clips < 0 || (channels != 1 && channels != 2)) return fail(DSP_EINVAL, "bad argument");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return fail(DSP_EINVAL, "bad stereo_mode");
    if (p->cfg.prefilter != DSP_PREFILTER_NONE) return fail(DSP_EINVAL, "the per-frame prefilter applies to independent float frames only");
    const int t = dsp_mfcc_frames_for(&p->cfg, samples_per_clip, max_frames);
    if (t == 0 || n_clips == 0) return 0;
    if (!d_pcm || !d_out) return fail(DSP_EINVAL, "NULL buffer");
    if (n_clips > 1 && clip_stride < samples_per_clip) return fail(DSP_EINVAL, "clip_stride < samples_per_clip");
    const int kind = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
    const int rc = run(p, d_pcm, d_out, n_clips * (long)t, t, clip_stride, stream, kind, false, samples_per_clip);
    return rc < 0 ? rc : t;
}


int dsp_mfcc_frames_host(dsp_mfcc_plan *p, const float *frames, long n_frames, float *out)
{
    if (!p || n_frames < 0 || (n_frames > 0 && (!frames || !out))) return fail(DSP_EINVAL, "bad argument");
    if (n_frames == 0) return DSP_OK;
    std::lock_guard<std::recursive_mutex> lock(p->mu);
    DSP_ON_DEVICE(p->device);
    const size_t in_b = (size_t)n_frames * p->cfg.frame_length * sizeof(float);
    const size_t out_b = (size_t)n_frames * p->cfg.n_mfcc * sizeof(float);
    int rc;
    if ((rc = reserve(&p->d_in, &p->in_cap, in_b)) < 0) return rc;
    if ((rc = reserve(&p->d_out, &p->out_cap, out_b)) < 0) return rc;
    DSP_HIP(hipMemcpyAsync(p->d_in, frames, in_b, hipMemcpyHostToDevice, nullptr));
    if (p->cfg.prefilter != DSP_PREFILTER_NONE) return fail(DSP_EINVAL, "prefiltered plans take device buffers (dsp_mfcc_frames_device)");
    if ((rc = run(p, p->d_in, p->d_out, n_frames, 0, 0, nullptr)) < 0) return rc;
    DSP_HIP(hipMemcpyAsync(out, p->d_out, out_b, hipMemcpyDeviceToHost, nullptr));
    DSP_HIP(hipStreamSynchronize(nullptr));
    return DSP_OK;
}

int dsp_mfcc_clips_host(dsp_mfcc_plan *p, const float *signal, long n_clips, int samples_per_clip,
                        long clip_stride, float *out, int max_frames)
{
    if (!p || n_clips < 0) return fail(DSP_EINVAL, "bad argument");
    const int t = dsp_mfcc_frames_for(&p->cfg, samples_per_clip, max_frames);
    if (t == 0 || n_clips == 0) return 0;
    if (!signal || !out) return fail(DSP_EINVAL, "NULL buffer");
    if (n_clips > 1 && clip_stride < samples_per_clip) return fail(DSP_EINVAL, "clip_stride < samples_per_clip");
    std::lock_guard<std::recursive_mutex> lock(p->mu);
    DSP_ON_DEVICE(p->device);
    // device copy is packed with an even stride so every frame start stays 8-byte aligned
    const long dstride = samples_per_clip + (samples_per_clip & 1);
    const size_t in_b = (size_t)n_clips * dstride * sizeof(float);
    const size_t out_b = (size_t)n_clips * t * p->cfg.n_mfcc * sizeof(float);
    int rc;
    if ((rc = reserve(&p->d_in, &p->in_cap, in_b)) < 0) return rc;
    if ((rc = reserve(&p->d_out, &p->out_cap, out_b)) < 0) return rc;
    DSP_HIP(hipMemcpy2DAsync(p->d_in, dstride * sizeof(float), signal, clip_stride * sizeof(float),
                             (size_t)samples_per_clip * sizeof(float), (size_t)n_clips, hipMemcpyHostToDevice, nullptr));
    if ((rc = run(p, p->d_in, p->d_out, n_clips * (long)t, t, dstride, nullptr, 0, false, samples_per_clip)) < 0) return rc;
    DSP_HIP(hipMemcpyAsync(out, p->d_out, out_b, hipMemcpyDeviceToHost, nullptr));
    DSP_HIP(hipStreamSynchronize(nullptr));
    return t;
}

}  // extern "C"

// ---- donut classifier path ------------------------------------------------------------

namespace {

struct ClassifyCtx {
    int device = -1;
    dsp::SpecTables *d_tab = nullptr;
    // workspace for one sub-batch
    float *d_x = nullptr, *d_sbp = nullptr;                // staged input (host entry points), 3000-7500 Hz PSD maps [clip][T][129]
    float *d_ck_bp = nullptr, *d_ck_mp = nullptr;          // [clip][T][kCkPerSegBp / Mp][8]: delay line of each filter at every segment start (3000-7500 Hz: and middle)
    float *d_mean_mp = nullptr;                            // [clip][T]: segment means of the 1000-3000 Hz output
    int *d_labels = nullptr, *d_hits = nullptr;            // d_hits: work list of clips with midpoints
    int *d_loud = nullptr;                                 // [clip][T]: time bins of the 1000-3000 Hz map above 70 dB; after the midpoints kernel: rows of the 3000-7500 Hz map the band sums read
    unsigned *d_minmax = nullptr;                          // [clip][2]: float bits of the smallest / largest positive cell of the 3000-7500 Hz map
    int *d_simd = nullptr;                                 // iir2_ckpt_kernel's per-CU SIMD load table (launch_iir2_ckpt)
    int *d_gate = nullptr;                                 // work list of the segments whose energy does not rule a loud cell out (IIR kernel)
    dsp::ClassifyTrace *d_trace = nullptr;
    long cap_clips = 0;                                    // per-clip arrays (labels, hits, trace, minmax)
    long cap_segs = 0;                                     // per-segment arrays: clips x segments per clip of the largest pass so far
    long cap_x_floats = 0;                                 // staging buffer of the host entry points, in floats (0: none)
    float keep_min_db = 70.0f;                             // the midpoint threshold d_tab->mp_keep_min was computed for
    bool gate_ok = false;                                  // SpecTables::gate_ok of d_tab
    std::mutex mu;
    hipEvent_t done = nullptr;                             // recorded behind the last call's work: the workspace is free once it has fired
    bool pending = false;
    dsp::SpanRing spans;                                   // ragged batches: the clips' spans on their way to the GPU (capi_util.hpp)
    void wait_idle()
    {
        if (pending && done) (void)hipEventSynchronize(done);
        pending = false;
    }
};
// One context (tables + grow-only workspace + mutex) PER DEVICE: threads that drive different GPUs from one process share nothing, and
// a device entry point returns once its work is enqueued -- the next call on that device makes its stream wait for the event this one
// leaves behind.  (Round 3: one process-wide context that moved between GPUs and blocked until the stream was idle.)
constexpr int kMaxDevices = 64;
ClassifyCtx g_cls_ctx[kMaxDevices];

// leaves the "workspace busy until here" event behind the call's work on every exit
struct ClsBusyMark {
    ClassifyCtx &c;
    hipStream_t st;
    ~ClsBusyMark()
    {
        if (c.done && hipEventRecord(c.done, st) == hipSuccess) c.pending = true;
        else { (void)hipGetLastError(); (void)hipStreamSynchronize(st); c.pending = false; }
    }
};

// 957 columns = 13.4 s: a midpoint needs a cluster of >= 12 columns (0.15 s at 14 ms per column) followed by a gap of >= 4
// (0.05 s), so 64 * 15 - 3 columns cannot hold more than the kMaxMidpoints = 64 a trace record has room for
constexpr int kMaxSpecColumns = 957;

bool valid_classify_cfg(const dsp_classify_config &c)
{
    auto fin = [](float v) { return v == v && v - v == 0.0f; };
    return fin(c.keep_lo) && fin(c.keep_hi) && fin(c.midpoint_db) && fin(c.middle_max) && fin(c.above_min) && fin(c.below_min) &&
           c.keep_lo < c.keep_hi;
}

dsp_classify_config default_classify_cfg()
{
    // sync/lib/classifier.cpp:67-68 (0.65 / 0.80), :436 (70 dB), :109 (100 / 200 / 80)
    return dsp_classify_config{0.65f, 0.80f, 70.0f, 100.0f, 200.0f, 80.0f};
}

static_assert(sizeof(dsp::ClassifyTrace) == sizeof(dsp_classify_trace), "trace layouts must match");

// the device of the host entry points: DSP_AMD_DEVICE or 0
int cls_host_device(int &device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback");
    const char *dev = std::getenv("DSP_AMD_DEVICE");
    device = dev ? std::atoi(dev) : 0;
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(DSP_EINVAL, "device index out of range");
    return DSP_OK;
}

// tables of the context of `device` (the caller holds g_cls.mu and has made the device current)
int cls_init(ClassifyCtx &g_cls, int device)
{
    if (g_cls.d_tab) return DSP_OK;
    g_cls.device = device;
    if (!g_cls.done) DSP_HIP(hipEventCreateWithFlags(&g_cls.done, hipEventDisableTiming));
    dsp::SpecTables t;
    dsp::build_spec_tables(16000, t);
    DSP_HIP(hipMalloc(&g_cls.d_tab, sizeof(t)));
    DSP_HIP(hipMemcpy(g_cls.d_tab, &t, sizeof(t), hipMemcpyHostToDevice));
    g_cls.gate_ok = t.gate_ok != 0;
    g_cls.keep_min_db = 70.0f;
    DSP_HIP(dsp::launch_spec_threshold(g_cls.d_tab, g_cls.keep_min_db, nullptr));
    {   // the recompute kernel's three-instruction PSD division is switched on only after it has been checked against the real
        // division on every float of its range, for this table's U, on this device (~1e9 values: a fraction of a millisecond)
        unsigned long long *d_bad = nullptr, bad = 1;
        DSP_HIP(hipMalloc(&d_bad, sizeof(bad)));
        hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(bad), nullptr);
        if (e == hipSuccess) e = dsp::launch_spec_div_verify(g_cls.d_tab, d_bad, nullptr);
        if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
        hipFree(d_bad);
        if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
        const int on = bad == 0 && !std::getenv("DSP_AMD_SPEC_EXACT_DIV") ? 1 : 0;
        DSP_HIP(hipMemcpy(reinterpret_cast<char *>(g_cls.d_tab) + offsetof(dsp::SpecTables, div_fast), &on, sizeof(on), hipMemcpyHostToDevice));
    }
    DSP_HIP(hipStreamSynchronize(nullptr));
    return DSP_OK;
}

int spec_bins(int n) { return n < dsp::kSpecSeg ? 0 : (n - dsp::kSpecSeg) / dsp::kSpecHop + 1; }
long cls_row(int n) { return ((long)n + 3) & ~3L; }      // workspace row: n floats rounded up to 16 bytes

void cls_free_workspace(ClassifyCtx &g_cls)
{
    for (void *p : {(void *)g_cls.d_x, (void *)g_cls.d_sbp, (void *)g_cls.d_ck_bp, (void *)g_cls.d_ck_mp, (void *)g_cls.d_loud, (void *)g_cls.d_gate, (void *)g_cls.d_simd,
                    (void *)g_cls.d_mean_mp, (void *)g_cls.d_labels, (void *)g_cls.d_hits, (void *)g_cls.d_trace, (void *)g_cls.d_minmax})
        if (p) hipFree(p);
    g_cls.d_x = g_cls.d_sbp = g_cls.d_ck_bp = g_cls.d_ck_mp = g_cls.d_mean_mp = nullptr;
    g_cls.d_labels = g_cls.d_hits = g_cls.d_loud = g_cls.d_gate = g_cls.d_simd = nullptr; g_cls.d_trace = nullptr; g_cls.d_minmax = nullptr;
    g_cls.cap_clips = 0; g_cls.cap_segs = 0; g_cls.cap_x_floats = 0;
}

void cls_release(ClassifyCtx &g_cls)      // (on g_cls.device, made current by the caller)
{
    g_cls.wait_idle();
    cls_free_workspace(g_cls);
    if (g_cls.d_tab) hipFree(g_cls.d_tab);
    g_cls.d_tab = nullptr;
    if (g_cls.done) (void)hipEventDestroy(g_cls.done);
    g_cls.done = nullptr;
    g_cls.spans.release();
    g_cls.device = -1;
}

// workspace of one sub-batch; need_x: also a staging buffer for the clips themselves (host entry points).  The per-segment arrays
// are [clip][T(n)] with the pass's own T, so what a pass needs of them is its PRODUCT clips x T: a ragged batch's pass of few long
// clips and its pass of many short ones share one allocation (sized by each dimension's maximum it would be their outer product --
// 32 GB of maps for 65 536 clips of which one is 13 s long).
int cls_reserve(ClassifyCtx &g_cls, long clips, int n, bool need_x)
{
    const long T = std::max(1, spec_bins(n));
    const long x_floats = need_x ? clips * cls_row(n) : 0;
    if (clips <= g_cls.cap_clips && clips * T <= g_cls.cap_segs && x_floats <= g_cls.cap_x_floats) return DSP_OK;
    g_cls.wait_idle();
    const long rows = std::max(clips, g_cls.cap_clips), segs = std::max(clips * T, g_cls.cap_segs), xf = std::max(x_floats, g_cls.cap_x_floats);
    cls_free_workspace(g_cls);
    if (xf > 0) DSP_HIP(hipMalloc(&g_cls.d_x, (size_t)xf * sizeof(float)));      // (staged int16 rows are at most as long)
    DSP_HIP(hipMalloc(&g_cls.d_sbp, (size_t)segs * dsp::kSpecBins * sizeof(float)));
    DSP_HIP(hipMalloc(&g_cls.d_ck_bp, (size_t)segs * dsp::kCkPerSegBp * 8 * sizeof(float)));
    DSP_HIP(hipMalloc(&g_cls.d_ck_mp, (size_t)segs * dsp::kCkPerSegMp * 8 * sizeof(float)));
    DSP_HIP(hipMalloc(&g_cls.d_loud, (size_t)segs * sizeof(int)));
    DSP_HIP(hipMalloc(&g_cls.d_minmax, (size_t)rows * 2 * sizeof(unsigned)));
    DSP_HIP(hipMalloc(&g_cls.d_simd, sizeof(int) * dsp::kSimdLoadCus * dsp::kSimdLoadStride));
    DSP_HIP(hipMalloc(&g_cls.d_gate, ((size_t)segs + 1) * sizeof(int)));      // work list of gated-in frames: count + frame numbers
    DSP_HIP(hipMalloc(&g_cls.d_mean_mp, (size_t)segs * sizeof(float)));
    DSP_HIP(hipMalloc(&g_cls.d_labels, (size_t)rows * sizeof(int)));
    DSP_HIP(hipMalloc(&g_cls.d_hits, (size_t)(rows + 1) * sizeof(int)));
    DSP_HIP(hipMalloc(&g_cls.d_trace, (size_t)rows * sizeof(dsp::ClassifyTrace)));
    g_cls.cap_clips = rows; g_cls.cap_segs = segs; g_cls.cap_x_floats = xf;
    return DSP_OK;
}

dsp::IirCoef coef_f32(double lo, double hi)
{
    double b[9], a[9];
    dsp_butter_bandpass(lo, hi, b, a);
    dsp::IirCoef c;
    for (int i = 0; i < 9; ++i) { c.b[i] = (float)b[i]; c.a[i] = (float)a[i]; }   // classifier.cpp:140-183: float literals
    return c;
}

// one sub-batch already resident at d_x (input kind `in`: 0 float, 1 / 2 / 3 int16 mono / stereo channel 0 / stereo average; row
// stride in samples per channel): labels (+ trace) into the workspace
// spans != nullptr: a ragged sub-batch (clip c at spans[c].off samples from d_x with spans[c].frames whole segments; n = the longest
// clip of the BATCH, total = samples in the buffer)
int cls_run(ClassifyCtx &g_cls, const dsp_classify_config &cfg, const void *d_x, int in, long clips, int n, long stride, hipStream_t st, bool want_trace,
            const dsp::ClipSpan *spans = nullptr, long total = 0)
{
    const dsp::IirCoef bp = coef_f32(3000, 7500), mp = coef_f32(1000, 3000);   // classifier.cpp:14-19, 438-442
    if (cfg.midpoint_db != g_cls.keep_min_db) {      // the table's threshold PSD value follows the configured dB threshold
        // (earlier calls on this context are ordered before this one by its event)
        DSP_HIP(dsp::launch_spec_threshold(g_cls.d_tab, cfg.midpoint_db, st));
        g_cls.keep_min_db = cfg.midpoint_db;
    }
    const dsp::ClassifyRule rule{cfg.keep_lo, cfg.keep_hi, cfg.middle_max, cfg.above_min, cfg.below_min};
    // ONE pass over the clips: both recurrences, the delay lines at every segment start, the 1000-3000 Hz segment means and
    // the energy gate.  No filtered signal is written; the spectrogram kernels recompute the segments they transform.
    DSP_HIP(dsp::launch_iir2_ckpt(d_x, clips, n, stride, bp, mp, g_cls.d_ck_bp, g_cls.d_ck_mp, g_cls.d_mean_mp, g_cls.d_gate, g_cls.d_tab, st, g_cls.d_simd, in, spans, total));
    // midpoints first (1000-3000 Hz map, as flags, gated frames only); the 3000-7500 Hz spectrogram and its band sums only for
    // clips that have midpoints
    DSP_HIP(dsp::launch_spec_from_ckpt(d_x, clips, n, stride, mp, g_cls.d_ck_mp, g_cls.d_mean_mp, g_cls.d_gate, nullptr, g_cls.d_tab,
                                       reinterpret_cast<float *>(g_cls.d_loud), true, st, nullptr, nullptr, in, spans));
    // DSP_AMD_CLASSIFY_FULL_MAPS=1: every row of the listed clips' maps is stored and read (the form before the need / minmax hand-over)
    static const bool full_maps = [] { const char *e = std::getenv("DSP_AMD_CLASSIFY_FULL_MAPS"); return e && std::atoi(e) != 0; }();
    unsigned *mm = full_maps ? nullptr : g_cls.d_minmax;
    const int *need = full_maps ? nullptr : g_cls.d_loud;
    DSP_HIP(dsp::launch_classify_midpoints(g_cls.d_loud, clips, n, 16000, g_cls.d_labels, g_cls.d_trace, g_cls.d_hits, st, want_trace, mm, spans));
    DSP_HIP(dsp::launch_spec_from_ckpt(d_x, clips, n, stride, bp, g_cls.d_ck_bp, nullptr, nullptr, g_cls.d_hits, g_cls.d_tab, g_cls.d_sbp, false, st, need, mm, in, spans));
    DSP_HIP(dsp::launch_classify_bands(g_cls.d_sbp, clips, n, 16000, g_cls.d_labels, g_cls.d_trace, g_cls.d_hits, st, rule, need, mm, spans));
    return DSP_OK;
}

// clips per pass through the workspace (42 KB per 1 s clip): 1024 blocks of 64 clips = the four IIR blocks a CU holds
constexpr long kClsSubBatch = 65536;

}  // namespace

extern "C" {

int dsp_classify_division_check(long long *mismatches)
{
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    unsigned long long *d_bad = nullptr, bad = 0;
    int on = 0;
    DSP_HIP(hipMalloc(&d_bad, sizeof(bad)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(bad), nullptr);
    if (e == hipSuccess) e = dsp::launch_spec_div_verify(g_cls.d_tab, d_bad, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&on, reinterpret_cast<const char *>(g_cls.d_tab) + offsetof(dsp::SpecTables, div_fast), sizeof(on), hipMemcpyDeviceToHost);
    hipFree(d_bad);
    if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
    if (mismatches) *mismatches = (long long)bad;
    return on ? 1 : 0;
}

int dsp_butter_bandpass_filter_f32(const float *data, long n_clips, int n, long stride, const float *b,
                                   const float *a, float *output)
{
    if (!data || !output || !b || !a || n_clips < 0 || n < 0 || (n_clips > 1 && stride < n)) return fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0 || n == 0) return DSP_OK;
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    float *dx = nullptr, *dy = nullptr;
    const size_t bytes = (size_t)n_clips * n * sizeof(float);
    DSP_HIP(hipMalloc(&dx, bytes));
    if (hipMalloc(&dy, bytes) != hipSuccess) { hipFree(dx); return fail(DSP_ENOMEM, "hipMalloc"); }
    dsp::IirCoef c;
    for (int i = 0; i < 9; ++i) { c.b[i] = b[i]; c.a[i] = a[i]; }
    hipError_t e = hipMemcpy2D(dx, (size_t)n * sizeof(float), data, (size_t)stride * sizeof(float), (size_t)n * sizeof(float), n_clips, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_iir_f32(dx, n_clips, n, n, c, dy, c, nullptr, nullptr);
    if (e == hipSuccess) e = hipMemcpy2D(output, (size_t)stride * sizeof(float), dy, (size_t)n * sizeof(float), (size_t)n * sizeof(float), n_clips, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dy);
    if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
    return DSP_OK;
}

int dsp_butter_bandpass_filter_f64(const double *data, long n_clips, int n, long stride, const double *b,
                                   const double *a, double *output)
{
    if (!data || !output || !b || !a || n_clips < 0 || n < 0 || (n_clips > 1 && stride < n)) return fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0 || n == 0) return DSP_OK;
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    double *dx = nullptr, *dy = nullptr;
    const size_t bytes = (size_t)n_clips * n * sizeof(double);
    DSP_HIP(hipMalloc(&dx, bytes));
    if (hipMalloc(&dy, bytes) != hipSuccess) { hipFree(dx); return fail(DSP_ENOMEM, "hipMalloc"); }
    dsp::IirCoefD c;
    for (int i = 0; i < 9; ++i) { c.b[i] = b[i]; c.a[i] = a[i]; }
    hipError_t e = hipMemcpy2D(dx, (size_t)n * sizeof(double), data, (size_t)stride * sizeof(double), (size_t)n * sizeof(double), n_clips, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_iir_f64(dx, n_clips, n, n, c, dy, nullptr);
    if (e == hipSuccess) e = hipMemcpy2D(output, (size_t)stride * sizeof(double), dy, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n_clips, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dy);
    if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
    return DSP_OK;
}

int dsp_compute_spectrogram_f32(const float *signal, int n, int fs, float *frequencies, float *times, float *sxx)
{
    if (!signal || !sxx || n < 0 || fs <= 0) return fail(DSP_EINVAL, "bad argument");
    const int T = spec_bins(n);
    if (frequencies)
        for (int k = 0; k < dsp::kSpecBins; ++k) frequencies[k] = (float)k * (float)fs / (float)dsp::kSpecSeg;   // classifier.cpp:248-251
    if (times)
        for (int t = 0; t < T; ++t) times[t] = ((float)(t * dsp::kSpecHop + dsp::kSpecSeg / 2)) / (float)fs;     // :254-258
    if (T == 0) return 0;
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    float *dx = nullptr, *ds = nullptr;
    dsp::SpecTables *dt = nullptr;               // fs enters only through the PSD scale U = fs * sum w^2 (classifier.cpp:296-301)
    DSP_HIP(hipMalloc(&dx, (size_t)n * sizeof(float)));
    const size_t sb = (size_t)dsp::kSpecBins * T * sizeof(float);
    if (hipMalloc(&ds, sb) != hipSuccess) { hipFree(dx); return fail(DSP_ENOMEM, "hipMalloc"); }
    hipError_t e = hipSuccess;
    if (fs != 16000) {
        dsp::SpecTables t;
        dsp::build_spec_tables(fs, t);
        e = hipMalloc(&dt, sizeof(t));
        if (e == hipSuccess) e = hipMemcpy(dt, &t, sizeof(t), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(dx, signal, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_spectrogram_f32(dx, 1, n, n, dt ? dt : g_cls.d_tab, ds, nullptr);
    if (e == hipSuccess) e = hipMemcpy(sxx, ds, sb, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(ds);
    if (dt) hipFree(dt);
    if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
    return T;
}

int dsp_compute_spectrogram_f64(const double *signal, int n, int fs, double *frequencies, double *times, double *sxx)
{
    if (!signal || !sxx || n < 0 || fs <= 0) return fail(DSP_EINVAL, "bad argument");
    const int T = spec_bins(n);
    if (frequencies)
        for (int k = 0; k < dsp::kSpecBins; ++k) frequencies[k] = (double)k * fs / dsp::kSpecSeg;                      // classifier.c:468-471
    if (times)
        for (int t = 0; t < T; ++t) times[t] = (double)(t * dsp::kSpecHop + dsp::kSpecSeg / 2) / fs;                    // :474-478
    if (T == 0) return 0;
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    double *dx = nullptr, *ds = nullptr;
    DSP_HIP(hipMalloc(&dx, (size_t)n * sizeof(double)));
    const size_t sb = (size_t)dsp::kSpecBins * T * sizeof(double);
    if (hipMalloc(&ds, sb) != hipSuccess) { hipFree(dx); return fail(DSP_ENOMEM, "hipMalloc"); }
    hipError_t e = hipMemcpy(dx, signal, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_spectrogram_f64(dx, 1, n, n, fs, ds, nullptr);
    if (e == hipSuccess) e = hipMemcpy(sxx, ds, sb, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(ds);
    if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
    return T;
}

#ifdef DSP_PF_STAMPS
__attribute__((visibility("default"))) int dsp_debug_pf_stamps(unsigned long long *out, int count)     // tools/pf_stamps.py
{
    DSP_HIP(hipDeviceSynchronize());
    DSP_HIP(dsp::read_pf_stamps(out, count));
    return DSP_OK;
}
#endif
#ifdef DSP_RC_STAMPS
__attribute__((visibility("default"))) int dsp_debug_rc_stamps(unsigned long long *out, int count)     // diagnostic builds only (tools/rc_stamps.py)
{
    DSP_HIP(hipDeviceSynchronize());
    DSP_HIP(dsp::read_rc_stamps(out, count));
    return DSP_OK;
}
__attribute__((visibility("default"))) int dsp_debug_bd_stamps(unsigned long long *out, int count)
{
    DSP_HIP(hipDeviceSynchronize());
    DSP_HIP(dsp::read_bd_stamps(out, count));
    return DSP_OK;
}
#endif

int dsp_sum_intense_f32(float lower, float upper, float half_range, const float *frequencies, int freq_bins,
                        const float *times, int time_bins, const float *db, float midpoint, float *out)
{
    if (!frequencies || !times || !db || !out || freq_bins <= 0 || time_bins <= 0) return fail(DSP_EINVAL, "bad argument");
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    const size_t nf = freq_bins, nt = time_bins, total = nf + nt + nf * nt + 1;
    float *d = nullptr;
    DSP_HIP(hipMalloc(&d, total * sizeof(float)));
    hipError_t e = hipMemcpy(d, frequencies, nf * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nf, times, nt * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nf + nt, db, nf * nt * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsp::launch_sum_intense(lower, upper, half_range, d, freq_bins, d + nf, time_bins, d + nf + nt, midpoint, d + nf + nt + nf * nt, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d + nf + nt + nf * nt, sizeof(float), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(DSP_EHIP, hipGetErrorString(e));
    return DSP_OK;
}

void dsp_classify_default_config(dsp_classify_config *cfg)
{
    if (cfg) *cfg = default_classify_cfg();
}

namespace {

int cls_bytes(int in) { return in == 0 ? 4 : (in == 1 ? 2 : 4); }     // bytes per sample, all channels

int cls_input_kind(int channels, int stereo_mode, int &in)
{
    if (channels != 1 && channels != 2) return fail(DSP_EINVAL, "channels must be 1 or 2");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return fail(DSP_EINVAL, "bad stereo_mode");
    in = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
    return DSP_OK;
}

int cls_host_entry(const dsp_classify_config *cfgp, const void *signal, int in, long n_clips, int n, long stride, int *labels, dsp_classify_trace *trace)
{
    if (!signal || !labels || n_clips < 0 || n < 0 || (n_clips > 1 && stride < n)) return fail(DSP_EINVAL, "bad argument");
    const dsp_classify_config cfg = cfgp ? *cfgp : default_classify_cfg();
    if (!valid_classify_cfg(cfg)) return fail(DSP_EINVAL, "classify config: thresholds must be finite with keep_lo < keep_hi");
    if (spec_bins(n) > kMaxSpecColumns) return fail(DSP_EINVAL, "clip too long (more than 957 spectrogram columns = 13.4 s at 16 kHz)");
    if (spec_bins(n) == 0) {      // clips shorter than one segment cannot fire the rule
        for (long c = 0; c < n_clips; ++c) labels[c] = 0;
        if (trace) std::memset(trace, 0, sizeof(*trace) * (size_t)n_clips);
        return DSP_OK;
    }
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    if ((rc = cls_init(g_cls, device)) < 0) return rc;
    const int bps = cls_bytes(in);
    const long row = (((long)n * bps + 15) & ~15L) / bps;            // staged rows start on 16 bytes
    if ((rc = cls_reserve(g_cls, std::min(kClsSubBatch, n_clips), n, true)) < 0) return rc;
    g_cls.wait_idle();                                               // the staging buffer is written by copies on the null stream
    ClsBusyMark mark{g_cls, nullptr};
    for (long c0 = 0; c0 < n_clips; c0 += kClsSubBatch) {
        const long cnt = std::min(kClsSubBatch, n_clips - c0);
        DSP_HIP(hipMemcpy2DAsync(g_cls.d_x, (size_t)row * bps, static_cast<const unsigned char *>(signal) + (size_t)c0 * stride * bps, (size_t)stride * bps,
                                 (size_t)n * bps, cnt, hipMemcpyHostToDevice, nullptr));
        if ((rc = cls_run(g_cls, cfg, g_cls.d_x, in, cnt, n, row, nullptr, trace != nullptr)) < 0) return rc;
        DSP_HIP(hipMemcpyAsync(labels + c0, g_cls.d_labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, nullptr));
        if (trace) DSP_HIP(hipMemcpyAsync(trace + c0, g_cls.d_trace, (size_t)cnt * sizeof(dsp::ClassifyTrace), hipMemcpyDeviceToHost, nullptr));
        DSP_HIP(hipStreamSynchronize(nullptr));
    }
    return DSP_OK;
}

int cls_device_entry(const dsp_classify_config *cfgp, const void *d_signal, int in, long n_clips, int n, long stride, int *d_labels, void *stream,
                     ClassifyCtx *own = nullptr)
{
    if (!d_signal || !d_labels || n_clips < 0 || n < 0 || (n_clips > 1 && stride < n)) return fail(DSP_EINVAL, "bad argument");
    const dsp_classify_config cfg = cfgp ? *cfgp : default_classify_cfg();
    if (!valid_classify_cfg(cfg)) return fail(DSP_EINVAL, "classify config: thresholds must be finite with keep_lo < keep_hi");
    if (spec_bins(n) > kMaxSpecColumns) return fail(DSP_EINVAL, "clip too long (more than 957 spectrogram columns = 13.4 s at 16 kHz)");
    if (n_clips == 0) return DSP_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_signal) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(DSP_EINVAL, "signal is not a device pointer");
    }
    if (attr.device < 0 || attr.device >= kMaxDevices) return fail(DSP_EINVAL, "device index out of range");
    if (own && own->device >= 0 && own->device != attr.device) return fail(DSP_EINVAL, "the context belongs to another device than the signal");
    ClassifyCtx &g_cls = own ? *own : g_cls_ctx[attr.device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(attr.device);
    int rc = cls_init(g_cls, attr.device);
    if (rc < 0) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (spec_bins(n) == 0) { DSP_HIP(hipMemsetAsync(d_labels, 0, (size_t)n_clips * sizeof(int), st)); return DSP_OK; }
    if ((rc = cls_reserve(g_cls, std::min(kClsSubBatch, n_clips), n, false)) < 0) return rc;
    if (g_cls.pending) DSP_HIP(hipStreamWaitEvent(st, g_cls.done, 0));      // the previous call's work on this workspace (any stream)
    ClsBusyMark mark{g_cls, st};
    for (long c0 = 0; c0 < n_clips; c0 += kClsSubBatch) {
        const long cnt = std::min(kClsSubBatch, n_clips - c0);
        const void *src = static_cast<const unsigned char *>(d_signal) + (size_t)c0 * stride * cls_bytes(in);
        if ((rc = cls_run(g_cls, cfg, src, in, cnt, n, stride, st, false)) < 0) return rc;
        DSP_HIP(hipMemcpyAsync(d_labels + c0, g_cls.d_labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToDevice, st));
    }
    return DSP_OK;
}

// Ragged batches (donut-classifier/classifier.c:286-297 reads one file of any length per run; its callers loop over files): the clips'
// spans from the host's offsets[n_clips + 1] (samples per channel from the buffer's start), every clip with the segments ITS length
// holds.  d_signal: the whole buffer on the GPU.  labels / trace (host, optional) are copied back when given, d_labels (device) otherwise.
int cls_ragged(const dsp_classify_config *cfgp, const void *d_signal, int device, int in, long n_clips, const long *offsets, int *d_labels, int *labels,
               dsp_classify_trace *trace, void *stream)
{
    const dsp_classify_config cfg = cfgp ? *cfgp : default_classify_cfg();
    if (!valid_classify_cfg(cfg)) return fail(DSP_EINVAL, "classify config: thresholds must be finite with keep_lo < keep_hi");
    int n_max = 0;
    for (long c = 0; c < n_clips; ++c) {
        const long n = offsets[c + 1] - offsets[c];
        if (offsets[c] < 0 || n < 0 || n > INT32_MAX) return fail(DSP_EINVAL, "offsets must be non-negative and non-decreasing, clips shorter than 2^31 samples");
        if (spec_bins((int)n) > kMaxSpecColumns) return fail(DSP_EINVAL, "clip " + std::to_string(c) + " too long (more than 957 spectrogram columns = 13.4 s at 16 kHz)");
        n_max = std::max(n_max, (int)n);
    }
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    DSP_ON_DEVICE(device);
    int rc = cls_init(g_cls, device);
    if (rc < 0) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (spec_bins(n_max) == 0) {                        // no clip holds a segment: no midpoints, label 0 (classifier.cpp:93-114)
        if (d_labels) DSP_HIP(hipMemsetAsync(d_labels, 0, (size_t)n_clips * sizeof(int), st));
        if (labels) std::memset(labels, 0, (size_t)n_clips * sizeof(int));
        if (trace) std::memset(trace, 0, (size_t)n_clips * sizeof(dsp_classify_trace));
        return DSP_OK;
    }
    // The clips run in order of length, longest first: the kernels take 64 clips per block and walk to the block's longest, so a block of
    // alike clips wastes nothing (measured on clips of 0.5 - 1.5 s in the caller's order: +54 % over the same samples in equal clips).
    // order[i] = the caller's index of the i-th clip as run; results go home through it (launch_scatter_records / on the host).
    if (n_clips >= (1L << 31)) return fail(DSP_EINVAL, "too many clips");
    std::vector<int> order((size_t)n_clips), segs((size_t)n_clips);
    for (long c = 0; c < n_clips; ++c) segs[c] = spec_bins((int)(offsets[c + 1] - offsets[c]));
    dsp::order_by_key_desc(segs.data(), n_clips, spec_bins(n_max), order.data());
    const size_t span_bytes = (size_t)n_clips * sizeof(dsp::ClipSpan), perm_bytes = (size_t)n_clips * sizeof(int);
    dsp::SpanRing::Slot *slot = nullptr;
    DSP_HIP(g_cls.spans.acquire(span_bytes + perm_bytes, &slot));
    dsp::ClipSpan *h = static_cast<dsp::ClipSpan *>(slot->h);
    for (long i = 0; i < n_clips; ++i) {
        const long c = order[i];
        h[i] = dsp::ClipSpan{offsets[c], (int)(offsets[c + 1] - offsets[c]), spec_bins((int)(offsets[c + 1] - offsets[c])), c, 0};
    }
    std::memcpy(static_cast<char *>(slot->h) + span_bytes, order.data(), perm_bytes);
    if (g_cls.pending) DSP_HIP(hipStreamWaitEvent(st, g_cls.done, 0));      // the previous call's work on this workspace (any stream)
    DSP_HIP(dsp::SpanRing::upload(slot, span_bytes + perm_bytes, st));
    const int *d_perm = reinterpret_cast<const int *>(static_cast<const char *>(slot->d) + span_bytes);
    std::vector<int> h_labels;
    std::vector<dsp::ClassifyTrace> h_trace;
    // Passes: as many clips as a pass of equal 1 s clips has SEGMENTS for (the workspaces are [clip][segments of the pass's longest clip]):
    // few clips per pass while they are long, the full 65 536 once they are short
    // (four times that before a pass is cut short: a small remainder pass costs a whole clip's sequential chain for few clips -- 0.5 - 1.5 s
    // clips in two passes measured 3.6 ms against 3.1 ms in one; the bound is there for batches with very long clips, ~11 GB of workspace)
    constexpr long kPassSegs = 4 * kClsSubBatch * 71;
    struct Pass { long c0, cnt; int n_row; };
    std::vector<Pass> passes;
    for (long c0 = 0; c0 < n_clips;) {
        const int t_row = std::max(1, h[c0].frames);                            // sorted: the pass's longest clip comes first
        const long cnt = std::min({kClsSubBatch, n_clips - c0, std::max(64L, kPassSegs / t_row)});
        passes.push_back(Pass{c0, cnt, (t_row - 1) * dsp::kSpecHop + dsp::kSpecSeg});
        c0 += cnt;
    }
    for (const Pass &ps : passes)
        if ((rc = cls_reserve(g_cls, ps.cnt, ps.n_row, false)) < 0) { dsp::SpanRing::mark(slot, st); return rc; }
    struct SlotMark { dsp::SpanRing::Slot *s; hipStream_t st; ~SlotMark() { dsp::SpanRing::mark(s, st); } } slot_mark{slot, st};
    ClsBusyMark mark{g_cls, st};
    const dsp::ClipSpan *d_spans = static_cast<const dsp::ClipSpan *>(slot->d);
    for (const Pass &ps : passes) {
        const long c0 = ps.c0, cnt = ps.cnt;
        if ((rc = cls_run(g_cls, cfg, d_signal, in, cnt, ps.n_row, 0, st, trace != nullptr, d_spans + c0, offsets[n_clips])) < 0) return rc;
        if (d_labels) DSP_HIP(dsp::launch_scatter_records(g_cls.d_labels, d_perm + c0, cnt, sizeof(int), d_labels, st));
        if (labels) {
            h_labels.resize((size_t)cnt);
            DSP_HIP(hipMemcpyAsync(h_labels.data(), g_cls.d_labels, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, st));
        }
        if (trace) {
            h_trace.resize((size_t)cnt);
            DSP_HIP(hipMemcpyAsync(h_trace.data(), g_cls.d_trace, (size_t)cnt * sizeof(dsp::ClassifyTrace), hipMemcpyDeviceToHost, st));
        }
        if (labels || trace) {
            DSP_HIP(hipStreamSynchronize(st));
            for (long i = 0; i < cnt; ++i) {
                if (labels) labels[order[c0 + i]] = h_labels[i];
                if (trace) std::memcpy(&trace[order[c0 + i]], &h_trace[i], sizeof(dsp_classify_trace));
            }
        }
    }
    return DSP_OK;
}

int cls_ragged_device_entry(const dsp_classify_config *cfgp, const void *d_signal, int in, long n_clips, const long *offsets, int *d_labels, void *stream)
{
    if (!d_signal || !d_labels || !offsets || n_clips < 0) return fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0) return DSP_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_signal) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(DSP_EINVAL, "signal is not a device pointer");
    }
    if (attr.device < 0 || attr.device >= kMaxDevices) return fail(DSP_EINVAL, "device index out of range");
    return cls_ragged(cfgp, d_signal, attr.device, in, n_clips, offsets, d_labels, nullptr, nullptr, stream);
}

int cls_ragged_host_entry(const dsp_classify_config *cfgp, const void *signal, int in, long n_clips, const long *offsets, int *labels, dsp_classify_trace *trace)
{
    if (!signal || !labels || !offsets || n_clips < 0) return fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0) return DSP_OK;
    if (offsets[n_clips] < offsets[0] || offsets[0] < 0) return fail(DSP_EINVAL, "offsets must be non-negative and non-decreasing, clips shorter than 2^31 samples");
    int device = 0;
    int rc = cls_host_device(device);
    if (rc < 0) return rc;
    // the whole buffer travels once (a buffer of its own: the contexts' staging rows are laid out for equal clips)
    void *d_flat = nullptr;
    const size_t bytes = (size_t)offsets[n_clips] * cls_bytes(in);
    {
        DSP_ON_DEVICE(device);
        DSP_HIP(hipMalloc(&d_flat, bytes + 16));
        const hipError_t e = hipMemcpy(d_flat, signal, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d_flat); DSP_HIP(e); }
    }
    rc = cls_ragged(cfgp, d_flat, device, in, n_clips, offsets, nullptr, labels, trace, nullptr);
    {
        dsp::DeviceScope on(device);
        (void)hipStreamSynchronize(nullptr);
        (void)hipFree(d_flat);
    }
    return rc;
}

}  // namespace

int dsp_classify_batch_host_cfg(const dsp_classify_config *cfgp, const float *signal, long n_clips, int n, long stride, int *labels,
                                dsp_classify_trace *trace)
{
    return cls_host_entry(cfgp, signal, 0, n_clips, n, stride, labels, trace);
}

int dsp_classify_batch_pcm16_host(const dsp_classify_config *cfgp, const int16_t *pcm, long n_clips, int n, long stride, int channels, int stereo_mode,
                                  int *labels, dsp_classify_trace *trace)
{
    int in = 0;
    const int rc = cls_input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : cls_host_entry(cfgp, pcm, in, n_clips, n, stride, labels, trace);
}

int dsp_classify_batch_pcm16_device(const dsp_classify_config *cfgp, const int16_t *d_pcm, long n_clips, int n, long stride, int channels,
                                    int stereo_mode, int *d_labels, void *stream)
{
    int in = 0;
    const int rc = cls_input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : cls_device_entry(cfgp, d_pcm, in, n_clips, n, stride, d_labels, stream);
}

int dsp_classify_batch_ragged_device(const dsp_classify_config *cfgp, const float *d_signal, long n_clips, const long *offsets, int *d_labels, void *stream)
{
    return cls_ragged_device_entry(cfgp, d_signal, 0, n_clips, offsets, d_labels, stream);
}

int dsp_classify_batch_ragged_pcm16_device(const dsp_classify_config *cfgp, const int16_t *d_pcm, long n_clips, const long *offsets, int channels,
                                           int stereo_mode, int *d_labels, void *stream)
{
    int in = 0;
    const int rc = cls_input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : cls_ragged_device_entry(cfgp, d_pcm, in, n_clips, offsets, d_labels, stream);
}

int dsp_classify_batch_ragged_host(const dsp_classify_config *cfgp, const float *signal, long n_clips, const long *offsets, int *labels,
                                   dsp_classify_trace *trace)
{
    return cls_ragged_host_entry(cfgp, signal, 0, n_clips, offsets, labels, trace);
}

int dsp_classify_batch_ragged_pcm16_host(const dsp_classify_config *cfgp, const int16_t *pcm, long n_clips, const long *offsets, int channels,
                                         int stereo_mode, int *labels, dsp_classify_trace *trace)
{
    int in = 0;
    const int rc = cls_input_kind(channels, stereo_mode, in);
    return rc < 0 ? rc : cls_ragged_host_entry(cfgp, pcm, in, n_clips, offsets, labels, trace);
}

/* A context of the caller's own: the default entry points share one workspace per device, so two calls on one device run one behind
 * the other; calls through different contexts (on different streams) may overlap. */
struct dsp_classify_ctx { ClassifyCtx c; int device; };

int dsp_classify_ctx_create(int device, dsp_classify_ctx **out)
{
    if (!out) return fail(DSP_EINVAL, "bad argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback");
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(DSP_EINVAL, "device index out of range");
    auto *ctx = new (std::nothrow) dsp_classify_ctx;
    if (!ctx) return fail(DSP_ENOMEM, "out of memory");
    ctx->device = device;
    {
        std::lock_guard<std::mutex> lock(ctx->c.mu);
        DSP_ON_DEVICE(device);
        const int rc = cls_init(ctx->c, device);
        if (rc < 0) { cls_release(ctx->c); delete ctx; return rc; }
    }
    *out = ctx;
    return DSP_OK;
}

void dsp_classify_ctx_destroy(dsp_classify_ctx *ctx)
{
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(ctx->c.mu);
        dsp::DeviceScope on(ctx->device);
        cls_release(ctx->c);
    }
    delete ctx;
}

int dsp_classify_batch_device_ctx(dsp_classify_ctx *ctx, const dsp_classify_config *cfgp, const float *d_signal, long n_clips, int n, long stride,
                                  int *d_labels, void *stream)
{
    if (!ctx) return fail(DSP_EINVAL, "bad argument");
    return cls_device_entry(cfgp, d_signal, 0, n_clips, n, stride, d_labels, stream, &ctx->c);
}

/* Test hook (no HIP call): holds the default classifier context of `device` for hold_ms milliseconds.  tests/test_capi_cpu.py runs it
 * from two threads: on two devices the holds overlap, on one device they queue. */
int dsp_debug_hold_classify_ctx(int device, int hold_ms)
{
    if (device < 0 || device >= kMaxDevices || hold_ms < 0 || hold_ms > 10000) return fail(DSP_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lock(g_cls_ctx[device].mu);
    std::this_thread::sleep_for(std::chrono::milliseconds(hold_ms));
    return DSP_OK;
}

int dsp_classify_stats(int device, long *gated_segments, long *listed_clips)
{
    if (device < 0 || device >= kMaxDevices) return fail(DSP_EINVAL, "device index out of range");
    ClassifyCtx &g_cls = g_cls_ctx[device];
    std::lock_guard<std::mutex> lock(g_cls.mu);
    if (g_cls.device < 0 || !g_cls.d_gate) return fail(DSP_EINVAL, "no classifier pass has run on this device");
    DSP_ON_DEVICE(device);
    g_cls.wait_idle();
    int ng = 0, nh = 0;
    DSP_HIP(hipMemcpy(&ng, g_cls.d_gate, sizeof(int), hipMemcpyDeviceToHost));
    DSP_HIP(hipMemcpy(&nh, g_cls.d_hits, sizeof(int), hipMemcpyDeviceToHost));
    if (gated_segments) *gated_segments = ng;
    if (listed_clips) *listed_clips = nh;
    return DSP_OK;
}

int dsp_classify_release(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    for (int d = 0; d < kMaxDevices && d < count; ++d) {
        if (device >= 0 && d != device) continue;
        ClassifyCtx &g_cls = g_cls_ctx[d];
        std::lock_guard<std::mutex> lock(g_cls.mu);
        if (g_cls.device < 0) continue;
        DSP_ON_DEVICE(d);
        cls_release(g_cls);
    }
    return DSP_OK;
}

int dsp_classify_batch_host(const float *signal, long n_clips, int n, long stride, int *labels, dsp_classify_trace *trace)
{
    return dsp_classify_batch_host_cfg(nullptr, signal, n_clips, n, stride, labels, trace);
}

int dsp_classify_batch_device_cfg(const dsp_classify_config *cfgp, const float *d_signal, long n_clips, int n, long stride,
                                  int *d_labels, void *stream)
{
    return cls_device_entry(cfgp, d_signal, 0, n_clips, n, stride, d_labels, stream);
}

int dsp_classify_batch_device(const float *d_signal, long n_clips, int n, long stride, int *d_labels, void *stream)
{
    return dsp_classify_batch_device_cfg(nullptr, d_signal, n_clips, n, stride, d_labels, stream);
}

// sync/lib/classifier.h:18 (find_midpoints): the midpoints are a by-product of the classify pipeline (its trace record)
int dsp_find_midpoints(const float *data, int num_frames, int fs, float *midpoints, int max_midpoints)
{
    if (!data || num_frames <= 0 || max_midpoints < 0 || (max_midpoints > 0 && !midpoints)) return fail(DSP_EINVAL, "bad argument");
    if (fs != 16000) return fail(DSP_EINVAL, "find_midpoints: only 16000 Hz has filter coefficients (classifier.cpp:138-191)");
    int label = 0;
    dsp_classify_trace tr;
    const int rc = dsp_classify_batch_host(data, 1, num_frames, num_frames, &label, &tr);
    if (rc < 0) return rc;
    for (int i = 0; i < tr.n_midpoints && i < max_midpoints; ++i) midpoints[i] = tr.midpoints[i];
    return tr.n_midpoints;
}

// sync/lib/classifier.h:19.  Same contract: 0/1, 0 also on failure (reason in dsp_last_error()).
int dsp_classify(float *data, int data_size)
{
    int label = 0;
    if (!data || data_size <= 0) { fail(DSP_EINVAL, "bad argument"); return 0; }
    const int rc = dsp_classify_batch_host(data, 1, data_size, data_size, &label, nullptr);
    if (rc < 0) {
        std::fprintf(stderr, "libdsp_amd: classify: %s\n", dsp_last_error());
        return 0;
    }
    return label;
}

}  // extern "C"

// ---- pooling + SVM ---------------------------------------------------------------------

struct dsp_svm {
    int device = 0;
    dsp::SvmModelDev m{};
    float *d_blob = nullptr;
};

extern "C" {

int dsp_mfcc_stats_device(const float *d_mfcc, long n_clips, int n_frames, int n_coef, float *d_feat, void *stream)
{
    if (n_clips < 0 || n_frames <= 0 || n_coef <= 0 || n_coef > 64 || (n_clips > 0 && (!d_mfcc || !d_feat)))
        return fail(DSP_EINVAL, "bad argument");
    if (n_clips == 0) return DSP_OK;
    // no handle here: launch on the GPU the caller's buffer lives on
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_mfcc) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(DSP_EINVAL, "d_mfcc is not a device pointer");
    }
    DSP_ON_DEVICE(attr.device);
    DSP_HIP(dsp::launch_mfcc_stats(d_mfcc, n_clips, n_frames, n_coef, d_feat, (hipStream_t)stream));
    return DSP_OK;
}

int dsp_svm_create(int device, int n_features, int n_sv, const float *offset, const float *scale,
                   const float *support_vectors, const float *coefficients, float gamma, float rho, float prob_a,
                   float prob_b, dsp_svm **out)
{
    if (!out || !offset || !scale || !support_vectors || !coefficients || n_features <= 0 || n_features > 256 || n_sv <= 0)
        return fail(DSP_EINVAL, "bad argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(DSP_ENODEV, "no HIP device: libdsp_amd has no CPU fallback");
    if (device < 0 || device >= n) return fail(DSP_EINVAL, "device index out of range");
    DSP_ON_DEVICE(device);
    auto *s = new dsp_svm;
    s->device = device;
    const size_t nf = n_features, ns = n_sv, total = 2 * nf + ns * nf + ns;
    if (hipMalloc(&s->d_blob, total * sizeof(float)) != hipSuccess) { delete s; return fail(DSP_ENOMEM, "hipMalloc"); }
    float *p = s->d_blob;
    hipError_t e = hipMemcpy(p, offset, nf * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p + nf, scale, nf * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p + 2 * nf, support_vectors, ns * nf * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p + 2 * nf + ns * nf, coefficients, ns * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(s->d_blob); delete s; return fail(DSP_EHIP, hipGetErrorString(e)); }
    s->m = {n_features, n_sv, gamma, rho, prob_a, prob_b, p, p + nf, p + 2 * nf, p + 2 * nf + ns * nf};
    *out = s;
    return DSP_OK;
}

void dsp_svm_destroy(dsp_svm *s)
{
    if (!s) return;
    dsp::DeviceScope dsp_device_scope_(s->device);
    if (s->d_blob) hipFree(s->d_blob);
    delete s;
}

}  // extern "C"

// Ragged batch -> spans in a ring slot (uploaded on `stream`).  offsets[n_clips + 1]: clip c is samples [offsets[c], offsets[c + 1]) per
// channel of the buffer; every clip must hold at least one frame.  *t_max: frames of the longest clip.
// The kernels deal the spans to their n_waves wavefronts in fixed order (wave w walks spans w, w + n_waves, ...), so the ORDER of the
// spans is the load balance: by frame count, longest first, and snaking -- left to right over the waves in even rounds, right to left in
// odd ones -- every wave's total is within a clip of the mean (in the caller's order: +14 % on clips of 0.5 - 1.5 s).
// (host only: no HIP call) fills h[n_clips]; returns DSP_OK or DSP_EINVAL with the reason
static int build_fused_spans(const dsp_mfcc_config &cfg, const long *offsets, long n_clips, int max_frames, long n_waves, dsp::ClipSpan *h, int *t_max)
{
    std::vector<int> frames((size_t)n_clips), order((size_t)n_clips);
    int tm = 0;
    for (long c = 0; c < n_clips; ++c) {
        const long n = offsets[c + 1] - offsets[c];
        if (offsets[c] < 0 || n < 0 || n > INT32_MAX) return fail(DSP_EINVAL, "offsets must be non-negative and non-decreasing, clips shorter than 2^31 samples");
        const int t = dsp_mfcc_frames_for(&cfg, (int)n, max_frames);
        if (t == 0) return fail(DSP_EINVAL, "clip " + std::to_string(c) + " of the ragged batch is shorter than one frame");
        frames[c] = t;
        tm = std::max(tm, t);
    }
    dsp::order_by_key_desc(frames.data(), n_clips, tm, order.data());
    n_waves = std::max(1L, n_waves);
    for (long i = 0; i < n_clips; ++i) {
        const long round = i / n_waves, j = i - round * n_waves;
        const long width = std::min(n_waves, n_clips - round * n_waves);          // the last round may be short
        const long pos = round * n_waves + ((round & 1) ? width - 1 - j : j);
        const long c = order[i];
        h[pos] = dsp::ClipSpan{offsets[c], (int)(offsets[c + 1] - offsets[c]), frames[c], c, 0};
    }
    *t_max = tm;
    return DSP_OK;
}

static int ragged_spans(dsp_mfcc_plan *p, const long *offsets, long n_clips, int max_frames, long n_waves, dsp::SpanRing::Slot **slot, int *t_max, void *stream)
{
    if (!offsets) return fail(DSP_EINVAL, "offsets is NULL");
    if (n_clips >= (1L << 31)) return fail(DSP_EINVAL, "too many clips");
    DSP_HIP(p->spans.acquire((size_t)n_clips * sizeof(dsp::ClipSpan), slot));
    const int rc = build_fused_spans(p->cfg, offsets, n_clips, max_frames, n_waves, static_cast<dsp::ClipSpan *>((*slot)->h), t_max);
    if (rc < 0) return rc;
    DSP_HIP(dsp::SpanRing::upload(*slot, (size_t)n_clips * sizeof(dsp::ClipSpan), (hipStream_t)stream));
    return DSP_OK;
}

/* Test hook (host only): the order a ragged batch of the fused clip kernels runs in -- spans[pos] = {start, samples, frames, caller's
 * index} as 4 longs per clip; wave w of n_waves walks pos = w, w + n_waves, ...  tests/test_capi_cpu.py checks it (also under ASan). */
extern "C" int dsp_debug_fused_spans(const dsp_mfcc_config *cfg, const long *offsets, long n_clips, int max_frames, long n_waves, long *out4)
{
    if (!cfg || !offsets || n_clips < 0 || (n_clips > 0 && !out4)) return fail(DSP_EINVAL, "bad argument");
    std::vector<dsp::ClipSpan> h((size_t)n_clips);
    int tm = 0;
    const int rc = build_fused_spans(*cfg, offsets, n_clips, max_frames, n_waves, h.data(), &tm);
    if (rc < 0) return rc;
    for (long i = 0; i < n_clips; ++i) { out4[4 * i] = h[i].off; out4[4 * i + 1] = h[i].n; out4[4 * i + 2] = h[i].frames; out4[4 * i + 3] = h[i].orig; }
    return tm;
}

// clip -> label in one kernel; in_kind 0 = float samples, 1 / 2 / 3 = int16 mono / stereo channel 0 / stereo average.
// offsets != nullptr: a ragged batch (clips of different lengths back to back or anywhere in the buffer; samples_per_clip / clip_stride unused)
static int scrubjay_fused(dsp_mfcc_plan *p, dsp_svm *s, const void *d_signal, int in_kind, long n_clips, int samples_per_clip,
                          long clip_stride, int max_frames, int *d_labels, float *d_decision, float *d_prob1, float *d_feat, void *stream,
                          const long *offsets = nullptr)
{
    if (!p || !s || n_clips < 0) return fail(DSP_EINVAL, "bad argument");
    if ((p->cfg.n_fft != 512 && p->cfg.n_fft != 2048) || (p->cfg.log_mode != DSP_LOG_PER_FRAME_MAX && p->cfg.log_mode != DSP_LOG_LOG10_FLOOR) ||
        p->cfg.prefilter != DSP_PREFILTER_NONE || p->kernel != DSP_KERNEL_WAVE)
        return fail(DSP_EINVAL, "the fused clip -> label path runs on the 512- and 2048-point wave-per-frame kernels, per-frame log modes");
    const bool aub2048 = p->cfg.n_fft == 2048 && (p->cfg.spectrum != DSP_SPECTRUM_POWER || p->cfg.log_mode == DSP_LOG_LOG10_FLOOR || p->cfg.framing == DSP_FRAMING_STREAM);
    if (in_kind != 0 && !aub2048 && (p->cfg.n_fft != 512 || p->cfg.frame_length != 400 || p->host.mel_gather != 3 ||
                                     !((p->host.dct_split == 4 && p->host.dct_len == 10) || (p->host.dct_split == 2 && p->host.dct_len == 20))))
        return fail(DSP_EINVAL, "int16 input of the fused clip -> label kernel: the reference framing (n_fft 512, frame 400, 40 mel filters, up to 20 coefficients) "
                                "or the scrubjay_infer.c front end (dsp_mfcc_scrubjay_infer_config)");
    if (s->m.n_features != 2 * p->cfg.n_mfcc || s->m.n_features > 64) return fail(DSP_EINVAL, "SVM n_features must equal 2 * n_mfcc (<= 64)");
    const bool ragged = offsets != nullptr;
    int t = ragged ? 1 : dsp_mfcc_frames_for(&p->cfg, samples_per_clip, max_frames);
    if (n_clips == 0) return 0;
    if (t == 0) return fail(DSP_EINVAL, "clips shorter than one frame have no features to pool");
    if (!d_signal || !d_labels) return fail(DSP_EINVAL, "NULL buffer");
    if (!ragged && n_clips > 1 && clip_stride < samples_per_clip) return fail(DSP_EINVAL, "clip_stride < samples_per_clip");
    if ((reinterpret_cast<uintptr_t>(d_signal) & (in_kind == 1 ? 3 : 7)) || (!ragged && n_clips > 1 && (clip_stride & 1)))
        return fail(DSP_EINVAL, "input must be 8-byte aligned (4 for mono int16) with an even clip stride");
    if (s->device != p->device) return fail(DSP_EINVAL, "plan and SVM live on different devices");
    DSP_ON_DEVICE(p->device);
    const int per_cu_f = p->cfg.n_fft == 2048 ? (p->blocks_per_cu > 0 ? p->blocks_per_cu : p->resident_blocks_2048_pool)
                                              : (p->blocks_per_cu > 0 ? p->blocks_per_cu : p->resident_blocks);
    const long blocks_f = std::max(1L, std::min((long)p->n_cu * per_cu_f, (n_clips + 3) / 4));
    dsp::SpanRing::Slot *slot = nullptr;
    if (ragged) {
        const int rc = ragged_spans(p, offsets, n_clips, max_frames, 4 * blocks_f, &slot, &t, stream);
        if (rc < 0) return rc;
    }
    dsp::Mfcc512Args a{};
    a.in = d_signal;
    a.in_kind = in_kind;
    a.out = nullptr;
    a.tables = p->d_tables;
    a.n_frames = n_clips * (long)t;
    a.n_clips = n_clips;
    a.spans = ragged ? static_cast<const dsp::ClipSpan *>(slot->d) : nullptr;
    a.clip_stride = ragged ? 0 : clip_stride;
    a.frames_per_clip = t;
    a.hop = p->cfg.hop_length;
    a.frame_len = p->cfg.frame_length;
    a.chunk = t;                                  // one wavefront walks one clip
    a.n_mels = p->cfg.n_mels;
    a.n_mfcc = p->cfg.n_mfcc;
    a.amin = p->cfg.amin;
    a.top_db = p->cfg.top_db;
    a.log_mode = p->cfg.log_mode;
    a.spectrum = p->cfg.spectrum;
    a.stream_framing = p->cfg.framing == DSP_FRAMING_STREAM;
    a.samples_per_clip = ragged ? 0 : samples_per_clip;
    a.pool.svm = s->m;
    a.pool.labels = d_labels;
    a.pool.decision = d_decision;
    a.pool.prob1 = d_prob1;
    a.pool.feat = d_feat;
    hipError_t e;
    if (p->cfg.n_fft == 2048) {      // scrubjay_infer.c's own framing (WIN_SIZE 2048, HOP_SIZE 1024): mfcc2048_kernel<POOL>
        e = dsp::launch_mfcc2048(a, p->d_tables2048, (int)blocks_f, (hipStream_t)stream, true);
    } else {
        e = dsp::launch_mfcc512_pool(a, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks_f, (hipStream_t)stream);
    }
    if (slot) dsp::SpanRing::mark(slot, (hipStream_t)stream);
    DSP_HIP(e);
    return t;
}

extern "C" {

int dsp_scrubjay_fused_device(dsp_mfcc_plan *p, dsp_svm *s, const float *d_signal, long n_clips, int samples_per_clip,
                              long clip_stride, int max_frames, int *d_labels, float *d_decision, float *d_prob1, float *d_feat,
                              void *stream)
{
    return scrubjay_fused(p, s, d_signal, 0, n_clips, samples_per_clip, clip_stride, max_frames, d_labels, d_decision, d_prob1, d_feat, stream);
}

int dsp_scrubjay_fused_pcm16_device(dsp_mfcc_plan *p, dsp_svm *s, const int16_t *d_pcm, long n_clips, int samples_per_clip, long clip_stride,
                                    int channels, int stereo_mode, int max_frames, int *d_labels, float *d_decision, float *d_prob1, float *d_feat,
                                    void *stream)
{
    if (channels != 1 && channels != 2) return fail(DSP_EINVAL, "channels must be 1 or 2");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return fail(DSP_EINVAL, "bad stereo_mode");
    const int kind = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
    return scrubjay_fused(p, s, d_pcm, kind, n_clips, samples_per_clip, clip_stride, max_frames, d_labels, d_decision, d_prob1, d_feat, stream);
}

int dsp_scrubjay_fused_ragged_device(dsp_mfcc_plan *p, dsp_svm *s, const float *d_signal, long n_clips, const long *offsets, int max_frames,
                                     int *d_labels, float *d_decision, float *d_prob1, float *d_feat, void *stream)
{
    if (!offsets) return fail(DSP_EINVAL, "offsets is NULL");
    return scrubjay_fused(p, s, d_signal, 0, n_clips, 0, 0, max_frames, d_labels, d_decision, d_prob1, d_feat, stream, offsets);
}

int dsp_scrubjay_fused_ragged_pcm16_device(dsp_mfcc_plan *p, dsp_svm *s, const int16_t *d_pcm, long n_clips, const long *offsets, int channels,
                                           int stereo_mode, int max_frames, int *d_labels, float *d_decision, float *d_prob1, float *d_feat,
                                           void *stream)
{
    if (!offsets) return fail(DSP_EINVAL, "offsets is NULL");
    if (channels != 1 && channels != 2) return fail(DSP_EINVAL, "channels must be 1 or 2");
    if (channels == 2 && stereo_mode != DSP_STEREO_CHANNEL0 && stereo_mode != DSP_STEREO_AVERAGE) return fail(DSP_EINVAL, "bad stereo_mode");
    const int kind = channels == 1 ? 1 : (stereo_mode == DSP_STEREO_CHANNEL0 ? 2 : 3);
    return scrubjay_fused(p, s, d_pcm, kind, n_clips, 0, 0, max_frames, d_labels, d_decision, d_prob1, d_feat, stream, offsets);
}

}  // extern "C"

int dsp::plan_device(const dsp_mfcc_plan *plan) { return plan ? plan->device : -1; }

// capi_util.hpp: the fused form of dsp_classify_signal_batch_device (capi_consumers.cpp)
int dsp::stop_fused_device(dsp_mfcc_plan *p, const dsp::StopModelDev &m, const void *d_signal, long n_clips, int samples_per_clip,
                           long clip_stride, int t, float *d_prob, void *stream, int in_kind, const long *offsets)
{
    (void)samples_per_clip;
    const bool ragged = offsets != nullptr;
    if (ragged) { t = 1; clip_stride = 0; }
    // the reference's shape on the default kernel: 512-point, per-frame log, 13 coefficients of 40 mel energies, complete frames
    if (p->cfg.n_fft != 512 || p->cfg.log_mode != DSP_LOG_PER_FRAME_MAX || p->cfg.prefilter != DSP_PREFILTER_NONE || p->kernel != DSP_KERNEL_WAVE ||
        p->host.dct_split != 4 || p->host.dct_len != 10 || m.n_coef != p->cfg.n_mfcc || m.units[0] > dsp::kStopFusedUnits || !m.fold_a || t <= 0 ||
        std::getenv("DSP_AMD_STOP_TWO_KERNELS"))
        return 0;
    if ((reinterpret_cast<uintptr_t>(d_signal) & (in_kind == 1 ? 3 : 7)) || (n_clips > 1 && (clip_stride & 1))) return 0;      // the two-kernel path reports it
    if (in_kind != 0 && (p->host.mel_gather != 3 || p->cfg.frame_length != 400)) return 0;
    if (m.max_frames <= 0) return fail(DSP_EINVAL, "stop model without frames");
    DSP_ON_DEVICE(p->device);
    dsp::SpanRing::Slot *slot = nullptr;
    const int per_cu = p->blocks_per_cu > 0 ? p->blocks_per_cu : p->resident_blocks;
    long blocks = std::max(1L, std::min((long)p->n_cu * per_cu, (n_clips + 3) / 4));
    blocks = dsp::mfcc512_stop_grid((int)blocks, m, in_kind, p->host.mel_gather, p->cfg.frame_length);      // what the launcher will start
    if (ragged) {      // frames past the model's max_frames are dropped (stop_detector.c:26-30): a clip's walk ends there
        const int rc = ragged_spans(p, offsets, n_clips, m.max_frames, 4 * blocks, &slot, &t, stream);
        if (rc < 0) return rc;
    }
    dsp::Mfcc512Args a{};
    a.in = d_signal;
    a.in_kind = in_kind;
    a.out = nullptr;
    a.tables = p->d_tables;
    a.n_frames = n_clips * (long)t;
    a.n_clips = n_clips;
    a.spans = ragged ? static_cast<const dsp::ClipSpan *>(slot->d) : nullptr;
    a.clip_stride = clip_stride;
    a.frames_per_clip = t;
    a.hop = p->cfg.hop_length;
    a.frame_len = p->cfg.frame_length;
    a.chunk = t;                                  // one wavefront walks one clip
    a.n_mels = p->cfg.n_mels;
    a.n_mfcc = p->cfg.n_mfcc;
    a.amin = p->cfg.amin;
    a.top_db = p->cfg.top_db;
    a.log_mode = 0;
    a.stop.m = m;
    a.stop.prob = d_prob;
    const hipError_t e = dsp::launch_mfcc512_stop(a, p->host.dct_split, p->host.dct_len, p->host.mel_gather, (int)blocks, (hipStream_t)stream);
    if (slot) dsp::SpanRing::mark(slot, (hipStream_t)stream);
    DSP_HIP(e);
    return 1;
}

extern "C" {

int dsp_svm_predict_device(dsp_svm *s, const float *d_feat, long n_clips, int *d_labels, float *d_decision,
                           float *d_prob1, void *stream)
{
    if (!s || n_clips < 0 || (n_clips > 0 && (!d_feat || !d_labels))) return fail(DSP_EINVAL, "bad argument");
    DSP_ON_DEVICE(s->device);
    DSP_HIP(dsp::launch_svm_predict(s->m, d_feat, n_clips, d_labels, d_decision, d_prob1, (hipStream_t)stream));
    return DSP_OK;
}

}  // extern "C"

// ---- the reference's entry point ------------------------------------------------

static dsp_mfcc_plan *g_default_plan = nullptr;
static std::mutex g_default_mu;

// 2fa/audio/word/c/mfcc.h:16-19.  Same contract as the reference: returns the
// frame count, 0 for "clip too short / no room"; a GPU failure also returns 0
// (no frames were produced) with the cause in dsp_last_error().
extern "C" int compute_mfcc(const float *signal, int num_samples, float *out_mfcc, int max_frames)
{
    dsp_mfcc_config cfg;
    dsp_mfcc_default_config(&cfg);
    if (dsp_mfcc_frames_for(&cfg, num_samples, max_frames) == 0) return 0;   // mfcc.c:117-119
    if (!signal || !out_mfcc) { fail(DSP_EINVAL, "NULL buffer"); return 0; }
    dsp_mfcc_plan *plan;
    {
        std::lock_guard<std::mutex> lock(g_default_mu);
        if (!g_default_plan) {
            const char *dev = std::getenv("DSP_AMD_DEVICE");
            if (dsp_mfcc_plan_create(&cfg, dev ? std::atoi(dev) : 0, &g_default_plan) < 0) {
                std::fprintf(stderr, "libdsp_amd: compute_mfcc: %s\n", dsp_last_error());
                return 0;
            }
        }
        plan = g_default_plan;
    }
    const int t = dsp_mfcc_clips_host(plan, signal, 1, num_samples, num_samples, out_mfcc, max_frames);
    if (t < 0) {
        std::fprintf(stderr, "libdsp_amd: compute_mfcc: %s\n", dsp_last_error());
        return 0;
    }
    return t;
}
