// capi_util.hpp -- error reporting shared by the translation units of the C ABI.
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dsp_amd.h"

struct dsp_mfcc_plan;
namespace dsp {
int capi_fail(int code, const std::string &msg);   // sets dsp_last_error() for this thread, returns code
struct StopModelDev;
// capi.cpp (owner of dsp_mfcc_plan): classify_signal in one kernel -- clip -> MFCC -> stop-word net, the MFCC matrix never written.
// Returns 1 when the fused kernel was enqueued, 0 when this plan / model shape has no fused form (the caller runs the two-kernel
// path), < 0 on error.  t = frames per clip (already capped at the model's max_frames).
// in_kind: 0 float samples, 1 / 2 / 3 int16 mono / stereo channel 0 / stereo average
int stop_fused_device(dsp_mfcc_plan *plan, const StopModelDev &m, const void *d_signal, long n_clips, int samples_per_clip,
                      long clip_stride, int t, float *d_prob, void *stream, int in_kind = 0, const long *offsets = nullptr);
// offsets != nullptr: a ragged batch (clip c = samples [offsets[c], offsets[c + 1]) per channel; samples_per_clip, clip_stride and t unused)
int plan_device(const dsp_mfcc_plan *plan);          // the GPU a plan lives on
}

// Ragged batches: the clips' spans (clip_span.hpp ClipSpan: start, samples, frames, caller's index -- 32 bytes per clip) travel to the GPU through a
// small ring of pinned host / device buffer pairs, so that a call neither waits for the stream it enqueues on nor shares a buffer with
// the call before it (which may still be running, on this stream or another).  A slot is reused only after the event recorded behind
// the kernels that read it.
namespace dsp {
struct ClipSpan;
struct SpanRing {
    static constexpr int kSlots = 4;
    struct Slot { void *h = nullptr, *d = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    Slot slot[kSlots];
    int next = 0;
    std::mutex mu;
    // a slot with room for `bytes`, its previous user finished; fill slot->h, then upload(), launch, then mark()
    hipError_t acquire(size_t bytes, Slot **out)
    {
        std::lock_guard<std::mutex> lock(mu);
        Slot &s = slot[next];
        next = (next + 1) % kSlots;
        hipError_t e;
        if (!s.done && (e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return e;
        if (s.used && (e = hipEventSynchronize(s.done)) != hipSuccess) return e;
        s.used = false;
        if (s.cap < bytes) {
            if (s.h) (void)hipHostFree(s.h);
            if (s.d) (void)hipFree(s.d);
            s.h = s.d = nullptr; s.cap = 0;
            const size_t cap = bytes + bytes / 2 + 4096;
            if ((e = hipHostMalloc(&s.h, cap, hipHostMallocDefault)) != hipSuccess) return e;
            if ((e = hipMalloc(&s.d, cap)) != hipSuccess) return e;
            s.cap = cap;
        }
        *out = &s;
        return hipSuccess;
    }
    static hipError_t upload(Slot *s, size_t bytes, hipStream_t st) { return hipMemcpyAsync(s->d, s->h, bytes, hipMemcpyHostToDevice, st); }
    // after the last kernel that reads the slot has been enqueued (also on error exits once upload() ran)
    static void mark(Slot *s, hipStream_t st) { s->used = hipEventRecord(s->done, st) == hipSuccess; if (!s->used) (void)hipStreamSynchronize(st); }
    void release()      // on the owner's device
    {
        std::lock_guard<std::mutex> lock(mu);
        for (Slot &s : slot) {
            if (s.used) (void)hipEventSynchronize(s.done);
            if (s.h) (void)hipHostFree(s.h);
            if (s.d) (void)hipFree(s.d);
            if (s.done) (void)hipEventDestroy(s.done);
            s = Slot{};
        }
    }
};
}

// order[i] = index of the i-th largest key, ties in input order (what std::stable_sort gives) -- by counting: a ragged batch of 125 000
// clips is ordered in well under a millisecond, where the comparison sort took longer than the kernel it was ordering for
namespace dsp {
inline void order_by_key_desc(const int *key, long n, int key_max, int *order)
{
    if (key_max < 0 || key_max > (1 << 22)) {               // (absurd key ranges: the comparison sort)
        for (long i = 0; i < n; ++i) order[i] = (int)i;
        std::stable_sort(order, order + n, [&](int a, int b) { return key[a] > key[b]; });
        return;
    }
    std::vector<long> start((size_t)key_max + 2, 0);
    for (long i = 0; i < n; ++i) ++start[(size_t)(key_max - key[i]) + 1];
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    for (long i = 0; i < n; ++i) order[start[(size_t)(key_max - key[i])]++] = (int)i;
}
}

#define DSP_CAPI_HIP(call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return dsp::capi_fail(DSP_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The entry points work on the GPU their plan / model / buffers live on, which need not be the caller's current device
// (torch keeps its own idea of it): switch for the duration of the call and put the caller's device back on every exit.
namespace dsp {
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device); else prev = -1;      // nothing to restore
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
}
#define DSP_ON_DEVICE(dev)                                                                                        \
    dsp::DeviceScope dsp_device_scope_(dev);                                                                      \
    if (dsp_device_scope_.err != hipSuccess)                                                                      \
        return dsp::capi_fail(DSP_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(dsp_device_scope_.err))
