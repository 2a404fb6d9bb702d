// capi_util.hpp -- error reporting shared by the translation units of the C ABI.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

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
                      long clip_stride, int t, float *d_prob, void *stream, int in_kind = 0);
int plan_device(const dsp_mfcc_plan *plan);          // the GPU a plan lives on
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
