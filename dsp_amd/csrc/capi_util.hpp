// capi_util.hpp -- error reporting shared by the translation units of the C ABI.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

#include "../../include/dsp_amd.h"

namespace dsp {
int capi_fail(int code, const std::string &msg);   // sets dsp_last_error() for this thread, returns code
}

#define DSP_CAPI_HIP(call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return dsp::capi_fail(DSP_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
