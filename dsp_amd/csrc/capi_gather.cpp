// capi_gather.cpp -- the path's one exchange step for a C host that drives several GPUs from ONE process ("RCCL over
// xGMI used only to gather the per-clip feature vectors"): include/dsp_amd.h dsp_gather_*.  One RCCL communicator per listed device
// (ncclCommInitAll), one grouped ncclAllGather per call -- the same collective on the same bytes as dsp_amd/dist.py issues through
// torch.distributed in the one-process-per-GPU layout.  The reference has no distributed code: there is no reference line
// to match here, 8e is the specification.
// RCCL is resolved at the first dsp_gather_create with dlopen("librccl.so.1") -- no link-time dependency (a host that never gathers
// never loads it, and a process that already holds an RCCL, e.g. PyTorch's, gets that one).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "capi_util.hpp"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return true;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { g_rccl.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return false; }
    auto sym = [&](const char *n) { return dlsym(h, n); };
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.GetErrorString) {
        g_rccl.error = "librccl lacks a collective entry point";
        dlclose(h);
        return false;
    }
    g_rccl.handle = h;
    return true;
}

int rccl_fail(const char *what, ncclResult_t r)
{
    return dsp::capi_fail(DSP_EHIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}

}  // namespace

struct dsp_gather {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};

extern "C" {

int dsp_gather_create(const int *devices, int n_devices, dsp_gather **out)
{
    if (!out) return dsp::capi_fail(DSP_EINVAL, "bad argument");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return dsp::capi_fail(DSP_EINVAL, "1 .. 64 devices");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return dsp::capi_fail(DSP_ENODEV, "no HIP device"); }
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= count) return dsp::capi_fail(DSP_EINVAL, "device index out of range");
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) return dsp::capi_fail(DSP_EINVAL, "a device is listed twice");
    }
    if (!rccl_load()) return dsp::capi_fail(DSP_ENODEV, g_rccl.error);
    auto *g = new (std::nothrow) dsp_gather;
    if (!g) return dsp::capi_fail(DSP_ENOMEM, "out of memory");
    g->devices.assign(devices, devices + n_devices);
    g->comms.assign((size_t)n_devices, nullptr);
    int prev = -1;
    (void)hipGetDevice(&prev);
    const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
    if (prev >= 0) (void)hipSetDevice(prev);
    if (r != ncclSuccess) { delete g; return rccl_fail("ncclCommInitAll", r); }
    *out = g;
    return DSP_OK;
}

void dsp_gather_destroy(dsp_gather *g)
{
    if (!g) return;
    for (ncclComm_t c : g->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    delete g;
}

int dsp_gather_n_devices(const dsp_gather *g) { return g ? (int)g->devices.size() : dsp::capi_fail(DSP_EINVAL, "bad argument"); }

int dsp_gather_all(dsp_gather *g, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank, void *const *streams)
{
    if (!g || !d_send || !d_recv) return dsp::capi_fail(DSP_EINVAL, "bad argument");
    const int n = (int)g->devices.size();
    for (int r = 0; r < n; ++r)
        if (!d_send[r] || !d_recv[r]) return dsp::capi_fail(DSP_EINVAL, "null buffer");
    if (bytes_per_rank == 0) return DSP_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    // one group: every rank's call is enqueued before any of them may wait for its peers (one thread drives all ranks)
    ncclResult_t res = g_rccl.GroupStart();
    for (int r = 0; r < n && res == ncclSuccess; ++r) {
        if (hipSetDevice(g->devices[r]) != hipSuccess) { (void)hipGetLastError(); res = ncclUnhandledCudaError; break; }
        res = g_rccl.AllGather(d_send[r], d_recv[r], bytes_per_rank, ncclInt8, g->comms[r], streams ? (hipStream_t)streams[r] : nullptr);
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (res != ncclSuccess) return rccl_fail("ncclAllGather", res);
    if (end != ncclSuccess) return rccl_fail("ncclGroupEnd", end);
    return DSP_OK;
}

}  // extern "C"
