// membench.hip -- read-streaming micro-benchmark for the MFCC frame layout
// (development aid).  Each wave walks 2 KiB "frames" like the MFCC kernel and
// reduces them to one float per frame, so only the load pattern is timed.
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4 *p)
{
    f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: 4 x dwordx2 per lane (z[l+64a]);  MODE 1: 2 x dwordx4 per lane (x[4l..], x[256+4l..])
// MODE 2: 2 x dwordx4, two frames in flight per wave (prefetch depth 2)
// MODE 3: 2 x dwordx4, nontemporal loads
template <int MODE>
__global__ __launch_bounds__(256) void stream(const float *__restrict__ in, float *__restrict__ out, long n_frames, int chunk)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * 4;
    for (long c0 = wave * chunk; c0 < n_frames; c0 += n_waves * chunk) {
        const long c1 = c0 + chunk < n_frames ? c0 + chunk : n_frames;
        if (MODE == 0) {
            float2 nxt[4];
            for (int a = 0; a < 4; ++a) nxt[a] = *reinterpret_cast<const float2 *>(in + c0 * 512 + 2 * (lane + 64 * a));
            for (long f = c0; f < c1; ++f) {
                float2 cur[4];
                for (int a = 0; a < 4; ++a) cur[a] = nxt[a];
                if (f + 1 < c1)
                    for (int a = 0; a < 4; ++a) nxt[a] = *reinterpret_cast<const float2 *>(in + (f + 1) * 512 + 2 * (lane + 64 * a));
                float s = (cur[0].x + cur[0].y) + (cur[1].x + cur[1].y) + (cur[2].x + cur[2].y) + (cur[3].x + cur[3].y);
                s += __shfl_xor(s, 1);
                if (lane < 13) out[f * 13 + lane] = s;
            }
        } else if (MODE == 1 || MODE == 3) {
            float4 nxt[2];
            for (int a = 0; a < 2; ++a) {
                const float4 *p = reinterpret_cast<const float4 *>(in + c0 * 512 + 4 * lane + 256 * a);
                nxt[a] = MODE == 3 ? nt_load(p) : *p;
            }
            for (long f = c0; f < c1; ++f) {
                float4 cur[2] = {nxt[0], nxt[1]};
                if (f + 1 < c1)
                    for (int a = 0; a < 2; ++a) {
                        const float4 *p = reinterpret_cast<const float4 *>(in + (f + 1) * 512 + 4 * lane + 256 * a);
                        nxt[a] = MODE == 3 ? nt_load(p) : *p;
                    }
                float s = (cur[0].x + cur[0].y) + (cur[0].z + cur[0].w) + (cur[1].x + cur[1].y) + (cur[1].z + cur[1].w);
                s += __shfl_xor(s, 1);
                if (lane < 13) out[f * 13 + lane] = s;
            }
        } else {
            float4 n1[2], n2[2];
            for (int a = 0; a < 2; ++a) n1[a] = *reinterpret_cast<const float4 *>(in + c0 * 512 + 4 * lane + 256 * a);
            if (c0 + 1 < c1)
                for (int a = 0; a < 2; ++a) n2[a] = *reinterpret_cast<const float4 *>(in + (c0 + 1) * 512 + 4 * lane + 256 * a);
            for (long f = c0; f < c1; ++f) {
                float4 cur[2] = {n1[0], n1[1]};
                n1[0] = n2[0]; n1[1] = n2[1];
                if (f + 2 < c1)
                    for (int a = 0; a < 2; ++a) n2[a] = *reinterpret_cast<const float4 *>(in + (f + 2) * 512 + 4 * lane + 256 * a);
                float s = (cur[0].x + cur[0].y) + (cur[0].z + cur[0].w) + (cur[1].x + cur[1].y) + (cur[1].z + cur[1].w);
                s += __shfl_xor(s, 1);
                if (lane < 13) out[f * 13 + lane] = s;
            }
        }
    }
}

// plain grid-stride float4 read-reduce: the "known good" streaming-read reference
__global__ __launch_bounds__(256) void plain_read(const float4 *__restrict__ in, float *__restrict__ out, long n4)
{
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    if (s == 123.456f) out[0] = s;
}

template <class F>
float time_ms(F f, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < iters; ++i) f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const long n = 1000000;
    float *in, *out;
    CK(hipMalloc(&in, n * 512 * sizeof(float)));
    CK(hipMalloc(&out, n * 13 * sizeof(float)));
    CK(hipMemset(in, 0x3c, n * 512 * sizeof(float)));
    const double gb = n * 2100.0 / 1e9;
    for (int bpc : {4, 5, 8}) {
        for (int chunk : {8, 32}) {
            const int blocks = 256 * bpc;
            float t0 = time_ms([&] { hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
            float t1 = time_ms([&] { hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
            float t2 = time_ms([&] { hipLaunchKernelGGL(stream<2>, dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
            float t3 = time_ms([&] { hipLaunchKernelGGL(stream<3>, dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
            printf("blocks/CU %d chunk %2d | x2: %.4f ms %6.0f GB/s | x4: %.4f ms %6.0f GB/s | x4 depth2: %.4f ms %6.0f GB/s | x4 nt: %.4f ms %6.0f GB/s\n",
                   bpc, chunk, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3, t3, gb / t3 * 1e3);
        }
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float t = time_ms([&] { hipLaunchKernelGGL(plain_read, dim3(blocks), dim3(256), 0, 0, (const float4 *)in, out, n * 128); }, 20);
        printf("plain float4 grid-stride read, %d blocks: %.4f ms %6.0f GB/s\n", blocks, t, n * 2048.0 / t / 1e6);
    }
    return 0;
}
