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
// MODE 4: 4 x dwordx2, nontemporal loads (the MFCC kernel's pattern), prefetch depth 2
// MODE 5 / 6: LDS-DMA (global_load_lds_dwordx4, default / nt policy) into a 3-frame per-wave ring, 2 frames
//             ahead, read back as z[l + 64 a] with ds_read_b64
typedef float f2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void stream(const float *__restrict__ in, float *__restrict__ out, long n_frames, int chunk)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * 4;
    for (long c0 = wave * chunk; c0 < n_frames; c0 += n_waves * chunk) {
        const long c1 = c0 + chunk < n_frames ? c0 + chunk : n_frames;
        if (MODE == 4) {
            f2v n1[4], n2[4];
            for (int a = 0; a < 4; ++a) n1[a] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(in + c0 * 512 + 2 * (lane + 64 * a)));
            if (c0 + 1 < c1)
                for (int a = 0; a < 4; ++a) n2[a] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(in + (c0 + 1) * 512 + 2 * (lane + 64 * a)));
            for (long f = c0; f < c1; ++f) {
                f2v cur[4];
                for (int a = 0; a < 4; ++a) { cur[a] = n1[a]; n1[a] = n2[a]; }
                if (f + 2 < c1)
                    for (int a = 0; a < 4; ++a) n2[a] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(in + (f + 2) * 512 + 2 * (lane + 64 * a)));
                float s = (cur[0].x + cur[0].y) + (cur[1].x + cur[1].y) + (cur[2].x + cur[2].y) + (cur[3].x + cur[3].y);
                s += __shfl_xor(s, 1);
                if (lane < 13) out[f * 13 + lane] = s;
            }
        } else if (MODE == 5 || MODE == 6) {
            extern __shared__ __attribute__((aligned(16))) char smem[];
            char *ring = smem + (threadIdx.x >> 6) * (3 * 2048);
            constexpr int AUX = MODE == 6 ? 2 : 0;
            auto dma = [&](long f, int slot) {
                const float *src = in + f * 512 + 4 * lane;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(ring + slot * 2048), 16, 0, AUX);
                __builtin_amdgcn_global_load_lds(src + 256, (__attribute__((address_space(3))) void *)(ring + slot * 2048 + 1024), 16, 0, AUX);
            };
            int slot = 0;
            dma(c0, 0);
            if (c0 + 1 < c1) dma(c0 + 1, 1);
            for (long f = c0; f < c1; ++f) {
                const int nslot = slot + 2 >= 3 ? slot - 1 : slot + 2;
                if (f + 2 < c1) { dma(f + 2, nslot); __builtin_amdgcn_s_waitcnt(0x0F74); }       // vmcnt(4)
                else if (f + 1 < c1) __builtin_amdgcn_s_waitcnt(0x0F72);                          // vmcnt(2)
                else __builtin_amdgcn_s_waitcnt(0x0F70);                                          // vmcnt(0)
                // inline asm: hipcc would put s_waitcnt vmcnt(0) before an ordinary LDS read while a DMA is in flight
                const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(ring + slot * 2048) + 8 * lane;
                f2v cur[4];
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\t"
                             "ds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(cur[0]), "=&v"(cur[1]), "=&v"(cur[2]), "=&v"(cur[3]) : "v"(addr) : "memory");
                float s = (cur[0].x + cur[0].y) + (cur[1].x + cur[1].y) + (cur[2].x + cur[2].y) + (cur[3].x + cur[3].y);
                s += __shfl_xor(s, 1);
                if (lane < 13) out[f * 13 + lane] = s;
                slot = slot + 1 >= 3 ? 0 : slot + 1;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
        } else if (MODE == 0) {
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

// MODE 7 / 8: the kernel's walk with prefetch ACROSS chunks (depth 2): 4 x dwordx2 nt / 2 x dwordx4 nt, so small
// chunks (down to one frame) can be compared; the 13-float result of 16 frames is stored at once, like the tile epilogue
template <int MODE, int SM>
__global__ __launch_bounds__(256) void walk(const float *__restrict__ in, float *__restrict__ out, long n_frames, int chunk, long wrap_mask = -1)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long n_waves = (long)gridDim.x * 4;
    const int ch = chunk;
    auto frame_of = [&](long i) { return ((i / ch) * n_waves + wave) * ch + i % ch; };
    constexpr int W = MODE == 7 ? 4 : 2;
    typedef float vt __attribute__((ext_vector_type(MODE == 7 ? 2 : 4)));
    auto load = [&](long f, vt (&r)[W]) {
        for (int a = 0; a < W; ++a)
            r[a] = __builtin_nontemporal_load(reinterpret_cast<const vt *>(in + f * 512 + (MODE == 7 ? 2 : 4) * lane + (512 / W) * a));
    };
    vt n1[W] = {}, n2[W] = {};
    long i = 0, f0 = frame_of(0), f1 = frame_of(1), f2;
    if (f0 >= n_frames) return;
    load(f0, n1);
    if (f1 < n_frames) load(f1, n2);
    float keep = 0.f;
    for (;; ++i) {
        vt cur[W];
        for (int a = 0; a < W; ++a) { cur[a] = n1[a]; n1[a] = n2[a]; }
        f2 = frame_of(i + 2);
        if (f2 < n_frames) load(f2, n2);
        float s = 0.f;
        for (int a = 0; a < W; ++a) s += MODE == 7 ? cur[a][0] + cur[a][1] : (cur[a][0] + cur[a][1]) + (cur[a][2] + cur[a][3]);
        s += __shfl_xor(s, 1);
        keep = (i & 15) == (lane >> 2) ? s : keep;          // lane group g keeps frame g of the tile
        if ((i & 15) == 15 || f1 >= n_frames) {
            const long fi = frame_of((i & ~15L) + (lane >> 2));
            if (SM == 1 || SM == 2) {          // scattered dword stores (the MFMA output layout), plain / nontemporal
                if (fi < n_frames && (lane >> 2) <= (i & 15)) {
                    float *o = out + fi * 13 + (lane & 3);
                    if (SM == 1) { o[0] = keep; o[4] = keep; o[8] = keep; if ((lane & 3) == 0) o[12] = keep; }
                    else { __builtin_nontemporal_store(keep, o); __builtin_nontemporal_store(keep, o + 4); __builtin_nontemporal_store(keep, o + 8); if ((lane & 3) == 0) __builtin_nontemporal_store(keep, o + 12); }
                }
            } else if (SM == 3 || SM == 4) {   // the tile's 16 x 13 floats as ONE contiguous dwordx4 store (52 lanes), plain / nt
                const long t0 = frame_of(i & ~15L);
                if (lane < 52 && t0 + 15 < n_frames && ch % 16 == 0) {
                    f4v v = {keep, keep, keep, keep};
                    f4v *o = reinterpret_cast<f4v *>(out + (t0 & wrap_mask) * 13) + lane;
                    if (SM == 3) *o = v; else __builtin_nontemporal_store(v, o);
                }
            } else if (keep == 123.456f) out[0] = keep;
        }
        if (f1 >= n_frames) break;
        f0 = f1; f1 = f2;
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
            float t4 = time_ms([&] { hipLaunchKernelGGL(stream<4>, dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
            float t5 = time_ms([&] { hipLaunchKernelGGL(stream<5>, dim3(blocks), dim3(256), 4 * 3 * 2048, 0, in, out, n, chunk); }, 20);
            float t6 = time_ms([&] { hipLaunchKernelGGL(stream<6>, dim3(blocks), dim3(256), 4 * 3 * 2048, 0, in, out, n, chunk); }, 20);
            printf("blocks/CU %d chunk %2d | x2: %.4f ms %6.0f GB/s | x4: %.4f ms %6.0f GB/s | x4 depth2: %.4f ms %6.0f GB/s | x4 nt: %.4f ms %6.0f GB/s\n"
                   "                      | x2 nt depth2: %.4f ms %6.0f GB/s | lds-dma: %.4f ms %6.0f GB/s | lds-dma nt: %.4f ms %6.0f GB/s\n",
                   bpc, chunk, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3, t3, gb / t3 * 1e3, t4, gb / t4 * 1e3, t5, gb / t5 * 1e3,
                   t6, gb / t6 * 1e3);
        }
    }
    for (int chunk : {8, 16, 32, 64}) {
        const int blocks = 256 * 4;
        float t[5];
        t[0] = time_ms([&] { hipLaunchKernelGGL((walk<8, 0>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
        t[1] = time_ms([&] { hipLaunchKernelGGL((walk<8, 1>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
        t[2] = time_ms([&] { hipLaunchKernelGGL((walk<8, 2>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
        t[3] = time_ms([&] { hipLaunchKernelGGL((walk<8, 3>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
        t[4] = time_ms([&] { hipLaunchKernelGGL((walk<8, 4>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
        float u = time_ms([&] { hipLaunchKernelGGL((walk<7, 3>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk); }, 20);
        float w1 = time_ms([&] { hipLaunchKernelGGL((walk<8, 3>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk, 0xFFFFL); }, 20);
        float w2 = time_ms([&] { hipLaunchKernelGGL((walk<8, 3>), dim3(blocks), dim3(256), 0, 0, in, out, n, chunk, 0xFFFFFL); }, 20);
        printf("   contiguous stores wrapped into 64 Ki frames (3.4 MB): %.4f ms; into 1 Mi frames (54 MB, all): %.4f ms\n", w1, w2);
        printf("walk x4 nt chunk %2d | no stores %.4f | scattered %.4f | scattered nt %.4f | contiguous %.4f | contiguous nt %.4f | x2 contiguous %.4f ms\n",
               chunk, t[0], t[1], t[2], t[3], t[4], u);
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float t = time_ms([&] { hipLaunchKernelGGL(plain_read, dim3(blocks), dim3(256), 0, 0, (const float4 *)in, out, n * 128); }, 20);
        printf("plain float4 grid-stride read, %d blocks: %.4f ms %6.0f GB/s\n", blocks, t, n * 2048.0 / t / 1e6);
    }
    return 0;
}
