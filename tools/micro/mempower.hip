// mempower.hip -- energy of the memory walk (development aid): runs ONE read pattern for a few seconds so that rocm-smi can
// sample clock and package power beside it (tools/micro/mempower.sh).  Patterns over 1 M x 2 KiB frames:
//   0 plain float4 grid-stride read        1 the MFCC kernel's walk: 4 x dwordx2 nontemporal, two frames in flight, 13-dword store per frame
//   2 pattern 1 with cacheable loads       3 the walk with 2 x dwordx4 nontemporal      4 pattern 1 without the stores
//   100 + aux: pattern 1 with buffer_load_dwordx2 and the cache-policy bits aux (1 = sc0, 2 = nt, 16 = sc1; gfx940 names)
//   hipcc --offload-arch=gfx950 -O3 -o mempower mempower.hip && ./mempower PATTERN SECONDS [rand]
//   rand: the input is uniform(-1, 1) noise instead of a constant (bus toggling is part of the energy)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void plain_read(const float4 *__restrict__ in, float *__restrict__ out, long n4)
{
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    if (s == 123.456f) out[0] = s;
}

template <bool NT, bool X4, bool STORE>
__global__ __launch_bounds__(256) void walk(const float *__restrict__ in, float *__restrict__ out, long n_frames, int chunk)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    auto ld = [&](long f, float (&v)[8]) {
        if (X4) {
            for (int a = 0; a < 2; ++a) {
                const f4v *p = reinterpret_cast<const f4v *>(in + f * 512 + 4 * lane + 256 * a);
                const f4v x = NT ? __builtin_nontemporal_load(p) : *p;
                v[4 * a] = x.x; v[4 * a + 1] = x.y; v[4 * a + 2] = x.z; v[4 * a + 3] = x.w;
            }
        } else {
            for (int a = 0; a < 4; ++a) {
                const f2v *p = reinterpret_cast<const f2v *>(in + f * 512 + 2 * (lane + 64 * a));
                const f2v x = NT ? __builtin_nontemporal_load(p) : *p;
                v[2 * a] = x.x; v[2 * a + 1] = x.y;
            }
        }
    };
    for (long c0 = wave * chunk; c0 < n_frames; c0 += n_waves * chunk) {
        const long c1 = c0 + chunk < n_frames ? c0 + chunk : n_frames;
        float n1[8], n2[8];
        ld(c0, n1);
        if (c0 + 1 < c1) ld(c0 + 1, n2);
        for (long f = c0; f < c1; ++f) {
            float cur[8];
            for (int a = 0; a < 8; ++a) { cur[a] = n1[a]; n1[a] = n2[a]; }
            if (f + 2 < c1) ld(f + 2, n2);
            float s = ((cur[0] + cur[1]) + (cur[2] + cur[3])) + ((cur[4] + cur[5]) + (cur[6] + cur[7]));
            s += __shfl_xor(s, 1);
            if (STORE) { if (lane < 13) out[f * 13 + lane] = s; }
            else if (s == 123.456f) out[0] = s;
        }
    }
}

typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int AUX>
__global__ __launch_bounds__(256) void walk_buf(const float *__restrict__ in, float *__restrict__ out, long n_frames, int chunk)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    auto ld = [&](long f, float (&v)[8]) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(in + f * 512), 0, 2048, 0x00027000);
        for (int a = 0; a < 4; ++a) {
            const u2v x = __builtin_amdgcn_raw_buffer_load_b64(r, 8 * (lane + 64 * a), 0, AUX);
            v[2 * a] = __uint_as_float(x.x); v[2 * a + 1] = __uint_as_float(x.y);
        }
    };
    for (long c0 = wave * chunk; c0 < n_frames; c0 += n_waves * chunk) {
        const long c1 = c0 + chunk < n_frames ? c0 + chunk : n_frames;
        float n1[8], n2[8];
        ld(c0, n1);
        if (c0 + 1 < c1) ld(c0 + 1, n2);
        for (long f = c0; f < c1; ++f) {
            float cur[8];
            for (int a = 0; a < 8; ++a) { cur[a] = n1[a]; n1[a] = n2[a]; }
            if (f + 2 < c1) ld(f + 2, n2);
            float s = ((cur[0] + cur[1]) + (cur[2] + cur[3])) + ((cur[4] + cur[5]) + (cur[6] + cur[7]));
            s += __shfl_xor(s, 1);
            if (lane < 13) out[f * 13 + lane] = s;
        }
    }
}

__global__ void fill_noise(float *p, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (float)(z >> 40) * (2.0f / 16777216.0f) - 1.0f;
    }
}

int main(int argc, char **argv)
{
    const int pattern = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
    const long n = 1000000;
    float *in, *out;
    if (hipMalloc(&in, n * 512 * sizeof(float)) != hipSuccess || hipMalloc(&out, n * 13 * sizeof(float)) != hipSuccess) return 1;
    (void)hipMemset(in, 0x3c, n * 512 * sizeof(float));
    if (argc > 3) hipLaunchKernelGGL(fill_noise, dim3(4096), dim3(256), 0, 0, in, n * 512);
    auto launch = [&] {
        switch (pattern) {
        case 0: hipLaunchKernelGGL(plain_read, dim3(2048), dim3(256), 0, 0, (const float4 *)in, out, n * 128); break;
        case 1: hipLaunchKernelGGL((walk<true, false, true>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        case 2: hipLaunchKernelGGL((walk<false, false, true>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        case 3: hipLaunchKernelGGL((walk<true, true, true>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        case 4: hipLaunchKernelGGL((walk<true, false, false>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
#define BUF(A) case 100 + A: hipLaunchKernelGGL((walk_buf<A>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        BUF(0) BUF(1) BUF(2) BUF(3) BUF(16) BUF(17) BUF(18) BUF(19)
#undef BUF
        default: break;
        }
    };
    launch();
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int i = 0; i < 200; ++i) launch();
        (void)hipDeviceSynchronize();
        launches += 200;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    printf("pattern %d: %.4f ms per pass, %.0f GB/s read (2048 B per frame)\n", pattern, el / launches * 1e3, n * 2048.0 * launches / el / 1e9);
    return 0;
}
