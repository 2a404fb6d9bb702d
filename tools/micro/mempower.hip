// mempower.hip -- energy of the memory walk (development aid): runs ONE read pattern for a few seconds so that rocm-smi can
// sample clock and package power beside it (tools/micro/mempower.sh).  Patterns over 1 M x 2 KiB frames:
//   0 plain float4 grid-stride read        1 the MFCC kernel's walk: 4 x dwordx2 nontemporal, two frames in flight, 13-dword store per frame
//   2 pattern 1 with cacheable loads       3 the walk with 2 x dwordx4 nontemporal      4 pattern 1 without the stores
//   hipcc --offload-arch=gfx950 -O3 -o mempower mempower.hip && ./mempower PATTERN SECONDS
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

int main(int argc, char **argv)
{
    const int pattern = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 6.0;
    const long n = 1000000;
    float *in, *out;
    if (hipMalloc(&in, n * 512 * sizeof(float)) != hipSuccess || hipMalloc(&out, n * 13 * sizeof(float)) != hipSuccess) return 1;
    (void)hipMemset(in, 0x3c, n * 512 * sizeof(float));
    auto launch = [&] {
        switch (pattern) {
        case 0: hipLaunchKernelGGL(plain_read, dim3(2048), dim3(256), 0, 0, (const float4 *)in, out, n * 128); break;
        case 1: hipLaunchKernelGGL((walk<true, false, true>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        case 2: hipLaunchKernelGGL((walk<false, false, true>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        case 3: hipLaunchKernelGGL((walk<true, true, true>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
        default: hipLaunchKernelGGL((walk<true, false, false>), dim3(1024), dim3(256), 0, 0, in, out, n, 8); break;
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
