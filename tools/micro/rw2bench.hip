// rw2bench.hip -- what HBM gives a kernel that reads one stream and writes two of the same size (the float64 IIR kernel's mix:
// 6.3 GB in, 12.6 GB out), fully coalesced 16-byte accesses, against the same bytes in the IIR kernel's shape (64 rows x 256-byte
// segments per tile).  Development aid.   hipcc --offload-arch=gfx950 -O3 -o rw2bench rw2bench.hip && ./rw2bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void copy2(const d2 *__restrict__ x, d2 *__restrict__ y1, d2 *__restrict__ y2, long n2)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
        const d2 v = __builtin_nontemporal_load(x + i);
        __builtin_nontemporal_store(v, y1 + i);
        __builtin_nontemporal_store(v * 2.0, y2 + i);
    }
}

// the IIR kernel's shape: a block owns 64 rows (clips) of n doubles and walks them in tiles of 32 doubles (256 B per row and tile)
__global__ __launch_bounds__(128) void tiles2(const double *__restrict__ x, double *__restrict__ y1, double *__restrict__ y2, int n)
{
    const long row0 = (long)blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int t0 = 0; t0 < n; t0 += 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 128 * k, r = e / 16, cc = (e % 16) * 2;
            const long off = (row0 + r) * n + t0 + cc;
            const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(x + off));
            __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(y1 + off));
            __builtin_nontemporal_store(v * 2.0, reinterpret_cast<d2 *>(y2 + off));
        }
    }
}

template <class F>
double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[3];
}

int main()
{
    const long clips = 49152; const int n = 16000;
    const long total = clips * n;
    double *x, *y1, *y2;
    if (hipMalloc(&x, total * 8) != hipSuccess || hipMalloc(&y1, total * 8) != hipSuccess || hipMalloc(&y2, total * 8) != hipSuccess) return 1;
    hipMemset(x, 0x3c, total * 8);
    const double gb = 3.0 * total * 8 / 1e9;
    for (int blocks : {2048, 4096, 8192}) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(copy2, dim3(blocks), dim3(256), 0, 0, (const d2 *)x, (d2 *)y1, (d2 *)y2, total / 2); });
        printf("coalesced read 1 + write 2, %d blocks: %.3f ms = %.2f TB/s\n", blocks, ms, gb / ms);
    }
    const double ms = time_ms([&] { hipLaunchKernelGGL(tiles2, dim3((unsigned)(clips / 64)), dim3(128), 0, 0, x, y1, y2, n); });
    printf("IIR-shaped tiles (768 blocks x 64 rows x 256 B segments): %.3f ms = %.2f TB/s\n", ms, gb / ms);
    return 0;
}
