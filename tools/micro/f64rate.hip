// f64rate.hip -- what a float64 VALU instruction costs on gfx950 (development aid): v_add_f64 / v_mul_f64 / v_fma_f64 beside
// v_add_f32 / v_fma_f32, as eight independent registers (issue rate) and as one dependent chain (latency), one wave per SIMD and two.
//   hipcc --offload-arch=gfx950 -O3 -o f64rate f64rate.hip && ./f64rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int REPS = 32, TRIPS = 400;

#define OP8(fmt, T) \
    asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[1]) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r[2]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[3]) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r[4]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[5]) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r[6]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[7]) : "v"(a), "v"(b));
#define OP1(fmt, T) \
    asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r[0]) : "v"(a), "v"(b));

template <typename T, int KIND, bool DEP>
__global__ __launch_bounds__(256) void k(T *out, T seed)
{
    const int lane = threadIdx.x & 63;
    T r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed + (T)(lane + i);
    T a = (T)1.0000001 + (T)lane * (T)1e-9, b = (T)0.5 + (T)lane * (T)1e-9;
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
        for (int j = 0; j < REPS; ++j) {
            if (KIND == 0) { if (DEP) { OP1("v_add_f64 %0, %1, %0", T) } else { OP8("v_add_f64 %0, %1, %0", T) } }
            else if (KIND == 1) { if (DEP) { OP1("v_mul_f64 %0, %1, %0", T) } else { OP8("v_mul_f64 %0, %1, %0", T) } }
            else if (KIND == 2) { if (DEP) { OP1("v_fma_f64 %0, %1, %2, %0", T) } else { OP8("v_fma_f64 %0, %1, %2, %0", T) } }
            else if (KIND == 3) { if (DEP) { OP1("v_add_f32_e32 %0, %1, %0", T) } else { OP8("v_add_f32_e32 %0, %1, %0", T) } }
            else if (KIND == 4) { if (DEP) { OP1("v_fma_f32 %0, %1, %2, %0", T) } else { OP8("v_fma_f32 %0, %1, %2, %0", T) } }
            else if (KIND == 5) { if (DEP) { OP1("v_mul_f32_e32 %0, %1, %0", T) } else { OP8("v_mul_f32_e32 %0, %1, %0", T) } }
        }
    }
    T s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T, int KIND, bool DEP>
void run(void *out, int w, const char *name)
{
    const int blocks = 256 * w;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, KIND, DEP>), dim3(blocks), dim3(256), 0, 0, (T *)out, (T)1);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<T, KIND, DEP>), dim3(blocks), dim3(256), 0, 0, (T *)out, (T)1);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double ns = t[2] * 1e6 / ((double)w * TRIPS * REPS * 8);
    printf("%-12s %-11s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, DEP ? "dependent" : "independent", w, ns, ns * 2.4);
}

int main()
{
    void *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    for (int w : {1, 2}) {
        run<double, 0, false>(out, w, "v_add_f64"); run<double, 0, true>(out, w, "v_add_f64");
        run<double, 1, false>(out, w, "v_mul_f64"); run<double, 1, true>(out, w, "v_mul_f64");
        run<double, 2, false>(out, w, "v_fma_f64"); run<double, 2, true>(out, w, "v_fma_f64");
        run<float, 3, false>(out, w, "v_add_f32"); run<float, 3, true>(out, w, "v_add_f32");
        run<float, 5, false>(out, w, "v_mul_f32"); run<float, 5, true>(out, w, "v_mul_f32");
        run<float, 4, false>(out, w, "v_fma_f32"); run<float, 4, true>(out, w, "v_fma_f32");
    }
    return 0;
}
