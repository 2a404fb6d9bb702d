// valubench.hip -- issue cost of the VALU instruction kinds the MFCC kernel uses
// (development aid).  Each kernel runs a long unrolled stream of one instruction
// kind on 8 independent registers; WAVES_PER_SIMD waves resident per SIMD.
// Reports SIMD-cycles per wave-instruction assuming a 2.1 GHz clock is NOT assumed:
// we print ns per instruction per SIMD and the ratio to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int REPS = 64;     // unrolled instructions per register per loop trip
constexpr int TRIPS = 200;

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float r[8];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 8; ++i) r[i] = seed + i + lane;
    const bool odd = lane & 1;
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
        for (int j = 0; j < REPS; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) r[i] = fmaf(r[i], 1.0001f, 0.5f);
                else if (KIND == 1) r[i] = r[i] + 0.5f;
                else if (KIND == 2) r[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r[i]), 0xB1, 0xF, 0xF, true));           // quad_perm mov
                else if (KIND == 3) r[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(r[i]), __float_as_int(r[(i + 1) & 7]), 0x128, 0xF, 0x3, false)); // row_ror:8 masked
                else if (KIND == 4) { auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[i]), __float_as_uint(r[(i + 1) & 7]), false, false); r[i] = __uint_as_float(p[0]); r[(i + 1) & 7] = __uint_as_float(p[1]); }
                else if (KIND == 5) { auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[i]), __float_as_uint(r[(i + 1) & 7]), false, false); r[i] = __uint_as_float(p[0]); r[(i + 1) & 7] = __uint_as_float(p[1]); }
                else if (KIND == 6) r[i] = odd ? r[(i + 1) & 7] : r[i];                       // v_cndmask
                else if (KIND == 7) r[i] = __builtin_amdgcn_logf(r[i]);
                else if (KIND == 8) r[i] = __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 5) << 2, __float_as_int(r[i])));
                else if (KIND == 9) r[i] = r[i] * r[(i + 3) & 7];
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(float *out, int blocks_per_cu, const char *name, double base)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double insts_per_simd = (double)blocks_per_cu * 1 /*waves per SIMD per block*/ * TRIPS * REPS * 8;
    const double ns = t[2] * 1e6 / insts_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms  %.3f ns per wave-instruction per SIMD  (x%.2f of fma)\n", name, blocks_per_cu, t[2], ns, base > 0 ? ns / base : 1.0);
    return ns;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 5}) {
        double base = run<0>(out, w, "v_fma_f32", 0);
        run<1>(out, w, "v_add_f32", base);
        run<9>(out, w, "v_mul_f32 (2 vgpr src)", base);
        run<2>(out, w, "v_mov_dpp quad_perm", base);
        run<3>(out, w, "v_mov_dpp row_ror bank_mask", base);
        run<4>(out, w, "v_permlane32_swap", base);
        run<5>(out, w, "v_permlane16_swap", base);
        run<6>(out, w, "v_cndmask", base);
        run<7>(out, w, "v_log_f32", base);
        run<8>(out, w, "ds_bpermute_b32", base);
    }
    return 0;
}
