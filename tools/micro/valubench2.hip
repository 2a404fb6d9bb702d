// valubench2.hip -- encoding-level issue cost on gfx950 (development aid): VOP2 (4-byte)
// vs VOP3 / DPP (8-byte) forms, via inline asm on 8 independent registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int REPS = 64, TRIPS = 200;

#define OP8(fmt) \
    asm volatile(fmt : "+v"(r0) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r1) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r2) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r3) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r4) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r5) : "v"(a), "v"(b)); \
    asm volatile(fmt : "+v"(r6) : "v"(a), "v"(b)); asm volatile(fmt : "+v"(r7) : "v"(a), "v"(b));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    const int lane = threadIdx.x & 63;
    float r0 = seed + lane, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f + lane * 1e-7f, b = 0.5f + lane * 1e-7f;
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
        for (int j = 0; j < REPS; ++j) {
            if (KIND == 0) { OP8("v_fmac_f32_e32 %0, %1, %2") }
            else if (KIND == 1) { OP8("v_fma_f32 %0, %1, %2, %0") }
            else if (KIND == 2) { OP8("v_cndmask_b32_e32 %0, %0, %1, vcc") }
            else if (KIND == 3) { OP8("v_cndmask_b32_e64 %0, %0, %1, vcc") }
            else if (KIND == 4) { OP8("v_add_f32_e32 %0, %1, %0") }
            else if (KIND == 5) { OP8("v_add_f32_e64 %0, %1, %0") }
            else if (KIND == 6) { OP8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
            else if (KIND == 7) { OP8("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
            else if (KIND == 8) { OP8("v_mul_f32_e32 %0, %1, %0") }
            else if (KIND == 9) { OP8("v_fma_f32 %0, %1, %2, -%0") }
            else if (KIND == 11) { OP8("v_sub_f32_e32 %0, %1, %0") }
            else if (KIND == 12) { OP8("v_max_f32_e32 %0, %1, %0") }
            else if (KIND == 13) { OP8("v_mov_b32_e32 %0, %1") }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

template <int KIND>
double run(float *out, int w, const char *name, double base)
{
    const int blocks = 256 * w;
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
    const double ns = t[2] * 1e6 / ((double)w * TRIPS * REPS * 8);
    printf("%-34s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (x%.2f)\n", name, w, ns, base > 0 ? ns / base : 1.0);
    return ns;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {2, 5, 8}) {
        double base = run<4>(out, w, "v_add_f32_e32 (VOP2)", 0);
        run<11>(out, w, "v_sub_f32_e32 (VOP2)", base);
        run<8>(out, w, "v_mul_f32_e32 (VOP2)", base);
        run<12>(out, w, "v_max_f32_e32 (VOP2)", base);
        run<13>(out, w, "v_mov_b32_e32 (VOP1)", base);
        run<0>(out, w, "v_fmac_f32_e32 (VOP2)", base);
        run<1>(out, w, "v_fma_f32 3 vgpr (VOP3)", base);
        run<9>(out, w, "v_fma_f32 3 vgpr neg (VOP3)", base);
        run<5>(out, w, "v_add_f32_e64 (VOP3)", base);
        run<2>(out, w, "v_cndmask_b32_e32 vcc (VOP2)", base);
        run<3>(out, w, "v_cndmask_b32_e64 (VOP3)", base);
        run<6>(out, w, "v_mov_b32_dpp", base);
        run<7>(out, w, "v_add_f32_dpp", base);
    }
    return 0;
}
