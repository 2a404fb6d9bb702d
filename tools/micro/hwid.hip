// hwid.hip -- where do the waves of a launch land?  (development aid)  Every wave records s_getreg(HW_ID) and XCC_ID;
// geometry as iir2_ckpt_kernel's: 768 blocks of 192 threads with 34 KB of LDS, blocks kept alive long enough to be co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(192) void k(unsigned *out, int spin)
{
    __shared__ float pad[8448];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    pad[threadIdx.x] = 0;
    unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
    float a = lane;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (lane == 0) { out[(blockIdx.x * 3 + w) * 2] = id; out[(blockIdx.x * 3 + w) * 2 + 1] = xcc + (a == 0.123f); }
    pad[threadIdx.x] = a;
}
int main()
{
    const int blocks = 768;
    unsigned *d; hipMalloc(&d, blocks * 3 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(192), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 3 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; ++b) { printf("block %3d:", b); for (int w = 0; w < 3; ++w) printf("  hw_id %08x xcc %x", h[(b * 3 + w) * 2], h[(b * 3 + w) * 2 + 1]); printf("\n"); }
    for (int b : {256, 257, 512, 513}) { printf("block %3d:", b); for (int w = 0; w < 3; ++w) printf("  hw_id %08x xcc %x", h[(b * 3 + w) * 2], h[(b * 3 + w) * 2 + 1]); printf("\n"); }
    // per (xcc, se, cu): waves per simd
    std::map<unsigned, std::vector<int>> cu;
    for (int i = 0; i < blocks * 3; ++i) {
        const unsigned id = h[i * 2], xcc = h[i * 2 + 1] & 0xf;
        const unsigned simd = (id >> 4) & 3, cuid = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
        auto &v = cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid];
        if (v.empty()) v.assign(4, 0);
        v[simd]++;
    }
    std::map<std::vector<int>, int> hist;
    for (auto &kv : cu) hist[kv.second]++;
    printf("%zu distinct (xcc, se, sh, cu); waves per SIMD patterns:\n", cu.size());
    for (auto &kv : hist) printf("  [%d %d %d %d] x %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    return 0;
}
