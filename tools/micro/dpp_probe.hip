#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__global__ void k(int *out)
{
    int v = threadIdx.x + 100;
    int r = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
    out[threadIdx.x] = r;
}
template <int CTRL>
void run(const char *name)
{
    int *d; (void)hipMalloc(&d, 64 * 4);
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    int h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s:", name);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\n");
}
int main()
{
    run<0x138>("wave_shr:1");
    run<0x112>("row_shr:2");
    run<0x118>("row_shr:8");
    run<0x142>("row_bcast:15");
    run<0x143>("row_bcast:31");
    return 0;
}
