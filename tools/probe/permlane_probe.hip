// Probe: prints what v_permlane32_swap / v_permlane16_swap return on gfx950 (which lanes of which operand).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o)
{
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    uint2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    uint2v q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    o[128 + threadIdx.x] = q[0];
    o[192 + threadIdx.x] = q[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"swap32[0]", "swap32[1]", "swap16[0]", "swap16[1]"};
    for (int r = 0; r < 4; r++) {
        printf("%s:", names[r]);
        for (int i = 0; i < 64; i += 8) printf(" %u", h[64 * r + i]);
        printf("\n");
    }
    return 0;
}
