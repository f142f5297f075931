// Issue rate of one SIMD as a wavefront sees it (gfx950): clocks per vector instruction for a dependent chain and for
// four independent chains, with 1, 2 and 4 wavefronts per SIMD; plain 32-bit adds, packed 16-bit adds, v_sad_u8, DPP moves.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/issue_rate.cpp -o tools/probe/issue_rate && tools/probe/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N (1 << 20)
template <int KIND, int ILP> __global__ void k(unsigned long long *out, int seed)
{
    unsigned a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
        for (int u = 0; u < 16 / ILP; u++) {
            if (KIND == 0) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(seed));
                if (ILP >= 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(seed));
                if (ILP >= 4) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(seed)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(seed)); }
            } else if (KIND == 1) {
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(seed));
                if (ILP >= 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(b) : "v"(seed));
                if (ILP >= 4) { asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c) : "v"(seed)); asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(d) : "v"(seed)); }
            } else if (KIND == 2) {
                asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(a) : "v"(seed));
                if (ILP >= 2) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(b) : "v"(seed));
                if (ILP >= 4) { asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(c) : "v"(seed)); asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(d) : "v"(seed)); }
            } else {
                asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
                if (ILP >= 2) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(b));
                if (ILP >= 4) { asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c)); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d)); }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a + b + c + d == 0x12345) out[1] = a;
}
template <int KIND, int ILP> void run(const char *name, unsigned long long *d)
{
    for (int waves : {1, 2, 4, 8}) { // wavefronts per SIMD: one workgroup of 4 * waves wavefronts on one CU
        hipLaunchKernelGGL((k<KIND, ILP>), dim3(1), dim3(64 * 4 * waves), 0, 0, d, 1);
        (void) hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void) hipEventCreate(&e0);
        (void) hipEventCreate(&e1);
        (void) hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND, ILP>), dim3(1), dim3(64 * 4 * waves), 0, 0, d, 1);
        (void) hipEventRecord(e1, 0);
        (void) hipEventSynchronize(e1);
        float ms = 0;
        (void) hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t = 0;
        (void) hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        printf("%-14s ILP %d  %d wave(s)/SIMD: %.2f memtime ticks = %.2f ns per instruction of one wavefront (%.2f clocks at 2.4 GHz)\n", name, ILP, waves, (double) t / N,
               1e6 * ms / N, 2.4e6 * ms / N);
    }
}
int main()
{
    unsigned long long *d;
    (void) hipMalloc(&d, 64);
    int clk = 0, mclk = 0;
    (void) hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    (void) hipDeviceGetAttribute(&mclk, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock %d kHz (s_memtime counts the constant 100 MHz clock on this part if ticks look ~24x small)\n", clk, mclk);
    run<0, 1>("v_add_u32", d); run<0, 4>("v_add_u32", d);
    run<0, 2>("v_add_u32", d);
    run<2, 1>("v_sad_u8", d); run<3, 1>("v_mov_dpp", d);
    return 0;
}
