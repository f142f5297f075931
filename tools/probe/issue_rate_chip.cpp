// Aggregate vector issue rate of the whole chip against wavefronts per SIMD (gfx950): k workgroups of four wavefronts per CU, every
// wavefront a stream of v_add_u32 (one dependent chain / four independent ones).  Companion of issue_rate.cpp (one CU).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/issue_rate_chip.cpp -o /tmp/issue_rate_chip && /tmp/issue_rate_chip
// (clocks are 2.4 GHz nominal: with every CU busy the part clocks lower -- the one-wavefront dependent chain reads 10.5 here, 9.0 alone)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N (1 << 16)
template <int ILP> __global__ __launch_bounds__(256) void kern(unsigned long long *out, int seed)
{
    unsigned a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7;
#pragma unroll 1
    for (int i = 0; i < N / 16; i++) {
#pragma unroll
        for (int u = 0; u < 16 / ILP; u++) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(seed));
            if (ILP >= 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(seed));
            if (ILP >= 4) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(seed)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(seed)); }
        }
    }
    if (a + b + c + d == 0x12345) out[1] = a;
}
template <int ILP> void run(unsigned long long *d)
{
    for (int nw : {1, 2, 3, 4, 6, 8}) { const int k = nw; // k workgroups of 4 wavefronts per CU = k wavefronts per SIMD when the dispatcher spreads them evenly
        hipLaunchKernelGGL((kern<ILP>), dim3(256 * k), dim3(256), 0, 0, d, 1);
        (void) hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
        (void) hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kern<ILP>), dim3(256 * k), dim3(256), 0, 0, d, 1);
        (void) hipEventRecord(e1, 0);
        (void) hipEventSynchronize(e1);
        float ms = 0;
        (void) hipEventElapsedTime(&ms, e0, e1);
        const double clocks = 2.4e6 * ms;               // per launch
        const double per_wave = clocks / N;             // clocks per counted instruction of one wavefront (if all run at once)
        printf("ILP %d  %d wavefront(s) per SIMD: %.2f clocks per instruction per wavefront -> %.3f vector instructions per clock per SIMD\n", ILP, k, per_wave, k / per_wave);
    }
}
int main()
{
    unsigned long long *d;
    (void) hipMalloc(&d, 64);
    run<1>(d);
    run<4>(d);
    return 0;
}
