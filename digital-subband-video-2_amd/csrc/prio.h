// prio.h -- what every streaming kernel does first.  Two build switches, both off in the product build (no instruction emitted):
//
//  * make prio   (-DDSV2_STREAM_PRIO=n): issue priority of the streaming kernels' wavefronts.  The motion search's wavefronts run at
//    priority 0; a value > 0 lets the short, memory-bound kernels of the other lockstep groups issue ahead of them on a shared SIMD.
//  * make census (-DDSV2_CENSUS): RESIDENCY UNDER LOAD.  rocprofv3's counters run the dispatches one at a time, so they cannot say how
//    many wavefronts a kernel keeps resident while four lockstep groups share the chip.  This does: the first thread of every
//    workgroup reads the 100 MHz real-time counter when it starts and when it leaves and adds (ticks x wavefronts of the group) to a
//    per-kernel-site tally, sharded 64 ways by compute unit so that the adds do not queue on one address.  Sum of ticks / 1e8 /
//    elapsed seconds / 1 024 SIMDs = mean resident wavefronts per SIMD of that kernel over the timed region -- SQ_WAVE_CYCLES read
//    in software, un-serialised.  Tallies are per translation unit (no relocatable device code in this build): each .hip registers
//    a reader with the host registry in dev.cpp; dsv2hip_census_read() returns them as (file, line, ticks, groups, wavefronts).
//    The persistent search kernels open their scope by hand (hme.hip).  bench.py reports it when DSV2_CENSUS=1
//    (tools/profile_round.sh part `census`).
#pragma once
#ifndef DSV2_STREAM_PRIO
#define DSV2_STREAM_PRIO 0
#endif
#ifdef DSV2_CENSUS
#include <hip/hip_runtime.h>
namespace dsv2 {
namespace census {
constexpr int kSites = 64, kShards = 64;
struct Tally {
    unsigned long long ticks, groups, waves, line;
};
static __device__ Tally g_tab[kSites][kShards];
struct Scope {
    int site, line;
    unsigned long long t0;
    bool mine;
    __device__ __forceinline__ Scope(int site_, int line_) : site(site_), line(line_), t0(0)
    {
        mine = (threadIdx.x | threadIdx.y | threadIdx.z) == 0;
        if (mine) {
            t0 = wall_clock64();
        }
    }
    __device__ __forceinline__ ~Scope()
    {
        if (mine) {
            const unsigned long long dt = wall_clock64() - t0;
            const unsigned nw = (blockDim.x * blockDim.y * blockDim.z + 63u) >> 6;
            // HW_ID: bits 8..11 CU id, 12 SH id, 13..15 SE id; XCC_ID bits 0..3 -- any spread over 64 shards will do
            const unsigned hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4); // hwreg(HW_REG_HW_ID, 0, 16)
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
            Tally *t = &g_tab[site][((hw >> 8) & 7u) | (xcc << 3)];
            atomicAdd(&t->ticks, dt * nw);
            atomicAdd(&t->groups, 1ull);
            atomicAdd(&t->waves, (unsigned long long) nw);
            t->line = (unsigned long long) line;
        }
    }
};
typedef void (*read_fn)(Tally *out);
typedef void (*reset_fn)();
void register_tu(const char *file, read_fn rd, reset_fn rs); // dev.cpp
static void read_tu(Tally *out)
{
    static Tally h[kSites][kShards];
    (void) hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tab), sizeof(h));
    for (int s = 0; s < kSites; s++) {
        Tally a = {0, 0, 0, 0};
        for (int k = 0; k < kShards; k++) {
            a.ticks += h[s][k].ticks;
            a.groups += h[s][k].groups;
            a.waves += h[s][k].waves;
            a.line = h[s][k].line ? h[s][k].line : a.line;
        }
        out[s] = a;
    }
}
static void reset_tu()
{
    static Tally z[kSites][kShards];
    (void) hipMemcpyToSymbol(HIP_SYMBOL(g_tab), z, sizeof(z));
}
static struct Registrar {
    Registrar() { register_tu(__BASE_FILE__, read_tu, reset_tu); }
} g_registrar;
} // namespace census
} // namespace dsv2
#define DSV2_CENSUS_SCOPE() dsv2::census::Scope census_scope_(__COUNTER__ % dsv2::census::kSites, __LINE__)
#else
#define DSV2_CENSUS_SCOPE() \
    do {                    \
    } while (0)
#endif
#define DSV2_KERNEL_PRIO()                                    \
    DSV2_CENSUS_SCOPE();                                      \
    do {                                                      \
        if (DSV2_STREAM_PRIO) {                               \
            __builtin_amdgcn_s_setprio(DSV2_STREAM_PRIO);     \
        }                                                     \
    } while (0)
