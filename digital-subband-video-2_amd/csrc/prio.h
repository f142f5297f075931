// prio.h -- issue priority of the streaming kernels' wavefronts (A/B build switch: make prio -> -DDSV2_STREAM_PRIO=n).
// The motion search's wavefronts run at priority 0; a value > 0 here lets the short, memory-bound kernels of the other lockstep
// groups issue ahead of them on a shared SIMD.  Default 0: no instruction is emitted.
#pragma once
#ifndef DSV2_STREAM_PRIO
#define DSV2_STREAM_PRIO 0
#endif
#define DSV2_KERNEL_PRIO()                                    \
    do {                                                      \
        if (DSV2_STREAM_PRIO) {                               \
            __builtin_amdgcn_s_setprio(DSV2_STREAM_PRIO);     \
        }                                                     \
    } while (0)
