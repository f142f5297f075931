// Which compute units does a CU-masked stream use?  For mask patterns over the device's CU bits, a kernel of many one-wave
// workgroups records (XCC id, SE, SH, CU) of every workgroup; the probe prints how many distinct CUs of which XCC were used.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/cumask_probe.cpp -o tools/probe/cumask_probe && tools/probe/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void where(unsigned *out)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { // 20 us: long enough for every allowed CU to be handed workgroups
    }
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);  // HW_ID: cu 8..11, sh 12, se 13..15
        unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
        out[blockIdx.x] = (xcc << 16) | (hw & 0xff00u);
    }
}
static void show(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t) mask.size(), mask.data()));
    const int n = 8192;
    unsigned *d, *h = (unsigned *) malloc(n * 4);
    CK(hipMalloc((void **) &d, n * 4));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per;
    for (int i = 0; i < n; i++) per[h[i] >> 16].insert(h[i] & 0xffffu);
    size_t tot = 0;
    printf("%-28s:", name);
    for (auto &kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  = %zu CUs\n", tot);
    if (getenv("VERBOSE")) {
        for (auto &kv : per) { printf("    xcc%u:", kv.first); for (unsigned c : kv.second) printf(" se%u.sh%u.cu%u", (c >> 13) & 7, (c >> 12) & 1, (c >> 8) & 15); printf("\n"); }
    }
    CK(hipFree(d));
    free(h);
    CK(hipStreamDestroy(s));
}
int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> m((size_t) words, 0xffffffffu);
    show("all", m);
    std::vector<uint32_t> a((size_t) words, 0u);
    for (int i = 0; i < 32; i++) a[(size_t) i / 32] |= 1u << (i % 32);
    show("bits 0..31", a);
    std::fill(a.begin(), a.end(), 0u);
    for (int i = 0; i < 8; i++) a[0] |= 1u << i;
    show("bits 0..7", a);
    std::fill(a.begin(), a.end(), 0u);
    for (int i = 0; i < p.multiProcessorCount / 2; i++) a[(size_t) i / 32] |= 1u << (i % 32);
    show("first half of the bits", a);
    std::fill(a.begin(), a.end(), 0u);
    for (int i = 0; i < p.multiProcessorCount; i += 2) a[(size_t) i / 32] |= 1u << (i % 32);
    show("even bits", a);
    std::fill(a.begin(), a.end(), 0u);
    for (int i = 0; i < p.multiProcessorCount; i++) if ((i / 8) % 2 == 0) a[(size_t) i / 32] |= 1u << (i % 32);
    show("bits with (i/8)%2==0", a);
    std::fill(a.begin(), a.end(), 0u);
    for (int i = 0; i < p.multiProcessorCount; i++) if ((i / 16) % 2 == 0) a[(size_t) i / 32] |= 1u << (i % 32);
    show("bits with (i/16)%2==0", a);
    return 0;
}
