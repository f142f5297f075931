// Which HIP streams share a hardware queue?  K test streams each get one ~2 ms spin kernel at the same moment; if they sit on
// distinct hardware queues the whole takes ~2 ms, if two share a queue ~4 ms, and so on.  Decoy streams (created before the
// test streams, used or not) show what earlier stream creation does to the mapping.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/queue_map.cpp -o tools/probe/queue_map && tools/probe/queue_map
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void spin(long long ticks, int *sink)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static double run(std::vector<hipStream_t> &ss, long long ticks)
{
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (auto s : ss) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, (int *) nullptr);
    for (auto s : ss) CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char **argv)
{
    int ndecoy = argc > 1 ? atoi(argv[1]) : 0, use_decoy = argc > 2 ? atoi(argv[2]) : 0, ntest = argc > 3 ? atoi(argv[3]) : 4, threaded = argc > 4 ? atoi(argv[4]) : 0;
    const long long ticks = 200000; // 100 MHz wall clock: 2 ms
    std::vector<hipStream_t> decoy((size_t) ndecoy), test((size_t) ntest);
    for (auto &s : decoy) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (use_decoy) run(decoy, 1000);
    if (threaded) { // each test stream created and first used by its own thread, all at once
        std::vector<std::thread> th;
        for (int i = 0; i < ntest; i++) th.emplace_back([&, i] { CK(hipStreamCreateWithFlags(&test[(size_t) i], hipStreamNonBlocking)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, test[(size_t) i], 1000, (int *) nullptr); CK(hipStreamSynchronize(test[(size_t) i])); });
        for (auto &t : th) t.join();
    } else {
        for (auto &s : test) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    run(test, 1000);
    double a = run(test, ticks), b = run(test, ticks);
    printf("decoys %d (%s), %d test streams%s: %.2f / %.2f ms for %d concurrent 2 ms kernels", ndecoy, use_decoy ? "used" : "unused", ntest, threaded ? " (threaded creation)" : "", a, b, ntest);
    if (ndecoy) { std::vector<hipStream_t> d4(decoy.begin(), decoy.begin() + (ndecoy < ntest ? ndecoy : ntest)); printf("; the first decoys instead: %.2f ms", run(d4, ticks)); }
    printf("\n");
    return 0;
}
