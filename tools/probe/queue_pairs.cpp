// Which streams share a hardware queue?  N streams created one after another; every pair (i, j) gets two 1 ms spin kernels at
// once: ~1 ms = different queues, ~2 ms = the same queue.  Prints the classes of streams that serialise with each other, then
// destroys some streams, creates new ones and prints the classes again.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/queue_pairs.cpp -o tools/probe/queue_pairs && GPU_MAX_HW_QUEUES=8 tools/probe/queue_pairs 16
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void spin(long long ticks)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}
static double pair_ms(hipStream_t a, hipStream_t b)
{
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100000);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 100000);
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static void classes(std::vector<hipStream_t> &s, const char *what)
{
    int n = (int) s.size();
    std::vector<int> cls((size_t) n, -1);
    int nc = 0;
    for (int i = 0; i < n; i++) {
        if (cls[(size_t) i] >= 0) continue;
        cls[(size_t) i] = nc;
        for (int j = i + 1; j < n; j++) {
            if (cls[(size_t) j] < 0 && pair_ms(s[(size_t) i], s[(size_t) j]) > 1.6) cls[(size_t) j] = nc;
        }
        nc++;
    }
    printf("%s: %d streams in %d classes:", what, n, nc);
    for (int i = 0; i < n; i++) printf(" %d", cls[(size_t) i]);
    printf("\n");
}
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<hipStream_t> s((size_t) n);
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (auto x : s) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, x, 100); }
    CK(hipDeviceSynchronize());
    classes(s, "created in order");
    classes(s, "again");
    // destroy the first four, create four new ones
    for (int i = 0; i < 4; i++) CK(hipStreamDestroy(s[(size_t) i]));
    for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&s[(size_t) i], hipStreamNonBlocking));
    classes(s, "first four re-created");
    // leave only six streams alive
    for (int i = 6; i < n; i++) CK(hipStreamDestroy(s[(size_t) i]));
    s.resize(6);
    classes(s, "six left");
    for (int k = 0; k < 3; k++) {
        std::vector<hipStream_t> t(4);
        for (auto &x : t) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        classes(t, "four fresh ones beside the six");
        for (auto x : t) CK(hipStreamDestroy(x));
    }
    return 0;
}
