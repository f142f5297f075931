// CPU harness of the submit queue behind dsv_enc / dsv_dec (csrc/batch.h: Coalescer): threads that each loop a synchronous call
// on an instance of their own, a step function that only sleeps.  No GPU call is made; built by `make -C oracle coalescer-test`
// (host-only compile of the library's own header), run by tests/test_coalescer_cpu.py, which reads the JSON line it prints.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../digital-subband-video-2_amd/csrc/batch.h"

using namespace dsv2;

struct Job {
    int id = 0, key = 0;
    int out = 0, batch = 0; // filled by the step: 7 * id + 1, and how many jobs shared the step
};

static std::atomic<int> g_mixed{0}, g_steps{0}, g_largest{0};
static int g_step_us = 2000;

static void step(Job *jobs, int n)
{
    for (int i = 0; i < n; i++) {
        g_mixed += jobs[i].key != jobs[0].key; // a step never mixes keys
        jobs[i].out = 7 * jobs[i].id + 1;
        jobs[i].batch = n;
    }
    g_steps++;
    int prev = g_largest.load();
    while (n > prev && !g_largest.compare_exchange_weak(prev, n)) {
    }
    std::this_thread::sleep_for(std::chrono::microseconds(g_step_us));
}

struct Result {
    unsigned long long calls, steps, largest, waited_us;
    int wrong, mixed;
    double seconds;
    double mean_batch;
};

// T threads x K calls; thread t uses key keys[t]; threads whose index is in `leaves_after` stop after that many calls and say so
static Result scenario(int T, int K, const std::vector<int> &keys, int leave_thread, int leave_after)
{
    Coalescer<Job> q;
    g_mixed = 0;
    g_steps = 0;
    g_largest = 0;
    std::atomic<int> wrong{0};
    std::atomic<long long> batch_sum{0}, ncalls{0};
    std::vector<int> who(T);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) {
        th.emplace_back([&, t] {
            const int calls = t == leave_thread ? leave_after : K;
            for (int k = 0; k < calls; k++) {
                Job j;
                j.id = t * 1000 + k;
                j.key = keys[t];
                q.submit(j, (unsigned long long) keys[t], &who[t], step);
                wrong += j.out != 7 * j.id + 1;
                batch_sum += j.batch;
                ncalls++;
            }
            q.forget(&who[t]);
        });
    }
    for (auto &x : th) {
        x.join();
    }
    Result r;
    const auto st = q.stats();
    r.calls = st.calls;
    r.steps = st.steps;
    r.largest = st.largest;
    r.waited_us = st.waited_us;
    r.wrong = wrong.load();
    r.mixed = g_mixed.load();
    r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    r.mean_batch = ncalls ? (double) batch_sum / (double) ncalls : 0;
    return r;
}

static void print(const char *name, const Result &r, bool last)
{
    printf("\"%s\": {\"calls\": %llu, \"steps\": %llu, \"largest\": %llu, \"waited_us\": %llu, \"wrong\": %d, \"mixed\": %d, \"seconds\": %.4f, "
           "\"mean_batch\": %.3f}%s",
           name, r.calls, r.steps, r.largest, r.waited_us, r.wrong, r.mixed, r.seconds, r.mean_batch, last ? "" : ", ");
}

int main(int argc, char **argv)
{
    g_step_us = argc > 1 ? atoi(argv[1]) : 2000;
    printf("{");
    print("one_caller", scenario(1, 40, {1}, -1, 0), false);
    print("four_callers", scenario(4, 40, {1, 1, 1, 1}, -1, 0), false);
    print("sixteen_callers", scenario(16, 30, std::vector<int>(16, 1), -1, 0), false);
    print("two_keys", scenario(6, 30, {1, 1, 1, 2, 2, 2}, -1, 0), false);
    print("one_leaves", scenario(2, 60, {1, 1}, 1, 5), true);
    printf("}\n");
    return 0;
}
