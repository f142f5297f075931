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

// ONE thread drives N instances of one key in turn (simulcast renditions, a loop over encoders): nobody else can arrive while a
// call blocks that thread, so no call may spend the window waiting for the thread's other instances
static Result one_thread_many(int N, int K)
{
    Coalescer<Job> q;
    g_mixed = 0;
    std::vector<int> who(N);
    int wrong = 0;
    long long batch_sum = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; k++) {
        for (int n = 0; n < N; n++) {
            Job j;
            j.id = n * 1000 + k;
            j.key = 1;
            q.submit(j, 1ull, &who[n], step);
            wrong += j.out != 7 * j.id + 1;
            batch_sum += j.batch;
        }
    }
    Result r;
    const auto st = q.stats();
    r.calls = st.calls;
    r.steps = st.steps;
    r.largest = st.largest;
    r.waited_us = st.waited_us;
    r.wrong = wrong;
    r.mixed = g_mixed.load();
    r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    r.mean_batch = (double) batch_sum / (double) (N * K);
    return r;
}

// two callers of one key; one of them STOPS calling without saying so (no forget: paused input, a thread that bailed out) while
// a long-running step of ANOTHER key keeps the queue busy all the time: after 100 ms the silent one must not be expected any more
static Result silent_leaver()
{
    Coalescer<Job> q;
    g_mixed = 0;
    std::vector<int> who(3);
    std::atomic<int> wrong{0};
    std::atomic<bool> stop{false};
    std::thread other([&] { // key 2: back-to-back steps, so that some step is running at every moment
        int k = 0;
        while (!stop.load()) {
            Job j;
            j.id = 2000 + k++;
            j.key = 2;
            q.submit(j, 2ull, &who[2], step);
        }
        q.forget(&who[2]);
    });
    std::thread quitter([&] {
        for (int k = 0; k < 3; k++) {
            Job j;
            j.id = 1000 + k;
            j.key = 1;
            q.submit(j, 1ull, &who[1], step);
        }
        // (no forget)
    });
    quitter.join();
    std::this_thread::sleep_for(std::chrono::milliseconds(150)); // the silent instance ages out -- although steps never stop running
    q.reset_stats();
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 40; k++) {
        Job j;
        j.id = k;
        j.key = 1;
        q.submit(j, 1ull, &who[0], step);
        wrong += j.out != 7 * j.id + 1;
    }
    Result r;
    const auto st = q.stats();
    stop = true;
    other.join();
    r.calls = st.calls;
    r.steps = st.steps;
    r.largest = st.largest;
    r.waited_us = st.waited_us;
    r.wrong = wrong.load();
    r.mixed = g_mixed.load();
    r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    r.mean_batch = 1;
    return r;
}

// the search token (batch.h: FifoToken): a holder that keeps it, a waiter that gives up after 50 ms, three more in line behind it.
// When the holder lets go every one of the three gets the token, in the order they asked (advisor finding, round 5: the ticket
// the first waiter abandoned was never served and every later waiter timed out too).
static Result token_timeout()
{
    FifoToken tok;
    std::atomic<int> timed_out{0}, got{0}, order_wrong{0};
    std::atomic<int> last{-1};
    const auto t0 = std::chrono::steady_clock::now();
    tok.acquire(std::chrono::seconds(1)); // the holder
    std::thread quitter([&] {
        try {
            tok.acquire(std::chrono::milliseconds(50));
            got++;
            tok.release();
        } catch (const FifoToken::TimedOut &) {
            timed_out++;
        }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
    std::vector<std::thread> later;
    for (int t = 0; t < 3; t++) {
        later.emplace_back([&, t] {
            try {
                tok.acquire(std::chrono::seconds(5));
                order_wrong += last.exchange(t) != t - 1;
                got++;
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                tok.release();
            } catch (const FifoToken::TimedOut &) {
                timed_out++;
            }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(10)); // (tickets in thread order)
    }
    quitter.join();                                              // gave up at ~50 ms, the holder still holding
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    tok.release();
    for (auto &x : later) {
        x.join();
    }
    // and a waiter that is FIRST in line when it gives up (its ticket == serving) must not stop the line either
    tok.acquire(std::chrono::seconds(1));
    std::thread front([&] {
        try {
            tok.acquire(std::chrono::milliseconds(20));
            got++;
            tok.release();
        } catch (const FifoToken::TimedOut &) {
            timed_out++;
        }
    });
    front.join();
    tok.release();
    try {
        tok.acquire(std::chrono::milliseconds(200));
        got++;
        tok.release();
    } catch (const FifoToken::TimedOut &) {
        timed_out++;
    }
    Result r{};
    r.calls = (unsigned long long) got.load();       // 3 in line + the last one = 4
    r.steps = (unsigned long long) timed_out.load(); // the two quitters = 2
    r.wrong = order_wrong.load();
    r.largest = tok.retired.size();                  // nothing left behind
    r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
    print("one_leaves", scenario(2, 60, {1, 1}, 1, 5), false);
    print("one_thread_four_instances", one_thread_many(4, 20), false);
    print("silent_leaver", silent_leaver(), false);
    print("token_timeout", token_timeout(), true);
    printf("}\n");
    return 0;
}
