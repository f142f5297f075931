// batch.h -- host-side plumbing shared by the lockstep batch engines of the encoder and the decoder:
// the worker pool the per-stream host phases run on, and the pinned/device arena the per-step job
// tables are built in.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "dev.h"

namespace dsv2 {

// Job tables of one step: built in pinned host memory, mirrored at the same offsets in device memory.
// take() hands out matching host / device views, upload() ships what was added since the last upload.
struct TableArena {
    uint8_t *h = nullptr, *d = nullptr;
    size_t cap = 0, used = 0, sent = 0;
    void reserve(size_t bytes)
    {
        used = sent = 0;
        if (bytes <= cap) {
            return;
        }
        if (cap) {
            HIPCHK(hipHostFree(h));
            HIPCHK(hipFree(d));
        }
        HIPCHK(hipHostMalloc((void **) &h, bytes, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **) &d, bytes));
        cap = bytes;
    }
    template <class T> T *take(size_t n, const T **dev)
    {
        used = (used + 15) & ~(size_t) 15;
        if (used + n * sizeof(T) > cap) {
            fatal("batch job table arena exhausted", __FILE__, __LINE__);
        }
        T *hp = (T *) (h + used);
        *dev = (const T *) (d + used);
        used += n * sizeof(T);
        return hp;
    }
    void upload(hipStream_t s)
    {
        if (used > sent) {
            HIPCHK(hipMemcpyAsync(d + sent, h + sent, used - sent, hipMemcpyHostToDevice, s));
            sent = used;
        }
    }
};

// Host phases run one task per stream on a process-wide pool of worker threads (created on first
// use, sized to the machine); the calling thread works too, and several lockstep groups may share
// the pool concurrently.
class WorkerPool {
  public:
    struct Batch {
        std::function<void(int)> fn;
        int n = 0;
        std::atomic<int> next{0}, left{0};
    };
    static WorkerPool &get()
    {
        static WorkerPool *pool = new WorkerPool(); // never destroyed: workers may outlive static teardown
        return *pool;
    }
    void run(int n, const std::function<void(int)> &fn)
    {
        auto b = std::make_shared<Batch>();
        b->fn = fn;
        b->n = n;
        b->left.store(n);
        {
            std::lock_guard<std::mutex> lk(mu_);
            active_.push_back(b);
        }
        cv_.notify_all();
        work(*b);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return b->left.load() == 0; });
    }

  private:
    WorkerPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned nthreads = hw ? (hw > 48 ? 48 : hw) : 8; // more only adds wake-ups: the per-stream tasks are short (measured 32 ... 128)
        if (const char *e = getenv("DSV2_HOST_THREADS")) {
            int v = atoi(e);
            nthreads = (unsigned) (v < 1 ? 1 : (v > 256 ? 256 : v)); // 1 = the calling thread does all the work
        }
        for (unsigned i = 0; i + 1 < nthreads; i++) {
            std::thread([this] { loop(); }).detach();
        }
    }
    void work(Batch &b)
    {
        for (;;) {
            int k = b.next.fetch_add(1);
            if (k >= b.n) {
                return;
            }
            b.fn(k);
            if (b.left.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_.notify_all();
            }
        }
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(mu_);
                for (;;) {
                    while (!active_.empty() && active_.front()->next.load() >= active_.front()->n) {
                        active_.pop_front(); // fully handed out
                    }
                    if (!active_.empty()) {
                        b = active_.front();
                        break;
                    }
                    cv_.wait(lk);
                }
            }
            work(*b);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<std::shared_ptr<Batch>> active_;
};

// Submit queue of the single-call entry points (dsv_enc, dsv_dec).  The reference's interface is one synchronous call per
// frame (dsv_encoder.h:190-199, dsv_decoder.h:54-61) and its only parallelism is one encoder per thread or process
// (parallel_encode_yuv.sh:31-52); a caller that keeps that shape -- T threads, each looping dsv_enc on its own encoder -- would
// give the GPU T lockstep steps of ONE picture each.  Here concurrent callers whose jobs can share a step (same key: picture
// geometry and what else a batch must agree on) are run as ONE lockstep step by whichever of them arrived first:
//   * the first caller to arrive becomes the LEADER; it waits -- bounded: `window` -- until the callers it can expect have
//     arrived, takes them, runs the batch on its own thread, and hands every caller its results; each call still returns when
//     ITS frame is done.  Whom it expects: a call blocks its THREAD, so what can still arrive is one call per other thread
//     that has driven an instance of this key within the last 100 ms, less the ones inside a running step.  (One thread that
//     drives several instances in turn -- simulcast renditions, a loop over encoders -- counts once: its other instances
//     cannot call while this one waits, and the plain path stays the plain path.)  An instance inside a running step does
//     not age out; one that stopped calling without saying so (paused input, a thread that bailed out) does, after 100 ms,
//     whatever else is running;
//   * a few callers run one step at a time (a caller that arrives while it runs joins the next one); with many callers
//     (>= 2 * kMinSplit) a leader takes only its share (1 / groups) and up to `groups` steps run side by side, one's host
//     phases and latency-bound kernels under another's;
//   * a single caller finds nobody to wait for and runs at once: the plain dsv_enc path.
// DSV2_COALESCE=0 turns it off, DSV2_COALESCE_US sets the longest wait (default 10 % of the last step, 100 us .. 2 ms); a crowd
// is split into two concurrent steps.
template <class JobT> class Coalescer {
  public:
    using RunFn = void (*)(JobT *jobs, int n);
    struct Stats {
        unsigned long long calls = 0, steps = 0, largest = 0, waited_us = 0;
    };
    // runs `job` (by value in, results copied back) as part of some lockstep step; `who` identifies the caller's instance
    void submit(JobT &job, unsigned long long key, const void *who, RunFn run)
    {
        using clock = std::chrono::steady_clock;
        Pending me;
        me.job = &job;
        me.key = key;
        me.who = who;
        std::unique_lock<std::mutex> lk(mu_);
        const auto now = clock::now();
        seen(who, key, now, std::this_thread::get_id());
        pending_.push_back(&me);
        st_.calls++;
        cv_.notify_all(); // (a collecting leader counts arrivals)
        for (;;) {
            if (me.done) {
                return;
            }
            // nobody is collecting and this key may start another step: this caller leads.  (A few callers run ONE step at
            // a time: whoever arrives while it runs waits for it, and the step after it takes everybody -- callers that start
            // out of phase fall into step after one round instead of each running alone for ever.)
            if (!me.taken && !collecting_ && steps_running(key) < max_steps(key, clock::now())) {
                break;
            }
            cv_.wait(lk);
        }
        collecting_ = true;
        const int groups = std::max(1, groups_);
        auto mine = [&] {
            int c = 0;
            for (Pending *p : pending_) {
                c += p->key == key && !p->taken;
            }
            return c;
        };
        auto want = [&] {
            int l = live(key, clock::now());
            int share = l >= 2 * kMinSplit ? (l + groups - 1) / groups : l;
            int free_callers = l - inflight(key);
            return std::max(1, std::min(share, free_callers));
        };
        const auto t0 = clock::now();
        const auto deadline = t0 + std::chrono::microseconds(window_us());
        while (mine() < want()) {
            if (cv_.wait_until(lk, deadline) == std::cv_status::timeout) {
                break;
            }
        }
        st_.waited_us += (unsigned long long) std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count();
        // take this key's callers in arrival order, up to the share
        std::vector<Pending *> batch;
        {
            const int cap = want();
            std::vector<Pending *> rest;
            for (Pending *p : pending_) {
                if (p->key == key && !p->taken && ((int) batch.size() < cap || p == &me)) {
                    p->taken = true;
                    batch.push_back(p);
                } else {
                    rest.push_back(p);
                }
            }
            pending_.swap(rest);
        }
        const unsigned long long step_id = ++step_seq_;
        {
            Running r;
            r.key = key;
            r.n = (int) batch.size();
            r.id = step_id;
            for (Pending *p : batch) {
                r.whos.push_back(p->who);
            }
            running_.push_back(std::move(r));
        }
        collecting_ = false;
        cv_.notify_all(); // whoever is left elects the next leader
        lk.unlock();
        std::vector<JobT> jobs;
        jobs.reserve(batch.size());
        for (Pending *p : batch) {
            jobs.push_back(*p->job);
        }
        const auto r0 = clock::now();
        run(jobs.data(), (int) jobs.size());
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - r0).count();
        lk.lock();
        last_step_us_ = us;
        st_.steps++;
        st_.largest = std::max<unsigned long long>(st_.largest, batch.size());
        {
            const auto done_at = clock::now();
            for (size_t i = 0; i < running_.size(); i++) {
                if (running_[i].id == step_id) {
                    for (const void *w : running_[i].whos) { // (the 100 ms start when the step hands its callers back)
                        for (Recent &r : recent_) {
                            if (r.who == w) {
                                r.at = done_at;
                            }
                        }
                    }
                    running_.erase(running_.begin() + (ptrdiff_t) i);
                    break;
                }
            }
        }
        for (size_t i = 0; i < batch.size(); i++) {
            *batch[i]->job = jobs[i];
            batch[i]->done = true;
        }
        cv_.notify_all();
    }
    // the instance will not call again (freed / end of stream): leaders stop expecting it
    void forget(const void *who)
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t i = 0; i < recent_.size(); i++) {
            if (recent_[i].who == who) {
                recent_.erase(recent_.begin() + (ptrdiff_t) i);
                break;
            }
        }
        cv_.notify_all();
    }
    Stats stats()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return st_;
    }
    void reset_stats()
    {
        std::lock_guard<std::mutex> lk(mu_);
        st_ = Stats();
    }
    static bool enabled()
    {
        static const bool on = !(getenv("DSV2_COALESCE") && atoi(getenv("DSV2_COALESCE")) == 0);
        return on;
    }

  private:
    static constexpr int kMinSplit = 4; // a crowd is split only when every part keeps at least this many callers
    struct Pending {
        JobT *job = nullptr;
        unsigned long long key = 0;
        const void *who = nullptr;
        bool taken = false, done = false;
    };
    struct Recent {
        const void *who;
        unsigned long long key;
        std::chrono::steady_clock::time_point at;
        std::thread::id tid; // the thread that last submitted for this instance
    };
    struct Running {
        unsigned long long key = 0, id = 0;
        int n = 0;
        std::vector<const void *> whos; // the instances inside this step
    };
    void seen(const void *who, unsigned long long key, std::chrono::steady_clock::time_point now, std::thread::id tid)
    {
        for (Recent &r : recent_) {
            if (r.who == who) {
                r.key = key;
                r.at = now;
                r.tid = tid;
                return;
            }
        }
        recent_.push_back(Recent{who, key, now, tid});
    }
    // callers that can be expected under `key`: the distinct THREADS behind its recently seen instances
    int live(unsigned long long key, std::chrono::steady_clock::time_point now)
    {
        std::vector<std::thread::id> tids;
        for (size_t i = 0; i < recent_.size();) {
            if (now - recent_[i].at > std::chrono::milliseconds(100) && !busy(recent_[i].who)) {
                recent_.erase(recent_.begin() + (ptrdiff_t) i); // has not called for 100 ms: no longer expected
                continue;
            }
            if (recent_[i].key == key && std::find(tids.begin(), tids.end(), recent_[i].tid) == tids.end()) {
                tids.push_back(recent_[i].tid);
            }
            i++;
        }
        return (int) tids.size();
    }
    bool busy(const void *who) // (an instance inside a running step does not age out: its step may take longer than the 100 ms)
    {
        for (const Running &r : running_) {
            if (std::find(r.whos.begin(), r.whos.end(), who) != r.whos.end()) {
                return true;
            }
        }
        for (const Pending *p : pending_) { // ... nor one that is waiting right now
            if (p->who == who) {
                return true;
            }
        }
        return false;
    }
    int steps_running(unsigned long long key)
    {
        int c = 0;
        for (const Running &r : running_) {
            c += r.key == key;
        }
        return c;
    }
    int max_steps(unsigned long long key, std::chrono::steady_clock::time_point now)
    {
        return live(key, now) >= 2 * kMinSplit ? std::max(1, groups_) : 1;
    }
    int inflight(unsigned long long key)
    {
        int c = 0;
        for (const Running &r : running_) {
            c += r.key == key ? r.n : 0;
        }
        return c;
    }
    long long window_us()
    {
        static const long long fixed = getenv("DSV2_COALESCE_US") ? atoll(getenv("DSV2_COALESCE_US")) : -1;
        if (fixed >= 0) {
            return fixed;
        }
        return std::min<long long>(2000, std::max<long long>(100, last_step_us_ / 10));
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Pending *> pending_;
    std::vector<Recent> recent_;
    std::vector<Running> running_;
    bool collecting_ = false;
    unsigned long long step_seq_ = 0;
    long long last_step_us_ = 0;
    const int groups_ = 2;
    Stats st_;
};

// A first-come-first-served token with a bounded wait (the encoder's search token, encoder.cpp).  A waiter that gives up RETIRES
// its ticket: `serving` never stops at a ticket nobody is waiting with, so one time-out fails one call and the next waiter still
// gets the token when the holder lets go (advisor, round 5: the abandoned ticket parked every later waiter for good).
// CPU harness: tests/coalescer_harness.cpp ("token_timeout").
struct FifoToken {
    struct TimedOut {};
    std::mutex mu;
    std::condition_variable cv;
    int free_slots = 1;
    unsigned long long next_ticket = 0, serving = 0;
    std::vector<unsigned long long> retired; // tickets beyond `serving` whose waiters gave up
    void skip_retired() // (mu held)
    {
        for (;;) {
            auto it = std::find(retired.begin(), retired.end(), serving);
            if (it == retired.end()) {
                return;
            }
            retired.erase(it);
            serving++;
        }
    }
    template <class Rep, class Period> void acquire(std::chrono::duration<Rep, Period> patience)
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long mine = next_ticket++;
        if (!cv.wait_for(lk, patience, [&] { return free_slots > 0 && serving == mine; })) {
            if (serving == mine) {
                serving++;
                skip_retired();
            } else {
                retired.push_back(mine);
            }
            lk.unlock();
            cv.notify_all(); // whoever is next in line may have been waiting for this ticket only
            throw TimedOut{};
        }
        serving++;
        skip_retired();
        free_slots--;
        if (free_slots > 0) {
            cv.notify_all(); // (more than one slot: the next in line may go as well)
        }
    }
    void release()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            free_slots++;
        }
        cv.notify_all();
    }
};

template <class F> inline void parallel_for(int n, F fn)
{
    if (n == 1) {
        fn(0);
        return;
    }
    WorkerPool::get().run(n, std::function<void(int)>(fn));
}

} // namespace dsv2
