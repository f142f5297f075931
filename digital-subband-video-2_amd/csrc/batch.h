// batch.h -- host-side plumbing shared by the lockstep batch engines of the encoder and the decoder:
// the worker pool the per-stream host phases run on, and the pinned/device arena the per-step job
// tables are built in.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

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

template <class F> inline void parallel_for(int n, F fn)
{
    if (n == 1) {
        fn(0);
        return;
    }
    WorkerPool::get().run(n, std::function<void(int)>(fn));
}

} // namespace dsv2
