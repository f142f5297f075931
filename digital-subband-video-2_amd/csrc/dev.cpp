// dev.cpp -- device memory, frame transfer and error plumbing for libdsv2hip.
#include "dev.h"

#include <string.h>
#include <sys/prctl.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <map>
#include <mutex>

namespace dsv2 {

thread_local long long t_launch_count = 0;

int trace_mode()
{
    static const int m = getenv("DSV2_TRACE") ? atoi(getenv("DSV2_TRACE")) : 0;
    return m;
}

// DSV2_TRACE=1: milliseconds since the library was loaded at a few points of a process's first step (what a short-lived
// caller -- one CLI process per closed-GOP segment, parallel_encode_yuv.sh:31-52 -- pays before its first packet)
static const std::chrono::steady_clock::time_point g_loaded = std::chrono::steady_clock::now();
void startup_mark(const char *what)
{
    static const bool on = (trace_mode() & 1) != 0;
    if (on) {
        fprintf(stderr, "[dsv2hip startup] %8.1f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_loaded).count(), what);
    }
}


[[noreturn]] void fatal(const char *what, const char *file, int line)
{
    fprintf(stderr, "[dsv2hip] FATAL: %s (%s:%d)\n", what, file, line);
    fflush(stderr);
    abort();
}

static std::once_flag g_dev_once;
static int g_dev_status = -1;

static void probe_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_dev_status = -1;
        return;
    }
    g_dev_status = 0;
}

int device_status()
{
    std::call_once(g_dev_once, probe_device);
    return g_dev_status;
}

// There is deliberately no CPU path behind this: a missing GPU is a hard error.
void ensure_device()
{
    if (device_status() != 0) {
        fatal("no usable HIP device: libdsv2hip has no CPU fallback", __FILE__, __LINE__);
    }
}

static int g_device_ordinal = 0;
void set_default_device(int ordinal) { g_device_ordinal = ordinal; }
// HIP's current device is per host thread: every entry point that may run on a new thread binds it
void bind_device()
{
    ensure_device();
    HIPCHK(hipSetDevice(g_device_ordinal));
}

bool device_arch_is(const char *prefix)
{
    ensure_device();
    static std::once_flag once;
    static char arch[256];
    std::call_once(once, [] {
        hipDeviceProp_t pr;
        HIPCHK(hipGetDeviceProperties(&pr, g_device_ordinal));
        snprintf(arch, sizeof(arch), "%s", pr.gcnArchName);
    });
    return strncmp(arch, prefix, strlen(prefix)) == 0;
}

void coef_dims(int format, int w, int h, int cw[3], int ch[3]) // frame.c:30-60
{
    int hs = DSV_FORMAT_H_SHIFT(format), vs = DSV_FORMAT_V_SHIFT(format);
    int c_w = (w + (1 << hs) - 1) >> hs, c_h = (h + (1 << vs) - 1) >> vs;
    c_w = (c_w + 1) & ~1;
    c_h = (c_h + 1) & ~1;
    cw[0] = w;
    ch[0] = h;
    cw[1] = cw[2] = c_w;
    ch[1] = ch[2] = c_h;
}

// ---- one device block per codec instance (dev.h: DevArena) ----
static thread_local DevArena *t_arena = nullptr;
static std::mutex g_arena_mu;
static std::map<uintptr_t, size_t> g_arenas; // live blocks: base -> bytes

void DevArena::create(size_t bytes)
{
    HIPCHK(hipMalloc((void **) &base, bytes));
    HIPCHK(hipMemset(base, 0, bytes));
    HIPCHK(hipStreamSynchronize(nullptr));
    cap = bytes;
    used = 0;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    g_arenas[(uintptr_t) base] = bytes;
}

void DevArena::destroy()
{
    if (!base) {
        return;
    }
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        g_arenas.erase((uintptr_t) base);
    }
    HIPCHK(hipFree(base));
    base = nullptr;
    cap = used = 0;
}

DevArenaScope::DevArenaScope(DevArena *a) : prev(t_arena) { t_arena = a; }
DevArenaScope::~DevArenaScope() { t_arena = prev; }

// allocations an instance arena could not hold (its size is an estimate kept in step with the allocations by hand: CodecDev::init);
// counted so that drift shows up in the tests (dsv2hip_arena_fallbacks) instead of as a silent extra hipMalloc per instance
std::atomic<long> g_arena_short{0};

long arena_fallbacks() { return g_arena_short.load(); }

hipError_t dev_alloc(void **p, size_t bytes)
{
    DevArena *a = t_arena;
    if (a && a->base) {
        const size_t off = (a->used + 255) & ~(size_t) 255;
        if (off + bytes <= a->cap) {
            *p = a->base + off;
            a->used = off + bytes;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && a && a->base) { // (the estimate of the block was short: this piece lives on its own, zeroed like the block)
        g_arena_short.fetch_add(1);
        HIPCHK(hipMemset(*p, 0, bytes));
        HIPCHK(hipStreamSynchronize(nullptr));
    }
    return e;
}

static bool in_live_arena(const void *p)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.upper_bound((uintptr_t) p);
    if (it == g_arenas.begin()) {
        return false;
    }
    --it;
    return (uintptr_t) p < it->first + it->second;
}

void dev_release(void *p)
{
    if (p && !in_live_arena(p)) {
        HIPCHK(hipFree(p));
    }
}

// hipMemset is asynchronous to the host and runs on the null stream, which the codec's non-blocking
// streams do not wait for: finish it before anything may be enqueued on those streams
void dev_zero(void *p, size_t bytes)
{
    if (t_arena && t_arena->base && in_live_arena(p)) {
        return; // (the block was zeroed when it was made)
    }
    HIPCHK(hipMemset(p, 0, bytes));
    HIPCHK(hipStreamSynchronize(nullptr));
}

// Host wait for a stream to drain WITHOUT occupying a core: the runtime's waits (hipStreamSynchronize, and
// hipEventSynchronize even on an event created for blocking synchronisation) poll, so a waiting thread shows
// ~100 % CPU.  The host threads of the lockstep groups wait most of the time, and host CPU time is the scarce
// resource once the kernels are fast (a container's CPU quota is shared with the entropy back end).  Here the
// thread sleeps between completion queries; the added latency is bounded by the sleep (tens of microseconds
// against waits of several milliseconds).  DSV2_SPIN_WAIT=1 restores the runtime's polling wait.
// Polling interval of the two waits below.  A single stream's step is a chain of latency-bound kernels with a host phase
// between them, and a wake-up that comes late (up to 120 us on the sparse schedule a multi-millisecond wait ends up in) two or
// three times per frame is 1 % of its frame: such a caller polls at 20 us throughout (DSV2_WAIT_FINE_MAX, default 1 stream).
// Several lockstep groups polling that finely at once was measured too: 4 x 2 streams lose 9 % (the queries contend with the
// other groups' launches inside the runtime), so batches keep the sparse schedule -- their host is shared by every rank.
static thread_local bool t_wait_fine = false;
void set_wait_fine(bool fine) { t_wait_fine = fine; }
static inline long wait_interval_ns(unsigned tries)
{
    if (t_wait_fine) {
        return tries < 8 ? 10000 : 20000;
    }
    return tries < 8 ? 15000 : (tries < 64 ? 40000 : 120000);
}

void stream_wait(hipStream_t s)
{
    static thread_local hipEvent_t ev = nullptr;
    if (!ev) {
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        prctl(PR_SET_TIMERSLACK, 2000ul, 0, 0, 0); // the default slack of 50 us would triple every short sleep
    }
    HIPCHK(hipEventRecord(ev, s));
    for (unsigned tries = 0;; tries++) {
        hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) {
            return;
        }
        if (e != hipErrorNotReady) {
            HIPCHK(e);
        }
        // short waits are answered quickly, long ones (tens of milliseconds of kernels) are polled sparsely: every
        // wake-up costs several microseconds of host CPU, and the host is shared by every rank of the node
        timespec ts = {0, wait_interval_ns(tries)};
        nanosleep(&ts, nullptr);
    }
}

// the same for an event that has been recorded already
void event_wait(hipEvent_t ev)
{
    for (unsigned tries = 0;; tries++) {
        hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) {
            return;
        }
        if (e != hipErrorNotReady) {
            HIPCHK(e);
        }
        timespec ts = {0, wait_interval_ns(tries)};
        nanosleep(&ts, nullptr);
    }
}

size_t dframe_bytes(int format, int w, int h)
{
    int hs = DSV_FORMAT_H_SHIFT(format), vs = DSV_FORMAT_V_SHIFT(format);
    int cw = (w + (1 << hs) - 1) >> hs, ch = (h + (1 << vs) - 1) >> vs;
    int pw[3] = {w, cw, cw}, ph[3] = {h, ch, ch};
    size_t off = 0;
    for (int c = 0; c < 3; c++) {
        off += (size_t) ((pw[c] + 2 * kBorder + 15) & ~15) * (size_t) (ph[c] + 2 * kBorder);
    }
    return off + 4096;
}

void dframe_alloc(DFrame *f, int format, int w, int h) // layout of frame.c:63-113, always bordered
{
    ensure_device();
    int hs = DSV_FORMAT_H_SHIFT(format), vs = DSV_FORMAT_V_SHIFT(format);
    int cw = (w + (1 << hs) - 1) >> hs, ch = (h + (1 << vs) - 1) >> vs;
    int pw[3] = {w, cw, cw}, ph[3] = {h, ch, ch};
    size_t off = 0;
    f->format = format;
    f->w = w;
    f->h = h;
    for (int c = 0; c < 3; c++) {
        int stride = (pw[c] + 2 * kBorder + 15) & ~15;
        f->p[c].stride = stride;
        f->p[c].w = pw[c];
        f->p[c].h = ph[c];
        f->plane_off[c] = off;
        f->plane_len[c] = (size_t) stride * (ph[c] + 2 * kBorder);
        off += f->plane_len[c];
    }
    f->bytes = off;
    // slack after the last plane: block reads may run a few bytes past the final border row
    HIPCHK(dev_alloc((void **) &f->alloc, off + 4096));
    dev_zero(f->alloc, off + 4096);
    for (int c = 0; c < 3; c++) {
        f->p[c].data = f->alloc + f->plane_off[c] + (size_t) f->p[c].stride * kBorder + kBorder;
    }
}

void dframe_free(DFrame *f)
{
    if (f->alloc) {
        dev_release(f->alloc);
        f->alloc = nullptr;
    }
}

void dframe_upload(DFrame *d, const DSV_FRAME *h, hipStream_t s)
{
    for (int c = 0; c < 3; c++) {
        const DSV_PLANE *hp = &h->planes[c];
        int w = hp->w < d->p[c].w ? hp->w : d->p[c].w;
        int rows = hp->h < d->p[c].h ? hp->h : d->p[c].h;
        HIPCHK(hipMemcpy2DAsync(d->p[c].data, d->p[c].stride, hp->data, hp->stride, w, rows, hipMemcpyHostToDevice, s));
    }
}

void dframe_download(const DFrame *d, DSV_FRAME *h, hipStream_t s)
{
    for (int c = 0; c < 3; c++) {
        DSV_PLANE *hp = &h->planes[c];
        int w = hp->w < d->p[c].w ? hp->w : d->p[c].w;
        int rows = hp->h < d->p[c].h ? hp->h : d->p[c].h;
        HIPCHK(hipMemcpy2DAsync(hp->data, hp->stride, d->p[c].data, d->p[c].stride, w, rows, hipMemcpyDeviceToHost, s));
    }
}

static bool same_layout(const DFrame *d, const DSV_FRAME *h)
{
    if (!h->border) {
        return false;
    }
    for (int c = 0; c < 3; c++) {
        if (h->planes[c].stride != d->p[c].stride || h->planes[c].w != d->p[c].w || h->planes[c].h != d->p[c].h) {
            return false;
        }
    }
    return true;
}

void dframe_upload_full(DFrame *d, const DSV_FRAME *h, hipStream_t s)
{
    if (!same_layout(d, h)) {
        fatal("dframe_upload_full: host frame is not a bordered frame of the same geometry", __FILE__, __LINE__);
    }
    for (int c = 0; c < 3; c++) {
        const uint8_t *hbase = h->planes[c].data - (size_t) h->planes[c].stride * kBorder - kBorder;
        HIPCHK(hipMemcpyAsync(d->alloc + d->plane_off[c], hbase, d->plane_len[c], hipMemcpyHostToDevice, s));
    }
}

void dframe_download_full(const DFrame *d, DSV_FRAME *h, hipStream_t s)
{
    if (!same_layout(d, h)) {
        fatal("dframe_download_full: host frame is not a bordered frame of the same geometry", __FILE__, __LINE__);
    }
    for (int c = 0; c < 3; c++) {
        // pixels + border only: the stride padding to the right of the border is left untouched
        uint8_t *hbase = h->planes[c].data - (size_t) h->planes[c].stride * kBorder - kBorder;
        HIPCHK(hipMemcpy2DAsync(hbase, h->planes[c].stride, d->alloc + d->plane_off[c], d->p[c].stride,
                                d->p[c].w + 2 * kBorder, d->p[c].h + 2 * kBorder, hipMemcpyDeviceToHost, s));
    }
}

void SbtScratch::ensure(size_t n, size_t n_ll)
{
    if (n_ll == 0 || n_ll > n) {
        n_ll = n;
    }
    if (n <= elems && n_ll <= elems_ll) {
        return;
    }
    release();
    for (int i = 0; i < 3; i++) {
        HIPCHK(hipMalloc((void **) &t[i], (i == 2 ? n : n_ll) * sizeof(int32_t)));
    }
    elems = n;
    elems_ll = n_ll;
}

void SbtScratch::borrow(int32_t *base, size_t n, size_t n_ll)
{
    release();
    if (n_ll == 0 || n_ll > n) {
        n_ll = n;
    }
    const size_t a = (n + 3) & ~(size_t) 3, b = (n_ll + 3) & ~(size_t) 3;
    t[2] = base;
    t[0] = base + a;
    t[1] = base + a + b;
    elems = n;
    elems_ll = n_ll;
    borrowed = true;
}

void SbtScratch::release()
{
    if (borrowed) {
        t[0] = t[1] = t[2] = nullptr;
        elems = elems_ll = 0;
        borrowed = false;
        return;
    }
    for (int i = 0; i < 3; i++) {
        if (t[i]) {
            HIPCHK(hipFree(t[i]));
            t[i] = nullptr;
        }
    }
    elems = elems_ll = 0;
}

} // namespace dsv2

// ---- residency census (prio.h; `make census`): host registry of the per-translation-unit tallies ---------------------------------
// The product build has no tallies: the two entry points exist (one ABI for every build) and report nothing.
namespace dsv2 {
namespace census {
struct Tally {
    unsigned long long ticks, groups, waves, line;
};
typedef void (*read_fn)(Tally *out);
typedef void (*reset_fn)();
struct Unit {
    const char *file;
    read_fn rd;
    reset_fn rs;
};
static Unit g_units[16];
static int g_nunits = 0;
void register_tu(const char *file, read_fn rd, reset_fn rs)
{
    if (g_nunits < 16) {
        g_units[g_nunits++] = Unit{file, rd, rs};
    }
}
} // namespace census
} // namespace dsv2

extern "C" {
void dsv2hip_census_reset(void)
{
    for (int u = 0; u < dsv2::census::g_nunits; u++) {
        dsv2::census::g_units[u].rs();
    }
}
/* one text line per kernel site that ran: "<file> <line> <ticks of the 100 MHz clock x wavefronts> <workgroups> <wavefronts>";
 * returns the number of bytes written (0: this build carries no census) */
int dsv2hip_census_read(char *out, int cap)
{
    int at = 0;
    for (int u = 0; u < dsv2::census::g_nunits; u++) {
        dsv2::census::Tally t[64];
        dsv2::census::g_units[u].rd(t);
        const char *f = strrchr(dsv2::census::g_units[u].file, '/');
        f = f ? f + 1 : dsv2::census::g_units[u].file;
        for (int s = 0; s < 64; s++) {
            if (t[s].groups && at < cap - 160) {
                at += snprintf(out + at, (size_t) (cap - at), "%s %llu %llu %llu %llu\n", f, t[s].line, t[s].ticks, t[s].groups, t[s].waves);
            }
        }
    }
    return at;
}
}
