// hostutil.cpp -- host-side helpers of the public API (include/dsv2_hip.h section 4):
// allocation, frame/buffer bookkeeping, logging, raw YUV file I/O.  These mirror the
// caller-visible behaviour of reference src/dsv.c:29-322 and src/frame.c:19-183, 437-446
// (zero-initialised allocations, reference-counted frames, 32-px bordered layout).
#include <string.h>

#include <atomic>

#include "dev.h"

// ---- pool of pinned (device-writable) blocks for decoder output pictures ------------------------------
// The batch decoder lets the GPU write a finished picture straight into the DSV_FRAME it returns; such a
// frame's pixel block comes from here and finds its way back when the caller releases the frame
// (dsv_frame_ref_dec -> dsv_free).  Blocks are recycled by exact size; the pool keeps at most kPoolMax idle.
#include <mutex>
#include <unordered_map>
#include <vector>

namespace dsv2 {
namespace {
std::mutex g_pool_mu;
std::unordered_map<void *, size_t> g_pool_live;                 // block -> size, for every block handed out
std::unordered_map<size_t, std::vector<void *>> g_pool_idle;    // size -> idle blocks
size_t g_pool_idle_count = 0;
constexpr size_t kPoolMax = 2048;
// blocks currently handed out: dsv_free() consults the pool (mutex + hash lookup) only while there are any, so that the
// encoder's many ordinary frees (packet buffers on the hot path) never touch the lock
std::atomic<size_t> g_pool_live_count{0};
} // namespace

void *pinned_pool_take(size_t bytes)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_idle.find(bytes);
        if (it != g_pool_idle.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_pool_idle_count--;
            g_pool_live[p] = bytes;
            g_pool_live_count++;
            return p;
        }
    }
    void *p = nullptr;
    HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[p] = bytes;
    g_pool_live_count++;
    return p;
}

bool pinned_pool_release(void *p)
{
    if (g_pool_live_count.load(std::memory_order_acquire) == 0) {
        return false; // (a block of the pool is only ever released by its holder, after the take that counted it)
    }
    size_t bytes;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_live.find(p);
        if (it == g_pool_live.end()) {
            return false;
        }
        bytes = it->second;
        g_pool_live.erase(it);
        g_pool_live_count--;
        if (g_pool_idle_count < kPoolMax) {
            g_pool_idle[bytes].push_back(p);
            g_pool_idle_count++;
            return true;
        }
    }
    HIPCHK(hipHostFree(p));
    return true;
}
} // namespace dsv2

extern "C" {

char *dsv_lvlname[5] = {(char *) "NONE", (char *) "ERROR", (char *) "WARNING", (char *) "INFO", (char *) "DEBUG"};

static std::atomic<int> g_log_level{1};
static std::atomic<unsigned> g_nalloc{0}, g_nfree{0};

void dsv_set_log_level(int level) { g_log_level = level; }
int dsv_get_log_level(void) { return g_log_level; }

void *dsv_alloc(int size)
{
    if (size < 0) {
        return NULL;
    }
    g_nalloc++;
    return calloc(1, (size_t) size + 16);
}

void dsv_free(void *ptr)
{
    if (ptr) {
        g_nfree++;
        if (dsv2::pinned_pool_release(ptr)) {
            return; // a pooled pinned block (decoder output picture): recycled, not freed
        }
        free(ptr);
    }
}

void dsv_memory_report(void)
{
    if (g_log_level >= 4) {
        printf("[DSV][DEBUG] allocations: %u, frees: %u\n", g_nalloc.load(), g_nfree.load());
    }
}

int dsv_lb2(unsigned n) // ceil(log2(n)), dsv.c:450
{
    unsigned i = 1;
    int l = 0;
    while (i < n) {
        i <<= 1;
        l++;
    }
    return l;
}

void dsv_mk_buf(DSV_BUF *buf, int size)
{
    buf->data = (uint8_t *) dsv_alloc(size);
    buf->len = (unsigned) size;
}

void dsv_buf_free(DSV_BUF *buf)
{
    if (buf && buf->data) {
        dsv_free(buf->data);
        buf->data = NULL;
    }
}

void dsv_mk_coefs(DSV_COEFS *c, int format, int width, int height)
{
    int cw[3], ch[3];
    dsv2::coef_dims(format, width, height, cw, ch);
    size_t n0 = (size_t) cw[0] * ch[0], n1 = (size_t) cw[1] * ch[1];
    DSV_SBC *base = (DSV_SBC *) dsv_alloc((int) ((n0 + 2 * n1) * sizeof(DSV_SBC)));
    for (int i = 0; i < 3; i++) {
        c[i].width = cw[i];
        c[i].height = ch[i];
    }
    c[0].data = base;
    c[1].data = base + n0;
    c[2].data = base + n0 + n1;
}

static void plane_dims(int format, int width, int height, int pw[3], int ph[3])
{
    int hs = DSV_FORMAT_H_SHIFT(format), vs = DSV_FORMAT_V_SHIFT(format);
    pw[0] = width;
    ph[0] = height;
    pw[1] = pw[2] = (width + (1 << hs) - 1) >> hs;
    ph[1] = ph[2] = (height + (1 << vs) - 1) >> vs;
}

DSV_FRAME *dsv_mk_frame(int format, int width, int height, int border)
{
    DSV_FRAME *f = (DSV_FRAME *) dsv_alloc(sizeof(DSV_FRAME));
    int pw[3], ph[3];
    int ext = border ? dsv2::kBorder : 0;
    size_t total = 0, off[3];

    plane_dims(format, width, height, pw, ph);
    f->refcount = 1;
    f->format = format;
    f->width = width;
    f->height = height;
    f->border = border ? 1 : 0;
    for (int c = 0; c < 3; c++) {
        DSV_PLANE *p = &f->planes[c];
        p->format = format;
        p->w = pw[c];
        p->h = ph[c];
        p->stride = (pw[c] + 2 * ext + 15) & ~15;
        p->len = p->stride * (ph[c] + 2 * ext);
        off[c] = total;
        total += (size_t) p->len;
    }
    f->alloc = (uint8_t *) dsv_alloc((int) total);
    for (int c = 0; c < 3; c++) {
        f->planes[c].data = f->alloc + off[c] + (size_t) f->planes[c].stride * ext + ext;
    }
    return f;
}

} // extern "C"

namespace dsv2 {
// a bordered frame whose pixel block is pinned host memory from the pool (contents undefined)
DSV_FRAME *mk_frame_pinned(int format, int width, int height)
{
    DSV_FRAME *f = (DSV_FRAME *) dsv_alloc(sizeof(DSV_FRAME));
    int pw[3], ph[3];
    size_t total = 0, off[3];
    plane_dims(format, width, height, pw, ph);
    f->refcount = 1;
    f->format = format;
    f->width = width;
    f->height = height;
    f->border = 1;
    for (int c = 0; c < 3; c++) {
        DSV_PLANE *p = &f->planes[c];
        p->format = format;
        p->w = pw[c];
        p->h = ph[c];
        p->stride = (pw[c] + 2 * kBorder + 15) & ~15;
        p->len = p->stride * (ph[c] + 2 * kBorder);
        off[c] = total;
        total += (size_t) p->len;
    }
    f->alloc = (uint8_t *) pinned_pool_take(total);
    for (int c = 0; c < 3; c++) {
        f->planes[c].data = f->alloc + off[c] + (size_t) f->planes[c].stride * kBorder + kBorder;
    }
    return f;
}
} // namespace dsv2

extern "C" {

DSV_FRAME *dsv_load_planar_frame(int format, void *data, int width, int height)
{
    DSV_FRAME *f = (DSV_FRAME *) dsv_alloc(sizeof(DSV_FRAME));
    int pw[3], ph[3];
    uint8_t *p = (uint8_t *) data;

    plane_dims(format, width, height, pw, ph);
    f->refcount = 1;
    f->format = format;
    f->width = width;
    f->height = height;
    for (int c = 0; c < 3; c++) {
        DSV_PLANE *pl = &f->planes[c];
        pl->format = format;
        pl->w = pw[c];
        pl->h = ph[c];
        pl->stride = pw[c];
        pl->len = pw[c] * ph[c];
        pl->data = p;
        p += pl->len;
    }
    return f; /* alloc stays NULL: the pixels belong to the caller */
}

DSV_FRAME *dsv_frame_ref_inc(DSV_FRAME *frame)
{
    if (!frame || frame->refcount <= 0) {
        dsv2::fatal("dsv_frame_ref_inc on a dead frame", __FILE__, __LINE__);
    }
    frame->refcount++;
    return frame;
}

void dsv_frame_ref_dec(DSV_FRAME *frame)
{
    if (!frame || frame->refcount <= 0) {
        dsv2::fatal("dsv_frame_ref_dec on a dead frame", __FILE__, __LINE__);
    }
    if (--frame->refcount == 0) {
        if (frame->alloc) {
            dsv_free(frame->alloc);
        }
        dsv_free(frame);
    }
}

DSV_FRAME *dsv_clone_frame(DSV_FRAME *s, int border)
{
    DSV_FRAME *d = dsv_mk_frame(s->format, s->width, s->height, border);
    dsv_frame_copy(d, s); /* extends the border on the GPU when `border` is set */
    return d;
}

void dsv_plane_xy(DSV_FRAME *frame, DSV_PLANE *out, int c, int x, int y)
{
    DSV_PLANE *p = &frame->planes[c];
    out->format = p->format;
    out->data = p->data + x + (ptrdiff_t) y * p->stride;
    out->stride = p->stride;
    out->len = p->len;
    out->w = p->w - x > 0 ? p->w - x : 0;
    out->h = p->h - y > 0 ? p->h - y : 0;
}

/* ---- raw planar YUV file I/O (dsv.c:109-305) ---- */

static size_t chroma_bytes(int w, int h, int subsamp)
{
    switch (subsamp) {
        case DSV_SUBSAMP_444: return (size_t) w * h;
        case DSV_SUBSAMP_422: return (size_t) (w / 2) * h;
        case DSV_SUBSAMP_420:
        case DSV_SUBSAMP_411: return (size_t) w * h / 4;
        case DSV_SUBSAMP_410: return (size_t) w * h / 16;
        default: dsv2::fatal("unsupported chroma format", __FILE__, __LINE__);
    }
}

static int classify_short_read(FILE *in, size_t got, size_t framesz)
{
    if (got == 0) {
        return -2; /* clean end of input */
    }
    long pos = ftell(in);
    if (pos >= 0 && ((size_t) pos % framesz) == 0) {
        return -2;
    }
    return -1;
}

int dsv_yuv_read_seq(FILE *in, uint8_t *o, int w, int h, int subsamp)
{
    if (!in) {
        return -1;
    }
    size_t framesz = (size_t) w * h + 2 * chroma_bytes(w, h, subsamp);
    size_t got = fread(o, 1, framesz, in);
    return got == framesz ? 0 : classify_short_read(in, got, framesz);
}

int dsv_yuv_read(FILE *in, int fno, uint8_t *o, int w, int h, int subsamp)
{
    if (!in || fno < 0) {
        return -1;
    }
    if (subsamp == DSV_SUBSAMP_UYVY) { /* packed 4:2:2 -> planar */
        size_t line = (size_t) w * 2;
        uint8_t *y = o, *u = o + (size_t) w * h, *v = u + (size_t) (w / 2) * h;
        uint8_t *tmp = (uint8_t *) malloc(line);
        if (fseek(in, (long) ((size_t) fno * w * h * 2), SEEK_SET)) {
            free(tmp);
            return -1;
        }
        for (int j = 0; j < h; j++) {
            if (fread(tmp, 1, line, in) != line) {
                free(tmp);
                return -1;
            }
            for (int i = 0; i < w / 2; i++) {
                *u++ = tmp[4 * i];
                *y++ = tmp[4 * i + 1];
                *v++ = tmp[4 * i + 2];
                *y++ = tmp[4 * i + 3];
            }
        }
        free(tmp);
        return 0;
    }
    size_t framesz = (size_t) w * h + 2 * chroma_bytes(w, h, subsamp);
    if (fseek(in, (long) ((size_t) fno * framesz), SEEK_SET)) {
        return classify_short_read(in, 1, framesz);
    }
    size_t got = fread(o, 1, framesz, in);
    return got == framesz ? 0 : classify_short_read(in, got, framesz);
}

int dsv_yuv_write_seq(FILE *out, DSV_PLANE *p)
{
    if (!out) {
        return -1;
    }
    for (int c = 0; c < 3; c++) {
        for (int y = 0; y < p[c].h; y++) {
            if (fwrite(p[c].data + (size_t) y * p[c].stride, (size_t) p[c].w, 1, out) != 1) {
                return -1;
            }
        }
    }
    return 0;
}

int dsv_yuv_write(FILE *out, int fno, DSV_PLANE *p)
{
    if (!out || fno < 0) {
        return -1;
    }
    size_t framesz = (size_t) p[0].w * p[0].h + (size_t) p[1].w * p[1].h + (size_t) p[2].w * p[2].h;
    if (fseek(out, (long) ((size_t) fno * framesz), SEEK_SET)) {
        return -1;
    }
    return dsv_yuv_write_seq(out, p);
}

} // extern "C"
