// dev.h -- device-side plumbing shared by every kernel group of libdsv2hip.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   * pictures: planar u8, every plane surrounded by a 32-pixel border, row stride
//     rounded up to 16 bytes -- byte-identical to the reference's host layout
//     (frame.c:63-113) so that block reads that run into the border see the same bytes;
//   * coefficient planes: int32, row stride = plane width, Mallat layout (frame.c:30-60);
//   * motion vectors: 16-byte DSV_MV records, raster order; blockdata: one flag byte per block.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/dsv2_hip.h"

namespace dsv2 {

[[noreturn]] void fatal(const char *what, const char *file, int line);

#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t e__ = (expr);                                       \
        if (e__ != hipSuccess) {                                       \
            fprintf(stderr, "[dsv2hip] HIP error %d (%s) ", (int) e__, \
                    hipGetErrorString(e__));                           \
            dsv2::fatal(#expr, __FILE__, __LINE__);                    \
        }                                                              \
    } while (0)

// every kernel launch goes through this macro so that the stage profile can report launch counts
extern thread_local long long t_launch_count;
#define DSV2_LAUNCH(...)                \
    do {                                \
        hipLaunchKernelGGL(__VA_ARGS__); \
        dsv2::t_launch_count++;         \
    } while (0)

constexpr int kBorder = 32; // dsv_internal.h:38
constexpr int kBlockP = 14; // dsv_internal.h:127

struct DPlane {
    uint8_t *data; // device pointer to pixel (0,0)
    int stride;
    int w, h;
};

// A launch's job record: entry i of the table (the host wrote it before the launch; nothing writes it during) or, for the
// single-call forms, the copy in the kernel arguments -- BY VALUE, THROUGH THE SCALAR CACHE, into scalar registers.
// `tab ? tab[i] : one` bound to a reference is a pointer into one of two address spaces: the compiler then fetches every field with a
// per-lane (flat) load of the same address, the record's pointers and strides live in vector registers, and every address the kernel
// forms from them is a 64-bit vector multiply-add instead of a scalar base plus a 32-bit lane offset.  (For the small records only --
// planes, plane pairs, the compaction's job: the 144-byte PlaneJob with its arrays ends up in scratch when copied this way, which costs
// more than the flat loads did; the transform and quantiser kernels keep the reference.)
#ifdef __HIPCC__
// a value / pointer every lane holds alike, told to the compiler as such: it moves to scalar registers and what is computed from it
// (addresses above all) to the scalar unit
template <class T> __device__ __forceinline__ T *uni_ptr(T *p)
{
    unsigned long long v = (unsigned long long) p;
    unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v);
    unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (v >> 32));
    return (T *) (((unsigned long long) hi << 32) | lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ DPlane uni(const DPlane &p) { return DPlane{uni_ptr(p.data), uni(p.stride), uni(p.w), uni(p.h)}; }

template <class T> __device__ __forceinline__ T job_of(const T *tab, unsigned i, const T &one)
{
    if (tab == nullptr) {
        return one;
    }
    T j;
    __builtin_memcpy(&j, (const __attribute__((address_space(4))) void *) (tab + i), sizeof(T));
    return j;
}
#endif

struct DFrame {
    uint8_t *alloc = nullptr;
    size_t bytes = 0;
    DPlane p[3];
    size_t plane_off[3]; // byte offset of each plane's storage (incl. border) inside alloc
    size_t plane_len[3];
    int format = 0, w = 0, h = 0;
};

struct DCoefs {
    int32_t *data;
    int w, h;
};

// per-block geometry + flags needed by adaptive stages
struct BlockMap {
    const uint8_t *bd; // device blockdata (may be null when unused)
    int nbh, nbv;
};

void dev_zero(void *p, size_t bytes); // synchronous zero fill of device memory (nothing inside an arena scope: the arena is zeroed once)
// One device allocation per codec instance (round 4).  Inside an arena scope (DevArenaScope on the calling thread) dev_alloc
// hands out 256-byte aligned pieces of ONE hipMalloc'd block -- an encoder instance made ~75 allocations, each a driver call
// when it is made and a device-synchronising one when it is freed: what a short-lived process (one CLI process per closed-GOP
// segment) pays before its first packet and again on its way out.  A request the block cannot hold falls back to hipMalloc;
// dev_release frees only what does not lie inside a live arena (the block itself is freed by its owner).
struct DevArena {
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
    void create(size_t bytes); // hipMalloc + zero fill + registration
    void destroy();
};
struct DevArenaScope {
    DevArenaScope(DevArena *a);
    ~DevArenaScope();
    DevArena *prev;
};
hipError_t dev_alloc(void **p, size_t bytes);
long arena_fallbacks(); // allocations an instance arena could not hold so far (0 unless CodecDev::init's estimate has drifted)
void dev_release(void *p);
size_t dframe_bytes(int format, int w, int h); // device bytes dframe_alloc asks for
void event_wait(hipEvent_t ev);        // ... for a recorded event
void set_wait_fine(bool fine);         // this thread's waits poll at 20 us (small batches: DSV2_WAIT_FINE_MAX) instead of up to 120 us
void stream_wait(hipStream_t s);       // host wait for the stream to drain: sleeps between completion queries (dev.cpp: 10 - 120 us apart), no busy spin
// pinned host blocks the GPU may write (hostutil.cpp): recycled through dsv_free
void *pinned_pool_take(size_t bytes);
bool pinned_pool_release(void *p);
DSV_FRAME *mk_frame_pinned(int format, int width, int height);
void dframe_alloc(DFrame *f, int format, int w, int h);
void dframe_free(DFrame *f);
// copy between a host DSV_FRAME (any stride, bordered or not) and a device frame: visible pixels only
void dframe_upload(DFrame *d, const DSV_FRAME *h, hipStream_t s);
void dframe_download(const DFrame *d, DSV_FRAME *h, hipStream_t s);
// whole storage including borders (host frame must be bordered with the reference layout)
void dframe_upload_full(DFrame *d, const DSV_FRAME *h, hipStream_t s);
void dframe_download_full(const DFrame *d, DSV_FRAME *h, hipStream_t s);

void coef_dims(int format, int w, int h, int cw[3], int ch[3]);

// scratch images for the transform: three int32 planes of the luma coefficient size
struct SbtScratch {
    int32_t *t[3] = {nullptr, nullptr, nullptr};
    size_t elems = 0, elems_ll = 0;
    // t[2] (row-pass temporary) holds a whole plane: n elements.  t[0] / t[1] only ever hold LL images of level >= 1, stored
    // with the plane's row stride: (ceil(h / 2) + 1) rows of it suffice -- n_ll elements (0: as large as t[2]).
    void ensure(size_t n, size_t n_ll = 0);
    void release();
    // the three images laid out in memory the caller owns (base: 16-byte aligned, scratch_elems(n, n_ll) int32): t[2], t[0], t[1]
    bool borrowed = false;
    void borrow(int32_t *base, size_t n, size_t n_ll);
    static size_t scratch_elems(size_t n, size_t n_ll) { return ((n + 3) & ~(size_t) 3) + 2 * ((n_ll + 3) & ~(size_t) 3); }
};
inline size_t sbt_ll_elems(int cw, int ch) { return (size_t) cw * (size_t) ((ch + 1) / 2 + 1); }

// One plane of one stream as the transform and the quantiser see it.  A device table of these
// (one per stream and plane of a lockstep batch) lets a single launch serve every stream.
struct PlaneJob {
    DPlane pic;        // working picture plane: residual in (forward level 1), reconstruction out (inverse level 1)
    int32_t *coefs;    // coefficient plane
    int32_t *t[3];     // scratch images with the plane's row stride: t[2] a whole plane, t[0] / t[1] the LL images (half the rows)
    const uint8_t *bd; // per-block flag bytes
    int32_t *qv;       // dense quantised values of this plane, scan order
    const DSV_MV *mvs; // motion field (P frames)
    int q;             // frame quantiser
    int qll;           // LL step size
    int qp[3][3];      // detail step sizes [level][subband - 1]
    int *tile_count;   // quantiser: per-1024-position nonzero counts of the stream's symbol list, or null (counted later)
    unsigned qv_base;  // scan position of this plane's first value within that list
};

// --- subband transform (sbt.hip) ------------------------------------------------
// forward: u8 plane -> coefs (cw x ch).  inverse: coefs -> u8 plane (coefs preserved).
void sbt_forward(hipStream_t s, const DPlane &src, DCoefs dst, SbtScratch &sc, int plane_idx, int isP,
                 int lossless, BlockMap bm);
void sbt_inverse(hipStream_t s, DPlane dst, DCoefs src, SbtScratch &sc, int q, int plane_idx, int isP,
                 int lossless, BlockMap bm);

// table forms: n jobs of identical geometry (cw x ch coefficient plane)
void sbt_forward_jobs(hipStream_t s, const PlaneJob *d_jobs, int n, int cw, int ch, int plane_idx, int isP, int lossless, int nbh,
                      int nbv);
void sbt_inverse_jobs(hipStream_t s, const PlaneJob *d_jobs, int n, int cw, int ch, int plane_idx, int isP, int lossless, int nbh,
                      int nbv);

// --- picture helpers (frame.hip) ---------------------------------------------------
void extend_plane(hipStream_t s, const DPlane &p);
void extend_frame(hipStream_t s, const DFrame &f, bool luma_only);
void ds2x_luma(hipStream_t s, const DPlane &src, const DPlane &dst);
void copy_frame_pixels(hipStream_t s, const DFrame &dst, const DFrame &src);
void copy_frame_full(hipStream_t s, const DFrame &dst, const DFrame &src);
// stream-batched forms: one launch works through a device-resident table of jobs
struct PlanePair {
    DPlane src, dst;
};
struct CopyJob {
    const void *src;
    void *dst;
    size_t bytes;
};
struct IngestJob {
    const uint8_t *src; // packed planar picture in HBM
    DPlane dst[3];
};
struct PlaneOutJob {
    DPlane src;
    uint8_t *dst; // pinned host memory, w * h bytes
};
struct To420Job { // one plane of a decoded picture on its way into a 4:2:0 output frame (frame.hip: k_to420)
    DPlane src, dst;
    int mode; // 0 copy, 1 from 4:4:4, 2 from 4:2:2, 3 from 4:1:1, 4 from "4:1:0"
};
void to420_batch(hipStream_t s, const To420Job *d_jobs, int n, int max_w, int max_h);
void ingest_uyvy_batch(hipStream_t s, const IngestJob *d_jobs, int n, int w, int h); // src = interleaved UYVY rows
void planes_to_host_batch(hipStream_t s, const PlaneOutJob *d_jobs, int n, int h);
void extend_planes(hipStream_t s, const DPlane *d_planes, int n, int max_w, int max_h);
void ds2x_planes(hipStream_t s, const PlanePair *d_pairs, int n, int dst_w, int dst_h);
void ds2x_planes4(hipStream_t s, const PlanePair *d_pairs, int n, int dst_w, int dst_h); // planes from dframe_alloc: 4 samples per thread
void copy_linear_batch(hipStream_t s, const CopyJob *d_jobs, int n, size_t max_bytes);
void zero_linear_batch(hipStream_t s, const CopyJob *d_jobs, int n, size_t max_bytes); // dst, bytes of each job
void copy_planes_batch(hipStream_t s, const PlanePair *d_pairs, int n, int w, int h);  // visible pixels, same-size planes
void ingest_batch(hipStream_t s, const IngestJob *d_jobs, int n, int w, int total_rows);
void ingest_batch16(hipStream_t s, const IngestJob *d_jobs, int n, int total_rows); // all plane widths % 16 == 0, <= 2048, 16-byte aligned sources

void ensure_device();
void set_default_device(int ordinal);
void bind_device(); // ensure_device + hipSetDevice(default ordinal) for the calling thread
int trace_mode();                    // DSV2_TRACE: bit 0 start-up marks, bit 1 wall-clock split of the lockstep steps (bit 2: absolute times, bit 3: every step)
void startup_mark(const char *what); // DSV2_TRACE=1
int device_status(); // 0 = usable HIP device present
bool device_arch_is(const char *prefix); // the default device's gcnArchName starts with `prefix`

} // namespace dsv2
