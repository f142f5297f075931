// frame.hip -- picture helpers that feed motion estimation / compensation and therefore
// the bitstream: border extension, 2x luma decimation, plane copies.
//
// Replaces reference src/frame.c: extend_plane (:358, with downsample_strip :251),
// dsv_ds2x_frame_luma (:211), dsv_frame_copy (:186).
//
// Border extension is NOT edge replication: each side strip is the rounded mean of
// groups of 4 edge pixels ((a+b+c+d+2)>>2, a trailing partial group takes a plain
// truncating mean), replicated across the 32-pixel border; corners average the two
// adjacent strip ends (frame.c:377-380).
#include "dev.h"
#include "prio.h"

namespace dsv2 {

// mean of edge group k taken along a line of `n` pixels starting at p with pixel step `step`
__device__ __forceinline__ int strip_value(const uint8_t *p, int step, int n, int k)
{
    int full = n >> 2;
    if (k < full) {
        const uint8_t *q = p + (size_t) (4 * k) * step;
        return (q[0] + q[step] + q[2 * step] + q[3 * step] + 2) >> 2;
    }
    int rem = n & 3, sum = 0;
    if (rem == 0) {
        return 0; // the reference's zero-initialised strip entry past the last group
    }
    const uint8_t *q = p + (size_t) (4 * full) * step;
    for (int i = 0; i < rem; i++) {
        sum += q[i * step];
    }
    return sum / rem;
}

// One work item per 4-byte piece of border.  Roles by linear id:
//   [0, 16h)                   : row y, side, piece -> the 32-pixel runs left and right of the row
//   [.., + 64 * ngx)           : border row, side, column group -> the 32 rows above / below
//   [.., + 4 * 32 * 8)         : the four corners
__device__ __forceinline__ void put4(uint8_t *dst, int v, int n)
{
    if (n == 4 && (((uintptr_t) dst) & 3) == 0) {
        *(uint32_t *) dst = (uint32_t) v * 0x01010101u;
    } else {
        for (int i = 0; i < n; i++) {
            dst[i] = (uint8_t) v;
        }
    }
}

__device__ __forceinline__ void extend_item(const DPlane &pl, int id)
{
    uint8_t *base = pl.data;
    int stride = pl.stride, w = pl.w, h = pl.h;
    int ngx = (w + 3) >> 2;
    if (id < 16 * h) {
        int y = id >> 4, side = (id >> 3) & 1, piece = id & 7;
        int v = strip_value(side ? base + (w - 1) : base, stride, h, y >> 2);
        uint8_t *row = base + (size_t) y * stride;
        put4(side ? row + w + 4 * piece : row - kBorder + 4 * piece, v, 4);
        return;
    }
    id -= 16 * h;
    if (id < 2 * kBorder * ngx) {
        int k = id % ngx, j = (id / ngx) % kBorder, side = id / (ngx * kBorder);
        int v = strip_value(side ? base + (size_t) (h - 1) * stride : base, 1, w, k);
        uint8_t *line = side ? base + (size_t) (h + j) * stride : base - (size_t) (j + 1) * stride;
        put4(line + 4 * k, v, min(4, w - 4 * k));
        return;
    }
    id -= 2 * kBorder * ngx;
    if (id < 4 * kBorder * 8) {
        int piece = id & 7, j = (id >> 3) % kBorder, corner = id / (8 * kBorder);
        int lastx = (w >> 2) - 1, lasty = (h >> 2) - 1;
        int v;
        uint8_t *dst;
        if (corner == 0) { // top-left
            v = (strip_value(base, 1, w, 0) + strip_value(base, stride, h, 0) + 1) >> 1;
            dst = base - (size_t) (j + 1) * stride - kBorder;
        } else if (corner == 1) { // top-right
            v = (strip_value(base, 1, w, lastx) + strip_value(base + (w - 1), stride, h, 0) + 1) >> 1;
            dst = base - (size_t) (j + 1) * stride + w;
        } else if (corner == 2) { // bottom-left
            v = (strip_value(base, stride, h, lasty) + strip_value(base + (size_t) (h - 1) * stride, 1, w, 0) + 1) >> 1;
            dst = base + (size_t) (h + j) * stride - kBorder;
        } else { // bottom-right
            v = (strip_value(base + (size_t) (h - 1) * stride, 1, w, lastx) + strip_value(base + (w - 1), stride, h, lasty) + 1) >> 1;
            dst = base + (size_t) (h + j) * stride + w;
        }
        put4(dst + 4 * piece, v, 4);
    }
}

static int extend_items(int w, int h) { return 16 * h + 2 * kBorder * ((w + 3) >> 2) + 4 * kBorder * 8; }

// tab == nullptr: the single plane `one`; otherwise blockIdx.y indexes a device table of planes
__global__ __launch_bounds__(256) void k_extend(const DPlane *__restrict__ tab, DPlane one)
{
    DSV2_KERNEL_PRIO();
    const DPlane pl = job_of(tab, blockIdx.y, one);
    extend_item(pl, blockIdx.x * blockDim.x + threadIdx.x);
}

void extend_plane(hipStream_t s, const DPlane &p)
{
    DSV2_LAUNCH(k_extend, dim3((extend_items(p.w, p.h) + 255) / 256), dim3(256), 0, s, nullptr, p);
}

void extend_frame(hipStream_t s, const DFrame &f, bool luma_only)
{
    for (int c = 0; c < (luma_only ? 1 : 3); c++) {
        extend_plane(s, f.p[c]);
    }
}

// n planes (device table d_planes); max_w / max_h bound the largest of them
void extend_planes(hipStream_t s, const DPlane *d_planes, int n, int max_w, int max_h)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_extend, dim3((extend_items(max_w, max_h) + 255) / 256, n), dim3(256), 0, s, d_planes, DPlane{});
}

// 2x2 rounded mean decimation of the luma plane (frame.c:211-234)
__global__ __launch_bounds__(256) void k_ds2x(const PlanePair *__restrict__ tab, PlanePair one)
{
    const PlanePair pp = job_of(tab, blockIdx.z, one);
    int x = blockIdx.x * 64 + threadIdx.x;
    int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= pp.dst.w || y >= pp.dst.h) {
        return;
    }
    int sstride = pp.src.stride;
    const uint8_t *sp = pp.src.data + (size_t) (2 * y) * sstride + 2 * x;
    pp.dst.data[(size_t) y * pp.dst.stride + x] = (uint8_t) ((sp[0] + sp[1] + sp[sstride] + sp[sstride + 1] + 2) >> 2);
}

// the same, four output samples per thread: two 8-byte loads (rows 2y, 2y + 1), byte-lane sums, one dword store.
// Needs 16-byte aligned plane origins and strides (every plane made by dframe_alloc); the last partial group of a row
// goes sample by sample.
constexpr int kDsRows = 4; // output rows per thread (y, y + 4, ...): all eight 8-byte loads issued before the first average
__global__ __launch_bounds__(256) void k_ds2x4(const PlanePair *__restrict__ tab, PlanePair one)
{
    DSV2_KERNEL_PRIO();
    const PlanePair pp = job_of(tab, blockIdx.z, one);
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y0 = blockIdx.y * (4 * kDsRows) + threadIdx.y;
    if (x >= pp.dst.w || y0 >= pp.dst.h) {
        return;
    }
    const int sstride = pp.src.stride;
    const bool whole = x + 4 <= pp.dst.w;
    uint2 a[kDsRows], b[kDsRows];
#pragma unroll
    for (int r = 0; r < kDsRows; r++) {
        const int y = y0 + 4 * r < pp.dst.h ? y0 + 4 * r : y0;
        const uint8_t *sp = pp.src.data + (size_t) (2 * y) * sstride + 2 * x;
        a[r] = whole ? *(const uint2 *) sp : make_uint2(0u, 0u);
        b[r] = whole ? *(const uint2 *) (sp + sstride) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int r = 0; r < kDsRows; r++) {
        const int y = y0 + 4 * r;
        if (y >= pp.dst.h) {
            break;
        }
        const uint8_t *sp = pp.src.data + (size_t) (2 * y) * sstride + 2 * x;
        uint8_t *dp = pp.dst.data + (size_t) y * pp.dst.stride + x;
        if (whole) {
            const uint32_t m = 0x00ff00ffu;
            // per 32-bit word: bytes (p0 p1 p2 p3) -> 16-bit fields (p0 + p1, p2 + p3)
            uint32_t lo = (a[r].x & m) + ((a[r].x >> 8) & m) + (b[r].x & m) + ((b[r].x >> 8) & m) + 0x00020002u;
            uint32_t hi = (a[r].y & m) + ((a[r].y >> 8) & m) + (b[r].y & m) + ((b[r].y >> 8) & m) + 0x00020002u;
            lo = (lo >> 2) & m; // outputs 0, 1 in bytes 0 and 2
            hi = (hi >> 2) & m; // outputs 2, 3
            *(uint32_t *) dp = (lo & 0xffu) | ((lo >> 8) & 0xff00u) | ((hi & 0xffu) << 16) | ((hi >> 16) << 24);
        } else {
            for (int i = 0; x + i < pp.dst.w; i++) {
                const uint8_t *q = sp + 2 * i;
                dp[i] = (uint8_t) ((q[0] + q[1] + q[sstride] + q[sstride + 1] + 2) >> 2);
            }
        }
    }
}

void ds2x_luma(hipStream_t s, const DPlane &src, const DPlane &dst)
{
    DSV2_LAUNCH(k_ds2x, dim3((dst.w + 63) / 64, (dst.h + 3) / 4), dim3(64, 4), 0, s, nullptr, PlanePair{src, dst});
}

void ds2x_planes(hipStream_t s, const PlanePair *d_pairs, int n, int dst_w, int dst_h)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_ds2x, dim3((dst_w + 63) / 64, (dst_h + 3) / 4, n), dim3(64, 4), 0, s, d_pairs, PlanePair{});
}

// all planes of the table come from dframe_alloc (16-byte aligned origins and strides): four samples per thread
void ds2x_planes4(hipStream_t s, const PlanePair *d_pairs, int n, int dst_w, int dst_h)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_ds2x4, dim3((dst_w + 255) / 256, (dst_h + 4 * kDsRows - 1) / (4 * kDsRows), n), dim3(64, 4), 0, s, d_pairs, PlanePair{});
}

// visible pixels of all planes, device to device (frame.c:186-203 without the extension)
void copy_frame_pixels(hipStream_t s, const DFrame &dst, const DFrame &src)
{
    for (int c = 0; c < 3; c++) {
        HIPCHK(hipMemcpy2DAsync(dst.p[c].data, dst.p[c].stride, src.p[c].data, src.p[c].stride, src.p[c].w, dst.p[c].h,
                                hipMemcpyDeviceToDevice, s));
    }
}

// whole storage including borders (same geometry required)
void copy_frame_full(hipStream_t s, const DFrame &dst, const DFrame &src)
{
    HIPCHK(hipMemcpyAsync(dst.alloc, src.alloc, src.bytes, hipMemcpyDeviceToDevice, s));
}

// ---- batched copies: blockIdx.y indexes a device table --------------------------------------------
// linear copy of `bytes` (multiple of 16, 16-byte aligned both sides)
__global__ __launch_bounds__(256) void k_copy_linear(const CopyJob *__restrict__ tab)
{
    const CopyJob &j = tab[blockIdx.y];
    const uint4 *sp = (const uint4 *) j.src;
    uint4 *dp = (uint4 *) j.dst;
    size_t n = j.bytes >> 4;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        dp[i] = sp[i];
    }
}

// zero fill of `bytes` (multiple of 16) at dst; src is ignored
__global__ __launch_bounds__(256) void k_zero_linear(const CopyJob *__restrict__ tab)
{
    const CopyJob &j = tab[blockIdx.y];
    uint4 *dp = (uint4 *) j.dst;
    size_t n = j.bytes >> 4;
    const uint4 z = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        dp[i] = z;
    }
}

void zero_linear_batch(hipStream_t s, const CopyJob *d_jobs, int n, size_t max_bytes)
{
    if (n <= 0) {
        return;
    }
    size_t vecs = max_bytes >> 4;
    int gx = (int) ((vecs + 256 * 8 - 1) / (256 * 8));
    DSV2_LAUNCH(k_zero_linear, dim3(gx < 1 ? 1 : gx, n), dim3(256), 0, s, d_jobs);
}

// visible pixels of one plane to another plane of the same size (both 4-byte aligned rows)
__global__ __launch_bounds__(256) void k_copy_plane(const PlanePair *__restrict__ tab)
{
    const PlanePair &pp = tab[blockIdx.z];
    int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (y >= pp.dst.h || x >= pp.dst.w) {
        return;
    }
    const uint8_t *sp = pp.src.data + (size_t) y * pp.src.stride + x;
    uint8_t *dp = pp.dst.data + (size_t) y * pp.dst.stride + x;
    if (x + 4 <= pp.dst.w) {
        *(uint32_t *) dp = *(const uint32_t *) sp;
    } else {
        for (int i = 0; x + i < pp.dst.w; i++) {
            dp[i] = sp[i];
        }
    }
}

void copy_planes_batch(hipStream_t s, const PlanePair *d_pairs, int n, int w, int h)
{
    if (n > 0) {
        DSV2_LAUNCH(k_copy_plane, dim3((w + 1023) / 1024, h, n), dim3(256), 0, s, d_pairs);
    }
}

void copy_linear_batch(hipStream_t s, const CopyJob *d_jobs, int n, size_t max_bytes)
{
    if (n <= 0) {
        return;
    }
    size_t vecs = max_bytes >> 4;
    int gx = (int) ((vecs + 256 * 8 - 1) / (256 * 8));
    DSV2_LAUNCH(k_copy_linear, dim3(gx < 1 ? 1 : gx, n), dim3(256), 0, s, d_jobs);
}

// packed planar picture (rows of exactly w bytes, planes back to back) -> the three padded planes
__global__ __launch_bounds__(256) void k_ingest(const IngestJob *__restrict__ tab)
{
    const IngestJob &j = tab[blockIdx.z];
    int y = blockIdx.y;
    const uint8_t *sp = j.src;
    int c = 0;
    while (c < 2 && y >= j.dst[c].h) { // plane of this row
        sp += (size_t) j.dst[c].w * j.dst[c].h;
        y -= j.dst[c].h;
        c++;
    }
    const DPlane &pl = j.dst[c];
    if (y >= pl.h) {
        return;
    }
    int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= pl.w) {
        return;
    }
    const uint8_t *srow = sp + (size_t) y * pl.w + x;
    uint8_t *drow = pl.data + (size_t) y * pl.stride + x;
    if (x + 4 <= pl.w && ((((uintptr_t) srow) | ((uintptr_t) drow)) & 3) == 0) {
        *(uint32_t *) drow = *(const uint32_t *) srow;
    } else {
        for (int i = 0; i < 4 && x + i < pl.w; i++) {
            drow[i] = srow[i];
        }
    }
}

// the same for pictures whose rows are multiples of 16 bytes in every plane (all the usual sizes): 16 bytes per thread and
// load, four rows per workgroup -- an eighth of the workgroups and a quarter of the memory instructions of k_ingest
__global__ __launch_bounds__(128) void k_ingest16(const IngestJob *__restrict__ tab)
{
    DSV2_KERNEL_PRIO();
    const IngestJob &j = tab[blockIdx.y];
    const int x = (int) threadIdx.x * 16;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int y = (int) blockIdx.x * 4 + r;
        const uint8_t *sp = j.src;
        int c = 0;
        while (c < 2 && y >= j.dst[c].h) { // plane of this row
            sp += (size_t) j.dst[c].w * j.dst[c].h;
            y -= j.dst[c].h;
            c++;
        }
        const DPlane &pl = j.dst[c];
        if (y < pl.h && x < pl.w) {
            *(uint4 *) (pl.data + (size_t) y * pl.stride + x) = *(const uint4 *) (sp + (size_t) y * pl.w + x);
        }
    }
}

// interleaved UYVY 4:2:2 picture (rows of 2 * w bytes: U0 Y0 V0 Y1 ...) -> the three padded planes; the
// de-interleave of dsv_yuv_read (dsv.c:177-205) done while the picture is ingested.  One thread per 4 luma pixels
// (8 source bytes -> 4 Y, 2 U, 2 V).
__global__ __launch_bounds__(256) void k_ingest_uyvy(const IngestJob *__restrict__ tab)
{
    const IngestJob &j = tab[blockIdx.z];
    const int y = blockIdx.y, w = j.dst[0].w;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y >= j.dst[0].h || x >= w) {
        return;
    }
    const uint8_t *s = j.src + (size_t) y * (size_t) (2 * w) + (size_t) 2 * x;
    uint8_t *dy = j.dst[0].data + (size_t) y * j.dst[0].stride + x;
    uint8_t *du = j.dst[1].data + (size_t) y * j.dst[1].stride + (x >> 1);
    uint8_t *dv = j.dst[2].data + (size_t) y * j.dst[2].stride + (x >> 1);
    if (x + 4 <= w && (((uintptr_t) s) & 7) == 0) {
        uint2 q = *(const uint2 *) s; // U0 Y0 V0 Y1 | U1 Y2 V1 Y3
        uint32_t yy = ((q.x >> 8) & 0xffu) | ((q.x >> 16) & 0xff00u) | ((q.y << 8) & 0xff0000u) | (q.y & 0xff000000u);
        *(uint32_t *) dy = yy; // x is a multiple of 4 and the plane origin is 16-byte aligned
        *(uint16_t *) du = (uint16_t) ((q.x & 0xffu) | ((q.y & 0xffu) << 8));
        *(uint16_t *) dv = (uint16_t) (((q.x >> 16) & 0xffu) | (((q.y >> 16) & 0xffu) << 8));
    } else {
        for (int i = 0; i < 4 && x + i < w; i += 2) { // (w is even for this format)
            du[i >> 1] = s[2 * i];
            dy[i] = s[2 * i + 1];
            dv[i >> 1] = s[2 * i + 2];
            dy[i + 1] = s[2 * i + 3];
        }
    }
}

// ---- decoder egress: chroma planes of a decoded picture converted to 4:2:0 -----------------------------------
// util.c:79-153 of the reference CLI (-out420p): 4:4:4 -> 4:2:2 -> 4:2:0 (two rounded pair averages, the second
// operand clamped at the plane edge), 4:2:2 -> 4:2:0, 4:1:1 -> 4:2:0, 4:1:0 -> 4:2:0; one thread per output sample,
// both steps of the 4:4:4 chain fused (the intermediate 4:2:2 sample is recomputed, never stored).
__device__ __forceinline__ int px(const DPlane &p, int x, int y) { return p.data[(size_t) y * p.stride + x]; }

__global__ __launch_bounds__(256) void k_to420(const To420Job *__restrict__ tab)
{
    const To420Job &j = tab[blockIdx.z];
    const DPlane &s = j.src, &d = j.dst;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= d.w || y >= d.h) {
        return;
    }
    int v;
    if (j.mode == 0) { // luma (and chroma that is 4:2:0 already): copy
        v = px(s, x, y);
    } else if (j.mode == 1) { // 4:4:4: conv444to422 then conv422to420
        int x0 = 2 * x, x1 = x0 < s.w - 1 ? x0 + 1 : s.w - 1;
        int y0 = 2 * y, y1 = y0 < s.h - 1 ? y0 + 1 : s.h - 1;
        int a = (px(s, x0, y0) + px(s, x1, y0) + 1) >> 1, b = (px(s, x0, y1) + px(s, x1, y1) + 1) >> 1;
        v = (a + b + 1) >> 1;
    } else if (j.mode == 2) { // 4:2:2: conv422to420
        int y0 = 2 * y, y1 = y0 < s.h - 1 ? y0 + 1 : s.h - 1;
        v = (px(s, x, y0) + px(s, x, y1) + 1) >> 1;
    } else if (j.mode == 3) { // 4:1:1: conv411to420
        int y0 = 2 * y, y1 = y0 < s.h - 1 ? y0 + 1 : s.h - 1;
        int sx = min(x >> 1, s.w - 1);
        v = (px(s, sx, y0) + px(s, sx, y1) + 1) >> 1;
    } else { // "4:1:0": conv410to420
        v = px(s, min(x >> 1, s.w - 1), min(y >> 1, s.h - 1));
    }
    d.data[(size_t) y * d.stride + x] = (uint8_t) v;
}

// small planes (the coarsest pyramid level) packed row after row into pinned host memory
__global__ __launch_bounds__(64) void k_plane_to_host(const PlaneOutJob *__restrict__ tab)
{
    const PlaneOutJob &j = tab[blockIdx.y];
    int y = blockIdx.x;
    if (y >= j.src.h) {
        return;
    }
    for (int x = threadIdx.x; x < j.src.w; x += 64) {
        j.dst[(size_t) y * j.src.w + x] = j.src.data[(size_t) y * j.src.stride + x];
    }
}

void planes_to_host_batch(hipStream_t s, const PlaneOutJob *d_jobs, int n, int h)
{
    if (n > 0) {
        DSV2_LAUNCH(k_plane_to_host, dim3(h, n), dim3(64), 0, s, d_jobs);
    }
}

void ingest_batch(hipStream_t s, const IngestJob *d_jobs, int n, int w, int total_rows)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_ingest, dim3((w + 1023) / 1024, total_rows, n), dim3(256), 0, s, d_jobs);
}

// every plane's width a multiple of 16 (and at most 2048), sources 16-byte aligned: the wide form
void ingest_batch16(hipStream_t s, const IngestJob *d_jobs, int n, int total_rows)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_ingest16, dim3((total_rows + 3) / 4, n), dim3(128), 0, s, d_jobs);
}

void ingest_uyvy_batch(hipStream_t s, const IngestJob *d_jobs, int n, int w, int h)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_ingest_uyvy, dim3((w + 1023) / 1024, h, n), dim3(256), 0, s, d_jobs);
}

void to420_batch(hipStream_t s, const To420Job *d_jobs, int n, int max_w, int max_h)
{
    if (n <= 0) {
        return;
    }
    DSV2_LAUNCH(k_to420, dim3((max_w + 255) / 256, max_h, n), dim3(256), 0, s, d_jobs);
}

} // namespace dsv2
