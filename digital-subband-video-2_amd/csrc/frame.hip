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

// One thread per border row segment / column group.  Roles by linear id:
//   [0, h)                : row y       -> left and right 32-pixel runs
//   [h, h + ngx)          : column group -> top and bottom 32 rows of 4 pixels
//   [h + ngx, h + ngx + 4*32): corner rows
__global__ __launch_bounds__(256) void k_extend(uint8_t *__restrict__ base, int stride, int w, int h)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    int ngx = (w + 3) >> 2;
    if (id < h) {
        int y = id;
        int lv = strip_value(base, stride, h, y >> 2);
        int rv = strip_value(base + (w - 1), stride, h, y >> 2);
        uint8_t *row = base + (size_t) y * stride;
        for (int i = 0; i < kBorder; i++) {
            row[-kBorder + i] = (uint8_t) lv;
            row[w + i] = (uint8_t) rv;
        }
        return;
    }
    id -= h;
    if (id < ngx) {
        int k = id;
        int tv = strip_value(base, 1, w, k);
        int bv = strip_value(base + (size_t) (h - 1) * stride, 1, w, k);
        int x0 = 4 * k, x1 = min(w, x0 + 4);
        for (int j = 0; j < kBorder; j++) {
            uint8_t *t = base - (size_t) (j + 1) * stride;
            uint8_t *b = base + (size_t) (h + j) * stride;
            for (int x = x0; x < x1; x++) {
                t[x] = (uint8_t) tv;
                b[x] = (uint8_t) bv;
            }
        }
        return;
    }
    id -= ngx;
    if (id < 4 * kBorder) {
        int corner = id / kBorder, j = id % kBorder;
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
        for (int i = 0; i < kBorder; i++) {
            dst[i] = (uint8_t) v;
        }
    }
}

void extend_plane(hipStream_t s, const DPlane &p)
{
    int total = p.h + ((p.w + 3) >> 2) + 4 * kBorder;
    hipLaunchKernelGGL(k_extend, dim3((total + 255) / 256), dim3(256), 0, s, p.data, p.stride, p.w, p.h);
}

void extend_frame(hipStream_t s, const DFrame &f, bool luma_only)
{
    for (int c = 0; c < (luma_only ? 1 : 3); c++) {
        extend_plane(s, f.p[c]);
    }
}

// 2x2 rounded mean decimation of the luma plane (frame.c:211-234)
__global__ __launch_bounds__(256) void k_ds2x(const uint8_t *__restrict__ src, int sstride, uint8_t *__restrict__ dst,
                                              int dstride, int dw, int dh)
{
    int x = blockIdx.x * 64 + threadIdx.x;
    int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) {
        return;
    }
    const uint8_t *sp = src + (size_t) (2 * y) * sstride + 2 * x;
    dst[(size_t) y * dstride + x] = (uint8_t) ((sp[0] + sp[1] + sp[sstride] + sp[sstride + 1] + 2) >> 2);
}

void ds2x_luma(hipStream_t s, const DPlane &src, const DPlane &dst)
{
    hipLaunchKernelGGL(k_ds2x, dim3((dst.w + 63) / 64, (dst.h + 3) / 4), dim3(64, 4), 0, s, src.data, src.stride, dst.data,
                       dst.stride, dst.w, dst.h);
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

} // namespace dsv2
