// fl_jpegdec.hip -- device half of the JPEG decode front end (gfx950): dequantisation + integer IDCT per 8x8 block, then
// chroma up-sampling + YCbCr -> RGB per pixel, with the arithmetic of zune-jpeg 0.4.14 as restated in
// oracle/fanlin_oracle_jpegdec.c (reference src/handler.rs:205-220).  Integer work throughout: results are required to be
// bit-identical to that oracle.
//
// jpeg_idct_kernel   one 8-lane group per block, 32 blocks per workgroup.  The block's coefficients arrive in zig-zag order,
//                    only up to the last non-zero one (the host decoder's compact blob); they are dequantised while being
//                    scattered into an LDS tile (9-word row pitch, 73-word block pitch: column and row passes are both
//                    conflict free), lane t then runs the butterfly on column t, then on row t, and stores 8 samples.
// jpeg_color_kernel  round 5: one thread = 4 pixels x 2 rows of a three-component YCbCr picture (4:2:0, 4:2:2 or 4:4:4) -- the luma
//                    and the three chroma rows the pair needs as unaligned dword loads, Cb / Cr interpolated (separable
//                    (3a + b + 2) >> 2 steps, vertical first), converted, stored as three dwords per row; a flat grid over
//                    (width / 4) x (height / 2) groups, so no lane idles on a 1920-pixel row.  The ends of a row are part of the
//                    fast form (color_group); grayscale, CMYK / YCCK, RGB and unusual samplings take the pixel-wise form (round 2's kernel).
// Both are memory-light (a few MB per picture) and sit in front of the resample kernel, whose input they produce in HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fl_jpegdec.h"

namespace fl {

namespace {

__constant__ uint8_t kUnzig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// the LDS words (9-word row pitch) coefficients 8 t .. 8 t + 7 of the zig-zag sequence go to, one byte each: ONE 8-byte load per thread of the
// fast form (eight byte loads from kUnzig before -- vector memory instructions, 720,000 of them per batch)
__constant__ uint64_t kUnzigTile[8] = {0x0b03020a12090100ull, 0x05040c141c241b13ull, 0x262e362d251d150dull, 0x1f170f07060e161eull, 0x283038403f372f27ull, 0x3931292119101820ull, 0x332b222a323a4241ull, 0x46453d343c44433bull};

constexpr int BLK_PITCH = 73; // 8 rows x 9 words + 1
constexpr int BLOCKS_PER_WG = 32;

__device__ __forceinline__ uint32_t clamp8(int v) { return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// clamp8(v >> 17), written as clamp-then-shift.  The natural form lets hipcc (ROCm 7.2) fuse pairs of them into gfx950's
// v_ashr_pk_u8_i32, whose destination keeps stale bits above bit 15 while the compiler treats them as zero: OR-ing the other
// two samples of a dword onto it corrupted bytes 2 and 3 of every stored dword (found by the bit-exact decode test).
__device__ __forceinline__ uint32_t sat17(int v)
{
    v = v < 0 ? 0 : (v > 0x1FFFFFF ? 0x1FFFFFF : v);
    return (uint32_t)v >> 17;
}

// the 1-D butterfly shared by both passes (stb_image / zune-jpeg idct_int): d[0..7] -> o[0..7] before the final shift
__device__ __forceinline__ void butterfly(const int *d, int bias, int *o)
{
    int p2 = d[2], p3 = d[6];
    int p1 = (p2 + p3) * 2217;
    int t2 = p1 + p3 * -7567;
    int t3 = p1 + p2 * 3135;
    p2 = d[0]; p3 = d[4];
    int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
    const int x0 = t0 + t3 + bias, x3 = t0 - t3 + bias, x1 = t1 + t2 + bias, x2 = t1 - t2 + bias;
    t0 = d[7]; t1 = d[5]; t2 = d[3]; t3 = d[1];
    p3 = t0 + t2;
    int p4 = t1 + t3;
    p1 = t0 + t3; p2 = t1 + t2;
    const int p5 = (p3 + p4) * 4816;
    t0 *= 1223; t1 *= 8410; t2 *= 12586; t3 *= 6149;
    p1 = p5 + p1 * -3685; p2 = p5 + p2 * -10497; p3 = p3 * -8034; p4 = p4 * -1597;
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    o[0] = x0 + t3; o[7] = x0 - t3;
    o[1] = x1 + t2; o[6] = x1 - t2;
    o[2] = x2 + t1; o[5] = x2 - t1;
    o[3] = x3 + t0; o[4] = x3 - t0;
}

// The blob's header is written by the host or by an earlier kernel and only read here: through the constant address space its
// fields are scalar loads (through a flat pointer the compiler must assume the kernel's own stores may change them, and every
// thread loaded every field it used -- 36 vector loads in the colour kernel).
typedef const __attribute__((address_space(4))) JpegBlobHeader *HdrPtr;
typedef const __attribute__((address_space(4))) JpegComponent &CompRef;

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegDecJob *__restrict__ jobs)
{
    __shared__ int tile[BLOCKS_PER_WG * BLK_PITCH];
    const JpegDecJob jb = jobs[blockIdx.y];
    const HdrPtr H = (HdrPtr)(uintptr_t)jb.blob;
    const uint32_t tid = threadIdx.x, t = tid & 7u, lb = tid >> 3;
    const uint32_t b = blockIdx.x * BLOCKS_PER_WG + lb;
    const uint32_t nblocks = H->nblocks;
    const bool live = b < nblocks;
    int *my = tile + lb * BLK_PITCH;
    // component of this block (block words are grouped by component)
    uint32_t ci = 0;
    if (live) { for (uint32_t k = 1; k < H->nc; ++k) if (b >= H->comp[k].block_base) ci = k; }
    const uint32_t word = live ? reinterpret_cast<const uint32_t *>(jb.blob + H->blocks_off)[b] : 0u;
    const uint32_t cnt = ((word >> 1) & 63u) + 1u, wide = word & 1u, first = word >> 7;
    const uint8_t *data = jb.blob + H->coef_off + (size_t)first * 2u;
    const int16_t *c16 = reinterpret_cast<const int16_t *>(data);
    const int8_t *c8 = reinterpret_cast<const int8_t *>(data) + 2u * kJpegWideHead;
    const bool full = live && wide && cnt == 64u && (reinterpret_cast<uintptr_t>(c16) & 15u) == 0u; // the same for the block's eight lanes
    if (!full) { // (a full block writes all 64 words of its tile itself)
#pragma unroll
        for (int k = 0; k < 9; ++k) my[t * 9 + k] = 0;
    }
    __syncthreads();
    if (live) {
        if (full) {
            // a full "wide" block (every block the device's entropy decoder writes): thread t takes coefficients 8 t .. 8 t + 7 of the
            // zig-zag sequence as ONE 16-byte load, their quantiser steps as another (the general loop: eight 2-byte loads each)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 cv = *(const __attribute__((address_space(1))) u32x4 *)(uintptr_t)(c16 + 8u * t);
            const u32x4 qv = *reinterpret_cast<const __attribute__((address_space(4))) u32x4 *>(&H->qt[ci][8u * t]);
            const uint64_t where = kUnzigTile[t];
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e) {
                const int q = (int)(int16_t)(cv[e >> 1] >> (16u * (e & 1u)));
                const int v = q * (int)(uint16_t)(qv[e >> 1] >> (16u * (e & 1u)));
                my[(uint32_t)(where >> (8u * e)) & 255u] = v;
            }
        } else
        for (uint32_t k = t; k < cnt; k += 8u) {
            const int q = (wide || k < kJpegWideHead) ? (int)c16[k] : (int)c8[k - kJpegWideHead];
            const int v = q * (int)H->qt[ci][k]; // dequantised in i32, as zune-jpeg does while decoding
            const uint32_t nat = kUnzig[k];
            my[(nat >> 3) * 9 + (nat & 7u)] = v;
        }
    }
    __syncthreads();
    int d[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = my[r * 9 + t]; // column t
    butterfly(d, 512, o);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) my[r * 9 + t] = o[r] >> 10;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = my[t * 9 + c]; // row t
    butterfly(d, 65536 + (128 << 17), o);
    if (live) {
        CompRef C = H->comp[ci];
        const uint32_t bi = b - C.block_base, by = bi / C.bw, bx = bi - by * C.bw;
        uint8_t *p = jb.planes + C.plane_off + (size_t)(by * 8u + t) * (C.bw * 8u) + bx * 8u;
        const uint32_t lo = sat17(o[0]) | (sat17(o[1]) << 8) | (sat17(o[2]) << 16) | (sat17(o[3]) << 24);
        const uint32_t hi = sat17(o[4]) | (sat17(o[5]) << 8) | (sat17(o[6]) << 16) | (sat17(o[7]) << 24);
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *(__attribute__((address_space(1))) u32x2 *)(uintptr_t)p = u32x2{lo, hi};
    }
}

// zune-jpeg's separable chroma interpolation at full-resolution position (x, y); sh / sv = 1 or 2
__device__ __forceinline__ int chroma_at(const uint8_t *plane, uint32_t pw, uint32_t cw, uint32_t chh, uint32_t sh, uint32_t sv, uint32_t x, uint32_t y)
{
    if (sv == 1u) {
        const uint8_t *row = plane + (size_t)y * pw;
        if (sh == 1u) return row[x];
        const uint32_t i = x >> 1;
        if (cw == 1u || x == 0u || x == 2u * cw - 1u) return row[i];
        return (x & 1u) ? (3 * row[i] + row[i + 1] + 2) >> 2 : (3 * row[i] + row[i - 1] + 2) >> 2;
    }
    const uint32_t r = y >> 1;
    int fr = (y & 1u) ? (int)r + 1 : (int)r - 1;
    fr = fr < 0 ? 0 : (fr > (int)chh - 1 ? (int)chh - 1 : fr);
    const uint8_t *nr = plane + (size_t)r * pw, *fa = plane + (size_t)fr * pw;
    if (sh == 1u) return (3 * nr[x] + fa[x] + 2) >> 2;
    const uint32_t ix = x >> 1;
    const int a = (3 * nr[ix] + fa[ix] + 2) >> 2;
    if (cw == 1u || x == 0u || x == 2u * cw - 1u) return a;
    const uint32_t k = (x & 1u) ? ix + 1u : ix - 1u;
    const int b = (3 * nr[k] + fa[k] + 2) >> 2;
    return (3 * a + b + 2) >> 2;
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
// (pointers read from a descriptor are flat to the compiler; planes and pixels are device memory by contract: global loads and stores)
typedef const __attribute__((address_space(1))) uint8_t *gcptr8;
typedef __attribute__((address_space(1))) uint8_t *gptr8;

// Four bytes at any address of a plane as ONE dword-aligned 8-byte load and a funnel shift (the chroma samples ix - 1 .. ix + 2 of a
// group start at an odd byte: a misaligned dword load is split by the memory pipeline).  Reads up to 3 bytes past the four -- inside
// the picture's scratch (the planes are followed by its pixels, fl_batch.cpp).
__device__ __forceinline__ uint32_t load4_at(gcptr8 p)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 __attribute__((aligned(4))) u32x2_a4;
    const uintptr_t a = (uintptr_t)p;
    const u32x2 w = *(const __attribute__((address_space(1))) u32x2_a4 *)(a & ~(uintptr_t)3);
    return __builtin_amdgcn_alignbit(w.y, w.x, ((uint32_t)a & 3u) * 8u);
}

// one pixel, every case: planes through zune-jpeg's interpolation (a plain read at full resolution), then the colour transform
__device__ __forceinline__ void color_pixel(const JpegDecJob &jb, HdrPtr H, uint32_t x, uint32_t y)
{
    const uint32_t nc = H->nc;
    uint8_t *o = jb.dst + ((size_t)y * H->width + x) * nc;
    int s[4];
    for (uint32_t i = 0; i < nc; ++i) {
        CompRef C = H->comp[i];
        s[i] = chroma_at(jb.planes + C.plane_off, C.bw * 8u, C.w, C.hpx, H->hmax / C.h, H->vmax / C.v, x, y);
    }
    if (nc == 1u) { o[0] = (uint8_t)s[0]; return; }
    if (nc == 4u) { // CMYK / YCCK: the raw samples, as JpegDecoder with out_colorspace = the input colour space returns them (handler.rs:417-419)
        o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; o[3] = (uint8_t)s[3];
        return;
    }
    if (H->is_rgb) { o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; return; }
    // zune-jpeg color_convert/scalar.rs: i16 arithmetic with 5/6-bit constants, arithmetic shifts
    const int cb = s[1] - 128, cr = s[2] - 128;
    o[0] = (uint8_t)clamp8(s[0] + ((45 * cr) >> 5));
    o[1] = (uint8_t)clamp8(s[0] - ((11 * cb + 23 * cr) >> 5));
    o[2] = (uint8_t)clamp8(s[0] + ((113 * cb) >> 6));
}

// Pixels x0 .. x0 + 3 (x0 a multiple of 4; fewer at the end of a row whose width is not) of rows y and y + 1 (y even), SH x SV chroma sampling.
// The ends of a row are part of it: zune-jpeg takes the nearest chroma sample as it is for the first pixel and for pixel 2 * c_w - 1
// (chroma_at above), which is the interpolation formula with the missing neighbour replaced by the sample itself -- (3 a + a + 2) >> 2 = a --
// so the group at x0 = 0 copies sample 0 over the one in front of the row and the last group copies sample c_w - 1 over those behind it.
// (Until round 5 the two end groups of every row went through color_pixel, one pixel and one header walk at a time, and one wave in four
// carried such a lane: more than half of the kernel's vector instructions.)
template <int SH, int SV>
__device__ __forceinline__ void color_group(const JpegDecJob &jb, uint32_t x0, uint32_t y, uint32_t rows)
{
    const uint32_t W = jb.width;
    gcptr8 py = (gcptr8)(jb.planes + jb.y_off);
    const uint32_t ypw = jb.y_pitch;
    uint32_t yv[2];
    yv[0] = *(const __attribute__((address_space(1))) uint32_t *)(py + (size_t)y * ypw + x0);
    yv[1] = rows > 1u ? *(const __attribute__((address_space(1))) uint32_t *)(py + (size_t)(y + 1u) * ypw + x0) : 0u;
    int cv[2][2][4]; // [Cb, Cr][row][pixel]
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        gcptr8 pl = (gcptr8)(jb.planes + (ci ? jb.cr_off : jb.cb_off));
        const uint32_t pw = jb.c_pitch;
        if (SH == 1 && SV == 1) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const uint32_t w = (rr == 0 || rows > 1u) ? *(const __attribute__((address_space(1))) uint32_t *)(pl + (size_t)(y + rr) * pw + x0) : 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) cv[ci][rr][k] = (int)((w >> (8 * k)) & 255u);
            }
            continue;
        }
        // samples ix - 1 .. ix + 2 (ix = x0 / 2) of the rows involved, vertically interpolated first
        const uint32_t ix = x0 >> 1;
        int a[2][4];
        if (SV == 1) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const uint32_t w = (rr == 0 || rows > 1u) ? load4_at(pl + (size_t)(y + rr) * pw + ix - 1u) : 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[rr][k] = (int)((w >> (8 * k)) & 255u);
            }
        } else {
            const uint32_t r = y >> 1; // both rows of the pair share the near chroma row; the far ones are r - 1 and r + 1, clamped
            const uint32_t ru = r ? r - 1u : 0u, rd = min(r + 1u, jb.c_rows - 1u);
            const uint32_t wn = load4_at(pl + (size_t)r * pw + ix - 1u), wu = load4_at(pl + (size_t)ru * pw + ix - 1u), wd = load4_at(pl + (size_t)rd * pw + ix - 1u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = (int)((wn >> (8 * k)) & 255u);
                a[0][k] = (3 * n + (int)((wu >> (8 * k)) & 255u) + 2) >> 2;
                a[1][k] = (3 * n + (int)((wd >> (8 * k)) & 255u) + 2) >> 2;
            }
        }
        const uint32_t last = jb.c_w - ix; // slot of the row's last sample (slot s holds sample ix - 1 + s): >= 1 for every group that has a pixel
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) { // a[rr][1] = sample ix, a[rr][2] = ix + 1
            a[rr][0] = x0 ? a[rr][0] : a[rr][1];
            a[rr][2] = last < 2u ? a[rr][1] : a[rr][2];
            a[rr][3] = last < 3u ? a[rr][2] : a[rr][3];
            cv[ci][rr][0] = (3 * a[rr][1] + a[rr][0] + 2) >> 2;
            cv[ci][rr][1] = (3 * a[rr][1] + a[rr][2] + 2) >> 2;
            cv[ci][rr][2] = (3 * a[rr][2] + a[rr][1] + 2) >> 2;
            cv[ci][rr][3] = (3 * a[rr][2] + a[rr][3] + 2) >> 2;
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        if (rr == 1 && rows < 2u) break;
        uint32_t b[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = (int)((yv[rr] >> (8 * k)) & 255u), cb = cv[0][rr][k] - 128, cr = cv[1][rr][k] - 128;
            b[3 * k] = clamp8(yy + ((45 * cr) >> 5));
            b[3 * k + 1] = clamp8(yy - ((11 * cb + 23 * cr) >> 5));
            b[3 * k + 2] = clamp8(yy + ((113 * cb) >> 6));
        }
        // (one 12-byte store per lane: a wave writes 768 contiguous bytes with ONE instruction -- three dword stores 12 bytes apart
        // touched every line three times)
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        typedef u32x3 __attribute__((aligned(1))) u32x3_unaligned;
        const u32x3 px = {b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24), b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24)};
        gptr8 o = (gptr8)jb.dst + ((size_t)(y + rr) * W + x0) * 3u;
        if (x0 + 4u <= W) *(__attribute__((address_space(1))) u32x3_unaligned *)o = px;
        else { // the last one to three pixels of a row
#pragma unroll
            for (uint32_t i = 0; i < 9u; ++i)
                if (i < 3u * (W - x0)) o[i] = (uint8_t)b[i];
        }
    }
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const JpegDecJob *__restrict__ jobs)
{
    const JpegDecJob jb = jobs[blockIdx.y];
    const uint32_t W = jb.width, Hh = jb.height;
    const uint32_t ngx = (W + 3u) / 4u, ngy = (Hh + 1u) / 2u;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= ngx * ngy) return;
    const uint32_t gy = idx / ngx, gx = idx - gy * ngx, x0 = 4u * gx, y = 2u * gy;
    const uint32_t rows = min(2u, Hh - y);
    // the fast form: YCbCr, luma at full resolution, both chroma planes 2x2, 2x1 or 1x1 (jpeg_color_job)
    const uint32_t mode = jb.mode;
    if (mode == 1u) color_group<2, 2>(jb, x0, y, rows);
    else if (mode == 2u) color_group<2, 1>(jb, x0, y, rows);
    else if (mode == 3u) color_group<1, 1>(jb, x0, y, rows);
    else {
        const HdrPtr H = (HdrPtr)(uintptr_t)jb.blob;
        for (uint32_t rr = 0; rr < rows; ++rr)
            for (uint32_t k = 0; k < 4u && x0 + k < W; ++k) color_pixel(jb, H, x0 + k, y + rr);
    }
}

} // namespace

hipError_t launch_jpeg_decode(const JpegDecJob *jobs, uint32_t njobs, uint32_t max_blocks, uint32_t max_w, uint32_t max_h, hipStream_t st)
{
    if (!njobs) return hipSuccess;
    // grid.y / grid.z are limited to 65535: callers split larger batches
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + BLOCKS_PER_WG - 1) / BLOCKS_PER_WG, njobs), dim3(256), 0, st, jobs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t groups = ((max_w + 3u) / 4u) * ((max_h + 1u) / 2u);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((groups + 255u) / 256u, njobs), dim3(256), 0, st, jobs);
    return hipGetLastError();
}

} // namespace fl
