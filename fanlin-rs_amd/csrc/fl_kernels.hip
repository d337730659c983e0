// fl_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels of the fanlin-rs image hot path.
//
// Arithmetic contract: every resample/blur tap is one fused
// f32 multiply-add applied in the reference's tap order (image 0.25.6
// imageops/sample.rs: vertical pass first into an unrounded f32 image, then
// the horizontal pass, clamp, round half away from zero).  Everything else
// (grayscale, invert, overlay/fill, YCbCr front ends) is evaluated with the
// reference's own operation order and must match it bit for bit, so this file
// is compiled with -ffp-contract=off and fuses only where __builtin_fmaf says so.
//
// No MFMA: the path has no dense contraction.  The hot kernel is
// resample_stream: it reads every source byte exactly once with coalesced
// 12/16-byte-per-lane buffer loads, keeps the <= 8 live output rows of the
// vertical pass in registers (weights are wave-uniform SGPR operands of
// v_pk_fma_f32), hands finished f32 rows to the horizontal pass through LDS,
// and writes rounded u8 pixels straight into the (letterboxed) destination.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "fl_kernels.h"
#include "fl_pixel.h"

#ifndef FL_WPREFETCH
#define FL_WPREFETCH 0 // 1 = fetch a row's weights from LDS one row ahead (measured: no gain)
#endif

#ifndef FL_RING_MULT
#define FL_RING_MULT 2    // prefetch ring of the streaming kernel = FL_RING_MULT blocks of FL_STREAM_DEPTH rows
#endif
#ifndef FL_LOAD_AUX
#define FL_LOAD_AUX 0     // cache policy bits of the streaming kernel's source-row loads (experiments: 2 = nt)
#endif
#ifndef FL_STREAM_DEPTH
#define FL_STREAM_DEPTH 4 // source rows per block of the streaming kernel
#endif

namespace fl {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

// image::color rgb_to_luma for u8: (2126 R + 7152 G + 722 B) / 10000, truncating (u32 maths).
__device__ __forceinline__ uint32_t luma_u8(uint32_t r, uint32_t g, uint32_t b)
{
    return (2126u * r + 7152u * g + 722u * b) / 10000u;
}

// FloatNearest + NumCast in horizontal_sample: clamp to [0,255], round half away from zero.
__device__ __forceinline__ uint32_t round_u8(float t)
{
    t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
    float f = __builtin_floorf(t);
    if (t - f >= 0.5f) f += 1.0f; // t - f is exact here
    return (uint32_t)f;
}

// One dword / byte to global memory, invisible to hipcc's s_waitcnt bookkeeping (see load_row below).
__device__ __forceinline__ void store_hidden_b32(void *p, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, off\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_hidden_b8(void *p, uint32_t v)
{
    asm volatile("global_store_byte %0, %1, off\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
}

// Writes one resampled pixel (MC rounded channels in c[]) to the destination.
// LB = letterboxed: destination is Rgba8, pixel converted with to_rgba() and blended onto the fill.
template <int MC, bool LB, bool HIDDEN = false>
__device__ __forceinline__ void store_pixel(uint8_t *dst, uint32_t pix_index, const uint32_t *c, uint32_t fill)
{
    if (LB) {
        uint32_t v;
        if (MC == 1) v = c[0] | (c[0] << 8) | (c[0] << 16) | (255u << 24);
        else if (MC == 2) v = blend_over_fill(fill, c[0], c[0], c[0], c[1]);
        else if (MC == 3) v = c[0] | (c[1] << 8) | (c[2] << 16) | (255u << 24);
        else v = blend_over_fill(fill, c[0], c[1], c[2], c[3]);
        if (HIDDEN) store_hidden_b32(reinterpret_cast<uint32_t *>(dst) + pix_index, v);
        else reinterpret_cast<uint32_t *>(dst)[pix_index] = v;
    } else {
        uint8_t *p = dst + (size_t)pix_index * MC;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            if (HIDDEN) store_hidden_b8(p + k, c[k]);
            else p[k] = (uint8_t)c[k];
        }
    }
}

// Applies the pre-op to one source pixel given as CS integer channels; writes MC floats.
template <int CS, int PRE>
__device__ __forceinline__ void preop_pixel(const uint32_t *s, float *v)
{
    if (PRE == PRE_GRAY && CS >= 3) {
        v[0] = (float)luma_u8(s[0], s[1], s[2]);
        if (CS == 4) v[1] = (float)s[3];
    } else if (PRE == PRE_INVERT) {
        constexpr int NC = (CS == 2 || CS == 4) ? CS - 1 : CS; // alpha is not inverted
#pragma unroll
        for (int k = 0; k < CS; ++k) v[k] = (float)(k < NC ? 255u - s[k] : s[k]);
    } else {
#pragma unroll
        for (int k = 0; k < CS; ++k) v[k] = (float)s[k];
    }
}

// ---------------------------------------------------------------------------
// Generic two-pass resample (any ratio, any size): vertical pass into an f32
// intermediate in HBM, horizontal pass out of it.  Used for up-scaling, for
// Gaussian blur (same machinery, ratio 1) and as the fallback of the fused
// streaming kernel.  Output-stationary: one thread per output sample.
// ---------------------------------------------------------------------------

template <int CS, int PRE>
__global__ __launch_bounds__(256) void vpass_generic_kernel(const Job *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                            float *__restrict__ mid, uint32_t job_base)
{
    constexpr int MC = mid_channels(CS, PRE);
    const Job jb = jobs[job_base + blockIdx.y];
    // flat over rows x columns of THIS job: narrow pictures (thumbnails) still fill their waves
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= jb.rh * jb.sw) return;
    const uint32_t oy = idx / jb.sw, x = idx - oy * jb.sw;
    const AxisTable *tab = reinterpret_cast<const AxisTable *>(arena + jb.vtab);
    const uint32_t left = arena[tab->left_off + oy];
    const uint32_t n = arena[tab->count_off + oy];
    const float *w = reinterpret_cast<const float *>(arena + tab->weights_off + arena[tab->woff_off + oy]);
    float acc[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) acc[k] = 0.0f;
    const uint8_t *p = jb.src + ((size_t)left * jb.sw + x) * CS;
    const size_t pitch = (size_t)jb.sw * CS;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t s[CS];
#pragma unroll
        for (int k = 0; k < CS; ++k) s[k] = p[k];
        float v[MC > CS ? MC : CS];
        preop_pixel<CS, PRE>(s, v);
        const float wi = w[i];
#pragma unroll
        for (int k = 0; k < MC; ++k) acc[k] = __builtin_fmaf(v[k], wi, acc[k]);
        p += pitch;
    }
    float *o = mid + (size_t)jb.mid_off + ((size_t)oy * jb.sw + x) * MC;
#pragma unroll
    for (int k = 0; k < MC; ++k) o[k] = acc[k];
}

template <int MC, bool LB, bool GROUPED>
__global__ __launch_bounds__(256) void hpass_generic_kernel(const Job *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                            const float *__restrict__ mid, uint32_t job_base)
{
    const Job jb = jobs[job_base + blockIdx.y];
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;  // flat over the kept (cropped) window of this job
    if (idx >= jb.ch * jb.cw) return;
    const uint32_t yy = idx / jb.cw, xx = idx - yy * jb.cw; // row / column inside the kept window
    const uint32_t x = jb.cx + xx, y = jb.cy + yy;
    const AxisTable *tab = reinterpret_cast<const AxisTable *>(arena + jb.htab);
    const uint32_t left = arena[tab->left_off + x];
    const uint32_t n = arena[tab->count_off + x];
    const float *w = reinterpret_cast<const float *>(arena + tab->weights_off + arena[tab->woff_off + x]);
    const float *p = mid + (size_t)jb.mid_off + ((size_t)y * jb.sw + left) * MC;
    // Horizontal summation order (also the oracle's FO_ARITH_FMA mode).  Lanczos3 resize (GROUPED): taps are
    // grouped by aligned blocks of 4 source pixels; inside a block one fused multiply-add per tap in
    // ascending order starting from 0, then the block sums are added in ascending order.  Gaussian blur:
    // one fused multiply-add per tap in tap order.
    float acc[MC], part[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) { acc[k] = 0.0f; part[k] = 0.0f; }
    for (uint32_t i = 0; i < n; ++i) {
        if (GROUPED && i != 0 && ((left + i) & 3u) == 0u) {
#pragma unroll
            for (int k = 0; k < MC; ++k) { acc[k] = acc[k] + part[k]; part[k] = 0.0f; }
        }
        const float wi = w[i];
#pragma unroll
        for (int k = 0; k < MC; ++k) part[k] = __builtin_fmaf(p[k], wi, part[k]);
        p += MC;
    }
#pragma unroll
    for (int k = 0; k < MC; ++k) acc[k] = acc[k] + part[k];
    uint32_t c[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) c[k] = round_u8(acc[k]);
    store_pixel<MC, LB>(jb.dst, (jb.oy + yy) * jb.dw + jb.ox + xx, c, jb.fill);
}

// ---------------------------------------------------------------------------
// Tiled two-pass resample (round 3): the generic path without its f32 intermediate in HBM.  One workgroup = one picture x
// one tile of kTileRows output rows x jb.pad1 output columns (a power of two the host chose so that the tile's source
// column window fits the LDS budget).  Vertical pass: wave w takes output rows w, w + 4, ...; lanes walk the window's
// source columns (coalesced byte loads, weights wave-uniform) and leave the unrounded f32 sums in LDS.  Horizontal pass:
// one thread per output pixel reads its taps from that LDS tile.  The arithmetic is the generic kernels' -- one fused
// multiply-add per tap in tap order vertically; horizontally Lanczos3 taps grouped by aligned blocks of 4 source pixels with
// the block sums added in ascending order (GROUPED, also the oracle's ARITH_FMA mode), Gaussian taps in tap order -- so the
// results are bit-identical to them.  Serves what neither the matrix-pipe nor the streaming kernel takes: up-scales, mild
// down-scales, odd pitches (image 0.25.6 imageops/sample.rs vertical_sample + horizontal_sample).
// ---------------------------------------------------------------------------
constexpr uint32_t kTileRows = 8; // (16 rows and 64 KB of LDS per workgroup left two workgroups per CU, and the kernel waited on its own loads)
constexpr uint32_t kTilePrefetch = 14; // source rows in flight per thread in the vertical pass (ratio 1: a band touches 14 rows)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned; // (gfx950 loads a dword from any byte address: one global_load_dword)

template <int CS, int PRE, bool LB, bool GROUPED>
__global__ __launch_bounds__(256) void resample_tile_kernel(const Job *__restrict__ jobs, const uint32_t *__restrict__ arena, uint32_t job_base,
                                                            uint32_t nbands)
{
    extern __shared__ float tile_mid[]; // [rows of the tile][source columns of its window][MC], then the horizontal weights of the tile's columns
    constexpr int MC = mid_channels(CS, PRE);
    const Job jb = jobs[job_base + blockIdx.y];
    const uint32_t tw = jb.pad1, tw_log = 31u - (uint32_t)__clz(tw);
    const uint32_t tiles_x = (jb.cw + tw - 1u) >> tw_log;
    if (blockIdx.x >= tiles_x * nbands) return;
    // One workgroup = one COLUMN of tiles (x range) of one band of rows: everything the horizontal pass needs -- the window's
    // first source column, its weights (staged in LDS), every thread's own (left, count, weight offset) -- is fetched once and
    // serves all tiles of the column; what is left per tile are the row tables of the vertical pass.  (A workgroup per tile spent
    // most of its 23 us in these chains of dependent table loads.)
    const uint32_t band = blockIdx.x / tiles_x, tx = blockIdx.x - band * tiles_x;
    const uint32_t band_rows = ((jb.ch + nbands - 1u) / nbands + kTileRows - 1u) / kTileRows * kTileRows;
    const uint32_t yb0 = jb.cy + band * band_rows, yb1 = min(yb0 + band_rows, jb.cy + jb.ch);
    if (yb0 >= yb1) return;
    const uint32_t x0 = jb.cx + (tx << tw_log), x1 = min(x0 + tw, jb.cx + jb.cw);
    const AxisTable *vt = reinterpret_cast<const AxisTable *>(arena + jb.vtab);
    const AxisTable *ht = reinterpret_cast<const AxisTable *>(arena + jb.htab);
    // the windows of an axis table move right monotonically: the column's window is [left of its first column, right end of its last)
    const uint32_t c0 = arena[ht->left_off + x0];
    const uint32_t ncols = arena[ht->left_off + x1 - 1u] + arena[ht->count_off + x1 - 1u] - c0;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const size_t pitch = (size_t)jb.sw * CS;
    // the horizontal weights of the columns lie back to back in the table: staged in LDS behind the f32 tile, coalesced
    // (read by output column in the horizontal pass they would be one cache line per lane)
    const uint32_t hw0 = arena[ht->woff_off + x0], hw1 = arena[ht->woff_off + x1 - 1u] + arena[ht->count_off + x1 - 1u];
    float *tile_w = tile_mid + kTileLdsFloats;
    float *tile_v = tile_w + kTileWeightFloats; // [source row of the band's window][output row of the band]: the vertical pass's weights
    const float *hweights = reinterpret_cast<const float *>(arena + ht->weights_off);
    for (uint32_t k = tid; k < hw1 - hw0; k += 256u) tile_w[k] = hweights[hw0 + k];
    // this thread's output column and the rows it takes in every tile (256 >> tw_log of them at a time)
    const uint32_t xx = tid & (tw - 1u), ysub = tid >> tw_log, ystep = 256u >> tw_log, ow = x1 - x0;
    const bool has_col = xx < ow;
    const uint32_t hx = x0 + min(xx, ow - 1u);
    const uint32_t hleft = arena[ht->left_off + hx], hn = arena[ht->count_off + hx];
    const float *hwp = tile_w + (arena[ht->woff_off + hx] - hw0);
    // vertical pass (all but the grayscale pre-op): the picture's band tables (fl_kernels.h TileVPlanHeader), and this thread's four bytes
    const TileVPlanHeader vp = *reinterpret_cast<const TileVPlanHeader *>(arena + jb.pad0);
    const uint32_t *vbands = arena + jb.pad0 + vp.bands_off;
    const float *vdense = reinterpret_cast<const float *>(arena + jb.pad0 + vp.dense_off);
    const uint32_t nbytes = ncols * (uint32_t)CS, b4 = tid * 4u;
    const uint32_t npitch = (ncols * (uint32_t)MC + 3u) & ~3u; // floats per row of the LDS tile: rows start 16-byte aligned, so a thread's four sums leave as one ds_write_b128
    const size_t col_off = (size_t)c0 * CS + b4;
    uint32_t inv = 0u; // Invert (color.rs Invert: 255 - c on the colour channels, alpha untouched) as an XOR mask of the four bytes
    if (PRE == PRE_INVERT) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t chn = (uint32_t)((col_off + j) % (uint32_t)CS);
            if (!((CS == 2 || CS == 4) && chn == (uint32_t)CS - 1u)) inv |= 0xffu << (8u * j);
        }
    }
    // Vertical pass, input-stationary by BYTE columns: without a pre-op that mixes channels a byte column is filtered like any other, so
    // a thread owns four consecutive bytes of the window, walks down the source rows the band's eight output rows touch -- each dword
    // loaded ONCE per band, kTilePrefetch of them in flight -- and feeds every row into all eight sums.  The weights are wave-uniform: a
    // dense [source row][output row] table in LDS, zero where a row lies outside an output row's window.  fma(v, 0, acc) leaves acc as
    // it is and a sum starts at +0, so each output row still sees exactly its taps, in tap order: the bits of the row-by-row form (one
    // workgroup-wide round of eight dependent loads per output row and 256 bytes; that form waited for the L2 78 % of the time,
    // profiles/r03_generic_sweep.txt).  The first rows of a band and its weight table are requested before the horizontal pass of
    // the band before it, so that their latency is spent under that pass.
    uint32_t rv = 0;                  // source rows of the band in hand (<= kTileVRows: the host checked)
    size_t goff = 0;                  // its first row's byte offset for this thread
    bool whole = false;               // every dword of this thread's column lies inside the source (all but the window's last thread in the picture's last rows)
    const uint8_t *pl = jb.src;       // the next row to request; it stops at the window's last row (requests past it repeat that row, unused)
    uint32_t ring[kTilePrefetch];
    float tvn[(kTileVRows * kTileRows) / 256u];
#pragma unroll
    for (uint32_t k = 0; k < kTilePrefetch; ++k) ring[k] = 0u;
    auto begin_band = [&](uint32_t y0) __attribute__((always_inline)) {
        const uint32_t bt = (y0 - jb.cy) / kTileRows; // band of the picture (the host's table is per picture, not per workgroup)
        const uint32_t top = vbands[2u * bt];
        rv = vbands[2u * bt + 1u];
        const float *dsrc = vdense + (size_t)bt * vp.rv_stride * kTileRows;
#pragma unroll
        for (uint32_t q = 0; q < (kTileVRows * kTileRows) / 256u; ++q) tvn[q] = tid + 256u * q < rv * kTileRows ? dsrc[tid + 256u * q] : 0.0f;
        goff = (size_t)top * pitch + col_off;
        whole = b4 < nbytes && goff + (size_t)(rv - 1u) * pitch + 4u <= (size_t)jb.src_bytes;
        if (whole) {
            pl = jb.src + goff;
#pragma unroll
            for (uint32_t k = 0; k < kTilePrefetch; ++k) {
                ring[k] = *reinterpret_cast<const u32_unaligned *>(pl);
                if (k + 1u < rv) pl += pitch;
            }
        }
    };
    auto publish_weights = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t q = 0; q < (kTileVRows * kTileRows) / 256u; ++q) tile_v[tid + 256u * q] = tvn[q];
    };
    if (PRE != PRE_GRAY) {
        begin_band(yb0);
        publish_weights();
        __syncthreads();
    }
    for (uint32_t y0 = yb0; y0 < yb1; y0 += kTileRows) {
        const uint32_t y1 = min(y0 + kTileRows, yb1);
        if (PRE != PRE_GRAY) {
            if (b4 < nbytes) {
                float acc[kTileRows][4];
#pragma unroll
                for (uint32_t o = 0; o < kTileRows; ++o)
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) acc[o][j] = 0.0f;
                auto add_row = [&](uint32_t r, uint32_t d) __attribute__((always_inline)) {
                    const f32x4 wa = *reinterpret_cast<const f32x4 *>(tile_v + r * kTileRows), wb = *reinterpret_cast<const f32x4 *>(tile_v + r * kTileRows + 4u);
                    const float v0 = (float)(d & 255u), v1 = (float)((d >> 8) & 255u), v2 = (float)((d >> 16) & 255u), v3 = (float)(d >> 24);
                    const float wr[kTileRows] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                    for (uint32_t o = 0; o < kTileRows; ++o) {
                        acc[o][0] = __builtin_fmaf(v0, wr[o], acc[o][0]);
                        acc[o][1] = __builtin_fmaf(v1, wr[o], acc[o][1]);
                        acc[o][2] = __builtin_fmaf(v2, wr[o], acc[o][2]);
                        acc[o][3] = __builtin_fmaf(v3, wr[o], acc[o][3]);
                    }
                };
                // (the choice between dwords and bytes is made once per band, outside the loops: a branch inside them puts every load in a
                // block of its own, and the compiler drains vmcnt at each join)
                if (whole) {
                    for (uint32_t rb = 0; rb < rv; rb += kTilePrefetch) {
#pragma unroll
                        for (uint32_t k = 0; k < kTilePrefetch; ++k) {
                            const uint32_t r = rb + k, d = ring[k] ^ inv;
                            if (rb + kTilePrefetch < rv) { // (wave-uniform: the last round requests nothing, the ring is free for the next band)
                                ring[k] = *reinterpret_cast<const u32_unaligned *>(pl);
                                if (r + kTilePrefetch + 1u < rv) pl += pitch;
                            }
                            if (r < rv) add_row(r, d);
                        }
                    }
                } else {
                    for (uint32_t r = 0; r < rv; ++r) {
                        const size_t a = goff + (size_t)r * pitch;
                        uint32_t d = 0u;
                        for (uint32_t j = 0; j < 4u && a + j < (size_t)jb.src_bytes; ++j) d |= (uint32_t)jb.src[a + j] << (8u * j);
                        add_row(r, d ^ inv);
                    }
                }
#pragma unroll
                for (uint32_t o = 0; o < kTileRows; ++o) // (the last thread's sums past the window land in the row's padding)
                    *reinterpret_cast<f32x4 *>(tile_mid + (size_t)o * npitch + b4) = f32x4{acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
            }
            if (y0 + kTileRows < yb1) begin_band(y0 + kTileRows); // the next band's first rows and weights: requested now, used after the horizontal pass
        } else {
            for (uint32_t ry = wave; ry < y1 - y0; ry += 4u) {
                const uint32_t oy = y0 + ry;
                const uint32_t left = arena[vt->left_off + oy], n = arena[vt->count_off + oy];
                const float *w = reinterpret_cast<const float *>(arena + vt->weights_off + arena[vt->woff_off + oy]);
                for (uint32_t col = lane; col < ncols; col += 64u) {
                    float acc[MC];
#pragma unroll
                    for (int k = 0; k < MC; ++k) acc[k] = 0.0f;
                    const uint8_t *p = jb.src + ((size_t)left * jb.sw + c0 + col) * CS;
                    for (uint32_t i = 0; i < n; ++i) {
                        uint32_t s[CS];
#pragma unroll
                        for (int k = 0; k < CS; ++k) s[k] = p[k];
                        float v[MC > CS ? MC : CS];
                        preop_pixel<CS, PRE>(s, v);
                        const float wi = w[i];
#pragma unroll
                        for (int k = 0; k < MC; ++k) acc[k] = __builtin_fmaf(v[k], wi, acc[k]);
                        p += pitch;
                    }
                    float *o = tile_mid + (size_t)ry * npitch + (size_t)col * MC;
#pragma unroll
                    for (int k = 0; k < MC; ++k) o[k] = acc[k];
                }
            }
        }
        __syncthreads();
        if (PRE != PRE_GRAY && y0 + kTileRows < yb1) publish_weights(); // (nobody reads this band's table any more; the barrier at the band's end publishes the next one)
        // Horizontal pass: a thread owns one output column and NR = tw / 32 of the tile's rows (ysub, ysub + ystep, ...), and takes
        // them through the taps TOGETHER: the weight, the block boundary of the grouped order and the tap's address are per column,
        // not per pixel (one pixel at a time spent two thirds of its vector instructions on them).
        if (has_col) {
            auto hpass = [&](auto nr_tag) __attribute__((always_inline)) {
                constexpr uint32_t NR = decltype(nr_tag)::value;
                const float *p = tile_mid + (size_t)ysub * npitch + (size_t)(hleft - c0) * MC;
                const uint32_t rstep = ystep * npitch;
                float acc[NR][MC], part[NR][MC];
#pragma unroll
                for (uint32_t q = 0; q < NR; ++q)
#pragma unroll
                    for (int k = 0; k < MC; ++k) { acc[q][k] = 0.0f; part[q][k] = 0.0f; }
                for (uint32_t i = 0; i < hn; ++i) {
                    if (GROUPED && i != 0 && ((hleft + i) & 3u) == 0u) {
#pragma unroll
                        for (uint32_t q = 0; q < NR; ++q)
#pragma unroll
                            for (int k = 0; k < MC; ++k) { acc[q][k] = acc[q][k] + part[q][k]; part[q][k] = 0.0f; }
                    }
                    const float wi = hwp[i];
#pragma unroll
                    for (uint32_t q = 0; q < NR; ++q)
#pragma unroll
                        for (int k = 0; k < MC; ++k) part[q][k] = __builtin_fmaf(p[q * rstep + k], wi, part[q][k]);
                    p += MC;
                }
#pragma unroll
                for (uint32_t q = 0; q < NR; ++q) {
                    const uint32_t yy = ysub + q * ystep;
                    if (yy >= y1 - y0) break; // (rows past a short last tile were computed on whatever the LDS held: never stored)
                    uint32_t cc[MC];
#pragma unroll
                    for (int k = 0; k < MC; ++k) cc[k] = round_u8(acc[q][k] + part[q][k]);
                    store_pixel<MC, LB>(jb.dst, (jb.oy + (y0 - jb.cy) + yy) * jb.dw + jb.ox + (x0 - jb.cx) + xx, cc, jb.fill);
                }
            };
            if (tw_log >= 8u) hpass(std::integral_constant<uint32_t, 8>{});
            else if (tw_log == 7u) hpass(std::integral_constant<uint32_t, 4>{});
            else if (tw_log == 6u) hpass(std::integral_constant<uint32_t, 2>{});
            else if (ysub < kTileRows) hpass(std::integral_constant<uint32_t, 1>{});
        }
        __syncthreads(); // the next tile's vertical pass overwrites the LDS tile
    }
}

// ---------------------------------------------------------------------------
// Pointwise placement: no resampling.  dst(x,y) = preop(src(x-ox+cx, y-oy+cy))
// converted to the destination layout, or the fill colour outside the placed
// window.  Covers grayscale/invert-only requests, letterbox-only requests and
// the border fill of resampled letterboxed images (BORDER_ONLY).
// ---------------------------------------------------------------------------

// FilterType::Nearest through image 0.25.6 sample.rs (support 0.0, box kernel): one tap of weight 1 at
// clamp(floor((o + 0.5) * ratio), 0, in - 1), ratio = in as f32 / out as f32 computed on the host
// (animated-GIF frames, reference handler.rs:336-341).
__device__ __forceinline__ uint32_t nearest_tap(uint32_t o, uint32_t ratio_bits, uint32_t in_size)
{
    const float c = ((float)o + 0.5f) * __uint_as_float(ratio_bits);
    const int left = (int)floorf(c);
    return (uint32_t)min(max(left, 0), (int)in_size - 1);
}

template <int CS, int PRE, bool LB, bool BORDER_ONLY, bool NEAREST = false>
__global__ __launch_bounds__(256) void place_kernel(const Job *__restrict__ jobs, uint32_t job_base)
{
    constexpr int MC = mid_channels(CS, PRE);
    const Job jb = jobs[job_base + blockIdx.y];
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;  // flat over the destination of this job
    if (idx >= jb.dh * jb.dw) return;
    const uint32_t y = idx / jb.dw, x = idx - y * jb.dw;
    const bool inside = x >= jb.ox && x < jb.ox + jb.cw && y >= jb.oy && y < jb.oy + jb.ch;
    if (!inside) {
        if (LB) reinterpret_cast<uint32_t *>(jb.dst)[y * jb.dw + x] = jb.fill;
        return;
    }
    if (BORDER_ONLY) return;
    uint32_t sy = y - jb.oy + jb.cy, sx = x - jb.ox + jb.cx;
    if (NEAREST) { sy = nearest_tap(sy, jb.vtab, jb.sh); sx = nearest_tap(sx, jb.htab, jb.sw); }
    const uint8_t *p = jb.src + ((size_t)sy * jb.sw + sx) * CS;
    uint32_t s[CS];
#pragma unroll
    for (int k = 0; k < CS; ++k) s[k] = p[k];
    float v[MC > CS ? MC : CS];
    preop_pixel<CS, PRE>(s, v);
    uint32_t c[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) c[k] = (uint32_t)v[k];
    store_pixel<MC, LB>(jb.dst, y * jb.dw + x, c, jb.fill);
}

// The same placement, four destination pixels of a row per thread: one thread per pixel with byte loads and stores ran a
// grayscale-only 1080p request at 0.2 of the HBM peak (tools/experiments/generic_sweep.py).  Where the four pixels lie inside the
// picture's window the source bytes come as CS dwords (gfx950 loads a dword from any byte address) and the result leaves as
// dwords; threads that straddle the window's edge fall back to the pixel-wise path above.  No arithmetic differs.
typedef uint32_t __attribute__((aligned(1))) u32_any; // a dword at any byte address
template <int CS, int PRE, bool LB>
__global__ __launch_bounds__(256) void place4_kernel(const Job *__restrict__ jobs, uint32_t job_base)
{
    constexpr int MC = mid_channels(CS, PRE);
    const Job jb = jobs[job_base + blockIdx.z];
    const uint32_t y = blockIdx.y, x0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (y >= jb.dh || x0 >= jb.dw) return;
    const bool row_in = y >= jb.oy && y < jb.oy + jb.ch;
    if (row_in && x0 >= jb.ox && x0 + 4u <= jb.ox + jb.cw) {
        const uint32_t sy = y - jb.oy + jb.cy, sx = x0 - jb.ox + jb.cx;
        const uint8_t *p = jb.src + ((size_t)sy * jb.sw + sx) * CS;
        uint32_t d[CS];
#pragma unroll
        for (int j = 0; j < CS; ++j) d[j] = *reinterpret_cast<const u32_any *>(p + 4 * j);
        uint32_t c[4][MC];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            uint32_t s4[CS];
#pragma unroll
            for (int k = 0; k < CS; ++k) { const int b = px * CS + k; s4[k] = (d[b >> 2] >> (8 * (b & 3))) & 255u; }
            float v[MC > CS ? MC : CS];
            preop_pixel<CS, PRE>(s4, v);
#pragma unroll
            for (int k = 0; k < MC; ++k) c[px][k] = (uint32_t)v[k];
        }
        if (LB) {
#pragma unroll
            for (int px = 0; px < 4; ++px) store_pixel<MC, true>(jb.dst, y * jb.dw + x0 + px, c[px], jb.fill);
        } else {
            uint8_t *o = jb.dst + ((size_t)y * jb.dw + x0) * MC;
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                uint32_t w = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int b = 4 * j + q; w |= c[b / MC][b % MC] << (8 * q); }
                *reinterpret_cast<u32_any *>(o + 4 * j) = w;
            }
        }
        return;
    }
    for (uint32_t x = x0; x < min(x0 + 4u, jb.dw); ++x) {
        const bool inside = row_in && x >= jb.ox && x < jb.ox + jb.cw;
        if (!inside) {
            if (LB) reinterpret_cast<uint32_t *>(jb.dst)[y * jb.dw + x] = jb.fill;
            continue;
        }
        const uint8_t *p = jb.src + ((size_t)(y - jb.oy + jb.cy) * jb.sw + (x - jb.ox + jb.cx)) * CS;
        uint32_t s1[CS];
#pragma unroll
        for (int k = 0; k < CS; ++k) s1[k] = p[k];
        float v[MC > CS ? MC : CS];
        preop_pixel<CS, PRE>(s1, v);
        uint32_t c1[MC];
#pragma unroll
        for (int k = 0; k < MC; ++k) c1[k] = (uint32_t)v[k];
        store_pixel<MC, LB>(jb.dst, y * jb.dw + x, c1, jb.fill);
    }
}

// ---------------------------------------------------------------------------
// EXIF orientation (DynamicImage::apply_orientation, image 0.25.6 metadata::Orientation + imageops::rotate90 /
// rotate180 / rotate270 / flip_horizontal / flip_vertical): a pure pixel permutation, one thread per
// destination pixel.  jb.sw x jb.sh = SOURCE size, jb.dw x jb.dh = oriented size, jb.fill = EXIF code 2..8.
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void orient_kernel(const Job *__restrict__ jobs, uint32_t job_base)
{
    const Job jb = jobs[job_base + blockIdx.z];
    const uint32_t y = blockIdx.y, x = blockIdx.x * 256u + threadIdx.x;
    if (y >= jb.dh || x >= jb.dw) return;
    const uint32_t W = jb.sw, H = jb.sh;
    uint32_t sx, sy;
    switch (jb.fill) {
    case 2: sx = W - 1u - x; sy = y; break;              // FlipHorizontal
    case 3: sx = W - 1u - x; sy = H - 1u - y; break;     // Rotate180
    case 4: sx = x; sy = H - 1u - y; break;              // FlipVertical
    case 5: sx = y; sy = x; break;                       // Rotate90FlipH (transpose)
    case 6: sx = y; sy = H - 1u - x; break;              // Rotate90
    case 7: sx = W - 1u - y; sy = H - 1u - x; break;     // Rotate270FlipH (transverse)
    case 8: sx = W - 1u - y; sy = x; break;              // Rotate270
    default: sx = x; sy = y; break;
    }
    const uint8_t *p = jb.src + ((size_t)sy * W + sx) * C;
    uint8_t *o = jb.dst + ((size_t)y * jb.dw + x) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = p[c];
}

// ---------------------------------------------------------------------------
// Fused streaming Lanczos3 down-scale: the hot kernel.
// ---------------------------------------------------------------------------

template <int CS> struct RowRaw;
template <> struct RowRaw<1> { uint32_t v; __device__ uint32_t dw(int) const { return v; } };
template <> struct RowRaw<2> { u32x2 v; __device__ uint32_t dw(int i) const { return i == 0 ? v.x : v.y; } };
template <> struct RowRaw<3> { u32x3 v; __device__ uint32_t dw(int i) const { return i == 0 ? v.x : i == 1 ? v.y : v.z; } };
template <> struct RowRaw<4> { u32x4 v; __device__ uint32_t dw(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; } };

// Source-row loads are ordinary (compiler-counted) raw buffer loads.  What keeps them in flight across
// the horizontal pass is that the kernel's only global STORES (store_pixel_hidden) are issued from
// inline asm: on gfx9 loads and stores share vmcnt, and once hipcc sees a store inside the row loop it
// drains every outstanding load at the loop header (s_waitcnt vmcnt(0)), turning the D-deep prefetch
// ring into one exposed HBM round trip per block.  Hidden stores are safe: vector-memory operations
// retire in order, so a younger store the compiler does not know about can only make its counted
// waits longer, never shorter.
template <int CS>
__device__ __forceinline__ void load_row(RowRaw<CS> &r, __amdgpu_buffer_rsrc_t rs, uint32_t voff)
{
    if constexpr (CS == 1) r.v = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, FL_LOAD_AUX);
    if constexpr (CS == 2) r.v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, FL_LOAD_AUX);
    if constexpr (CS == 3) r.v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, 0, FL_LOAD_AUX);
    if constexpr (CS == 4) r.v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, FL_LOAD_AUX);
}

// PXL pixels of CS bytes, packed in CS dwords -> PXL*MC floats with the pre-op applied.
template <int CS, int PRE>
__device__ __forceinline__ void convert_row(const RowRaw<CS> &raw, float *v)
{
    constexpr int MC = mid_channels(CS, PRE);
    uint32_t d[CS];
#pragma unroll
    for (int k = 0; k < CS; ++k) d[k] = raw.dw(k);
    if (PRE == PRE_INVERT) {
        // 255 - c on every colour byte; alpha bytes (LumaA / Rgba) keep their value.
        constexpr uint32_t m = (CS == 2) ? 0x00ff00ffu : 0x00ffffffu; // bytes that are colour, per pixel-aligned dword
#pragma unroll
        for (int k = 0; k < CS; ++k) {
            if (CS == 2 || CS == 4) d[k] = d[k] ^ m; else d[k] = ~d[k];
        }
    }
    if (PRE == PRE_GRAY && CS >= 3) {
#pragma unroll
        for (int p = 0; p < PXL; ++p) {
            uint32_t s[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const int b = p * CS + c;
                s[c] = (d[b >> 2] >> (8 * (b & 3))) & 255u;
            }
            v[p * MC] = (float)luma_u8(s[0], s[1], s[2]);
            if (CS == 4) v[p * MC + 1] = (float)s[3];
        }
    } else {
#pragma unroll
        for (int b = 0; b < PXL * CS; ++b) v[b] = (float)((d[b >> 2] >> (8 * (b & 3))) & 255u); // v_cvt_f32_ubyteN
    }
}


extern __shared__ __attribute__((aligned(16))) float fl_lds[];

// Workgroup barrier that orders LDS traffic only.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Takes the finished vertical row out of one accumulator slot (registers) and re-arms the slot.
template <int NH>
__device__ __forceinline__ void take_slot(f32x2 *acc, float *e)
{
#pragma unroll
    for (int j = 0; j < NH; ++j) { e[2 * j] = acc[j].x; e[2 * j + 1] = acc[j].y; acc[j] = f32x2{0.0f, 0.0f}; }
}

// NA = accumulator slots (output rows alive per source row; the host picks the smallest that fits),
// D  = source rows kept in flight per lane.
// UA (Rgb8 only): source rows that are not dword aligned (pitch or base pointer not a multiple of 4).  Dword
// buffer loads ignore the two low address bits, so each lane loads the 16 aligned bytes that cover its 12 and
// funnel-shifts them by the row's byte phase (v_alignbyte_b32; the phase is wave-uniform because a lane's
// own offset, 12 * lane, is a multiple of 4).
template <int CS, int PRE, bool LB, int NA, int D, bool UA>
__global__ __launch_bounds__(256) void resample_stream_kernel(const Job *__restrict__ jobs,
                                                              const StreamItem *__restrict__ items,
                                                              const uint32_t *__restrict__ arena
)
{
    // experiments only (-DFL_ABLATE=mask): 1 = no horizontal pass, 2 = no flush/barrier, 4 = no FMAs
#ifdef FL_ABLATE
    constexpr uint32_t ablate = FL_ABLATE;
#else
    constexpr uint32_t ablate = 0;
#endif
    constexpr int MC = mid_channels(CS, PRE);
    constexpr int NV = PXL * MC;
    constexpr uint32_t T = 256;
    float *lds = fl_lds;

    const StreamItem it = items[blockIdx.x];
    const Job jb = jobs[it.job];
    const uint32_t tid = threadIdx.x;
    const uint32_t nxs = it.x1 - it.x0;

    // LDS: [ sch: 2 x SCHED_CHUNK RowSched | WT: jmax x T float4 | PO: jmax x T u32 | pbuf: (nxs * ks + 1) x (float4, or float
    // for single-channel rows: the smaller buffer lets a third workgroup fit a CU) ]
    constexpr uint32_t SCH_WORDS = SCHED_CHUNK * (sizeof(RowSched) / 4);
    uint32_t *sch = reinterpret_cast<uint32_t *>(lds);
    const uint32_t jmax = it.jmax, kmax = it.kmax, ks = it.ks;
    f32x4 *wt = reinterpret_cast<f32x4 *>(lds + 2 * SCH_WORDS);
    uint32_t *po = reinterpret_cast<uint32_t *>(lds + 2 * SCH_WORDS + jmax * T * 4);
    const uint32_t pbuf_off = 2 * SCH_WORDS + jmax * T * 5; // float offset, a multiple of 4
    using PT = typename std::conditional<MC == 1, float, f32x4>::type; // one partial sum (all channels of one output column)
    PT *pbuf = reinterpret_cast<PT *>(lds + pbuf_off);

    // stage the strip's horizontal tables and zero the partial-sum buffer (slots no lane writes stay 0 forever)
    {
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(arena + it.wt_off);
        for (uint32_t i = tid; i < jmax * T; i += T) { wt[i] = wsrc[i]; po[i] = arena[it.po_off + i]; }
        for (uint32_t i = tid; i < nxs * ks + 1; i += T) pbuf[i] = PT{};
        for (uint32_t i = tid; i < SCH_WORDS; i += T) sch[i] = arena[it.sched_off + i]; // first schedule chunk
    }

    // raw buffer descriptor (stride 0): num_records = image bytes, so the hardware range-checks every lane
    const uint32_t base_phase = UA ? (uint32_t)(reinterpret_cast<uintptr_t>(jb.src) & 3u) : 0u;
    // (UA: the range is rounded up to whole dwords, otherwise the hardware zeroes the last, partly valid dword;
    //  an aligned dword that holds one valid byte lies in the same page as that byte, so this cannot fault)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(jb.src) - base_phase, 0,
                                                                        (int)(UA ? ((jb.src_bytes + base_phase + 3u) & ~3u) : jb.src_bytes), 0x00020000);
    const uint32_t pitch = jb.sw * CS;
    const uint32_t voff = (it.sx0 + tid * PXL) * CS + it.r0 * pitch + base_phase;
    const uint32_t pix_base = (jb.oy - jb.cy) * jb.dw + jb.ox + (it.x0 - jb.cx);

    // Letterbox border: every workgroup paints the part of the fill frame that lies next to its own band
    // and strip (first/last band own the rows above/below the picture, first/last strip the side margins),
    // so no separate fill kernel runs over the destination.
    if (LB) {
        const uint32_t dx0 = it.x0 == jb.cx ? 0u : jb.ox + it.x0 - jb.cx;
        const uint32_t dx1 = it.x1 == jb.cx + jb.cw ? jb.dw : jb.ox + it.x1 - jb.cx;
        const uint32_t dy0 = it.y0 == jb.cy ? 0u : jb.oy + it.y0 - jb.cy;
        const uint32_t dy1 = it.y1 == jb.cy + jb.ch ? jb.dh : jb.oy + it.y1 - jb.cy;
        uint32_t *d32 = reinterpret_cast<uint32_t *>(jb.dst);
        // (hidden stores, like every store of this kernel: see load_row)
        // rows above and below the placed picture
        const uint32_t wcols = dx1 - dx0;
        const uint32_t top_rows = dy0 < jb.oy ? min(dy1, jb.oy) - dy0 : 0u;
        for (uint32_t i = tid; i < top_rows * wcols; i += T) store_hidden_b32(d32 + (dy0 + i / wcols) * jb.dw + dx0 + i % wcols, jb.fill);
        const uint32_t by0 = max(dy0, jb.oy + jb.ch);
        const uint32_t bot_rows = dy1 > by0 ? dy1 - by0 : 0u;
        for (uint32_t i = tid; i < bot_rows * wcols; i += T) store_hidden_b32(d32 + (by0 + i / wcols) * jb.dw + dx0 + i % wcols, jb.fill);
        // side margins of the rows that hold the picture
        const uint32_t my0 = max(dy0, jb.oy), my1 = min(dy1, jb.oy + jb.ch);
        const uint32_t mrows = my1 > my0 ? my1 - my0 : 0u;
        const uint32_t lcols = dx0 < jb.ox ? min(dx1, jb.ox) - dx0 : 0u;
        for (uint32_t i = tid; i < mrows * lcols; i += T) store_hidden_b32(d32 + (my0 + i / lcols) * jb.dw + dx0 + i % lcols, jb.fill);
        const uint32_t rx0 = max(dx0, jb.ox + jb.cw);
        const uint32_t rcols = dx1 > rx0 ? dx1 - rx0 : 0u;
        for (uint32_t i = tid; i < mrows * rcols; i += T) store_hidden_b32(d32 + (my0 + i / rcols) * jb.dw + rx0 + i % rcols, jb.fill);
    }

    // accumulators live as register PAIRS so that every tap is one v_pk_fma_f32 (two fused multiply-adds)
    static_assert(NV % 2 == 0, "PXL is even");
    constexpr int NH = NV / 2;
    f32x2 acc[NA][NH];
#pragma unroll
    for (int s = 0; s < NA; ++s)
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[s][k] = f32x2{0.0f, 0.0f};

    // whole byte offset goes through voffset: rows past the image end are range-checked by the buffer descriptor and read 0
    constexpr int LW = UA ? 4 : CS; // dwords loaded per lane and row
    // rows in flight per lane: R.  HBM latency under load is longer than one block of FMAs, so the ring holds two blocks
    // where the register file allows it (Rgba8 accumulators and the unaligned variant's 4-dword rows already fill it)
    constexpr int RM = (!UA && CS <= 3) ? FL_RING_MULT : 1, R = D * RM;
    RowRaw<LW> ring[R];
#pragma unroll
    for (int k = 0; k < R; ++k) load_row<LW>(ring[k], rs, UA ? ((voff + k * pitch) & ~3u) : voff + k * pitch);

    __syncthreads();

    // Main loop: blocks of D source rows.  Inside a block the code is straight line (loads, byte->f32
    // conversion, one v_pk_fma_f32 per accumulator pair with the weight as an SGPR operand); rows that
    // complete an output row only record it, and the completed rows are flushed to LDS and run through
    // the horizontal pass between blocks.  The host guarantees that a completed slot is not re-armed
    // before the end of its block (build_row_sched, `defer` argument).
    // The row schedule (8 weights + live/emit masks per source row) is staged through LDS in chunks of
    // SCHED_CHUNK rows, double buffered: fetching it row by row with scalar loads exposes one scalar-cache
    // miss per row, which was the largest single stall of the loop.  The next chunk is fetched into two
    // VGPRs at the start of a chunk and written to the other LDS buffer at its end.
    const uint32_t nrows = it.r1 - it.r0; // the host pads the schedule to a whole number of chunks
    uint32_t g0 = 0, g1 = 0;
    // One block of D rows per pass of the inner loop; HALF selects which D registers of the ring the block consumes
    // and refills.  The inner loop is fully unrolled, so HALF is a constant and the ring stays in fixed registers.
    for (uint32_t rb0 = 0; rb0 < nrows; rb0 += R) { // nrows is a multiple of SCHED_CHUNK, hence of R
#pragma unroll
    for (int HALF = 0; HALF < RM; ++HALF) {
        const uint32_t rb = rb0 + HALF * D;
        const uint32_t in_chunk = rb % SCHED_CHUNK;
        const uint32_t *schc = sch + ((rb / SCHED_CHUNK) & 1u) * SCH_WORDS + in_chunk * (sizeof(RowSched) / 4);
        const bool fetch_next = in_chunk == 0 && rb + SCHED_CHUNK < nrows;
        if (fetch_next) {
            const uint32_t *nx = arena + it.sched_off + (size_t)(rb / SCHED_CHUNK + 1) * SCH_WORDS;
            g0 = nx[tid];
            if (tid + T < SCH_WORDS) g1 = nx[tid + T];
        }
        // block summary (row 0 of the block): which slots complete inside this block, first completed output row
        const u32x4 meta = *reinterpret_cast<const u32x4 *>(schc + 8);
        const uint32_t sch_base = (uint32_t)(schc - sch); // dword offset of this block's first row inside the dynamic LDS block
        // Row weights: wave-uniform LDS reads (broadcast), fetched ONE row ahead so that a row's FMAs never
        // wait for LDS.  Dead slots carry weight 0, so the accumulate below is branch free: acc + v * 0
        // leaves a finished or not yet started slot untouched, and skipping it with scalar branches costs
        // more than the idle v_pk_fma_f32 it saves.  (The opaque offsets stop hipcc from hoisting all D rows'
        // reads to the top of the block, which costs 8 VGPRs per row of look-ahead.)
        f32x4 wr03[D], wr47[D];
        {
            uint32_t so = sch_base;
            asm volatile("" : "+v"(so));
            wr03[0] = *reinterpret_cast<const f32x4 *>(fl_lds + so);
            wr47[0] = *reinterpret_cast<const f32x4 *>(fl_lds + so + 4);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const uint32_t ri = rb + k;
            if (k + 1 < D && FL_WPREFETCH) {
                uint32_t so = sch_base + (k + 1) * 12;
                asm volatile("" : "+v"(so));
                wr03[k + 1 < D ? k + 1 : 0] = *reinterpret_cast<const f32x4 *>(fl_lds + so);
                wr47[k + 1 < D ? k + 1 : 0] = *reinterpret_cast<const f32x4 *>(fl_lds + so + 4);
            }
            if (k > 0 && !FL_WPREFETCH) {
                uint32_t so = sch_base + k * 12;
                asm volatile("" : "+v"(so));
                wr03[k] = *reinterpret_cast<const f32x4 *>(fl_lds + so);
                wr47[k] = *reinterpret_cast<const f32x4 *>(fl_lds + so + 4);
            }
            const f32x4 w03 = wr03[k], w47 = wr47[k];
            // unconditional refill: rows past the band are harmless extra reads, rows past the image read 0
            // (convert first, refill second: the slot's registers are dead by then, so the refill lands in place)
            float v[NV];
            if constexpr (UA) {
                const uint32_t ph = __builtin_amdgcn_readfirstlane((voff + ri * pitch) & 3u); // same in every lane
                RowRaw<CS> al;
                al.v.x = __builtin_amdgcn_alignbyte(ring[HALF * D + k].v.y, ring[HALF * D + k].v.x, ph);
                al.v.y = __builtin_amdgcn_alignbyte(ring[HALF * D + k].v.z, ring[HALF * D + k].v.y, ph);
                al.v.z = __builtin_amdgcn_alignbyte(ring[HALF * D + k].v.w, ring[HALF * D + k].v.z, ph);
                convert_row<CS, PRE>(al, v);
            } else {
                convert_row<CS, PRE>(ring[HALF * D + k], v); // hipcc waits for this row only: the R - 1 younger rows stay in flight
            }
            // Keep the refill below the conversion: hoisted above it, the refill needs fresh registers and the
            // ring is then rotated with v_mov behind a vmcnt(0) at the loop end.  The empty asm makes the
            // refill's address depend on every converted value, so all reads of the old row precede it.
            uint32_t roff = voff + (ri + R) * pitch;
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(roff) : "v"(v[j]));
            load_row<LW>(ring[HALF * D + k], rs, UA ? (roff & ~3u) : roff);
            if (!(ablate & 4u)) {
#pragma unroll
                for (int s = 0; s < NA; ++s) {
                    const float w = s == 0 ? w03.x : s == 1 ? w03.y : s == 2 ? w03.z : s == 3 ? w03.w
                                  : s == 4 ? w47.x : s == 5 ? w47.y : s == 6 ? w47.z : w47.w;
                    const f32x2 ww = {w, w};
#pragma unroll
                    for (int j = 0; j < NH; ++j)
                        acc[s][j] = __builtin_elementwise_fma(f32x2{v[2 * j], v[2 * j + 1]}, ww, acc[s][j]);
                }
            }
            // One row at a time: left alone, hipcc converts all D rows and reads all D weight sets up front
            // and sinks every FMA to the end of the block, which costs 20 VGPRs per row of look-ahead (and a
            // wave per SIMD).  The empty asm pins each accumulator's value here, the barrier pins the rest.
#pragma unroll
            for (int s = 0; s < NA; ++s)
#pragma unroll
                for (int j = 0; j < NH; ++j) asm volatile("" : "+v"(acc[s][j])); // pinned as pairs: keeps v_pk_fma_f32
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t em = __builtin_amdgcn_readfirstlane(meta.y); // outputs complete in order: first_out, first_out + 1, ...
        uint32_t oy = __builtin_amdgcn_readfirstlane(meta.z);
        // g0/g1 were loaded by the compiler's own bookkeeping; publish the next chunk before its first use
        if (in_chunk + D == SCHED_CHUNK && rb + D < nrows) {
            uint32_t *nb = sch + (((rb / SCHED_CHUNK) + 1) & 1u) * SCH_WORDS;
            nb[tid] = g0;
            if (tid + T < SCH_WORDS) nb[tid + T] = g1;
            lds_barrier();
        }
        if (ablate & 2u) em = 0;
        while (em) { // wave-uniform; usually zero or one iteration
            const uint32_t s = oy % NA;
            em &= ~(1u << s);
            float e[NV]; // the finished f32 row: this lane's PXL pixels x MC channels
            switch (s) {
            case 0: take_slot<NH>(acc[0], e); break;
            case 1: take_slot<NH>(acc[NA > 1 ? 1 : 0], e); break;
            case 2: take_slot<NH>(acc[NA > 2 ? 2 : 0], e); break;
            case 3: take_slot<NH>(acc[NA > 3 ? 3 : 0], e); break;
            case 4: take_slot<NH>(acc[NA > 4 ? 4 : 0], e); break;
            case 5: take_slot<NH>(acc[NA > 5 ? 5 : 0], e); break;
            case 6: take_slot<NH>(acc[NA > 6 ? 6 : 0], e); break;
            default: take_slot<NH>(acc[NA > 7 ? 7 : 0], e); break;
            }
            if (!(ablate & 1u) && !(ablate & 16u)) {
                // Horizontal pass, step 1 (all lanes): this lane's 4 pixels -> one partial sum per output
                // column whose window they touch.  Weights and target slots come from per-lane tables, so
                // every LDS access is lane-contiguous (no bank conflicts).
#pragma unroll 4
                for (uint32_t j = 0; j < jmax; ++j) { // jmax is a multiple of 4 (host pads with zero weights -> dummy slot)
                    const f32x4 w = wt[j * T + tid];
                    const uint32_t slot = po[j * T + tid];
                    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int p = 0; p < PXL; ++p) {
                        const float wp = p == 0 ? w.x : p == 1 ? w.y : p == 2 ? w.z : w.w;
#pragma unroll
                        for (int c = 0; c < MC; ++c) part[c] = __builtin_fmaf(e[p * MC + c], wp, part[c]);
                    }
                    if constexpr (MC == 1) {
                        *reinterpret_cast<float *>(reinterpret_cast<char *>(pbuf) + (slot >> 2)) = part[0]; // table offsets are in 16-byte slots
                    } else {
                        f32x4 q;
                        q.x = part[0]; q.y = part[1]; q.z = part[2]; q.w = part[3];
                        *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(pbuf) + slot) = q;
                    }
                }
            }
            // LDS-only hand-off (no vmcnt drain: the prefetched rows and the pixel stores stay in flight)
            if (!(ablate & 8u)) lds_barrier();
            // step 2 (one lane per output column): add the partial sums in ascending pixel order.  The partial
            // sums are fetched into registers first and the second barrier sits right behind the fetch, so
            // the other waves go back to the vertical pass while the additions, rounding and store run.
            // partial sums held in registers across the barrier; Rgba8 rows (16 accumulators per slot) cannot afford 12 x 4
            constexpr uint32_t KREG = (MC == 4) ? 2 : 12;
            PT q[KREG];
            const PT *pp = pbuf + tid * ks;
            const bool reducer = !(ablate & 1u) && !(ablate & 32u) && tid < nxs;
            if (reducer) {
#pragma unroll
                for (uint32_t k = 0; k < KREG; ++k) q[k] = k < kmax ? pp[k] : PT{};
            }
            float sum[MC];
#pragma unroll
            for (int c = 0; c < MC; ++c) sum[c] = 0.0f;
            auto add_partial = [&](const PT &r) {
                if constexpr (MC == 1) sum[0] = sum[0] + r;
                else {
                    sum[0] = sum[0] + r.x;
                    if constexpr (MC > 1) sum[1] = sum[1] + r.y;
                    if constexpr (MC > 2) sum[2] = sum[2] + r.z;
                    if constexpr (MC > 3) sum[3] = sum[3] + r.w;
                }
            };
            if (reducer && kmax > KREG) { // long windows (ratio > ~7): finish the fetch before releasing the buffer
#pragma unroll
                for (uint32_t k = 0; k < KREG; ++k) add_partial(q[k]);
                for (uint32_t k = KREG; k < kmax; ++k) add_partial(pp[k]);
            }
            if (!(ablate & 8u)) lds_barrier();
            if (reducer) {
                if (kmax <= KREG) {
#pragma unroll
                    for (uint32_t k = 0; k < KREG; ++k) add_partial(q[k]); // slots past kmax hold +0: adding them changes nothing
                }
                uint32_t c8[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) c8[c] = round_u8(sum[c]);
                store_pixel<MC, LB, true>(jb.dst, pix_base + oy * jb.dw + tid, c8, jb.fill);
            }
            ++oy;
        }
    }
    }
}

// ---------------------------------------------------------------------------
// Separable Gaussian blur (image 0.25.6 imageops::blur = the same two-pass machinery with ratio 1 and
// support 2*sigma: 41..81 taps, windows truncated and renormalised at the borders).
//
// One workgroup = one image x a band of BLUR_TY output rows x a tile of <= T - (taps-1) output columns.
//   vertical pass   lane <-> source column (tile + halo).  Every source row of the band's window is loaded
//                   once (u8 -> f32 once) and accumulated into the BLUR_TY output rows in registers; the
//                   weights are wave-uniform (dense [row][BLUR_TY] table staged in LDS, broadcast reads).
//   hand-off        the BLUR_TY unrounded f32 rows go to LDS, lane-contiguous.
//   horizontal pass lane <-> output column.  Neighbouring lanes read neighbouring pixels (ratio 1), so the
//                   LDS reads are conflict free; one weight read feeds BLUR_TY rows x C channels of FMAs.
// ---------------------------------------------------------------------------

constexpr int BLUR_TY = 8;        // output rows per workgroup (colour)
constexpr int BLUR_TY_MONO = 8;   // ... when one channel is filtered (16 measured slower: 0.80 vs 0.61 ms, more zero-weight FMAs per band)
__host__ __device__ inline int blur_ty(uint32_t channels_filtered) { return channels_filtered == 1 ? BLUR_TY_MONO : BLUR_TY; }
constexpr uint32_t BLUR_MAXTAPS = 128;            // sigma <= 20 gives 81 taps
__host__ __device__ constexpr uint32_t blur_midw(uint32_t threads) { return threads + BLUR_MAXTAPS; } // columns of the f32 hand-off rows in LDS

__host__ __device__ inline uint32_t blur_tiles_t(uint32_t w, uint32_t taps, uint32_t threads) { const uint32_t cap = threads - (taps - 1); return (w + cap - 1) / cap; }
// Lanes per workgroup: a tile of tw output columns needs tw + taps - 1 lanes in the vertical pass, so the width that
// wastes the fewest lane slots wins (300 columns, 41 taps: 1 tile of 384 lanes instead of 2 of 256)
__host__ __device__ inline uint32_t blur_threads(uint32_t w, uint32_t taps)
{
    uint32_t best = 256, cost = 0xffffffffu;
    for (uint32_t t = 256; t <= 512; t += 128) {
        const uint32_t c = blur_tiles_t(w, taps, t) * t;
        if (c < cost) { cost = c; best = t; }
    }
    return best;
}
__host__ __device__ inline uint32_t blur_tiles(uint32_t w, uint32_t taps) { return blur_tiles_t(w, taps, blur_threads(w, taps)); }

// CS = channels stored per pixel, C = channels filtered.  C < CS only for opaque Rgba8 pictures (every
// letterboxed output of an opaque source): the alpha plane is the constant 255 (any normalised filter
// maps it to 255 again, far from a rounding boundary) and, for a grey picture on a grey fill, R = G = B, so
// one channel is filtered and replicated -- identical arithmetic on identical inputs, bit-identical output.
template <int CS, int C, int TY, int THREADS>
__global__ __launch_bounds__(THREADS) void blur_tile_kernel(const Job *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                        uint32_t job_base)
{
    constexpr int MS = C == 3 ? 4 : C; // floats per pixel in LDS
    constexpr uint32_t T = THREADS;
    // XCD-aware numbering: workgroups are handed to the 8 XCDs round-robin in launch order, and each XCD has its own
    // L2.  The bands of one picture re-read each other's halo rows (41-81 taps against 8 output rows), so all
    // workgroups of a picture are given launch indices that are congruent modulo 8: picture p of every group of 8
    // lives on XCD p % 8 and its halo re-reads hit that L2.
    uint32_t job_i, bid;
    {
        const uint32_t gx = gridDim.x, l = blockIdx.y * gx + blockIdx.x;
        const uint32_t set = l / (8u * gx), m = l - set * 8u * gx;
        const uint32_t inset = min(8u, gridDim.y - set * 8u);
        bid = m / inset;
        job_i = set * 8u + (m - bid * inset);
    }
    const Job jb = jobs[job_base + job_i];
    const uint32_t w = jb.sw, h = jb.sh;
    // one table block per (w, h, sigma): header -> this workgroup's tile and band records -> bulk copies
    const uint32_t *blk = arena + jb.pad0;
    const BlurPlanHeader hd = *reinterpret_cast<const BlurPlanHeader *>(blk);
    const uint32_t nt = hd.nt, htaps = hd.htaps, tw_full = hd.tw_full;
    if (bid >= nt * hd.nb) return;
    const uint32_t band = bid / nt, tile = bid % nt;
    const uint32_t x0 = tile * tw_full, tw = min(tw_full, w - x0);
    const uint32_t y0 = band * TY, ty = min((uint32_t)TY, h - y0);
    const uint32_t tid = threadIdx.x;
    const uint32_t cl = blk[hd.tiles_off + 2 * tile], ncols = blk[hd.tiles_off + 2 * tile + 1]; // ncols <= T by construction
    const uint32_t top = blk[hd.bands_off + 2 * band], nrows = blk[hd.bands_off + 2 * band + 1];

    // LDS: [ wv: rv x TY | mid: TY x (T + htaps) x MS | wh: htaps x nrows_h (distinct weight vectors) ]
    float *wv = fl_lds;
    const uint32_t wv_floats = (hd.rv * TY + 3u) & ~3u;
    constexpr uint32_t midw = blur_midw(THREADS); // compile-time row pitch: row offsets fold into the ds_read immediates
    float *mid = fl_lds + wv_floats;
    float *wh = mid + TY * midw * MS;

    {
        const float *vsrc = reinterpret_cast<const float *>(blk + hd.vdense_off) + (size_t)band * hd.rv * TY;
        for (uint32_t i = tid; i < hd.rv * TY; i += T) wv[i] = vsrc[i];
    }
    // columns past the tile's source window are only ever read with zero weights, but must hold finite values
    // (loops are written without integer division: it costs ~40 instructions per element on this ISA)
#pragma unroll
    for (int o = 0; o < TY; ++o)
        for (uint32_t cidx = ncols + tid; cidx < midw; cidx += T) {
#pragma unroll
            for (int c = 0; c < MS; ++c) mid[(o * midw + cidx) * MS + c] = 0.0f;
        }
    // horizontal weights: this column's first tap and the id of its weight vector; the distinct vectors (one for all
    // interior columns, one per column within 2 sigma of a border) are copied tap-major
    const uint32_t *tt = blk + hd.htiles_off + (size_t)tile * (tw_full * 2);
    const uint2 hcol = tid < tw ? *reinterpret_cast<const uint2 *>(tt + 2 * tid) : uint2{0u, 0u};
    const uint32_t hleft = hcol.x, hrow = hcol.y, nrh = hd.nrows_h;
    {
        const float *src = reinterpret_cast<const float *>(blk + hd.hrows_off);
        for (uint32_t i = tid; i < htaps * nrh; i += T) wh[i] = src[i];
    }
    __syncthreads();

    // ---- vertical pass ----
    float acc[TY][C];
#pragma unroll
    for (int o = 0; o < TY; ++o)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[o][c] = 0.0f;
    if (tid < ncols) {
        // raw buffer loads: a pointer read from a descriptor is "generic" to hipcc and would become flat_load,
        // which also counts on lgkmcnt and so serialises against every LDS weight read
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(jb.src), 0, (int)jb.src_bytes, 0x00020000);
        const uint32_t off0 = (top * w + cl + tid) * CS, pitch = w * CS;
        // the image is L2/MALL resident but a load is still ~1 us away: keep PF rows in flight per lane
        constexpr int PF = 12;
        uint32_t ring[PF][CS == 4 ? 1 : C];
        auto fetch = [&](uint32_t r, uint32_t *d) {
            const uint32_t o = off0 + r * pitch; // rows past the image end are range-checked and read 0 (never used)
            if constexpr (CS == 4) d[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
            else {
#pragma unroll
                for (int c = 0; c < C; ++c) d[c] = __builtin_amdgcn_raw_buffer_load_b8(rs, o + c, 0, 0);
            }
        };
#pragma unroll
        for (int k = 0; k < PF; ++k) fetch(k, ring[k]);
        for (uint32_t rb = 0; rb < nrows; rb += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const uint32_t r = rb + k;
                float v[C];
                if constexpr (CS == 4) {
                    const uint32_t d = ring[k][0];
#pragma unroll
                    for (int c = 0; c < C; ++c) v[c] = (float)((d >> (8 * c)) & 255u);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) v[c] = (float)ring[k][c];
                }
                fetch(r + PF, ring[k]);
                if (r < nrows) {
                    f32x4 wq[TY / 4];
#pragma unroll
                    for (int g = 0; g < TY / 4; ++g) wq[g] = *reinterpret_cast<const f32x4 *>(wv + r * TY + 4 * g);
#pragma unroll
                    for (int o = 0; o < TY; ++o) {
                        const f32x4 q4 = wq[o / 4];
                        const float wo = (o & 3) == 0 ? q4.x : (o & 3) == 1 ? q4.y : (o & 3) == 2 ? q4.z : q4.w;
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[o][c] = __builtin_fmaf(v[c], wo, acc[o][c]);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < TY; ++o) {
            float *m = mid + (o * midw + tid) * MS;
#pragma unroll
            for (int c = 0; c < C; ++c) m[c] = acc[o][c];
        }
    }
    __syncthreads();

    // ---- horizontal pass (tap order: one fused multiply-add per tap) ----
    if (tid < tw) {
#pragma unroll
        for (int o = 0; o < TY; ++o)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[o][c] = 0.0f;
        const float *m0 = mid + hleft * MS;
#pragma unroll 4
        for (uint32_t i = 0; i < htaps; ++i) {
            const float wi = wh[i * nrh + hrow];
#pragma unroll
            for (int o = 0; o < TY; ++o) {
                const float *px = m0 + (o * midw + i) * MS;
                if constexpr (MS == 4) {
                    const f32x4 q = *reinterpret_cast<const f32x4 *>(px);
                    acc[o][0] = __builtin_fmaf(q.x, wi, acc[o][0]);
                    if constexpr (C > 1) acc[o][1] = __builtin_fmaf(q.y, wi, acc[o][1]);
                    if constexpr (C > 2) acc[o][2] = __builtin_fmaf(q.z, wi, acc[o][2]);
                    if constexpr (C > 3) acc[o][3] = __builtin_fmaf(q.w, wi, acc[o][3]);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[o][c] = __builtin_fmaf(px[c], wi, acc[o][c]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < TY; ++o) {
            if ((uint32_t)o < ty) {
                uint32_t c8[CS];
#pragma unroll
                for (int c = 0; c < C; ++c) c8[c] = round_u8(acc[o][c]);
                if constexpr (CS == 4 && C == 1) { c8[1] = c8[0]; c8[2] = c8[0]; c8[3] = 255u; }
                if constexpr (CS == 4 && C == 3) c8[3] = 255u;
                store_pixel<CS, false>(jb.dst, (y0 + o) * w + x0 + tid, c8, 0u);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Encoder colour front ends
// ---------------------------------------------------------------------------

// DynamicImage get_pixel -> Rgba<u8>: Luma -> (l,l,l,255), LumaA -> (l,l,l,a), Rgb -> (r,g,b,255)
// image 0.25.6 codecs/jpeg/encoder.rs rgb_to_ycbcr (f32, truncating) on 8x8-padded planes
// (copy_blocks_ycbcr / pixel_at_or_near replicate the last column and row).
__global__ __launch_bounds__(256) void jfif444_kernel(const FrontendJob *__restrict__ fjobs, uint32_t job_base)
{
    const FrontendJob fj = fjobs[job_base + blockIdx.z];
    const uint32_t y = blockIdx.y;
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (y >= fj.plane_h || x >= fj.plane_w) return;
    const uint32_t sx = x < fj.w ? x : fj.w - 1u, sy = y < fj.h ? y : fj.h - 1u;
    uint32_t ri, gi, bi, ai;
    load_rgba(fj.src + ((size_t)sy * fj.w + sx) * fj.c, fj.c, ri, gi, bi, ai);
    const float max = 255.0f;
    const float r = (float)ri, g = (float)gi, b = (float)bi;
    const float yy = 76.245f / max * r + 149.685f / max * g + 29.07f / max * b;
    const float cb = -43.0185f / max * r - 84.4815f / max * g + 127.5f / max * b + 128.0f;
    const float cr = 127.5f / max * r - 106.7685f / max * g - 20.7315f / max * b + 128.0f;
    const size_t plane = (size_t)fj.plane_w * fj.plane_h, o = (size_t)y * fj.plane_w + x;
    fj.dst[o] = sat_u8(yy);
    fj.dst[plane + o] = sat_u8(cb);
    fj.dst[2 * plane + o] = sat_u8(cr);
}

// libwebp dsp/yuv.h fixed point (YUV_FIX = 16) + picture_csp_enc.c gamma-corrected 2x2 chroma averaging.
__device__ __forceinline__ int webp_clip_uv(int uv, int rounding)
{
    uv = (uv + rounding + (128 << 18)) >> 18;
    return ((uv & ~0xff) == 0) ? uv : (uv < 0) ? 0 : 255;
}
__device__ __forceinline__ int webp_linear_to_gamma(const int32_t *lin2gam, uint32_t base_value, int shift)
{
    const int v = (int)(base_value << shift);
    const int tab_pos = v >> 9;              // GAMMA_TAB_FIX + 2
    const int x = v & 511;                   // (kGammaTabScale << 2) - 1
    const int y = lin2gam[tab_pos + 1] * x + lin2gam[tab_pos] * (512 - x);
    return (y + 64) >> 7;                    // kGammaTabRounder, GAMMA_TAB_FIX
}

// One 2x2 block of libwebp's chroma down-sampling (picture_csp_enc.c AccumulateRGB / AccumulateRGBA): the four pixels
// (edge blocks repeat the last column / row: libwebp's SUM2 with shift 1, or step = 0 / rgb_stride = 0, are the same
// numbers) are averaged in linear light; a block that is neither fully opaque nor fully transparent weights them by
// alpha: LinearToGammaWeighted = LinearToGamma((sum a_i * GammaToLinear(c_i) * kInvAlpha[a]) >> 17), kInvAlpha[a] = 2^19 / a.
struct WebpBlock {
    uint32_t sr = 0, sg = 0, sb = 0;     // plain sums of GammaToLinear
    uint32_t wr = 0, wg = 0, wb = 0;     // alpha-weighted sums
    uint32_t a = 0;
    bool translucent = false;
    __device__ __forceinline__ void add(const int32_t *gam2lin, uint32_t r, uint32_t g, uint32_t b, uint32_t alpha)
    {
        const uint32_t lr = (uint32_t)gam2lin[r], lg = (uint32_t)gam2lin[g], lb = (uint32_t)gam2lin[b];
        sr += lr; sg += lg; sb += lb;
        wr += alpha * lr; wg += alpha * lg; wb += alpha * lb;
        a += alpha;
        translucent |= alpha != 255u;
    }
    __device__ __forceinline__ void finish(const int32_t *lin2gam, int &r, int &g, int &b) const
    {
        if (a == 4u * 255u || a == 0u) {
            r = webp_linear_to_gamma(lin2gam, sr, 0);
            g = webp_linear_to_gamma(lin2gam, sg, 0);
            b = webp_linear_to_gamma(lin2gam, sb, 0);
        } else {
            const uint32_t inv = (1u << 19) / a; // rare path: a real division is fine
            r = webp_linear_to_gamma(lin2gam, (wr * inv) >> 17, 0);
            g = webp_linear_to_gamma(lin2gam, (wg * inv) >> 17, 0);
            b = webp_linear_to_gamma(lin2gam, (wb * inv) >> 17, 0);
        }
    }
};

__global__ __launch_bounds__(256) void webp420_kernel(const FrontendJob *__restrict__ fjobs, const uint32_t *__restrict__ arena,
                                                      uint32_t gamma_off, uint32_t job_base)
{
    const FrontendJob fj = fjobs[job_base + blockIdx.z];
    const uint32_t by = blockIdx.y;
    const uint32_t bx = blockIdx.x * 256u + threadIdx.x;
    if (by >= fj.chroma_h || bx >= fj.chroma_w) return;
    const int32_t *gam2lin = reinterpret_cast<const int32_t *>(arena + gamma_off);       // [256]
    const int32_t *lin2gam = gam2lin + 256;                                               // [33]
    const uint32_t w = fj.w, h = fj.h, c = fj.c;
    const uint32_t x0 = 2u * bx, y0 = 2u * by;
    const uint32_t x1 = x0 + 1u < w ? x0 + 1u : x0;     // odd width: SUM2 path
    const uint32_t y1 = y0 + 1u < h ? y0 + 1u : y0;     // odd height: rgb_stride = 0
    uint8_t *Y = fj.dst, *U = fj.dst + (size_t)w * h, *V = U + (size_t)fj.chroma_w * fj.chroma_h, *A = V + (size_t)fj.chroma_w * fj.chroma_h;
    WebpBlock blk;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t px = (k & 1) ? x1 : x0, py = (k & 2) ? y1 : y0;
        uint32_t r, g, b, a;
        load_rgba(fj.src + ((size_t)py * w + px) * c, c, r, g, b, a);
        // luma and alpha of each real pixel (duplicates of the edge replicate are rewritten with the same value)
        const int luma = 16839 * (int)r + 33059 * (int)g + 6420 * (int)b;
        Y[(size_t)py * w + px] = (uint8_t)((luma + (1 << 15) + (16 << 16)) >> 16);
        A[(size_t)py * w + px] = (uint8_t)a;
        blk.add(gam2lin, r, g, b, a);
    }
    int r, g, b;
    blk.finish(lin2gam, r, g, b);
    U[(size_t)by * fj.chroma_w + bx] = (uint8_t)webp_clip_uv(-9719 * r - 19081 * g + 28800 * b, 1 << 17);
    V[(size_t)by * fj.chroma_w + bx] = (uint8_t)webp_clip_uv(+28800 * r - 24116 * g - 4684 * b, 1 << 17);
    if (blk.translucent && fj.status) atomicOr(fj.status, 1u);
}

// Rgba8 fast paths of the front ends (every letterboxed output is Rgba8): 4 pixels per thread, dword loads,
// dword stores per plane.  Same arithmetic as the generic kernels above.
__global__ __launch_bounds__(256) void jfif444_rgba_kernel(const FrontendJob *__restrict__ fjobs, uint32_t job_base)
{
    const FrontendJob fj = fjobs[job_base + blockIdx.y];
    const uint32_t q = fj.plane_w >> 2;                        // 4-pixel groups per plane row (plane_w is a multiple of 8)
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;      // flat over rows x groups: narrow planes still fill the waves
    if (idx >= q * fj.plane_h) return;
    const uint32_t y = idx / q, x = (idx - y * q) * 4u;
    const uint32_t sy = y < fj.h ? y : fj.h - 1u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fj.src), 0, (int)(fj.w * fj.h * 4u), 0x00020000);
    uint32_t yy = 0, cb = 0, cr = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sx = x + k < fj.w ? x + k : fj.w - 1u; // replicate the last column into the padding
        const uint32_t d = __builtin_amdgcn_raw_buffer_load_b32(rs, (sy * fj.w + sx) * 4u, 0, 0);
        uint32_t a, b, c;
        jfif_px(d, a, b, c);
        yy |= a << (8 * k); cb |= b << (8 * k); cr |= c << (8 * k);
    }
    const size_t plane = (size_t)fj.plane_w * fj.plane_h, o = (size_t)y * fj.plane_w + x;
    uint8_t *dst = fj.dst;
    *reinterpret_cast<uint32_t *>(dst + o) = yy;
    *reinterpret_cast<uint32_t *>(dst + plane + o) = cb;
    *reinterpret_cast<uint32_t *>(dst + 2 * plane + o) = cr;
}

__global__ __launch_bounds__(256) void webp420_rgba_kernel(const FrontendJob *__restrict__ fjobs, const uint32_t *__restrict__ arena,
                                                           uint32_t gamma_off, uint32_t job_base)
{
    const FrontendJob fj = fjobs[job_base + blockIdx.y];
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x; // flat over chroma samples
    if (idx >= fj.chroma_w * fj.chroma_h) return;
    const uint32_t by = idx / fj.chroma_w, bx = idx - by * fj.chroma_w;
    const int32_t *gam2lin = reinterpret_cast<const int32_t *>(arena + gamma_off);
    const int32_t *lin2gam = gam2lin + 256;
    const uint32_t w = fj.w, h = fj.h;
    const uint32_t x0 = 2u * bx, y0 = 2u * by;
    const uint32_t x1 = x0 + 1u < w ? x0 + 1u : x0;
    const uint32_t y1 = y0 + 1u < h ? y0 + 1u : y0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fj.src), 0, (int)(w * h * 4u), 0x00020000);
    uint8_t *Y = fj.dst, *U = fj.dst + (size_t)w * h, *V = U + (size_t)fj.chroma_w * fj.chroma_h, *A = V + (size_t)fj.chroma_w * fj.chroma_h;
    WebpBlock blk;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t px = (k & 1) ? x1 : x0, py = (k & 2) ? y1 : y0;
        const uint32_t d = __builtin_amdgcn_raw_buffer_load_b32(rs, (py * w + px) * 4u, 0, 0);
        const uint32_t r = d & 255u, g = (d >> 8) & 255u, b = (d >> 16) & 255u, a = d >> 24;
        const int luma = 16839 * (int)r + 33059 * (int)g + 6420 * (int)b;
        Y[(size_t)py * w + px] = (uint8_t)((luma + (1 << 15) + (16 << 16)) >> 16);
        A[(size_t)py * w + px] = (uint8_t)a;
        blk.add(gam2lin, r, g, b, a);
    }
    int r, g, b;
    blk.finish(lin2gam, r, g, b);
    U[(size_t)by * fj.chroma_w + bx] = (uint8_t)webp_clip_uv(-9719 * r - 19081 * g + 28800 * b, 1 << 17);
    V[(size_t)by * fj.chroma_w + bx] = (uint8_t)webp_clip_uv(+28800 * r - 24116 * g - 4684 * b, 1 << 17);
    if (blk.translucent && fj.status) atomicOr(fj.status, 1u);
}

// reference src/handler.rs:423-438: per pixel (Y, Cb, Cr, K) -> (clamp(R), clamp(G), clamp(B), 255 - K), f32 with
// truncating casts, evaluated in the reference's operation order (this file is built with -ffp-contract=off)
__device__ __forceinline__ uint32_t ycck_pixel(uint32_t d)
{
    const float y = (float)(d & 255u), cb = (float)((d >> 8) & 255u), cr = (float)((d >> 16) & 255u);
    float r = y + 1.40200f * cr - 179.456f;
    float g = y - 0.34414f * cb - 0.71414f * cr + 135.45984f;
    float b = y + 1.77200f * cb - 226.816f;
    r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
    g = g < 0.0f ? 0.0f : (g > 255.0f ? 255.0f : g);
    b = b < 0.0f ? 0.0f : (b > 255.0f ? 255.0f : b);
    return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | ((255u - (d >> 24)) << 24);
}

__global__ __launch_bounds__(256) void ycck_to_cmyk_kernel(uint32_t *__restrict__ px, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    px[i] = ycck_pixel(px[i]);
}

// ---------------------------------------------------------------------------
// CMYK_8 -> RGB_8 through a baked device-link CLUT: what lcms2's transform_pixels does for the transform the
// reference builds (src/handler.rs:469-493: CMYK profile -> sRGB, Perceptual, NO_CACHE).  Little CMS 2 optimises
// that transform into ONE 17^4 x 3 u16 table (cmsopt.c OptimizeByResampling) and evaluates every pixel with
// cmsintrp.c Eval4Inputs: tetrahedral interpolation over inputs 1..3 on the two table slices that bracket
// input 0, then a linear blend, all in 16.16 fixed point with 32-bit wrap-around.  This kernel is that
// arithmetic (formatters Unroll4Bytes / Pack3Bytes included); the table is baked on the host
// (fl_cmyk.cpp).  Table nodes are padded to 4 x u16 so that a node is one 8-byte load.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int32_t lcms_to_fixed_domain(int32_t a)
{
    return (int32_t)((uint32_t)a + (uint32_t)((int32_t)((uint32_t)a + 0x7fffu) / 0xffff));
}

struct ClutNode { int32_t c[3]; };
__device__ __forceinline__ ClutNode clut_load(const uint2 *__restrict__ t, uint32_t idx)
{
    const uint2 v = t[idx];
    return ClutNode{{(int32_t)(v.x & 0xffffu), (int32_t)(v.x >> 16), (int32_t)(v.y & 0xffffu)}};
}

// tetrahedral interpolation inside slice `base`: the six cases of Eval4Inputs collapse to "walk the cube
// from (0,0,0) to (1,1,1) along the axes in descending order of their fractions" (ties give equal sums)
__device__ __forceinline__ void clut_tetra(const uint2 *__restrict__ t, uint32_t base, uint32_t X0, uint32_t X1, uint32_t Y0,
                                           uint32_t Y1, uint32_t Z0, uint32_t Z1, int32_t rx, int32_t ry, int32_t rz, uint32_t out[3])
{
    uint32_t a1, a2; // the two intermediate vertices
    int32_t r1, r2, r3;
    if (rx >= ry && ry >= rz)      { a1 = X1 + Y0 + Z0; a2 = X1 + Y1 + Z0; r1 = rx; r2 = ry; r3 = rz; }
    else if (rx >= rz && rz >= ry) { a1 = X1 + Y0 + Z0; a2 = X1 + Y0 + Z1; r1 = rx; r2 = rz; r3 = ry; }
    else if (rz >= rx && rx >= ry) { a1 = X0 + Y0 + Z1; a2 = X1 + Y0 + Z1; r1 = rz; r2 = rx; r3 = ry; }
    else if (ry >= rx && rx >= rz) { a1 = X0 + Y1 + Z0; a2 = X1 + Y1 + Z0; r1 = ry; r2 = rx; r3 = rz; }
    else if (ry >= rz && rz >= rx) { a1 = X0 + Y1 + Z0; a2 = X0 + Y1 + Z1; r1 = ry; r2 = rz; r3 = rx; }
    else                           { a1 = X0 + Y0 + Z1; a2 = X0 + Y1 + Z1; r1 = rz; r2 = ry; r3 = rx; }
    const ClutNode v0 = clut_load(t, base + X0 + Y0 + Z0), v1 = clut_load(t, base + a1), v2 = clut_load(t, base + a2),
                   v3 = clut_load(t, base + X1 + Y1 + Z1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t rest = (uint32_t)(v1.c[k] - v0.c[k]) * (uint32_t)r1 + (uint32_t)(v2.c[k] - v1.c[k]) * (uint32_t)r2 +
                              (uint32_t)(v3.c[k] - v2.c[k]) * (uint32_t)r3;
        const int32_t f = lcms_to_fixed_domain((int32_t)rest);
        out[k] = (uint32_t)(v0.c[k] + (((int32_t)((uint32_t)f + 0x8000u)) >> 16)) & 0xffffu;
    }
}

__device__ __forceinline__ uint32_t cmyk_clut_pixel(const uint2 *__restrict__ t, uint32_t grid, uint32_t d)
{
    const uint32_t domain = grid - 1u;
    const uint32_t sz = 1u, sy = grid, sx = grid * grid, sk = grid * grid * grid; // node strides of inputs 3, 2, 1, 0
    uint32_t i0[4], step[4];
    int32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (d >> (8 * k)) & 255u;
        const uint32_t in16 = b * 257u;                                  // FROM_8_TO_16
        const int32_t f = lcms_to_fixed_domain((int32_t)(in16 * domain));
        i0[k] = (uint32_t)f >> 16;
        r[k] = f & 0xffff;
        step[k] = b == 255u ? 0u : 1u;                                   // Input == 0xFFFF: no upper neighbour
    }
    const uint32_t K0 = sk * i0[0], K1 = K0 + sk * step[0];
    const uint32_t X0 = sx * i0[1], X1 = X0 + sx * step[1];
    const uint32_t Y0 = sy * i0[2], Y1 = Y0 + sy * step[2];
    const uint32_t Z0 = sz * i0[3], Z1 = Z0 + sz * step[3];
    uint32_t t1[3], t2[3];
    clut_tetra(t, K0, X0, X1, Y0, Y1, Z0, Z1, r[1], r[2], r[3], t1);
    clut_tetra(t, K1, X0, X1, Y0, Y1, Z0, Z1, r[1], r[2], r[3], t2);
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t dif = (t2[k] - t1[k]) * (uint32_t)r[0] + 0x8000u;      // LinearInterp
        dif = ((dif >> 16) + t1[k]) & 0xffffu;
        o |= ((dif * 65281u + 8388608u) >> 24) << (8 * k);              // FROM_16_TO_8
    }
    return o;
}

// 4 pixels per thread: one 16-byte load, three dword stores.  n4 = ceil(n / 4); the buffers are padded to that.
template <bool YCCK>
__global__ __launch_bounds__(256) void cmyk_clut_kernel(const uint4 *__restrict__ src, uint32_t *__restrict__ dst,
                                                        const uint2 *__restrict__ clut, uint32_t grid, uint64_t n4)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const uint4 q = src[i];
    uint32_t p[4] = {q.x, q.y, q.z, q.w}, o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (YCCK) p[k] = ycck_pixel(p[k]);
        o[k] = cmyk_clut_pixel(clut, grid, p[k]);
    }
    dst[i * 3 + 0] = o[0] | (o[1] << 24);
    dst[i * 3 + 1] = (o[1] >> 8) | (o[2] << 16);
    dst[i * 3 + 2] = (o[2] >> 16) | (o[3] << 8);
}


// ---------------------------------------------------------------------------
// launch wrappers (called from the host runtime; all asynchronous on `stream`)
// ---------------------------------------------------------------------------

#define FL_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return e__; } while (0)

template <int CS, int PRE>
static hipError_t launch_vpass_t(const LaunchGeneric &g, hipStream_t st)
{
    dim3 grid((g.max_sw * g.max_rh + 255u) / 256u, g.njobs);
    hipLaunchKernelGGL((vpass_generic_kernel<CS, PRE>), grid, dim3(256), 0, st, g.jobs, g.arena, g.mid, g.job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_vpass_generic(const LaunchGeneric &g, hipStream_t st)
{
#define FL_CASE(C_, P_) if (g.cs == C_ && g.pre == P_) return launch_vpass_t<C_, P_>(g, st)
    FL_CASE(1, PRE_NONE); FL_CASE(1, PRE_GRAY); FL_CASE(1, PRE_INVERT);
    FL_CASE(2, PRE_NONE); FL_CASE(2, PRE_GRAY); FL_CASE(2, PRE_INVERT);
    FL_CASE(3, PRE_NONE); FL_CASE(3, PRE_GRAY); FL_CASE(3, PRE_INVERT);
    FL_CASE(4, PRE_NONE); FL_CASE(4, PRE_GRAY); FL_CASE(4, PRE_INVERT);
#undef FL_CASE
    return hipErrorInvalidValue;
}

template <int MC, bool LB>
static hipError_t launch_hpass_t(const LaunchGeneric &g, hipStream_t st)
{
    dim3 grid((g.max_cw * g.max_ch + 255u) / 256u, g.njobs);
    if (g.grouped) hipLaunchKernelGGL((hpass_generic_kernel<MC, LB, true>), grid, dim3(256), 0, st, g.jobs, g.arena, g.mid, g.job_base);
    else hipLaunchKernelGGL((hpass_generic_kernel<MC, LB, false>), grid, dim3(256), 0, st, g.jobs, g.arena, g.mid, g.job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

template <int CS, int PRE>
static hipError_t launch_tile_t(const LaunchGeneric &g, hipStream_t st)
{
    // grid.x: (tile columns of the widest kept window at the narrowest tile width) x bands of rows -- enough bands that the chip sees
    // a few thousand workgroups also when the batch is one picture
    const uint32_t cols = (g.max_cw + g.tile_w_min - 1u) / g.tile_w_min;
    const uint32_t max_bands = (g.max_ch + kTileRows - 1u) / kTileRows;
    const uint32_t nbands = std::max(1u, std::min(max_bands, (4096u + cols * g.njobs - 1u) / (cols * g.njobs)));
    dim3 grid(cols * nbands, g.njobs);
    const size_t lds = (size_t)(kTileLdsFloats + kTileWeightFloats + kTileVRows * kTileRows) * sizeof(float);
#define FL_TILE_LAUNCH(LB_, GR_)                                                                                                          \
    do {                                                                                                                                  \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_tile_kernel<CS, PRE, LB_, GR_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                                    \
        hipLaunchKernelGGL((resample_tile_kernel<CS, PRE, LB_, GR_>), grid, dim3(256), lds, st, g.jobs, g.arena, g.job_base, nbands);     \
    } while (0)
    if (g.letterbox) { if (g.grouped) FL_TILE_LAUNCH(true, true); else FL_TILE_LAUNCH(true, false); }
    else { if (g.grouped) FL_TILE_LAUNCH(false, true); else FL_TILE_LAUNCH(false, false); }
#undef FL_TILE_LAUNCH
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_tile_resample(const LaunchGeneric &g, hipStream_t st)
{
#define FL_CASE(C_, P_) if (g.cs == C_ && g.pre == P_) return launch_tile_t<C_, P_>(g, st)
    FL_CASE(1, PRE_NONE); FL_CASE(1, PRE_GRAY); FL_CASE(1, PRE_INVERT);
    FL_CASE(2, PRE_NONE); FL_CASE(2, PRE_GRAY); FL_CASE(2, PRE_INVERT);
    FL_CASE(3, PRE_NONE); FL_CASE(3, PRE_GRAY); FL_CASE(3, PRE_INVERT);
    FL_CASE(4, PRE_NONE); FL_CASE(4, PRE_GRAY); FL_CASE(4, PRE_INVERT);
#undef FL_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_hpass_generic(const LaunchGeneric &g, hipStream_t st)
{
    const uint32_t mc = mid_channels(g.cs, g.pre);
#define FL_CASE(M_) if (mc == M_) return g.letterbox ? launch_hpass_t<M_, true>(g, st) : launch_hpass_t<M_, false>(g, st)
    FL_CASE(1); FL_CASE(2); FL_CASE(3); FL_CASE(4);
#undef FL_CASE
    return hipErrorInvalidValue;
}

template <int CS, int PRE>
static hipError_t launch_place_t(const LaunchGeneric &g, bool border_only, hipStream_t st)
{
    dim3 grid((g.max_dw * g.max_dh + 255u) / 256u, g.njobs);
    if (g.nearest) {
        if (g.letterbox) hipLaunchKernelGGL((place_kernel<CS, PRE, true, false, true>), grid, dim3(256), 0, st, g.jobs, g.job_base);
        else hipLaunchKernelGGL((place_kernel<CS, PRE, false, false, true>), grid, dim3(256), 0, st, g.jobs, g.job_base);
    } else if (border_only) {
        if (!g.letterbox) return hipSuccess;
        hipLaunchKernelGGL((place_kernel<CS, PRE, true, true>), grid, dim3(256), 0, st, g.jobs, g.job_base);
    } else {
        const bool by_four = g.max_dh <= 65535u && g.njobs <= 65535u && !g.no_place4; // (grid limits of the y and z dimensions)
        dim3 grid4(((g.max_dw + 3u) / 4u + 255u) / 256u, g.max_dh, g.njobs);
        if (g.letterbox) {
            if (by_four) hipLaunchKernelGGL((place4_kernel<CS, PRE, true>), grid4, dim3(256), 0, st, g.jobs, g.job_base);
            else hipLaunchKernelGGL((place_kernel<CS, PRE, true, false>), grid, dim3(256), 0, st, g.jobs, g.job_base);
        } else {
            if (by_four) hipLaunchKernelGGL((place4_kernel<CS, PRE, false>), grid4, dim3(256), 0, st, g.jobs, g.job_base);
            else hipLaunchKernelGGL((place_kernel<CS, PRE, false, false>), grid, dim3(256), 0, st, g.jobs, g.job_base);
        }
    }
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_place(const LaunchGeneric &g, bool border_only, hipStream_t st)
{
#define FL_CASE(C_, P_) if (g.cs == C_ && g.pre == P_) return launch_place_t<C_, P_>(g, border_only, st)
    FL_CASE(1, PRE_NONE); FL_CASE(1, PRE_GRAY); FL_CASE(1, PRE_INVERT);
    FL_CASE(2, PRE_NONE); FL_CASE(2, PRE_GRAY); FL_CASE(2, PRE_INVERT);
    FL_CASE(3, PRE_NONE); FL_CASE(3, PRE_GRAY); FL_CASE(3, PRE_INVERT);
    FL_CASE(4, PRE_NONE); FL_CASE(4, PRE_GRAY); FL_CASE(4, PRE_INVERT);
#undef FL_CASE
    return hipErrorInvalidValue;
}

size_t blur_lds_bytes(uint32_t w, uint32_t channels, uint32_t vtaps, uint32_t htaps)
{
    const size_t BLUR_TY = (size_t)blur_ty(channels);
    const uint32_t ms = channels == 3 ? 4 : channels;
    const uint32_t nt = blur_tiles(w, htaps), tw = (w + nt - 1) / nt;
    const size_t wv = (((size_t)(BLUR_TY + vtaps) * BLUR_TY + 3) & ~(size_t)3); // rv <= BLUR_TY + vtaps - 1
    (void)tw;
    const size_t rows_h = std::min<size_t>(w, 2 * (size_t)htaps); // distinct horizontal weight vectors: <= htaps (borders) + 1 (interior)
    return (wv + (size_t)BLUR_TY * blur_midw(blur_threads(w, htaps)) * ms + (size_t)htaps * rows_h) * sizeof(float);
}

uint32_t blur_tile_count(uint32_t w, uint32_t htaps) { return blur_tiles(w, htaps); }
uint32_t blur_lanes(uint32_t w, uint32_t htaps) { return blur_threads(w, htaps); }
uint32_t blur_band_rows(uint32_t channels_filtered) { return (uint32_t)blur_ty(channels_filtered); }

uint32_t blur_grid_x(uint32_t w, uint32_t h, uint32_t htaps, uint32_t channels_filtered)
{
    const uint32_t ty = (uint32_t)blur_ty(channels_filtered);
    return blur_tiles(w, htaps) * ((h + ty - 1) / ty);
}

bool blur_tile_supported(uint32_t htaps) { return htaps >= 1 && htaps <= BLUR_MAXTAPS; }

template <int CS, int C, int THREADS>
static hipError_t launch_blur_tt(const LaunchGeneric &g, uint32_t grid_x, size_t lds, hipStream_t st)
{
    auto k = blur_tile_kernel<CS, C, (C == 1 ? BLUR_TY_MONO : BLUR_TY), THREADS>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(grid_x, g.njobs), dim3(THREADS), lds, st, g.jobs, g.arena, g.job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

// g.blur_lanes = lanes per workgroup chosen for the group (blur_threads of its pictures)
template <int CS, int C>
static hipError_t launch_blur_t(const LaunchGeneric &g, uint32_t grid_x, size_t lds, hipStream_t st)
{
    switch (g.blur_lanes) {
    case 256: return launch_blur_tt<CS, C, 256>(g, grid_x, lds, st);
    case 384: return launch_blur_tt<CS, C, 384>(g, grid_x, lds, st);
    case 512: return launch_blur_tt<CS, C, 512>(g, grid_x, lds, st);
    }
    return hipErrorInvalidValue;
}

// g.cs = channels stored, g.pre = channels filtered (see blur_tile_kernel)
hipError_t launch_blur_tile(const LaunchGeneric &g, uint32_t grid_x, size_t lds, hipStream_t st)
{
    if (g.cs == 4 && g.pre == 1) return launch_blur_t<4, 1>(g, grid_x, lds, st);
    if (g.cs == 4 && g.pre == 3) return launch_blur_t<4, 3>(g, grid_x, lds, st);
    switch (g.cs) {
    case 1: return launch_blur_t<1, 1>(g, grid_x, lds, st);
    case 2: return launch_blur_t<2, 2>(g, grid_x, lds, st);
    case 3: return launch_blur_t<3, 3>(g, grid_x, lds, st);
    case 4: return launch_blur_t<4, 4>(g, grid_x, lds, st);
    }
    return hipErrorInvalidValue;
}

size_t stream_lds_bytes(uint32_t jmax, uint32_t nxs, uint32_t ks, uint32_t mid_channels)
{
    const size_t pbuf = (((size_t)nxs * ks + 1) * (mid_channels == 1 ? 4 : 16) + 15) & ~(size_t)15;
    return 2 * SCHED_CHUNK * sizeof(RowSched) + (size_t)jmax * 256 * 20 + pbuf;
}

uint32_t stream_lanes() { return 256; }

uint32_t stream_block_rows() { return FL_STREAM_DEPTH; }
static_assert(SCHED_CHUNK % (FL_STREAM_DEPTH * FL_RING_MULT) == 0 && FL_RING_MULT >= 1 && FL_RING_MULT <= 2, "schedule chunks must hold whole ring rounds");

bool stream_supported(uint32_t cs, uint32_t pre)
{
    (void)pre;
    return cs >= 1 && cs <= 4;
}

template <int CS, int PRE, bool LB, int NA, int D, bool UA>
static hipError_t launch_stream_v(const LaunchStream &s, hipStream_t st)
{
    auto k = resample_stream_kernel<CS, PRE, LB, NA, D, UA>;
    if (s.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(s.nitems), dim3(256), s.lds_bytes, st, s.jobs, s.items, s.arena);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

template <int CS, int PRE>
static hipError_t launch_stream_t(const LaunchStream &s, hipStream_t st)
{
    constexpr int D = FL_STREAM_DEPTH;
    if constexpr (CS == 3) {
        if (s.unaligned) {
            if (s.letterbox) return s.nacc <= 7 ? launch_stream_v<CS, PRE, true, 7, D, true>(s, st) : launch_stream_v<CS, PRE, true, 8, D, true>(s, st);
            return s.nacc <= 7 ? launch_stream_v<CS, PRE, false, 7, D, true>(s, st) : launch_stream_v<CS, PRE, false, 8, D, true>(s, st);
        }
    }
    if (s.letterbox) return s.nacc <= 7 ? launch_stream_v<CS, PRE, true, 7, D, false>(s, st) : launch_stream_v<CS, PRE, true, 8, D, false>(s, st);
    return s.nacc <= 7 ? launch_stream_v<CS, PRE, false, 7, D, false>(s, st) : launch_stream_v<CS, PRE, false, 8, D, false>(s, st);
}

hipError_t launch_stream(const LaunchStream &s, hipStream_t st)
{
#define FL_CASE(C_, P_) if (s.cs == C_ && s.pre == P_) return launch_stream_t<C_, P_>(s, st)
    FL_CASE(1, PRE_NONE); FL_CASE(1, PRE_INVERT);   // (grayscale of Luma / LumaA is the identity: the host maps it to PRE_NONE)
    FL_CASE(2, PRE_NONE); FL_CASE(2, PRE_INVERT);
    FL_CASE(3, PRE_NONE); FL_CASE(3, PRE_GRAY); FL_CASE(3, PRE_INVERT);
    FL_CASE(4, PRE_NONE); FL_CASE(4, PRE_GRAY); FL_CASE(4, PRE_INVERT);
#undef FL_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_orient(const LaunchGeneric &g, hipStream_t st)
{
    dim3 grid((g.max_dw + 255u) / 256u, g.max_dh, g.njobs);
    switch (g.cs) {
    case 1: hipLaunchKernelGGL(orient_kernel<1>, grid, dim3(256), 0, st, g.jobs, g.job_base); break;
    case 2: hipLaunchKernelGGL(orient_kernel<2>, grid, dim3(256), 0, st, g.jobs, g.job_base); break;
    case 3: hipLaunchKernelGGL(orient_kernel<3>, grid, dim3(256), 0, st, g.jobs, g.job_base); break;
    case 4: hipLaunchKernelGGL(orient_kernel<4>, grid, dim3(256), 0, st, g.jobs, g.job_base); break;
    default: return hipErrorInvalidValue;
    }
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_cmyk_clut(const void *src, void *dst, const void *clut, uint32_t grid, uint64_t n_pixels, bool ycck, hipStream_t st)
{
    const uint64_t n4 = (n_pixels + 3) / 4;
    if (n4 == 0) return hipSuccess;
    const dim3 g((unsigned)((n4 + 255) / 256));
    if (ycck) hipLaunchKernelGGL(cmyk_clut_kernel<true>, g, dim3(256), 0, st, static_cast<const uint4 *>(src), static_cast<uint32_t *>(dst),
                                 static_cast<const uint2 *>(clut), grid, n4);
    else hipLaunchKernelGGL(cmyk_clut_kernel<false>, g, dim3(256), 0, st, static_cast<const uint4 *>(src), static_cast<uint32_t *>(dst),
                            static_cast<const uint2 *>(clut), grid, n4);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_ycck_to_cmyk(uint32_t *px, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ycck_to_cmyk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, px, n);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_jfif444(const FrontendJob *fjobs, uint32_t job_base, uint32_t njobs, uint32_t max_pw, uint32_t max_ph,
                          bool all_rgba_aligned, hipStream_t st)
{
    if (all_rgba_aligned) {
        dim3 grid4(((max_pw / 4u) * max_ph + 255u) / 256u, njobs);
        hipLaunchKernelGGL(jfif444_rgba_kernel, grid4, dim3(256), 0, st, fjobs, job_base);
        FL_LAUNCH_CHECK();
        return hipSuccess;
    }
    dim3 grid((max_pw + 255u) / 256u, max_ph, njobs);
    hipLaunchKernelGGL(jfif444_kernel, grid, dim3(256), 0, st, fjobs, job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_webp420(const FrontendJob *fjobs, const uint32_t *arena, uint32_t gamma_off, uint32_t job_base,
                          uint32_t njobs, uint32_t max_cw, uint32_t max_ch, bool all_rgba_aligned, hipStream_t st)
{
    if (all_rgba_aligned) {
        dim3 gridf((max_cw * max_ch + 255u) / 256u, njobs);
        hipLaunchKernelGGL(webp420_rgba_kernel, gridf, dim3(256), 0, st, fjobs, arena, gamma_off, job_base);
        FL_LAUNCH_CHECK();
        return hipSuccess;
    }
    dim3 grid((max_cw + 255u) / 256u, max_ch, njobs);
    hipLaunchKernelGGL(webp420_kernel, grid, dim3(256), 0, st, fjobs, arena, gamma_off, job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace fl
