// fl_jpeg.hip -- the JPEG encoder's back half on gfx950: colour conversion + forward DCT + quantisation, then
// Huffman coding, byte stuffing and framing, so that what leaves the GPU is the finished JFIF stream.
//
// Replaces `JpegEncoder::new_with_quality(&mut buffer, q).encode_image(&img)` (reference src/handler.rs:274-278),
// i.e. image 0.25.6 src/codecs/jpeg/encoder.rs (encode_image, encode_rgb, BitWriter::write_block / write_bits /
// huffman_encode, build_* header helpers) and src/codecs/jpeg/transform.rs (fdct): baseline, three components,
// all sampling factors 1x1, Annex K quantisation tables scaled by quality, Annex K Huffman tables.
//
// Four kernels per batch:
//   jpeg_dct_quant_kernel   one wave per 8x8 block: the wave's 64 lanes are the 64 samples / coefficients.  The
//                           integer DCT of transform.rs (IJG jfdctint) is linear up to its final shift of each pass,
//                           so a lane computes ITS coefficient as an 8-term integer dot product with a constant
//                           matrix derived at compile time from the butterfly itself (int32 wrap-around arithmetic
//                           is a ring: the sums are identical bit for bit).
//                           The same wave also sizes the block's AC code (lane k = zig-zag coefficient k; zero runs
//                           come from a ballot).
//   jpeg_scan_kernel        one workgroup per picture: adds the DC code sizes (they need the previous block) and scans
//                           the block sizes into bit offsets; clears the bit-stream scratch; writes the header.
//   jpeg_emit_kernel        one wave per block again: every lane forms its code word(s), a wave prefix sum places them,
//                           LDS atomics assemble the block, word stores (atomic at the two shared edges) write it.
//   jpeg_stuff_kernel       one workgroup per picture: pad_byte, 0xFF -> 0xFF00 stuffing by a second scan, EOI, length.
// HBM traffic is the pixels once (the input is the 240 KB picture the resample kernel just wrote, L2 resident) plus
// the coefficient scratch; the kernels are VALU / LDS bound, not bandwidth bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fl_jpeg_tables.h"
#include "fl_kernels.h"
#include "fl_pixel.h"

namespace fl {

namespace {

// ---------------------------------------------------------------- constant tables --

// transform.rs constants (CONST_BITS = 13)
constexpr int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
                  F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

struct Mat8 { int32_t a[8][8]; };

// The 1-D butterfly of transform.rs::fdct without its rounding constants and shifts: out[k] is linear in in[].
constexpr void fdct_linear(const int32_t in[8], int32_t out[8])
{
    int32_t t0 = in[0] + in[7], t1 = in[1] + in[6], t2 = in[2] + in[5], t3 = in[3] + in[4];
    const int32_t t10 = t0 + t3, t11 = t1 + t2;
    int32_t t12 = t0 - t3, t13 = t1 - t2;
    t0 = in[0] - in[7]; t1 = in[1] - in[6]; t2 = in[2] - in[5]; t3 = in[3] - in[4];
    out[0] = t10 + t11;
    out[4] = t10 - t11;
    int32_t z1 = (t12 + t13) * F_0_541;
    out[2] = z1 + t12 * F_0_765;
    out[6] = z1 - t13 * F_1_847;
    t12 = t0 + t2;
    t13 = t1 + t3;
    z1 = (t12 + t13) * F_1_175;
    t12 = t12 * (-F_0_390) + z1;
    t13 = t13 * (-F_1_961) + z1;
    z1 = (t0 + t3) * (-F_0_899);
    out[1] = t0 * F_1_501 + z1 + t12;
    out[7] = t3 * F_0_298 + z1 + t13;
    z1 = (t1 + t2) * (-F_2_562);
    out[3] = t1 * F_3_072 + z1 + t13;
    out[5] = t2 * F_2_053 + z1 + t12;
}

constexpr Mat8 make_fdct_matrix()
{
    Mat8 m{};
    for (int j = 0; j < 8; ++j) {
        int32_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e[j] = 1;
        fdct_linear(e, o);
        for (int k = 0; k < 8; ++k) m.a[k][j] = o[k];
    }
    return m;
}

__constant__ Mat8 kFdct = make_fdct_matrix();

// natural index -> zig-zag position (inverse of encoder.rs UNZIGZAG)
__constant__ uint8_t kZigzagPos[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                                       41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                                       46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// Annex K Huffman tables (fl_jpeg_tables.h) expanded to (length << 16 | code) at compile time -- encoder.rs build_huff_lut
struct HuffLut { uint32_t e[256]; };

constexpr HuffLut make_lut(const HuffSpec &s)
{
    HuffLut t{};
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < s.len[l - 1]; ++i, ++k) t.e[s.val[k]] = ((uint32_t)l << 16) | code++;
        code <<= 1;
    }
    return t;
}

struct HuffAll { HuffLut ac[2], dc[2]; };
constexpr HuffAll make_all() { return HuffAll{{make_lut(kAcLuma), make_lut(kAcChroma)}, {make_lut(kDcLuma), make_lut(kDcChroma)}}; }
__constant__ HuffAll kHuff = make_all();

// ---------------------------------------------------------------- wave helpers --

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

// LDS traffic that stays inside one wave needs no hardware barrier (a wave's DS instructions execute in order);
// this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exclusive scan across the 256 threads of the workgroup; *total = sum.  s_w: 4 words of LDS.
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    __syncthreads(); // s_w may still be read from the previous call
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) { const uint32_t t = s_w[i]; if (i < wave) base += t; sum += t; }
    *total = sum;
    return base + inc - v;
}

__device__ __forceinline__ uint32_t coef_size(int32_t v) // encode_coefficient: bits of |v|
{
    const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
    return mag ? 32u - (uint32_t)__clz(mag) : 0u;
}

// What lane k >= 1 of a wave contributes to the bit stream of one block (BitWriter::write_block): a non-zero AC
// coefficient (zig-zag position k) with the zero run in front of it, ZRL codes included; lane 63 the end-of-block
// code when coefficient 63 is zero.  v is meaningful for lanes >= 1 only (lane 0 = DC, coded separately).
// ac = the component's 256-entry AC look-up table in LDS.  Returns the bit count, bits right-aligned in *bits.
__device__ __forceinline__ uint32_t ac_lane_code(int32_t v, uint32_t lane, const uint32_t *ac, uint64_t *bits)
{
    const uint64_t nzmask = __ballot(v != 0 && lane != 0u);
    uint32_t nb = 0;
    uint64_t b = 0;
    if (lane != 0u && v != 0) {
        const uint32_t size = coef_size(v);
        const uint32_t value = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
        const uint64_t lower = nzmask & ((1ull << lane) - 1ull);
        const uint32_t prev = lower ? 63u - (uint32_t)__clzll(lower) : 0u;     // position of the previous coded coefficient
        const uint32_t run = lane - prev - 1u;
        const uint32_t zrl = ac[0xF0], e = ac[((run & 15u) << 4) | size];
        const uint32_t zl = zrl >> 16, zc = zrl & 0xffffu;
        for (uint32_t i = 0; i < (run >> 4); ++i) { b = (b << zl) | zc; nb += zl; }  // while zero_run > 15 { 0xF0 }
        b = (b << (e >> 16)) | (e & 0xffffu);
        b = (b << size) | value;
        nb += (e >> 16) + size;
    } else if (lane == 63u) {
        const uint32_t e = ac[0]; // EOB
        nb = e >> 16;
        b = e & 0xffffu;
    }
    *bits = b;
    return nb;
}

// ---------------------------------------------------------------- kernel 1: colour + FDCT + quantise (+ AC bit count) --

constexpr int kBlocksPerWave = 8;                       // consecutive blocks one wave walks through
constexpr int kBlocksPerWg = 4 * kBlocksPerWave;

__device__ __forceinline__ void stage_ac_luts(uint32_t *s_ac)
{
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) s_ac[i] = kHuff.ac[i >> 8].e[i & 255u];
    __syncthreads();
}

__device__ __forceinline__ uint32_t block_pixel(const JpegJob &jb, uint32_t blk, uint32_t r, uint32_t c)
{
    const uint32_t brow = blk / jb.bx, bcol = blk - brow * jb.bx;
    // copy_blocks_ycbcr / pixel_at_or_near: pixels past the right / bottom edge repeat the last column / row
    uint32_t px = bcol * 8u + c, py = brow * 8u + r;
    px = px < jb.w ? px : jb.w - 1u;
    py = py < jb.h ? py : jb.h - 1u;
    uint32_t pr, pg, pb, pa;
    load_rgba(jb.src + ((size_t)py * jb.w + px) * jb.c, jb.c, pr, pg, pb, pa);
    return pr | (pg << 8) | (pb << 16);
}

__global__ __launch_bounds__(256) void jpeg_dct_quant_kernel(const JpegJob *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                             uint32_t job_base)
{
    __shared__ int32_t s_a[4][64], s_b[4][64];
    __shared__ uint32_t s_ac[512];
    const JpegJob jb = jobs[job_base + blockIdx.y];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, r = lane >> 3, c = lane & 7u;
    const uint32_t nblocks = jb.bx * jb.by;
    if (blockIdx.x * kBlocksPerWg >= nblocks) return; // whole workgroup idle (uniform)
    stage_ac_luts(s_ac);
    const uint32_t first = blockIdx.x * kBlocksPerWg + wave * kBlocksPerWave;
    if (first >= nblocks) return;                      // wave-uniform; from here on waves never synchronise with each other
    const uint32_t last = min(first + (uint32_t)kBlocksPerWave, nblocks);
    int32_t m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m1[j] = kFdct.a[c][j]; m2[j] = kFdct.a[r][j]; }
    const uint8_t *qt = reinterpret_cast<const uint8_t *>(arena + jb.tab_off) + 624;
    const float ql = (float)qt[lane], qc = (float)qt[64 + lane];
    const uint32_t zz = kZigzagPos[lane];
    int32_t *ta = s_a[wave], *tb = s_b[wave];
    uint32_t rgb = block_pixel(jb, first, r, c);
    for (uint32_t blk = first; blk < last; ++blk) {
        uint32_t smp[3];
        jfif_px(rgb, smp[0], smp[1], smp[2]);
        if (blk + 1u < last) rgb = block_pixel(jb, blk + 1u, r, c); // in flight while this block is transformed
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {
            ta[lane] = (int32_t)smp[comp];
            wave_lds_sync();
            // Pass 1 (rows): lane (r, c) produces horizontal frequency c of row r
            int32_t p = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) p += m1[j] * ta[r * 8 + j];
            int32_t v1;
            if (c == 0) v1 = (p - 8 * 128) << 2;          // level shift folded in, scaled by 2^PASS1_BITS
            else if (c == 4) v1 = p << 2;
            else v1 = (p + (1 << 10)) >> 11;              // CONST_BITS - PASS1_BITS
            tb[lane] = v1;
            wave_lds_sync();
            // Pass 2 (columns): lane (r, c) produces vertical frequency r of column c
            int32_t p2 = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) p2 += m2[j] * tb[j * 8 + c];
            int32_t d;
            if (r == 0 || r == 4) d = (p2 + 2) >> 2;
            else d = (p2 + (1 << 14)) >> 15;              // CONST_BITS + PASS1_BITS
            // encode_rgb "Quantization": ((d / 8) as f32 / f32::from(q)).round() as i32
            const int32_t qv = (int32_t)roundf(__fdiv_rn((float)(d / 8), comp ? qc : ql));
            // natural -> zig-zag order through LDS: lane k then owns zig-zag coefficient k
            ta[zz] = qv;
            wave_lds_sync();
            const int32_t zv = ta[lane];
            wave_lds_sync();
            const uint32_t unit = blk * 3u + comp;
            jb.coef[(size_t)unit * 64 + lane] = (int16_t)zv;
            uint64_t bits;
            const uint32_t nb = ac_lane_code(zv, lane, s_ac + (comp ? 256 : 0), &bits);
            const uint32_t ac_bits = wave_sum(nb);
            if (lane == 0u) jb.unit_off[unit] = ac_bits;   // the DC code is added by the scan (it needs the previous block)
        }
    }
}

// ---------------------------------------------------------------- kernel 2: bit offsets of all blocks --

__device__ __forceinline__ int32_t dc_diff(const JpegJob &jb, uint32_t u)
{
    // differential DC against the same component's previous block (encode_rgb: y_dcprev / cb_dcprev / cr_dcprev)
    return (int32_t)jb.coef[(size_t)u * 64] - (u >= 3u ? (int32_t)jb.coef[(size_t)(u - 3u) * 64] : 0);
}

__global__ __launch_bounds__(256) void jpeg_scan_kernel(const JpegJob *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                        uint32_t job_base)
{
    __shared__ uint32_t s_w[4], s_carry;
    const JpegJob jb = jobs[job_base + blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t nunits = jb.bx * jb.by * 3u;
    if (tid == 0u) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < nunits; base += 256u) {
        const uint32_t u = base + tid;
        uint32_t len = 0;
        if (u < nunits) {
            const uint32_t size = coef_size(dc_diff(jb, u));
            len = jb.unit_off[u] + (kHuff.dc[(u % 3u) ? 1 : 0].e[size] >> 16) + size;
        }
        uint32_t chunk;
        const uint32_t ex = wg_exclusive_scan(len, s_w, &chunk);
        const uint32_t carry = s_carry;
        if (u < nunits) jb.unit_off[u] = carry + ex;
        __syncthreads();
        if (tid == 0u) s_carry = carry + chunk;
        __syncthreads();
    }
    const uint32_t total_bits = s_carry;
    const uint32_t nbytes = (total_bits + 7u) >> 3;
    if ((uint64_t)nbytes + 16u > jb.raw_cap) { // cannot happen with the scratch the host sizes; never write out of bounds
        if (tid == 0u) { jb.unit_off[nunits] = 0u; jb.result[1] = 0u; atomicOr(&jb.result[0], FL_JPEG_RESULT_OVERFLOW); }
        return;
    }
    if (tid == 0u) jb.unit_off[nunits] = total_bits;
    const uint32_t nwords = (total_bits + 31u) / 32u + 1u;
    for (uint32_t i = tid; i < nwords; i += 256u) jb.raw[i] = 0u;
    // everything in front of the scan data
    const uint8_t *hdr = reinterpret_cast<const uint8_t *>(arena + jb.tab_off);
    for (uint32_t i = tid; i < kJpegHeaderBytes; i += 256u) if (i < jb.dst_cap) jb.dst[i] = hdr[i];
}

// ---------------------------------------------------------------- kernel 3: emit the code words --

constexpr int kUnitsPerWave = 8;

__global__ __launch_bounds__(256) void jpeg_emit_kernel(const JpegJob *__restrict__ jobs, uint32_t job_base)
{
    __shared__ uint32_t s_unit[4][64];
    __shared__ uint32_t s_ac[512];
    const JpegJob jb = jobs[job_base + blockIdx.y];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t nunits = jb.bx * jb.by * 3u;
    if (blockIdx.x * (4u * kUnitsPerWave) >= nunits) return; // whole workgroup idle (uniform)
    stage_ac_luts(s_ac);
    const uint32_t first = blockIdx.x * (4u * kUnitsPerWave) + wave * kUnitsPerWave;
    if (first >= nunits) return;
    if (jb.unit_off[nunits] == 0u) return; // the scan refused this picture
    uint32_t *buf = s_unit[wave];
    buf[lane] = 0u;
    // everything this wave will read, requested up front: the block loop below then runs out of registers
    int32_t v[kUnitsPerWave], dcp[kUnitsPerWave];
    uint32_t off[kUnitsPerWave + 1];
#pragma unroll
    for (int k = 0; k < kUnitsPerWave; ++k) {
        const uint32_t u = min(first + (uint32_t)k, nunits - 1u);
        v[k] = jb.coef[(size_t)u * 64 + lane];
        dcp[k] = u >= 3u ? (int32_t)jb.coef[(size_t)(u - 3u) * 64] : 0; // same component, previous block
    }
#pragma unroll
    for (int k = 0; k <= kUnitsPerWave; ++k) off[k] = jb.unit_off[min(first + (uint32_t)k, nunits)]; // nunits + 1 entries
#pragma unroll
    for (int k = 0; k < kUnitsPerWave; ++k) {
        const uint32_t u = first + (uint32_t)k;
        if (u >= nunits) break;
        const uint32_t table = (u % 3u) ? 1u : 0u;
        uint64_t bits;
        uint32_t nb = ac_lane_code(v[k], lane, s_ac + table * 256u, &bits);
        if (lane == 0u) {
            const int32_t diff = v[k] - dcp[k];
            const uint32_t size = coef_size(diff);
            const uint32_t value = (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << size) - 1u);
            const uint32_t e = kHuff.dc[table].e[size];
            nb = (e >> 16) + size;
            bits = ((uint64_t)(e & 0xffffu) << size) | value;
        }
        const uint32_t inc = wave_inclusive_scan(nb, lane);
        const uint32_t unit_bits = off[k + 1] - off[k];
        wave_lds_sync();
        if (nb) {
            const uint32_t p = (off[k] & 31u) + (inc - nb);
            const uint32_t wi = p >> 5, sh = p & 31u;
            const uint64_t left = bits << (64u - nb);            // left-aligned code word(s), at most 59 bits
            const uint64_t a = left >> sh;
            const uint32_t w0 = (uint32_t)(a >> 32), w1 = (uint32_t)a, w2 = sh ? (uint32_t)((left << (64u - sh)) >> 32) : 0u;
            if (w0) atomicOr(&buf[wi], w0);
            if (w1) atomicOr(&buf[wi + 1u], w1);
            if (w2) atomicOr(&buf[wi + 2u], w2);
        }
        wave_lds_sync();
        const uint32_t nw = ((off[k] & 31u) + unit_bits + 31u) >> 5; // <= 54: a block codes to at most 22 + 63 * 26 bits
        if (lane < nw) {
            const uint32_t word = __builtin_bswap32(buf[lane]);  // memory order = stream order
            buf[lane] = 0u;
            uint32_t *g = jb.raw + (off[k] >> 5) + lane;
            if (lane == 0u || lane == nw - 1u) atomicOr(g, word); // boundary words are shared with the neighbouring blocks
            else *g = word;
        }
    }
}

// ---------------------------------------------------------------- kernel 4: pad_byte, 0xFF stuffing, EOI --

__global__ __launch_bounds__(256) void jpeg_stuff_kernel(const JpegJob *__restrict__ jobs, uint32_t job_base)
{
    __shared__ uint32_t s_w[4], s_carry;
    const JpegJob jb = jobs[job_base + blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t nunits = jb.bx * jb.by * 3u;
    const uint32_t total_bits = jb.unit_off[nunits];
    if (total_bits == 0u) return; // refused by the scan (result words already say so)
    const uint32_t nbytes = (total_bits + 7u) >> 3, limit = jb.dst_cap;
    // BitWriter::pad_byte = write_bits(0x7F, 7): the last partial byte is filled with ones
    const uint32_t pad_word = (total_bits & 7u) ? (0xFFu >> (total_bits & 7u)) << (8u * ((total_bits >> 3) & 3u)) : 0u;
    if (tid == 0u) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < nbytes; base += 1024u) {
        const uint32_t i = base + tid * 4u;
        uint32_t word = i < nbytes ? jb.raw[i >> 2] : 0u;
        if ((i >> 2) == (total_bits >> 5)) word |= pad_word;
        const uint32_t valid = i < nbytes ? (nbytes - i < 4u ? nbytes - i : 4u) : 0u;
        uint32_t ff = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) ff += (k < valid && ((word >> (8u * k)) & 255u) == 255u) ? 1u : 0u;
        uint32_t chunk;
        const uint32_t ex = wg_exclusive_scan(ff, s_w, &chunk);
        const uint32_t carry = s_carry;
        uint32_t o = kJpegHeaderBytes + i + carry + ex;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            if (k < valid) {
                const uint32_t b = (word >> (8u * k)) & 255u;
                if (o < limit) jb.dst[o] = (uint8_t)b;
                ++o;
                if (b == 255u) { if (o < limit) jb.dst[o] = 0u; ++o; }
            }
        }
        __syncthreads();
        if (tid == 0u) s_carry = carry + chunk;
        __syncthreads();
    }
    if (tid == 0u) {
        const uint32_t end = kJpegHeaderBytes + nbytes + s_carry;
        if ((uint64_t)end + 2u <= limit) {
            jb.dst[end] = 0xFF; jb.dst[end + 1u] = 0xD9; // EOI
            jb.result[1] = end + 2u;
        } else {
            jb.result[1] = 0u;
            atomicOr(&jb.result[0], FL_JPEG_RESULT_OVERFLOW);
        }
    }
}

} // namespace

#define FL_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return e__; } while (0)

hipError_t launch_jpeg_encode(const JpegJob *jobs, const uint32_t *arena, uint32_t job_base, uint32_t njobs, uint32_t max_blocks,
                              hipStream_t st)
{
    if (!njobs || !max_blocks) return hipSuccess;
    hipLaunchKernelGGL(jpeg_dct_quant_kernel, dim3((max_blocks + kBlocksPerWg - 1) / kBlocksPerWg, njobs), dim3(256), 0, st, jobs, arena, job_base);
    FL_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_scan_kernel, dim3(njobs), dim3(256), 0, st, jobs, arena, job_base);
    FL_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_emit_kernel, dim3((max_blocks * 3u + 4u * kUnitsPerWave - 1u) / (4u * kUnitsPerWave), njobs), dim3(256), 0, st, jobs, job_base);
    FL_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_stuff_kernel, dim3(njobs), dim3(256), 0, st, jobs, job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace fl
