// fl_jpeg.hip -- the JPEG encoder's back half on gfx950: colour conversion + forward DCT + quantisation, then
// Huffman coding, byte stuffing and framing, so that what leaves the GPU is the finished JFIF stream.
//
// Replaces `JpegEncoder::new_with_quality(&mut buffer, q).encode_image(&img)` (reference src/handler.rs:274-278),
// i.e. image 0.25.6 src/codecs/jpeg/encoder.rs (encode_image, encode_rgb, BitWriter::write_block / write_bits /
// huffman_encode, build_* header helpers) and src/codecs/jpeg/transform.rs (fdct): baseline, three components,
// all sampling factors 1x1, Annex K quantisation tables scaled by quality, Annex K Huffman tables.
//
// Two kernels per batch:
//   jpeg_dct_quant_kernel   one wave per 8x8 block: the wave's 64 lanes are the 64 samples / coefficients.  The
//                           integer DCT of transform.rs (IJG jfdctint) is linear up to its final shift of each pass,
//                           so a lane computes ITS coefficient as an 8-term integer dot product with a constant
//                           matrix derived at compile time from the butterfly itself (int32 wrap-around arithmetic
//                           is a ring: the sums are identical bit for bit).
//   jpeg_entropy_kernel     one workgroup per image.  Phase A: every wave sizes whole blocks in parallel (lane k =
//                           zig-zag coefficient k; run lengths come from a ballot).  Scan: bit offset of every block.
//                           Phase B: the same lanes emit their code words at their offsets (LDS atomics per block,
//                           then word stores).  Phase C: 0xFF byte stuffing by a second scan, header, EOI, length.
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

// ---------------------------------------------------------------- kernel 1: colour + FDCT + quantise --

constexpr int kBlocksPerWg = 4;

__global__ __launch_bounds__(256) void jpeg_dct_quant_kernel(const JpegJob *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                             uint32_t job_base)
{
    __shared__ int32_t s_a[kBlocksPerWg][64], s_b[kBlocksPerWg][64];
    const JpegJob jb = jobs[job_base + blockIdx.y];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, r = lane >> 3, c = lane & 7u;
    const uint32_t nblocks = jb.bx * jb.by;
    const uint32_t blk = blockIdx.x * kBlocksPerWg + wave;
    if (blockIdx.x * kBlocksPerWg >= nblocks) return; // whole workgroup idle (uniform)
    const bool live = blk < nblocks;
    const uint32_t brow = live ? blk / jb.bx : 0u, bcol = live ? blk - brow * jb.bx : 0u;
    // copy_blocks_ycbcr / pixel_at_or_near: pixels past the right / bottom edge repeat the last column / row
    uint32_t px = bcol * 8u + c, py = brow * 8u + r;
    px = px < jb.w ? px : jb.w - 1u;
    py = py < jb.h ? py : jb.h - 1u;
    uint32_t pr, pg, pb, pa;
    load_rgba(jb.src + ((size_t)py * jb.w + px) * jb.c, jb.c, pr, pg, pb, pa);
    uint32_t smp[3];
    jfif_px(pr | (pg << 8) | (pb << 16), smp[0], smp[1], smp[2]);
    int32_t m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m1[j] = kFdct.a[c][j]; m2[j] = kFdct.a[r][j]; }
    const uint8_t *qt = reinterpret_cast<const uint8_t *>(arena + jb.tab_off) + 624;
    const uint32_t zz = kZigzagPos[lane];
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        s_a[wave][lane] = (int32_t)smp[comp];
        __syncthreads();
        // Pass 1 (rows): lane (r, c) produces horizontal frequency c of row r
        int32_t p = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) p += m1[j] * s_a[wave][r * 8 + j];
        int32_t v1;
        if (c == 0) v1 = (p - 8 * 128) << 2;          // level shift folded in, scaled by 2^PASS1_BITS
        else if (c == 4) v1 = p << 2;
        else v1 = (p + (1 << 10)) >> 11;              // CONST_BITS - PASS1_BITS
        s_b[wave][lane] = v1;
        __syncthreads();
        // Pass 2 (columns): lane (r, c) produces vertical frequency r of column c
        int32_t p2 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) p2 += m2[j] * s_b[wave][j * 8 + c];
        int32_t d;
        if (r == 0 || r == 4) d = (p2 + 2) >> 2;
        else d = (p2 + (1 << 14)) >> 15;              // CONST_BITS + PASS1_BITS
        // encode_rgb "Quantization": ((d / 8) as f32 / f32::from(q)).round() as i32
        const float q = (float)qt[(comp ? 64 : 0) + lane];
        const int32_t qv = (int32_t)roundf(__fdiv_rn((float)(d / 8), q));
        if (live) jb.coef[((size_t)blk * 3 + comp) * 64 + zz] = (int16_t)qv;
    }
}

// ---------------------------------------------------------------- kernel 2: entropy coding + framing --

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

// What lane `lane` of a wave contributes to the bit stream of one block (write_block in encoder.rs):
// lane 0 the DC difference, lane k a non-zero AC coefficient with the zero run in front of it (ZRL codes
// included), lane 63 the end-of-block code when coefficient 63 is zero.  Returns the bit count, bits right-aligned.
__device__ __forceinline__ uint32_t lane_code(int32_t v, uint32_t lane, uint32_t table, const uint32_t *s_ac, const uint32_t *s_dc,
                                              uint64_t *bits)
{
    const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
    const uint32_t size = mag ? 32u - (uint32_t)__clz(mag) : 0u;                 // encode_coefficient
    const uint32_t value = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << size) - 1u);
    const uint64_t nzmask = __ballot(v != 0 && lane != 0u);
    uint32_t nb = 0;
    uint64_t b = 0;
    if (lane == 0u) {
        const uint32_t e = s_dc[table * 16u + size];
        nb = (e >> 16) + size;
        b = ((uint64_t)(e & 0xffffu) << size) | value;
    } else if (v != 0) {
        const uint64_t lower = nzmask & ((1ull << lane) - 1ull);
        const uint32_t prev = lower ? 63u - (uint32_t)__clzll(lower) : 0u;     // position of the previous coded coefficient
        const uint32_t run = lane - prev - 1u;
        const uint32_t zrl = s_ac[table * 256u + 0xF0u], e = s_ac[table * 256u + (((run & 15u) << 4) | size)];
        const uint32_t zl = zrl >> 16, zc = zrl & 0xffffu;
        for (uint32_t i = 0; i < (run >> 4); ++i) { b = (b << zl) | zc; nb += zl; }  // while zero_run > 15 { 0xF0 }
        b = (b << (e >> 16)) | (e & 0xffffu);
        b = (b << size) | value;
        nb += (e >> 16) + size;
    } else if (lane == 63u) {
        const uint32_t e = s_ac[table * 256u]; // EOB
        nb = e >> 16;
        b = e & 0xffffu;
    }
    *bits = b;
    return nb;
}

__global__ __launch_bounds__(256) void jpeg_entropy_kernel(const JpegJob *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                           uint32_t job_base)
{
    __shared__ uint32_t s_ac[2 * 256], s_dc[2 * 16], s_w[4], s_unit[4][72], s_carry;
    const JpegJob jb = jobs[job_base + blockIdx.x];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t nunits = jb.bx * jb.by * 3u;
    for (uint32_t i = tid; i < 512u; i += 256u) s_ac[i] = kHuff.ac[i >> 8].e[i & 255u];
    if (tid < 32u) s_dc[tid] = kHuff.dc[tid >> 4].e[tid & 15u];
    for (uint32_t i = lane; i < 72u; i += 64u) s_unit[wave][i] = 0u;
    __syncthreads();

    // ---- phase A: bits per block -------------------------------------------------------------
    for (uint32_t u = wave; u < nunits; u += 4u) {
        int32_t v = jb.coef[(size_t)u * 64 + lane];
        if (lane == 0u && u >= 3u) v -= jb.coef[(size_t)(u - 3u) * 64]; // differential DC against the same component's previous block
        uint64_t bits;
        const uint32_t nb = lane_code(v, lane, (u % 3u) ? 1u : 0u, s_ac, s_dc, &bits);
        const uint32_t total = wave_sum(nb);
        if (lane == 0u) jb.unit_off[u] = total;
    }
    __threadfence_block();
    __syncthreads();

    // ---- scan: bit offset of every block ------------------------------------------------------
    if (tid == 0u) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < nunits; base += 256u) {
        const uint32_t u = base + tid;
        const uint32_t len = u < nunits ? jb.unit_off[u] : 0u;
        uint32_t chunk;
        const uint32_t ex = wg_exclusive_scan(len, s_w, &chunk);
        const uint32_t carry = s_carry;
        if (u < nunits) jb.unit_off[u] = carry + ex;
        __syncthreads();
        if (tid == 0u) s_carry = carry + chunk;
        __syncthreads();
    }
    const uint32_t total_bits = s_carry;
    if (tid == 0u) jb.unit_off[nunits] = total_bits;
    const uint32_t nbytes = (total_bits + 7u) >> 3;
    if ((uint64_t)nbytes + 16u > jb.raw_cap) { // cannot happen with the scratch the host sizes; never write out of bounds
        if (tid == 0u) { jb.result[1] = 0u; atomicOr(&jb.result[0], FL_JPEG_RESULT_OVERFLOW); }
        return;
    }
    const uint32_t nwords = (total_bits + 31u) / 32u + 1u;
    for (uint32_t i = tid; i < nwords; i += 256u) jb.raw[i] = 0u;
    __threadfence_block();
    __syncthreads();

    // ---- phase B: emit ---------------------------------------------------------------------------
    for (uint32_t u = wave; u < nunits; u += 4u) {
        int32_t v = jb.coef[(size_t)u * 64 + lane];
        if (lane == 0u && u >= 3u) v -= jb.coef[(size_t)(u - 3u) * 64];
        uint64_t bits;
        const uint32_t nb = lane_code(v, lane, (u % 3u) ? 1u : 0u, s_ac, s_dc, &bits);
        const uint32_t inc = wave_inclusive_scan(nb, lane);
        const uint32_t unit_bits = __shfl(inc, 63, 64);
        const uint32_t off = jb.unit_off[u];
        if (nb) {
            const uint32_t p = (off & 31u) + (inc - nb);
            const uint32_t wi = p >> 5, sh = p & 31u;
            const uint64_t left = bits << (64u - nb);            // left-aligned code word(s)
            const uint64_t a = left >> sh;
            const uint32_t w0 = (uint32_t)(a >> 32), w1 = (uint32_t)a, w2 = sh ? (uint32_t)((left << (64u - sh)) >> 32) : 0u;
            if (w0) atomicOr(&s_unit[wave][wi], w0);
            if (w1) atomicOr(&s_unit[wave][wi + 1u], w1);
            if (w2) atomicOr(&s_unit[wave][wi + 2u], w2);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t nw = ((off & 31u) + unit_bits + 31u) >> 5; // <= 56
        if (lane < nw) {
            const uint32_t word = __builtin_bswap32(s_unit[wave][lane]); // memory order = stream order
            s_unit[wave][lane] = 0u;
            uint32_t *g = jb.raw + (off >> 5) + lane;
            if (lane == 0u || lane == nw - 1u) atomicOr(g, word);        // boundary words are shared with the neighbours
            else *g = word;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence_block();
    __syncthreads();

    // ---- phase C: pad_byte, 0xFF stuffing, framing ----------------------------------------------
    uint8_t *raw8 = reinterpret_cast<uint8_t *>(jb.raw);
    if (tid == 0u && (total_bits & 7u)) raw8[total_bits >> 3] |= (uint8_t)(0xFFu >> (total_bits & 7u)); // write_bits(0x7F, 7)
    const uint8_t *hdr = reinterpret_cast<const uint8_t *>(arena + jb.tab_off);
    const uint32_t hdr_len = kJpegHeaderBytes, limit = jb.dst_cap;
    for (uint32_t i = tid; i < hdr_len; i += 256u) if (i < limit) jb.dst[i] = hdr[i];
    __threadfence_block();
    __syncthreads();
    if (tid == 0u) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < nbytes; base += 1024u) {
        const uint32_t i = base + tid * 4u;
        const uint32_t word = i < nbytes ? jb.raw[i >> 2] : 0u;
        const uint32_t valid = i < nbytes ? (nbytes - i < 4u ? nbytes - i : 4u) : 0u;
        uint32_t ff = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) ff += (k < valid && ((word >> (8u * k)) & 255u) == 255u) ? 1u : 0u;
        uint32_t chunk;
        const uint32_t ex = wg_exclusive_scan(ff, s_w, &chunk);
        const uint32_t carry = s_carry;
        uint32_t o = hdr_len + i + carry + ex;
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
        const uint32_t end = hdr_len + nbytes + s_carry; // + EOI
        if ((uint64_t)end + 2u <= limit) {
            jb.dst[end] = 0xFF; jb.dst[end + 1u] = 0xD9;
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
    hipLaunchKernelGGL(jpeg_entropy_kernel, dim3(njobs), dim3(256), 0, st, jobs, arena, job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace fl
