// fl_jpeg.hip -- the JPEG encoder's back half on gfx950: colour conversion + forward DCT + quantisation, then
// Huffman coding, byte stuffing and framing, so that what leaves the GPU is the finished JFIF stream.
//
// Replaces `JpegEncoder::new_with_quality(&mut buffer, q).encode_image(&img)` (reference src/handler.rs:274-278),
// i.e. image 0.25.6 src/codecs/jpeg/encoder.rs (encode_image, encode_rgb, BitWriter::write_block / write_bits /
// huffman_encode, build_* header helpers) and src/codecs/jpeg/transform.rs (fdct): baseline, three components,
// all sampling factors 1x1, Annex K quantisation tables scaled by quality, Annex K Huffman tables.
//
// Two kernels per batch:
//   jpeg_dct_quant_kernel   one wave per 8x8 block (a wave walks through 8 consecutive blocks): the 64 lanes are the
//                           64 samples / coefficients.  The integer DCT of transform.rs (IJG jfdctint) is linear up
//                           to the final shift of each pass, so a lane computes ITS coefficient as an 8-term integer
//                           dot product with a constant matrix derived at compile time from the butterfly itself
//                           (int32 wrap-around arithmetic is a ring: the sums are identical bit for bit).  The same
//                           wave then Huffman-codes the block's AC coefficients: lane k = zig-zag coefficient k, a
//                           ballot gives every lane its zero run, a DPP prefix sum places the code words, LDS
//                           atomics assemble them.  Out: quantised DC, AC bit count, AC bits (from bit 0).
//   jpeg_pack_kernel        one workgroup per picture: DC code sizes (they need the previous block) + AC sizes ->
//                           exclusive scan -> bit offset of every block; the blocks' bits are shifted into place in an
//                           LDS window of the stream (atomics), then pad_byte, 0xFF -> 0xFF00 stuffing by a second
//                           scan, header, EOI, length.  Long streams are handled window after window.
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

constexpr bool fdct_matrix_fits_i16()
{
    const Mat8 m = make_fdct_matrix();
    for (int k = 0; k < 8; ++k)
        for (int j = 0; j < 8; ++j)
            if (m.a[k][j] > 32767 || m.a[k][j] < -32768) return false;
    return true;
}
static_assert(fdct_matrix_fits_i16(), "the passes use v_dot2_i32_i16: matrix entries must be 16-bit");

typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i16x2 as_i16x2(uint32_t v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ uint32_t pack_i16(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
// 8-term dot product of packed 16-bit operands, exact in 32 bits (v_dot2_i32_i16: two multiply-adds per instruction)
__device__ __forceinline__ int32_t dot8_i16(const uint32_t m[4], const uint4 v)
{
    int32_t p = __builtin_amdgcn_sdot2(as_i16x2(m[0]), as_i16x2(v.x), 0, false);
    p = __builtin_amdgcn_sdot2(as_i16x2(m[1]), as_i16x2(v.y), p, false);
    p = __builtin_amdgcn_sdot2(as_i16x2(m[2]), as_i16x2(v.z), p, false);
    return __builtin_amdgcn_sdot2(as_i16x2(m[3]), as_i16x2(v.w), p, false);
}

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

template <int N>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v)
{
    // lane i receives lane i - N of its row of 16; lanes without a source get 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, false);
}

// Inclusive prefix sum over the 64 lanes in six DPP adds: four steps inside the rows of 16, then the row totals travel by
// row_bcast:15 (rows 1 and 3 take the last lane of the row before) and row_bcast:31 (rows 2 and 3 take lane 31).
// (VALU latency only; ds_bpermute based shuffles cost an LDS round trip per step, v_readlane + v_cndmask per row six more
// vector instructions -- this kernel runs at 97 % VALU issue, profiles/r03_jpeg_pmc.txt, so instructions are its time.)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
    (void)lane;
    v += dpp_row_shr<1>(v);
    v += dpp_row_shr<2>(v);
    v += dpp_row_shr<4>(v);
    v += dpp_row_shr<8>(v);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 row_mask:0xa
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 row_mask:0xc
    return v;
}

// LDS traffic that stays inside one wave needs no hardware barrier (a wave's DS instructions execute in order);
// this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr uint32_t kPackThreads = 512; // jpeg_pack_kernel: one workgroup per picture; its phases are chains of short dependent steps
constexpr uint32_t kPackWaves = kPackThreads / 64;

// exclusive scan across the threads of the pack workgroup; *total = sum.  s_w: kPackWaves words of LDS.
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    __syncthreads(); // s_w may still be read from the previous call
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < kPackWaves; ++i) { const uint32_t t = s_w[i]; if (i < wave) base += t; sum += t; }
    *total = sum;
    return base + inc - v;
}

__device__ __forceinline__ uint32_t coef_size(int32_t v) // encode_coefficient: bits of |v|
{
    const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
    return mag ? 32u - (uint32_t)__clz(mag) : 0u;
}

// Huffman coding of one block's AC coefficients by one wave (BitWriter::write_block without the DC term): lane k
// holds zig-zag coefficient k (lane 0, the DC term, does not take part).  A non-zero lane emits its (run, size)
// code and value bits, preceded by ZRL codes when the zero run in front of it exceeds 15; lane 63 emits the
// end-of-block code when coefficient 63 is zero.  The code words are OR-ed into `tu` (LDS, zero on entry) from
// bit 0, most significant bit first.  Returns the number of bits.  ac = the component's 256-entry table in LDS;
// lt_lo / lt_hi = mask of the lanes below this one.
__device__ __forceinline__ uint32_t code_ac_block(int32_t zv, uint32_t lane, uint32_t lt_lo, uint32_t lt_hi, const uint32_t *ac,
                                                  uint32_t *tu)
{
    const bool nz = zv != 0 && lane != 0u;
    const uint64_t mask = __ballot(nz);
    const uint32_t eob = ac[0];
    if (mask == 0ull) { // DC-only block (wave-uniform): just the end-of-block code
        if (lane == 0u) tu[0] = (eob & 0xffffu) << (32u - (eob >> 16));
        return eob >> 16;
    }
    const uint32_t mag = (uint32_t)(zv < 0 ? -zv : zv);
    const uint32_t size = mag ? 32u - (uint32_t)__clz(mag) : 0u;               // encode_coefficient
    const uint32_t value = (uint32_t)(zv + (zv >> 31)) & ((1u << size) - 1u);  // negative: (v - 1) & mask
    const uint32_t lo = (uint32_t)mask & lt_lo, hi = (uint32_t)(mask >> 32) & lt_hi;
    const uint32_t prev = hi ? 63u - (uint32_t)__clz(hi) : (lo ? 31u - (uint32_t)__clz(lo) : 0u); // previous coded coefficient
    const uint32_t run = lane - prev - 1u;
    uint32_t nb = 0, code = 0;
    if (nz) {
        const uint32_t e = ac[((run & 15u) << 4) | size];
        nb = (e >> 16) + size;                       // <= 16 + 10
        code = ((e & 0xffffu) << size) | value;
    } else if (lane == 63u) {
        nb = eob >> 16;
        code = eob & 0xffffu;
    }
    if (__ballot(nz && run > 15u) == 0ull) {
        // common case, no ZRL anywhere in the block: every lane's code fits one 32-bit word
        const uint32_t inc = wave_inclusive_scan(nb, lane);
        if (nb) {
            const uint32_t p = inc - nb, wi = p >> 5, sh = p & 31u;
            const uint32_t left = code << (32u - nb);
            const uint32_t w1 = (left << 1) << (31u - sh);
            atomicOr(&tu[wi], left >> sh);
            if (w1) atomicOr(&tu[wi + 1u], w1);
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    // "while zero_run > 15 { huffman_encode(0xF0) }": up to three ZRL codes in front, 59 bits in all
    uint64_t bits = code;
    if (nz) {
        const uint32_t zrl = ac[0xF0], zl = zrl >> 16, zc = zrl & 0xffffu;
        uint64_t pre = 0;
        uint32_t pl = 0;
        for (uint32_t i = 0; i < (run >> 4); ++i) { pre = (pre << zl) | zc; pl += zl; }
        bits |= pre << nb;
        nb += pl;
    }
    const uint32_t inc = wave_inclusive_scan(nb, lane);
    if (nb) {
        const uint32_t p = inc - nb, wi = p >> 5, sh = p & 31u;
        const uint64_t left = bits << (64u - nb);
        const uint64_t a = left >> sh;
        const uint32_t w0 = (uint32_t)(a >> 32), w1 = (uint32_t)a, w2 = sh ? (uint32_t)((left << (64u - sh)) >> 32) : 0u;
        if (w0) atomicOr(&tu[wi], w0);
        if (w1) atomicOr(&tu[wi + 1u], w1);
        if (w2) atomicOr(&tu[wi + 2u], w2);
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
}

// ---------------------------------------------------------------- kernel 1: colour + FDCT + quantise (+ AC bit count) --

constexpr int kBlocksPerWave = 8;                       // consecutive blocks one wave walks through
constexpr int kBlocksPerWg = 4 * kBlocksPerWave;

__device__ __forceinline__ void stage_ac_luts(uint32_t *s_ac)
{
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) s_ac[i] = kHuff.ac[i >> 8].e[i & 255u];
    __syncthreads();
}

__device__ __forceinline__ uint32_t block_pixel(const JpegJob &jb, uint32_t brow, uint32_t bcol, uint32_t r, uint32_t c)
{
    // copy_blocks_ycbcr / pixel_at_or_near: pixels past the right / bottom edge repeat the last column / row
    uint32_t px = bcol * 8u + c, py = brow * 8u + r;
    px = px < jb.w ? px : jb.w - 1u;
    py = py < jb.h ? py : jb.h - 1u;
    if (jb.c == 4u && (reinterpret_cast<uintptr_t>(jb.src) & 3u) == 0u) // every letterboxed picture: one dword per pixel
        return reinterpret_cast<const uint32_t *>(jb.src)[(size_t)py * jb.w + px] & 0xffffffu;
    uint32_t pr, pg, pb, pa;
    load_rgba(jb.src + ((size_t)py * jb.w + px) * jb.c, jb.c, pr, pg, pb, pa);
    return pr | (pg << 8) | (pb << 16);
}

// Forward DCT + quantisation of one component of one block: lane (r, c) in, lane k = zig-zag coefficient k out.
__device__ __forceinline__ int32_t dct_quant_unit(uint32_t sample, uint32_t lane, uint32_t r, uint32_t c, const uint32_t (&m1)[4], const uint32_t (&m2)[4],
                                                  uint4 k1, uint32_t q, uint32_t magic, uint32_t zz, int16_t *ta16, int16_t *tb16, int32_t *ta)
{
    ta16[lane] = (int16_t)sample;
    wave_lds_sync();
    // Pass 1 (rows): lane (r, c) produces horizontal frequency c of row r
    const int32_t p = dot8_i16(m1, *reinterpret_cast<const uint4 *>(ta16 + r * 8)); // one 16-byte read: the row's 8 samples
    // transform.rs: c = 0: (p - 8 * 128) << PASS1_BITS (level shift folded in), c = 4: p << PASS1_BITS, else (p + 2^10) >> 11
    // (CONST_BITS - PASS1_BITS).  One form for all lanes, (p << s1 + a1) >> 11 with per-lane s1 = 13 or 0 (the sums of columns 0
    // and 4 are below 2^12, so nothing is shifted out): no lane-dependent branches in a kernel whose time is its vector instructions.
    const int32_t v1 = ((p << k1.x) + (int32_t)k1.y) >> 11;
    tb16[c * 8 + r] = (int16_t)v1;                // |v1| <= 255 * 8 * 4; transposed: the column pass reads 8 consecutive values
    wave_lds_sync();
    // Pass 2 (columns): lane (r, c) produces vertical frequency r of column c: rows 0 and 4 (p2 + 2) >> 2, else (p2 + 2^14) >> 15
    const int32_t p2 = dot8_i16(m2, *reinterpret_cast<const uint4 *>(tb16 + c * 8));
    const int32_t d = (p2 + (int32_t)k1.z) >> k1.w;
    // encode_rgb "Quantization": ((d / 8) as f32 / f32::from(q)).round() as i32.  |d / 8| <= 2048 and q <= 255,
    // so the f32 quotient cannot round onto or across a half (nearest miss: 1 / (2q) >= 2^-9 away, f32 error
    // <= 2^-14): it equals the exact round-half-away  sign(n) * floor((2|n| + q) / 2q), taken with the
    // per-coefficient reciprocal ceil(2^32 / 2q) from the table block (exact for 2|n| + q < 2^13).
    // i32 division truncates toward zero: |d / 8| = |d| >> 3, and the sign of a non-zero quotient is d's.
    const int32_t sg = d >> 31;
    const uint32_t an = (uint32_t)((d ^ sg) - sg) >> 3;
    const uint32_t rq = __umulhi(2u * an + q, magic);
    const int32_t qv = ((int32_t)rq ^ sg) - sg;
    // natural -> zig-zag order through LDS: lane k then owns zig-zag coefficient k
    ta[zz] = qv;
    wave_lds_sync();
    const int32_t zv = ta[lane];
    wave_lds_sync();
    return zv;
}

// One wave per block; the three components of a block are transformed one after the other, then Huffman-coded TOGETHER:
// the kernel is bound by vector-instruction issue (97 % VALU, profiles/r03_jpeg_pmc.txt), a block's three 64-coefficient
// units rarely hold more than a few dozen non-zero coefficients between them, and coding works on non-zero coefficients
// only -- so they are compacted into one list (rank of a lane among its unit's non-zero lanes = v_mbcnt of the ballot), one
// lane per code word: (run, size) lookup, value bits, ZRL prefixes, a segmented prefix sum for the bit positions and the LDS
// ORs then run once per block instead of once per component.  A block with more than 64 code words (non-zero coefficients
// plus end-of-block codes) takes the per-component path, code_ac_block, which is also what every stream is checked against
// (the oracle encoder codes blocks the reference's way, one BitWriter call after the other).
__global__ __launch_bounds__(256) void jpeg_dct_quant_kernel(const JpegJob *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                             uint32_t job_base)
{
    __shared__ int32_t s_a[4][64];                       // zig-zag exchange
    __shared__ __attribute__((aligned(16))) int16_t s_a16[4][64], s_b16[4][64]; // samples / pass-1 results (both fit 16 bits)
    __shared__ uint32_t s_ac[512], s_u[4][3][64];        // AC code tables (luma, chroma); per wave and component: the unit's AC bits
    __shared__ uint32_t s_ent[4][72];                    // per wave: [0] a pad entry, then the block's compacted code-word list
    const JpegJob jb = jobs[job_base + blockIdx.y];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, r = lane >> 3, c = lane & 7u;
    const uint32_t nblocks = jb.bx * jb.by;
    if (blockIdx.x * kBlocksPerWg >= nblocks) return; // whole workgroup idle (uniform)
#pragma unroll
    for (int k = 0; k < 3; ++k) s_u[wave][k][lane] = 0u;
    if (lane == 0) s_ent[wave][0] = 7u << 24;          // (component 7: no real entry continues it)
    stage_ac_luts(s_ac);
    const uint32_t first = blockIdx.x * kBlocksPerWg + wave * kBlocksPerWave;
    if (first >= nblocks) return;                      // wave-uniform; from here on waves never synchronise with each other
    const uint32_t last = min(first + (uint32_t)kBlocksPerWave, nblocks);
    uint32_t m1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m1[j] = pack_i16(kFdct.a[c][2 * j], kFdct.a[c][2 * j + 1]);
        m2[j] = pack_i16(kFdct.a[r][2 * j], kFdct.a[r][2 * j + 1]);
    }
    const uint8_t *qt = reinterpret_cast<const uint8_t *>(arena + jb.tab_off) + 624;
    const uint32_t ql = qt[lane], qc = qt[64 + lane];
    const uint32_t *magic = reinterpret_cast<const uint32_t *>(qt + 128);
    const uint32_t ml = magic[lane], mc = magic[64 + lane];
    const uint32_t lt_lo = lane >= 32u ? 0xffffffffu : (1u << lane) - 1u, lt_hi = lane >= 32u ? (1u << (lane - 32u)) - 1u : 0u;
    const uint32_t zz = kZigzagPos[lane];
    // per-lane shift / rounding constants of the two passes (see dct_quant_unit)
    const uint4 k1 = {(c == 0u || c == 4u) ? 13u : 0u, c == 0u ? (uint32_t)(-(8 * 128) * 8192) : (c == 4u ? 0u : 1u << 10),
                      (r == 0u || r == 4u) ? 2u : 1u << 14, (r == 0u || r == 4u) ? 2u : 15u};
    int32_t *ta = s_a[wave];
    int16_t *ta16 = s_a16[wave], *tb16 = s_b16[wave];
    uint32_t *ent = s_ent[wave];
    uint32_t brow = first / jb.bx, bcol = first - brow * jb.bx; // (one division per wave; the walk below steps them)
    uint32_t rgb = block_pixel(jb, brow, bcol, r, c);
    for (uint32_t blk = first; blk < last; ++blk) {
        uint32_t smp[3];
        jfif_px(rgb, smp[0], smp[1], smp[2]);
        // a block of one colour (the fill frame of a letterboxed picture: 75 of the 950 blocks of config 1, 31 % of config 0's)
        const bool flat = __ballot(rgb != (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb)) == 0ull;
        if (++bcol == jb.bx) { bcol = 0; ++brow; }
        if (blk + 1u < last) rgb = block_pixel(jb, brow, bcol, r, c); // in flight while this block is transformed
        if (flat) {
            // Both passes of the transform are linear up to their final shifts and every row of the matrix but the first sums to zero:
            // a constant block s has pass-1 column 0 = 4 (8 s - 1024) in every row, nothing else, and pass 2 turns that into
            // d[0] = (32 (8 s - 1024) + 2) >> 2 = 64 (s - 128); all other d are (0 + rounding) >> shift = 0.  The DC term goes through
            // the quantiser of dct_quant_unit, each component's AC code is its end-of-block code alone.
            const uint32_t q0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)ql), q0c = (uint32_t)__builtin_amdgcn_readfirstlane((int)qc);
            const uint32_t m0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)ml), m0c = (uint32_t)__builtin_amdgcn_readfirstlane((int)mc);
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) {
                const int32_t d = 64 * ((int32_t)smp[comp] - 128);
                const int32_t sg = d >> 31;
                const uint32_t an = (uint32_t)((d ^ sg) - sg) >> 3;
                const uint32_t rq = __umulhi(2u * an + (comp ? q0c : q0l), comp ? m0c : m0l);
                const int32_t qv = ((int32_t)rq ^ sg) - sg;
                const uint32_t e = s_ac[comp ? 256u : 0u]; // (run 0, size 0): end of block
                if (lane == 0u) {
                    const uint32_t unit = blk * 3u + (uint32_t)comp;
                    jb.acbits[(size_t)unit * kAcWordsPerUnit] = (e & 0xffffu) << (32u - (e >> 16));
                    jb.meta[unit] = ((uint32_t)qv & 0xffffu) | ((e >> 16) << 16);
                }
            }
            continue;
        }
        int32_t zv[3];
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) zv[comp] = dct_quant_unit(smp[comp], lane, r, c, m1, m2, k1, comp ? qc : ql, comp ? mc : ml, zz, ta16, tb16, ta);
        // ---- the block's code words: per component its non-zero AC coefficients in zig-zag order, then an end-of-block code
        // unless coefficient 63 is coded (BitWriter::write_block) ----
        uint64_t mask[3];
        uint32_t base[3], total = 0;
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {
            mask[comp] = __ballot(zv[comp] != 0 && lane != 0u);
            base[comp] = total;
            total += (uint32_t)__popcll(mask[comp]) + ((mask[comp] >> 63) ? 0u : 1u);
        }
        uint32_t ac_bits[3];
        if (total <= 64u) {
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) {
                const uint32_t lo = (uint32_t)mask[comp], hi = (uint32_t)(mask[comp] >> 32);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u)); // non-zero lanes below this one
                if (zv[comp] != 0 && lane != 0u) ent[1u + base[comp] + rank] = ((uint32_t)zv[comp] & 0xffffu) | (lane << 16) | ((uint32_t)comp << 24); // value, zig-zag position, component
                if (lane == 0u && !(mask[comp] >> 63)) ent[base[comp] + (uint32_t)__popcll(mask[comp]) + 1u] = (64u << 16) | ((uint32_t)comp << 24); // position 64: end of block
            }
            wave_lds_sync();
            const bool act = lane < total;
            const uint32_t prev = ent[lane], cur = act ? ent[lane + 1u] : (64u << 16);
            wave_lds_sync();
            const uint32_t comp = (cur >> 24) & 3u, pos = (cur >> 16) & 127u;
            const bool is_eob = pos == 64u;
            const uint32_t prevpos = ((prev ^ cur) >> 24) ? 0u : ((prev >> 16) & 127u); // a unit's first code word: its run starts behind the DC term
            const uint32_t run = pos - prevpos - 1u;
            const int32_t v = (int32_t)(int16_t)(cur & 0xffffu);
            const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
            const uint32_t size = is_eob ? 0u : 32u - (uint32_t)__clz(mag | 1u); // encode_coefficient (a listed coefficient is not zero)
            const uint32_t value = (uint32_t)(v + (v >> 31)) & ((1u << size) - 1u);                 // negative: (v - 1) & mask
            const uint32_t *tab = s_ac + (comp ? 256u : 0u);
            const uint32_t e = tab[is_eob ? 0u : (((run & 15u) << 4) | size)];
            uint32_t nb = act ? (e >> 16) + size : 0u;              // <= 16 + 10
            const uint32_t code = ((e & 0xffffu) << size) | value;
            const uint32_t nzrl = (act && !is_eob) ? run >> 4 : 0u; // "while zero_run > 15 { huffman_encode(0xF0) }": up to three in front
            uint32_t *tuc = s_u[wave][comp];
            uint32_t inc, pbit;
            if (__ballot(nzrl != 0u) == 0ull) {
                // no ZRL anywhere in the block: every code word fits one 32-bit word
                inc = wave_inclusive_scan(nb, lane);
                const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)base[1] - 1), s2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)base[2] - 1);
                pbit = inc - nb - (comp == 0u ? 0u : (comp == 1u ? s1 : s2));
                if (nb) {
                    const uint32_t wi = pbit >> 5, sh = pbit & 31u;
                    const uint32_t left = code << (32u - nb);
                    const uint32_t w1 = (left << 1) << (31u - sh);
                    atomicOr(&tuc[wi], left >> sh);
                    if (w1) atomicOr(&tuc[wi + 1u], w1);
                }
                ac_bits[0] = s1; ac_bits[1] = s2 - s1;
            } else {
                const uint32_t zrl = tab[0xF0], zl = zrl >> 16, zc = zrl & 0xffffu;
                uint64_t pre = 0;
#pragma unroll
                for (uint32_t i = 0; i < 3; ++i) if (i < nzrl) pre = (pre << zl) | zc;
                const uint64_t bits = (pre << nb) | code; // <= 3 * 16 + 26 bits
                nb += nzrl * zl;
                inc = wave_inclusive_scan(nb, lane);
                const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)base[1] - 1), s2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)base[2] - 1);
                pbit = inc - nb - (comp == 0u ? 0u : (comp == 1u ? s1 : s2));
                if (nb) {
                    const uint32_t wi = pbit >> 5, sh = pbit & 31u;
                    const uint64_t left = bits << (64u - nb);
                    const uint64_t a2 = left >> sh;
                    const uint32_t w0 = (uint32_t)(a2 >> 32), w1 = (uint32_t)a2, w2 = sh ? (uint32_t)((left << (64u - sh)) >> 32) : 0u;
                    if (w0) atomicOr(&tuc[wi], w0);
                    if (w1) atomicOr(&tuc[wi + 1u], w1);
                    if (w2) atomicOr(&tuc[wi + 2u], w2);
                }
                ac_bits[0] = s1; ac_bits[1] = s2 - s1;
            }
            ac_bits[2] = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63) - ac_bits[0] - ac_bits[1];
            wave_lds_sync();
        } else {
            // more code words than lanes: component by component, as the reference writes them
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) {
                ac_bits[comp] = code_ac_block(zv[comp], lane, lt_lo, lt_hi, s_ac + (comp ? 256 : 0), s_u[wave][comp]);
                wave_lds_sync();
            }
        }
        // ---- out: the units' AC bits (from bit 0 of their own buffers: the pack kernel shifts them into place), quantised DC, AC bit count
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) {
            const uint32_t unit = blk * 3u + (uint32_t)comp;
            uint32_t *tu = s_u[wave][comp];
            const uint32_t nw = (ac_bits[comp] + 31u) >> 5;          // <= 52 words: 63 * 26 bits
            if (lane < nw) { jb.acbits[(size_t)unit * kAcWordsPerUnit + lane] = tu[lane]; tu[lane] = 0u; }
            if (lane == 0u) jb.meta[unit] = ((uint32_t)zv[comp] & 0xffffu) | (ac_bits[comp] << 16); // quantised DC, AC bit count
        }
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------- kernel 2: offsets, assembly, stuffing, framing --

constexpr uint32_t kWinWords = 8192; // LDS window of the bit stream: 32 KB = 262,144 bits

// ORs the 32-bit word `val`, whose most significant bit sits at stream bit g, into the window [wbase, wbase + kWinWords)
__device__ __forceinline__ void win_or(uint32_t *win, uint32_t wbase, uint64_t g, uint32_t val)
{
    const uint64_t W = g >> 5;
    const uint32_t sh = (uint32_t)g & 31u;
    const uint32_t hi = val >> sh, lo = sh ? val << (32u - sh) : 0u;
    if (hi && W >= wbase && W < (uint64_t)wbase + kWinWords) atomicOr(&win[W - wbase], hi);
    if (lo && W + 1 >= wbase && W + 1 < (uint64_t)wbase + kWinWords) atomicOr(&win[W + 1 - wbase], lo);
}

__device__ __forceinline__ uint32_t dc_code(const JpegJob &jb, uint32_t u, uint32_t meta, uint32_t *len)
{
    // differential DC against the same component's previous block (encode_rgb: y_dcprev / cb_dcprev / cr_dcprev)
    const int32_t dc = (int16_t)(meta & 0xffffu), prev = u >= 3u ? (int32_t)(int16_t)(jb.meta[u - 3u] & 0xffffu) : 0;
    const int32_t diff = dc - prev;
    const uint32_t size = coef_size(diff);
    const uint32_t value = (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << size) - 1u);
    const uint32_t e = kHuff.dc[(u % 3u) ? 1 : 0].e[size];
    *len = (e >> 16) + size;                                   // <= 9 + 11 or 11 + 11 bits
    return (((e & 0xffffu) << size) | value) << (32u - *len);  // left-aligned
}

__global__ __launch_bounds__(kPackThreads) void jpeg_pack_kernel(const JpegJob *__restrict__ jobs, const uint32_t *__restrict__ arena,
                                                                 uint32_t job_base)
{
    __shared__ uint32_t s_win[kWinWords];
    __shared__ uint32_t s_w[kPackWaves], s_carry, s_lo, s_hi;
    const JpegJob jb = jobs[job_base + blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t nunits = jb.bx * jb.by * 3u;
    constexpr uint32_t T = kPackThreads;

    // ---- bit offset of every block: DC code size (needs the previous block) + AC size, exclusive scan -- and, in the same pass,
    // the block's code goes into the first window of the stream (win_or drops what lies beyond it): the words a thread will
    // place are requested together with its meta word, BEFORE the scan, so that a picture whose stream fits one window (every
    // 300 x 200 picture) costs one chain of dependent loads per 512 blocks instead of three ----
    for (uint32_t i = tid; i < kWinWords; i += T) s_win[i] = 0u;
    if (tid == 0u) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < nunits; base += T) {
        const uint32_t u = base + tid;
        uint32_t len = 0, dl = 0, dcw = 0, nw = 0, a0 = 0, a1 = 0;
        if (u < nunits) {
            const uint32_t m = jb.meta[u];
            nw = ((m >> 16) + 31u) >> 5;
            if (nw > 0u) a0 = jb.acbits[(size_t)u * kAcWordsPerUnit];
            if (nw > 1u) a1 = jb.acbits[(size_t)u * kAcWordsPerUnit + 1u];
            dcw = dc_code(jb, u, m, &dl);
            len = dl + (m >> 16);
        }
        uint32_t chunk;
        const uint32_t ex = wg_exclusive_scan(len, s_w, &chunk);
        const uint32_t carry = s_carry;
        if (u < nunits) {
            const uint32_t off = carry + ex;
            jb.unit_off[u] = off;
            win_or(s_win, 0u, off, dcw);
            if (nw > 0u) win_or(s_win, 0u, (uint64_t)off + dl, a0);
            if (nw > 1u) win_or(s_win, 0u, (uint64_t)off + dl + 32u, a1);
            for (uint32_t w = 2; w < nw; ++w) win_or(s_win, 0u, (uint64_t)off + dl + 32u * w, jb.acbits[(size_t)u * kAcWordsPerUnit + w]);
        }
        __syncthreads();
        if (tid == 0u) s_carry = carry + chunk;
        __syncthreads();
    }
    const uint32_t total_bits = s_carry;
    if (tid == 0u) jb.unit_off[nunits] = total_bits;
    __threadfence_block();
    __syncthreads();
    const uint32_t nbytes = (total_bits + 7u) >> 3, limit = jb.dst_cap;
    // everything in front of the scan data
    const uint8_t *hdr = reinterpret_cast<const uint8_t *>(arena + jb.tab_off);
    for (uint32_t i = tid; i < kJpegHeaderBytes; i += T) if (i < limit) jb.dst[i] = hdr[i];

    uint32_t ff_before = 0; // stuffed bytes emitted by earlier windows (same value in every thread)
    for (uint32_t wbase = 0; wbase * 32ull < total_bits; wbase += kWinWords) {
        const uint64_t wb = (uint64_t)wbase * 32u, we = wb + (uint64_t)kWinWords * 32u;
        if (wbase != 0u) { // (the first window was filled by the pass above)
            for (uint32_t i = tid; i < kWinWords; i += T) s_win[i] = 0u;
            // blocks that touch this window: offsets are ascending, so two binary searches bound them
            if (tid == 0u) {
                uint32_t lo = 0, hi = nunits;            // first u with unit_off[u + 1] > wb
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (jb.unit_off[mid + 1u] > wb) hi = mid; else lo = mid + 1u; }
                s_lo = lo;
                uint32_t lo2 = lo, hi2 = nunits;         // first u with unit_off[u] >= we
                while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2) >> 1; if (jb.unit_off[mid] >= we) hi2 = mid; else lo2 = mid + 1u; }
                s_hi = lo2;
            }
        } else if (tid == 0u) { s_lo = 0u; s_hi = 0u; }
        __syncthreads();
        const uint32_t u_lo = s_lo, u_hi = s_hi;
        // 2 lanes per block (a block's AC code is 1-3 words in ordinary pictures): lane 0 of the pair also places the DC code
        for (uint32_t ub = u_lo; ub < u_hi; ub += T / 2u) {
            const uint32_t u = ub + (tid >> 1), j = tid & 1u;
            if (u < u_hi) {
                const uint32_t m = jb.meta[u], off = jb.unit_off[u];
                uint32_t dl;
                const uint32_t dcw = dc_code(jb, u, m, &dl);
                if (j == 0u) win_or(s_win, wbase, off, dcw);
                const uint32_t nw = ((m >> 16) + 31u) >> 5;
                for (uint32_t w = j; w < nw; w += 2u)
                    win_or(s_win, wbase, (uint64_t)off + dl + 32u * w, jb.acbits[(size_t)u * kAcWordsPerUnit + w]);
            }
        }
        __syncthreads();
        // BitWriter::pad_byte = write_bits(0x7F, 7): the last partial byte is filled with ones
        if (tid == 0u && (total_bits & 7u) && (total_bits >> 5) >= wbase && (total_bits >> 5) < wbase + kWinWords)
            s_win[(total_bits >> 5) - wbase] |= (0xFFu >> (total_bits & 7u)) << (24u - 8u * ((total_bits >> 3) & 3u));
        __syncthreads();
        // 0xFF -> 0xFF 0x00 stuffing: scan of the 0xFF counts, then every thread writes its four bytes
        const uint32_t win_bytes = (uint32_t)min((uint64_t)kWinWords * 4u, (uint64_t)nbytes - (uint64_t)wbase * 4u);
        if (tid == 0u) s_carry = 0u;
        __syncthreads();
        for (uint32_t base = 0; base < win_bytes; base += 4u * T) {
            const uint32_t i = base + tid * 4u;
            const uint32_t word = i < win_bytes ? s_win[i >> 2] : 0u;     // stream order = most significant byte first
            const uint32_t valid = i < win_bytes ? (win_bytes - i < 4u ? win_bytes - i : 4u) : 0u;
            uint32_t ff = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) ff += (k < valid && ((word >> (24u - 8u * k)) & 255u) == 255u) ? 1u : 0u;
            uint32_t chunk;
            const uint32_t ex = wg_exclusive_scan(ff, s_w, &chunk);
            const uint32_t carry = s_carry;
            uint64_t o = (uint64_t)kJpegHeaderBytes + (uint64_t)wbase * 4u + i + ff_before + carry + ex;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < valid) {
                    const uint32_t b = (word >> (24u - 8u * k)) & 255u;
                    if (o < limit) jb.dst[o] = (uint8_t)b;
                    ++o;
                    if (b == 255u) { if (o < limit) jb.dst[o] = 0u; ++o; }
                }
            }
            __syncthreads();
            if (tid == 0u) s_carry = carry + chunk;
            __syncthreads();
        }
        ff_before += s_carry;
        __syncthreads();
    }
    if (tid == 0u) {
        const uint64_t end = (uint64_t)kJpegHeaderBytes + nbytes + ff_before;
        if (end + 2u <= limit) {
            jb.dst[end] = 0xFF; jb.dst[end + 1u] = 0xD9; // EOI
            jb.result[1] = (uint32_t)(end + 2u);
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
    hipLaunchKernelGGL(jpeg_pack_kernel, dim3(njobs), dim3(kPackThreads), 0, st, jobs, arena, job_base);
    FL_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace fl
