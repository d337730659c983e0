// fl_wtile.hip -- the window-tile matrix-pipe kernel for gfx950 (design, arithmetic and table layout: fl_wtile.h).
// Reference: image 0.25.6 imageops/sample.rs vertical_sample + horizontal_sample behind resize_exact and blur
// (src/handler.rs:229-255 of the reference).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <type_traits>

#include "fl_mfma.h"
#include "fl_pixel.h"
#include "fl_wtile.h"

namespace fl {

namespace {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned; // (gfx950 loads 16 bytes from any byte address)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef __attribute__((address_space(1))) uint32_t *gptr32;
typedef __attribute__((address_space(1))) uint8_t *gptr8;
typedef __attribute__((address_space(1))) u32_unaligned *gptr32u;

extern __shared__ __attribute__((aligned(16))) uint8_t wt_lds[];

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the rows requested for the
// NEXT step (the whole point of requesting them early) and for the pixel stores of the previous one (measured: 0.26 + 0.17 ms of the
// 1.30 ms at ratio 1.92 were exactly those two waits).  Everything the barriers of this kernel order lives in LDS.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// NSLOT x NKMAX = kWtOperandRegs: a wave keeps the horizontal operands of NSLOT N-tiles of up to NKMAX K-steps each in registers
// for its whole walk (<1, 6> also serves strips whose N-tiles mostly share ONE operand block: a blur's interior columns).
// WAVES: 8, or 4 for single-register-set plans of small pictures (round 5: a 300 x 200 blur is thirteen short steps of three barriers
// each -- with 4 waves per workgroup TWO workgroups share a CU (the register file holds 8 waves of 256 registers either way) and one
// picture's barriers and LDS round trips hide behind the other's work).
template <int NSLOT, int NKMAX, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 8 / WAVES) void resample_wtile_kernel(const Job *__restrict__ jobs, const WtItem *__restrict__ items,
                                                                       const uint32_t *__restrict__ arena, uint32_t lb, uint32_t invert, uint32_t framed)
{
    static_assert(NSLOT * NKMAX == (int)kWtOperandRegs, "operand register budget");
    constexpr uint32_t kWtWaves = (uint32_t)WAVES, kWtThreads = 64u * (uint32_t)WAVES; // (shadow the header's: this instantiation's workgroup)
    constexpr uint32_t WPRE = (kWtMaxKV * 192u + kWtThreads - 1u) / kWtThreads;      // 16-byte pieces of vertical operands a thread carries
    // experiments only (-DFL_ABLATE=mask, tools/build_ablate.sh with ABL_FILE=fl_wtile.hip): 1 = no vertical pass, 2 = no horizontal pass,
    // 4 = no stores to the destination, 8 = no row / operand traffic after a band's first window, 16 = horizontal MFMAs without the
    // recombination and the byte writes
#ifdef FL_ABLATE
    constexpr uint32_t ablate = FL_ABLATE;
#else
    constexpr uint32_t ablate = 0;
#endif
    const WtItem it = items[blockIdx.x];
    const Job jb = jobs[it.job];
    const uint32_t *plan = arena + it.plan_off;
    const WtHeader hd = *reinterpret_cast<const WtHeader *>(plan);
    const WtStrip sp = reinterpret_cast<const WtStrip *>(plan + hd.strip_off)[it.strip];
    const WtMTile *mts = reinterpret_cast<const WtMTile *>(plan + hd.mt_off);
    const WtNTile *nts = reinterpret_cast<const WtNTile *>(plan + hd.nt_off);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    const uint32_t SP = sp.sp, spw = SP >> 4, RR = hd.ring_rows;
    const uint32_t tn = sp.n1 - sp.n0, OP = wt_out_pitch(tn);
    // LDS: [ring: RR rows x SP] [planes: 3 x 16 rows x SP] [vertical operands of the M-tile: nkv_max x 3 KB] [output tile: 16 x OP bytes] [the strip's N-tile records]
    uint8_t *ring = wt_lds;
    uint8_t *planes = ring + RR * SP;
    u32x4 *wv_lds = reinterpret_cast<u32x4 *>(planes + 48u * SP);
    uint8_t *otile = reinterpret_cast<uint8_t *>(wv_lds) + hd.nkv_max * 3072u;
    u32x4 *ntl = reinterpret_cast<u32x4 *>(otile + 16u * OP);
    auto rmod = [&](uint32_t r) -> uint32_t { return r - RR * __umulhi(r, hd.ring_magic); }; // r mod RR (r < 65536)
    for (uint32_t k = tid; k < tn; k += kWtThreads) ntl[k] = reinterpret_cast<const u32x4 *>(nts + sp.n0)[k];

    // ---- source rows -> ring.  A thread owns one 16-byte column of the window and every rpp-th row; up to kWtPrefetch pieces stay in
    // registers between the request (before the horizontal pass) and the LDS write (after it). ----
    const uint32_t tcol = tid % spw, trow = tid / spw, rpp = kWtThreads / spw; // (one division per workgroup)
    const uint32_t pitch = hd.src_rowbytes;
    const gptr8 gsrc = (gptr8)jb.src;
    // Invert (reference src/handler.rs:226-228, color.rs Invert: 255 - c on the colour channels, alpha untouched) as an XOR on the way into
    // the ring (in `put`): a 16-byte piece starts on a pixel boundary for 2 and 4 channels, and 1 and 3 channels have no alpha
    const uint32_t inv = !invert ? 0u : hd.cs == 4u ? 0x00ffffffu : hd.cs == 2u ? 0x00ff00ffu : 0xffffffffu;
    auto fetch = [&](uint32_t r) -> u32x4 {
        const uint32_t row = min(r, hd.src_rows - 1u); // rows past the picture carry zero weights: any finite bytes do
        if (NSLOT == 1 && framed) { // (blur plans; this instantiation only: the others have no register to spare)
            // the plan's image is a frame of value jb.fill around the picture at jb.src (jb.rw x jb.rh bytes, tightly packed, at
            // column jb.cx, row jb.cy): reference src/handler.rs:238-248 builds exactly that image before img.blur()
            const uint32_t sr = row - jb.cy, sc = sp.col0 + 16u * tcol - jb.cx; // (wrap around for rows / columns in front of the picture)
            u32x4 v = {jb.fill, jb.fill, jb.fill, jb.fill};
            if (sr >= jb.rh) return v;
            const uint32_t o2 = sr * jb.rw + sc;
            if (sc < jb.rw && sc + 16u <= jb.rw) return *(const __attribute__((address_space(1))) u32x4_unaligned *)(gsrc + o2);
            for (uint32_t b = 0; b < 16u; ++b)
                if (sc + b < jb.rw) v[b >> 2] = (v[b >> 2] & ~(255u << (8u * (b & 3u)))) | ((uint32_t)gsrc[o2 + b] << (8u * (b & 3u)));
            return v;
        }
        const uint32_t off = row * pitch + sp.col0 + 16u * tcol;
        if (off + 16u <= jb.src_bytes) return *(const __attribute__((address_space(1))) u32x4_unaligned *)(gsrc + off);
        u32x4 v = {0u, 0u, 0u, 0u};
        for (uint32_t b = 0; b < 16u && off + b < jb.src_bytes; ++b) v[b >> 2] |= (uint32_t)gsrc[off + b] << (8u * (b & 3u));
        return v;
    };
    // (the XOR sits on the LDS side: applied to the load's result it would make `request` wait for the bytes it has just asked for --
    // measured: 1.29 -> 1.68 ms at ratio 1.92)
    auto put = [&](uint32_t r, u32x4 v) { *reinterpret_cast<u32x4 *>(ring + rmod(r) * SP + 16u * tcol) = v ^ inv; };
    u32x4 pre[kWtPrefetch];
    uint32_t pre_r0 = 0, pre_r1 = 0;      // rows requested for the coming step: [pre_r0, pre_r1)
    u32x4 wpre[WPRE];                     // ... and its vertical operands (nkv x 192 16-byte pieces over the workgroup's threads)
    uint32_t wpre_n = 0;
    auto request = [&](uint32_t r0, uint32_t r1, const WtMTile &m) __attribute__((always_inline)) {
        pre_r0 = r0; pre_r1 = r1;
#pragma unroll
        for (uint32_t k = 0; k < kWtPrefetch; ++k) {
            const uint32_t r = r0 + trow + k * rpp;
            if (trow < rpp && r < r1) pre[k] = fetch(r);
        }
        wpre_n = m.nk * 192u;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(plan + m.ops);
#pragma unroll
        for (uint32_t k = 0; k < WPRE; ++k)
            if (tid + k * kWtThreads < wpre_n) wpre[k] = src[tid + k * kWtThreads];
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t k = 0; k < kWtPrefetch; ++k) {
            const uint32_t r = pre_r0 + trow + k * rpp;
            if (trow < rpp && r < pre_r1) put(r, pre[k]);
        }
        // (more new rows than the registers held: the rest now, synchronously -- a band's first window, ratios just below the
        // streaming kernel's range)
        if (trow < rpp)
            for (uint32_t r = pre_r0 + trow + kWtPrefetch * rpp; r < pre_r1; r += rpp) put(r, fetch(r));
#pragma unroll
        for (uint32_t k = 0; k < WPRE; ++k)
            if (tid + k * kWtThreads < wpre_n) wv_lds[tid + k * kWtThreads] = wpre[k];
    };

    // ---- this wave's horizontal operands: its N-tiles are the same in every M-tile ----
    u32x4 hb[NSLOT][NKMAX][3];
    uint32_t cached[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const uint32_t nt = sp.n0 + wave + kWtWaves * (uint32_t)s;
        cached[s] = 0xffffffffu;
        uint32_t nk = 0;
        if (NSLOT == 1 && sp.common_ops != 0xffffffffu && sp.common_nk <= (uint32_t)NKMAX) { cached[s] = sp.common_ops; nk = sp.common_nk; }
        // (only what the NKMAX register steps hold: a tile with more K-steps -- a blur of sigma > ~21 whose tiles need 7 or 8 --
        // stays uncached and goes through fly(), which walks all of its steps; cached with its first NKMAX steps it would
        // match okA below and pair() would drop the rest of its taps)
        else if (nt < sp.n1 && nts[nt].nk <= (uint32_t)NKMAX) { cached[s] = nts[nt].ops; nk = nts[nt].nk; }
        const u32x4 *src = reinterpret_cast<const u32x4 *>(plan + (cached[s] != 0xffffffffu ? cached[s] : 0u));
#pragma unroll
        for (int k = 0; k < NKMAX; ++k)
#pragma unroll
            for (int d = 0; d < 3; ++d) hb[s][k][d] = (uint32_t)k < nk ? src[(k * 3 + d) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    }

    const uint32_t sh = hd.hs - 6u, slo = sh - 8u, s4 = 32u - sh, s3 = 24u - sh;
    const int32_t rnd = 1 << (slo - 1u);
    // the planes' offsets (2^22, 128 * 2^8, 128) times the weight sum, and the final rounding's half (fl_mfma.hip): a multiple of 2^8,
    // so it rides in as the first C operand of the L3 chain (which enters the sum shifted left by s3 <= 8) instead of an add per output
    constexpr int32_t round_add = -132112384 + (1 << 19);
    static_assert(round_add % 256 == 0, "folded into L3");
    const int32_t c3 = round_add >> s3;

    // nine products of one K-step of one N-tile: five accumulator chains (the first step starts them from constants: the matrix
    // instruction takes 0 as an inline operand, and the rounding constant of the recombination's right shift rides in L1)
    auto step = [&](auto first_tag, i32x4 (&L)[5], const u32x4 &a2u, const u32x4 &a1u, const u32x4 &a0u, const u32x4 &b2u, const u32x4 &b1u, const u32x4 &b0u) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const i32x4 a2 = __builtin_bit_cast(i32x4, a2u), a1 = __builtin_bit_cast(i32x4, a1u), a0 = __builtin_bit_cast(i32x4, a0u);
        const i32x4 b2 = __builtin_bit_cast(i32x4, b2u), b1 = __builtin_bit_cast(i32x4, b1u), b0 = __builtin_bit_cast(i32x4, b0u);
        const i32x4 z = {0, 0, 0, 0};
        // (the WEIGHTS go in as the matrix unit's first operand, the plane bytes as its second -- both are "16 bytes of K per lane", so the
        // registers are the same either way -- and the product comes out transposed: lane (row i, g) holds output bytes 4 g .. 4 g + 3
        // of ITS row, one dword of the output tile, instead of four rows of one byte column)
        L[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b2, a2, FIRST ? z : L[4], 0, 0, 0);
        L[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, a2, FIRST ? i32x4{c3, c3, c3, c3} : L[3], 0, 0, 0);
        L[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0, a2, FIRST ? z : L[2], 0, 0, 0);
        L[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0, a1, FIRST ? i32x4{rnd, rnd, rnd, rnd} : L[1], 0, 0, 0);
        L[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b0, a0, FIRST ? z : L[0], 0, 0, 0);
        L[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b2, a1, L[3], 0, 0, 0);
        L[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, a1, L[2], 0, 0, 0);
        L[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b1, a0, L[1], 0, 0, 0);
        L[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b2, a0, L[2], 0, 0, 0);
    };
    // sum = L4 2^32 + L3 2^24 + L2 2^16 + L1 2^8 + L0 in units of 2^-(14 + hs) of a pixel step -> 2^-20, then the byte
    auto emit = [&](const i32x4 (&L)[5], uint32_t jt) __attribute__((always_inline)) {
        if (ablate & 16u) { asm volatile("" : : "v"(L[0]), "v"(L[1]), "v"(L[2]), "v"(L[3]), "v"(L[4])); return; }
        uint32_t packed = 0u; // lane (row i, g): output bytes 4 g .. 4 g + 3 of the tile's row i
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int32_t low = (((L[2][r] << 8) + L[1][r]) + (L[0][r] >> 8)) >> slo;
            const int32_t x = (int32_t)(((uint32_t)L[4][r] << s4) + ((uint32_t)L[3][r] << s3) + (uint32_t)low);
            // (clamp-then-shift: clamp(x >> 20, 0, 255) is the pattern hipcc fuses into gfx950's broken v_ashr_pk_u8_i32)
            packed |= ((uint32_t)min(max(x, 0), (256 << 20) - 1) >> 20) << (8 * r);
        }
        // one dword per lane, every bank once (wt_out_pitch); round 4 wrote four bytes per lane, sixteen lanes into the same four dwords
        *reinterpret_cast<uint32_t *>(otile + i * OP + 16u * jt + 4u * g) = packed;
    };
    // Two N-tiles whose operands sit in registers, side by side: their 2 x 9 matrix instructions per K-step are independent of one another
    // (two waves per SIMD leave a lone tile's accumulator chains exposed).  K-steps past a tile's own count multiply by zero operands.
    auto pair = [&](auto sa_tag, auto sb_tag, uint32_t jtA, uint32_t jtB, bool hasB) __attribute__((always_inline)) {
        constexpr int SA = decltype(sa_tag)::value, SB = decltype(sb_tag)::value;
        const u32x4 tA = ntl[jtA], tB = ntl[hasB ? jtB : jtA];
        const uint8_t *pa = planes + i * SP + (tA[0] - sp.col0) + 16u * g, *pb = planes + i * SP + (tB[0] - sp.col0) + 16u * g;
        const uint32_t nkp = hasB ? max(tA[1], tB[1]) : tA[1];
        i32x4 LA[5], LB[5];
#pragma unroll
        for (int k = 0; k < NKMAX; ++k) {
            if (k > 0 && (uint32_t)k >= nkp) break; // (every tile has a first K-step)
            const u32x4 a2 = *reinterpret_cast<const u32x4 *>(pa + 64u * k), a1 = *reinterpret_cast<const u32x4 *>(pa + 64u * k + 16u * SP),
                        a0 = *reinterpret_cast<const u32x4 *>(pa + 64u * k + 32u * SP);
            const u32x4 c2 = *reinterpret_cast<const u32x4 *>(pb + 64u * k), c1 = *reinterpret_cast<const u32x4 *>(pb + 64u * k + 16u * SP),
                        c0 = *reinterpret_cast<const u32x4 *>(pb + 64u * k + 32u * SP);
            if (k == 0) {
                step(std::true_type{}, LA, a2, a1, a0, hb[SA][k][0], hb[SA][k][1], hb[SA][k][2]);
                if (hasB) step(std::true_type{}, LB, c2, c1, c0, hb[SB][k][0], hb[SB][k][1], hb[SB][k][2]);
            } else {
                step(std::false_type{}, LA, a2, a1, a0, hb[SA][k][0], hb[SA][k][1], hb[SA][k][2]);
                if (hasB) step(std::false_type{}, LB, c2, c1, c0, hb[SB][k][0], hb[SB][k][1], hb[SB][k][2]);
            }
        }
        emit(LA, jtA);
        if (hasB) emit(LB, jtB);
    };
    // a tile that shares nothing (a blur's border columns): operands from the L2
    auto fly = [&](uint32_t jt) __attribute__((always_inline)) {
        const u32x4 t = ntl[jt];
        const uint8_t *pa = planes + i * SP + (t[0] - sp.col0) + 16u * g;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(plan + t[2]);
        i32x4 L[5];
        {
            const u32x4 b2 = src[lane], b1 = src[64u + lane], b0 = src[128u + lane];
            const u32x4 a2 = *reinterpret_cast<const u32x4 *>(pa), a1 = *reinterpret_cast<const u32x4 *>(pa + 16u * SP), a0 = *reinterpret_cast<const u32x4 *>(pa + 32u * SP);
            step(std::true_type{}, L, a2, a1, a0, b2, b1, b0);
        }
        for (uint32_t k = 1; k < t[1]; ++k) {
            const u32x4 b2 = src[(k * 3u + 0u) * 64u + lane], b1 = src[(k * 3u + 1u) * 64u + lane], b0 = src[(k * 3u + 2u) * 64u + lane];
            const u32x4 a2 = *reinterpret_cast<const u32x4 *>(pa + 64u * k), a1 = *reinterpret_cast<const u32x4 *>(pa + 64u * k + 16u * SP),
                        a0 = *reinterpret_cast<const u32x4 *>(pa + 64u * k + 32u * SP);
            step(std::false_type{}, L, a2, a1, a0, b2, b1, b0);
        }
        emit(L, jt);
    };
    // output tile -> destination: 32 threads per row
    auto store_rows = [&](uint32_t mt, uint32_t rr) __attribute__((always_inline)) {
        const uint32_t t32 = tid & 31u, y = 16u * mt + rr;
        const uint32_t b0 = 16u * sp.n0, b1 = min(16u * sp.n1, hd.nout);
        if (y >= hd.rows) return;
        const uint8_t *orow = otile + rr * OP;
        const uint32_t cs = hd.cs, px0 = b0 / cs, npx = (b1 - b0) / cs;
        const size_t drow = (size_t)(jb.oy + y) * jb.dw + jb.ox + px0;
        if (lb) {
            gptr32 d = (gptr32)(reinterpret_cast<uint32_t *>(jb.dst) + drow);
            if (cs == 3u) { // four pixels = three dwords of the tile
                for (uint32_t q = 4u * t32; q < npx; q += 128u) {
                    const uint32_t *p = reinterpret_cast<const uint32_t *>(orow + 3u * q);
                    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
                    const uint32_t v0 = (w0 & 0xffffffu) | 0xff000000u, v1 = (w0 >> 24) | ((w1 & 0xffffu) << 8) | 0xff000000u,
                                   v2 = (w1 >> 16) | ((w2 & 0xffu) << 16) | 0xff000000u, v3 = (w2 >> 8) | 0xff000000u;
                    d[q] = v0;
                    if (q + 1u < npx) d[q + 1u] = v1;
                    if (q + 2u < npx) d[q + 2u] = v2;
                    if (q + 3u < npx) d[q + 3u] = v3;
                }
            } else {
                for (uint32_t q = t32; q < npx; q += 32u) {
                    const uint8_t *p = orow + q * cs;
                    uint32_t v;
                    if (cs == 1u) v = p[0] * 0x010101u | 0xff000000u;
                    else if (cs == 2u) v = blend_over_fill(jb.fill, p[0], p[0], p[0], p[1]);
                    else v = blend_over_fill(jb.fill, p[0], p[1], p[2], p[3]);
                    d[q] = v;
                }
            }
        } else {
            const gptr8 d = (gptr8)(jb.dst + drow * cs);
            const uint32_t nb = b1 - b0;
            for (uint32_t b = 4u * t32; b < nb; b += 128u) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(orow + b);
                if (b + 4u <= nb) *(gptr32u)(d + b) = v;
                else for (uint32_t k = 0; b + k < nb; ++k) d[b + k] = (uint8_t)(v >> (8u * k));
            }
        }
    };
    auto store_tile = [&](uint32_t mt) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t r0 = 0; r0 < 16u; r0 += kWtThreads / 32u) store_rows(mt, r0 + (tid >> 5));
    };

    WtMTile m = mts[it.mt0];
    request(m.kr0, m.kr0 + 32u * m.nk, m); // the band's first window
    commit();
    lds_barrier();
    for (uint32_t mt = it.mt0; mt < it.mt1; ++mt) {
        const bool more = mt + 1u < it.mt1;
        const WtMTile mn = mts[more ? mt + 1u : mt]; // (a scalar load: back long before the request below needs it)
        // the next step's rows and operands: requested first, in flight during both passes (a request behind the vertical pass
        // came back after the horizontal one had finished: 2 us of HBM latency against 1.5 us of work)
        if (more && !(ablate & 8u)) request(max(m.kr0 + 32u * m.nk, mn.kr0), mn.kr0 + 32u * mn.nk, mn);
        else { pre_r0 = pre_r1 = 0; wpre_n = 0; }
        if (mt > it.mt0 && !(ablate & 4u)) store_tile(mt - 1u);
        // ---- vertical: two column tiles side by side; one copy of the pass per K-step count (static loops: no partly defined
        // register arrays, which cost hipcc hundreds of spilled registers here) ----
        auto vertical = [&](auto nk_tag) __attribute__((always_inline)) {
            constexpr uint32_t NKV = decltype(nk_tag)::value;
            u32x4 wv[NKV][3]; // (eight waves read the same 3 KB per K-step: the largest LDS traffic of this pass)
#pragma unroll
            for (uint32_t k = 0; k < NKV; ++k)
#pragma unroll
                for (uint32_t t = 0; t < 3; ++t) wv[k][t] = wv_lds[(k * 3u + t) * 64u + lane];
            uint32_t radr[NKV]; // lane's address inside a column tile: 8 rows of a lane group, two 8-byte halves per row
#pragma unroll
            for (uint32_t k = 0; k < NKV; ++k) radr[k] = (rmod(m.kr0 + 32u * k + 8u * g) + (i >> 1)) * SP + 8u * (i & 1u);
            auto to_planes = [&](const f32x4 &acc, uint32_t ct) __attribute__((always_inline)) {
                // sums = value * 2^-9 -> 2^22 + round((value - 128) * 2^14) in the mantissa -> three byte planes (fl_mfma.hip, full width)
                const uint32_t x0 = __builtin_bit_cast(uint32_t, __builtin_fmaf(acc[0], 8388608.0f, 10485760.0f)), x1 = __builtin_bit_cast(uint32_t, __builtin_fmaf(acc[1], 8388608.0f, 10485760.0f)),
                               x2 = __builtin_bit_cast(uint32_t, __builtin_fmaf(acc[2], 8388608.0f, 10485760.0f)), x3 = __builtin_bit_cast(uint32_t, __builtin_fmaf(acc[3], 8388608.0f, 10485760.0f));
                const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x3, x2, 0x05010400u);
                const uint32_t p0 = __builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u;
                const uint32_t p1 = __builtin_amdgcn_perm(t1, t0, 0x07060302u) ^ 0x80808080u;
                const uint32_t p2 = __builtin_amdgcn_perm(x1, x0, 0x0c0c0602u) | __builtin_amdgcn_perm(x3, x2, 0x06020c0cu);
                uint8_t *pw = planes + i * SP + 16u * ct + 4u * g; // lane (output row i, columns 4 g .. 4 g + 3)
                *reinterpret_cast<uint32_t *>(pw) = p2;
                *reinterpret_cast<uint32_t *>(pw + 16u * SP) = p1;
                *reinterpret_cast<uint32_t *>(pw + 32u * SP) = p0;
            };
            auto to_f16 = [&](const v2i &raw) __attribute__((always_inline)) -> f16x8 {
                u32x4 a;
                a[0] = __builtin_amdgcn_perm(0u, (uint32_t)raw[0], 0x0c010c00u);
                a[1] = __builtin_amdgcn_perm(0u, (uint32_t)raw[0], 0x0c030c02u);
                a[2] = __builtin_amdgcn_perm(0u, (uint32_t)raw[1], 0x0c010c00u);
                a[3] = __builtin_amdgcn_perm(0u, (uint32_t)raw[1], 0x0c030c02u);
                return __builtin_bit_cast(f16x8, a);
            };
            uint32_t ct = wave;
            for (; ct + kWtWaves < spw; ct += 2u * kWtWaves) {
                const uint32_t ctb = ct + kWtWaves;
                f32x4 accA = {0.0f, 0.0f, 0.0f, 0.0f}, accB = {0.0f, 0.0f, 0.0f, 0.0f};
                v2i rawA[NKV], rawB[NKV];
#pragma unroll
                for (uint32_t k = 0; k < NKV; ++k) {
                    rawA[k] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + radr[k] + 16u * ct));
                    rawB[k] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + radr[k] + 16u * ctb));
                }
#pragma unroll
                for (uint32_t k = 0; k < NKV; ++k) {
                    const f16x8 av = to_f16(rawA[k]), bv = to_f16(rawB[k]);
#pragma unroll
                    for (uint32_t t = 0; t < 3; ++t) {
                        accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[k][t]), accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, __builtin_bit_cast(f16x8, wv[k][t]), accB, 0, 0, 0);
                    }
                }
                to_planes(accA, ct);
                to_planes(accB, ctb);
            }
            if (ct < spw) { // the wave's last, single column tile
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                v2i raw[NKV];
#pragma unroll
                for (uint32_t k = 0; k < NKV; ++k) raw[k] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + radr[k] + 16u * ct));
#pragma unroll
                for (uint32_t k = 0; k < NKV; ++k) {
                    const f16x8 av = to_f16(raw[k]);
#pragma unroll
                    for (uint32_t t = 0; t < 3; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[k][t]), acc, 0, 0, 0);
                }
                to_planes(acc, ct);
            }
        };
        if (!(ablate & 1u)) {
            if (m.nk == 1u) vertical(std::integral_constant<uint32_t, 1>{});
            else if (m.nk == 2u) vertical(std::integral_constant<uint32_t, 2>{});
            else if (m.nk == 3u) vertical(std::integral_constant<uint32_t, 3>{});
            else vertical(std::integral_constant<uint32_t, 4>{});
        }
        lds_barrier();
        // ---- horizontal: this wave's N-tiles w, w + 8, ... in pairs ----
        if (!(ablate & 2u)) {
            const uint32_t mine = wave < tn ? (tn - wave + kWtWaves - 1u) / kWtWaves : 0u; // tiles of this wave
#define FL_WT_PAIR(S_) \
            if constexpr (NSLOT > S_) { \
                constexpr int SB_ = (S_ + 1 < NSLOT) ? S_ + 1 : S_; \
                const uint32_t ja = wave + kWtWaves * S_##u, jb2 = ja + kWtWaves; \
                if (S_##u < mine) { \
                    const bool inB = S_##u + 1u < mine && SB_ != S_; /* a second tile in this pair of slots */ \
                    const bool okA = ntl[ja][2] == cached[S_], okB = inB && ntl[jb2][2] == cached[SB_]; \
                    if (okA) pair(std::integral_constant<int, S_>{}, std::integral_constant<int, SB_>{}, ja, jb2, okB); else fly(ja); \
                    if (inB && !(okA && okB)) fly(jb2); \
                } \
                __builtin_amdgcn_sched_barrier(0); /* one pair at a time: hoisting the next pairs' LDS reads up here costs more registers than the file has */ \
            }
            if constexpr (NSLOT > 1) {
                FL_WT_PAIR(0) FL_WT_PAIR(2) FL_WT_PAIR(4)
                for (uint32_t s = NSLOT; s < mine; ++s) fly(wave + kWtWaves * s); // (never: the planner gives a wave at most NSLOT tiles)
            } else {
                // one register set: every tile that shares it (a blur's interior columns) runs from registers, two at a time
                for (uint32_t s = 0; s < mine; s += 2u) {
                    const uint32_t ja = wave + kWtWaves * s, jb2 = ja + kWtWaves;
                    const bool okA = ntl[ja][2] == cached[0], inB = s + 1u < mine, okB = inB && ntl[jb2][2] == cached[0];
                    if (okA) pair(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ja, jb2, okB); else fly(ja);
                    if (inB && !(okA && okB)) fly(jb2);
                }
            }
#undef FL_WT_PAIR
        }
        commit();
        lds_barrier();
        m = mn;
    }
    if (!(ablate & 4u)) store_tile(it.mt1 - 1u);
}

template <int NSLOT, int NKMAX, int WAVES>
hipError_t launch_wtile_t(const LaunchWtile &m, hipStream_t st)
{
    // the attribute is per function and device: set once per (instantiation, device), not with every launch
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_wtile_kernel<NSLOT, NKMAX, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    resample_wtile_kernel<NSLOT, NKMAX, WAVES><<<m.nitems, 64 * WAVES, m.lds_bytes, st>>>(m.jobs, m.items, m.arena, m.letterbox, m.invert, m.framed);
    return hipGetLastError();
}

} // namespace

hipError_t launch_wtile(const LaunchWtile &m, hipStream_t st)
{
    if (!m.nitems) return hipSuccess;
    if (m.lds_bytes > 160u * 1024u) return hipErrorInvalidValue;
    if (m.framed && m.nslot != 1) return hipErrorInvalidValue; // (fl_batch.cpp asks for the framed source only with single-register-set plans)
    if (m.nslot == 6 && m.nkmax == 1) return launch_wtile_t<6, 1, 8>(m, st);
    if (m.nslot == 3 && m.nkmax == 2) return launch_wtile_t<3, 2, 8>(m, st);
    if (m.nslot == 2 && m.nkmax == 3) return launch_wtile_t<2, 3, 8>(m, st);
    // (single-register-set plans that leave room for a second workgroup on the CU run with 4 waves per workgroup)
    if (m.nslot == 1 && m.nkmax == 6) return (m.lds_bytes <= 78u * 1024u && m.half_waves) ? launch_wtile_t<1, 6, 4>(m, st) : launch_wtile_t<1, 6, 8>(m, st);
    return hipErrorInvalidValue;
}

} // namespace fl
