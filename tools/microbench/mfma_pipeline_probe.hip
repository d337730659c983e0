// Feasibility probe (development tool, not product): instruction skeleton of an LDS-staged, MFMA-based resample kernel on
// a 1080p Rgb8 batch, with dummy weights.  One workgroup (8 waves) owns a 2048-byte column strip of one picture and streams
// its rows once:
//   global_load_lds_dwordx4 (wave-private 2 x 8 KB ring, no VGPRs, no barriers) -> ds_read_b64_tr_b8 (hardware transpose)
//   -> v_perm to f16 (0x6400 | byte) -> v_mfma_f32_16x16x32_f16 x (2 live output tiles x 2 weight terms)
//   -> per finished 16-row tile: f32 -> 16-bit fixed point -> byte digits -> v_mfma_i32_16x16x64_i8 horizontal pass
//   -> ds_add_u32 into a shared [16][300] tile -> barrier -> RGBA8 stores.
//   hipcc --offload-arch=gfx950 -O3 mfma_pipeline_probe.hip -o mfma_pipeline_probe && ./mfma_pipeline_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 8, THREADS = WAVES * 64;
constexpr int KROWS = 32;                 // source rows per K-block
constexpr int WCOLS = 256;                // byte columns per wave
constexpr int RING = 2 * KROWS * WCOLS;   // bytes per wave
constexpr int NOUT = 300;                 // outputs (x, channel) per strip
constexpr int OUT_TILE = 16 * NOUT * 4;

extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

template <int MODE> // bit 0: skip vertical MFMAs, bit 1: skip the flush (horizontal stage), bit 2: skip tr reads + perms,
                    // bit 3: plain LDS stores instead of ds_add, bit 4: no barriers / final pass, bit 5: horizontal weights from registers,
                    // bit 6: skip the horizontal MFMAs, bit 7: one 8 KB buffer per wave instead of two
__global__ __launch_bounds__(THREADS, 1) void pipe(const uint8_t *__restrict__ src, const u32x4 *__restrict__ wtab, const u32x4 *__restrict__ htab,
                                                   uint32_t *__restrict__ dst, uint32_t pitch, uint32_t rows, uint32_t img_bytes, uint32_t nstrips,
                                                   uint32_t strip_stride, uint32_t nkb)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t img = blockIdx.x / nstrips, strip = blockIdx.x - img * nstrips;
    const uint8_t *base = src + (size_t)img * img_bytes + strip * strip_stride + wave * WCOLS;
    uint8_t *ring = lds + OUT_TILE + wave * RING;
    uint32_t *otile = reinterpret_cast<uint32_t *>(lds);
    for (uint32_t k = tid; k < 16 * NOUT; k += THREADS) otile[k] = 0;

    // load instruction u of a K-block: row octet u >> 1, column half u & 1; lane: tile t = lane >> 3, row q = lane & 7
    const uint32_t lq = lane & 7u, lt = lane >> 3;
    auto issue = [&](uint32_t s, uint32_t buf) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t ro = u >> 1, h = u & 1u;
            uint32_t row = s * KROWS + ro * 8u + lq;
            row = row < rows ? row : rows - 1u;
            const uint8_t *gp = base + (size_t)row * pitch + h * 128u + ((lt ^ (ro & 1u)) * 16u);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gp,
                                             (void __attribute__((address_space(3))) *)(ring + buf * (RING / 2) + u * 1024u), 16, 0, 0);
        }
    };
    f32x4 acc[2][16];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) acc[s2][ct] = f32x4{0, 0, 0, 0};

    const uint32_t traddr = (i >> 1) * 16u + (i & 1u) * 8u;
    uint32_t tile = 0, flushed = 0;
    issue(0, 0);
    if (!(MODE & 128)) issue(1, 1);
    for (uint32_t s = 0; s < nkb; ++s) {
        const uint32_t buf = (MODE & 128) ? 0u : (s & 1u);
        // weights of this K-block: 2 live tiles x 2 terms
        u32x4 wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = wtab[(s * 4 + k) * 64 + lane];
        if (s + 1 < nkb && !(MODE & 128)) __builtin_amdgcn_s_waitcnt(0x0f70 | 8 /* vmcnt(8): the next K-block's loads may stay in flight */);
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        v2i raw[16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const uint32_t off = buf * (RING / 2) + (2u * g + (ct >> 3)) * 1024u + (((ct & 7) ^ (g & 1u)) * 128u) + traddr;
            if (MODE & 4) raw[ct] = *reinterpret_cast<v2i *>(ring + off);
            else raw[ct] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + off));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        if (MODE & 128) { if (s + 1 < nkb) issue(s + 1, 0); }
        else if (s + 2 < nkb) issue(s + 2, buf);
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            u32x4 a;
            if (MODE & 4) { a[0] = raw[ct][0]; a[1] = raw[ct][1]; a[2] = raw[ct][0] ^ 1; a[3] = raw[ct][1] ^ 1; }
            else {
                a[0] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][0], 0x04010400u);
                a[1] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][0], 0x04030402u);
                a[2] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][1], 0x04010400u);
                a[3] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][1], 0x04030402u);
            }
            const f16x8 av = __builtin_bit_cast(f16x8, a);
            if (MODE & 1) { acc[0][ct][0] += (float)a[0]; acc[1][ct][1] += (float)a[1]; acc[0][ct][2] += (float)a[2]; acc[1][ct][3] += (float)a[3]; }
            else {
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[0]), acc[0][ct], 0, 0, 0);
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[1]), acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[2]), acc[1][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[3]), acc[1][ct], 0, 0, 0);
            }
        }
        // a 16-row output tile finishes every 3.2 K-blocks (102.4 source rows)
        const uint32_t done = ((s + 1u) * 10u) / 32u;
        if (!(MODE & 2) && done > flushed) {
            flushed = done;
            const uint32_t set = tile & 1u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x4 ahi, alo;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 v = set ? acc[1][4 * c + a] : acc[0][4 * c + a];
                    const uint32_t x0 = (uint32_t)(int)__builtin_rintf(__builtin_fmaf(v[0], 64.0f, -73728.0f)), x1 = (uint32_t)(int)__builtin_rintf(__builtin_fmaf(v[1], 64.0f, -73728.0f)),
                                   x2 = (uint32_t)(int)__builtin_rintf(__builtin_fmaf(v[2], 64.0f, -73728.0f)), x3 = (uint32_t)(int)__builtin_rintf(__builtin_fmaf(v[3], 64.0f, -73728.0f));
                    const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x3, x2, 0x05010400u); // (x0.b0, x1.b0, x0.b1, x1.b1)
                    alo[a] = __builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u;
                    ahi[a] = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (t == 2 && (c & 1)) continue; // 2.5 output tiles per chunk on average
                    const uint32_t hidx = ((wave * 4 + c) * 3 + t) * 2;
                    u32x4 h1, h0;
                    if (MODE & 32) { h1 = wv[t]; h0 = wv[t + 1]; } else { h1 = htab[hidx * 64 + lane]; h0 = htab[(hidx + 1) * 64 + lane]; }
                    i32x4 t2, t1, t0;
                    if (MODE & 64) { t2 = __builtin_bit_cast(i32x4, ahi) + __builtin_bit_cast(i32x4, h1); t1 = __builtin_bit_cast(i32x4, alo) ^ __builtin_bit_cast(i32x4, h0); t0 = t1 + t2; }
                    else {
                        t2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ahi), __builtin_bit_cast(i32x4, h1), i32x4{0, 0, 0, 0}, 0, 0, 0);
                        t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ahi), __builtin_bit_cast(i32x4, h0), i32x4{0, 0, 0, 0}, 0, 0, 0);
                        t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, alo), __builtin_bit_cast(i32x4, h1), t1, 0, 0, 0);
                        t0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, alo), __builtin_bit_cast(i32x4, h0), i32x4{0, 0, 0, 0}, 0, 0, 0);
                    }
                    const uint32_t ob = (wave * 36u + c * 9u + t * 4u + i) % NOUT;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t p = ((uint32_t)t2[r] << 16) + ((uint32_t)t1[r] << 8) + (uint32_t)t0[r];
                        if (MODE & 8) otile[(4u * g + r) * NOUT + ob] = p;
                        else __hip_atomic_fetch_add(&otile[(4u * g + r) * NOUT + ob], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) { if (set) acc[1][ct] = f32x4{0, 0, 0, 0}; else acc[0][ct] = f32x4{0, 0, 0, 0}; }
            if (!(MODE & 16)) {
            __syncthreads();
            for (uint32_t k = tid; k < 16 * 100; k += THREADS) {
                const uint32_t row = k / 100u, xo = k - row * 100u;
                uint32_t *o = otile + row * NOUT + 3u * xo;
                const uint32_t r = min(max((int)(o[0] + 0x40400000u) >> 23, 0), 255), gg = min(max((int)(o[1] + 0x40400000u) >> 23, 0), 255), b = min(max((int)(o[2] + 0x40400000u) >> 23, 0), 255);
                o[0] = 0; o[1] = 0; o[2] = 0;
                dst[((size_t)img * 200u + (tile * 16u + row) % 200u) * 300u + strip * 100u + xo] = r | (gg << 8) | (b << 16) | 0xff000000u;
            }
            __syncthreads();
            }
            ++tile;
        }
    }
    float keep = 0;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) keep += acc[0][ct][0] + acc[1][ct][1];
    if (keep == 123.456f) dst[blockIdx.x] = (uint32_t)keep;
}


// ---- second structure: the horizontal stage of a finished tile is staggered between the two halves of the workgroup
// (odd waves flush one K-block later), its weights are fetched one chunk ahead, the f32 -> fixed-point step is one FMA
// (magic constant), and the barrier + RGBA pass runs two K-blocks after the tile finished, at the top of the loop.
template <int STAG, int PRE, int FIN>
__global__ __launch_bounds__(THREADS, 1) void pipe2(const uint8_t *__restrict__ src, const u32x4 *__restrict__ wtab, const u32x4 *__restrict__ htab,
                                                    uint32_t *__restrict__ dst, uint32_t pitch, uint32_t rows, uint32_t img_bytes, uint32_t nstrips,
                                                    uint32_t strip_stride, uint32_t nkb)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t img = blockIdx.x / nstrips, strip = blockIdx.x - img * nstrips;
    const uint8_t *base = src + (size_t)img * img_bytes + strip * strip_stride + wave * WCOLS;
    uint8_t *ring = lds + OUT_TILE + wave * RING;
    uint32_t *otile = reinterpret_cast<uint32_t *>(lds);
    for (uint32_t k = tid; k < 16 * NOUT; k += THREADS) otile[k] = 0;
    __syncthreads();
    const uint32_t lq = lane & 7u, lt = lane >> 3;
    auto issue = [&](uint32_t s, uint32_t buf) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t ro = u >> 1, h = u & 1u;
            uint32_t row = s * KROWS + ro * 8u + lq;
            row = row < rows ? row : rows - 1u;
            const uint8_t *gp = base + (size_t)row * pitch + h * 128u + ((lt ^ (ro & 1u)) * 16u);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gp,
                                             (void __attribute__((address_space(3))) *)(ring + buf * (RING / 2) + u * 1024u), 16, 0, 0);
        }
    };
    f32x4 acc[2][16];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) acc[s2][ct] = f32x4{0, 0, 0, 0};
    const uint32_t traddr = (i >> 1) * 16u + (i & 1u) * 8u;
    const uint32_t stag = STAG ? (wave & 1u) : 0u;
    uint32_t flushed = 0, finalized = 0, need_b2 = 0;
    auto tiles_after = [](int s) -> uint32_t { return s < 0 ? 0u : ((uint32_t)(s + 1) * 10u) / 32u; };

    auto load_h = [&](u32x4 *h, int c) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t == 2 && (c & 1)) continue;
            const uint32_t hidx = ((wave * 4 + c) * 3 + t) * 2;
            h[2 * t] = htab[hidx * 64 + lane];
            h[2 * t + 1] = htab[(hidx + 1) * 64 + lane];
        }
    };
    auto flush = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
        u32x4 hb[4][6];
        if (PRE) load_h(hb[0], 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (PRE) { if (c + 1 < 4) load_h(hb[c + 1], c + 1); } else load_h(hb[c], c);
            u32x4 ahi, alo;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 v = acc[set][4 * c + a];
                // 1.5 * 2^23 + bias: the sum's low mantissa bits are the rounded fixed-point value
                const uint32_t x0 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[0], 64.0f, 12509184.0f)), x1 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[1], 64.0f, 12509184.0f)),
                               x2 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[2], 64.0f, 12509184.0f)), x3 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[3], 64.0f, 12509184.0f));
                const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x3, x2, 0x05010400u);
                alo[a] = __builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u;
                ahi[a] = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t == 2 && (c & 1)) continue;
                const u32x4 h1 = hb[c][2 * t], h0 = hb[c][2 * t + 1];
                i32x4 t2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ahi), __builtin_bit_cast(i32x4, h1), i32x4{0, 0, 0, 0}, 0, 0, 0);
                i32x4 t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ahi), __builtin_bit_cast(i32x4, h0), i32x4{0, 0, 0, 0}, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, alo), __builtin_bit_cast(i32x4, h1), t1, 0, 0, 0);
                i32x4 t0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, alo), __builtin_bit_cast(i32x4, h0), i32x4{0, 0, 0, 0}, 0, 0, 0);
                const uint32_t ob = (wave * 36u + c * 9u + t * 4u + i) % NOUT;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t p = ((uint32_t)t2[r] << 16) + ((uint32_t)t1[r] << 8) + (uint32_t)t0[r];
                    __hip_atomic_fetch_add(&otile[(4u * g + r) * NOUT + ob], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) acc[set][ct] = f32x4{0, 0, 0, 0};
    };
    auto final_pass = [&](uint32_t tile) {
        for (uint32_t k = tid; k < 16 * 100; k += THREADS) {
            const uint32_t row = k / 100u, xo = k - row * 100u;
            uint32_t *o = otile + row * NOUT + 3u * xo;
            const uint32_t r = min(max((int)(o[0] + 0x40400000u) >> 23, 0), 255), gg = min(max((int)(o[1] + 0x40400000u) >> 23, 0), 255), b = min(max((int)(o[2] + 0x40400000u) >> 23, 0), 255);
            o[0] = 0; o[1] = 0; o[2] = 0;
            dst[((size_t)img * 200u + (tile * 16u + row) % 200u) * 300u + strip * 100u + xo] = r | (gg << 8) | (b << 16) | 0xff000000u;
        }
    };

    issue(0, 0);
    issue(1, 1);
    for (uint32_t s = 0; s < nkb; ++s) {
        const uint32_t buf = s & 1u;
        if (FIN) {
            if (need_b2) { __syncthreads(); need_b2 = 0; }
            if (tiles_after((int)s - 2) > finalized) { __syncthreads(); final_pass(finalized); ++finalized; need_b2 = 1; }
        }
        u32x4 wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = wtab[(s * 4 + k) * 64 + lane];
        if (s + 1 < nkb) __builtin_amdgcn_s_waitcnt(0x0f70 | 8); else __builtin_amdgcn_s_waitcnt(0x0f70);
        v2i raw[16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const uint32_t off = buf * (RING / 2) + (2u * g + (ct >> 3)) * 1024u + (((ct & 7) ^ (g & 1u)) * 128u) + traddr;
            raw[ct] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + off));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (s + 2 < nkb) issue(s + 2, buf);
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            u32x4 a;
            a[0] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][0], 0x04010400u);
            a[1] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][0], 0x04030402u);
            a[2] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][1], 0x04010400u);
            a[3] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][1], 0x04030402u);
            const f16x8 av = __builtin_bit_cast(f16x8, a);
            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[0]), acc[0][ct], 0, 0, 0);
            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[1]), acc[0][ct], 0, 0, 0);
            acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[2]), acc[1][ct], 0, 0, 0);
            acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[3]), acc[1][ct], 0, 0, 0);
        }
        if (tiles_after((int)s - (int)stag) > flushed) {
            if (flushed & 1u) flush(std::integral_constant<int, 1>{}); else flush(std::integral_constant<int, 0>{});
            ++flushed;
        }
    }
    float keep = 0;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) keep += acc[0][ct][0] + acc[1][ct][1];
    if (keep == 123.456f) dst[blockIdx.x] = (uint32_t)keep;
}

template <int STAG, int PRE, int FIN>
static void run2(const char *name, const uint8_t *src, const u32x4 *wtab, const u32x4 *htab, uint32_t *dst, int nimg)
{
    const uint32_t W = 1920, H = 1080, pitch = W * 3, img_bytes = pitch * H, nstrips = 3, strip_stride = 1856, nkb = (H + KROWS - 1) / KROWS;
    const size_t ldsb = OUT_TILE + WAVES * RING;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe2<STAG, PRE, FIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    pipe2<STAG, PRE, FIN><<<nimg * nstrips, THREADS, ldsb>>>(src, wtab, htab, dst, pitch, H, img_bytes, nstrips, strip_stride, nkb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) pipe2<STAG, PRE, FIN><<<nimg * nstrips, THREADS, ldsb>>>(src, wtab, htab, dst, pitch, H, img_bytes, nstrips, strip_stride, nkb);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = (double)nimg * img_bytes / 1e9;
    printf("%-64s %.3f ms  (%.2f TB/s of source bytes)\n", name, ms, gb / ms);
}


// ---- third structure: one 8 KB buffer per wave (the next K-block is requested as soon as the transposed reads have
// left it), no workgroup barrier at all -- finished tiles are summed into one of two [16][300] LDS tiles and the wave
// that arrives last (an LDS counter) converts the tile to RGBA8 by itself -- and the horizontal weights come from an LDS
// copy when the geometry repeats (HLDS: 18 distinct operands for 1920 -> 300) or from global memory (HLDS = 0).
constexpr int P3_OT = 16 * NOUT * 4, P3_CNT = 2 * P3_OT, P3_HT = P3_CNT + 64, P3_RING = P3_HT + 18 * 1024;
template <int HLDS, int FIN>
__global__ __launch_bounds__(THREADS, 1) void pipe3(const uint8_t *__restrict__ src, const u32x4 *__restrict__ wtab, const u32x4 *__restrict__ htab,
                                                    uint32_t *__restrict__ dst, uint32_t pitch, uint32_t rows, uint32_t img_bytes, uint32_t nstrips,
                                                    uint32_t strip_stride, uint32_t nkb)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t img = blockIdx.x / nstrips, strip = blockIdx.x - img * nstrips;
    const uint8_t *base = src + (size_t)img * img_bytes + strip * strip_stride + wave * WCOLS;
    uint8_t *ring = lds + P3_RING + wave * (RING / 2);
    uint32_t *otile = reinterpret_cast<uint32_t *>(lds);
    uint32_t *counter = reinterpret_cast<uint32_t *>(lds + P3_CNT);
    const u32x4 *hlds = reinterpret_cast<const u32x4 *>(lds + P3_HT);
    for (uint32_t k = tid; k < 2 * 16 * NOUT + 16; k += THREADS) otile[k] = 0;
    for (uint32_t k = tid; k < 18 * 64; k += THREADS) reinterpret_cast<u32x4 *>(lds + P3_HT)[k] = htab[k];
    __syncthreads();
    const uint32_t lq = lane & 7u, lt = lane >> 3;
    auto issue = [&](uint32_t s) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t ro = u >> 1, h = u & 1u;
            uint32_t row = s * KROWS + ro * 8u + lq;
            row = row < rows ? row : rows - 1u;
            const uint8_t *gp = base + (size_t)row * pitch + h * 128u + ((lt ^ (ro & 1u)) * 16u);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gp, (void __attribute__((address_space(3))) *)(ring + u * 1024u), 16, 0, 0);
        }
    };
    f32x4 acc[2][16];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) acc[s2][ct] = f32x4{0, 0, 0, 0};
    const uint32_t traddr = (i >> 1) * 16u + (i & 1u) * 8u;
    uint32_t tile = 0, flushed = 0;
    issue(0);
    for (uint32_t s = 0; s < nkb; ++s) {
        u32x4 wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = wtab[(s * 4 + k) * 64 + lane];
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
        v2i raw[16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const uint32_t off = (2u * g + (ct >> 3)) * 1024u + (((ct & 7) ^ (g & 1u)) * 128u) + traddr;
            raw[ct] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + off));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        if (s + 1 < nkb) issue(s + 1);
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            u32x4 a;
            a[0] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][0], 0x04010400u);
            a[1] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][0], 0x04030402u);
            a[2] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][1], 0x04010400u);
            a[3] = __builtin_amdgcn_perm(0x64646464u, (uint32_t)raw[ct][1], 0x04030402u);
            const f16x8 av = __builtin_bit_cast(f16x8, a);
            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[0]), acc[0][ct], 0, 0, 0);
            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[1]), acc[0][ct], 0, 0, 0);
            acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[2]), acc[1][ct], 0, 0, 0);
            acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[3]), acc[1][ct], 0, 0, 0);
        }
        const uint32_t done = ((s + 1u) * 10u) / 32u;
        if (done > flushed) {
            flushed = done;
            const uint32_t set = tile & 1u;
            uint32_t *ot = otile + set * (16 * NOUT);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x4 ahi, alo;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 v = set ? acc[1][4 * c + a] : acc[0][4 * c + a];
                    const uint32_t x0 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[0], 64.0f, 12509184.0f)), x1 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[1], 64.0f, 12509184.0f)),
                                   x2 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[2], 64.0f, 12509184.0f)), x3 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[3], 64.0f, 12509184.0f));
                    const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x3, x2, 0x05010400u);
                    alo[a] = __builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u;
                    ahi[a] = __builtin_amdgcn_perm(t1, t0, 0x07060302u);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (t == 2 && (c & 1)) continue;
                    u32x4 h1, h0;
                    if (HLDS) { const uint32_t hidx = (((wave * 4 + c) % 3u) * 3 + t) * 2; h1 = hlds[hidx * 64 + lane]; h0 = hlds[(hidx + 1) * 64 + lane]; }
                    else { const uint32_t hidx = ((wave * 4 + c) * 3 + t) * 2; h1 = htab[hidx * 64 + lane]; h0 = htab[(hidx + 1) * 64 + lane]; }
                    i32x4 t2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ahi), __builtin_bit_cast(i32x4, h1), i32x4{0, 0, 0, 0}, 0, 0, 0);
                    i32x4 t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ahi), __builtin_bit_cast(i32x4, h0), i32x4{0, 0, 0, 0}, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, alo), __builtin_bit_cast(i32x4, h1), t1, 0, 0, 0);
                    i32x4 t0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, alo), __builtin_bit_cast(i32x4, h0), i32x4{0, 0, 0, 0}, 0, 0, 0);
                    const uint32_t ob = (wave * 36u + c * 9u + t * 4u + i) % NOUT;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t p = ((uint32_t)t2[r] << 16) + ((uint32_t)t1[r] << 8) + (uint32_t)t0[r];
                        __hip_atomic_fetch_add(&ot[(4u * g + r) * NOUT + ob], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) { if (set) acc[1][ct] = f32x4{0, 0, 0, 0}; else acc[0][ct] = f32x4{0, 0, 0, 0}; }
            if (FIN) {
                uint32_t old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&counter[set], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                if ((old & 7u) == 7u) { // the last of the 8 waves: this tile is complete
                    for (uint32_t k = lane; k < 16 * 100; k += 64) {
                        const uint32_t row = k / 100u, xo = k - row * 100u;
                        uint32_t *o = ot + row * NOUT + 3u * xo;
                        const uint32_t r = min(max((int)(o[0] + 0x40400000u) >> 23, 0), 255), gg = min(max((int)(o[1] + 0x40400000u) >> 23, 0), 255), b = min(max((int)(o[2] + 0x40400000u) >> 23, 0), 255);
                        o[0] = 0; o[1] = 0; o[2] = 0;
                        dst[((size_t)img * 200u + (tile * 16u + row) % 200u) * 300u + strip * 100u + xo] = r | (gg << 8) | (b << 16) | 0xff000000u;
                    }
                }
            }
            ++tile;
        }
    }
    float keep = 0;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) keep += acc[0][ct][0] + acc[1][ct][1];
    if (keep == 123.456f) dst[blockIdx.x] = (uint32_t)keep;
}

template <int HLDS, int FIN>
static void run3(const char *name, const uint8_t *src, const u32x4 *wtab, const u32x4 *htab, uint32_t *dst, int nimg)
{
    const uint32_t W = 1920, H = 1080, pitch = W * 3, img_bytes = pitch * H, nstrips = 3, strip_stride = 1856, nkb = (H + KROWS - 1) / KROWS;
    const size_t ldsb = P3_RING + WAVES * (RING / 2);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe3<HLDS, FIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    pipe3<HLDS, FIN><<<nimg * nstrips, THREADS, ldsb>>>(src, wtab, htab, dst, pitch, H, img_bytes, nstrips, strip_stride, nkb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) pipe3<HLDS, FIN><<<nimg * nstrips, THREADS, ldsb>>>(src, wtab, htab, dst, pitch, H, img_bytes, nstrips, strip_stride, nkb);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = (double)nimg * img_bytes / 1e9;
    printf("%-64s %.3f ms  (%.2f TB/s of source bytes; lds %zu)\n", name, ms, gb / ms, ldsb);
}

template <int MODE>
static void run(const char *name, const uint8_t *src, const u32x4 *wtab, const u32x4 *htab, uint32_t *dst, int nimg)
{
    const uint32_t W = 1920, H = 1080, pitch = W * 3, img_bytes = pitch * H, nstrips = 3, strip_stride = 1856, nkb = (H + KROWS - 1) / KROWS;
    const size_t ldsb = OUT_TILE + WAVES * RING;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pipe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    pipe<MODE><<<nimg * nstrips, THREADS, ldsb>>>(src, wtab, htab, dst, pitch, H, img_bytes, nstrips, strip_stride, nkb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) pipe<MODE><<<nimg * nstrips, THREADS, ldsb>>>(src, wtab, htab, dst, pitch, H, img_bytes, nstrips, strip_stride, nkb);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = (double)nimg * img_bytes / 1e9;
    printf("%-64s %.3f ms  (%.2f TB/s of source bytes; 3 strips of 2048 B at stride %u = %.3fx bytes; lds %zu)\n", name, ms, gb / ms, strip_stride, 3 * 2048.0 / pitch, ldsb);
}

int main()
{
    const int nimg = 1024;
    const size_t bytes = (size_t)nimg * 1920 * 1080 * 3;
    uint8_t *src;
    CK(hipMalloc(&src, bytes + 8192));
    {
        std::vector<uint32_t> h(1 << 20);
        for (auto &v : h) v = (uint32_t)rand() * 2654435761u;
        for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy(src + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice));
    }
    u32x4 *wtab, *htab;
    {
        std::vector<uint32_t> h(34 * 4 * 64 * 4);
        for (auto &v : h) v = 0x2c002c00u + (uint32_t)(rand() & 0x03ff03ff); // small f16 values
        CK(hipMalloc(&wtab, h.size() * 4));
        CK(hipMemcpy(wtab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> h2(8 * 4 * 3 * 2 * 64 * 4);
        for (auto &v : h2) v = (uint32_t)rand();
        CK(hipMalloc(&htab, h2.size() * 4));
        CK(hipMemcpy(htab, h2.data(), h2.size() * 4, hipMemcpyHostToDevice));
    }
    uint32_t *dst;
    CK(hipMalloc(&dst, (size_t)nimg * 300 * 200 * 4));
    run<7>("global_load_lds ring only (plain LDS reads, no MFMA, no flush)", src, wtab, htab, dst, nimg);
    run<3>("+ ds_read_b64_tr_b8 + f16 perms", src, wtab, htab, dst, nimg);
    run<2>("+ vertical f16 MFMAs (64 per K-block and wave)", src, wtab, htab, dst, nimg);
    run<0>("+ horizontal stage (i8 MFMA, ds_add, barrier, RGBA stores)", src, wtab, htab, dst, nimg);
    run<8>("  horizontal stage with plain LDS stores instead of ds_add", src, wtab, htab, dst, nimg);
    run<16>("  horizontal stage without barriers / final pass", src, wtab, htab, dst, nimg);
    run<32>("  horizontal stage with weights from registers", src, wtab, htab, dst, nimg);
    run<64>("  horizontal stage without its MFMAs", src, wtab, htab, dst, nimg);
    run<8 + 16 + 32>("  horizontal stage: conversion + MFMA + stores only", src, wtab, htab, dst, nimg);
    run<8 + 16 + 32 + 64>("  horizontal stage: conversion + stores only", src, wtab, htab, dst, nimg);
    run<7 + 128>("one 8 KB buffer per wave: ring only", src, wtab, htab, dst, nimg);
    run<2 + 128>("one 8 KB buffer per wave: + tr reads, perms, vertical MFMAs", src, wtab, htab, dst, nimg);
    run<128>("one 8 KB buffer per wave: + horizontal stage", src, wtab, htab, dst, nimg);
    run3<0, 1>("pipe3: no barriers, last wave converts; weights from global", src, wtab, htab, dst, nimg);
    run3<1, 1>("pipe3: no barriers, last wave converts; weights from LDS", src, wtab, htab, dst, nimg);
    run3<1, 0>("pipe3: weights from LDS, no RGBA pass at all", src, wtab, htab, dst, nimg);
    run2<0, 0, 1>("pipe2: deferred barrier + RGBA pass, magic-constant conversion", src, wtab, htab, dst, nimg);
    run2<0, 1, 1>("pipe2: + weights one chunk ahead", src, wtab, htab, dst, nimg);
    run2<1, 0, 1>("pipe2: + odd waves flush one K-block later", src, wtab, htab, dst, nimg);
    run2<1, 1, 1>("pipe2: + both", src, wtab, htab, dst, nimg);
    run2<1, 1, 0>("pipe2: both, no barrier / RGBA pass at all", src, wtab, htab, dst, nimg);
    return 0;
}
