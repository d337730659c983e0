// Feasibility probe (development tool, not product): the instruction skeleton of an MFMA-based resample kernel,
// with dummy weights, timed on a 1080p Rgb8 batch.  Answers "how fast can source rows stream through
//   buffer_load -> xor 0x80 -> 4x4 byte transposes (v_perm) -> v_mfma_i32_16x16x64_i8 x 3 weight digits
//   -> digit combine -> f32 -> LDS tile [16 output rows][chunk columns] -> v_mfma_f32_16x16x4_f32 horizontal"
// before any of it is built for real.
//   hipcc --offload-arch=gfx950 -O3 vstage_probe.hip -o vstage_probe && ./vstage_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;
constexpr int CHUNK_B = WAVES * 128;  // byte columns per workgroup chunk
constexpr int XPITCH = CHUNK_B + 4;   // floats per X-tile row (bank spread for the b128 writes)
constexpr int NKB = 3;                // K-blocks (64 source rows each) per 16-row output tile

__device__ __forceinline__ void transpose4x4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t *o)
{
    const uint32_t u0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
    const uint32_t u1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
    const uint32_t u2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
    const uint32_t u3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(u2, u0, 0x05040100u);
    o[1] = __builtin_amdgcn_perm(u2, u0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(u3, u1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(u3, u1, 0x07060302u);
}

extern __shared__ __attribute__((aligned(16))) float lds[];

// MODE bit 0: horizontal f32-MFMA emulation; bit 1: skip the vertical MFMAs (loads + VALU only); bit 2: skip transposes
template <int MODE>
__global__ __launch_bounds__(THREADS, 2) void vstage(const uint8_t *__restrict__ src, const i32x4 *__restrict__ wtab, float *__restrict__ out,
                                                     uint32_t pitch, uint32_t img_bytes, uint32_t nchunks, uint32_t ntiles, uint32_t hm_per_wave)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t img = blockIdx.x / nchunks, chunk = blockIdx.x - img * nchunks;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)img * img_bytes, 0, (int)img_bytes, 0x00020000);
    const uint32_t voff = (16u * g) * pitch + chunk * CHUNK_B + wave * 128u + 8u * i; // row 16g of the K-block, this lane's 8 bytes
    float *xt = lds;                              // [16][XPITCH]
    float *htab = lds + 16 * XPITCH;              // 4096 floats of dummy horizontal weights
    for (uint32_t k = tid; k < 4096; k += THREADS) htab[k] = 1.0f / (float)(k + 1);

    const uint32_t nsteps = ntiles * NKB;
    auto row0 = [&](uint32_t s) { const uint32_t t = s / NKB, kb = s - t * NKB; const uint32_t r = 102u * t; return (r > 19u ? r - 19u : 0u) + 64u * kb; };

    u32x2 ra[16], rb[16];
    i32x4 wa[3], wb[3];
    auto issue = [&](u32x2 *r, i32x4 *w, uint32_t s) {
#pragma unroll
        for (int d = 0; d < 3; ++d) w[d] = wtab[(s * 3 + d) * 64 + lane];
        const uint32_t soff = row0(s) * pitch;
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + j * pitch, soff, 0);
    };
    i32x4 acc[8][3];
    auto zero_acc = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[q][d] = i32x4{0, 0, 0, 0};
    };
    auto process = [&](u32x2 *r, i32x4 *w) {
        uint32_t op[8][4]; // operand q = byte column q of this lane's 8: 16 rows as 4 dwords
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t o[4];
            if (MODE & 4) { o[0] = r[4 * b].x; o[1] = r[4 * b + 1].x; o[2] = r[4 * b + 2].x; o[3] = r[4 * b + 3].x; }
            else transpose4x4(r[4 * b].x ^ 0x80808080u, r[4 * b + 1].x ^ 0x80808080u, r[4 * b + 2].x ^ 0x80808080u, r[4 * b + 3].x ^ 0x80808080u, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) op[q][b] = o[q];
            if (MODE & 4) { o[0] = r[4 * b].y; o[1] = r[4 * b + 1].y; o[2] = r[4 * b + 2].y; o[3] = r[4 * b + 3].y; }
            else transpose4x4(r[4 * b].y ^ 0x80808080u, r[4 * b + 1].y ^ 0x80808080u, r[4 * b + 2].y ^ 0x80808080u, r[4 * b + 3].y ^ 0x80808080u, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) op[4 + q][b] = o[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const i32x4 a = {(int)op[q][0], (int)op[q][1], (int)op[q][2], (int)op[q][3]};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (MODE & 2) { acc[q][d].x += a.x ^ w[d].x; acc[q][d].y += a.y; acc[q][d].z += a.z; acc[q][d].w += a.w; }
                else acc[q][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w[d], acc[q][d], 0, 0, 0);
            }
        }
    };
    f32x4 hacc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto finish_tile = [&](uint32_t t) {
        // digit combine -> f32 -> X tile.  Result element (m = 4g + reg, n = i): m = byte-column lane 4g+reg of the operand, n = output row
        const float bias = 128.0f + (float)(i + t);
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                f32x4 xv;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const i32x4 a0 = acc[q4 * 4 + q][0], a1 = acc[q4 * 4 + q][1], a2 = acc[q4 * 4 + q][2];
                    const int v0 = reg == 0 ? a0.x : reg == 1 ? a0.y : reg == 2 ? a0.z : a0.w;
                    const int v1 = reg == 0 ? a1.x : reg == 1 ? a1.y : reg == 2 ? a1.z : a1.w;
                    const int v2 = reg == 0 ? a2.x : reg == 1 ? a2.y : reg == 2 ? a2.z : a2.w;
                    const int tsum = v0 + (v1 << 8) + (v2 << 16);
                    const float x = __builtin_fmaf((float)tsum, 2.384185791015625e-07f, bias);
                    if (q == 0) xv.x = x; else if (q == 1) xv.y = x; else if (q == 2) xv.z = x; else xv.w = x;
                }
                // columns 8*(4g+reg) + 4*q4 + q of this wave's 128
                *reinterpret_cast<f32x4 *>(xt + i * XPITCH + wave * 128u + 8u * (4u * g + reg) + 4u * q4) = xv;
            }
        }
        __syncthreads();
        if (MODE & 1) {
            // horizontal emulation: hm_per_wave f32 MFMAs, each with one X read (A) and one gathered weight (B)
            for (uint32_t k = 0; k < hm_per_wave; ++k) {
                const uint32_t px = (wave * hm_per_wave + k) * 4u + g;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = xt[i * XPITCH + ((px * 3u + c) & (CHUNK_B - 1))];
                    uint32_t idx = px - 6u * i;
                    const float b = idx < 40u ? htab[(i * 41u + idx) & 4095u] : 0.0f;
                    hacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, hacc[c], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    };

    issue(ra, wa, 0);
    zero_acc();
    for (uint32_t s = 0; s < nsteps; s += 2) {
        issue(rb, wb, s + 1 < nsteps ? s + 1 : s);
        process(ra, wa);
        if ((s + 1) % NKB == 0) { finish_tile(s / NKB); zero_acc(); }
        issue(ra, wa, s + 2 < nsteps ? s + 2 : s);
        if (s + 1 < nsteps) {
            process(rb, wb);
            if ((s + 2) % NKB == 0) { finish_tile((s + 1) / NKB); zero_acc(); }
        }
    }
    float keep = hacc[0].x + hacc[1].y + hacc[2].z + xt[tid];
    if (keep == 123.456f) out[blockIdx.x] = keep;
}

template <int MODE>
static void run(const char *name, const uint8_t *src, const i32x4 *wtab, float *out, int nimg, uint32_t hm)
{
    const uint32_t W = 1920, H = 1080, C = 3, pitch = W * C, img_bytes = pitch * H;
    const uint32_t nchunks = (pitch + CHUNK_B - 1) / CHUNK_B, ntiles = 11;
    const size_t ldsb = (16 * XPITCH + 4096) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&vstage<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    vstage<MODE><<<nimg * nchunks, THREADS, ldsb>>>(src, wtab, out, pitch, img_bytes, nchunks, ntiles, hm);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) vstage<MODE><<<nimg * nchunks, THREADS, ldsb>>>(src, wtab, out, pitch, img_bytes, nchunks, ntiles, hm);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = (double)nimg * img_bytes / 1e9;
    printf("%-58s %.3f ms for %d images  (%.2f TB/s of source bytes; %u chunks/image x %u tiles)\n", name, ms, nimg, gb / ms, nchunks, ntiles);
}


// ---- sliding-window variant: every source row is loaded once; K-blocks of KROWS rows complete S output rows each; the
// partial sums of the other 16 - S rows are carried into the next K-block as the C operand of the digit-0 MFMA
// (shifted S lanes down by DPP); completed rows go to the X tile, the horizontal stage runs when it holds 16 rows.
template <int LB, int MODE>
__global__ __launch_bounds__(THREADS, 2) void vslide(const uint8_t *__restrict__ src, const i32x4 *__restrict__ wtab, float *__restrict__ out,
                                                     uint32_t pitch, uint32_t rows, uint32_t img_bytes, uint32_t nchunks, uint32_t chunk_stride, uint32_t nkb,
                                                     uint32_t krows, uint32_t hm_per_wave)
{
    constexpr int LW = LB / 4;       // dwords per lane and row
    constexpr int NOP = LB;          // MFMA operands (byte columns) per lane
    constexpr int WB = 16 * LB;      // byte columns per wave
    constexpr int CW = WAVES * WB;   // byte columns per workgroup
    constexpr int XP = CW + 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t img = blockIdx.x / nchunks, chunk = blockIdx.x - img * nchunks;
    const uint8_t *base = src + (size_t)img * img_bytes;
    const uint32_t voff = (16u * g) * pitch + chunk * chunk_stride + wave * WB + LB * i;
    float *xt = lds;
    float *htab = lds + 16 * XP;
    for (uint32_t k = tid; k < 4096; k += THREADS) htab[k] = 1.0f / (float)(k + 1);

    uint32_t ra[16][LW], rb[16][LW];
    i32x4 wa[3], wb[3];
    auto issue = [&](uint32_t (*r)[LW], i32x4 *w, uint32_t s) {
#pragma unroll
        for (int d = 0; d < 3; ++d) w[d] = wtab[(s * 3 + d) * 64 + lane];
        // rows past this K-block read 0 without touching memory: the descriptor ends where the K-block ends
        const uint32_t r0 = s * krows, r1 = min(r0 + krows, rows);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(r1 * pitch), 0x00020000);
        const uint32_t soff = r0 * pitch;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (LW == 2) { auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + j * pitch, soff, 0); r[j][0] = v[0]; r[j][1] = v[1]; }
            if constexpr (LW == 3) { auto v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff + j * pitch, soff, 0); r[j][0] = v[0]; r[j][1] = v[1]; r[j][2] = v[2]; }
        }
    };
    i32x4 carry[NOP];
#pragma unroll
    for (int q = 0; q < NOP; ++q) carry[q] = i32x4{0, 0, 0, 0};
    f32x4 hacc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint32_t fill = 0;
    auto process = [&](uint32_t (*r)[LW], i32x4 *w, uint32_t s) {
        const float bias = 128.0f + (float)(i + s);
#pragma unroll
        for (int dw = 0; dw < LW; ++dw) {
            uint32_t op[4][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                uint32_t o[4];
                transpose4x4(r[4 * b][dw] ^ 0x80808080u, r[4 * b + 1][dw] ^ 0x80808080u, r[4 * b + 2][dw] ^ 0x80808080u, r[4 * b + 3][dw] ^ 0x80808080u, o);
#pragma unroll
                for (int q = 0; q < 4; ++q) op[q][b] = o[q];
            }
            f32x4 xv[4]; // [reg] over q
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i32x4 a = {(int)op[q][0], (int)op[q][1], (int)op[q][2], (int)op[q][3]};
                i32x4 a0, a1, a2;
                if (MODE & 2) { a0 = carry[dw * 4 + q] + a; a1 = a ^ w[1]; a2 = a ^ w[2]; }
                else {
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w[0], carry[dw * 4 + q], 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w[1], i32x4{0, 0, 0, 0}, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w[2], i32x4{0, 0, 0, 0}, 0, 0, 0);
                }
                i32x4 t = a0 + (a1 << 8) + (a2 << 16);
                // completed rows (lanes i < 8) -> f32
                const float x0 = __builtin_fmaf((float)t.x, 2.384185791015625e-07f, bias), x1 = __builtin_fmaf((float)t.y, 2.384185791015625e-07f, bias),
                            x2 = __builtin_fmaf((float)t.z, 2.384185791015625e-07f, bias), x3 = __builtin_fmaf((float)t.w, 2.384185791015625e-07f, bias);
                if (q == 0) { xv[0].x = x0; xv[1].x = x1; xv[2].x = x2; xv[3].x = x3; }
                if (q == 1) { xv[0].y = x0; xv[1].y = x1; xv[2].y = x2; xv[3].y = x3; }
                if (q == 2) { xv[0].z = x0; xv[1].z = x1; xv[2].z = x2; xv[3].z = x3; }
                if (q == 3) { xv[0].w = x0; xv[1].w = x1; xv[2].w = x2; xv[3].w = x3; }
                // carry: lane i takes lane i + 8 of the same 16-lane row, rows shifted in are zero
                carry[dw * 4 + q].x = __builtin_amdgcn_update_dpp(0, t.x, 0x108, 0xf, 0xf, true);
                carry[dw * 4 + q].y = __builtin_amdgcn_update_dpp(0, t.y, 0x108, 0xf, 0xf, true);
                carry[dw * 4 + q].z = __builtin_amdgcn_update_dpp(0, t.z, 0x108, 0xf, 0xf, true);
                carry[dw * 4 + q].w = __builtin_amdgcn_update_dpp(0, t.w, 0x108, 0xf, 0xf, true);
            }
            if (i < 8u) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    *reinterpret_cast<f32x4 *>(xt + (fill + i) * XP + wave * WB + LB * (4u * g + reg) + 4u * dw) = xv[reg];
            }
        }
        fill += 8u;
        if (fill == 16u) {
            fill = 0;
            __syncthreads();
            if (MODE & 1) {
                for (uint32_t k = 0; k < hm_per_wave; ++k) {
                    const uint32_t px = (wave * hm_per_wave + k) * 4u + g;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float a = xt[i * XP + ((px * 3u + c) % CW)];
                        uint32_t idx = px - 6u * i;
                        const float b = idx < 40u ? htab[(i * 41u + idx) & 4095u] : 0.0f;
                        hacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, hacc[c], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    };
    issue(ra, wa, 0);
    for (uint32_t s = 0; s < nkb; s += 2) {
        issue(rb, wb, s + 1 < nkb ? s + 1 : s);
        process(ra, wa, s);
        issue(ra, wa, s + 2 < nkb ? s + 2 : s);
        if (s + 1 < nkb) process(rb, wb, s + 1);
    }
    float keep = hacc[0].x + hacc[1].y + hacc[2].z + xt[tid] + (float)carry[0].x;
    if (keep == 123.456f) out[blockIdx.x] = keep;
}

template <int LB, int MODE>
static void run_slide(const char *name, const uint8_t *src, const i32x4 *wtab, float *out, int nimg, uint32_t nchunks, uint32_t hm)
{
    const uint32_t W = 1920, H = 1080, C = 3, pitch = W * C, img_bytes = pitch * H;
    const uint32_t krows = 51, nkb = (H + krows - 1) / krows;
    const uint32_t cw = WAVES * 16 * LB, chunk_stride = (pitch - cw + nchunks - 2) / (nchunks - 1) / 12 * 12;
    const size_t ldsb = (16 * (cw + 4) + 4096) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&vslide<LB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    vslide<LB, MODE><<<nimg * nchunks, THREADS, ldsb>>>(src, wtab, out, pitch, H, img_bytes, nchunks, chunk_stride, nkb, krows, hm);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) vslide<LB, MODE><<<nimg * nchunks, THREADS, ldsb>>>(src, wtab, out, pitch, H, img_bytes, nchunks, chunk_stride, nkb, krows, hm);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = (double)nimg * img_bytes / 1e9;
    printf("%-66s %.3f ms  (%.2f TB/s of source bytes; %u strips of %u B, stride %u, %u K-blocks, lds %zu)\n", name, ms, gb / ms, nchunks, cw, chunk_stride, nkb, ldsb);
}

int main()
{
    const int nimg = 1024;
    const size_t bytes = (size_t)nimg * 1920 * 1080 * 3;
    uint8_t *src;
    CK(hipMalloc(&src, bytes + 4096));
    {
        std::vector<uint32_t> h(1 << 20);
        for (auto &v : h) v = (uint32_t)rand() * 2654435761u;
        for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy(src + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice));
    }
    i32x4 *wtab;
    const size_t wn = (size_t)11 * NKB * 3 * 64;
    {
        std::vector<int> h(wn * 4);
        for (auto &v : h) v = rand();
        CK(hipMalloc(&wtab, wn * 16));
        CK(hipMemcpy(wtab, h.data(), wn * 16, hipMemcpyHostToDevice));
    }
    float *out;
    CK(hipMalloc(&out, 1 << 20));
    run<6>("loads + combine + X-tile writes only", src, wtab, out, nimg, 0);
    run<2>("+ xor + transposes (no MFMA)", src, wtab, out, nimg, 0);
    run<0>("vertical stage: + 72 i8 MFMAs per wave-tile", src, wtab, out, nimg, 0);
    run<1>("vertical + horizontal emulation (15 x 3 f32 MFMA / wave-tile)", src, wtab, out, nimg, 15);
    run<1>("vertical + horizontal emulation (8 x 3 f32 MFMA / wave-tile)", src, wtab, out, nimg, 8);
    run_slide<8, 2>("slide LB=8: loads + VALU (no MFMA), 7 strips", src, wtab, out, nimg, 7, 0);
    run_slide<8, 0>("slide LB=8: vertical stage, 7 strips", src, wtab, out, nimg, 7, 0);
    run_slide<8, 1>("slide LB=8: vertical + horizontal emu (15x3), 7 strips", src, wtab, out, nimg, 7, 15);
    run_slide<12, 2>("slide LB=12: loads + VALU (no MFMA), 5 strips", src, wtab, out, nimg, 5, 0);
    run_slide<12, 0>("slide LB=12: vertical stage, 5 strips", src, wtab, out, nimg, 5, 0);
    run_slide<12, 1>("slide LB=12: vertical + horizontal emu (21x3), 5 strips", src, wtab, out, nimg, 5, 21);
    run_slide<12, 1>("slide LB=12: vertical + horizontal emu (10x3), 5 strips", src, wtab, out, nimg, 5, 10);
    return 0;
}
