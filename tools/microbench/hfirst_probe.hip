// Feasibility probe (development tool, not product): horizontal-FIRST resample skeleton with dummy weights.
//   source rows in their natural layout (lane (g, n): 16 bytes of row n) -> xor 0x80 -> v_mfma_i32_16x16x64_i8 against
//   register-resident weight operands (3 signed-byte digits) -> digit combine -> ds_add into a [16 rows][outputs] i32 tile
//   -> vertical pass on 6.4x fewer columns (VALU, from LDS).
// No transposes, small prefetch ring, tiny LDS tile.  Timed on a 1080p Rgb8 batch like vstage_probe.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NPAIR = 6;   // (K-block, M-tile) pairs with a non-zero weight operand per wave (2 K-blocks x ~2.75 M-tiles)
constexpr int NMT = 4;     // distinct M-tiles (16 outputs each) a wave contributes to
constexpr int TPITCH = 193; // i32 per tile row (180 outputs of a strip, odd pitch)

extern __shared__ __attribute__((aligned(16))) int lds_i[];

// MODE bit 0: vertical-second emulation; bit 1: no MFMA (VALU stand-in); bit 2: no ds_add
template <int WAVES, int MODE, int DEPTH = 3>
__global__ __launch_bounds__(WAVES * 64) void hfirst(const uint8_t *__restrict__ src, const i32x4 *__restrict__ wtab, float *__restrict__ out,
                                                     uint32_t pitch, uint32_t rows, uint32_t img_bytes, uint32_t nstrips, uint32_t strip_stride)
{
    constexpr uint32_t T = WAVES * 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, n = lane & 15u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t img = blockIdx.x / nstrips, strip = blockIdx.x - img * nstrips;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)img * img_bytes, 0, (int)img_bytes, 0x00020000);
    const uint32_t voff = n * pitch + strip * strip_stride + wave * 128u + 16u * g; // row n of the step, this lane's 16 bytes of K-block 0
    int *tile = lds_i; // 2 x [16][TPITCH]
    for (uint32_t k = tid; k < 2 * 16 * TPITCH; k += T) tile[k] = 0;

    // weight operands: loaded once, resident for the whole image
    i32x4 wa[NPAIR][3];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
        for (int d = 0; d < 3; ++d) wa[p][d] = wtab[((wave * NPAIR + p) * 3 + d) * 64 + lane];
    // which M-tile each pair feeds and which K-block it reads (static pattern: kb0 -> tiles 0,1,2; kb1 -> tiles 1,2,3)
    const uint32_t nsteps = (rows + 15u) / 16u;
    u32x4 ring[DEPTH][2];
    auto issue = [&](int slot, uint32_t s) {
        const uint32_t soff = s * 16u * pitch;
        ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 64u, soff, 0);
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) issue(k, k < (int)nsteps ? k : 0);
    __syncthreads();
    f32x2 vacc[7] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    auto step = [&](int slot, uint32_t s) {
        i32x4 b[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32x4 r = ring[slot][k];
            b[k] = i32x4{(int)(r.x ^ 0x80808080u), (int)(r.y ^ 0x80808080u), (int)(r.z ^ 0x80808080u), (int)(r.w ^ 0x80808080u)};
        }
        issue(slot, s + DEPTH < nsteps ? s + DEPTH : s);
        int *tl = tile + (s & 1u) * 16 * TPITCH;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            i32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
                const int kb = p / 3, pmt = (p % 3) + kb; // pairs 0..2: K-block 0 -> tiles 0..2, pairs 3..5: K-block 1 -> tiles 1..3
                if (pmt != mt) continue;
                if (MODE & 2) { a0 += wa[p][0] ^ b[kb]; a1 += wa[p][1]; a2 += wa[p][2]; }
                else {
                    a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[p][0], b[kb], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[p][1], b[kb], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[p][2], b[kb], a2, 0, 0, 0);
                }
            }
            const i32x4 t = a0 + (a1 << 8) + (a2 << 16);
            // result element (m = 4g + reg, n): output 16*mt + 4g + reg of this wave's window, source row n of the step
            if (!(MODE & 4)) {
                int *p = tl + n * TPITCH + ((wave * 18u + mt * 16u + 4u * g) % 176u);
                atomicAdd(p + 0, t.x); atomicAdd(p + 1, t.y); atomicAdd(p + 2, t.z); atomicAdd(p + 3, t.w);
            } else { vacc[0].x += (float)(t.x + t.y + t.z + t.w); }
        }
        __syncthreads();
        if ((MODE & 1) && wave < 2) {
            // vertical pass of the previous... this step's tile: lane <-> 2 adjacent output columns, 16 rows, 7 live output rows
            const uint32_t col = (wave * 64u + lane) * 2u;
            if (col < 180u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int2 v = *reinterpret_cast<const int2 *>(tl + r * TPITCH + col + (col & 1u));
                    *reinterpret_cast<int2 *>(tl + r * TPITCH + col + (col & 1u)) = int2{0, 0};
                    const f32x2 x = {__builtin_fmaf((float)v.x, 2.384185791015625e-07f, 128.0f), __builtin_fmaf((float)v.y, 2.384185791015625e-07f, 128.0f)};
#pragma unroll
                    for (int sl = 0; sl < 7; ++sl) {
                        const float w = __int_as_float(tile[2 * 16 * TPITCH - 1 - ((r * 7 + sl) & 127)] | 0x3c000000);
                        vacc[sl] = __builtin_elementwise_fma(x, f32x2{w, w}, vacc[sl]);
                    }
                }
            }
        }
    };
    for (uint32_t s = 0; s < nsteps; s += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
            if (s + k < nsteps) step(k, s + k);
    }
    float keep = 0;
    for (int k = 0; k < 7; ++k) keep += vacc[k].x + vacc[k].y;
    if (keep == 123.456f) out[blockIdx.x] = keep;
}

template <int WAVES, int MODE, int DEPTH = 3>
static void run(const char *name, const uint8_t *src, const i32x4 *wtab, float *out, int nimg)
{
    const uint32_t W = 1920, H = 1080, C = 3, pitch = W * C, img_bytes = pitch * H;
    const uint32_t cw = WAVES * 128, nstrips = (pitch + (cw * 9 / 10) - 1) / (cw * 9 / 10);
    const uint32_t stride = nstrips > 1 ? (pitch - cw + nstrips - 2) / (nstrips - 1) / 64 * 64 : 0;
    const size_t ldsb = 2 * 16 * TPITCH * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hfirst<WAVES, MODE, DEPTH><<<nimg * nstrips, WAVES * 64, ldsb>>>(src, wtab, out, pitch, H, img_bytes, nstrips, stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hfirst<WAVES, MODE, DEPTH><<<nimg * nstrips, WAVES * 64, ldsb>>>(src, wtab, out, pitch, H, img_bytes, nstrips, stride);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = (double)nimg * img_bytes / 1e9;
    printf("%-64s %.3f ms  (%.2f TB/s of source bytes; %u strips of %u B at stride %u = %.2fx bytes)\n", name, ms, gb / ms, nstrips, cw, stride,
           (double)nstrips * cw / pitch);
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
    const size_t wn = (size_t)16 * NPAIR * 3 * 64;
    {
        std::vector<int> h(wn * 4);
        for (auto &v : h) v = rand();
        CK(hipMalloc(&wtab, wn * 16));
        CK(hipMemcpy(wtab, h.data(), wn * 16, hipMemcpyHostToDevice));
    }
    float *out;
    CK(hipMalloc(&out, 1 << 20));
    run<8, 6>("8 waves: loads + xor + VALU stand-in, no LDS adds", src, wtab, out, nimg);
    run<8, 4>("8 waves: + 18 i8 MFMAs per step, no LDS adds", src, wtab, out, nimg);
    run<8, 0>("8 waves: + ds_add of the partial sums + barrier per step", src, wtab, out, nimg);
    run<8, 1>("8 waves: + vertical pass on the tile (2 waves)", src, wtab, out, nimg);
    run<10, 0>("10 waves: horizontal stage + ds_add", src, wtab, out, nimg);
    run<10, 1>("10 waves: + vertical pass on the tile (2 waves)", src, wtab, out, nimg);
    run<10, 0, 6>("10 waves, prefetch 6 steps: horizontal stage + ds_add", src, wtab, out, nimg);
    run<10, 4, 6>("10 waves, prefetch 6 steps: horizontal stage, no LDS, no barrier", src, wtab, out, nimg);
    run<10, 6, 6>("10 waves, prefetch 6 steps: loads + xor only", src, wtab, out, nimg);
    run<10, 6, 3>("10 waves, prefetch 3 steps: loads + xor only", src, wtab, out, nimg);
    run<10, 4, 3>("10 waves, prefetch 3 steps: horizontal stage, no LDS, no barrier", src, wtab, out, nimg);
    return 0;
}
