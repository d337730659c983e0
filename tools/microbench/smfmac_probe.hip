// Microbenchmark / probe (development tool, not product): operand structure and issue rate of gfx950's 2:4-sparse integer
// matrix instruction v_smfmac_i32_16x16x128_i8, found empirically (the ISA text is not available in this environment).
//   hipcc --offload-arch=gfx950 -O3 smfmac_probe.hip -o smfmac_probe && ./smfmac_probe
//
// Why: the horizontal pass of the matrix-pipe resample kernel multiplies by a banded weight matrix whose K index runs over
// INTERLEAVED channel bytes -- an output of channel c has non-zero weights only on bytes of channel c, i.e. at most 2 of every
// 4 consecutive bytes for Rgb8 (1 of 4 for Rgba8): exactly the 2:4 structure the sparse instruction skips.
//
// Hypothesis H (CDNA3 convention carried to the double-K form):
//   A (4 VGPRs, 16 bytes per lane): lane 16 g + m holds the 16 STORED bytes of row m for dense positions 32 g .. 32 g + 31:
//       stored byte 2 t + e (t = 0..7, e = 0..1) sits at dense position 32 g + 4 t + idx(t, e),
//       idx(t, e) = bits [4 t + 2 e + 1 : 4 t + 2 e] of the lane's index register
//   B (8 VGPRs, 32 bytes per lane): lane 16 g + n holds B[k = 32 g + j][n] in byte j
//   D[m][n] in register r of lane 16 (m / 4) + n, r = m % 4 (as every 16x16 form)
// The probe does not assume H: for every (g, stored byte s, index code) it puts a single 1 into A and position codes into B
// and reads off WHICH byte of B was multiplied.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void probe(const v4i *a, const v8i *b, const int *idx, v4i *d)
{
    const int l = threadIdx.x, c = blockIdx.x;
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a[c * 64 + l], b[c * 64 + l], acc, idx[c * 64 + l], 0, 0);
    d[c * 64 + l] = acc;
}

// rates: MODE 0 = dense v_mfma_i32_16x16x64_i8, 1 = sparse v_smfmac_i32_16x16x128_i8; NACC independent accumulators
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(int *out, int iters, int seed)
{
    v4i acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = v4i{0, 0, 0, 0};
    v4i a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7};
    v8i b = {seed, seed + 1, seed + 2, seed + 3, seed + 4, seed + 5, seed + 6, seed + 7};
    v4i b4 = {seed, seed + 1, seed + 2, seed + 3};
    const int ix = 0x44444444 ^ (int)threadIdx.x; // (any valid-looking pattern)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (MODE == 0) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b4, acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, acc[k], ix, 0, 0);
        }
    }
    int s = 0;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC>
static void rate(const char *name)
{
    int *out;
    const int blocks = 256 * 2, iters = 4096;
    CK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<MODE, NACC><<<blocks, 256>>>(out, 16, 3);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<MODE, NACC><<<blocks, 256>>>(out, iters, 3);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: blocks * 4 waves / (256 CUs * 4 SIMDs) waves, each iters * NACC instructions
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * NACC;
    printf("%-46s %8.3f ms  -> %.1f ns per instruction and SIMD (= %.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    CK(hipFree(out));
}

int main()
{
    // ---- structure ------------------------------------------------------------------------------------------------------
    // case c = (g, s, code): A = 1 in stored byte s of lane 16 g + 5 (row m = 5), index field of that byte = code, every
    // other field of every lane = the pattern {0, 1} per pair; B byte j of lane 16 g' + n = (32 g' + j) - 64 (a position code
    // that fits a signed byte), so D[5][n] = code of the dense position that was read.
    const int NC = 4 * 16 * 4;
    std::vector<int32_t> A(NC * 64 * 4, 0), B(NC * 64 * 8), IDX(NC * 64), D(NC * 64 * 4);
    for (int c = 0; c < NC; ++c) {
        const int g = c >> 6, s = (c >> 2) & 15, code = c & 3;
        for (int lane = 0; lane < 64; ++lane) {
            int8_t *bb = reinterpret_cast<int8_t *>(&B[(c * 64 + lane) * 8]);
            for (int j = 0; j < 32; ++j) bb[j] = (int8_t)(32 * (lane >> 4) + j - 64);
            uint32_t ix = 0x44444444u; // every pair: idx0 = 0, idx1 = 1
            if (lane == 16 * g + 5) {
                reinterpret_cast<int8_t *>(&A[(c * 64 + lane) * 4])[s] = 1;
                ix = (ix & ~(3u << (2 * s))) | ((uint32_t)code << (2 * s));
            }
            IDX[c * 64 + lane] = (int32_t)ix;
        }
    }
    v4i *da, *dd; v8i *db; int *di;
    CK(hipMalloc(&da, A.size() * 4)); CK(hipMalloc(&db, B.size() * 4)); CK(hipMalloc(&di, IDX.size() * 4)); CK(hipMalloc(&dd, D.size() * 4));
    CK(hipMemcpy(da, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, IDX.data(), IDX.size() * 4, hipMemcpyHostToDevice));
    probe<<<NC, 64>>>(da, db, di, dd);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost));
    printf("measured map: A lane group g, stored byte s, index code 0 -> (B lane group, B byte); '+' = codes 1..3 read the next three bytes\n");
    for (int g = 0; g < 4; ++g) {
        printf("  g %d:", g);
        for (int s2 = 0; s2 < 16; ++s2) {
            int p[4];
            for (int code = 0; code < 4; ++code) p[code] = D[(((g * 16 + s2) * 4 + code) * 64 + 16) * 4 + 1] + 64;
            const bool lin = p[1] == p[0] + 1 && p[2] == p[0] + 2 && p[3] == p[0] + 3;
            printf(" %d:(%d,%d)%s", s2, p[0] >> 5, p[0] & 31, lin ? "+" : "?");
        }
        printf("\n");
    }
    int agree = 0, other = 0;
    for (int c = 0; c < NC; ++c) {
        const int g = c >> 6, s = (c >> 2) & 15, code = c & 3;
        // D[m = 5][n]: lane 16 * (5 / 4) + n, register 5 % 4 = 1
        int pos = -1000; bool uniform = true, elsewhere = false;
        for (int n = 0; n < 16; ++n) {
            const int v = D[(c * 64 + 16 + n) * 4 + 1] + 64;
            if (n == 0) pos = v; else if (v != pos) uniform = false;
        }
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r)
                if (!(lane >> 4 == 1 && r == 1) && D[(c * 64 + lane) * 4 + r] != 0) elsewhere = true;
        const int want = 32 * g + 4 * (s >> 1) + code;
        if (uniform && !elsewhere && pos == want) ++agree;
        else {
            if (other < 0) printf("  case g %d stored byte %2d code %d: read dense position %d%s%s (hypothesis: %d)\n", g, s, code, pos, uniform ? "" : " [not uniform over n]", elsewhere ? " [other rows non-zero]" : "", want);
            ++other;
        }
    }
    printf("structure: %d of %d cases read the dense position hypothesis H names; %d do not\n", agree, NC, other);

    // ---- a random 2:4 matrix against the plain sum -----------------------------------------------------------------------
    {
        srand(5);
        std::vector<int32_t> A1(64 * 4), B1(64 * 8), I1(64), D1(64 * 4);
        std::vector<int> Ad(16 * 128, 0), Bd(128 * 16);
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, m = lane & 15;
            uint32_t ix = 0;
            int8_t *aa = reinterpret_cast<int8_t *>(&A1[lane * 4]);
            for (int t = 0; t < 8; ++t) {
                int p0 = rand() % 4, p1 = rand() % 4;
                while (p1 == p0) p1 = rand() % 4;
                if (p0 > p1) std::swap(p0, p1);
                ix |= (uint32_t)(p0 | (p1 << 2)) << (4 * t);
                aa[2 * t] = (int8_t)(rand() % 255 - 127); aa[2 * t + 1] = (int8_t)(rand() % 255 - 127);
                Ad[m * 128 + 32 * g + 4 * t + p0] = aa[2 * t];
                Ad[m * 128 + 32 * g + 4 * t + p1] = aa[2 * t + 1];
            }
            I1[lane] = (int32_t)ix;
            int8_t *bb = reinterpret_cast<int8_t *>(&B1[lane * 8]);
            for (int j = 0; j < 32; ++j) { bb[j] = (int8_t)(rand() % 255 - 127); Bd[(32 * g + j) * 16 + m] = bb[j]; }
        }
        CK(hipMemcpy(da, A1.data(), A1.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, B1.data(), B1.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(di, I1.data(), I1.size() * 4, hipMemcpyHostToDevice));
        probe<<<1, 64>>>(da, db, di, dd);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D1.data(), dd, D1.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (lane >> 4) + r, n = lane & 15;
                long want = 0;
                for (int k = 0; k < 128; ++k) want += (long)Ad[m * 128 + k] * Bd[k * 16 + n];
                if (want != D1[lane * 4 + r]) ++bad;
            }
        printf("random 2:4 matrix (ascending index pairs): %d of 256 results differ from the plain sum under H\n", bad);
    }
    // ---- rates ------------------------------------------------------------------------------------------------------------
    rate<0, 1>("dense  16x16x64 i8, 1 accumulator (dependent)");
    rate<0, 4>("dense  16x16x64 i8, 4 accumulators");
    rate<1, 1>("sparse 16x16x128 i8, 1 accumulator (dependent)");
    rate<1, 4>("sparse 16x16x128 i8, 4 accumulators");
    return 0;
}
