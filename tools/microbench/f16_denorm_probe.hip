// Microbenchmark / probe (development tool, not product): does v_mfma_f32_16x16x32_f16 on gfx950 honour SUBNORMAL f16 inputs?
//   hipcc --offload-arch=gfx950 -O3 f16_denorm_probe.hip -o f16_denorm_probe && ./f16_denorm_probe
//
// Why it matters (fl_mfma.hip, round 4): a byte b zero-extended to 16 bits IS the f16 subnormal b * 2^-24, so the transposed
// rows become MFMA operands with one v_perm_b32 per two bytes and NO bias (round 2/3 used 0x6400 | b = 1024 + b, whose bias
// costs the f32 accumulator three bits).  With weights as three f16 terms (hi + mid + lo = the f32 weight exactly) the
// vertical pass is then u8 x f32 -> f32, the reference's own width.
//  1. A = subnormal f16 (bytes), B = normal f16: D must equal the exact sum (small integers: exactly representable).
//  2. A = subnormal, B = subnormal: products of 2^-48 scale must survive in the f32 result.
//  3. A = bytes, B = three-term weights of random f32 values: D vs a float64 sum (error must be f32-rounding sized).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// nterm B operands per lane, accumulated into one D
__global__ void probe(const u32x4 *a, const u32x4 *b, int nterm, f32x4 *d)
{
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    const f16x8 av = __builtin_bit_cast(f16x8, a[l]);
    for (int t = 0; t < nterm; ++t) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, b[t * 64 + l]), c, 0, 0, 0);
    d[l] = c;
}

static uint16_t f16_bits(double v)
{
    const float f = (float)v;
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 255u) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}
static double f16_value(uint16_t h)
{
    const int s = (h & 0x8000u) ? -1 : 1;
    const int e = (h >> 10) & 31, m = h & 0x3ff;
    if (e == 0) return s * ldexp((double)m, -24);
    return s * ldexp((double)(m | 0x400), e - 25);
}

int main()
{
    // operand element (lane, j): A[row = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][col = lane & 15]
    std::vector<uint16_t> A(64 * 8), B(3 * 64 * 8);
    std::vector<double> Aval(16 * 32), Bval(32 * 16);
    u32x4 *da, *db; f32x4 *dd;
    CK(hipMalloc(&da, 64 * 16)); CK(hipMalloc(&db, 3 * 64 * 16)); CK(hipMalloc(&dd, 64 * 16));
    std::vector<float> D(64 * 4);
    srand(7);
    for (int test = 0; test < 3; ++test) {
        std::fill(B.begin(), B.end(), 0);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int m = lane & 15, k = 8 * (lane >> 4) + j;
                const uint16_t byte = (uint16_t)(rand() & 255);
                A[lane * 8 + j] = byte;                     // zero-extended byte = f16 subnormal byte * 2^-24
                Aval[m * 32 + k] = ldexp((double)byte, -24);
            }
        int nterm = 1;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int n = lane & 15, k = 8 * (lane >> 4) + j;
                if (test == 0) {                            // small integers times 2^10: normal f16
                    const int w = (rand() % 31) - 15;
                    B[lane * 8 + j] = f16_bits(ldexp((double)w, 10));
                    Bval[k * 16 + n] = ldexp((double)w, 10);
                } else if (test == 1) {                     // subnormal B too
                    const int w = (rand() % 63) - 31;
                    B[lane * 8 + j] = f16_bits(ldexp((double)w, -24));
                    Bval[k * 16 + n] = ldexp((double)w, -24);
                } else {                                    // f32 weights (|w| < 0.5, some tiny) * 2^15 as three f16 terms
                    float w = ((float)rand() / RAND_MAX - 0.5f) * ((rand() & 3) ? 0.6f : 1e-4f);
                    const double ws = ldexp((double)w, 15);
                    const uint16_t t0 = f16_bits(ws), t1 = f16_bits(ws - f16_value(t0)), t2 = f16_bits(ws - f16_value(t0) - f16_value(t1));
                    B[(0 * 64 + lane) * 8 + j] = t0; B[(1 * 64 + lane) * 8 + j] = t1; B[(2 * 64 + lane) * 8 + j] = t2;
                    const double rep = f16_value(t0) + f16_value(t1) + f16_value(t2);
                    if (rep != ws) printf("  (three terms do not represent w = %g exactly: off by %g relative)\n", w, (rep - ws) / ws);
                    Bval[k * 16 + n] = ws;
                    nterm = 3;
                }
            }
        CK(hipMemcpy(da, A.data(), 64 * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, B.data(), 3 * 64 * 16, hipMemcpyHostToDevice));
        probe<<<1, 64>>>(da, db, nterm, dd);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dd, 64 * 16, hipMemcpyDeviceToHost));
        double maxrel = 0, maxabs = 0; int exact = 0, zeros = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (lane >> 4) + r, n = lane & 15;
                double want = 0, mag = 0;
                for (int k = 0; k < 32; ++k) { want += Aval[m * 32 + k] * Bval[k * 16 + n]; mag += fabs(Aval[m * 32 + k] * Bval[k * 16 + n]); }
                const double got = D[lane * 4 + r];
                if (got == want) ++exact;
                if (got == 0.0 && want != 0.0) ++zeros;
                maxabs = std::max(maxabs, fabs(got - want));
                if (mag > 0) maxrel = std::max(maxrel, fabs(got - want) / mag);
            }
        printf("test %d (%s): %d of 256 results exact, %d flushed to zero, max |err| %.3g, max |err| / sum|products| %.3g (f32 eps = 6e-8)\n", test,
               test == 0 ? "A subnormal bytes x B normal integers" : test == 1 ? "A subnormal x B subnormal" : "A bytes x B = three f16 terms of f32 weights, three MFMAs",
               exact, zeros, maxabs, maxrel);
    }
    return 0;
}
