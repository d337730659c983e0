// Microbenchmark / probe (development tool, not product): checks on the device the operand structure of the two
// matrix instructions the MFMA resample kernel is built on, and measures their issue rates next to VALU work.
//   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
//
//  1. v_mfma_i32_16x16x64_i8: claim  D[m][n] = sum_{g<4, j<16} A(lane 16g+m).byte[j] * B(lane 16g+n).byte[j],
//     result element (m = 4*(lane>>4) + reg, n = lane&15) in register `reg` of `lane`.
//  2. v_mfma_f32_16x16x4_f32: claim  D[m][n] = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))) bit for bit with
//     a_g = A(lane 16g+m), b_g = B(lane 16g+n), same result layout.
//  3. v_perm_b32 selector semantics (4x4 byte transpose in 8 instructions).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_i8(const i32x4 *a, const i32x4 *b, i32x4 *d)
{
    const int l = threadIdx.x;
    i32x4 c = {0, 0, 0, 0};
    d[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], c, 0, 0, 0);
}

__global__ void probe_f32(const float *a, const float *b, const f32x4 *c, f32x4 *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c[l], 0, 0, 0);
}

__device__ __forceinline__ void transpose4x4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t *o)
{
    // __builtin_amdgcn_perm(s0, s1, sel): byte k of the result = byte sel[k] of {s0 (4..7), s1 (0..3)}
    const uint32_t u0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u); // r0.b0 r1.b0 r0.b1 r1.b1
    const uint32_t u1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u); // r0.b2 r1.b2 r0.b3 r1.b3
    const uint32_t u2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
    const uint32_t u3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(u2, u0, 0x05040100u); // u0.b0 u0.b1 u2.b0 u2.b1
    o[1] = __builtin_amdgcn_perm(u2, u0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(u3, u1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(u3, u1, 0x07060302u);
}

__global__ void probe_perm(const uint32_t *in, uint32_t *out)
{
    const int l = threadIdx.x;
    uint32_t o[4];
    transpose4x4(in[4 * l], in[4 * l + 1], in[4 * l + 2], in[4 * l + 3], o);
    for (int k = 0; k < 4; ++k) out[4 * l + k] = o[k];
}

// ---- rates -------------------------------------------------------------------------------------------------
// MODE 0: i8 MFMA only; 1: f32 MFMA only; 2: i8 MFMA + NV v_perm per MFMA; 3: f32 MFMA + NV VALU per MFMA; 4: VALU only (v_perm)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t *out, int iters, uint32_t seed)
{
    i32x4 a = {(int)(seed + threadIdx.x), (int)seed * 3, (int)seed * 5, (int)seed * 7};
    i32x4 b = {(int)(seed ^ threadIdx.x), 11, 13, 17};
    i32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 facc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float fa = (float)threadIdx.x, fb = 1.0001f;
    uint32_t u[8];
    for (int k = 0; k < 8; ++k) u[k] = seed * (k + 1) + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0 || MODE == 2) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
            if (MODE == 1 || MODE == 3) facc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, facc[k], 0, 0, 0);
            if (MODE >= 2) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[(v + k) % 8]) : "v"(u[(v + k + 1) % 8]), "v"(u[(v + k + 2) % 8]), "s"(0x07050301u));
            }
        }
    }
    uint32_t s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w + (uint32_t)(facc[k].x + facc[k].y + facc[k].z + facc[k].w);
    for (int k = 0; k < 8; ++k) s += u[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
static void run_rate(const char *name, int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, threads = 256, iters = 2048;
    uint32_t *out;
    CK(hipMalloc(&out, sizeof(uint32_t) * blocks * threads));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    rate_kernel<MODE, NV><<<blocks, threads>>>(out, iters, 7u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) rate_kernel<MODE, NV><<<blocks, threads>>>(out, iters, 7u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    // per SIMD: wg_per_cu waves (one wave of each workgroup lands on each SIMD), each iters*4 MFMA (+ NV VALU each)
    const double clk = 2.4e9 * ms * 1e-3;
    const double mfma_per_simd = (double)wg_per_cu * iters * 4;
    printf("%-34s waves/SIMD %d  %.3f ms  -> %.1f clk per MFMA-slot per SIMD (at 2.4 GHz nominal)\n", name, wg_per_cu, ms, clk / mfma_per_simd);
    CK(hipFree(out));
}

int main()
{
    // ---- 1. i8 structure ----
    {
        std::vector<int8_t> A(64 * 16), B(64 * 16);
        srand(1);
        for (auto &v : A) v = (int8_t)(rand() % 256 - 128);
        for (auto &v : B) v = (int8_t)(rand() % 256 - 128);
        i32x4 *da, *db, *dd;
        CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice));
        probe_i8<<<1, 64>>>(da, db, dd);
        std::vector<int32_t> D(256);
        CK(hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (l >> 4) + r, n = l & 15;
                int32_t ref = 0;
                for (int g = 0; g < 4; ++g)
                    for (int j = 0; j < 16; ++j) ref += (int32_t)A[(16 * g + m) * 16 + j] * (int32_t)B[(16 * g + n) * 16 + j];
                if (ref != D[l * 4 + r]) ++bad;
            }
        printf("i8 16x16x64 structure: %s (%d of 256 mismatch)\n", bad ? "MISMATCH" : "ok", bad);
    }
    // ---- 2. f32 chain ----
    {
        std::vector<float> A(64), B(64), C(256), D(256);
        srand(2);
        for (auto &v : A) v = (float)(rand() % 100000) * 1.37e-3f - 50.0f;
        for (auto &v : B) v = (float)(rand() % 100000) * 3.1e-6f - 0.1f;
        for (auto &v : C) v = (float)(rand() % 1000) * 0.77f;
        float *da, *db; f32x4 *dc, *dd;
        CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(da, A.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, B.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, C.data(), 1024, hipMemcpyHostToDevice));
        probe_f32<<<1, 64>>>(da, db, dc, dd);
        CK(hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost));
        int bad_asc = 0, bad_desc = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (l >> 4) + r, n = l & 15;
                float up = C[l * 4 + r], dn = C[l * 4 + r];
                for (int g = 0; g < 4; ++g) up = fmaf(A[16 * g + m], B[16 * g + n], up);
                for (int g = 3; g >= 0; --g) dn = fmaf(A[16 * g + m], B[16 * g + n], dn);
                uint32_t x, y, z;
                memcpy(&x, &up, 4); memcpy(&y, &D[l * 4 + r], 4); memcpy(&z, &dn, 4);
                if (x != y) ++bad_asc;
                if (z != y) ++bad_desc;
            }
        printf("f32 16x16x4 fma chain: ascending-k %s (%d), descending-k (%d)\n", bad_asc ? "MISMATCH" : "ok bit-exact", bad_asc, bad_desc);
    }
    // ---- 3. perm ----
    {
        std::vector<uint32_t> in(256), out(256);
        for (int i = 0; i < 256; ++i) in[i] = (uint32_t)rand() * 2654435761u;
        uint32_t *di, *dout;
        CK(hipMalloc(&di, 1024)); CK(hipMalloc(&dout, 1024));
        CK(hipMemcpy(di, in.data(), 1024, hipMemcpyHostToDevice));
        probe_perm<<<1, 64>>>(di, dout);
        CK(hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int q = 0; q < 4; ++q) {
                uint32_t ref = 0;
                for (int r = 0; r < 4; ++r) ref |= ((in[4 * l + r] >> (8 * q)) & 255u) << (8 * r);
                if (ref != out[4 * l + q]) ++bad;
            }
        printf("4x4 byte transpose by v_perm_b32: %s (%d)\n", bad ? "MISMATCH" : "ok", bad);
    }
    // ---- 4. rates ----
    for (int w = 1; w <= 2; ++w) {
        run_rate<0, 0>("i8 16x16x64 alone", w);
        run_rate<1, 0>("f32 16x16x4 alone", w);
        run_rate<2, 2>("i8 16x16x64 + 2 v_perm", w);
        run_rate<2, 4>("i8 16x16x64 + 4 v_perm", w);
        run_rate<2, 8>("i8 16x16x64 + 8 v_perm", w);
        run_rate<3, 4>("f32 16x16x4 + 4 v_perm", w);
        run_rate<3, 8>("f32 16x16x4 + 8 v_perm", w);
        run_rate<3, 16>("f32 16x16x4 + 16 v_perm", w);
        run_rate<4, 8>("8 v_perm alone", w);
    }
    return 0;
}
