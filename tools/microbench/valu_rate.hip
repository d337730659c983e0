// Microbenchmark (development tool, not product): per-instruction VALU issue
// rates on gfx950 that the resample kernel design depends on, and the HBM
// streaming-read rate of the row-streaming access pattern.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NACC = 16;
constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(float* out, float w_in, unsigned seed)
{
    float a[NACC];
    f32x2 p[NACC];
    unsigned u[NACC];
    for (int k = 0; k < NACC; ++k) { a[k] = (float)(threadIdx.x + k); p[k] = f32x2{a[k], a[k] + 1.0f}; u[k] = seed * (k + 1) + threadIdx.x; }
    float w = w_in;
    f32x2 w2 = f32x2{w_in, w_in};
    unsigned hw = 0x3c003c00u; // two f16 1.0
    f32x2 wv2 = f32x2{w_in + threadIdx.x * 0.0f, w_in};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(a[(k + 1) % NACC]), "s"(w));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(p[(k + 1) % NACC]), "s"(w2));
            if (OP == 2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(u[k]), "s"(hw));
            if (OP == 3) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[k]) : "v"(u[k]));
            if (OP == 4) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[k]) : "v"(u[(k + 1) % NACC]), "v"(u[(k + 2) % NACC]), "s"(0x07050301u));
            if (OP == 5) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(u[k]) : "v"(u[(k + 1) % NACC]), "s"(hw));
            if (OP == 6) asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(a[(k + 1) % NACC]), "s"(w));
            if (OP == 7) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[k]) : "v"(u[k]), "v"(u[(k + 1) % NACC]));
            if (OP == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[k]) : "v"(p[(k + 1) % NACC]), "v"(wv2));
            if (OP == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k]) : "v"(p[(k + 1) % NACC]), "v"(p[(k + 2) % NACC]));
            if (OP == 10) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(a[(k + 1) % NACC]), "v"(a[(k + 2) % NACC]));
        }
    }
    float s = 0;
    for (int k = 0; k < NACC; ++k) s += a[k] + p[k].x + p[k].y + (float)u[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Row-streaming read: each lane reads LANEB bytes per row, rows are ROWB bytes apart,
// a workgroup walks `rows` rows of its own image.  Mimics the resample vertical pass.
template <int LANEB>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char* src, unsigned* out, unsigned rowb, unsigned rows, size_t img_bytes)
{
    const unsigned char* base = src + (size_t)blockIdx.x * img_bytes;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)img_bytes, 0x00020000);
    unsigned acc = 0;
    unsigned voff = threadIdx.x * LANEB;
    if (voff >= rowb) return;
#pragma unroll 8
    for (unsigned r = 0; r < rows; ++r) {
        if (LANEB == 12) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, r * rowb, 0); acc += v.x ^ v.y ^ v.z; }
        if (LANEB == 16) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, r * rowb, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int OP>
double run_valu(const char* name, int flop_per_lane_op)
{
    int blocks = 256 * 8, threads = 256;
    float* out; CK(hipMalloc(&out, sizeof(float) * blocks * threads));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    valu_kernel<OP><<<blocks, threads>>>(out, 1.0001f, 7u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) valu_kernel<OP><<<blocks, threads>>>(out, 1.0001f, 7u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double lane_ops = (double)blocks * threads * ITERS * NACC * (OP == 6 ? 2 : 1);
    double wave_instr_per_s = lane_ops / 64 / (ms * 1e-3);
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s  %7.2f G wave-instr/s  (%.1f TFLOP/s at %d flop/op)  wave-instr/clk/CU@2.4GHz=%.2f\n",
           name, ms, lane_ops / (ms * 1e-3) / 1e12, wave_instr_per_s / 1e9,
           lane_ops * flop_per_lane_op / (ms * 1e-3) / 1e12, flop_per_lane_op, wave_instr_per_s / 256 / 2.4e9);
    CK(hipFree(out));
    return ms;
}

template <int LANEB>
void run_stream(const char* name, unsigned nimg, unsigned threads)
{
    unsigned rowb = 5760, rows = 1080; size_t img = (size_t)rowb * rows;
    unsigned char* src; unsigned* out;
    CK(hipMalloc(&src, img * nimg)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, img * nimg));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    stream_kernel<LANEB><<<nimg, threads>>>(src, out, rowb, rows, img);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) stream_kernel<LANEB><<<nimg, threads>>>(src, out, rowb, rows, img);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-28s nimg=%u threads=%u %8.3f ms  %8.1f GB/s\n", name, nimg, threads, ms, (double)img * nimg / (ms * 1e-3) / 1e9);
    CK(hipFree(src)); CK(hipFree(out));
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    run_valu<0>("v_fma_f32 (sgpr w)", 2);
    run_valu<1>("v_pk_fma_f32 (sgpr w)", 4);
    run_valu<2>("v_dot2_f32_f16", 4);
    run_valu<7>("v_dot2c_f32_f16", 4);
    run_valu<3>("v_cvt_f32_ubyte1", 1);
    run_valu<4>("v_perm_b32", 1);
    run_valu<5>("v_dot4_i32_i8", 8);
    run_valu<6>("v_mul_f32+v_add_f32", 1);
    run_valu<8>("v_pk_fma_f32 (vgpr w, op_sel)", 4);
    run_valu<9>("v_pk_fma_f32 (3 vgpr pairs)", 4);
    run_valu<10>("v_fmac_f32 (vgpr)", 2);
    run_stream<12>("stream b96 512thr", 1024, 512);
    run_stream<16>("stream b128 384thr", 1024, 384);
    run_stream<12>("stream b96 512thr", 2048, 512);
    return 0;
}
