// store_probe.hip -- what a few pixel stores cost a wave that streams rows through LDS-DMA with one request group in flight
// (the matrix-pipe resample kernel's load pattern: 8 x global_load_lds_dwordx4 = 8 KB per wave and step, then s_waitcnt).
// Variants: 0 no store; 1 store, then the requests, wait vmcnt(0); 2 requests, then store, wait vmcnt(0);
//           3 requests, then store, wait vmcnt(1) (the store, youngest, is not waited for); 4 = 3 with the store as 4 instructions
//           and vmcnt(4).   A store happens every third step.
//           5 = 3 with the same 1 KB as ONE global_store_dwordx4; 6 = bursts: every 24th step (staggered over the waves) 8 x dwordx4 (the
//           bytes of 5); 7 = every 24th step 2 x dwordx4 (the bytes of 1-3); 8 = 1 with a global_load_dword in the store's place.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/store_probe.hip -o tools/microbench/store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__shared__ __attribute__((aligned(16))) uint8_t ring[8 * 8192];

template <int V>
__global__ __launch_bounds__(512, 1) void probe(const uint8_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t steps, size_t wg_stride)
{
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t *base = src + (size_t)blockIdx.x * wg_stride + wave * 8192u;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(ring + wave * 8192u));
    uint32_t *d = dst + (size_t)blockIdx.x * 65536u + wave * 4096u;
    auto issue = [&](uint32_t s) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t off = s * 65536u + u * 1024u + lane * 16u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(ring_lds + u * 1024u) : "memory");
        }
    };
    auto store = [&](uint32_t s, uint32_t n) {
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t off = ((s * 4u + k) & 15u) * 256u + lane * 4u; // 256 contiguous bytes per instruction
            asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(s), "s"(d) : "memory");
        }
    };
    auto store4 = [&](uint32_t s, uint32_t n) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 val = {s, s, s, s};
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t off = ((s + k) & 7u) * 1024u + lane * 16u; // 1 KB contiguous per instruction
            asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(val), "s"(d) : "memory");
        }
    };
    issue(0);
    for (uint32_t s = 0; s < steps; ++s) {
        const bool st = (s % 3u) == 2u;
        if (V == 3 && st) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (V == 4 && st) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (V == 5 && st) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_sleep(8); // (a little work between landing and the next request, as the kernel's transposed reads)
        const bool st_next = ((s + 1u) % 3u) == 2u;
        if (V == 1 && st_next) store(s, 1);
        if (s + 1u < steps) issue(s + 1u);
        if ((V == 2 || V == 3) && st_next) store(s, 1);
        if (V == 4 && st_next) store(s, 4);
        if (V == 5 && st_next) store4(s, 1);
        const bool burst = ((s + 3u * wave) % 24u) == 23u;
        if (V == 6 && burst) store4(s, 8);
        if (V == 7 && burst) store4(s, 2);
        if (V == 8 && st_next) {
            uint32_t got;
            const uint32_t off = (s & 15u) * 256u + lane * 4u;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(got) : "v"(off), "s"(d) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main()
{
    const uint32_t nwg = 256 * 8, steps = 48;
    const size_t wg_stride = (size_t)steps * 65536u;
    uint8_t *src; uint32_t *dst;
    if (hipMalloc(&src, nwg * wg_stride) != hipSuccess || hipMalloc(&dst, (size_t)nwg * 65536u * 4u) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, nwg * wg_stride);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void (*ks[9])(const uint8_t *, uint32_t *, uint32_t, size_t) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>};
    for (int rep = 0; rep < 3; ++rep)
        for (int v = 0; v < 9; ++v) {
            for (int w = 0; w < 3; ++w) ks[v]<<<nwg, 512>>>(src, dst, steps, wg_stride);
            hipEventRecord(e0);
            for (int w = 0; w < 20; ++w) ks[v]<<<nwg, 512>>>(src, dst, steps, wg_stride);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("variant %d: %.3f ms per launch, %.2f TB/s of loaded bytes\n", v, ms, nwg * (double)wg_stride / (ms * 1e-3) / 1e12);
        }
    return 0;
}
