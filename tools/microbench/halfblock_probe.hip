// halfblock_probe.hip -- does a wave's load path run faster when its K-block (32 rows x 256 bytes, 8 x global_load_lds_dwordx4)
// is requested as two half-blocks, so that something is always in flight?  The matrix-pipe resample kernel's addressing:
// 1920x1080 Rgb8 pictures (row pitch 5760), three strips of 2048 bytes per row, one workgroup = one picture x one strip, wave w
// owns byte columns 256 w .. + 255, load u covers row octet u >> 1, column half u & 1 (lane: row lane & 7, 16-byte tile lane >> 3).
//   variant 0: the kernel's scheme: wait for the whole K-block, (reads), request the next one
//   variant 1: half-blocks by ROW octets (0,1 | 2,3): wait H0, (reads), request H0 of the next block, wait H1, (reads), request H1
//   variant 2: half-blocks by COLUMN halves (the two 128-byte halves of a row segment at different times)
//   variant 3: as 0, but the next K-block is requested BEFORE the (reads) -- an upper bound: needs a second ring
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/halfblock_probe.hip -o tools/microbench/halfblock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__shared__ __attribute__((aligned(16))) uint8_t ring[2 * 8 * 8192];

template <int V>
__global__ __launch_bounds__(512, 1) void probe(const uint8_t *__restrict__ src, uint32_t nkb, uint32_t sleep_units)
{
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t pic = blockIdx.x / 3u, strip = blockIdx.x % 3u;
    const uint32_t pitch = 5760u, byte0 = strip * 1856u;
    const uint8_t *base = src + (size_t)pic * pitch * 1080u;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(ring + wave * 8192u));
    const uint32_t lq = lane & 7u, lt = lane >> 3;
    auto issue = [&](uint32_t s, uint32_t u, uint32_t buf) {
        const uint32_t row = min(32u * s + 8u * (u >> 1) + lq, 1079u);
        const uint32_t col = min(byte0 + wave * 256u + (u & 1u) * 128u + lt * 16u, pitch - 16u);
        const uint32_t off = row * pitch + col;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(ring_lds + buf * 65536u + u * 1024u) : "memory");
    };
    auto work = [&](uint32_t units) { for (uint32_t k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(8); };
    for (uint32_t u = 0; u < 8; ++u) issue(0, V == 2 ? ((u & 3u) * 2u + (u >> 2)) : u, 0); // (variant 2 issues column half 0 of all octets first)
    for (uint32_t s = 0; s < nkb; ++s) {
        const bool more = s + 1u < nkb;
        if (V == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            work(sleep_units);
            if (more) for (uint32_t u = 0; u < 8; ++u) issue(s + 1u, u, 0);
            work(3u * sleep_units); // (the compute that overlaps the load)
        } else if (V == 1 || V == 2) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            work(sleep_units / 2u);
            if (more) for (uint32_t k = 0; k < 4; ++k) issue(s + 1u, V == 1 ? k : 2u * k, 0);
            if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            work(sleep_units - sleep_units / 2u);
            if (more) for (uint32_t k = 0; k < 4; ++k) issue(s + 1u, V == 1 ? 4u + k : 2u * k + 1u, 0);
            work(3u * sleep_units);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (more) for (uint32_t u = 0; u < 8; ++u) issue(s + 1u, u, (s + 1u) & 1u);
            work(sleep_units);
            work(3u * sleep_units);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char **argv)
{
    const uint32_t npic = 1024, nkb = 34;
    const size_t bytes = (size_t)npic * 5760u * 1080u;
    uint8_t *src;
    if (hipMalloc(&src, bytes + 65536) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void (*ks[4])(const uint8_t *, uint32_t, uint32_t) = {probe<0>, probe<1>, probe<2>, probe<3>};
    for (uint32_t sl = 0; sl <= 8; sl += 2)
        for (int rep = 0; rep < 2; ++rep)
            for (int v = 0; v < 4; ++v) {
                for (int w = 0; w < 3; ++w) ks[v]<<<npic * 3, 512>>>(src, nkb, sl);
                (void)hipEventRecord(e0);
                for (int w = 0; w < 20; ++w) ks[v]<<<npic * 3, 512>>>(src, nkb, sl);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
                printf("work units %u variant %d: %.3f ms per launch, %.2f TB/s of requested bytes (%.3f of the 8 TB/s peak for the 6.616 GB of config 1)\n", sl, v, ms,
                       npic * 3.0 * nkb * 65536.0 / (ms * 1e-3) / 1e12, 6.616e9 / (ms * 1e-3) / 8e12);
            }
    return 0;
}
