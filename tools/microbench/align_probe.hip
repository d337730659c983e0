// align_probe.hip -- what a strip whose first byte is not on a 128-byte line costs the matrix-pipe kernel's load path, and whether
// splitting a wave's 256-byte row piece at the LINE boundary (instead of in the middle) buys it back.  Addressing as in
// fl_mfma.hip: 1920x1080 Rgb8 (pitch 5760), one workgroup = one picture x one strip of 2048 bytes, wave w owns bytes 256 w .. + 255
// of the strip, a K-block = 32 rows = 8 x global_load_lds_dwordx4 per wave (row octet u >> 1, half u & 1; lane: row lane & 7,
// 16-byte piece lane >> 3 of the half).
//   mode 0: the planner's strips (first bytes 0 / 1872 / 3792: the inner ones 80 bytes past a line)
//   mode 1: the same bytes per strip from line starts (0 / 1792 / 3712) -- what an aligned plan would read
//   mode 2: the planner's strips, halves rotated: instruction "half 0" takes the eight pieces of the one whole line inside the
//           256 bytes, "half 1" the pieces before and after it (3 line requests per row instead of 4)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/align_probe.hip -o tools/microbench/align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__shared__ __attribute__((aligned(16))) uint8_t ring[8 * 8192];

__global__ __launch_bounds__(512, 1) void probe(const uint8_t *__restrict__ src, uint32_t nkb, uint32_t sleep_units, uint32_t mode, uint32_t only_strip)
{
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t pic = blockIdx.x / 3u, strip = only_strip < 3u ? only_strip : blockIdx.x % 3u;
    const uint32_t pitch = 5760u;
    const uint32_t b_plan[3] = {0u, 1872u, 3792u}, b_line[3] = {0u, 1792u, 3712u};
    const uint32_t byte0 = mode == 1u ? b_line[strip] : b_plan[strip];
    const uint8_t *base = src + (size_t)pic * pitch * 1080u;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(ring + wave * 8192u));
    const uint32_t lq = lane & 7u, lt = lane >> 3;
    const uint32_t rot = mode == 2u ? ((128u - byte0 % 128u) % 128u) / 16u : 0u; // pieces in front of the first line boundary
    auto issue = [&](uint32_t s, uint32_t u) {
        const uint32_t row = min(32u * s + 8u * (u >> 1) + lq, 1079u);
        const uint32_t piece = ((u & 1u) * 8u + lt + rot) & 15u;
        const uint32_t col = min(byte0 + wave * 256u + piece * 16u, pitch - 16u);
        const uint32_t off = row * pitch + col;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(ring_lds + u * 1024u) : "memory");
    };
    auto work = [&](uint32_t units) { for (uint32_t k = 0; k < units; ++k) __builtin_amdgcn_s_sleep(8); };
    for (uint32_t u = 0; u < 8; ++u) issue(0, u);
    for (uint32_t s = 0; s < nkb; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        work(sleep_units);
        if (s + 1u < nkb) for (uint32_t u = 0; u < 8; ++u) issue(s + 1u, u);
        work(3u * sleep_units);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main()
{
    const uint32_t npic = 1024, nkb = 34;
    const size_t bytes = (size_t)npic * 5760u * 1080u;
    uint8_t *src;
    if (hipMalloc(&src, bytes + 65536) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[3] = {"planner's strips (0 / 1872 / 3792)", "line-aligned strips (0 / 1792 / 3712)", "planner's strips, halves split at the line boundary"};
    for (uint32_t sl = 0; sl <= 4; sl += 2)
        for (uint32_t strip = 0; strip <= 3; ++strip)
            for (uint32_t mode = 0; mode < 3; ++mode) {
                for (int w = 0; w < 3; ++w) probe<<<npic * 3, 512>>>(src, nkb, sl, mode, strip);
                (void)hipEventRecord(e0);
                for (int w = 0; w < 20; ++w) probe<<<npic * 3, 512>>>(src, nkb, sl, mode, strip);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
                printf("work units %u, %s, %s: %.3f ms per launch\n", sl, strip < 3 ? (strip == 0 ? "every workgroup on strip 0" : strip == 1 ? "every workgroup on strip 1" : "every workgroup on strip 2") : "strips 0, 1, 2 mixed",
                       names[mode], ms);
            }
    return 0;
}
