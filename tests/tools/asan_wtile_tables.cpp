// Host-only harness for tests/test_wtile.py: the window-tile kernel's table builder (csrc/fl_mfma_tables.cpp build_wtile_plan) and the
// host run of its tables (oracle/wtile_model.cpp) over random geometries, built with -fsanitize=address,undefined: any out-of-bounds
// access in the builders (windows at picture borders, strips, operand blocks) aborts the run.  Prints "<plans> <rejected>".
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

extern "C" int wtile_model_run(const uint8_t *src, uint32_t sw, uint32_t sh, uint32_t cs, uint32_t rw, uint32_t rh, float blur_sigma,
                                       uint8_t *dst, uint32_t info[8]);

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 200;
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&](uint32_t lo, uint32_t hi) { s = s * 6364136223846793005ull + 1442695040888963407ull; return lo + (uint32_t)((s >> 33) % (hi - lo + 1)); };
    int ok = 0, rejected = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t cs = rnd(1, 4), sw = rnd(1, 300), sh = rnd(1, 300);
        std::vector<uint8_t> src((size_t)sw * sh * cs);
        for (auto &b : src) b = (uint8_t)rnd(0, 255);
        uint32_t info[8];
        int r;
        if (rnd(0, 2) == 0) {
            static const float sig[] = {0.3f, 1.0f, 4.0f, 10.0f, 20.0f};
            std::vector<uint8_t> dst(src.size());
            r = wtile_model_run(src.data(), sw, sh, cs, 0, 0, sig[rnd(0, 4)], dst.data(), info);
        } else {
            const uint32_t rw = rnd(1, 400), rh = rnd(1, 400);
            std::vector<uint8_t> dst((size_t)rw * rh * cs);
            r = wtile_model_run(src.data(), sw, sh, cs, rw, rh, 0.0f, dst.data(), info);
        }
        if (r) ++ok; else ++rejected;
    }
    printf("%d %d\n", ok, rejected);
    return 0;
}
