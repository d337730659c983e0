// wtile_model.cpp -- TEST INFRASTRUCTURE (like the rest of oracle/: only tests/, smoke() and bench's cpu_baseline may use this directory).
// The window-tile matrix-pipe kernel's arithmetic (fanlin-rs_amd/csrc/fl_wtile.h) run on the host FROM THE KERNEL'S OWN TABLES, operand
// for operand: built together with the product's table builder (csrc/fl_mfma_tables.cpp, csrc/fl_tables.cpp: host-only code) into
// oracle/libwtile_model.so.  What the tests hold the device's bytes against (tests/test_wtile.py), and -- against the C oracle -- what
// checks the table builder without a device.  Reference: image 0.25.6 imageops/sample.rs behind resize_exact and blur
// (/root/reference/src/handler.rs:229-255).  Nothing in libfanlin_gpu.so contains or calls it.
#include <math.h>
#include <string.h>

#include <vector>

#include "../fanlin-rs_amd/csrc/fl_mfma.h"
#include "../fanlin-rs_amd/csrc/fl_tables.h"
#include "../fanlin-rs_amd/csrc/fl_wtile.h"

extern "C" int wtile_model_run(const uint8_t *src, uint32_t sw, uint32_t sh, uint32_t cs, uint32_t rw, uint32_t rh, float blur_sigma,
                                       uint8_t *dst, uint32_t info[8])
{
    using namespace fl;
    if (info) for (int k = 0; k < 8; ++k) info[k] = 0;
    if (!sw || !sh || !cs || cs > 4) return 0;
    HostAxis v, h;
    if (blur_sigma > 0.0f) {
        rw = sw; rh = sh;
        build_axis(sh, sh, FILTER_GAUSSIAN, blur_sigma, v);
        build_axis(sw, sw, FILTER_GAUSSIAN, blur_sigma, h);
    } else {
        if (!rw || !rh) return 0;
        build_axis(sh, rh, FILTER_LANCZOS3, 0.0f, v);
        build_axis(sw, rw, FILTER_LANCZOS3, 0.0f, h);
    }
    HostWtPlan p;
    build_wtile_plan(v, h, cs, 0, 0, rw, rh, p);
    if (!p.ok) return 0;
    const uint32_t *blk = p.blk.data();
    const WtHeader hd = *reinterpret_cast<const WtHeader *>(blk);
    const WtMTile *mts = reinterpret_cast<const WtMTile *>(blk + hd.mt_off);
    const WtNTile *nts = reinterpret_cast<const WtNTile *>(blk + hd.nt_off);
    const WtStrip *strips = reinterpret_cast<const WtStrip *>(blk + hd.strip_off);
    if (info) {
        info[0] = hd.n_mt; info[1] = hd.n_nt; info[2] = hd.n_strips; info[3] = hd.hs; info[4] = p.nslot; info[5] = p.nkmax; info[6] = p.lds_bytes;
        info[7] = (uint32_t)p.blk.size();
    }
    if (!src || !dst) return 1;
    auto f16 = [](uint32_t hbits) -> double {
        const int sgn = (hbits & 0x8000u) ? -1 : 1, e = (hbits >> 10) & 31, m = hbits & 0x3ff;
        return e == 0 ? sgn * ldexp((double)m, -24) : sgn * ldexp((double)(m | 0x400), e - 25);
    };
    const uint32_t rowbytes = sw * cs, nout = rw * cs;
    const uint32_t sh_ = hd.hs - 6u, slo = sh_ - 8u, s4 = 32u - sh_, s3 = 24u - sh_;
    const int32_t rnd = 1 << (slo - 1u), round_add = -132112384 + (1 << 19);
    std::vector<int32_t> iv((size_t)16 * rowbytes); // the M-tile's intermediate rows: 2^22 + round((value - 128) * 2^14)
    for (uint32_t mt = 0; mt < hd.n_mt; ++mt) {
        const WtMTile m = mts[mt];
        const uint32_t *ops = blk + m.ops;
        for (uint32_t n = 0; n < 16; ++n) {
            // the row's vertical weights, decoded from the operands: lane 16 g + n, K-step k, word jj / 2, half jj & 1, three terms
            std::vector<double> w(32u * m.nk, 0.0);
            for (uint32_t k = 0; k < m.nk; ++k)
                for (uint32_t g = 0; g < 4; ++g)
                    for (uint32_t jj = 0; jj < 8; ++jj)
                        for (uint32_t t = 3; t-- > 0;)
                            w[32u * k + 8u * g + jj] += f16((ops[((k * 3u + t) * 64u + 16u * g + n) * 4u + jj / 2u] >> (16u * (jj & 1u))) & 0xffffu);
            for (uint32_t col = 0; col < rowbytes; ++col) {
                double s = 0.0;
                for (uint32_t r = 0; r < 32u * m.nk; ++r) {
                    if (w[r] == 0.0) continue;
                    const uint32_t row = std::min(m.kr0 + r, sh - 1u);
                    s += w[r] * (double)src[(size_t)row * rowbytes + col];
                }
                // sums = value * 2^-9 in the kernel (weights x 2^15, bytes x 2^-24); fmaf(sum, 2^23, 1.25 * 2^23) rounds to nearest even
                const double x = s / 32768.0;
                iv[(size_t)n * rowbytes + col] = (int32_t)(4194304.0 + nearbyint((x - 128.0) * 16384.0));
            }
        }
        for (uint32_t nt = 0; nt < hd.n_nt; ++nt) {
            const WtNTile t = nts[nt];
            const int8_t *hb = reinterpret_cast<const int8_t *>(blk + t.ops);
            for (uint32_t n = 0; n < 16; ++n) {
                const uint32_t o = 16u * nt + n;
                if (o >= nout) continue;
                for (uint32_t rr = 0; rr < 16; ++rr) {
                    const uint32_t y = 16u * mt + rr;
                    if (y >= rh) continue;
                    int64_t L[5] = {0, rnd, 0, 0, 0};
                    for (uint32_t k = 0; k < t.nk; ++k)
                        for (uint32_t g = 0; g < 4; ++g)
                            for (uint32_t jj = 0; jj < 16; ++jj) {
                                const uint32_t col = t.kc0 + 64u * k + 16u * g + jj;
                                const int32_t d2 = hb[(((size_t)k * 3 + 0) * 64 + 16 * g + n) * 16 + jj], d1 = hb[(((size_t)k * 3 + 1) * 64 + 16 * g + n) * 16 + jj],
                                              d0 = hb[(((size_t)k * 3 + 2) * 64 + 16 * g + n) * 16 + jj];
                                if (!(d2 | d1 | d0)) continue;
                                if (col >= rowbytes) return 0; // a weight on a byte that does not exist: a table bug
                                const int32_t x = iv[(size_t)rr * rowbytes + col];
                                const int32_t a2 = x >> 16, a1 = (int32_t)((x >> 8) & 255) - 128, a0 = (int32_t)(x & 255) - 128;
                                L[4] += (int64_t)a2 * d2; L[3] += (int64_t)a2 * d1 + (int64_t)a1 * d2; L[2] += (int64_t)a2 * d0 + (int64_t)a1 * d1 + (int64_t)a0 * d2;
                                L[1] += (int64_t)a1 * d0 + (int64_t)a0 * d1; L[0] += (int64_t)a0 * d0;
                            }
                    const int32_t low = (int32_t)(((L[2] * 256 + L[1]) + (L[0] >> 8)) >> slo); // (the device shifts; a negative left operand is only UB on paper before C++20)
                    const int32_t pp = (int32_t)(((uint32_t)L[4] << s4) + ((uint32_t)L[3] << s3) + (uint32_t)low);
                    const int32_t xx = pp + round_add;
                    dst[(size_t)y * nout + o] = (uint8_t)((uint32_t)std::min(std::max(xx, 0), (256 << 20) - 1) >> 20);
                }
            }
        }
    }
    (void)strips;
    return 1;
}
