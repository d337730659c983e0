// fl_mfma_tables.cpp -- host-side tables of the matrix-pipe resample kernel (layout: fl_mfma.h).
// The weights themselves come from build_axis (image 0.25.6 imageops/sample.rs, evaluated in f32 as the reference does);
// this file only re-arranges them as MFMA operands.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>

#include "fl_mfma.h"
#include "fl_wtile.h"

namespace fl {

namespace {

// IEEE binary16, round to nearest even (host side; the device consumes the bits)
uint16_t f16_bits(double v)
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
        const int shift = 14 - e; // 14..24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r; // may carry into the exponent: still the right value
    return (uint16_t)(sign | r);
}

double f16_value(uint16_t h)
{
    const int s = (h & 0x8000u) ? -1 : 1;
    const int e = (h >> 10) & 31;
    const int m = h & 0x3ff;
    if (e == 0) return s * ldexp((double)m, -24);
    return s * ldexp((double)(m | 0x400), e - 25);
}

} // namespace

void build_mfma_plan(const HostAxis &v, const HostAxis &h, uint32_t cs, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, HostMfmaPlan &out, uint32_t max_outputs,
                     MfmaArith arith)
{
    out = HostMfmaPlan();
    out.arith = arith;
    const bool full = arith == MFMA_ARITH_FULL;
    const uint32_t NTERM = full ? 3u : 2u;  // f16 terms per vertical weight
    const uint32_t NDIG = full ? 3u : 2u;   // signed byte digits per horizontal weight
    if (cs < 1 || cs > 4) return;
    if (cw == 0 || ch == 0 || cy + ch > v.out_size || cx + cw > h.out_size) return;
    const uint32_t sh = v.in_size, sw = h.in_size;
    // ---- vertical: tiles of 16 output rows, K-blocks of 32 source rows --------------------------------------------
    const uint32_t NT = (ch + 15u) / 16u, NKB = (sh + kMfmaKRows - 1u) / kMfmaKRows;
    if (NT >= 0xffffu) return;
    std::vector<uint32_t> A(NT), B(NT);
    for (uint32_t j = 0; j < NT; ++j) {
        uint32_t a = 0xffffffffu, b = 0;
        for (uint32_t n = 0; n < 16 && 16 * j + n < ch; ++n) {
            const uint32_t oy = cy + 16 * j + n;
            a = std::min(a, v.left[oy]);
            b = std::max(b, v.left[oy] + v.count[oy]);
        }
        A[j] = a; B[j] = b;
        if (j && (A[j] < A[j - 1] || B[j] < B[j - 1])) return; // windows must move down monotonically
    }
    out.tiles.resize(NT);
    for (uint32_t j = 0; j < NT; ++j) out.tiles[j] = {A[j] / kMfmaKRows, (B[j] - 1u) / kMfmaKRows};
    // one tile finishes per K-block -- except that a short last tile may end in the same K-block as its predecessor (both
    // windows are cut off by the picture's last row): the kernel then takes one more pass, on the all-zero K-block, for it
    for (uint32_t j = 0; j + 1 < NT; ++j)
        if (out.tiles[j + 1].kb_last == out.tiles[j].kb_last) {
            if (j + 2 != NT) return;
            out.tail = 1;
        }
    for (uint32_t j = 0; j + 2 < NT; ++j)
        if (out.tiles[j + 2].kb_first <= out.tiles[j].kb_last) return; // the accumulator set of tile j is free again before tile j + 2 starts
    // (one K-block more than the picture has: all-zero weights, for the kernel's passes after the last rows)
    out.vmeta.assign(NKB + 1u, 0xffffu);
    out.vw.assign((size_t)(NKB + 1u) * 2 * NTERM * 64 * 4, 0u);
    for (uint32_t j = 0; j < NT; ++j) {
        for (uint32_t s = out.tiles[j].kb_first; s <= out.tiles[j].kb_last; ++s) {
            // accumulator set of tile j during K-block s: 0 if it is the older (lower) of the K-block's live tiles, 1 if the tile
            // before it is still alive -- when that one completes, the kernel moves set 1 to set 0
            const uint32_t set = (j > 0 && out.tiles[j - 1].kb_last >= s) ? 1u : 0u;
            out.vmeta[s] |= 1u << (16 + set);
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t g = lane >> 4, n = lane & 15u;
                if (16 * j + n >= ch) continue;
                const uint32_t oy = cy + 16 * j + n;
                for (uint32_t jj = 0; jj < 8; ++jj) {
                    const uint32_t r = kMfmaKRows * s + 8 * g + jj;
                    if (r < v.left[oy] || r >= v.left[oy] + v.count[oy]) continue;
                    // packed: 256 w as two f16 terms (22-23 bits); full: 2^15 w as three (hi + mid + lo is the f32 weight itself
                    // for every |w| >= 2^-16; smaller ones are exact to 2^-39 absolute: f16 subnormals end at 2^-24)
                    const double w = ldexp((double)v.weights[v.woff[oy] + (r - v.left[oy])], (int)(full ? kMfmaVScaleLog2Full : kMfmaVScaleLog2));
                    double rest = w;
                    const size_t base = ((((size_t)s * 2 + set) * NTERM) * 64 + lane) * 4 + jj / 2;
                    for (uint32_t t = 0; t < NTERM; ++t) {
                        const uint16_t wt = f16_bits(rest);
                        if ((wt & 0x7c00u) == 0x7c00u) { out = HostMfmaPlan(); return; } // (a weight of 2 or more: not a down-scale)
                        rest -= f16_value(wt);
                        out.vw[base + (size_t)t * 64 * 4] |= (uint32_t)wt << (16 * (jj & 1u));
                    }
                }
            }
        }
        {   // the K-block that completes the tile (the extra, all-zero one for a last tile that ends with its predecessor)
            uint32_t &m = out.vmeta[(out.tail && j + 1 == NT) ? NKB : out.tiles[j].kb_last];
            m = (m & 0xffff0000u) | j;
        }
    }
    out.ntiles = NT; out.nkb = NKB; out.y0 = cy; out.rows = ch;

    // ---- horizontal: strips of <= 2048 source bytes -----------------------------------------------------------------
    float maxw = 0.0f;
    for (uint32_t x = cx; x < cx + cw; ++x)
        for (uint32_t k = 0; k < h.count[x]; ++k) maxw = std::max(maxw, fabsf(h.weights[h.woff[x] + k]));
    // fixed-point scale of the horizontal weights: the largest that keeps every weight inside NDIG balanced byte digits
    // (packed: 14..17 bits; full: 24 or not at all -- no weight reaches 1/2 for any ratio the kernel takes, and the kernel's shifts are
    // compile-time constants)
    const int hs_max = full ? 24 : 17, hs_min = full ? 24 : 14;
    const double qlimit = full ? 8355711.0 : 32639.0; // 127 * 256^(NDIG-1) + ... + 127
    int hs = hs_max;
    while (hs >= hs_min && ldexp((double)maxw, hs) > qlimit - 640.0) --hs;
    if (hs < hs_min) return;
    // Fewest strips that fit (each <= kMfmaStripBytes of source per row incl. its 16-byte alignment, <= max_outputs outputs), taken
    // greedily from the left under a cap on the pixels per strip; the smallest cap that still gives that count evens them out.
    // (Equal strips, round 2's rule, needed a fourth strip for 1080p -> 256 columns: the edge strips have one halo, the inner ones
    // two, so 88 + 80 + 88 fits where 86 + 85 + 85 does not.  The result does not depend on the split: the horizontal sums are
    // exact integers.)
    // Where a strip starts in the row.  The kernel moves 16-byte pieces, so any multiple of 16 works -- but a wave's 256-byte row piece that
    // does not start on a 128-byte line touches three lines instead of two, and the load path alone runs 8 % slower on such strips
    // (tools/microbench/align_probe.hip, profiles/r05_align_probe.txt: 1.139 against 1.049 ms for the flagship's strips; splitting the
    // requests at the line boundary instead buys back a quarter of that).  Rows that start on a line (pitch a multiple of 128: 1080p
    // Rgb8 is 45 lines) get line-aligned strips; the up to 112 bytes a strip loses on its left are part of the fit test below.
    // (only where that costs no extra strip: a strip is a whole walk over the picture's rows)
    uint32_t al = (cs * sw) % 128u == 0u ? 128u : 16u;
    const uint32_t max_px = max_outputs / cs;
    std::vector<uint32_t> bounds; // x0 of every strip, then cx + cw
    auto split = [&](uint32_t cap, std::vector<uint32_t> *b) -> uint32_t {
        uint32_t n = 0;
        if (b) b->clear();
        for (uint32_t x0 = cx; x0 < cx + cw; ++n) {
            uint32_t L = h.left[x0], R = h.left[x0] + h.count[x0], x1 = x0;
            while (x1 < cx + cw && x1 - x0 < cap) {
                const uint32_t L2 = std::min(L, h.left[x1]), R2 = std::max(R, h.left[x1] + h.count[x1]);
                if (cs * R2 - (cs * L2) / al * al > kMfmaStripBytes) break;
                L = L2; R = R2; ++x1;
            }
            if (x1 == x0) return 0; // one output column alone does not fit
            if (b) b->push_back(x0);
            x0 = x1;
        }
        if (b) b->push_back(cx + cw);
        return n;
    };
    if (!max_px) return;
    uint32_t ns = split(max_px, nullptr);
    if (al > 16u) {
        al = 16u;
        const uint32_t ns16 = split(max_px, nullptr);
        if (ns && ns <= ns16) al = 128u; else ns = ns16;
    }
    if (!ns) return;
    uint32_t cap = (cw + ns - 1u) / ns;
    while (cap < max_px && split(cap, nullptr) != ns) ++cap;
    if (split(cap, &bounds) != ns) return;
    // quantised weights, one vector per output column; the largest tap absorbs the rounding so that the sum is exactly 2^hs
    std::vector<std::vector<int32_t>> hq(cw);
    for (uint32_t x = cx; x < cx + cw; ++x) {
        std::vector<int32_t> &q = hq[x - cx];
        q.resize(h.count[x]);
        int64_t sum = 0;
        uint32_t big = 0;
        for (uint32_t k = 0; k < h.count[x]; ++k) {
            q[k] = (int32_t)llround(ldexp((double)h.weights[h.woff[x] + k], hs));
            sum += q[k];
            if (abs(q[k]) > abs(q[big])) big = k;
        }
        q[big] += (int32_t)(((int64_t)1 << hs) - sum);
        if ((double)abs(q[big]) > qlimit) return;
    }
    for (size_t si = 0; si + 1 < bounds.size(); ++si) {
        HostMfmaPlan::Strip S;
        const uint32_t x0 = bounds[si], x1 = bounds[si + 1];
        uint32_t L = 0xffffffffu;
        for (uint32_t x = x0; x < x1; ++x) L = std::min(L, h.left[x]);
        S.hdr.x0 = x0; S.hdr.x1 = x1; S.hdr.byte0 = (cs * L) / al * al; S.hdr.nout = (x1 - x0) * cs; S.hdr.hs = (uint32_t)hs; S.hdr.slots = 2;
        const int32_t nout = (int32_t)S.hdr.nout;
        // operand 0 is all zeros: the tile slots a chunk does not need multiply by it and add into the dummy column, which keeps
        // the kernel's horizontal stage free of branches (12 matrix instructions back to back per chunk instead of 4 + a wait)
        const uint32_t CE = 1u + NDIG; // words per tile slot: first output, then one operand index per digit (high digit first)
        S.ctab.assign(kMfmaWaves * 4 * 3 * CE, 0);
        for (size_t k = 0; k < S.ctab.size(); k += CE) S.ctab[k] = 0x40000000;
        std::map<std::string, uint32_t> seen;
        seen.emplace(std::string(1024, '\0'), 0u);
        S.ops.assign(256, 0u);
        auto weight_of = [&](int32_t o, uint32_t col) -> int32_t { // weight of strip byte column `col` in output o
            if (o < 0 || o >= nout) return 0;
            const uint32_t abs_b = S.hdr.byte0 + col;
            if (abs_b >= cs * sw) return 0;
            const uint32_t px = abs_b / cs, chn = abs_b % cs, x = x0 + (uint32_t)o / cs;
            if ((uint32_t)o % cs != chn || px < h.left[x] || px >= h.left[x] + h.count[x]) return 0;
            return hq[x - cx][px - h.left[x]];
        };
        for (uint32_t w = 0; w < kMfmaWaves; ++w)
            for (uint32_t c = 0; c < 4; ++c) {
                const uint32_t col0 = kMfmaWaveCols * w + 64u * c;
                int32_t omin = 0x7fffffff, omax = -1;
                for (int32_t o = 0; o < nout; ++o)
                    for (uint32_t col = col0; col < col0 + 64u; ++col)
                        if (weight_of(o, col) != 0) { omin = std::min(omin, o); omax = std::max(omax, o); break; }
                if (omax < 0) continue;
                const uint32_t nt = (uint32_t)(omax - omin) / 16u + 1u;
                if (nt > 3) return;
                S.hdr.slots = std::max(S.hdr.slots, std::max(nt, 2u));
                for (uint32_t t = 0; t < nt; ++t) {
                    const int32_t base = omin + 16 * (int32_t)t;
                    std::string op[3] = {std::string(1024, '\0'), std::string(1024, '\0'), std::string(1024, '\0')};
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const uint32_t g = lane >> 4, n = lane & 15u;
                        for (uint32_t a = 0; a < 4; ++a)
                            for (uint32_t r = 0; r < 4; ++r) {
                                // balanced digits, low to high: q = sum d_k 256^k with every d_k in [-128, 127]
                                int32_t q = weight_of(base + (int32_t)n, col0 + 16u * a + 4u * g + r);
                                for (uint32_t d = NDIG; d-- > 0;) {
                                    const int32_t lo = d ? ((q + 128) & 255) - 128 : q;
                                    op[d][lane * 16 + 4 * a + r] = (char)(int8_t)lo;
                                    q = (q - lo) / 256;
                                }
                            }
                    }
                    int32_t *e = &S.ctab[((w * 4 + c) * 3 + t) * CE];
                    e[0] = base;
                    for (uint32_t d = 0; d < NDIG; ++d) {
                        auto it = seen.find(op[d]);
                        if (it == seen.end()) {
                            it = seen.emplace(op[d], (uint32_t)seen.size()).first;
                            S.ops.resize((size_t)seen.size() * 256);
                            memcpy(S.ops.data() + (size_t)it->second * 256, op[d].data(), 1024);
                        }
                        e[1 + d] = (int32_t)it->second;
                    }
                }
            }
        S.hdr.n_ops = (uint32_t)seen.size();
        out.strips.push_back(std::move(S));
    }
    out.ok = true;
}

// The plan a geometry runs with.  Strips of up to kMfmaMaxStripOutputs outputs leave LDS room for the horizontal operands; where
// that bound -- not the 2048 source bytes -- is what cuts the picture into strips (ratios below ~4.7 for Rgb8: 1080p -> 480, 512,
// 640 columns), the wide layout (kMfmaMaxStripOutputsWide outputs per strip, operands read from the L2) needs fewer strips, and a
// strip is a whole walk over the picture's rows by one workgroup, whatever its width.
void choose_mfma_plan(const HostAxis &v, const HostAxis &h, uint32_t cs, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, HostMfmaPlan &out, MfmaArith arith)
{
    build_mfma_plan(v, h, cs, cx, cy, cw, ch, out, kMfmaMaxStripOutputs, arith);
    out.wide = false;
    // (1- and 2-channel sources keep the narrow layout: a wide row is up to 728 pixels there, and the conversion's 24 sums per
    // lane pushed hipcc into scratch -- whose reloads wait on vmcnt, i.e. for the K-block in flight)
    if (!out.ok || out.strips.size() < 2 || cw * cs <= kMfmaMaxStripOutputs || cs < 3) return;
    HostMfmaPlan w;
    build_mfma_plan(v, h, cs, cx, cy, cw, ch, w, kMfmaMaxStripOutputsWide, arith);
    if (w.ok && w.strips.size() < out.strips.size()) {
        out = std::move(w);
        out.wide = true;
    }
}

// ---- the window-tile kernel's tables (fl_wtile.h) -------------------------------------------------------------------------------
void build_wtile_plan(const HostAxis &v, const HostAxis &h, uint32_t cs, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, HostWtPlan &out)
{
    out = HostWtPlan();
    if (cs < 1 || cs > 4 || cw == 0 || ch == 0 || cy + ch > v.out_size || cx + cw > h.out_size) return;
    const uint32_t sh = v.in_size, sw = h.in_size, nout = cw * cs;
    const uint32_t NMT = (ch + 15u) / 16u, NNT = (nout + 15u) / 16u;
    if (sh >= 65536u || sw * cs >= (1u << 20)) return;
    std::vector<uint32_t> ops;                 // operand words, deduplicated per tile (a blur's interior tiles all share one block)
    std::map<std::string, uint32_t> seen;      // block bytes -> word offset inside `ops`
    auto intern = [&](const std::vector<uint32_t> &b) -> uint32_t {
        std::string key(reinterpret_cast<const char *>(b.data()), b.size() * 4);
        auto it = seen.find(key);
        if (it != seen.end()) return it->second;
        const uint32_t off = (uint32_t)ops.size();
        ops.insert(ops.end(), b.begin(), b.end());
        seen.emplace(std::move(key), off);
        return off;
    };
    // ---- vertical: per M-tile a K window of whole 32-row steps starting at a multiple of 8 rows ----
    std::vector<WtMTile> mts(NMT);
    uint32_t nkv_max = 0, prev_kr0 = 0;
    for (uint32_t j = 0; j < NMT; ++j) {
        uint32_t a = 0xffffffffu, b = 0;
        for (uint32_t n = 0; n < 16 && 16 * j + n < ch; ++n) {
            const uint32_t oy = cy + 16 * j + n;
            a = std::min(a, v.left[oy]);
            b = std::max(b, v.left[oy] + v.count[oy]);
        }
        const uint32_t kr0 = a & ~7u, nk = (b - kr0 + kMfmaKRows - 1u) / kMfmaKRows;
        if (nk == 0 || nk > kWtMaxKV) return;
        if (j && kr0 < prev_kr0) return; // the ring only moves down (a window may END earlier than the one before it -- the picture's last rows: nothing new to load then)
        prev_kr0 = kr0;
        std::vector<uint32_t> blk((size_t)nk * 3 * 64 * 4, 0u);
        for (uint32_t k = 0; k < nk; ++k)
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t g = lane >> 4, n = lane & 15u;
                if (16 * j + n >= ch) continue;
                const uint32_t oy = cy + 16 * j + n;
                for (uint32_t jj = 0; jj < 8; ++jj) {
                    const uint32_t r = kr0 + kMfmaKRows * k + 8 * g + jj;
                    if (r < v.left[oy] || r >= v.left[oy] + v.count[oy]) continue;
                    double rest = ldexp((double)v.weights[v.woff[oy] + (r - v.left[oy])], (int)kMfmaVScaleLog2Full);
                    for (uint32_t t = 0; t < 3; ++t) {
                        const uint16_t wt = f16_bits(rest);
                        if ((wt & 0x7c00u) == 0x7c00u) return; // a weight of 2 or more
                        rest -= f16_value(wt);
                        blk[(((size_t)k * 3 + t) * 64 + lane) * 4 + jj / 2] |= (uint32_t)wt << (16 * (jj & 1u));
                    }
                }
            }
        mts[j] = {kr0, nk, intern(blk), 0u};
        nkv_max = std::max(nkv_max, nk);
    }
    // ---- horizontal weights: fixed point at the finest of 2^-24 / 2^-23 / 2^-22 whose three balanced digits hold the largest ----
    float maxw = 0.0f;
    for (uint32_t x = cx; x < cx + cw; ++x)
        for (uint32_t k = 0; k < h.count[x]; ++k) maxw = std::max(maxw, fabsf(h.weights[h.woff[x] + k]));
    const double qlimit = 8355711.0; // 127 * 65536 + 127 * 256 + 127
    int hs = 24;
    while (hs >= 22 && ldexp((double)maxw, hs) > qlimit - 640.0) --hs;
    if (hs < 22) return;
    std::vector<std::vector<int32_t>> hq(cw);
    for (uint32_t x = cx; x < cx + cw; ++x) {
        std::vector<int32_t> &q = hq[x - cx];
        q.resize(h.count[x]);
        int64_t sum = 0;
        uint32_t big = 0;
        for (uint32_t k = 0; k < h.count[x]; ++k) {
            q[k] = (int32_t)llround(ldexp((double)h.weights[h.woff[x] + k], hs));
            sum += q[k];
            if (abs(q[k]) > abs(q[big])) big = k;
        }
        q[big] += (int32_t)(((int64_t)1 << hs) - sum);
        if ((double)abs(q[big]) > qlimit) return;
    }
    auto weight_of = [&](uint32_t o, uint32_t col) -> int32_t { // weight of source byte `col` of a row in output byte o
        if (o >= nout || col >= cs * sw) return 0;
        const uint32_t px = col / cs, x = cx + o / cs;
        if (o % cs != col % cs || px < h.left[x] || px >= h.left[x] + h.count[x]) return 0;
        return hq[x - cx][px - h.left[x]];
    };
    std::vector<WtNTile> nts(NNT);
    uint32_t nkh_max = 0;
    for (uint32_t j = 0; j < NNT; ++j) {
        uint32_t a = 0xffffffffu, b = 0;
        for (uint32_t n = 0; n < 16 && 16 * j + n < nout; ++n) {
            const uint32_t o = 16 * j + n, x = cx + o / cs;
            a = std::min(a, cs * h.left[x] + o % cs);
            b = std::max(b, cs * (h.left[x] + h.count[x] - 1u) + o % cs + 1u);
        }
        const uint32_t kc0 = a & ~15u, nk = (b - kc0 + 63u) / 64u;
        if (nk == 0 || nk > kWtMaxKH) return;
        if (j && kc0 < nts[j - 1].kc0) return;
        std::vector<uint32_t> blk((size_t)nk * 3 * 64 * 4, 0u);
        uint8_t *bytes = reinterpret_cast<uint8_t *>(blk.data());
        for (uint32_t k = 0; k < nk; ++k)
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t g = lane >> 4, n = lane & 15u;
                for (uint32_t jj = 0; jj < 16; ++jj) {
                    int32_t q = weight_of(16 * j + n, kc0 + 64u * k + 16u * g + jj);
                    for (uint32_t d = 3; d-- > 0;) { // balanced digits, low to high; operand 0 holds the highest
                        const int32_t lo = d ? ((q + 128) & 255) - 128 : q;
                        bytes[((((size_t)k * 3 + d) * 64 + lane) * 16) + jj] = (uint8_t)(int8_t)lo;
                        q = (q - lo) / 256;
                    }
                }
            }
        nts[j] = {kc0, nk, intern(blk), 0u};
        nkh_max = std::max(nkh_max, nk);
    }
    // ---- strips: as many N-tiles as the LDS and the wave's operand registers allow ----
    uint32_t ring_rows = 0;
    for (auto &m : mts) ring_rows = std::max(ring_rows, 32u * m.nk);
    std::map<uint32_t, uint32_t> freq;
    for (auto &t : nts) freq[t.ops]++;
    uint32_t common_n = 0;
    for (auto &kv : freq) common_n = std::max(common_n, kv.second);
    const bool uniform = NNT >= 4 && common_n * 2 >= NNT; // most column tiles share their operands: one register set serves them
    // register split of the kernel instantiation: 6 x 1, 3 x 2, 2 x 3 or 1 x 6 (N-tiles per wave x K-steps)
    uint32_t nkmax = nkh_max <= 3u ? nkh_max : kWtOperandRegs;
    if (uniform) nkmax = kWtOperandRegs;
    else if (nkh_max > kWtOperandRegs) return; // (tiles of 7-8 K-steps that share nothing: not a geometry this kernel is for)
    const uint32_t nslot = kWtOperandRegs / nkmax;
    const uint32_t step = cs == 3 ? 3u : 1u;     // a strip starts on a pixel boundary: 16 n0 must be a multiple of cs
    const uint32_t tn_regs = uniform ? 0xffffu : kWtWaves * nslot;
    auto strip_of = [&](uint32_t n0, uint32_t n1) {
        WtStrip S{};
        S.n0 = n0; S.n1 = n1; S.col0 = nts[n0].kc0;
        uint32_t end = 0;
        for (uint32_t j = n0; j < n1; ++j) end = std::max(end, nts[j].kc0 + 64u * nts[j].nk);
        uint32_t spw = (end - S.col0 + 15u) / 16u;
        if (!(spw & 1u)) ++spw;
        S.sp = 16u * spw;
        S.lds_bytes = wt_lds_bytes(ring_rows, S.sp, nkv_max, n1 - n0);
        return S;
    };
    auto split = [&](uint32_t cap, std::vector<WtStrip> *b) -> uint32_t {
        uint32_t n = 0;
        if (b) b->clear();
        for (uint32_t n0 = 0; n0 < NNT; ++n) {
            uint32_t n1 = n0;
            while (n1 < NNT) {
                const uint32_t t = std::min(NNT, n1 + step);
                if (t - n0 > cap || t - n0 > tn_regs || strip_of(n0, t).lds_bytes > kWtLdsBudget) break;
                n1 = t;
            }
            if (n1 == n0) return 0;
            if (b) b->push_back(strip_of(n0, n1));
            n0 = n1;
        }
        return n;
    };
    const uint32_t ns = split(0xffffu, nullptr);
    if (!ns) return;
    uint32_t cap = ((NNT + ns - 1u) / ns + step - 1u) / step * step;
    while (split(cap, nullptr) != ns) cap += step;
    std::vector<WtStrip> strips;
    if (split(cap, &strips) != ns) return;
    uint32_t lds_max = 0;
    for (auto &S : strips) {
        std::map<uint32_t, uint32_t> f;
        for (uint32_t j = S.n0; j < S.n1; ++j) f[nts[j].ops]++;
        S.common_ops = 0xffffffffu;
        uint32_t best = 1;
        for (auto &kv : f) if (kv.second > best) { best = kv.second; S.common_ops = kv.first; }
        for (uint32_t j = S.n0; j < S.n1; ++j) if (nts[j].ops == S.common_ops) S.common_nk = nts[j].nk;
        lds_max = std::max(lds_max, S.lds_bytes);
    }
    // ---- the block ----
    WtHeader hd{};
    hd.n_mt = NMT; hd.n_nt = NNT; hd.n_strips = (uint32_t)strips.size(); hd.cs = cs;
    hd.rows = ch; hd.nout = nout; hd.src_rows = sh; hd.src_rowbytes = sw * cs; hd.hs = (uint32_t)hs;
    hd.ring_rows = ring_rows; hd.ring_magic = (uint32_t)((((uint64_t)1 << 32) + ring_rows - 1u) / ring_rows);
    hd.nkv_max = nkv_max; hd.nkh_max = nkh_max;
    const uint32_t hw = sizeof(WtHeader) / 4;
    hd.mt_off = hw;
    hd.nt_off = hd.mt_off + NMT * (uint32_t)(sizeof(WtMTile) / 4);
    hd.strip_off = hd.nt_off + NNT * (uint32_t)(sizeof(WtNTile) / 4);
    const uint32_t ops_off = (hd.strip_off + hd.n_strips * (uint32_t)(sizeof(WtStrip) / 4) + 3u) & ~3u; // operands are read 16 bytes at a time
    for (auto &m : mts) m.ops += ops_off;
    for (auto &t : nts) t.ops += ops_off;
    for (auto &S : strips) if (S.common_ops != 0xffffffffu) S.common_ops += ops_off;
    out.blk.assign((size_t)ops_off + ops.size(), 0u);
    memcpy(out.blk.data(), &hd, sizeof(hd));
    memcpy(out.blk.data() + hd.mt_off, mts.data(), mts.size() * sizeof(WtMTile));
    memcpy(out.blk.data() + hd.nt_off, nts.data(), nts.size() * sizeof(WtNTile));
    memcpy(out.blk.data() + hd.strip_off, strips.data(), strips.size() * sizeof(WtStrip));
    memcpy(out.blk.data() + ops_off, ops.data(), ops.size() * 4);
    out.nslot = nslot; out.nkmax = nkmax; out.n_mt = NMT; out.n_strips = hd.n_strips; out.lds_bytes = lds_max;
    out.ok = true;
}

} // namespace fl
