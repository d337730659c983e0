// fl_tables.cpp -- see fl_tables.h.  Compiled with -ffp-contract=off: the
// window bounds depend on exact f32 evaluation order (floor/ceil of
// (o + 0.5) * ratio -/+ support), as in image 0.25.6 imageops/sample.rs.
#include "fl_tables.h"
#include "fl_jpeg_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>

namespace fl {

namespace {

constexpr float kPi = 3.14159265358979323846f; // f32::consts::PI

// sample.rs sinc(): a = t * PI; t == 0 ? 1 : sin(a) / a
inline float sinc(float t)
{
    const float a = t * kPi;
    return t == 0.0f ? 1.0f : sinf(a) / a;
}

// sample.rs lanczos3_kernel(x) = lanczos(x, 3.0)
inline float lanczos3(float x) { return fabsf(x) < 3.0f ? sinc(x) * sinc(x / 3.0f) : 0.0f; }

// sample.rs gaussian(x, r) = ((2 PI).sqrt() * r).recip() * (-x.powi(2) / (2.0 * r.powi(2))).exp()
inline float gaussian(float x, float r)
{
    const float norm = 1.0f / (sqrtf(2.0f * kPi) * r);
    return norm * expf(-(x * x) / (2.0f * (r * r)));
}

inline int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

} // namespace

void build_axis(uint32_t in_size, uint32_t out_size, Filter filter, float sigma, HostAxis &out)
{
    out = HostAxis();
    out.in_size = in_size;
    out.out_size = out_size;
    out.left.resize(out_size);
    out.count.resize(out_size);
    out.woff.resize(out_size);
    const float support = filter == FILTER_GAUSSIAN ? 2.0f * sigma : 3.0f;
    const float ratio = (float)in_size / (float)out_size;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = support * sratio;
    out.weights.reserve((size_t)out_size * (size_t)(2.0f * src_support + 3.0f));
    for (uint32_t o = 0; o < out_size; ++o) {
        float input = ((float)o + 0.5f) * ratio;
        int64_t left = (int64_t)floorf(input - src_support);
        left = clamp64(left, 0, (int64_t)in_size - 1);
        int64_t right = (int64_t)ceilf(input + src_support);
        right = clamp64(right, left + 1, (int64_t)in_size);
        input = input - 0.5f;
        const size_t base = out.weights.size();
        float sum = 0.0f;
        for (int64_t i = left; i < right; ++i) {
            const float x = ((float)i - input) / sratio;
            const float w = filter == FILTER_GAUSSIAN ? gaussian(x, sigma) : lanczos3(x);
            out.weights.push_back(w);
            sum += w;
        }
        for (size_t k = base; k < out.weights.size(); ++k) out.weights[k] /= sum;
        out.left[o] = (uint32_t)left;
        out.count[o] = (uint32_t)(right - left);
        out.woff[o] = (uint32_t)base;
        out.max_taps = std::max(out.max_taps, out.count[o]);
    }
}

void resize_dimensions(uint32_t width, uint32_t height, uint32_t nwidth, uint32_t nheight, bool fill, uint32_t &ow,
                       uint32_t &oh)
{
    const double wratio = (double)nwidth / (double)width;
    const double hratio = (double)nheight / (double)height;
    const double ratio = fill ? std::max(wratio, hratio) : std::min(wratio, hratio);
    const uint64_t nw = std::max<uint64_t>((uint64_t)round((double)width * ratio), 1);
    const uint64_t nh = std::max<uint64_t>((uint64_t)round((double)height * ratio), 1);
    if (nw > (uint64_t)UINT32_MAX) {
        const double r = (double)UINT32_MAX / (double)width;
        ow = UINT32_MAX;
        oh = std::max<uint32_t>((uint32_t)round((double)height * r), 1);
    } else if (nh > (uint64_t)UINT32_MAX) {
        const double r = (double)UINT32_MAX / (double)height;
        ow = std::max<uint32_t>((uint32_t)round((double)width * r), 1);
        oh = UINT32_MAX;
    } else {
        ow = (uint32_t)nw;
        oh = (uint32_t)nh;
    }
}

bool build_row_sched(const HostAxis &v, uint32_t y0, uint32_t y1, uint32_t nacc, uint32_t block, uint32_t &r0,
                     uint32_t &r1, std::vector<RowSched> &out)
{
    out.clear();
    if (y0 >= y1 || y1 > v.out_size || nacc < 1 || nacc > (uint32_t)NACC || block < 1) return false;
    r0 = v.left[y0];
    r1 = r0;
    for (uint32_t o = y0; o < y1; ++o) {
        // windows must start and end in output order: that is what makes "slot = o mod NACC" collision free
        if (o > y0 && (v.left[o] < v.left[o - 1] || v.left[o] + v.count[o] < v.left[o - 1] + v.count[o - 1])) return false;
        r1 = std::max(r1, v.left[o] + v.count[o]);
    }
    // whole blocks for the kernel loop, whole chunks for its LDS staging of the schedule
    const uint32_t padded = (r1 - r0 + block - 1) / block * block;
    out.assign((padded + SCHED_CHUNK - 1) / SCHED_CHUNK * SCHED_CHUNK, RowSched());
    for (auto &e : out) memset(&e, 0, sizeof(e));
    // deferred flush: output o + nacc must start in a later block than the one in which output o completes
    for (uint32_t o = y0; o + nacc < y1; ++o) {
        const uint32_t done_block = (v.left[o] + v.count[o] - 1 - r0) / block;
        const uint32_t rearm_block = (v.left[o + nacc] - r0) / block;
        if (rearm_block <= done_block) return false;
    }
    for (uint32_t o = y0; o < y1; ++o) {
        const uint32_t slot = o % nacc;
        for (uint32_t i = 0; i < v.count[o]; ++i) {
            RowSched &e = out[v.left[o] + i - r0];
            if (e.live & (1u << slot)) return false; // two live outputs would share a slot
            e.live |= 1u << slot;
            e.w[slot] = v.weights[v.woff[o] + i];
        }
        RowSched &last = out[v.left[o] + v.count[o] - 1 - r0];
        if (!last.emit) last.first_out = o; // outputs complete in order, so the first one seen is the lowest
        last.emit |= 1u << slot;
    }
    // the kernel flushes at block ends: fold every block's emits into its first row
    for (size_t b0 = 0; b0 < out.size(); b0 += block) {
        uint32_t em = 0, first = 0;
        for (size_t k = 0; k < block; ++k) {
            RowSched &e = out[b0 + k];
            if (e.emit && !em) first = e.first_out;
            em |= e.emit;
            e.emit = 0;
            e.first_out = 0;
        }
        out[b0].emit = em;
        out[b0].first_out = first;
    }
    return true;
}

void build_strip(const HostAxis &h, uint32_t x0, uint32_t x1, uint32_t lanes, uint32_t ppl, HostStrip &out)
{
    out = HostStrip();
    out.x0 = x0;
    out.x1 = x1;
    uint32_t lo = UINT32_MAX, hi = 0;
    for (uint32_t x = x0; x < x1; ++x) {
        lo = std::min(lo, h.left[x]);
        hi = std::max(hi, h.left[x] + h.count[x]);
    }
    out.sx0 = lo - (lo % ppl);
    out.sx1 = hi;
    const uint32_t n = x1 - x0;
    // first / last contributing lane of every column
    std::vector<uint32_t> ta(n), tb(n);
    for (uint32_t x = x0; x < x1; ++x) {
        ta[x - x0] = (h.left[x] - out.sx0) / ppl;
        tb[x - x0] = (h.left[x] + h.count[x] - 1 - out.sx0) / ppl;
        out.kmax = std::max(out.kmax, tb[x - x0] - ta[x - x0] + 1);
    }
    out.ks = out.kmax | 1u;
    // columns touched by every lane (windows are monotone, so they form a contiguous range)
    std::vector<uint32_t> xa(lanes, 0), cnt(lanes, 0);
    for (uint32_t t = 0; t < lanes; ++t) {
        uint32_t first = UINT32_MAX, last = 0;
        for (uint32_t xl = 0; xl < n; ++xl)
            if (ta[xl] <= t && t <= tb[xl]) { first = std::min(first, xl); last = xl; }
        if (first != UINT32_MAX) { xa[t] = first; cnt[t] = last - first + 1; }
        out.jmax = std::max(out.jmax, cnt[t]);
    }
    {   // the kernel unrolls its column loop by 4 and finishes odd counts in a remainder loop
#ifdef FL_EXPERIMENT
        const char *pad = getenv("FLGPU_JMAX_PAD"); // experiment builds only: 4 restores the padded tables
        const uint32_t m = pad ? (uint32_t)std::max(1, atoi(pad)) : 1u;
#else
        const uint32_t m = 1u;
#endif
        out.jmax = std::max(1u, (out.jmax + m - 1u) / m * m);
    }
    const uint32_t dummy = n * out.ks * 16u;
    out.wt.assign((size_t)out.jmax * lanes * 4, 0.0f);
    out.po.assign((size_t)out.jmax * lanes, dummy);
    for (uint32_t t = 0; t < lanes; ++t)
        for (uint32_t j = 0; j < cnt[t]; ++j) {
            const uint32_t xl = xa[t] + j, x = x0 + xl;
            if (!(ta[xl] <= t && t <= tb[xl])) continue; // cannot happen for monotone windows
            for (uint32_t p = 0; p < ppl && p < 4; ++p) {
                const int64_t i = (int64_t)out.sx0 + (int64_t)t * ppl + p - (int64_t)h.left[x];
                if (i >= 0 && i < (int64_t)h.count[x]) out.wt[((size_t)j * lanes + t) * 4 + p] = h.weights[h.woff[x] + (uint32_t)i];
            }
            out.po[(size_t)j * lanes + t] = (xl * out.ks + (t - ta[xl])) * 16u;
        }
}

void build_blur_plan(const HostAxis &v, const HostAxis &h, uint32_t nt, uint32_t ty, std::vector<uint32_t> &out)
{
    const uint32_t w = h.out_size, hh = v.out_size, htaps = h.max_taps;
    const uint32_t tw_full = (w + nt - 1) / nt, nb = (hh + ty - 1) / ty;
    uint32_t rv = 0;
    std::vector<uint32_t> tiles(2 * nt, 0u), bands(2 * nb, 0u);
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t x0 = t * tw_full;
        if (x0 >= w) continue;
        const uint32_t x1 = std::min(w, x0 + tw_full) - 1;
        tiles[2 * t] = h.left[x0];
        tiles[2 * t + 1] = h.left[x1] + h.count[x1] - h.left[x0];
    }
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t y0 = b * ty, y1 = std::min(hh, y0 + ty) - 1;
        bands[2 * b] = v.left[y0];
        bands[2 * b + 1] = v.left[y1] + v.count[y1] - v.left[y0];
        rv = std::max(rv, bands[2 * b + 1]);
    }
    BlurPlanHeader hd{};
    hd.nt = nt; hd.nb = nb; hd.tw_full = tw_full; hd.htaps = htaps; hd.rv = rv;
    const uint32_t hw = sizeof(BlurPlanHeader) / 4;
    hd.tiles_off = hw;
    hd.bands_off = hd.tiles_off + 2 * nt;
    hd.vdense_off = (hd.bands_off + 2 * nb + 3u) & ~3u;
    hd.htiles_off = hd.vdense_off + nb * rv * ty;
    // distinct horizontal weight vectors (zero padded to htaps): interior columns all share one
    std::map<std::vector<uint32_t>, uint32_t> ids;
    std::vector<uint32_t> col_row(w);
    std::vector<std::vector<uint32_t>> rows;
    for (uint32_t x = 0; x < w; ++x) {
        std::vector<uint32_t> key(htaps, 0u);
        memcpy(key.data(), h.weights.data() + h.woff[x], h.count[x] * sizeof(float));
        auto it = ids.find(key);
        if (it == ids.end()) { it = ids.emplace(key, (uint32_t)rows.size()).first; rows.push_back(key); }
        col_row[x] = it->second;
    }
    hd.nrows_h = (uint32_t)rows.size();
    hd.hrows_off = hd.htiles_off + nt * tw_full * 2;
    out.assign((size_t)hd.hrows_off + (size_t)htaps * hd.nrows_h, 0u);
    memcpy(out.data(), &hd, sizeof(hd));
    memcpy(out.data() + hd.tiles_off, tiles.data(), tiles.size() * 4);
    memcpy(out.data() + hd.bands_off, bands.data(), bands.size() * 4);
    float *vd = reinterpret_cast<float *>(out.data() + hd.vdense_off);
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t y0 = b * ty, top = bands[2 * b];
        for (uint32_t o = 0; o < ty && y0 + o < hh; ++o) {
            const uint32_t y = y0 + o;
            for (uint32_t i = 0; i < v.count[y]; ++i)
                vd[((size_t)b * rv + (v.left[y] + i - top)) * ty + o] = v.weights[v.woff[y] + i];
        }
    }
    for (uint32_t t = 0; t < nt; ++t) {
        uint32_t *base = out.data() + hd.htiles_off + (size_t)t * tw_full * 2;
        const uint32_t x0 = t * tw_full;
        if (x0 >= w) break;
        const uint32_t cl = h.left[x0];
        for (uint32_t j = 0; j < tw_full && x0 + j < w; ++j) {
            base[2 * j] = h.left[x0 + j] - cl;
            base[2 * j + 1] = col_row[x0 + j];
        }
    }
    uint32_t *hr = out.data() + hd.hrows_off;
    for (uint32_t r = 0; r < hd.nrows_h; ++r)
        for (uint32_t i = 0; i < htaps; ++i) hr[(size_t)i * hd.nrows_h + r] = rows[r][i];
}

void build_webp_gamma(std::vector<uint32_t> &out)
{
    // libwebp src/enc/picture_csp_enc.c: kGamma = 0.80, GAMMA_FIX = 12, GAMMA_TAB_FIX = 7
    const double kGamma = 0.80;
    const int kGammaFix = 12, kGammaTabFix = 7, kGammaTabSize = 1 << (kGammaFix - kGammaTabFix);
    const int kGammaScale = (1 << kGammaFix) - 1;
    const double scale = (double)(1 << kGammaTabFix) / kGammaScale;
    const double norm = 1. / 255.;
    out.clear();
    for (int v = 0; v <= 255; ++v) out.push_back((uint32_t)(uint16_t)(pow(norm * v, kGamma) * kGammaScale + .5));
    for (int v = 0; v <= kGammaTabSize; ++v) out.push_back((uint32_t)(int)(255. * pow(scale * v, 1. / kGamma) + .5));
}

void build_jpeg_tables(uint32_t width, uint32_t height, uint32_t quality, std::vector<uint32_t> &out)
{
    std::vector<uint8_t> b;
    b.reserve(kJpegTableBlockBytes);
    uint8_t q[2][64];
    // "Derive our quantization table scaling value using the libjpeg algorithm"
    uint32_t scale = std::min(std::max(quality, 1u), 100u);
    scale = scale < 50 ? 5000 / scale : 200 - scale * 2;
    for (int i = 0; i < 64; ++i) {
        q[0][i] = (uint8_t)std::min(std::max(((uint32_t)kStdLumaQ[i] * scale + 50) / 100, 1u), 255u);
        q[1][i] = (uint8_t)std::min(std::max(((uint32_t)kStdChromaQ[i] * scale + 50) / 100, 1u), 255u);
    }
    auto segment = [&](uint8_t marker, const std::vector<uint8_t> &d) {
        b.push_back(0xFF); b.push_back(marker);
        b.push_back((uint8_t)((d.size() + 2) >> 8)); b.push_back((uint8_t)(d.size() + 2));
        b.insert(b.end(), d.begin(), d.end());
    };
    b.push_back(0xFF); b.push_back(0xD8);                                                      // SOI
    segment(0xE0, {'J', 'F', 'I', 'F', 0, 1, 2, 0, 0, 1, 0, 1, 0, 0});                         // JFIF 1.2, PixelDensity::default() = 1:1 aspect
    std::vector<uint8_t> d = {8, (uint8_t)(height >> 8), (uint8_t)height, (uint8_t)(width >> 8), (uint8_t)width, 3};
    for (int c = 0; c < 3; ++c) { d.push_back((uint8_t)(c + 1)); d.push_back(0x11); d.push_back((uint8_t)(c ? 1 : 0)); }
    segment(0xC0, d);                                                                          // SOF0
    for (int t = 0; t < 2; ++t) {                                                              // DQT, zig-zag order
        d.assign(1, (uint8_t)t);
        for (int k = 0; k < 64; ++k) d.push_back(q[t][kUnzigzag[k]]);
        segment(0xDB, d);
    }
    const struct { uint8_t cls, dest; const HuffSpec *s; } hts[4] = {{0, 0, &kDcLuma}, {1, 0, &kAcLuma}, {0, 1, &kDcChroma}, {1, 1, &kAcChroma}};
    for (const auto &h : hts) {                                                                // DHT
        d.assign(1, (uint8_t)((h.cls << 4) | h.dest));
        d.insert(d.end(), h.s->len, h.s->len + 16);
        d.insert(d.end(), h.s->val, h.s->val + h.s->n);
        segment(0xC4, d);
    }
    d = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};                                              // SOS
    segment(0xDA, d);
    b.resize(624, 0);
    b.insert(b.end(), q[0], q[0] + 64);
    b.insert(b.end(), q[1], q[1] + 64);
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            // reciprocal of 2q for the kernel's exact round-half-away division (see jpeg_dct_quant_kernel)
            const uint64_t d2 = 2ull * q[t][i];
            const uint32_t m = (uint32_t)((((uint64_t)1 << 32) + d2 - 1) / d2);
            const uint8_t *mb = reinterpret_cast<const uint8_t *>(&m);
            b.insert(b.end(), mb, mb + 4);
        }
    out.assign(kJpegTableBlockBytes / 4, 0);
    memcpy(out.data(), b.data(), kJpegTableBlockBytes);
}

} // namespace fl
