// fl_jpeghuff.cpp -- host half of the JPEG decode front end: marker parsing (ITU-T T.81 Annex B) and the sequential
// Huffman decoder (Annex F.2), table driven, writing the compact coefficient blob the device kernels consume
// (fl_jpegdec.h).  Reference: src/handler.rs:205-220 (ImageReader -> JpegDecoder::new -> DynamicImage::from_decoder,
// zune-jpeg 0.4.14) and handler.rs:206 (decoder.orientation(): the EXIF tag).
#include "fl_jpegdec.h"

#include <string.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <map>
#include <vector>

namespace fl {

namespace {

inline uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

constexpr int kAcBits = 10;  // lookahead of the single-pass decoder's one-step AC table (4 KB per table; 11 bits measured the same)
constexpr int kFastBits = 9; // lookahead of the one-step tables (11 bits resolve a few more symbols at once and cost as much again in L1 misses: measured equal)

struct Huff {
    // kFastBits lookahead: (length << 8) | symbol, 0 = longer than that
    uint16_t fast[1 << kFastBits];
    int32_t maxcode[18];  // per length, -1 = none; [17] = sentinel
    int32_t valoff[17];   // symbol index of the first code of a length minus that code
    uint8_t vals[256];
    uint8_t counts[16];   // codes per length 1..16 (the DHT segment's, for the device's table)
    // The single-pass decoder's table (AC and DC tables alike): kAcBits of lookahead resolve code + magnitude (or the end-of-block and
    // ZRL codes) in one step: bits 0-4 = bits to drop, 5-8 = zero run, 12 = no value (13 = end of block, else ZRL),
    // 16-31 = value; 0 = slow path
    uint32_t fastx[1 << kAcBits];
    bool present = false;
};

bool build_huff(Huff &h, const uint8_t *bits /*[16] counts of lengths 1..16*/, const uint8_t *vals, int total)
{
    memcpy(h.vals, vals, (size_t)total);
    memcpy(h.counts, bits, 16);
    memset(h.fast, 0, sizeof(h.fast));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valoff[l] = k - code;
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            if (l <= kFastBits) {
                const int first = code << (kFastBits - l), cnt = 1 << (kFastBits - l);
                if (first + cnt > (1 << kFastBits)) return false;
                for (int f = 0; f < cnt; ++f) h.fast[first + f] = (uint16_t)((l << 8) | vals[k]);
            }
        }
        if (code > (1 << l)) return false; // over-subscribed
        h.maxcode[l] = bits[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    // code + magnitude (or end-of-block / ZRL) in one lookup: the common case of AC coefficients is a small value after a short run
    for (int i = 0; i < (1 << kAcBits); ++i) {
        h.fastx[i] = 0;
        // the code at the top of the kAcBits: canonical search (table construction only)
        int len = 0, sym = -1;
        for (int l = 1; l <= kAcBits; ++l) {
            const int c = i >> (kAcBits - l);
            if (h.maxcode[l] >= 0 && c <= h.maxcode[l] && c + h.valoff[l] >= 0 && c + h.valoff[l] < total) { len = l; sym = h.vals[c + h.valoff[l]]; break; }
        }
        if (sym < 0) continue;
        const int run = sym >> 4, mag = sym & 15;
        if (mag == 0) {
            if (run == 0) h.fastx[i] = (uint32_t)len | (3u << 12);
            else if (run == 15) h.fastx[i] = (uint32_t)len | (1u << 12);
            continue; // (other run / 0 symbols are not baseline codes: the slow path treats them as it always did)
        }
        if (len + mag > kAcBits) continue;
        int v = ((i << len) & ((1 << kAcBits) - 1)) >> (kAcBits - mag);
        if (v < (1 << (mag - 1))) v += (int)(~0u << mag) + 1;
        h.fastx[i] = (uint32_t)(len + mag) | ((uint32_t)run << 5) | ((uint32_t)(uint16_t)(int16_t)v << 16);
    }
    h.present = true;
    return true;
}

struct BitReader {
    const uint8_t *d;
    size_t n, pos;
    uint64_t buf = 0;
    int cnt = 0;
    int marker = 0; // a marker was met: zeros are supplied from here on

    void fill()
    {
        // fast path: the next 8 bytes hold no 0xFF (no stuffing, no marker): take as many whole bytes as fit in one go
        if (!marker && pos + 8 <= n && cnt <= 56) {
            uint64_t raw;
            memcpy(&raw, d + pos, 8);
            const uint64_t inv = ~raw;
            if (!((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull)) {
                const int k = (64 - cnt) >> 3; // 1..8 bytes
                const uint64_t be = __builtin_bswap64(raw);
                buf |= (k == 8 ? be : (be >> (64 - 8 * k)) << (64 - cnt - 8 * k));
                pos += (size_t)k;
                cnt += 8 * k;
                return;
            }
        }
        while (cnt <= 56) {
            uint32_t b = 0;
            if (!marker && pos < n) {
                b = d[pos++];
                if (b == 0xFF) {
                    const uint32_t b2 = pos < n ? d[pos] : 0xD9;
                    if (b2 == 0) pos++;
                    else { marker = (int)b2; pos++; b = 0; }
                }
            }
            buf |= (uint64_t)b << (56 - cnt);
            cnt += 8;
        }
    }
    inline uint32_t peek(int k) { return (uint32_t)(buf >> (64 - k)); }
    inline void drop(int k) { buf <<= k; cnt -= k; }
};

inline int decode_sym(BitReader &br, const Huff &h)
{
    if (br.cnt < 16) br.fill();
    const uint32_t f = h.fast[br.peek(kFastBits)];
    if (f) { br.drop((int)(f >> 8)); return (int)(f & 255u); }
    // longer than the lookahead: canonical search
    uint32_t code = br.peek(kFastBits + 1);
    int l = kFastBits + 1;
    while (l <= 16 && (int32_t)code > h.maxcode[l]) { ++l; code = br.peek(l); }
    if (l > 16) return -1;
    br.drop(l);
    const int idx = (int)code + h.valoff[l];
    return idx >= 0 && idx < 256 ? h.vals[idx] : -1;
}

inline int receive_extend(BitReader &br, int s)
{
    if (!s) return 0;
    if (br.cnt < s) br.fill();
    const int v = (int)br.peek(s);
    br.drop(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

int exif_orientation(const uint8_t *p, size_t len)
{
    if (len < 14 || memcmp(p, "Exif\0\0", 6)) return 0;
    const uint8_t *t = p + 6;
    const size_t n = len - 6;
    bool le;
    if (t[0] == 'I' && t[1] == 'I') le = true; else if (t[0] == 'M' && t[1] == 'M') le = false; else return 0;
    auto rd16 = [&](size_t o) -> uint32_t { return le ? (uint32_t)(t[o] | (t[o + 1] << 8)) : (uint32_t)((t[o] << 8) | t[o + 1]); };
    auto rd32 = [&](size_t o) -> uint32_t {
        return le ? ((uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) | ((uint32_t)t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24))
                  : (((uint32_t)t[o] << 24) | ((uint32_t)t[o + 1] << 16) | ((uint32_t)t[o + 2] << 8) | (uint32_t)t[o + 3]);
    };
    if (rd16(2) != 42) return 0;
    const size_t ifd = rd32(4);
    if (ifd + 2 > n) return 0;
    const uint32_t cnt = rd16(ifd);
    for (uint32_t i = 0; i < cnt; ++i) {
        const size_t e = ifd + 2 + 12u * i;
        if (e + 12 > n) return 0;
        // (the tag counts only as what the specification makes it, one SHORT: an entry of another type or count is skipped and
        // the search goes on, as far as is known what image 0.25.6's Orientation::from_exif_chunk does)
        if (rd16(e) == 0x0112 && rd16(e + 2) == 3 && rd32(e + 4) == 1) { const uint32_t v = rd16(e + 8); return v >= 1 && v <= 8 ? (int)v : 0; }
    }
    return 0;
}

struct Parsed {
    JpegInfo info;
    struct C { uint32_t id, h, v, tq, td, ta; } c[4];
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    Huff ht[2][4];
    size_t scan_pos = 0;
    size_t sos_marker = 0; // offset of the first SOS marker's 0xFF
    bool one_scan = false; // SOS names every component in frame order
    bool single_pass = false; // sequential process, one interleaved scan: decoded straight into the blob
    std::map<uint32_t, std::vector<uint8_t>> icc_chunks;
    uint32_t icc_count = 0;
};

// DQT, DHT and DRI segments (they may also stand between the scans of a multi-scan file): 0 ok, -1 malformed
int table_segment(uint32_t m, const uint8_t *p, uint32_t pl, Parsed &P, bool want_tables)
{
    if (m == 0xDB) {
        uint32_t o = 0;
        while (o < pl) {
            const uint32_t pq = p[o] >> 4, tq = p[o] & 15u;
            if (tq > 3 || pq > 1 || o + 1 + 64 * (pq + 1) > pl) return -1;
            o++;
            for (int k = 0; k < 64; ++k) P.qt[tq][k] = pq ? (uint16_t)be16(p + o + 2 * k) : p[o + k];
            o += 64 * (pq + 1);
            P.have_qt[tq] = true;
        }
    } else if (m == 0xC4) {
        uint32_t o = 0;
        while (o < pl) {
            const uint32_t tc = p[o] >> 4, th = p[o] & 15u;
            if (tc > 1 || th > 3 || o + 17 > pl) return -1;
            int total = 0;
            for (int l = 0; l < 16; ++l) total += p[o + 1 + l];
            if (total > 256 || o + 17 + (uint32_t)total > pl) return -1;
            if (want_tables && !build_huff(P.ht[tc][th], p + o + 1, p + o + 17, total)) return -1;
            o += 17 + (uint32_t)total;
        }
    } else if (m == 0xDD) {
        if (pl < 2) return -1;
        P.info.restart_interval = be16(p);
    }
    return 0;
}

// -1 malformed; 0 ok (info.supported says whether the scan can be decoded here)
int parse(const uint8_t *d, size_t n, Parsed &P, bool want_tables)
{
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
    size_t pos = 2;
    bool got_sof = false;
    for (;;) {
        if (pos + 4 > n || d[pos] != 0xFF) return -1;
        while (pos < n && d[pos] == 0xFF) pos++;
        if (pos >= n) return -1;
        const uint32_t m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;
        if (pos + 2 > n) return -1;
        const uint32_t len = be16(d + pos);
        if (len < 2 || pos + len > n) return -1;
        const uint8_t *p = d + pos + 2;
        const uint32_t pl = len - 2;
        if (m == 0xDB || m == 0xC4 || m == 0xDD) {
            if (table_segment(m, p, pl, P, want_tables) != 0) return -1;
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (pl < 6 || got_sof) return -1; // (a second frame header: T.81 allows one per image; zune-jpeg rejects it too)
            P.info.progressive = m == 0xC2;
            P.info.sof = m;
            P.info.precision = p[0];
            P.info.height = be16(p + 1);
            P.info.width = be16(p + 3);
            P.info.components = p[5];
            const uint32_t nc = p[5];
            if (nc < 1 || nc > 4 || pl < 6 + 3 * nc || !P.info.width || !P.info.height) return -1;
            for (uint32_t i = 0; i < nc; ++i) {
                P.c[i].id = p[6 + 3 * i]; P.c[i].h = p[7 + 3 * i] >> 4; P.c[i].v = p[7 + 3 * i] & 15u; P.c[i].tq = p[8 + 3 * i];
                if (P.c[i].h < 1 || P.c[i].h > 4 || P.c[i].v < 1 || P.c[i].v > 4 || P.c[i].tq > 3) return -1;
                P.info.hmax = P.c[i].h > P.info.hmax ? P.c[i].h : P.info.hmax;
                P.info.vmax = P.c[i].v > P.info.vmax ? P.c[i].v : P.info.vmax;
            }
            got_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            // lossless / differential / arithmetic: a JPEG, but not one for this path
            P.info.progressive = 1;
            if (pl >= 6) { P.info.precision = p[0]; P.info.height = be16(p + 1); P.info.width = be16(p + 3); P.info.components = p[5]; }
            P.info.supported = 0;
            return P.info.width && P.info.height ? 0 : -1;
        } else if (m == 0xEE) {
            if (pl >= 12 && !memcmp(p, "Adobe", 5)) P.info.adobe_transform = p[11];
        } else if (m == 0xE2) {
            // ICC profile, possibly in several chunks: "ICC_PROFILE\0" seq count data (decoder.icc_profile(), handler.rs:447)
            if (pl > 14 && !memcmp(p, "ICC_PROFILE\0", 12)) P.icc_chunks[p[12]] = std::vector<uint8_t>(p + 14, p + pl), P.icc_count = p[13];
        } else if (m == 0xE1) {
            if (!P.info.exif_orientation) P.info.exif_orientation = (uint32_t)exif_orientation(p, pl);
        } else if (m == 0xDA) {
            if (!got_sof || pl < 1) return -1;
            const uint32_t ns = p[0], nc = P.info.components;
            if (ns < 1 || ns > 4 || pl < 1 + 2 * ns + 3) return -1;
            P.one_scan = ns == nc;
            for (uint32_t i = 0; i < ns && P.one_scan; ++i) {
                if (P.c[i].id != p[1 + 2 * i]) { P.one_scan = false; break; }
                P.c[i].td = p[2 + 2 * i] >> 4; P.c[i].ta = p[2 + 2 * i] & 15u;
                if (P.c[i].td > 3 || P.c[i].ta > 3) return -1;
            }
            P.scan_pos = pos + len;
            P.sos_marker = pos - 2;
            // What the decoder takes: 8-bit Huffman processes -- baseline / extended sequential (SOF0, SOF1) in one scan or several,
            // and progressive (SOF2: spectral selection and successive approximation, assembled on the host, T.81 Annex G) --
            // with 1, 3 or 4 components, every plane at full or half resolution per direction.
            bool ok = (P.info.sof == 0xC0 || P.info.sof == 0xC1 || P.info.sof == 0xC2) && P.info.precision == 8 && (nc == 1 || nc == 3 || nc == 4);
            for (uint32_t i = 0; ok && nc > 1 && i < nc; ++i) // every plane at full or half resolution per direction
                ok = P.info.hmax % P.c[i].h == 0 && P.info.vmax % P.c[i].v == 0 && P.info.hmax / P.c[i].h <= 2 && P.info.vmax / P.c[i].v <= 2;
            P.info.supported = ok ? 1u : 0u;
            P.single_pass = ok && P.info.sof != 0xC2 && P.one_scan;
            if (P.icc_count && P.icc_chunks.size() == P.icc_count) {
                for (uint32_t k = 1; k <= P.icc_count; ++k) {
                    auto it = P.icc_chunks.find(k);
                    if (it == P.icc_chunks.end()) { P.info.icc.clear(); break; }
                    P.info.icc.insert(P.info.icc.end(), it->second.begin(), it->second.end());
                }
            }
            return 0;
        }
        pos += len;
    }
}


void layout(const Parsed &P, JpegBlobHeader &H)
{
    const JpegInfo &I = P.info;
    memset(&H, 0, sizeof(H));
    H.magic = 0x31444a46u; // "FJD1"
    H.width = I.width; H.height = I.height; H.nc = I.components;
    H.hmax = I.components == 1 ? 1u : I.hmax;
    H.vmax = I.components == 1 ? 1u : I.vmax;
    H.is_rgb = I.components == 3 && I.adobe_transform == 0;
    H.adobe_transform = (uint32_t)(I.adobe_transform + 1);
    const uint32_t mcux = (I.width + 8 * H.hmax - 1) / (8 * H.hmax), mcuy = (I.height + 8 * H.vmax - 1) / (8 * H.vmax);
    uint32_t nb = 0, po = 0;
    for (uint32_t i = 0; i < I.components; ++i) {
        JpegComponent &c = H.comp[i];
        c.h = I.components == 1 ? 1u : P.c[i].h; // a single component is never interleaved: its MCU is one block
        c.v = I.components == 1 ? 1u : P.c[i].v;
        c.bw = mcux * c.h; c.bh = mcuy * c.v;
        c.w = (I.width * c.h + H.hmax - 1) / H.hmax;
        c.hpx = (I.height * c.v + H.vmax - 1) / H.vmax;
        c.block_base = nb;
        c.plane_off = po;
        c.tq = P.c[i].tq;
        nb += c.bw * c.bh;
        po += c.bw * c.bh * 64u;
        memcpy(H.qt[i], P.qt[P.c[i].tq], 128);
    }
    H.nblocks = nb;
    H.plane_bytes = po;
    H.blocks_off = (uint32_t)sizeof(JpegBlobHeader);
    H.coef_off = H.blocks_off + nb * 4u;
}


// ---- files of several scans: progressive (T.81 Annex G) and sequential files that code their components one after the other ----
// The coefficients of the whole picture are assembled on the host, scan by scan, and packed into the same blob the
// single-pass decoder writes: the device half (dequantisation, IDCT, up-sampling, colour) does not know the difference.

struct Scan {
    uint32_t ns = 0, ci[4] = {0, 0, 0, 0}; // components of the scan as frame indices
    uint32_t td[4] = {0, 0, 0, 0}, ta[4] = {0, 0, 0, 0};
    uint32_t ss = 0, se = 63, ah = 0, al = 0;
};

// `z` without its r lowest set bits (r <= 15): where a run of r zero coefficients ends.  One PDEP where the CPU has BMI2 (the run length
// is data: a loop of r steps is a mispredicted branch per symbol of a refinement scan), the loop elsewhere.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) // (this file is host code, but hipcc also runs its device pass over it)
__attribute__((target("bmi2"))) static uint64_t select_from_bmi2(uint64_t z, unsigned r) { return __builtin_ia32_pdep_di(~0ull << r, z); }
#endif
static uint64_t select_from_loop(uint64_t z, unsigned r) { for (unsigned i = 0; i < r && z; ++i) z &= z - 1ull; return z; }
static uint64_t (*const select_from)(uint64_t, unsigned) = [] {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("bmi2")) return &select_from_bmi2;
#endif
    return &select_from_loop;
}();

// One scan into coef[block][64] (zig-zag order).  0 ok, -1 malformed.
// nzm[block]: bit k = coefficient k is non-zero (round 5).  The refinement scans of a progressive file visit every coefficient of every
// block to ask "non-zero already?" -- 2 M look-ups per luma scan of a 1080p picture, most of them "no" -- and the packing step looked
// for each block's last coefficient the same way; with the mask a correction-bit pass walks the set bits only (count-trailing-zeros).
// The bit buffer of the two AC loops lives in registers for the length of a block (br is visible to the out-of-line refill).
int decode_scan(BitReader &br, const Parsed &P, const JpegBlobHeader &H, const Scan &S, bool progressive, int16_t *coef, uint64_t *nzm)
{
    const bool dc_scan = S.ss == 0;
    uint32_t mcus_x, mcus_y;
    if (S.ns > 1) { mcus_x = H.comp[0].bw / H.comp[0].h; mcus_y = H.comp[0].bh / H.comp[0].v; }
    else { const JpegComponent &c = H.comp[S.ci[0]]; mcus_x = (c.w + 7u) / 8u; mcus_y = (c.hpx + 7u) / 8u; } // a lone component: its own blocks, no MCU padding
    int pred[4] = {0, 0, 0, 0};
    uint32_t eobrun = 0, rst_left = P.info.restart_interval;
    const int p1 = 1 << S.al, m1 = -(1 << S.al);
    // The bit buffer lives in registers for the length of the scan (br is visible to the out-of-line refill, so its members are memory to
    // the compiler); `need`, `bit`, `bits`, `sym` keep br's meaning: zeros behind a marker or the end.  br gets it back around the restart
    // handling and on every way out (Back).
    uint64_t buf = br.buf;
    int bc = br.cnt;
    auto need = [&](int k) { if (bc < k) { br.buf = buf; br.cnt = bc; br.fill(); buf = br.buf; bc = br.cnt; } };
    auto bit = [&]() -> int { need(1); const int v = (int)(buf >> 63); buf <<= 1; --bc; return v; };
    auto bits = [&](int k) -> int { if (!k) return 0; need(k); const int v = (int)(buf >> (64 - k)); buf <<= k; bc -= k; return v; };
    auto sym = [&](const Huff &h) -> int {
        need(16);
        const uint32_t f = h.fast[buf >> (64 - kFastBits)];
        if (f) { buf <<= (f >> 8); bc -= (int)(f >> 8); return (int)(f & 255u); }
        br.buf = buf; br.cnt = bc;
        const int r = decode_sym(br, h);
        buf = br.buf; bc = br.cnt;
        return r;
    };
    struct Back { BitReader &br; uint64_t &buf; int &bc; ~Back() { br.buf = buf; br.cnt = bc; } } back_{br, buf, bc};
    for (uint32_t my = 0; my < mcus_y; ++my)
        for (uint32_t mx = 0; mx < mcus_x; ++mx) {
            if (P.info.restart_interval && rst_left == 0) {
                br.buf = buf; br.cnt = bc;
                struct Reload { BitReader &br; uint64_t &buf; int &bc; ~Reload() { buf = br.buf; bc = br.cnt; } } reload_{br, buf, bc};
                if (!br.marker) {
                    br.pos -= (size_t)(br.cnt / 8);
                    if (br.pos + 2 > br.n || br.d[br.pos] != 0xFF || br.d[br.pos + 1] < 0xD0 || br.d[br.pos + 1] > 0xD7) return -1;
                    br.pos += 2;
                } else if (br.marker >= 0xD0 && br.marker <= 0xD7) br.marker = 0;
                else return -1;
                br.buf = 0; br.cnt = 0;
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                eobrun = 0;
                rst_left = P.info.restart_interval;
            }
            for (uint32_t k0 = 0; k0 < S.ns; ++k0) {
                const uint32_t ci = S.ci[k0];
                const JpegComponent &c = H.comp[ci];
                const uint32_t nh = S.ns > 1 ? c.h : 1u, nv = S.ns > 1 ? c.v : 1u;
                for (uint32_t v = 0; v < nv; ++v)
                    for (uint32_t h = 0; h < nh; ++h) {
                        const uint32_t bx = mx * nh + h, by = my * nv + v;
                        const size_t bi = (size_t)c.block_base + (size_t)by * c.bw + bx;
                        int16_t *b = coef + bi * 64;
                        if (progressive && dc_scan) { // (the DC scans do not touch the masks)
                            if (S.ah == 0) { // G.1.2.1 first DC scan: the difference, scaled by 2^Al
                                const int t = sym(P.ht[0][S.td[k0]]);
                                if (t < 0 || t > 11) return -1;
                                { const int v = bits(t); pred[ci] += t ? (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v) : 0; }
                                const int val = pred[ci] * (1 << S.al);
                                if (val < -32768 || val > 32767) return -1;
                                b[0] = (int16_t)val;
                            } else if (bit()) b[0] = (int16_t)(b[0] | p1); // refinement: one more bit of every DC term
                            continue;
                        }
                        uint64_t nz = nzm[bi];
                        struct Keep { uint64_t *slot; uint64_t &nz; ~Keep() { *slot = nz; } } keep_{nzm + bi, nz}; // (the mask goes back on every way out of the block)
                        if (!progressive) {
                            // sequential: DC difference, then the AC coefficients up to the end-of-block code (F.2.2)
                            const Huff &hd = P.ht[0][S.td[k0]], &ha = P.ht[1][S.ta[k0]];
                            const int t = sym(hd);
                            if (t < 0 || t > 11) return -1;
                            { const int v = bits(t); pred[ci] += t ? (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v) : 0; }
                            if (pred[ci] < -32768 || pred[ci] > 32767) return -1;
                            b[0] = (int16_t)pred[ci];
                            for (int k = 1; k < 64;) {
                                const int rs = sym(ha);
                                if (rs < 0) return -1;
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                k += r;
                                if (k > 63) return -1;
                                const int v = bits(sz);
                                b[k] = (int16_t)(v < (1 << (sz - 1)) ? v - (1 << sz) + 1 : v);
                                nz |= 1ull << k;
                                ++k;
                            }
                        } else if (S.ah == 0) { // G.1.2.2 first AC scan of the band [ss, se]
                            if (eobrun) { --eobrun; continue; }
                            const Huff &ha = P.ht[1][S.ta[k0]];
                            for (int k = (int)S.ss; k <= (int)S.se;) {
                                const int rs = sym(ha);
                                if (rs < 0) return -1;
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz == 0) {
                                    if (r == 15) { k += 16; continue; }
                                    eobrun = (1u << r) - 1u + (uint32_t)bits(r); // EOBr: this block and eobrun more end here
                                    break;
                                }
                                k += r;
                                if (k > (int)S.se) return -1;
                                const int v = bits(sz);
                                const int val = (v < (1 << (sz - 1)) ? v - (1 << sz) + 1 : v) * (1 << S.al);
                                if (val < -32768 || val > 32767) return -1;
                                b[k] = (int16_t)val;
                                if (val) nz |= 1ull << k;
                                ++k;
                            }
                        } else { // G.1.2.3 refinement of the band: one more bit of the coefficients already non-zero, and new +-1 << Al ones
                            int k = (int)S.ss;
                            const Huff &ha = P.ht[1][S.ta[k0]];
                            const uint64_t band = (~0ull << S.ss) & (S.se >= 63 ? ~0ull : ((1ull << (S.se + 1)) - 1ull));
                            // correction bits for the non-zero coefficients in `m` (ascending): read together, applied one by one
                            auto refine_set = [&](uint64_t m) {
                                int c = __builtin_popcountll(m);
                                while (c > 0) {
                                    const int take = c > 32 ? 32 : c;
                                    const uint32_t corr = (uint32_t)bits(take);
                                    for (int i = take - 1; i >= 0; --i) {
                                        const int kk = __builtin_ctzll(m);
                                        m &= m - 1ull;
                                        // (no branch on the data: the bit, "this bit of the coefficient is still clear" and the sign as arithmetic)
                                        const int cf = b[kk];
                                        const int add = (int)((corr >> i) & 1u) & (int)!(cf & p1);
                                        b[kk] = (int16_t)(cf + add * (cf >= 0 ? p1 : m1));
                                    }
                                    c -= take;
                                }
                            };
                            if (!eobrun) {
                                for (; k <= (int)S.se; ++k) {
                                    const int rs = sym(ha);
                                    if (rs < 0) return -1;
                                    const int r = rs >> 4;
                                    const int sz = rs & 15;
                                    int val = 0;
                                    if (sz) {
                                        if (sz != 1) return -1;
                                        val = bit() ? p1 : m1;
                                    } else if (r != 15) {
                                        eobrun = (1u << r) + (uint32_t)bits(r); // this block included
                                        break;
                                    }
                                    // skip r coefficients that are still zero, refining the non-zero ones met on the way: the walk ends AT the
                                    // (r + 1)-th zero from k on (or behind the band) -- r <= 15 zeros dropped from the mask of zeros, then its lowest bit
                                    const uint64_t z = select_from(~nz & band & (~0ull << k), (unsigned)r);
                                    const int pos = z ? __builtin_ctzll(z) : (int)S.se + 1;
                                    refine_set(nz & band & (~0ull << k) & (pos >= 64 ? ~0ull : ((1ull << pos) - 1ull)));
                                    k = pos;
                                    if (sz) { if (k > (int)S.se) return -1; b[k] = (int16_t)val; nz |= 1ull << k; }
                                }
                            }
                            if (eobrun) {
                                // the rest of the band: correction bits for the coefficients that are non-zero already -- the set bits from k on
                                if (k <= (int)S.se) refine_set(nz & band & (~0ull << k));
                                --eobrun;
                            }
                        }
                    }
            }
            if (P.info.restart_interval) rst_left--;
        }
    return 0;
}

// Walks the segments from the first SOS marker on: tables, scans, EOI.  0 ok, -1 malformed, -2 not covered.
int decode_scans(const uint8_t *d, size_t n, Parsed &P, const JpegBlobHeader &H, int16_t *coef, uint64_t *nzm)
{
    const bool progressive = P.info.sof == 0xC2;
    size_t pos = P.sos_marker;
    uint32_t scans = 0;
    uint64_t work_blocks = 0; // blocks visited by the scans so far
    for (;;) {
        if (pos + 2 > n) return scans ? 0 : -1; // (no EOI: what has been decoded stands, as decoders in the field do)
        if (d[pos] != 0xFF) return -1;
        while (pos < n && d[pos] == 0xFF) pos++;
        if (pos >= n) return scans ? 0 : -1;
        const uint32_t m = d[pos++];
        if (m == 0xD9) return scans ? 0 : -1;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) return -1;
        const uint32_t len = be16(d + pos);
        if (len < 2 || pos + len > n) return -1;
        const uint8_t *p = d + pos + 2;
        const uint32_t pl = len - 2;
        if (m == 0xDB || m == 0xC4 || m == 0xDD) { if (table_segment(m, p, pl, P, true) != 0) return -1; pos += len; continue; }
        if (m != 0xDA) { pos += len; continue; }
        Scan S;
        S.ns = p[0];
        if (S.ns < 1 || S.ns > 4 || pl < 1 + 2 * S.ns + 3) return -1;
        for (uint32_t i = 0; i < S.ns; ++i) {
            uint32_t ci = 0;
            while (ci < P.info.components && P.c[ci].id != p[1 + 2 * i]) ++ci;
            if (ci == P.info.components) return -1;
            if (i && ci <= S.ci[i - 1]) return -1; // B.2.3: in frame order
            S.ci[i] = ci; S.td[i] = p[2 + 2 * i] >> 4; S.ta[i] = p[2 + 2 * i] & 15u;
            if (S.td[i] > 3 || S.ta[i] > 3) return -1;
        }
        S.ss = p[1 + 2 * S.ns]; S.se = p[2 + 2 * S.ns]; S.ah = p[3 + 2 * S.ns] >> 4; S.al = p[3 + 2 * S.ns] & 15u;
        if (progressive) {
            if (S.ss > S.se || S.se > 63 || S.al > 13 || S.ah > 13 || (S.ss == 0 && S.se != 0) || (S.ss > 0 && S.ns != 1)) return -1;
        } else if (S.ss != 0 || S.se != 63 || S.ah != 0 || S.al != 0) return -1;
        for (uint32_t i = 0; i < S.ns; ++i) {
            if ((S.ss == 0 && S.ah == 0 && !P.ht[0][S.td[i]].present) || ((S.ss > 0 || !progressive) && !P.ht[1][S.ta[i]].present)) return -1;
        }
        if (S.ns > 1) { // interleaved: the MCU must not exceed 10 blocks (B.2.3)
            uint32_t nb = 0;
            for (uint32_t i = 0; i < S.ns; ++i) nb += H.comp[S.ci[i]].h * H.comp[S.ci[i]].v;
            if (nb > 10) return -1;
        }
        // A bound on the WORK a file may ask for, not only on its scan count: every scan walks all blocks of its components (an
        // end-of-band run still touches each of them), so a few hundred bytes per scan can cost 10^7 block visits.  Real encoders
        // write ~10 scans, i.e. each block is visited ~6 times; beyond 32 visits per block of the frame on average the file goes to
        // the caller's own decoder (-2 = unsupported), before the scan is decoded.
        {
            uint64_t visit = 0;
            for (uint32_t i = 0; i < S.ns; ++i) visit += (uint64_t)H.comp[S.ci[i]].bw * H.comp[S.ci[i]].bh;
            work_blocks += visit;
            if (work_blocks > 32ull * H.nblocks + 4096ull) return -2;
        }
        BitReader br{d, n, pos + len};
        if (decode_scan(br, P, H, S, progressive, coef, nzm) != 0) return -1;
        ++scans;
        if (scans > 1000) return -2;
        // the next marker: either the reader ran into it, or it lies in the bytes not yet read
        if (br.marker) { pos = br.pos - 2; continue; }
        size_t q = br.pos;
        while (q + 1 < n && !(d[q] == 0xFF && d[q + 1] != 0x00 && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7) && d[q + 1] != 0xFF)) ++q;
        pos = q;
    }
}

} // namespace

int jpeg_parse_info(const uint8_t *data, size_t n, JpegInfo &info)
{
    Parsed P;
    const int rc = parse(data, n, P, false);
    info = P.info;
    return rc;
}

void jpeg_color_job(const JpegBlobHeader &H, JpegDecJob &j)
{
    j.mode = 0; j.width = H.width; j.height = H.height;
    j.y_off = j.y_pitch = j.cb_off = j.cr_off = j.c_pitch = j.c_rows = j.c_w = 0;
    if (H.nc == 3u && !H.is_rgb && H.comp[0].h == H.hmax && H.comp[0].v == H.vmax && H.comp[1].h == H.comp[2].h && H.comp[1].v == H.comp[2].v &&
        H.comp[1].h && H.comp[1].v) {
        const uint32_t sh = H.hmax / H.comp[1].h, sv = H.vmax / H.comp[1].v;
        j.mode = (sh == 2u && sv == 2u) ? 1u : (sh == 2u && sv == 1u) ? 2u : (sh == 1u && sv == 1u) ? 3u : 0u;
        j.y_off = H.comp[0].plane_off; j.y_pitch = H.comp[0].bw * 8u;
        j.cb_off = H.comp[1].plane_off; j.cr_off = H.comp[2].plane_off; j.c_pitch = H.comp[1].bw * 8u;
        j.c_rows = H.comp[1].hpx; j.c_w = H.comp[1].w;
    }
}

size_t jpeg_blob_bound(const JpegInfo &I)
{
    if (!I.supported) return 0;
    const uint32_t hmax = I.components == 1 ? 1u : I.hmax, vmax = I.components == 1 ? 1u : I.vmax;
    const size_t mcux = (I.width + 8 * hmax - 1) / (8 * hmax), mcuy = (I.height + 8 * vmax - 1) / (8 * vmax);
    // blocks per MCU <= components * hmax * vmax
    const size_t nb = mcux * mcuy * (I.components == 1 ? 1u : (size_t)I.components * hmax * vmax);
    return sizeof(JpegBlobHeader) + nb * 4 + nb * 128 + 64;
}

size_t jpeg_stage_bound(size_t file_bytes)
{
    return sizeof(JpegBlobHeader) + sizeof(JpegHuffStage) + 4u * kJhTableWords * 4u + file_bytes + 96u + 4u * std::min<size_t>(kJhMaxRestarts, file_bytes / 2u);
}

// Host half of the DEVICE entropy decoder (fl_jpeghuff_dev.hip): header parse, code tables in the device's layout, the entropy-coded
// segment with its 0xFF00 stuffing removed.  ~30 us for a 300 KB file, against ~1.8 ms of Huffman decoding.
int jpeg_entropy_stage(const uint8_t *data, size_t n, uint8_t *out, size_t cap, size_t *used)
{
    Parsed P;
    int rc = parse(data, n, P, true);
    if (rc) return rc;
    if (!P.info.supported || !P.single_pass) return -2;
    const uint32_t nc = P.info.components;
    if (nc != 1 && nc != 3) return -2;
    for (uint32_t i = 0; i < nc; ++i) if (!P.have_qt[P.c[i].tq]) return -1;
    if (cap < jpeg_stage_bound(n - P.scan_pos)) return -2;
    JpegBlobHeader H;
    layout(P, H);
    JpegHuffStage S;
    memset(&S, 0, sizeof(S));
    // the code tables the scan names: at most four distinct ones fit the device's LDS (baseline files have two of each class at most)
    int slot_of[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
    uint32_t nslots = 0;
    uint32_t *tables = reinterpret_cast<uint32_t *>(out + sizeof(JpegBlobHeader) + sizeof(JpegHuffStage));
    memset(tables, 0, 4u * kJhTableWords * 4u);
    int pair_of[4] = {-1, -1, -1, -1}; // per table slot: the slot of the AC table whose end-of-block code may follow its symbols (-2: more than one)
    const Huff *huff_of[4] = {nullptr, nullptr, nullptr, nullptr};
    for (uint32_t i = 0; i < nc; ++i)
        for (int cls = 0; cls < 2; ++cls) {
            const uint32_t id = cls ? P.c[i].ta : P.c[i].td;
            if (!P.ht[cls][id].present) return -1;
            if (slot_of[cls][id] < 0) {
                if (nslots == 4) return -2;
                slot_of[cls][id] = (int)nslots;
                const Huff &h = P.ht[cls][id];
                huff_of[nslots] = &h;
                uint32_t *t = tables + (size_t)nslots * kJhTableWords;
                // the device's layout (fl_jpeghuff_dev.hip): look[2^kJhLookBits] halfwords = code length | symbol << 8 for every code of
                // up to kJhLookBits bits (canonical order, as build_huff assigns them; 0 = a longer code or none), maxcode, valoff, vals
                constexpr uint32_t kLookWords = (1u << kJhLookBits) / 2u;
                static_assert(kLookWords + 18 + 17 + 64 <= kJhTableWords, "device table layout (fl_jpegdec.h kJhTableWords)");
                uint16_t *look = reinterpret_cast<uint16_t *>(t);
                uint32_t code = 0, k = 0;
                for (uint32_t l = 1; l <= kJhLookBits; ++l) {
                    for (uint32_t q = 0; q < h.counts[l - 1]; ++q, ++k, ++code) {
                        const uint32_t first = code << (kJhLookBits - l), cnt = 1u << (kJhLookBits - l);
                        const uint16_t e = (uint16_t)(l | ((uint32_t)h.vals[k] << 8));
                        for (uint32_t f = 0; f < cnt; ++f) look[first + f] = e; // (build_huff has checked the code space)
                    }
                    code <<= 1;
                }
                memcpy(t + kLookWords, h.maxcode, 18 * 4);
                memcpy(t + kLookWords + 18, h.valoff, 17 * 4);
                memcpy(t + kLookWords + 35, h.vals, 256);
                ++nslots;
            }
            (cls ? S.ac_tab : S.dc_tab)[i] = (uint8_t)slot_of[cls][id];
        }
    for (uint32_t i = 0; i < nc; ++i) {
        const int d = S.dc_tab[i], a = S.ac_tab[i];
        pair_of[d] = (pair_of[d] == -1 || pair_of[d] == a) ? a : -2;
        pair_of[a] = (pair_of[a] == -1 || pair_of[a] == a) ? a : -2; // (a slot used as DC table by one component and as AC table by another: no pairing)
    }
    // An end-of-block code right behind a symbol's magnitude bits, all inside the lookahead: bits 5-7 of the entry = its length, and the
    // device consumes it with the symbol (one decoding step for the "DC difference, end of block" blocks of flat regions, one step less
    // for every block whose last coefficient is a short code).  Symbols that carry no coefficient (end of block, ZRL) and DC categories
    // above 11 stay plain.
    for (uint32_t sl = 0; sl < nslots; ++sl) {
        if (pair_of[sl] < 0) continue;
        const Huff &ha = *huff_of[pair_of[sl]];
        uint32_t el = 0, ec = 0, code = 0, k = 0;
        for (uint32_t l = 1; l <= 16 && !el; ++l) {
            for (uint32_t q = 0; q < ha.counts[l - 1]; ++q, ++k, ++code)
                if (ha.vals[k] == 0x00) { el = l; ec = code; break; }
            code <<= 1;
        }
        if (!el || el > 7) continue;
        const bool is_ac = (uint32_t)pair_of[sl] == sl;
        uint16_t *look = reinterpret_cast<uint16_t *>(tables + (size_t)sl * kJhTableWords);
        for (uint32_t idx = 0; idx < (1u << kJhLookBits); ++idx) {
            const uint32_t e = look[idx], l = e & 31u, sym = e >> 8, sz = sym & 15u;
            if (!l || (is_ac ? sz == 0u : sym > 11u)) continue;
            const uint32_t tot = l + sz + el;
            if (tot > kJhLookBits) continue;
            if (((idx >> (kJhLookBits - tot)) & ((1u << el) - 1u)) == ec) look[idx] = (uint16_t)(e | (el << 5));
        }
    }
    // blocks of one MCU in coding order (A.2.3): component by component, rows of its h x v group
    uint32_t bpm = 0;
    for (uint32_t i = 0; i < nc; ++i)
        for (uint32_t v = 0; v < H.comp[i].v; ++v)
            for (uint32_t hh = 0; hh < H.comp[i].h; ++hh) {
                if (bpm >= 10) return -2;
                S.blk_comp[bpm] = (uint8_t)i; S.blk_h[bpm] = (uint8_t)hh; S.blk_v[bpm] = (uint8_t)v;
                ++bpm;
            }
    S.bpm = bpm;
    S.mcux = H.comp[0].bw / H.comp[0].h; S.mcuy = H.comp[0].bh / H.comp[0].v;
    S.total_blocks = H.nblocks;
    if ((uint64_t)S.mcux * S.mcuy * bpm != H.nblocks) return -2;
    S.tables_off = (uint32_t)(sizeof(JpegBlobHeader) + sizeof(JpegHuffStage));
    S.stream_off = (uint32_t)((S.tables_off + 4u * kJhTableWords * 4u + 15u) & ~15u);
    // the segment without its stuffing: 0xFF 0x00 -> 0xFF; it ends at the first marker that is not a restart marker.  A restart marker
    // (only in a file that announced an interval) is taken out as well: the next interval starts at the byte that follows in the copy.
    uint8_t *o = out + S.stream_off;
    const uint8_t *p = data + P.scan_pos, *end = data + n;
    std::vector<uint32_t> rst;
    uint32_t next_rst = 0; // RSTm markers count modulo 8 (B.2.1)
    while (p < end) {
        const uint8_t *ff = static_cast<const uint8_t *>(memchr(p, 0xFF, (size_t)(end - p)));
        if (!ff) { memcpy(o, p, (size_t)(end - p)); o += end - p; break; }
        memcpy(o, p, (size_t)(ff - p)); o += ff - p;
        if (ff + 1 >= end) break;
        if (ff[1] == 0x00) { *o++ = 0xFF; p = ff + 2; continue; }
        if (ff[1] == 0xFF) { p = ff + 1; continue; }              // fill bytes in front of a marker
        if (ff[1] >= 0xD0 && ff[1] <= 0xD7) {
            // without an announced interval, out of sequence, or in a number the device's tables are not sized for: the host decoder's case
            if (!P.info.restart_interval || ff[1] != 0xD0 + next_rst || rst.size() >= kJhMaxRestarts) return -2;
            next_rst = (next_rst + 1u) & 7u;
            rst.push_back((uint32_t)(o - (out + S.stream_off)));
            p = ff + 2;
            continue;
        }
        break;                                                      // EOI or any other marker: the scan is over
    }
    if (P.info.restart_interval) {
        // the host decoder wants exactly one marker behind every interval but the last (fl_jpeghuff.cpp jpeg_entropy_decode); a file that
        // differs is broken one way or another, and which way is for that decoder to say
        const uint64_t mcus = (uint64_t)S.mcux * S.mcuy, want = (mcus + P.info.restart_interval - 1u) / P.info.restart_interval;
        if (rst.size() + 1u != want) return -2;
        for (size_t i = 1; i < rst.size(); ++i) if (rst[i] <= rst[i - 1]) return -2; // (an empty interval)
        if (!rst.empty() && rst[0] == 0u) return -2;
    }
    const size_t stream_bytes = (size_t)(o - (out + S.stream_off));
    if (stream_bytes > (1u << 28)) return -2;                      // (bit positions are 32-bit on the device)
    // A picture of one colour is a PERIODIC stream ("DC difference 0, end of block" over and over, ~5 bits per block): a decoder started at
    // a wrong bit falls into a shifted parse that is just as valid and never meets the true one, so the device's chain of states would have
    // to walk the file subsequence by subsequence (tests/tools/fuzz_jpegdec.py: every flat picture ended in the host retry, after ~3 ms of
    // kernels).  Such files are tiny and the host decodes them in a fraction of that: below 7 bits per block the file is not for this path.
    if ((uint64_t)stream_bytes * 8u < 7ull * H.nblocks) return -2;
    memset(o, 0xFF, 32);                                           // the device reads whole words, up to 16 bytes past the end
    S.stream_bits = (uint32_t)(stream_bytes * 8u);
    S.staged_bytes = (uint32_t)(S.stream_off + ((stream_bytes + 16u + 15u) & ~(size_t)15u));
    S.rst_mcus = P.info.restart_interval;
    S.n_rst = (uint32_t)rst.size();
    S.rst_off = S.staged_bytes;
    if (!rst.empty()) {
        if ((size_t)S.staged_bytes + rst.size() * 4u + 16u > cap) return -2;
        memcpy(out + S.rst_off, rst.data(), rst.size() * 4u);
        S.staged_bytes = (uint32_t)((S.rst_off + rst.size() * 4u + 15u) & ~(size_t)15u);
    }
    // the header describes the blob the DEVICE builds: block words, then 64 halfwords per block
    H.magic = kJhMagic;
    H.blocks_off = (uint32_t)sizeof(JpegBlobHeader);
    H.coef_off = (H.blocks_off + H.nblocks * 4u + 15u) & ~15u;
    H.total_bytes = S.staged_bytes;
    memcpy(out, &H, sizeof(H));
    memcpy(out + sizeof(H), &S, sizeof(S));
    if (used) *used = S.staged_bytes;
    return 0;
}

int jpeg_entropy_decode(const uint8_t *data, size_t n, uint8_t *blob, size_t cap, size_t *used)
{
    Parsed P;
    int rc = parse(data, n, P, true);
    if (rc) return rc;
    if (!P.info.supported) return -2;
    const uint32_t nc = P.info.components;
    for (uint32_t i = 0; i < nc; ++i) if (!P.have_qt[P.c[i].tq]) return -1;
    JpegBlobHeader H;
    layout(P, H);
    if ((size_t)H.coef_off + (size_t)H.nblocks * 128 + 64 > cap) return -1;
    uint32_t *words = reinterpret_cast<uint32_t *>(blob + H.blocks_off);
    uint8_t *coef = blob + H.coef_off; // block data, 2-byte aligned, see fl_jpegdec.h
    size_t nhalf = 0;                  // halfwords written
    if (!P.single_pass) {
        // progressive, or sequential in several scans: assemble all coefficients first (decode_scans), then pack them
        // (the two arrays are the thread's: a worker that decodes progressive files one after the other pays for the pages once)
        thread_local std::vector<int16_t> all;
        thread_local std::vector<uint64_t> nzm;
        all.assign((size_t)H.nblocks * 64, 0);
        nzm.assign(H.nblocks, 0ull);
        rc = decode_scans(data, n, P, H, all.data(), nzm.data());
        // (a thread keeps the arrays of ordinary pictures between calls -- up to 64 MB, a 33-megapixel frame -- and gives a larger one back)
        struct Release { std::vector<int16_t> &a; std::vector<uint64_t> &m; ~Release() { if (a.capacity() > ((size_t)64 << 20) / 2) { std::vector<int16_t>().swap(a); std::vector<uint64_t>().swap(m); } } } release_{all, nzm};
        if (rc) return rc;
        for (uint32_t i = 0; i < nc; ++i) memcpy(H.qt[i], P.qt[P.c[i].tq], 128); // (a table may have been redefined between scans)
        for (uint32_t bi = 0; bi < H.nblocks; ++bi) {
            const int16_t *blk = all.data() + (size_t)bi * 64;
            // the last non-zero coefficient and "every coefficient behind the head fits a byte" from the mask of non-zero ones
            const uint64_t nz = nzm[bi] & ~1ull;
            const int last = nz ? 63 - __builtin_clzll(nz) : 0;
            bool narrow = true;
            for (uint64_t m = kJpegWideHead < 64 ? nz & (~0ull << kJpegWideHead) : 0ull; m; m &= m - 1ull) {
                const int k = __builtin_ctzll(m);
                if (blk[k] < -128 || blk[k] > 127) { narrow = false; break; }
            }
            const uint32_t cnt = (uint32_t)last + 1;
            if (nhalf >= ((size_t)1 << 25)) return -2; // block words carry 25 offset bits
            words[bi] = ((uint32_t)nhalf << 7) | ((cnt - 1u) << 1) | (narrow ? 0u : 1u);
            uint8_t *o = coef + nhalf * 2;
            const uint32_t head = narrow ? (cnt < kJpegWideHead ? cnt : kJpegWideHead) : cnt;
            size_t bytes = head * 2;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
            if (narrow && cnt > head) {
                // coefficients head .. 63 as bytes in one go (SSE2: part of x86-64), written at o + 2 head - head so that byte k lands at
                // o[head + k]; behind the last coefficient they are zeros, the padding byte among them; the blob has 64 bytes of slack
                // behind its last block (checked above) and the next block overwrites what this one wrote too far
                for (int q = 0; q < 4; ++q) {
                    const __m128i lo = _mm_loadu_si128(reinterpret_cast<const __m128i *>(blk + 16 * q)), hi = _mm_loadu_si128(reinterpret_cast<const __m128i *>(blk + 16 * q + 8));
                    _mm_storeu_si128(reinterpret_cast<__m128i *>(o + head + 16 * q), _mm_packs_epi16(lo, hi));
                }
                memcpy(o, blk, head * 2);
                bytes += cnt - head;
                bytes += bytes & 1u;
            } else
#endif
            {
                memcpy(o, blk, head * 2);
                for (uint32_t k = head; k < cnt; ++k) o[bytes++] = (uint8_t)(int8_t)blk[k];
                if (bytes & 1u) o[bytes++] = 0;
            }
            nhalf += bytes / 2;
        }
        size_t ncoef2 = (nhalf + 7) & ~(size_t)7;
        H.total_bytes = (uint32_t)(H.coef_off + ncoef2 * 2);
        memcpy(blob, &H, sizeof(H));
        if (used) *used = H.total_bytes;
        return 0;
    }
    for (uint32_t i = 0; i < nc; ++i)
        if (!P.ht[0][P.c[i].td].present || !P.ht[1][P.c[i].ta].present) return -1;
    BitReader br{data, n, P.scan_pos};
    int pred[4] = {0, 0, 0, 0};
    const uint32_t mcux = H.comp[0].bw / H.comp[0].h, mcuy = H.comp[0].bh / H.comp[0].v;
    uint32_t rst_left = P.info.restart_interval;
    for (uint32_t my = 0; my < mcuy; ++my)
        for (uint32_t mx = 0; mx < mcux; ++mx) {
            if (P.info.restart_interval && rst_left == 0) {
                // F.2.2.4: the interval ends byte aligned on RSTn; predictors start again at zero
                if (!br.marker) {
                    // fewer than 8 padding bits are buffered in a valid stream; whole bytes would be data nobody coded
                    br.pos -= (size_t)(br.cnt / 8);
                    if (br.pos + 2 > n || data[br.pos] != 0xFF || data[br.pos + 1] < 0xD0 || data[br.pos + 1] > 0xD7) return -1;
                    br.pos += 2;
                } else if (br.marker >= 0xD0 && br.marker <= 0xD7) {
                    br.marker = 0;
                } else return -1;
                br.buf = 0; br.cnt = 0;
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                rst_left = P.info.restart_interval;
            }
            for (uint32_t i = 0; i < nc; ++i) {
                const JpegComponent &c = H.comp[i];
                const Huff &hd = P.ht[0][P.c[i].td], &ha = P.ht[1][P.c[i].ta];
                for (uint32_t v = 0; v < c.v; ++v)
                    for (uint32_t h = 0; h < c.h; ++h) {
                        const uint32_t bx = mx * c.h + h, by = my * c.v + v;
                        if (nhalf >= ((size_t)1 << 25)) return -2; // block words carry 25 offset bits
                        // The block is written in its blob form as it is decoded (fl_jpegdec.h: the first kJpegWideHead
                        // coefficients as i16, the rest as i8 up to the last non-zero one): the 72 bytes it can take are cleared
                        // and only the coded coefficients are stored.  A coefficient past the head that does not fit a byte
                        // makes the block "wide" (all i16): rare, and handled by decoding the block again the plain way.
                        uint8_t *o = coef + nhalf * 2;
                        memset(o, 0, 72);
                        const BitReader at_block = br;
                        const int pred_before = pred[i];
                        // The bit buffer lives in registers for the length of the block (br is visible to the out-of-line refill,
                        // so its members are memory to the compiler).
                        uint64_t buf = br.buf;
                        int bc = br.cnt;
                        {   // DC difference: category + magnitude bits in one look-up when they fit the table's reach (as for AC below)
                            if (bc < 32) { br.buf = buf; br.cnt = bc; br.fill(); buf = br.buf; bc = br.cnt; }
                            const uint32_t e = hd.fastx[buf >> (64 - kAcBits)];
                            if (e) {
                                const int nb = (int)(e & 31u);
                                buf <<= nb; bc -= nb;
                                // (the table is the AC one's shape: a DC symbol is a bare category, so an entry with a run, or the
                                // ZRL flag, is a symbol no DC table may hold -- what the slow path rejects as t > 11)
                                if (e & (1u << 12)) { if (!(e & (1u << 13))) return -1; }
                                else if ((e >> 5) & 15u) return -1;
                                else pred[i] += (int)(int16_t)(e >> 16);
                            } else {
                                br.buf = buf; br.cnt = bc;
                                const int t = decode_sym(br, hd);
                                if (t < 0 || t > 11) return -1;
                                pred[i] += receive_extend(br, t);
                                buf = br.buf; bc = br.cnt;
                            }
                        }
                        if (pred[i] < -32768 || pred[i] > 32767) return -1;
                        const int16_t dc = (int16_t)pred[i];
                        memcpy(o, &dc, 2);
                        int last = 0;
                        uint32_t wide = 0;
                        for (int k = 1; k < 64;) {
                            if (bc < 32) {
                                uint64_t raw;
                                bool done = false;
                                if (!br.marker && br.pos + 8 <= n) {
                                    memcpy(&raw, data + br.pos, 8);
                                    const uint64_t inv = ~raw;
                                    if (!((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull)) { // no 0xFF among them
                                        // whole bytes that fit are counted; the bits of the next byte that also land in the
                                        // buffer are the ones the next refill ORs onto themselves
                                        buf |= __builtin_bswap64(raw) >> bc;
                                        br.pos += (size_t)((63 - bc) >> 3);
                                        bc |= 56;
                                        done = true;
                                    }
                                }
                                if (!done) { br.buf = buf; br.cnt = bc; br.fill(); buf = br.buf; bc = br.cnt; }
                            }
                            int val;
                            const uint32_t e = ha.fastx[buf >> (64 - kAcBits)];
                            if (e) {
                                const int nb = (int)(e & 31u);
                                buf <<= nb; bc -= nb;
                                if (e & (1u << 12)) {
                                    if (e & (1u << 13)) break; // end of block
                                    k += 16;                   // ZRL
                                    continue;
                                }
                                k += (int)((e >> 5) & 15u);
                                val = (int16_t)(e >> 16);
                            } else {
                                br.buf = buf; br.cnt = bc;
                                const int rs = decode_sym(br, ha);
                                if (rs < 0) return -1;
                                const int r = rs >> 4, s = rs & 15;
                                if (s == 0) {
                                    buf = br.buf; bc = br.cnt;
                                    if (r == 15) { k += 16; continue; }
                                    break;
                                }
                                k += r;
                                val = receive_extend(br, s);
                                buf = br.buf; bc = br.cnt;
                            }
                            if (k > 63) return -1;
                            if (k < (int)kJpegWideHead) { const int16_t v16 = (int16_t)val; memcpy(o + 2 * k, &v16, 2); }
                            else { o[kJpegWideHead + k] = (uint8_t)(int8_t)val; wide |= (uint32_t)(val + 128) >> 8; }
                            last = k;
                            ++k;
                        }
                        br.buf = buf; br.cnt = bc;
                        uint32_t *word = &words[c.block_base + by * c.bw + bx];
                        if (!wide) {
                            const size_t bytes = last < (int)kJpegWideHead ? 2u * ((size_t)last + 1u) : (size_t)last + kJpegWideHead + 1u;
                            *word = ((uint32_t)nhalf << 7) | ((uint32_t)last << 1);
                            nhalf += (bytes + 1u) / 2u;
                            continue;
                        }
                        // wide block: once more, into 64 halfwords
                        br = at_block;
                        pred[i] = pred_before;
                        int16_t blk[64];
                        memset(blk, 0, sizeof(blk));
                        {
                            const int t = decode_sym(br, hd);
                            if (t < 0 || t > 11) return -1;
                            pred[i] += receive_extend(br, t);
                        }
                        blk[0] = (int16_t)pred[i];
                        last = 0;
                        for (int k = 1; k < 64;) {
                            const int rs = decode_sym(br, ha);
                            if (rs < 0) return -1;
                            const int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r == 15) { k += 16; continue; }
                                break;
                            }
                            k += r;
                            if (k > 63) return -1;
                            blk[k] = (int16_t)receive_extend(br, s);
                            last = k;
                            ++k;
                        }
                        *word = ((uint32_t)nhalf << 7) | ((uint32_t)last << 1) | 1u;
                        memcpy(o, blk, 2u * ((size_t)last + 1u));
                        nhalf += (size_t)last + 1u;
                    }
            }
            if (P.info.restart_interval) rst_left--;
        }
    size_t ncoef = nhalf;
    ncoef = (ncoef + 7) & ~(size_t)7;
    H.total_bytes = (uint32_t)(H.coef_off + ncoef * 2);
    memcpy(blob, &H, sizeof(H));
    if (used) *used = H.total_bytes;
    return 0;
}

} // namespace fl
