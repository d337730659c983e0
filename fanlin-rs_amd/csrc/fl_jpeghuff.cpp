// fl_jpeghuff.cpp -- host half of the JPEG decode front end: marker parsing (ITU-T T.81 Annex B) and the sequential
// Huffman decoder (Annex F.2), table driven, writing the compact coefficient blob the device kernels consume
// (fl_jpegdec.h).  Reference: src/handler.rs:205-220 (ImageReader -> JpegDecoder::new -> DynamicImage::from_decoder,
// zune-jpeg 0.4.14) and handler.rs:206 (decoder.orientation(): the EXIF tag).
#include "fl_jpegdec.h"

#include <string.h>

#include <map>

namespace fl {

namespace {

inline uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

struct Huff {
    // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
    uint16_t fast[512];
    int32_t maxcode[18];  // per length, -1 = none; [17] = sentinel
    int32_t valoff[17];   // symbol index of the first code of a length minus that code
    uint8_t vals[256];
    // AC tables only: code AND magnitude bits inside the 9-bit lookahead -> (value << 8) | (run << 4) | total bits, 0 = slow path
    int16_t fastac[512];
    bool present = false;
};

bool build_huff(Huff &h, const uint8_t *bits /*[16] counts of lengths 1..16*/, const uint8_t *vals, int total)
{
    memcpy(h.vals, vals, (size_t)total);
    memset(h.fast, 0, sizeof(h.fast));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valoff[l] = k - code;
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            if (l <= 9) {
                const int first = code << (9 - l), cnt = 1 << (9 - l);
                if (first + cnt > 512) return false;
                for (int f = 0; f < cnt; ++f) h.fast[first + f] = (uint16_t)((l << 8) | vals[k]);
            }
        }
        if (code > (1 << l)) return false; // over-subscribed
        h.maxcode[l] = bits[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    // short code + short magnitude in one lookup (the common case of AC coefficients: small values after short runs)
    for (int i = 0; i < 512; ++i) {
        h.fastac[i] = 0;
        const uint32_t f = h.fast[i];
        if (!f) continue;
        const int len = (int)(f >> 8), rs = (int)(f & 255u), run = rs >> 4, mag = rs & 15;
        if (mag == 0 || len + mag > 9) continue;
        int v = ((i << len) & 511) >> (9 - mag);           // the magnitude bits that follow the code
        if (v < (1 << (mag - 1))) v += (int)(~0u << mag) + 1; // F.2.2.1 EXTEND
        if (v >= -128 && v <= 127) h.fastac[i] = (int16_t)((v * 256) | (run << 4) | (len + mag));
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
    const uint32_t f = h.fast[br.peek(9)];
    if (f) { br.drop((int)(f >> 8)); return (int)(f & 255u); }
    // longer than 9 bits: canonical search
    uint32_t code = br.peek(10);
    int l = 10;
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
    bool one_scan = false; // SOS names every component in frame order
    std::map<uint32_t, std::vector<uint8_t>> icc_chunks;
    uint32_t icc_count = 0;
};

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
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (pl < 6 || got_sof) return -1; // (a second frame header: T.81 allows one per image; zune-jpeg rejects it too)
            P.info.progressive = m == 0xC2;
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
        } else if (m == 0xDD) {
            if (pl < 2) return -1;
            P.info.restart_interval = be16(p);
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
            if (pl < 1 + 2 * ns + 3) return -1;
            P.one_scan = ns == nc;
            for (uint32_t i = 0; i < ns && P.one_scan; ++i) {
                if (P.c[i].id != p[1 + 2 * i]) { P.one_scan = false; break; }
                P.c[i].td = p[2 + 2 * i] >> 4; P.c[i].ta = p[2 + 2 * i] & 15u;
                if (P.c[i].td > 3 || P.c[i].ta > 3) return -1;
            }
            P.scan_pos = pos + len;
            bool ok = !P.info.progressive && P.info.precision == 8 && P.one_scan && (nc == 1 || nc == 3 || nc == 4);
            for (uint32_t i = 0; ok && nc > 1 && i < nc; ++i) // every plane at full or half resolution per direction
                ok = P.info.hmax % P.c[i].h == 0 && P.info.vmax % P.c[i].v == 0 && P.info.hmax / P.c[i].h <= 2 && P.info.vmax / P.c[i].v <= 2;
            P.info.supported = ok ? 1u : 0u;
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

} // namespace

int jpeg_parse_info(const uint8_t *data, size_t n, JpegInfo &info)
{
    Parsed P;
    const int rc = parse(data, n, P, false);
    info = P.info;
    return rc;
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

int jpeg_entropy_decode(const uint8_t *data, size_t n, uint8_t *blob, size_t cap, size_t *used)
{
    Parsed P;
    int rc = parse(data, n, P, true);
    if (rc) return rc;
    if (!P.info.supported) return -2;
    const uint32_t nc = P.info.components;
    for (uint32_t i = 0; i < nc; ++i)
        if (!P.have_qt[P.c[i].tq] || !P.ht[0][P.c[i].td].present || !P.ht[1][P.c[i].ta].present) return -1;
    JpegBlobHeader H;
    layout(P, H);
    if ((size_t)H.coef_off + (size_t)H.nblocks * 128 + 64 > cap) return -1;
    uint32_t *words = reinterpret_cast<uint32_t *>(blob + H.blocks_off);
    uint8_t *coef = blob + H.coef_off; // block data, 2-byte aligned, see fl_jpegdec.h
    size_t nhalf = 0;                  // halfwords written
    BitReader br{data, n, P.scan_pos};
    int pred[4] = {0, 0, 0, 0};
    const uint32_t mcux = H.comp[0].bw / H.comp[0].h, mcuy = H.comp[0].bh / H.comp[0].v;
    uint32_t rst_left = P.info.restart_interval;
    int16_t blk[64];
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
                        const int t = decode_sym(br, hd);
                        if (t < 0 || t > 11) return -1;
                        pred[i] += receive_extend(br, t);
                        if (pred[i] < -32768 || pred[i] > 32767) return -1;
                        blk[0] = (int16_t)pred[i];
                        int last = 0;
                        bool narrow = true;
                        for (int k = 1; k < 64;) {
                            if (br.cnt < 16) br.fill();
                            const int fa = ha.fastac[br.peek(9)];
                            if (fa) { // code + magnitude in one step
                                k += (fa >> 4) & 15;
                                if (k > 63) return -1;
                                br.drop(fa & 15);
                                while (last + 1 < k) blk[++last] = 0;
                                blk[k] = (int16_t)(fa >> 8);
                                last = k;
                                ++k;
                                continue;
                            }
                            const int rs = decode_sym(br, ha);
                            if (rs < 0) return -1;
                            const int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r == 15) { k += 16; continue; }
                                break;
                            }
                            k += r;
                            if (k > 63) return -1;
                            while (last + 1 < k) blk[++last] = 0;
                            const int val = receive_extend(br, s);
                            blk[k] = (int16_t)val;
                            if (k >= (int)kJpegWideHead && (val < -128 || val > 127)) narrow = false;
                            last = k;
                            ++k;
                        }
                        const uint32_t cnt = (uint32_t)last + 1;
                        if (nhalf >= ((size_t)1 << 25)) return -2; // block words carry 25 offset bits
                        words[c.block_base + by * c.bw + bx] = ((uint32_t)nhalf << 7) | ((cnt - 1u) << 1) | (narrow ? 0u : 1u);
                        uint8_t *o = coef + nhalf * 2;
                        const uint32_t head = narrow ? (cnt < kJpegWideHead ? cnt : kJpegWideHead) : cnt;
                        memcpy(o, blk, head * 2);
                        size_t bytes = head * 2;
                        for (uint32_t k = head; k < cnt; ++k) o[bytes++] = (uint8_t)(int8_t)blk[k];
                        if (bytes & 1u) o[bytes++] = 0;
                        nhalf += bytes / 2;
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
