/*
 * fanlin_oracle_jpegdec.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): baseline JPEG decoding as the
 * reference performs it at src/handler.rs:205-220 through `image 0.25.6` -> `zune-jpeg 0.4.14`
 * (Cargo.lock:4482-4483).
 *
 * PARITY UNPINNED.  zune-jpeg's sources are not under /root/reference and no Rust toolchain exists here.  The
 * entropy decoder is fully specified by ITU-T T.81 (Annex F) and is restated from the standard -- here with the
 * bit-by-bit procedure of F.2.2.3 (DECODE / RECEIVE / EXTEND), deliberately the simplest possible form, so that it
 * shares nothing with the table-driven decoder of the product (csrc/fl_jpeghuff.cpp).  The three lossy-stage choices a
 * decoder is free in are restated from the published zune-jpeg 0.4.14 sources as recalled:
 *   idct     src/idct/scalar.rs `idct_int`: the 12-bit fixed point butterfly of stb_image (constants 2217, -7567, 3135,
 *            4816, 1223, 8410, 12586, 6149, -3685, -10497, -8034, -1597), column pass + 512 >> 10, row pass
 *            + 65536 + (128 << 17) >> 17, clamped to 0..255; dequantisation happens before it, in i32
 *   upsample src/upsampler/scalar.rs: horizontal (3*a + b + 2) >> 2 with the two end samples copied, vertical
 *            (3*near + far + 2) >> 2, h2v2 = vertical into a scratch row THEN horizontal on the rounded values (two
 *            roundings, unlike libjpeg's single 16ths rounding); rows beyond the picture repeat the edge row
 *   colour   src/color_convert/scalar.rs: r = y + ((45 * cr) >> 5), g = y - ((11 * cb + 23 * cr) >> 5),
 *            b = y + ((113 * cb) >> 6) on cb - 128 / cr - 128 in i16, clamped (5/6-bit constants, NOT libjpeg's 16-bit ones)
 * tests/test_jpeg_decode.py bounds the distance of this restatement to libjpeg-turbo's JDCT_ISLOW decoder (Pillow) on
 * the reference's own images/lenna.jpg and on synthetic streams; that bound (<= 4 LSB, mean < 1) is what is pinned,
 * not equality with zune-jpeg.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fanlin_oracle.h"

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;        /* blocks per row / column, padded to whole MCUs */
    int pw, ph;        /* plane size in samples (bw*8, bh*8) */
    int w, h_px;       /* real size of this component's plane: ceil(W * h / hmax), ceil(H * v / vmax) */
    uint8_t *plane;
    int pred;
    int sink_base; /* index of this component's first block in the coefficient sink */
} jd_comp;

typedef struct {
    const uint8_t *d;
    size_t n, pos;
    uint32_t bitbuf;
    int bitcnt;
    int marker; /* marker met inside the entropy-coded segment (0 = none) */
    /* Huffman tables, T.81 Annex C / F.2.2.3 */
    uint8_t bits[2][4][17];
    uint8_t vals[2][4][256];
    int mincode[2][4][17], maxcode[2][4][18], valptr[2][4][17];
    int have_ht[2][4];
    uint16_t qt[4][64]; /* zig-zag order, as stored in the file */
    int have_qt[4];
    int W, H, nc, hmax, vmax, restart, progressive, precision;
    int adobe_transform; /* -1 = no APP14 */
    int orientation;
    int16_t *sink; /* optional: quantised coefficients, [block][64] in zig-zag order, components one after the other */
    jd_comp c[4];
} jd;

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int u16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

static void build_ht(jd *j, int tc, int th)
{
    /* C.2 + F.2.2.3: codes of each length are consecutive; mincode / maxcode / valptr per length */
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        j->valptr[tc][th][l] = k;
        j->mincode[tc][th][l] = code;
        code += j->bits[tc][th][l];
        k += j->bits[tc][th][l];
        j->maxcode[tc][th][l] = j->bits[tc][th][l] ? code - 1 : -1;
        code <<= 1;
    }
    j->maxcode[tc][th][17] = 0x7fffffff;
}

/* EXIF orientation (tag 0x0112) from an APP1 "Exif\0\0" segment; 0 if absent / malformed */
static int exif_orientation(const uint8_t *p, int len)
{
    if (len < 14 || memcmp(p, "Exif\0\0", 6)) return 0;
    const uint8_t *t = p + 6;
    const int n = len - 6;
    int le;
    if (t[0] == 'I' && t[1] == 'I') le = 1; else if (t[0] == 'M' && t[1] == 'M') le = 0; else return 0;
#define RD16(o) (le ? (t[o] | (t[(o) + 1] << 8)) : ((t[o] << 8) | t[(o) + 1]))
#define RD32(o) (le ? ((uint32_t)t[o] | ((uint32_t)t[(o) + 1] << 8) | ((uint32_t)t[(o) + 2] << 16) | ((uint32_t)t[(o) + 3] << 24)) \
                    : (((uint32_t)t[o] << 24) | ((uint32_t)t[(o) + 1] << 16) | ((uint32_t)t[(o) + 2] << 8) | (uint32_t)t[(o) + 3]))
    if (RD16(2) != 42) return 0;
    uint32_t ifd = RD32(4);
    if (ifd + 2 > (uint32_t)n) return 0;
    int cnt = RD16(ifd);
    for (int i = 0; i < cnt; ++i) {
        uint32_t e = ifd + 2 + 12u * (uint32_t)i;
        if (e + 12 > (uint32_t)n) return 0;
        if (RD16(e) == 0x0112) { int v = RD16(e + 8); return v >= 1 && v <= 8 ? v : 0; }
    }
    return 0;
#undef RD16
#undef RD32
}

/* Parses everything up to and including SOS.  0 = ok, -1 = malformed, -2 = not a baseline sequential 8-bit Huffman stream */
static int parse(jd *j)
{
    if (j->n < 4 || j->d[0] != 0xFF || j->d[1] != 0xD8) return -1;
    j->pos = 2;
    j->adobe_transform = -1;
    int got_sof = 0;
    for (;;) {
        if (j->pos + 4 > j->n) return -1;
        if (j->d[j->pos] != 0xFF) return -1;
        while (j->pos < j->n && j->d[j->pos] == 0xFF) j->pos++; /* fill bytes */
        if (j->pos >= j->n) return -1;
        const int m = j->d[j->pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -1;
        if (j->pos + 2 > j->n) return -1;
        const int len = u16(j->d + j->pos);
        if (len < 2 || j->pos + (size_t)len > j->n) return -1;
        const uint8_t *p = j->d + j->pos + 2;
        const int pl = len - 2;
        if (m == 0xDB) { /* DQT */
            int o = 0;
            while (o < pl) {
                const int pq = p[o] >> 4, tq = p[o] & 15;
                if (tq > 3 || pq > 1) return -1;
                o++;
                if (o + 64 * (pq + 1) > pl) return -1;
                for (int k = 0; k < 64; ++k) { j->qt[tq][k] = pq ? (uint16_t)u16(p + o + 2 * k) : p[o + k]; }
                o += 64 * (pq + 1);
                j->have_qt[tq] = 1;
            }
        } else if (m == 0xC4) { /* DHT */
            int o = 0;
            while (o < pl) {
                const int tc = p[o] >> 4, th = p[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > pl) return -1;
                int total = 0;
                j->bits[tc][th][0] = 0;
                for (int l = 1; l <= 16; ++l) { j->bits[tc][th][l] = p[o + l]; total += p[o + l]; }
                o += 17;
                if (total > 256 || o + total > pl) return -1;
                memcpy(j->vals[tc][th], p + o, (size_t)total);
                o += total;
                build_ht(j, tc, th);
                j->have_ht[tc][th] = 1;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { /* SOF0 / SOF1 (extended sequential, Huffman) / SOF2 (progressive) */
            if (pl < 6) return -1;
            j->progressive = m == 0xC2;
            j->precision = p[0];
            j->H = u16(p + 1);
            j->W = u16(p + 3);
            j->nc = p[5];
            if (j->nc < 1 || j->nc > 4 || pl < 6 + 3 * j->nc || !j->W || !j->H) return -1;
            for (int i = 0; i < j->nc; ++i) {
                j->c[i].id = p[6 + 3 * i];
                j->c[i].h = p[7 + 3 * i] >> 4;
                j->c[i].v = p[7 + 3 * i] & 15;
                j->c[i].tq = p[8 + 3 * i];
                if (j->c[i].h < 1 || j->c[i].h > 4 || j->c[i].v < 1 || j->c[i].v > 4 || j->c[i].tq > 3) return -1;
                if (j->c[i].h > j->hmax) j->hmax = j->c[i].h;
                if (j->c[i].v > j->vmax) j->vmax = j->c[i].v;
            }
            got_sof = 1;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return -2; /* lossless, differential or arithmetic coding */
        } else if (m == 0xDD) {
            if (pl < 2) return -1;
            j->restart = u16(p);
        } else if (m == 0xEE) {
            if (pl >= 12 && !memcmp(p, "Adobe", 5)) j->adobe_transform = p[11];
        } else if (m == 0xE1) {
            if (!j->orientation) j->orientation = exif_orientation(p, pl);
        } else if (m == 0xDA) { /* SOS */
            if (!got_sof || pl < 1) return -1;
            const int ns = p[0];
            if (pl < 1 + 2 * ns + 3) return -1;
            if (j->progressive || j->precision != 8) return -2;
            if (ns != j->nc) return -2; /* only one scan holding every component (what encoders emit for baseline files) */
            for (int i = 0; i < ns; ++i) {
                int ci = -1;
                for (int k = 0; k < j->nc; ++k) if (j->c[k].id == p[1 + 2 * i]) ci = k;
                if (ci != i) return -2;
                j->c[ci].td = p[2 + 2 * i] >> 4;
                j->c[ci].ta = p[2 + 2 * i] & 15;
                if (j->c[ci].td > 3 || j->c[ci].ta > 3) return -1;
            }
            j->pos += (size_t)len;
            return 0;
        }
        j->pos += (size_t)len;
    }
}

/* one bit of the entropy-coded segment; byte stuffing (FF 00) removed, a marker ends the data (zeros are supplied) */
static int next_bit(jd *j)
{
    if (j->bitcnt == 0) {
        int b = 0;
        if (!j->marker && j->pos < j->n) {
            b = j->d[j->pos++];
            if (b == 0xFF) {
                int b2 = j->pos < j->n ? j->d[j->pos] : 0xD9;
                if (b2 == 0) j->pos++;
                else { j->marker = b2; j->pos++; b = 0; }
            }
        }
        j->bitbuf = (uint32_t)b;
        j->bitcnt = 8;
    }
    j->bitcnt--;
    return (int)((j->bitbuf >> j->bitcnt) & 1u);
}

static int decode_sym(jd *j, int tc, int th)
{
    /* F.2.2.3 DECODE */
    int code = next_bit(j), l = 1;
    while (l <= 16 && code > j->maxcode[tc][th][l]) { code = (code << 1) | next_bit(j); ++l; }
    if (l > 16) return -1;
    return j->vals[tc][th][j->valptr[tc][th][l] + code - j->mincode[tc][th][l]];
}

static int receive_extend(jd *j, int s)
{
    if (!s) return 0;
    int v = 0;
    for (int i = 0; i < s; ++i) v = (v << 1) | next_bit(j);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; /* F.2.2.1 EXTEND */
}

/* zune-jpeg 0.4.14 src/idct/scalar.rs idct_int (= stb_image stbi__idct_block); in[] dequantised, natural order */
static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

static void idct_block(const int32_t *in, uint8_t *out, int stride)
{
    int32_t v[64];
    int all_ac_zero = 1;
    for (int i = 1; i < 64; ++i) if (in[i]) { all_ac_zero = 0; break; }
    if (all_ac_zero) {
        /* whole-block shortcut of idct_int: (dc + 4 + 1024) >> 3, clamped -- what the full path gives for a DC-only block */
        const uint8_t s = clamp8((in[0] + 4 + 1024) >> 3);
        for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) out[y * stride + x] = s;
        return;
    }
#define F2F(x) ((int32_t)((x) * 4096.0 + 0.5))
    for (int c = 0; c < 8; ++c) {
        const int32_t *d = in + c;
        if (!d[8] && !d[16] && !d[24] && !d[32] && !d[40] && !d[48] && !d[56]) {
            const int32_t dc = d[0] * 4;
            for (int r = 0; r < 8; ++r) v[r * 8 + c] = dc;
            continue;
        }
        int32_t p2 = d[16], p3 = d[48];
        int32_t p1 = (p2 + p3) * 2217;
        int32_t t2 = p1 + p3 * -7567;
        int32_t t3 = p1 + p2 * 3135;
        p2 = d[0]; p3 = d[32];
        int32_t t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        const int32_t x0 = t0 + t3 + 512, x3 = t0 - t3 + 512, x1 = t1 + t2 + 512, x2 = t1 - t2 + 512;
        t0 = d[56]; t1 = d[40]; t2 = d[24]; t3 = d[8];
        p3 = t0 + t2;
        int32_t p4 = t1 + t3;
        p1 = t0 + t3; p2 = t1 + t2;
        const int32_t p5 = (p3 + p4) * 4816;
        t0 *= 1223; t1 *= 8410; t2 *= 12586; t3 *= 6149;
        p1 = p5 + p1 * -3685; p2 = p5 + p2 * -10497; p3 = p3 * -8034; p4 = p4 * -1597;
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        v[0 * 8 + c] = (x0 + t3) >> 10; v[7 * 8 + c] = (x0 - t3) >> 10;
        v[1 * 8 + c] = (x1 + t2) >> 10; v[6 * 8 + c] = (x1 - t2) >> 10;
        v[2 * 8 + c] = (x2 + t1) >> 10; v[5 * 8 + c] = (x2 - t1) >> 10;
        v[3 * 8 + c] = (x3 + t0) >> 10; v[4 * 8 + c] = (x3 - t0) >> 10;
    }
    for (int r = 0; r < 8; ++r) {
        const int32_t *d = v + r * 8;
        int32_t p2 = d[2], p3 = d[6];
        int32_t p1 = (p2 + p3) * 2217;
        int32_t t2 = p1 + p3 * -7567;
        int32_t t3 = p1 + p2 * 3135;
        p2 = d[0]; p3 = d[4];
        int32_t t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        const int32_t bias = 65536 + (128 << 17);
        const int32_t x0 = t0 + t3 + bias, x3 = t0 - t3 + bias, x1 = t1 + t2 + bias, x2 = t1 - t2 + bias;
        t0 = d[7]; t1 = d[5]; t2 = d[3]; t3 = d[1];
        p3 = t0 + t2;
        int32_t p4 = t1 + t3;
        p1 = t0 + t3; p2 = t1 + t2;
        const int32_t p5 = (p3 + p4) * 4816;
        t0 *= 1223; t1 *= 8410; t2 *= 12586; t3 *= 6149;
        p1 = p5 + p1 * -3685; p2 = p5 + p2 * -10497; p3 = p3 * -8034; p4 = p4 * -1597;
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
        uint8_t *o = out + r * stride;
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
#undef F2F
}

static int decode_block(jd *j, jd_comp *c, int bx, int by)
{
    int32_t blk[64];
    memset(blk, 0, sizeof(blk));
    const int t = decode_sym(j, 0, c->td);
    if (t < 0 || t > 11) return -1;
    c->pred += receive_extend(j, t);
    int16_t *sink = j->sink ? j->sink + ((size_t)c->sink_base + (size_t)by * c->bw + bx) * 64 : NULL;
    if (sink) sink[0] = (int16_t)c->pred;
    blk[0] = c->pred * (int32_t)j->qt[c->tq][0];
    for (int k = 1; k < 64;) {
        const int rs = decode_sym(j, 1, c->ta);
        if (rs < 0) return -1;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r == 15) { k += 16; continue; }
            break; /* EOB */
        }
        k += r;
        if (k > 63) return -1;
        const int val = receive_extend(j, s);
        if (sink) sink[k] = (int16_t)val;
        blk[kZigzag[k]] = val * (int32_t)j->qt[c->tq][k];
        ++k;
    }
    idct_block(blk, c->plane + (size_t)by * 8 * c->pw + (size_t)bx * 8, c->pw);
    return 0;
}

/* chroma sample at full resolution: zune-jpeg's separable two-step interpolation (see the file header) */
static int up_h(const uint8_t *row, int n, int x)
{
    /* upsample_horizontal: out[0] = in[0], out[2n-1] = in[n-1], otherwise (3 * near + far + 2) >> 2 */
    const int i = x >> 1;
    if (n == 1) return row[0];
    if (x == 0) return row[0];
    if (x == 2 * n - 1) return row[n - 1];
    return (x & 1) ? (3 * row[i] + row[i + 1] + 2) >> 2 : (3 * row[i] + row[i - 1] + 2) >> 2;
}

int fo_jpeg_adobe_transform(const uint8_t *data, size_t n) /* -1 = no APP14 Adobe segment, else its transform byte */
{
    jd *j = (jd *)calloc(1, sizeof(jd));
    if (!j) return -1;
    j->d = data; j->n = n;
    const int rc = parse(j);
    const int t = (rc == 0 || rc == -2) ? j->adobe_transform : -1;
    free(j);
    return t;
}

int fo_jpeg_info(const uint8_t *data, size_t n, uint32_t *w, uint32_t *h, uint32_t *c, uint32_t *orientation)
{
    jd *j = (jd *)calloc(1, sizeof(jd));
    if (!j) return -1;
    j->d = data; j->n = n;
    const int rc = parse(j);
    if (rc == 0 || rc == -2) {
        if (w) *w = (uint32_t)j->W;
        if (h) *h = (uint32_t)j->H;
        if (c) *c = (uint32_t)j->nc;
        if (orientation) *orientation = (uint32_t)j->orientation;
    }
    free(j);
    return rc;
}

/* out: W*H*1 (grayscale files), W*H*3 (YCbCr / RGB files) or W*H*4 (raw CMYK / YCCK samples).  0 = ok, -1 malformed, -2 unsupported */
static int decode_impl(const uint8_t *data, size_t n, uint8_t *out, int16_t *sink, size_t sink_blocks, uint32_t *nblocks_out)
{
    jd *j = (jd *)calloc(1, sizeof(jd));
    if (!j) return -1;
    j->d = data; j->n = n;
    int rc = parse(j);
    if (rc) { free(j); return rc; }
    if (j->nc != 1 && j->nc != 3 && j->nc != 4) { free(j); return -2; }
    if (j->nc == 1) { j->c[0].h = j->c[0].v = 1; j->hmax = j->vmax = 1; } /* a single component is never interleaved: its MCU is one block */
    /* chroma planes must be full size or exactly half in a direction (what the upsampler of zune-jpeg covers) */
    for (int i = 0; i < j->nc; ++i) {
        if (j->hmax % j->c[i].h || j->vmax % j->c[i].v || j->hmax / j->c[i].h > 2 || j->vmax / j->c[i].v > 2) { free(j); return -2; }
        if (!j->have_qt[j->c[i].tq] || !j->have_ht[0][j->c[i].td] || !j->have_ht[1][j->c[i].ta]) { free(j); return -1; }
    }
    const int mcux = (j->W + 8 * j->hmax - 1) / (8 * j->hmax), mcuy = (j->H + 8 * j->vmax - 1) / (8 * j->vmax);
    for (int i = 0; i < j->nc; ++i) {
        jd_comp *c = &j->c[i];
        c->bw = mcux * c->h; c->bh = mcuy * c->v;
        c->pw = c->bw * 8; c->ph = c->bh * 8;
        c->w = (j->W * c->h + j->hmax - 1) / j->hmax;
        c->h_px = (j->H * c->v + j->vmax - 1) / j->vmax;
        c->plane = (uint8_t *)calloc((size_t)c->pw * c->ph, 1);
        if (!c->plane) { rc = -1; goto done; }
    }
    {
        int total = 0;
        for (int i = 0; i < j->nc; ++i) { j->c[i].sink_base = total; total += j->c[i].bw * j->c[i].bh; }
        if (nblocks_out) *nblocks_out = (uint32_t)total;
        if (sink) {
            if ((size_t)total > sink_blocks) { rc = -1; goto done; }
            memset(sink, 0, (size_t)total * 128);
            j->sink = sink;
        }
    }
    {
        int rst_left = j->restart;
        for (int my = 0; my < mcuy && !rc; ++my)
            for (int mx = 0; mx < mcux && !rc; ++mx) {
                if (j->restart && rst_left == 0) {
                    /* byte align, expect RSTn, reset predictors (F.2.2.4) */
                    j->bitcnt = 0;
                    if (!j->marker) { /* the marker has not been met yet: it must be next */
                        if (j->pos + 2 <= j->n && j->d[j->pos] == 0xFF && j->d[j->pos + 1] >= 0xD0 && j->d[j->pos + 1] <= 0xD7) j->pos += 2;
                        else { rc = -1; break; }
                    } else if (j->marker >= 0xD0 && j->marker <= 0xD7) j->marker = 0;
                    else { rc = -1; break; }
                    for (int i = 0; i < j->nc; ++i) j->c[i].pred = 0;
                    rst_left = j->restart;
                }
                for (int i = 0; i < j->nc && !rc; ++i) {
                    jd_comp *c = &j->c[i];
                    for (int v = 0; v < c->v && !rc; ++v)
                        for (int h = 0; h < c->h && !rc; ++h) rc = decode_block(j, c, mx * c->h + h, my * c->v + v);
                }
                if (j->restart) rst_left--;
            }
    }
    if (rc || !out) goto done;
    {
        /* Adobe transform 0 with three components = the samples ARE R, G, B; four components (CMYK / YCCK) come out raw, as
         * JpegDecoder with out_colorspace = the input colour space returns them (reference src/handler.rs:417-419) */
        const int is_rgb = j->nc == 3 && j->adobe_transform == 0;
        for (int y = 0; y < j->H; ++y)
            for (int x = 0; x < j->W; ++x) {
                int s[4] = {0, 0, 0, 0};
                for (int i = 0; i < j->nc; ++i) {
                    const jd_comp *c = &j->c[i];
                    const int sh = j->hmax / c->h, sv = j->vmax / c->v; /* 1 or 2 */
                    const int cw = c->w, chh = c->h_px;
                    if (sv == 1 && sh == 1) s[i] = c->plane[(size_t)y * c->pw + x];
                    else if (sv == 1) s[i] = up_h(c->plane + (size_t)y * c->pw, cw, x);
                    else {
                        /* upsample_vertical: output row 2r is nearest to r with r - 1 as the far row, row 2r + 1 has r + 1;
                         * rows outside the plane repeat the edge row */
                        const int r = y >> 1;
                        int far = (y & 1) ? r + 1 : r - 1;
                        if (far < 0) far = 0;
                        if (far > chh - 1) far = chh - 1;
                        const uint8_t *near_row = c->plane + (size_t)r * c->pw, *far_row = c->plane + (size_t)far * c->pw;
                        if (sh == 1) s[i] = (3 * near_row[x] + far_row[x] + 2) >> 2;
                        else {
                            /* h2v2: the vertical step produces a whole (rounded) row, the horizontal step runs on it */
                            const int n2 = cw, ix = x >> 1;
                            int a = (3 * near_row[ix] + far_row[ix] + 2) >> 2;
                            if (n2 == 1 || x == 0 || x == 2 * n2 - 1) s[i] = a;
                            else {
                                const int k = (x & 1) ? ix + 1 : ix - 1;
                                const int b = (3 * near_row[k] + far_row[k] + 2) >> 2;
                                s[i] = (3 * a + b + 2) >> 2;
                            }
                        }
                    }
                }
                uint8_t *o = out + ((size_t)y * j->W + x) * j->nc;
                if (j->nc == 1) o[0] = (uint8_t)s[0];
                else if (j->nc == 4) { o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; o[3] = (uint8_t)s[3]; }
                else if (is_rgb) { o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2]; }
                else {
                    const int16_t yy = (int16_t)s[0], cb = (int16_t)(s[1] - 128), cr = (int16_t)(s[2] - 128);
                    o[0] = clamp8(yy + ((45 * cr) >> 5));
                    o[1] = clamp8(yy - ((11 * cb + 23 * cr) >> 5));
                    o[2] = clamp8(yy + ((113 * cb) >> 6));
                }
            }
    }
done:
    for (int i = 0; i < 4; ++i) free(j->c[i].plane);
    free(j);
    return rc;
}

int fo_jpeg_decode(const uint8_t *data, size_t n, uint8_t *out) { return decode_impl(data, n, out, NULL, 0, NULL); }

/* Entropy decoding only: the quantised coefficients of every block ([block][64], zig-zag order; blocks of component 0 in
 * raster order of its padded plane, then component 1, ...).  *nblocks receives the block count even if sink is NULL. */
int fo_jpeg_coefficients_of(const uint8_t *data, size_t n, int16_t *sink, size_t sink_blocks, uint32_t *nblocks)
{
    uint8_t dummy = 0;
    return decode_impl(data, n, sink ? NULL : NULL, sink, sink_blocks, nblocks) + 0 * dummy;
}
