/*
 * fanlin_oracle_jpeg.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), JPEG encoder back half.
 *
 * Restates image 0.25.6 src/codecs/jpeg/encoder.rs + transform.rs as driven by the reference at
 * src/handler.rs:274-278 (JpegEncoder::new_with_quality(q).encode_image(&DynamicImage)): baseline, three
 * components, every sampling factor 1x1, Annex K tables scaled the libjpeg way, the integer forward DCT
 * (a port of IJG jfdctint.c: 13-bit constants, 2 extra bits after the row pass, output scaled by 8),
 * quantisation `((d / 8) as f32 / q as f32).round()`, the Annex K Huffman tables and the segment order
 * SOI, APP0 (JFIF 1.2, aspect 1:1), SOF0, DQT x2, DHT x4, SOS, entropy data, EOI.
 * PARITY UNPINNED for the same reason as fanlin_oracle.h; what IS checked independently (tests/test_jpeg.py):
 * the tables against the DQT / DHT segments libjpeg itself writes, the DCT against a float64 DCT-II, and the
 * streams by decoding them with libjpeg (Pillow).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fanlin_oracle.h"

/* ITU T.81 Annex K, tables K.1 / K.2, natural (row-major) order -- encoder.rs STD_LUMA_QTABLE / STD_CHROMA_QTABLE */
static const uint8_t STD_LUMA_Q[64] = {
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t STD_CHROMA_Q[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
/* zig-zag position -> natural index (encoder.rs UNZIGZAG) */
static const uint8_t UNZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
/* Annex K tables K.3 - K.6 */
static const uint8_t DC_LUMA_LEN[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_CHROMA_LEN[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALUES[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_LUMA_LEN[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_LUMA_VALUES[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t AC_CHROMA_LEN[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_CHROMA_VALUES[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* encoder.rs JpegEncoder::new_with_quality: the libjpeg scaling of the Annex K tables */
void fo_jpeg_qtables(int quality, uint8_t luma[64], uint8_t chroma[64])
{
    uint32_t scale = (uint32_t)(quality < 1 ? 1 : (quality > 100 ? 100 : quality));
    scale = scale < 50 ? 5000 / scale : 200 - scale * 2;
    for (int i = 0; i < 64; ++i) {
        uint32_t a = ((uint32_t)STD_LUMA_Q[i] * scale + 50) / 100, b = ((uint32_t)STD_CHROMA_Q[i] * scale + 50) / 100;
        luma[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
        chroma[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
}

/* transform.rs fdct == IJG jfdctint.c (release 8/9 form) */
enum { CONST_BITS = 13, PASS1_BITS = 2 };
enum {
    FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
    FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069, FIX_2_053119869 = 16819,
    FIX_2_562915447 = 20995, FIX_3_072711026 = 25172
};

static void fdct_1d(const int32_t in[8], int32_t out[8], int pass)
{
    int32_t t0 = in[0] + in[7], t1 = in[1] + in[6], t2 = in[2] + in[5], t3 = in[3] + in[4];
    int32_t t10 = t0 + t3, t12 = t0 - t3, t11 = t1 + t2, t13 = t1 - t2;
    t0 = in[0] - in[7]; t1 = in[1] - in[6]; t2 = in[2] - in[5]; t3 = in[3] - in[4];
    const int sh = pass == 1 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
    if (pass == 1) {
        out[0] = (t10 + t11 - 8 * 128) << PASS1_BITS; /* unsigned -> signed conversion folded in */
        out[4] = (t10 - t11) << PASS1_BITS;
    } else {
        t10 += 1 << (PASS1_BITS - 1); /* fudge factor for the final descale */
        out[0] = (t10 + t11) >> PASS1_BITS;
        out[4] = (t10 - t11) >> PASS1_BITS;
    }
    int32_t z1 = (t12 + t13) * FIX_0_541196100 + (1 << (sh - 1));
    out[2] = (z1 + t12 * FIX_0_765366865) >> sh;
    out[6] = (z1 - t13 * FIX_1_847759065) >> sh;
    t12 = t0 + t2;
    t13 = t1 + t3;
    z1 = (t12 + t13) * FIX_1_175875602 + (1 << (sh - 1));
    t12 = t12 * (-FIX_0_390180644) + z1;
    t13 = t13 * (-FIX_1_961570560) + z1;
    z1 = (t0 + t3) * (-FIX_0_899976223);
    t0 = t0 * FIX_1_501321110 + z1 + t12;
    t3 = t3 * FIX_0_298631336 + z1 + t13;
    z1 = (t1 + t2) * (-FIX_2_562915447);
    t1 = t1 * FIX_3_072711026 + z1 + t13;
    t2 = t2 * FIX_2_053119869 + z1 + t12;
    out[1] = t0 >> sh; out[3] = t1 >> sh; out[5] = t2 >> sh; out[7] = t3 >> sh;
}

void fo_jpeg_fdct(const uint8_t samples[64], int32_t coeffs[64])
{
    int32_t a[8], b[8];
    for (int y = 0; y < 8; ++y) {
        for (int k = 0; k < 8; ++k) a[k] = samples[y * 8 + k];
        fdct_1d(a, b, 1);
        for (int k = 0; k < 8; ++k) coeffs[y * 8 + k] = b[k];
    }
    for (int x = 7; x >= 0; --x) {
        for (int k = 0; k < 8; ++k) a[k] = coeffs[k * 8 + x];
        fdct_1d(a, b, 2);
        for (int k = 0; k < 8; ++k) coeffs[k * 8 + x] = b[k];
    }
}

/* encoder.rs encode_rgb, "Quantization": ((d / 8) as f32 / f32::from(q)).round() as i32, natural order */
static int32_t quantise(int32_t d, uint8_t q) { return (int32_t)roundf((float)(d / 8) / (float)q); }

/* Quantised coefficients of every block, in ZIG-ZAG order, unit = (block_row * blocks_x + block_col) * 3 + component. */
int fo_jpeg_coefficients(const fo_image *src, int quality, int16_t *out)
{
    uint32_t pw, ph;
    uint8_t *planes = (uint8_t *)malloc((size_t)((src->w + 7) & ~7u) * ((src->h + 7) & ~7u) * 3);
    if (!planes || fo_jpeg_ycbcr444(src, planes, &pw, &ph)) { free(planes); return -1; }
    uint8_t q[2][64];
    fo_jpeg_qtables(quality, q[0], q[1]);
    const uint32_t bx = pw / 8, by = ph / 8;
    for (uint32_t r = 0; r < by; ++r)
        for (uint32_t c = 0; c < bx; ++c)
            for (int comp = 0; comp < 3; ++comp) {
                uint8_t s[64];
                int32_t d[64];
                const uint8_t *pl = planes + (size_t)comp * pw * ph;
                for (int y = 0; y < 8; ++y) memcpy(s + y * 8, pl + (size_t)(r * 8 + y) * pw + c * 8, 8);
                fo_jpeg_fdct(s, d);
                int16_t *o = out + ((size_t)(r * bx + c) * 3 + comp) * 64;
                for (int k = 0; k < 64; ++k) o[k] = (int16_t)quantise(d[UNZIGZAG[k]], q[comp ? 1 : 0][UNZIGZAG[k]]);
            }
    free(planes);
    return 0;
}

/* ------------------------------------------------------------ entropy coding -- */
typedef struct { uint8_t len[256]; uint16_t code[256]; } huff_table;

/* encoder.rs build_huff_lut: canonical codes from (code-length counts, values) -- T.81 Annex C */
static void build_huff(const uint8_t lens[16], const uint8_t *values, huff_table *t)
{
    memset(t, 0, sizeof(*t));
    uint32_t code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < lens[l - 1]; ++i, ++k) { t->len[values[k]] = (uint8_t)l; t->code[values[k]] = (uint16_t)code++; }
        code <<= 1;
    }
}

typedef struct { uint8_t *p; size_t n, cap; uint32_t acc; uint8_t nbits; } bitw;

static void put_byte(bitw *w, uint8_t b) { if (w->n < w->cap) w->p[w->n] = b; w->n++; }
/* encoder.rs BitWriter::write_bits: MSB first, 0xFF is followed by a stuffed 0x00 */
static void write_bits(bitw *w, uint16_t bits, uint8_t size)
{
    if (size == 0) return;
    w->nbits = (uint8_t)(w->nbits + size);
    w->acc |= (uint32_t)bits << (32 - w->nbits);
    while (w->nbits >= 8) {
        const uint8_t b = (uint8_t)(w->acc >> 24);
        put_byte(w, b);
        if (b == 0xFF) put_byte(w, 0x00);
        w->nbits = (uint8_t)(w->nbits - 8);
        w->acc <<= 8;
    }
}
static void encode_coefficient(int32_t c, uint8_t *size, uint16_t *value)
{
    uint16_t mag = (uint16_t)(c < 0 ? -c : c);
    uint8_t n = 0;
    while (mag > 0) { mag >>= 1; n++; }
    const uint16_t mask = (uint16_t)((1u << n) - 1u);
    *size = n;
    *value = (uint16_t)((c < 0 ? (uint16_t)(c - 1) : (uint16_t)c) & mask);
}
static int32_t write_block(bitw *w, const int16_t zz[64], int32_t prevdc, const huff_table *dc, const huff_table *ac)
{
    uint8_t size; uint16_t value;
    encode_coefficient(zz[0] - prevdc, &size, &value);
    write_bits(w, dc->code[size], dc->len[size]);
    write_bits(w, value, size);
    uint32_t run = 0;
    for (int k = 1; k < 64; ++k) {
        if (zz[k] == 0) { run++; continue; }
        while (run > 15) { write_bits(w, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
        encode_coefficient(zz[k], &size, &value);
        const uint8_t sym = (uint8_t)((run << 4) | size);
        write_bits(w, ac->code[sym], ac->len[sym]);
        write_bits(w, value, size);
        run = 0;
    }
    if (zz[63] == 0) write_bits(w, ac->code[0x00], ac->len[0x00]);
    return zz[0];
}

static void put_segment(bitw *w, uint8_t marker, const uint8_t *data, size_t n)
{
    put_byte(w, 0xFF); put_byte(w, marker);
    put_byte(w, (uint8_t)((n + 2) >> 8)); put_byte(w, (uint8_t)(n + 2));
    for (size_t i = 0; i < n; ++i) put_byte(w, data[i]);
}

/* Everything in front of the entropy-coded data (623 bytes).  Returns the length; writes at most cap bytes. */
size_t fo_jpeg_header(uint32_t width, uint32_t height, int quality, uint8_t *out, size_t cap)
{
    bitw w = {out, 0, cap, 0, 0};
    uint8_t buf[256], q[2][64];
    fo_jpeg_qtables(quality, q[0], q[1]);
    put_byte(&w, 0xFF); put_byte(&w, 0xD8);                                               /* SOI */
    const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 2, 0, 0, 1, 0, 1, 0, 0};          /* build_jfif_header, PixelDensity::default() */
    put_segment(&w, 0xE0, jfif, sizeof(jfif));
    size_t n = 0;                                                                        /* build_frame_header */
    buf[n++] = 8; buf[n++] = (uint8_t)(height >> 8); buf[n++] = (uint8_t)height; buf[n++] = (uint8_t)(width >> 8); buf[n++] = (uint8_t)width;
    buf[n++] = 3;
    for (int c = 0; c < 3; ++c) { buf[n++] = (uint8_t)(c + 1); buf[n++] = 0x11; buf[n++] = (uint8_t)(c ? 1 : 0); }
    put_segment(&w, 0xC0, buf, n);
    for (int t = 0; t < 2; ++t) {                                                         /* build_quantization_segment */
        buf[0] = (uint8_t)t;
        for (int k = 0; k < 64; ++k) buf[1 + k] = q[t][UNZIGZAG[k]];
        put_segment(&w, 0xDB, buf, 65);
    }
    const struct { uint8_t cls, dest; const uint8_t *len, *val; size_t nval; } hts[4] = {
        {0, 0, DC_LUMA_LEN, DC_VALUES, 12}, {1, 0, AC_LUMA_LEN, AC_LUMA_VALUES, 162},
        {0, 1, DC_CHROMA_LEN, DC_VALUES, 12}, {1, 1, AC_CHROMA_LEN, AC_CHROMA_VALUES, 162}};
    for (int t = 0; t < 4; ++t) {                                                         /* build_huffman_segment */
        buf[0] = (uint8_t)((hts[t].cls << 4) | hts[t].dest);
        memcpy(buf + 1, hts[t].len, 16);
        memcpy(buf + 17, hts[t].val, hts[t].nval);
        put_segment(&w, 0xC4, buf, 17 + hts[t].nval);
    }
    n = 0;                                                                               /* build_scan_header */
    buf[n++] = 3;
    for (int c = 0; c < 3; ++c) { buf[n++] = (uint8_t)(c + 1); buf[n++] = (uint8_t)(c ? 0x11 : 0x00); }
    buf[n++] = 0; buf[n++] = 63; buf[n++] = 0;
    put_segment(&w, 0xDA, buf, n);
    return w.n;
}

/* JpegEncoder::new_with_quality(q).encode_image(&DynamicImage) -- reference src/handler.rs:274-278.
 * Returns the stream length (which may exceed cap; then only cap bytes were written), or 0 on error. */
size_t fo_jpeg_encode(const fo_image *src, int quality, uint8_t *out, size_t cap)
{
    if (src->w == 0 || src->h == 0 || src->w > 65535 || src->h > 65535) return 0;
    const uint32_t bx = (src->w + 7) / 8, by = (src->h + 7) / 8;
    int16_t *coef = (int16_t *)malloc((size_t)bx * by * 3 * 64 * sizeof(int16_t));
    if (!coef || fo_jpeg_coefficients(src, quality, coef)) { free(coef); return 0; }
    bitw w = {out, 0, cap, 0, 0};
    w.n = fo_jpeg_header(src->w, src->h, quality, out, cap);
    huff_table dcl, acl, dcc, acc;
    build_huff(DC_LUMA_LEN, DC_VALUES, &dcl); build_huff(AC_LUMA_LEN, AC_LUMA_VALUES, &acl);
    build_huff(DC_CHROMA_LEN, DC_VALUES, &dcc); build_huff(AC_CHROMA_LEN, AC_CHROMA_VALUES, &acc);
    int32_t prev[3] = {0, 0, 0};
    for (size_t b = 0; b < (size_t)bx * by; ++b)
        for (int c = 0; c < 3; ++c) prev[c] = write_block(&w, coef + (b * 3 + c) * 64, prev[c], c ? &dcc : &dcl, c ? &acc : &acl);
    write_bits(&w, 0x7F, 7);            /* pad_byte */
    put_byte(&w, 0xFF); put_byte(&w, 0xD9); /* EOI */
    free(coef);
    return w.n;
}
