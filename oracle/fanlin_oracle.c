/*
 * fanlin_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See fanlin_oracle.h for the scope statement and the "PARITY UNPINNED" note.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math [-mfma] -shared -fPIC
 * (-ffp-contract=off keeps `t + v*w` as two roundings, which is what rustc
 * emits for the reference; FO_ARITH_FMA goes through explicit fmaf()).
 *
 * Loop structure deliberately mirrors image 0.25.6 (scalar, 4 padded channels,
 * kernel called through a function pointer once per weight, f32 intermediate
 * image between the vertical and the horizontal pass) so that timing this
 * code is a fair stand-in for timing the reference's CPU path.
 */
#include "fanlin_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void fo_free(void *p) { free(p); }

/* ---------------------------------------------------------------- utils -- */

/* image 0.25.6 src/math/utils.rs::resize_dimensions (f64; round = half away
 * from zero; the u32-overflow branches are kept for completeness). */
void fo_resize_dimensions(uint32_t width, uint32_t height, uint32_t nwidth, uint32_t nheight,
                          int fill, uint32_t *ow, uint32_t *oh)
{
    double wratio = (double)nwidth / (double)width;
    double hratio = (double)nheight / (double)height;
    double ratio = fill ? fmax(wratio, hratio) : fmin(wratio, hratio);
    double fw = round((double)width * ratio);
    double fh = round((double)height * ratio);
    uint64_t nw = (uint64_t)fw; if (nw < 1) nw = 1;
    uint64_t nh = (uint64_t)fh; if (nh < 1) nh = 1;
    if (nw > (uint64_t)UINT32_MAX) {
        double r = (double)UINT32_MAX / (double)width;
        uint32_t h2 = (uint32_t)round((double)height * r);
        *ow = UINT32_MAX; *oh = h2 < 1 ? 1 : h2;
    } else if (nh > (uint64_t)UINT32_MAX) {
        double r = (double)UINT32_MAX / (double)height;
        uint32_t w2 = (uint32_t)round((double)width * r);
        *ow = w2 < 1 ? 1 : w2; *oh = UINT32_MAX;
    } else {
        *ow = (uint32_t)nw; *oh = (uint32_t)nh;
    }
}

/* ------------------------------------------------------------- kernels -- */

typedef struct fo_filter {
    float (*kernel)(float x, float param); /* Box<dyn Fn(f32)->f32> in the reference */
    float support;
    float param;
} fo_filter;

/* sample.rs::sinc */
static float k_sinc(float t)
{
    float a = t * 3.14159265358979323846f; /* f32::consts::PI */
    if (t == 0.0f) return 1.0f;
    return sinf(a) / a;
}
/* sample.rs::lanczos(x, 3.0) via lanczos3_kernel */
static float k_lanczos3(float x, float unused)
{
    (void)unused;
    if (fabsf(x) < 3.0f) return k_sinc(x) * k_sinc(x / 3.0f);
    return 0.0f;
}
/* sample.rs::gaussian(x, r):
 * ((2.0 * PI).sqrt() * r).recip() * (-x.powi(2) / (2.0 * r.powi(2))).exp() */
static float k_gaussian(float x, float r)
{
    float norm = 1.0f / (sqrtf(2.0f * 3.14159265358979323846f) * r);
    float e = expf(-(x * x) / (2.0f * (r * r)));
    return norm * e;
}
/* sample.rs::box_kernel / triangle_kernel (Nearest is used by the GIF path). */
static float k_box(float x, float unused) { (void)x; (void)unused; return 1.0f; }
static float k_triangle(float x, float unused)
{
    (void)unused;
    if (fabsf(x) < 1.0f) return 1.0f - fabsf(x);
    return 0.0f;
}

static int make_filter(int filter, float sigma, fo_filter *f)
{
    switch (filter) {
    case FO_FILTER_LANCZOS3: f->kernel = k_lanczos3; f->support = 3.0f; f->param = 0.0f; return 0;
    case FO_FILTER_GAUSSIAN: f->kernel = k_gaussian; f->support = 2.0f * sigma; f->param = sigma; return 0;
    case FO_FILTER_NEAREST:  f->kernel = k_box;      f->support = 0.0f; f->param = 0.0f; return 0;
    case FO_FILTER_TRIANGLE: f->kernel = k_triangle; f->support = 1.0f; f->param = 0.0f; return 0;
    }
    return -1;
}

static int64_t clamp_i64(int64_t a, int64_t lo, int64_t hi) { return a < lo ? lo : (a > hi ? hi : a); }

/* One output sample's window and normalised weights -- the body shared by
 * vertical_sample and horizontal_sample in sample.rs.  ws must hold in_size floats. */
static void window_weights(const fo_filter *f, uint32_t in_size, float ratio, float sratio,
                           float src_support, uint32_t out, uint32_t *left_o, uint32_t *right_o, float *ws)
{
    float input = ((float)out + 0.5f) * ratio;
    int64_t left = (int64_t)floorf(input - src_support);
    left = clamp_i64(left, 0, (int64_t)in_size - 1);
    int64_t right = (int64_t)ceilf(input + src_support);
    right = clamp_i64(right, left + 1, (int64_t)in_size);
    input = input - 0.5f;

    float sum = 0.0f;
    uint32_t n = 0;
    for (int64_t i = left; i < right; ++i) {
        float w = f->kernel(((float)i - input) / sratio, f->param);
        ws[n++] = w;
        sum += w;
    }
    for (uint32_t k = 0; k < n; ++k) ws[k] /= sum;
    *left_o = (uint32_t)left;
    *right_o = (uint32_t)right;
}

long fo_build_weights(uint32_t in_size, uint32_t out_size, int filter, float sigma,
                      uint32_t *left, uint32_t *count, uint32_t *offset, float *weights, size_t cap)
{
    fo_filter f;
    if (make_filter(filter, sigma, &f) || in_size == 0 || out_size == 0) return -1;
    float ratio = (float)in_size / (float)out_size;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = f.support * sratio;
    float *ws = (float *)malloc(sizeof(float) * (size_t)in_size);
    if (!ws) return -1;
    size_t total = 0;
    for (uint32_t o = 0; o < out_size; ++o) {
        uint32_t l, r;
        window_weights(&f, in_size, ratio, sratio, src_support, o, &l, &r, ws);
        uint32_t n = r - l;
        if (total + n > cap) { free(ws); return -1; }
        left[o] = l; count[o] = n; offset[o] = (uint32_t)total;
        memcpy(weights + total, ws, sizeof(float) * n);
        total += n;
    }
    offset[out_size] = (uint32_t)total;
    free(ws);
    return (long)total;
}

/* ------------------------------------------------------- pointwise ops -- */

/* color.rs: SRGB_LUMA = [2126, 7152, 722], SRGB_LUMA_DIV = 10000, u32 maths,
 * truncating division.  Rgb8 -> Luma8, Rgba8 -> LumaA8 (alpha kept),
 * Luma8 / LumaA8 unchanged (DynamicImage::grayscale). */
int fo_grayscale(const fo_image *src, fo_image *dst)
{
    uint32_t dc = (src->c == 3) ? 1 : (src->c == 4) ? 2 : src->c;
    size_t n = (size_t)src->w * src->h;
    uint8_t *out = (uint8_t *)malloc((n * dc) > 0 ? n * dc : 1);
    if (!out) return -1;
    if (src->c <= 2) {
        memcpy(out, src->px, n * dc);
    } else {
        for (size_t i = 0; i < n; ++i) {
            const uint8_t *p = src->px + i * src->c;
            uint32_t l = 2126u * p[0] + 7152u * p[1] + 722u * p[2];
            out[i * dc] = (uint8_t)(l / 10000u);
            if (src->c == 4) out[i * dc + 1] = p[3];
        }
    }
    dst->w = src->w; dst->h = src->h; dst->c = dc; dst->px = out;
    return 0;
}

/* color.rs Invert: colour channels max - c; alpha (LumaA, Rgba) untouched. */
void fo_invert(fo_image *img)
{
    size_t n = (size_t)img->w * img->h;
    uint32_t c = img->c, nc = (c == 2 || c == 4) ? c - 1 : c;
    for (size_t i = 0; i < n; ++i)
        for (uint32_t k = 0; k < nc; ++k) img->px[i * c + k] = (uint8_t)(255 - img->px[i * c + k]);
}

/* image 0.25.6 metadata::Orientation::from_exif + DynamicImage::apply_orientation, written as the
 * imageops loops it calls: rotate90 puts (x,y) at (h-1-y, x), rotate270 at (y, w-1-x), rotate180 at
 * (w-1-x, h-1-y); the two "FlipH" variants rotate first and mirror afterwards. */
static void put_px(fo_image *d, uint32_t x, uint32_t y, const uint8_t *p) { memcpy(d->px + ((size_t)y * d->w + x) * d->c, p, d->c); }
static int rotate_into(const fo_image *s, int quarter_turns, fo_image *d)
{
    d->c = s->c;
    d->w = (quarter_turns & 1) ? s->h : s->w;
    d->h = (quarter_turns & 1) ? s->w : s->h;
    d->px = (uint8_t *)malloc((size_t)d->w * d->h * d->c + 1);
    if (!d->px) return -1;
    for (uint32_t y = 0; y < s->h; ++y)
        for (uint32_t x = 0; x < s->w; ++x) {
            const uint8_t *p = s->px + ((size_t)y * s->w + x) * s->c;
            if (quarter_turns == 1) put_px(d, s->h - y - 1, x, p);
            else if (quarter_turns == 2) put_px(d, s->w - x - 1, s->h - y - 1, p);
            else if (quarter_turns == 3) put_px(d, y, s->w - x - 1, p);
            else put_px(d, x, y, p);
        }
    return 0;
}
static void flip_h_in_place(fo_image *m)
{
    uint8_t t[4];
    for (uint32_t y = 0; y < m->h; ++y)
        for (uint32_t x = 0; x < m->w / 2; ++x) {
            uint8_t *a = m->px + ((size_t)y * m->w + x) * m->c, *b = m->px + ((size_t)y * m->w + (m->w - 1 - x)) * m->c;
            memcpy(t, a, m->c); memcpy(a, b, m->c); memcpy(b, t, m->c);
        }
}
static void flip_v_in_place(fo_image *m)
{
    uint8_t t[4];
    for (uint32_t y = 0; y < m->h / 2; ++y)
        for (uint32_t x = 0; x < m->w; ++x) {
            uint8_t *a = m->px + ((size_t)y * m->w + x) * m->c, *b = m->px + ((size_t)(m->h - 1 - y) * m->w + x) * m->c;
            memcpy(t, a, m->c); memcpy(a, b, m->c); memcpy(b, t, m->c);
        }
}
int fo_apply_orientation(const fo_image *src, int exif, fo_image *dst)
{
    switch (exif) {
    case 2: if (rotate_into(src, 0, dst)) return -1; flip_h_in_place(dst); return 0;   /* FlipHorizontal */
    case 3: return rotate_into(src, 2, dst);                                           /* Rotate180 */
    case 4: if (rotate_into(src, 0, dst)) return -1; flip_v_in_place(dst); return 0;   /* FlipVertical */
    case 5: if (rotate_into(src, 1, dst)) return -1; flip_h_in_place(dst); return 0;   /* Rotate90FlipH */
    case 6: return rotate_into(src, 1, dst);                                           /* Rotate90 */
    case 7: if (rotate_into(src, 3, dst)) return -1; flip_h_in_place(dst); return 0;   /* Rotate270FlipH */
    case 8: return rotate_into(src, 3, dst);                                           /* Rotate270 */
    default: return rotate_into(src, 0, dst);                                          /* NoTransforms */
    }
}

/* ------------------------------------------------------------ resample -- */

static inline float acc_step(float t, float v, float w, int arith)
{
    if (arith == FO_ARITH_FMA) return fmaf(v, w, t);
    float p = v * w; /* -ffp-contract=off: two roundings, as in the reference */
    return t + p;
}

/* sample.rs::vertical_sample: u8 (c channels, padded to 4 with 255 by
 * Pixel::channels4) -> Rgba32F, no rounding, no clamping. */
static float *vertical_sample_u8(const fo_image *img, uint32_t new_h, const fo_filter *f, int arith)
{
    uint32_t width = img->w, height = img->h, c = img->c;
    float *out = (float *)malloc(sizeof(float) * 4 * (size_t)width * new_h + 16);
    float *ws = (float *)malloc(sizeof(float) * (size_t)height);
    if (!out || !ws) { free(out); free(ws); return NULL; }
    float ratio = (float)height / (float)new_h;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = f->support * sratio;
    for (uint32_t outy = 0; outy < new_h; ++outy) {
        uint32_t left, right;
        window_weights(f, height, ratio, sratio, src_support, outy, &left, &right, ws);
        uint32_t n = right - left;
        for (uint32_t x = 0; x < width; ++x) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
            for (uint32_t i = 0; i < n; ++i) {
                const uint8_t *p = img->px + ((size_t)(left + i) * width + x) * c;
                float v0 = (float)p[0];
                float v1 = c > 1 ? (float)p[1] : 255.0f;
                float v2 = c > 2 ? (float)p[2] : 255.0f;
                float v3 = c > 3 ? (float)p[3] : 255.0f;
                float w = ws[i];
                t0 = acc_step(t0, v0, w, arith);
                t1 = acc_step(t1, v1, w, arith);
                t2 = acc_step(t2, v2, w, arith);
                t3 = acc_step(t3, v3, w, arith);
            }
            float *o = out + ((size_t)outy * width + x) * 4;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
        }
    }
    free(ws);
    return out;
}

/* FloatNearest + NumCast: clamp(t, 0, 255) then f32::round (half away from zero). */
static inline uint8_t to_u8_nearest(float t)
{
    if (t < 0.0f) t = 0.0f; else if (t > 255.0f) t = 255.0f;
    return (uint8_t)roundf(t);
}

/* sample.rs::horizontal_sample: Rgba32F -> u8, first c channels kept. */
static uint8_t *horizontal_sample_f32(const float *img, uint32_t width, uint32_t height, uint32_t c,
                                      uint32_t new_w, const fo_filter *f, int arith, int grouped)
{
    uint8_t *out = (uint8_t *)malloc((size_t)new_w * height * c + 16);
    float *ws = (float *)malloc(sizeof(float) * (size_t)width);
    if (!out || !ws) { free(out); free(ws); return NULL; }
    float ratio = (float)width / (float)new_w;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = f->support * sratio;
    for (uint32_t outx = 0; outx < new_w; ++outx) {
        uint32_t left, right;
        window_weights(f, width, ratio, sratio, src_support, outx, &left, &right, ws);
        uint32_t n = right - left;
        for (uint32_t y = 0; y < height; ++y) {
            float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (arith == FO_ARITH_FMA && grouped) {
                /* The HIP kernels' horizontal order for the Lanczos3 resize: taps grouped by aligned blocks of 4 source
                 * pixels, one fused multiply-add per tap inside a block (from 0), block sums
                 * added in ascending order. */
                float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (uint32_t i = 0; i < n; ++i) {
                    if (i != 0 && ((left + i) & 3u) == 0u)
                        for (int k = 0; k < 4; ++k) { t[k] = t[k] + part[k]; part[k] = 0.0f; }
                    const float *p = img + ((size_t)y * width + left + i) * 4;
                    for (int k = 0; k < 4; ++k) part[k] = fmaf(p[k], ws[i], part[k]);
                }
                for (int k = 0; k < 4; ++k) t[k] = t[k] + part[k];
            } else {
                for (uint32_t i = 0; i < n; ++i) {
                    const float *p = img + ((size_t)y * width + left + i) * 4;
                    float w = ws[i];
                    t[0] = acc_step(t[0], p[0], w, arith);
                    t[1] = acc_step(t[1], p[1], w, arith);
                    t[2] = acc_step(t[2], p[2], w, arith);
                    t[3] = acc_step(t[3], p[3], w, arith);
                }
            }
            uint8_t *o = out + ((size_t)y * new_w + outx) * c;
            for (uint32_t k = 0; k < c; ++k) o[k] = to_u8_nearest(t[k]);
        }
    }
    free(ws);
    return out;
}

static int clone_image(const fo_image *src, fo_image *dst)
{
    size_t n = (size_t)src->w * src->h * src->c;
    uint8_t *p = (uint8_t *)malloc(n ? n : 1);
    if (!p) return -1;
    memcpy(p, src->px, n);
    dst->w = src->w; dst->h = src->h; dst->c = src->c; dst->px = p;
    return 0;
}

/* grouped: FO_ARITH_FMA sums the horizontal taps by aligned blocks of 4 source pixels (resize kernels);
 * the blur kernels accumulate tap by tap. */
static int sample_two_pass(const fo_image *src, uint32_t nw, uint32_t nh, const fo_filter *f, int arith, int grouped, fo_image *dst)
{
    float *tmp = vertical_sample_u8(src, nh, f, arith);
    if (!tmp) return -1;
    uint8_t *out = horizontal_sample_f32(tmp, src->w, nh, src->c, nw, f, arith, grouped);
    free(tmp);
    if (!out) return -1;
    dst->w = nw; dst->h = nh; dst->c = src->c; dst->px = out;
    return 0;
}

/* imageops::resize: empty -> empty, same size -> copy, else vertical then horizontal. */
int fo_resize_exact(const fo_image *src, uint32_t nw, uint32_t nh, int filter, int arith, fo_image *dst)
{
    fo_filter f;
    if (make_filter(filter, 0.0f, &f)) return -1;
    if (src->w == 0 || src->h == 0) {
        dst->w = nw; dst->h = nh; dst->c = src->c;
        dst->px = (uint8_t *)calloc((size_t)nw * nh * src->c + 1, 1);
        return dst->px ? 0 : -1;
    }
    if (nw == src->w && nh == src->h) return clone_image(src, dst);
    return sample_two_pass(src, nw, nh, &f, arith, 1, dst);
}

/* DynamicImage::resize */
int fo_resize(const fo_image *src, uint32_t nw, uint32_t nh, int filter, int arith, fo_image *dst)
{
    if (nw == src->w && nh == src->h) return clone_image(src, dst);
    uint32_t w2, h2;
    fo_resize_dimensions(src->w, src->h, nw, nh, 0, &w2, &h2);
    return fo_resize_exact(src, w2, h2, filter, arith, dst);
}

/* imageops::crop_imm + to_image, with crop_dimms clamping */
static int crop_image(const fo_image *src, uint32_t x, uint32_t y, uint32_t w, uint32_t h, fo_image *dst)
{
    if (x > src->w) x = src->w;
    if (y > src->h) y = src->h;
    if (h > src->h - y) h = src->h - y;
    if (w > src->w - x) w = src->w - x;
    uint8_t *p = (uint8_t *)malloc((size_t)w * h * src->c + 1);
    if (!p) return -1;
    for (uint32_t r = 0; r < h; ++r)
        memcpy(p + (size_t)r * w * src->c, src->px + ((size_t)(y + r) * src->w + x) * src->c, (size_t)w * src->c);
    dst->w = w; dst->h = h; dst->c = src->c; dst->px = p;
    return 0;
}

/* DynamicImage::resize_to_fill */
int fo_resize_to_fill(const fo_image *src, uint32_t nw, uint32_t nh, int filter, int arith, fo_image *dst)
{
    uint32_t w2, h2;
    fo_resize_dimensions(src->w, src->h, nw, nh, 1, &w2, &h2);
    fo_image mid;
    if (fo_resize_exact(src, w2, h2, filter, arith, &mid)) return -1;
    uint64_t ratio = (uint64_t)mid.w * nh;
    uint64_t nratio = (uint64_t)nw * mid.h;
    int rc;
    if (nratio > ratio) rc = crop_image(&mid, 0, (mid.h - nh) / 2, nw, nh, dst);
    else                rc = crop_image(&mid, (mid.w - nw) / 2, 0, nw, nh, dst);
    free(mid.px);
    return rc;
}

/* imageops::blur (0.25.6): Filter{gaussian(x, sigma), support 2 sigma}, same-size two-pass */
int fo_blur(const fo_image *src, float sigma, int arith, fo_image *dst)
{
    fo_filter f;
    if (sigma <= 0.0f) sigma = 1.0f;
    make_filter(FO_FILTER_GAUSSIAN, sigma, &f);
    if (src->w == 0 || src->h == 0) return clone_image(src, dst);
    return sample_two_pass(src, src->w, src->h, &f, arith, 0, dst);
}

/* ----------------------------------------------------------- letterbox -- */

/* GenericImageView::get_pixel for DynamicImage: to_rgba().into_color() */
static inline void to_rgba8(const uint8_t *p, uint32_t c, uint8_t o[4])
{
    switch (c) {
    case 1: o[0] = o[1] = o[2] = p[0]; o[3] = 255; break;
    case 2: o[0] = o[1] = o[2] = p[0]; o[3] = p[1]; break;
    case 3: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 255; break;
    default: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; break;
    }
}

/* color.rs `impl Blend for Rgba<T>` (f32 src-over, truncating NumCast back to u8) */
static void blend_rgba8(uint8_t bg[4], const uint8_t fg[4])
{
    if (fg[3] == 0) return;
    if (fg[3] == 255) { memcpy(bg, fg, 4); return; }
    const float max_t = 255.0f;
    float bg_r = (float)bg[0] / max_t, bg_g = (float)bg[1] / max_t, bg_b = (float)bg[2] / max_t, bg_a = (float)bg[3] / max_t;
    float fg_r = (float)fg[0] / max_t, fg_g = (float)fg[1] / max_t, fg_b = (float)fg[2] / max_t, fg_a = (float)fg[3] / max_t;
    float alpha_final = bg_a + fg_a - bg_a * fg_a;
    if (alpha_final == 0.0f) return;
    float bg_r_a = bg_r * bg_a, bg_g_a = bg_g * bg_a, bg_b_a = bg_b * bg_a;
    float fg_r_a = fg_r * fg_a, fg_g_a = fg_g * fg_a, fg_b_a = fg_b * fg_a;
    float out_r_a = fg_r_a + bg_r_a * (1.0f - fg_a);
    float out_g_a = fg_g_a + bg_g_a * (1.0f - fg_a);
    float out_b_a = fg_b_a + bg_b_a * (1.0f - fg_a);
    float out_r = out_r_a / alpha_final, out_g = out_g_a / alpha_final, out_b = out_b_a / alpha_final;
    bg[0] = (uint8_t)(max_t * out_r);
    bg[1] = (uint8_t)(max_t * out_g);
    bg[2] = (uint8_t)(max_t * out_b);
    bg[3] = (uint8_t)(max_t * alpha_final);
}

int fo_letterbox(const fo_image *top, uint32_t w, uint32_t h, const uint8_t fill[3], fo_image *dst)
{
    uint8_t *bg = (uint8_t *)malloc((size_t)w * h * 4 + 1);
    if (!bg) return -1;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        bg[i * 4 + 0] = fill[0]; bg[i * 4 + 1] = fill[1]; bg[i * 4 + 2] = fill[2]; bg[i * 4 + 3] = 255;
    }
    /* handler.rs:241-246: offsets are abs_diff / 2, always >= 0 */
    uint32_t dx = (w > top->w ? w - top->w : top->w - w) / 2;
    uint32_t dy = (h > top->h ? h - top->h : top->h - h) / 2;
    /* imageops::overlay_bounds_ext with non-negative origin: clip to the bottom image */
    uint32_t rw = 0, rh = 0;
    if (dx < w) { rw = w - dx; if (rw > top->w) rw = top->w; }
    if (dy < h) { rh = h - dy; if (rh > top->h) rh = top->h; }
    for (uint32_t y = 0; y < rh; ++y)
        for (uint32_t x = 0; x < rw; ++x) {
            uint8_t fg[4];
            to_rgba8(top->px + ((size_t)y * top->w + x) * top->c, top->c, fg);
            blend_rgba8(bg + ((size_t)(dy + y) * w + dx + x) * 4, fg);
        }
    dst->w = w; dst->h = h; dst->c = 4; dst->px = bg;
    return 0;
}

/* ---------------------------------------------------------- whole path -- */

int fo_process_pixels(const fo_image *src, const fo_params *p, int arith, fo_image *dst)
{
    fo_image img, oriented;
    /* handler.rs:221-223 */
    if (fo_apply_orientation(src, p->orientation, &oriented)) return -1;
    /* handler.rs:224-228: grayscale XOR invert, grayscale wins */
    if (p->grayscale) {
        int rc = fo_grayscale(&oriented, &img);
        free(oriented.px);
        if (rc) return -1;
    } else {
        img = oriented;
        if (p->inverse) fo_invert(&img);
    }
    /* handler.rs:229-249 */
    if (p->has_dims) {
        uint32_t width = p->w, height = p->h;
        if (width != img.w || height != img.h) {
            fo_image r;
            const int filter = p->filter == FO_FILTER_NEAREST ? FO_FILTER_NEAREST : FO_FILTER_LANCZOS3;
            int rc = p->crop ? fo_resize_to_fill(&img, width, height, filter, arith, &r)
                             : fo_resize(&img, width, height, filter, arith, &r);
            free(img.px);
            if (rc) return -1;
            img = r;
        }
        if (width > img.w || height > img.h) {
            fo_image r;
            int rc = fo_letterbox(&img, width, height, p->fill, &r);
            free(img.px);
            if (rc) return -1;
            img = r;
        }
    }
    /* handler.rs:250-255 */
    if (p->blur_sigma > 0.0f) {
        fo_image r;
        int rc = fo_blur(&img, p->blur_sigma, arith, &r);
        free(img.px);
        if (rc) return -1;
        img = r;
    }
    *dst = img;
    return 0;
}

/* ------------------------------------------------- encoder front ends -- */

/* codecs/jpeg/encoder.rs::rgb_to_ycbcr, f32, truncating `as u8` (saturating in Rust) */
static inline uint8_t sat_u8(float v)
{
    if (!(v > 0.0f)) return 0; /* also NaN -> 0, as Rust's `as u8` */
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

int fo_jpeg_ycbcr444(const fo_image *src, uint8_t *out, uint32_t *pw_o, uint32_t *ph_o)
{
    uint32_t w = src->w, h = src->h, c = src->c;
    if (w == 0 || h == 0) return -1;
    uint32_t pw = (w + 7) & ~7u, ph = (h + 7) & ~7u;
    uint8_t *Y = out, *Cb = out + (size_t)pw * ph, *Cr = out + 2 * (size_t)pw * ph;
    const float max = 255.0f;
    for (uint32_t y = 0; y < ph; ++y)
        for (uint32_t x = 0; x < pw; ++x) {
            /* pixel_at_or_near: replicate the last column / row into the padding */
            uint32_t sx = x < w ? x : w - 1, sy = y < h ? y : h - 1;
            uint8_t px[4];
            to_rgba8(src->px + ((size_t)sy * w + sx) * c, c, px);
            float r = (float)px[0], g = (float)px[1], b = (float)px[2];
            float yy = 76.245f / max * r + 149.685f / max * g + 29.07f / max * b;
            float cb = -43.0185f / max * r - 84.4815f / max * g + 127.5f / max * b + 128.0f;
            float cr = 127.5f / max * r - 106.7685f / max * g - 20.7315f / max * b + 128.0f;
            size_t o = (size_t)y * pw + x;
            Y[o] = sat_u8(yy); Cb[o] = sat_u8(cb); Cr[o] = sat_u8(cr);
        }
    *pw_o = pw; *ph_o = ph;
    return 0;
}

/* libwebp src/enc/picture_csp_enc.c gamma tables + src/dsp/yuv.h fixed point */
enum { YUV_FIX = 16, YUV_HALF = 1 << (YUV_FIX - 1) };
enum { GAMMA_FIX = 12, GAMMA_TAB_FIX = 7, GAMMA_TAB_SIZE = 1 << (GAMMA_FIX - GAMMA_TAB_FIX) };
static int g_lin2gam[GAMMA_TAB_SIZE + 1];
static uint16_t g_gam2lin[256];
static int g_gamma_ok = 0;

static void init_gamma_tables(void)
{
    if (g_gamma_ok) return;
    const double kGamma = 0.80;
    const int kGammaScale = (1 << GAMMA_FIX) - 1;
    const double scale = (double)(1 << GAMMA_TAB_FIX) / kGammaScale;
    const double norm = 1. / 255.;
    for (int v = 0; v <= 255; ++v) g_gam2lin[v] = (uint16_t)(pow(norm * v, kGamma) * kGammaScale + .5);
    for (int v = 0; v <= GAMMA_TAB_SIZE; ++v) g_lin2gam[v] = (int)(255. * pow(scale * v, 1. / kGamma) + .5);
    g_gamma_ok = 1;
}

static inline int lin2gam_interp(int v)
{
    const int kGammaTabScale = 1 << GAMMA_TAB_FIX;
    const int tab_pos = v >> (GAMMA_TAB_FIX + 2);
    const int x = v & ((kGammaTabScale << 2) - 1);
    const int v0 = g_lin2gam[tab_pos];
    const int v1 = g_lin2gam[tab_pos + 1];
    return v1 * x + v0 * ((kGammaTabScale << 2) - x);
}
static inline int linear_to_gamma(uint32_t base_value, int shift)
{
    const int kGammaTabRounder = (1 << GAMMA_TAB_FIX) >> 1;
    const int y = lin2gam_interp((int)(base_value << shift));
    return (y + kGammaTabRounder) >> GAMMA_TAB_FIX;
}
static inline int clip_uv(int uv, int rounding)
{
    uv = (uv + rounding + (128 << (YUV_FIX + 2))) >> (YUV_FIX + 2);
    return ((uv & ~0xff) == 0) ? uv : (uv < 0) ? 0 : 255;
}
static inline int rgb_to_y(int r, int g, int b, int rounding)
{
    const int luma = 16839 * r + 33059 * g + 6420 * b;
    return (luma + rounding + (16 << YUV_FIX)) >> YUV_FIX;
}
static inline int rgb_to_u(int r, int g, int b, int rounding) { return clip_uv(-9719 * r - 19081 * g + 28800 * b, rounding); }
static inline int rgb_to_v(int r, int g, int b, int rounding) { return clip_uv(+28800 * r - 24116 * g - 4684 * b, rounding); }

int fo_webp_yuv420(const fo_image *src, uint8_t *out)
{
    if (src->c != 4) return -1;
    init_gamma_tables();
    uint32_t w = src->w, h = src->h;
    uint32_t uvw = (w + 1) >> 1, uvh = (h + 1) >> 1;
    uint8_t *Y = out, *U = out + (size_t)w * h, *V = U + (size_t)uvw * uvh, *A = V + (size_t)uvw * uvh;
    int has_alpha = 0;
    for (size_t i = 0; i < (size_t)w * h; ++i) if (src->px[i * 4 + 3] != 255) { has_alpha = 1; break; }
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *p = src->px + ((size_t)y * w + x) * 4;
            Y[(size_t)y * w + x] = (uint8_t)rgb_to_y(p[0], p[1], p[2], YUV_HALF);
            if (has_alpha) A[(size_t)y * w + x] = p[3];
        }
    /* AccumulateRGB / AccumulateRGBA + WebPConvertRGBA32ToUV.  Every chroma sample averages a 2x2 block in linear light;
     * a block that is neither fully opaque nor fully transparent weights its pixels by alpha (LinearToGammaWeighted:
     * sum of a_i * GammaToLinear(c_i), times the reciprocal table kInvAlpha[a] = (1 << 19) / a, >> 17).  Odd width /
     * height repeat the last column / row (step = 0 / rgb_stride = 0 in libwebp). */
    for (uint32_t by = 0; by < uvh; ++by) {
        uint32_t y0 = 2 * by, y1 = (2 * by + 1 < h) ? 2 * by + 1 : y0;
        for (uint32_t bx = 0; bx < uvw; ++bx) {
            uint32_t x0 = 2 * bx, x1 = (x0 + 1 < w) ? x0 + 1 : x0;
            const uint8_t *t[4] = {src->px + ((size_t)y0 * w + x0) * 4, src->px + ((size_t)y0 * w + x1) * 4,
                                   src->px + ((size_t)y1 * w + x0) * 4, src->px + ((size_t)y1 * w + x1) * 4};
            const uint32_t a = (uint32_t)t[0][3] + t[1][3] + t[2][3] + t[3][3];
            int rgb[3];
            for (int k = 0; k < 3; ++k) {
                if (a == 4 * 0xff || a == 0) {
                    /* SUM4 (or SUM2 with shift 1 at an odd edge: the same number) */
                    uint32_t s4 = (uint32_t)g_gam2lin[t[0][k]] + g_gam2lin[t[1][k]] + g_gam2lin[t[2][k]] + g_gam2lin[t[3][k]];
                    rgb[k] = linear_to_gamma(s4, 0);
                } else {
                    const uint32_t sum = t[0][3] * (uint32_t)g_gam2lin[t[0][k]] + t[1][3] * (uint32_t)g_gam2lin[t[1][k]] +
                                         t[2][3] * (uint32_t)g_gam2lin[t[2][k]] + t[3][3] * (uint32_t)g_gam2lin[t[3][k]];
                    const uint32_t inv = (1u << 19) / a;               /* kInvAlpha[a], kAlphaFix = 19 */
                    rgb[k] = linear_to_gamma((sum * inv) >> (19 - 2), 0); /* DIVIDE_BY_ALPHA */
                }
            }
            U[(size_t)by * uvw + bx] = (uint8_t)rgb_to_u(rgb[0], rgb[1], rgb[2], YUV_HALF << 2);
            V[(size_t)by * uvw + bx] = (uint8_t)rgb_to_v(rgb[0], rgb[1], rgb[2], YUV_HALF << 2);
        }
    }
    return has_alpha;
}

/* reference src/handler.rs:423-438 (in-repo, restated exactly) */
void fo_ycck_to_cmyk(uint8_t *raw, size_t n_pixels)
{
    for (size_t i = 0; i < n_pixels * 4; i += 4) {
        float y = (float)raw[i], cb = (float)raw[i + 1], cr = (float)raw[i + 2];
        float r = y + 1.40200f * cr - 179.456f;
        float g = y - 0.34414f * cb - 0.71414f * cr + 135.45984f;
        float b = y + 1.77200f * cb - 226.816f;
        r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
        g = g < 0.0f ? 0.0f : (g > 255.0f ? 255.0f : g);
        b = b < 0.0f ? 0.0f : (b > 255.0f ? 255.0f : b);
        raw[i] = (uint8_t)r; raw[i + 1] = (uint8_t)g; raw[i + 2] = (uint8_t)b;
        raw[i + 3] = (uint8_t)(255 - raw[i + 3]);
    }
}

/* ------------------------------------------------------------ CMYK CLUT -- */
/* Little CMS 2 (2.12 .. 2.16 share this code) cmsintrp.c: _cmsToFixedDomain, Eval4Inputs, LinearInterp; lcms2_internal.h
 * FROM_8_TO_16 / FROM_16_TO_8.  cmsS15Fixed16Number is int32 and the library relies on wrap-around, so the sums
 * are formed in uint32 here. */
static int32_t to_fixed_domain(int32_t a) { return (int32_t)((uint32_t)a + (uint32_t)((int32_t)((uint32_t)a + 0x7fffu) / 0xffff)); }

static void eval3_tetra(const uint16_t *t, uint32_t nout, int32_t X0, int32_t X1, int32_t Y0, int32_t Y1, int32_t Z0, int32_t Z1,
                        int32_t rx, int32_t ry, int32_t rz, uint16_t *out)
{
#define DENS(i, j, k) ((int32_t)t[(i) + (j) + (k) + ch])
    for (uint32_t ch = 0; ch < nout; ++ch) {
        int32_t c0 = DENS(X0, Y0, Z0), c1, c2, c3;
        if (rx >= ry && ry >= rz) {
            c1 = DENS(X1, Y0, Z0) - c0; c2 = DENS(X1, Y1, Z0) - DENS(X1, Y0, Z0); c3 = DENS(X1, Y1, Z1) - DENS(X1, Y1, Z0);
        } else if (rx >= rz && rz >= ry) {
            c1 = DENS(X1, Y0, Z0) - c0; c2 = DENS(X1, Y1, Z1) - DENS(X1, Y0, Z1); c3 = DENS(X1, Y0, Z1) - DENS(X1, Y0, Z0);
        } else if (rz >= rx && rx >= ry) {
            c1 = DENS(X1, Y0, Z1) - DENS(X0, Y0, Z1); c2 = DENS(X1, Y1, Z1) - DENS(X1, Y0, Z1); c3 = DENS(X0, Y0, Z1) - c0;
        } else if (ry >= rx && rx >= rz) {
            c1 = DENS(X1, Y1, Z0) - DENS(X0, Y1, Z0); c2 = DENS(X0, Y1, Z0) - c0; c3 = DENS(X1, Y1, Z1) - DENS(X1, Y1, Z0);
        } else if (ry >= rz && rz >= rx) {
            c1 = DENS(X1, Y1, Z1) - DENS(X0, Y1, Z1); c2 = DENS(X0, Y1, Z0) - c0; c3 = DENS(X0, Y1, Z1) - DENS(X0, Y1, Z0);
        } else if (rz >= ry && ry >= rx) {
            c1 = DENS(X1, Y1, Z1) - DENS(X0, Y1, Z1); c2 = DENS(X0, Y1, Z1) - DENS(X0, Y0, Z1); c3 = DENS(X0, Y0, Z1) - c0;
        } else {
            c1 = c2 = c3 = 0;
        }
        const int32_t rest = (int32_t)((uint32_t)c1 * (uint32_t)rx + (uint32_t)c2 * (uint32_t)ry + (uint32_t)c3 * (uint32_t)rz);
        out[ch] = (uint16_t)(c0 + (((int32_t)((uint32_t)to_fixed_domain(rest) + 0x8000u)) >> 16));
    }
#undef DENS
}

void fo_cmyk_to_rgb(const uint8_t *cmyk, size_t n_pixels, const uint16_t *clut, uint32_t grid, uint8_t *rgb)
{
    const int32_t domain = (int32_t)grid - 1;
    const int32_t opta0 = 3, opta1 = 3 * (int32_t)grid, opta2 = opta1 * (int32_t)grid, opta3 = opta2 * (int32_t)grid;
    for (size_t p = 0; p < n_pixels; ++p) {
        uint16_t in[4], t1[3], t2[3];
        for (int k = 0; k < 4; ++k) in[k] = (uint16_t)((cmyk[p * 4 + k] << 8) | cmyk[p * 4 + k]); /* FROM_8_TO_16 */
        const int32_t fk = to_fixed_domain((int32_t)in[0] * domain), fx = to_fixed_domain((int32_t)in[1] * domain),
                      fy = to_fixed_domain((int32_t)in[2] * domain), fz = to_fixed_domain((int32_t)in[3] * domain);
        const int32_t k0 = fk >> 16, x0 = fx >> 16, y0 = fy >> 16, z0 = fz >> 16;
        const int32_t rk = fk & 0xffff, rx = fx & 0xffff, ry = fy & 0xffff, rz = fz & 0xffff;
        const int32_t K0 = opta3 * k0, K1 = K0 + (in[0] == 0xffff ? 0 : opta3);
        const int32_t X0 = opta2 * x0, X1 = X0 + (in[1] == 0xffff ? 0 : opta2);
        const int32_t Y0 = opta1 * y0, Y1 = Y0 + (in[2] == 0xffff ? 0 : opta1);
        const int32_t Z0 = opta0 * z0, Z1 = Z0 + (in[3] == 0xffff ? 0 : opta0);
        eval3_tetra(clut + K0, 3, X0, X1, Y0, Y1, Z0, Z1, rx, ry, rz, t1);
        eval3_tetra(clut + K1, 3, X0, X1, Y0, Y1, Z0, Z1, rx, ry, rz, t2);
        for (int k = 0; k < 3; ++k) {
            uint32_t dif = (uint32_t)((int32_t)t2[k] - (int32_t)t1[k]) * (uint32_t)rk + 0x8000u; /* LinearInterp */
            dif = (dif >> 16) + t1[k];
            const uint16_t o16 = (uint16_t)dif;
            rgb[p * 3 + k] = (uint8_t)(((uint32_t)o16 * 65281u + 8388608u) >> 24);               /* FROM_16_TO_8 */
        }
    }
}
