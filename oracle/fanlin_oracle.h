/*
 * fanlin_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the pixel arithmetic that livesense-inc/fanlin-rs
 * executes inside State::process_image (reference src/handler.rs:185-309).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (fanlin-rs_amd/csrc) never does.
 *
 * PARITY UNPINNED: the arithmetic lives in third-party crates that are not
 * vendored under /root/reference (image 0.25.6, Cargo.lock:1948; webp 0.3.0 /
 * libwebp-sys 0.9.6, Cargo.lock:4019,2156) and no Rust toolchain exists in
 * this environment, so the reference itself could not be run.  No reference
 * test pins a pixel (reference src/main.rs:457-468 asserts status + MIME
 * only).  The restatement follows the published source of those crates at
 * the pinned versions; every function names the upstream routine it follows.
 */
#ifndef FANLIN_ORACLE_H
#define FANLIN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Tightly packed, row-major, top-left origin, interleaved u8 channels:
 * 1 = Luma8, 2 = LumaA8, 3 = Rgb8, 4 = Rgba8 (image::DynamicImage layouts). */
typedef struct fo_image {
    uint32_t w, h, c;
    uint8_t *px; /* malloc'd by the oracle when it is an output; fo_free() it */
} fo_image;

/* How t += v * w is evaluated in the resample loops.
 * FO_ARITH_REF: separate f32 multiply then f32 add -- what rustc emits for the
 *               reference (no contraction).  This is THE reference arithmetic.
 * FO_ARITH_FMA: the arithmetic of the HIP kernels, used to prove the kernels
 *               bit-exact against a CPU restatement: one fused fmaf per tap; the
 *               vertical pass in tap order; the horizontal pass of the Lanczos3 resize
 *               grouped by aligned blocks of 4 source pixels (block sums added in
 *               ascending order), the horizontal pass of the blur in tap order.
 *               Differs from FO_ARITH_REF by <= 1 LSB. */
enum { FO_ARITH_REF = 0, FO_ARITH_FMA = 1 };

enum { FO_FILTER_LANCZOS3 = 0, FO_FILTER_GAUSSIAN = 1, FO_FILTER_NEAREST = 2, FO_FILTER_TRIANGLE = 3 };

/* Accessor outputs of query::Query (reference src/query.rs:28-78). */
typedef struct fo_params {
    int has_dims;       /* Query::dimensions().is_some()            query.rs:28-33 */
    uint32_t w, h;
    uint8_t fill[3];    /* Query::fill_color()                      query.rs:35-49 */
    int crop;           /* Query::cropping()                        query.rs:55-57 */
    float blur_sigma;   /* Query::blur(): 0.0 or clamp(v,10,20)     query.rs:59-62 */
    int grayscale;      /* Query::grayscale()                       query.rs:64-66 */
    int inverse;        /* Query::inverse()                         query.rs:68-70 */
    int orientation;    /* EXIF orientation 1..8 (decoder.orientation(), handler.rs:206); 0 = none */
    int filter;         /* FO_FILTER_LANCZOS3 (process_image, handler.rs:233,235) or FO_FILTER_NEAREST (process_gif, 338,340) */
} fo_params;

void fo_free(void *p);

/* image 0.25.6 src/math/utils.rs::resize_dimensions */
void fo_resize_dimensions(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, int fill,
                          uint32_t *ow, uint32_t *oh);

/* Weight table of one axis, image 0.25.6 src/imageops/sample.rs
 * (vertical_sample / horizontal_sample share this index + weight maths).
 * left[o], count[o] (o < out_size); weights are written packed, output o's
 * taps at weights[offset[o] .. offset[o]+count[o]).  offset has out_size+1
 * entries.  Returns total taps, or -1 if `cap` floats is too small. */
long fo_build_weights(uint32_t in_size, uint32_t out_size, int filter, float sigma,
                      uint32_t *left, uint32_t *count, uint32_t *offset,
                      float *weights, size_t cap);

/* DynamicImage::apply_orientation(Orientation::from_exif(code)) (handler.rs:221-223) */
int fo_apply_orientation(const fo_image *src, int exif, fo_image *dst);

/* image::imageops::colorops::grayscale / grayscale_alpha (color.rs rgb_to_luma). */
int fo_grayscale(const fo_image *src, fo_image *dst);
/* image::imageops::colorops::invert (color.rs Invert impls), in place. */
void fo_invert(fo_image *img);

/* image::imageops::resize(image, nw, nh, filter) */
int fo_resize_exact(const fo_image *src, uint32_t nw, uint32_t nh, int filter, int arith, fo_image *dst);
/* DynamicImage::resize (aspect preserving) and ::resize_to_fill (cover + centre crop) */
int fo_resize(const fo_image *src, uint32_t nw, uint32_t nh, int filter, int arith, fo_image *dst);
int fo_resize_to_fill(const fo_image *src, uint32_t nw, uint32_t nh, int filter, int arith, fo_image *dst);
/* image::imageops::blur(image, sigma) as of 0.25.6 */
int fo_blur(const fo_image *src, float sigma, int arith, fo_image *dst);
/* ImageBuffer::from_pixel(w,h,Rgba[r,g,b,255]) + imageops::overlay (handler.rs:238-248) */
int fo_letterbox(const fo_image *top, uint32_t w, uint32_t h, const uint8_t fill[3], fo_image *dst);

/* The pixel part of State::process_image, handler.rs:224-255, in its order. */
int fo_process_pixels(const fo_image *src, const fo_params *p, int arith, fo_image *dst);

/* JPEG encoder colour front end of image 0.25.6 codecs/jpeg/encoder.rs
 * (rgb_to_ycbcr + copy_blocks_ycbcr edge replication).  Planes are
 * pw x ph with pw = ceil8(w), ph = ceil8(h); out = Y | Cb | Cr, 3*pw*ph bytes. */
int fo_jpeg_ycbcr444(const fo_image *src, uint8_t *out, uint32_t *pw, uint32_t *ph);

/* libwebp picture_csp_enc.c ImportYUVAFromRGBA (no dithering, no sharp yuv):
 * Y w*h | U ((w+1)/2)*((h+1)/2) | V same.  Source must be Rgba8.
 * Returns 1 if the picture has non-opaque alpha (alpha plane then follows V), 0 if not. */
int fo_webp_yuv420(const fo_image *src, uint8_t *out);

/* handler.rs:423-438, YCCK -> "CMYK with inverted K" pointwise loop, in place on n pixels x 4. */
void fo_ycck_to_cmyk(uint8_t *raw, size_t n_pixels);

/* JPEG encoder back half (fanlin_oracle_jpeg.c): image 0.25.6 codecs/jpeg/encoder.rs + transform.rs as called at
 * handler.rs:274-278.  Coefficients are quantised, zig-zag ordered, unit = (block_row*blocks_x + block_col)*3 + comp. */
void fo_jpeg_qtables(int quality, uint8_t luma[64], uint8_t chroma[64]);
void fo_jpeg_fdct(const uint8_t samples[64], int32_t coeffs[64]);
int fo_jpeg_coefficients(const fo_image *src, int quality, int16_t *out);
size_t fo_jpeg_header(uint32_t width, uint32_t height, int quality, uint8_t *out, size_t cap);
size_t fo_jpeg_encode(const fo_image *src, int quality, uint8_t *out, size_t cap);

/* lcms2 transform_pixels for the transform of handler.rs:469-488 (CMYK_8 -> RGB_8), given the 17^4 device-link
 * table Little CMS precomputes for it (cmsopt.c OptimizeByResampling): formatters Unroll4Bytes / Pack3Bytes and
 * cmsintrp.c Eval4Inputs, in its 16.16 fixed point.  clut = grid^4 x 3 u16, node index ((c*grid+m)*grid+y)*grid+k.
 * PINNED: tests compare this function with the system's liblcms2 itself (tests/golden/cmyk_lcms2.npz and, where
 * the library is present, live). */
void fo_cmyk_to_rgb(const uint8_t *cmyk, size_t n_pixels, const uint16_t *clut, uint32_t grid, uint8_t *rgb);

/* ---- baseline JPEG decoding (fanlin_oracle_jpegdec.c): reference src/handler.rs:205-220 via zune-jpeg 0.4.14 ---- */
/* 0 = ok, -1 = malformed, -2 = valid JPEG this decoder does not cover (progressive, arithmetic, 12-bit, several scans) */
int fo_jpeg_info(const uint8_t *data, size_t n, uint32_t *w, uint32_t *h, uint32_t *components, uint32_t *exif_orientation);
/* out: w*h (one component: Luma8), w*h*3 (Rgb8) or w*h*4 (four components: raw CMYK / YCCK samples) bytes */
int fo_jpeg_adobe_transform(const uint8_t *data, size_t n);
int fo_jpeg_decode(const uint8_t *data, size_t n, uint8_t *out);
/* entropy decoding only: quantised coefficients [block][64] in zig-zag order, component after component */
int fo_jpeg_coefficients_of(const uint8_t *data, size_t n, int16_t *sink, size_t sink_blocks, uint32_t *nblocks);

#ifdef __cplusplus
}
#endif
#endif
