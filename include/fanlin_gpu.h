/*
 * fanlin_gpu.h -- C ABI of the MI355X-native image hot path for fanlin-rs.
 *
 * This is the drop-in boundary: a reference-side shim (Rust `extern "C"`
 * block) binds exactly these entry points and keeps the
 * rest of fanlin-rs (axum handler, routing, origin fetch, codecs' entropy
 * stages) unchanged.  Plain pointers and sizes only; no C++ or torch types;
 * no exceptions cross this boundary; every function returns an flgpu_status
 * (0 = ok) unless it returns a pointer.
 *
 * What each entry point replaces in the reference (paths relative to the
 * fanlin-rs repository root):
 *
 *   flgpu_query_parse / flgpu_query_*   src/query.rs:3-94   (query::Query + accessors)
 *   FLGPU_ACCEPT_* flags                src/content.rs:12-48 (content::Format)
 *   flgpu_params_from_query             src/handler.rs:224-261 (which accessors feed which step)
 *   flgpu_plan_output                   image::math::utils::resize_dimensions +
 *                                       DynamicImage::resize / resize_to_fill geometry +
 *                                       the letterbox rule at src/handler.rs:229-249
 *   flgpu_transform*                    img.apply_orientation (src/handler.rs:221-223) and the image-crate calls at
 *                                       src/handler.rs:225,227,233,
 *                                       235,240-247,253 (grayscale, invert, resize,
 *                                       resize_to_fill, from_pixel + overlay, blur)
 *   front_end = FLGPU_FE_JFIF444        colour front end of jpeg::JpegEncoder::encode_image,
 *                                       src/handler.rs:274-278
 *   front_end = FLGPU_FE_WEBP420        WebPPictureImportRGBA + ARGB->YUV420 inside
 *                                       webp::Encoder::encode, src/handler.rs:295-297
 *   front_end = FLGPU_FE_JPEG           the whole jpeg::JpegEncoder::new_with_quality(q).encode_image(&img),
 *                                       src/handler.rs:274-278 (FDCT, quantiser, Huffman coder, framing)
 *   params.filter = NEAREST             the per-frame pipeline of process_gif, src/handler.rs:327-353
 *   flgpu_process_image                 State::process_image after the decoder as one call, src/handler.rs:198-308
 *   flgpu_ycck_to_cmyk                  the YCCK loop of convert_jpeg_color_if_needed, src/handler.rs:423-438
 *   flgpu_set_cmyk_profile / _clut      create_cmyk_to_rgb_converter + CMYK2RGB::with_icc_profile,
 *                                       src/main.rs:74-76, src/handler.rs:469-488
 *   flgpu_cmyk_to_rgb[_device]          CMYK2RGB::convert = lcms2 transform_pixels, src/handler.rs:446-462,490-492
 *   FLGPU_IMG_JPEG_SOURCE / flgpu_process_jpeg   JpegDecoder::new + DynamicImage::from_decoder, src/handler.rs:205-220
 *   flgpu_create / flgpu_destroy        lifetime of handler::State, src/handler.rs:14-21,36-52
 *   flgpu_config.devices / flgpu_plan_shards   the one shared Arc<State> behind all tokio workers, src/main.rs:108-112
 */
#ifndef FANLIN_GPU_H
#define FANLIN_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLGPU_ABI_VERSION 6

typedef enum flgpu_status {
    FLGPU_OK = 0,
    FLGPU_ERR_INVALID_ARG = 1,   /* null pointer, zero-sized image, channels not in 1..4, ... */
    FLGPU_ERR_UNSUPPORTED = 2,   /* valid request the device path does not cover */
    FLGPU_ERR_NO_DEVICE = 3,     /* no usable HIP device: the library never falls back to the CPU */
    FLGPU_ERR_OOM = 4,           /* host or device allocation failed */
    FLGPU_ERR_DEVICE = 5,        /* a HIP call failed; see flgpu_last_error() */
    FLGPU_ERR_PARSE = 6,         /* query string rejected (axum would answer 400) */
    FLGPU_ERR_BUFFER_TOO_SMALL = 7,
    FLGPU_ERR_SHUTDOWN = 8       /* context is being destroyed */
} flgpu_status;

/* Decoded image as the image crate holds it: tightly packed rows, top-left
 * origin, interleaved u8 channels; 1 = Luma8, 2 = LumaA8, 3 = Rgb8, 4 = Rgba8. */
typedef struct flgpu_image {
    uint8_t *data;      /* host pointer (flgpu_transform, _batch) or device pointer (_batch_device) */
    uint64_t capacity;  /* src: bytes valid at data; dst: bytes writable at data */
    uint32_t width, height, channels;
    uint32_t flags;     /* out: FLGPU_IMG_* */
    uint64_t bytes;     /* out: bytes produced at data (pixels, planes or an encoded stream) */
} flgpu_image;

#define FLGPU_IMG_FRONTEND_PLANES 1u  /* dst->data holds encoder planes, not interleaved pixels */
#define FLGPU_IMG_PINNED          8u  /* in (src and dst of flgpu_transform): data comes from flgpu_host_alloc, i.e. is page-locked:
                                         the copy through the library's own pinned staging blocks is skipped */
#define FLGPU_IMG_ENCODED         4u  /* JPEG: data holds an encoded stream of `bytes` bytes */
#define FLGPU_IMG_JPEG_SOURCE     16u /* in (src of flgpu_transform / flgpu_transform_batch): data holds a baseline JPEG FILE of `capacity`
                                         bytes instead of pixels, width / height / channels say what it decodes to (flgpu_jpeg_info):
                                         the library decodes it itself -- Huffman decoding on the calling thread, dequantisation, IDCT,
                                         chroma up-sampling and YCbCr -> RGB on the device -- replacing JpegDecoder::new +
                                         DynamicImage::from_decoder (src/handler.rs:205-220).  ~1 MB crosses PCIe instead of 6.2 MB
                                         for a 1080p picture.  Streams it does not cover: FLGPU_ERR_UNSUPPORTED (decode on the host). */
#define FLGPU_IMG_HAS_ALPHA       2u  /* WEBP420: some pixel is not opaque: the picture is WEBP_YUV420A for libwebp, i.e. the A
                                         plane behind V must be handed to WebPEncode too (for opaque pictures it is all 255) */

/* Encoder colour front end to run after the pixel pipeline. */
typedef enum flgpu_front_end {
    FLGPU_FE_NONE = 0,     /* dst = interleaved pixels, out_c channels */
    FLGPU_FE_JFIF444 = 1,  /* dst = Y | Cb | Cr, each plane_w x plane_h (multiples of 8, edge replicated) */
    FLGPU_FE_WEBP420 = 2,  /* dst = Y (w x h) | U | V (each ceil(w/2) x ceil(h/2)) | A (w x h), BT.601 limited range: what
                              libwebp's WebPPictureARGBToYUVA makes of the RGBA picture, alpha-weighted chroma included */
    FLGPU_FE_JPEG = 3      /* dst = the finished JFIF stream of JpegEncoder::new_with_quality(q).encode_image(&img)
                              (src/handler.rs:274-278): baseline, 3 components, 4:4:4; dst->bytes long */
} flgpu_front_end;

/* content::Format bits (src/content.rs:15-16). */
#define FLGPU_ACCEPT_WEBP 1u
#define FLGPU_ACCEPT_AVIF 2u

/* Output container chosen at src/handler.rs:256-261. */
typedef enum flgpu_out_format { FLGPU_OUT_KEEP = 0, FLGPU_OUT_WEBP = 1, FLGPU_OUT_AVIF = 2 } flgpu_out_format;

/* query::Query, field for field (src/query.rs:4-15); has_* = Option::is_some. */
typedef struct flgpu_query {
    uint8_t has_w, has_h, has_rgb, has_quality, has_crop, has_blur, has_grayscale, has_inverse, has_avif, has_webp;
    uint8_t quality, crop, blur, grayscale, inverse, avif, webp;
    uint8_t reserved;
    uint32_t w, h;
    char rgb[112];      /* NUL terminated copy of the raw `rgb` value (truncated if longer) */
} flgpu_query;

/* Exactly the accessor outputs of query::Query that the pixel pipeline consumes. */
enum flgpu_filter { FLGPU_FILTER_LANCZOS3 = 0, FLGPU_FILTER_NEAREST = 1 };

typedef struct flgpu_params {
    uint32_t has_dims;                 /* Query::dimensions().is_some() */
    uint32_t w, h;
    uint8_t fill_r, fill_g, fill_b;    /* Query::fill_color() */
    uint8_t crop;                      /* Query::cropping() */
    float blur_sigma;                  /* Query::blur(): 0.0 or 10.0..=20.0 */
    uint8_t grayscale, inverse;        /* Query::grayscale(), Query::inverse() */
    uint8_t quality;                   /* Query::quality() (carried for the host encoder) */
    uint8_t front_end;                 /* flgpu_front_end */
    uint8_t orientation;               /* EXIF orientation 1..8 from decoder.orientation() (src/handler.rs:206,221-223); 0 or 1 = none */
    uint8_t filter;                    /* flgpu_filter: Lanczos3 for stills (handler.rs:233,235), Nearest for GIF frames (handler.rs:338,340) */
    uint8_t reserved[2];
} flgpu_params;

/* Geometry decided on the host before any pixel is touched. */
typedef struct flgpu_plan {
    uint32_t src_w, src_h;             /* source size after img.apply_orientation() (swapped for EXIF 5..8) */
    uint32_t mid_c;                    /* channels after grayscale/invert */
    uint32_t resampled;                /* 1 if a Lanczos3 pass runs */
    uint32_t resized_w, resized_h;     /* resize_exact target */
    uint32_t crop_x, crop_y;           /* resize_to_fill centre-crop origin */
    uint32_t letterboxed;              /* 1 if overlay onto the fill colour runs (output becomes Rgba8) */
    uint32_t place_x, place_y;         /* overlay offset */
    uint32_t out_w, out_h, out_c;      /* pixel image after letterbox/blur */
    uint32_t plane_w, plane_h;         /* front end luma plane size */
    uint32_t chroma_w, chroma_h;       /* front end chroma plane size */
    uint64_t pixel_bytes;              /* out_w * out_h * out_c */
    uint64_t out_bytes;                /* bytes the call writes to dst (pixels or planes); for FLGPU_FE_JPEG a planning bound
                                          (about 1 byte per sample) that ordinary pictures stay far below */
    uint64_t max_out_bytes;            /* FLGPU_FE_JPEG: the worst case of the format (every 8x8 block of every component at its
                                          longest code, every byte stuffed): a dst of this capacity can never be too small, as
                                          JpegEncoder::encode_image into a Vec never fails (src/handler.rs:274-278).  Otherwise
                                          equal to out_bytes. */
} flgpu_plan;

#define FLGPU_MAX_DEVICES 8
typedef struct flgpu_config {
    int32_t device;            /* HIP device ordinal; -1 = current device (used when n_devices <= 1) */
    uint32_t max_batch;        /* request-queue flush size PER DEVICE (0 = default 16); a flush also ends at 16 MB of pixel sources
                                  per device (three 1080p buffers: the uploads in front of its first kernel), JPEG files not counted */
    uint32_t flush_timeout_us; /* request-queue flush timer (0 = default 200) */
    uint32_t profile;          /* 1 = bracket kernels with HIP events and report them in flgpu_stats */
    uint32_t queue_lanes;      /* flgpu_transform: batches kept in flight at once PER DEVICE (each lane has its own stream
                                  and scratch, so one batch's PCIe transfers overlap another's kernels); 0 = default: 4 lanes, plus 4 overflow
                                  lanes that only take a full batch that is waiting while those four are busy (more callers than
                                  4 x max_batch); a value of 1..8 = exactly that many lanes */
    uint32_t n_devices;        /* 0 or 1: one GPU (`device`).  2..FLGPU_MAX_DEVICES: ONE context for the GPUs of a node, as
                                  the reference shares one Arc<State> between all its workers (src/main.rs:108-112): every
                                  flushed batch of the request queue and every batch call is split into n_devices contiguous
                                  shards balanced by algorithmic bytes (W*H*C + output bytes), shard k runs on devices[k],
                                  results come back in request order.  An ordinal may repeat (two shards on one GPU). */
    uint32_t use_embedded_profile; /* config `use_embedded_profile` (src/handler.rs:19,446-458): CMYK / YCCK JPEG SOURCES decoded by the
                                      library are converted with their own embedded ICC profile when they carry a usable one */
    uint32_t decode_threads;   /* flgpu_transform with FLGPU_IMG_JPEG_SOURCE: callers that may run the host half of the decoder (Huffman
                                  decoding, ~2 ms of CPU per 1080p file) at the same time; the others wait their turn.  0 = the CPUs this
                                  process may use (cgroup quota / affinity mask): more runnable decoders than CPUs only lengthens every
                                  request's decode -- the reference bounds its in-flight requests the same way (max_clients, src/main.rs:108-110) */
    int32_t devices[FLGPU_MAX_DEVICES];
} flgpu_config;

typedef struct flgpu_stats {
    uint64_t images;              /* images transformed */
    uint64_t batches;             /* kernel batches launched */
    uint64_t queue_flushes;       /* request-queue flushes */
    uint64_t tables_built;        /* weight tables built (cache misses) */
    uint64_t resample_launches;   /* launches of the fused resample kernels (streaming + matrix-pipe) */
    double resample_ms;           /* summed HIP-event time of those launches (profile = 1) */
    uint64_t resample_src_bytes;  /* algorithmic bytes read by those launches */
    uint64_t resample_dst_bytes;  /* algorithmic bytes written by those launches */
    uint64_t generic_launches;    /* launches of the two-pass generic resample kernels (through an LDS tile, or through HBM) */
    uint64_t blur_launches;
    double blur_ms;
    uint64_t frontend_launches;
    double frontend_ms;
    uint64_t cmyk_pixels;         /* pixels converted CMYK -> RGB */
    uint64_t cmyk_tables_baked;   /* device-link tables baked from ICC profiles */
    uint64_t jpeg_sources;        /* FLGPU_IMG_JPEG_SOURCE pictures decoded on the device */
    uint64_t jpeg_file_bytes;     /* their file bytes ... */
    uint64_t jpeg_upload_bytes;   /* ... and what crossed PCIe for them (coefficient blobs) */
    uint64_t mfma_launches;       /* launches of the matrix-pipe kernels (fl_mfma.hip, fl_wtile.hip; the latter also for blurs) */
    uint64_t jpeg_device_huffman; /* of jpeg_sources: files whose entropy-coded segment was decoded on the device too (fl_jpeghuff_dev.hip) */
    uint64_t jpeg_device_huffman_retries; /* ... of which the device gave up on and the host decoded after all */
    uint64_t wtile_launches;      /* of mfma_launches: launches of the window-tile matrix-pipe kernel (fl_wtile.hip: mild ratios, up-scales, blurs) */
} flgpu_stats;

typedef struct flgpu_ctx flgpu_ctx;

/* ---- request model (host only, no device needed) ---------------------- */

/* Parses "w=300&h=200&rgb=32,32,32" (the part after '?', or a whole URI) with
 * the semantics of axum::extract::Query<query::Query>: unknown keys ignored,
 * a known key with an unparsable value is an error.  Returns FLGPU_ERR_PARSE
 * where axum would reject the request with 400. */
int flgpu_query_parse(const char *query_string, flgpu_query *out);
int flgpu_query_dimensions(const flgpu_query *q, uint32_t *w, uint32_t *h); /* 1 = Some */
void flgpu_query_fill_color(const flgpu_query *q, uint8_t *r, uint8_t *g, uint8_t *b);
uint8_t flgpu_query_quality(const flgpu_query *q);
int flgpu_query_cropping(const flgpu_query *q);
float flgpu_query_blur(const flgpu_query *q);
int flgpu_query_grayscale(const flgpu_query *q);
int flgpu_query_inverse(const flgpu_query *q);
int flgpu_query_use_avif(const flgpu_query *q);
int flgpu_query_use_webp(const flgpu_query *q);
int flgpu_query_as_is(const flgpu_query *q);
int flgpu_query_unsupported_scale_size(const flgpu_query *q);

/* Query + content::Format -> pipeline parameters and the output container
 * (src/handler.rs:256-261).  front_end is set to WEBP420 / JFIF444 / NONE from
 * the chosen container and `input_is_jpeg` (JPEG stays JPEG, anything else
 * keeps its own encoder and gets pixels). */
int flgpu_params_from_query(const flgpu_query *q, uint32_t accept_flags, int input_is_jpeg,
                            flgpu_params *params, int *out_format);
/* input_is_jpeg: 0 = no, 1 = JPEG and the caller's encoder wants Y/Cb/Cr planes (FLGPU_FE_JFIF444),
 * 2 = JPEG and the library finishes the stream (FLGPU_FE_JPEG). */

/* Pure function: output geometry for a source of sw x sh x sc. */
int flgpu_plan_output(const flgpu_params *p, uint32_t sw, uint32_t sh, uint32_t sc, flgpu_plan *plan);

/* ---- CMYK / YCCK JPEG sources (reference src/handler.rs:398-493) ------------- */

#define FLGPU_CMYK_GRID 17u          /* nodes per axis of the device-link table (Little CMS default for 4 inputs) */
#define FLGPU_CMYK_INPUT_YCCK 1u     /* pixels are (Y, Cb, Cr, K): run the loop of handler.rs:423-438 first */

/* Replaces create_cmyk_to_rgb_converter / CMYK2RGB::with_icc_profile (src/main.rs:74-76, src/handler.rs:469-488):
 * bakes the CMYK_8 -> sRGB RGB_8, Intent::Perceptual transform of `icc` into the 17^4 table Little CMS itself
 * interpolates, using the system's liblcms2 (dlopen).  FLGPU_ERR_INVALID_ARG = not a usable CMYK profile (the
 * reference's `None`), FLGPU_ERR_UNSUPPORTED = liblcms2 missing on this host. */
int flgpu_set_cmyk_profile(flgpu_ctx *ctx, const uint8_t *icc, uint64_t icc_len);
int flgpu_cmyk_bake_available(void); /* 1 if liblcms2 could be loaded */
/* The same table handed over / read back as grid^4 x 3 u16 (R, G, B), node index ((c*grid + m)*grid + y)*grid + k:
 * for hosts that bake it themselves and for copying rank 0's table to the other GPUs. */
int flgpu_set_cmyk_clut(flgpu_ctx *ctx, uint32_t grid, const uint16_t *rgb_nodes);
int flgpu_get_cmyk_clut(flgpu_ctx *ctx, uint16_t *rgb_nodes, uint64_t capacity_entries, uint32_t *grid);
/* A context that spans several devices bakes the table once and hands it from devices[0] to the others: 2 = by one
 * ncclBroadcast (RCCL over xGMI; one rank per distinct GPU, librccl loaded on demand), 1 = by plain copies (shards sharing
 * a GPU, or no RCCL on the host), 0 = single-device context or no table yet. */
int flgpu_cmyk_distribution(flgpu_ctx *ctx);
/* Self-test of that RCCL path on ONE device: loads librccl as the distribution does, resolves the same five symbols, creates
 * a one-rank communicator (ncclCommInitAll), broadcasts 250,563 bytes out of place as "ncclUint8" between group calls,
 * checks that exactly those bytes arrived, destroys the communicator.  FLGPU_ERR_UNSUPPORTED = no RCCL on this host (the
 * distribution then uses copies).  info (may be NULL): [0] RCCL version, [1] bytes that arrived intact, [2] 1 if nothing was
 * written past them, [3] 1 once the communicator has been destroyed. */
int flgpu_rccl_selftest(int device, uint32_t info[4]);
/* Replaces CMYK2RGB::convert = lcms2 transform_pixels (src/handler.rs:490-492) and, with
 * FLGPU_CMYK_INPUT_YCCK, the YCCK loop in front of it (423-438).  n_pixels x 4 bytes in, n_pixels x 3 bytes out.
 * `embedded_icc` (may be NULL) is the JPEG's own profile when use_embedded_profile is set: it is baked once and
 * cached by content; if it cannot be used the configured profile is (handler.rs:446-458).  With no usable
 * profile at all: FLGPU_ERR_UNSUPPORTED (the reference returns None and decodes the JPEG the ordinary way). */
int flgpu_cmyk_to_rgb(flgpu_ctx *ctx, const uint8_t *cmyk, uint64_t n_pixels, uint8_t *rgb,
                      const uint8_t *embedded_icc, uint64_t icc_len, uint32_t flags);
/* Device-resident variant with the configured profile: d_cmyk 16-byte aligned and readable up to a multiple of 4
 * pixels, d_rgb 4-byte aligned and writable up to a multiple of 4 pixels; returns after enqueueing. */
int flgpu_cmyk_to_rgb_device(flgpu_ctx *ctx, const void *d_cmyk, void *d_rgb, uint64_t n_pixels, uint32_t flags,
                             void *hip_stream);

/* ---- device context ---------------------------------------------------- */

/* Returns NULL on failure; *status (optional) receives the reason. */
flgpu_ctx *flgpu_create(const flgpu_config *cfg, int *status);
void flgpu_destroy(flgpu_ctx *ctx);

/* One image, host memory, blocking.  Thread-safe: concurrent callers are
 * packed into shared kernel launches by the context's request queue. */
int flgpu_transform(flgpu_ctx *ctx, const flgpu_image *src, const flgpu_params *p, flgpu_image *dst);

/* State::process_image after the decoder, as one call (src/handler.rs:198-308 minus decoding and the host-only
 * encoders): `decoded` = DynamicImage::from_decoder's pixels, `exif_orientation` = decoder.orientation() (1..8),
 * `query_string` = the request's query, `accept_flags` = content::Format, `input_format` = what with_guessed_format
 * said.  The outcome is one of flgpu_result_kind:
 *   AS_IS         params.as_is() (handler.rs:202-204): nothing was done, serve the original bytes
 *   JPEG_STREAM   input was JPEG and no other container was negotiated: dst holds the finished "image/jpeg" body
 *   WEBP_PLANES   lossy WebP was negotiated (handler.rs:286-297): dst holds Y | U | V for WebPEncode
 *   PIXELS        everything else (PNG, AVIF, lossless WebP, GIF frames, ...): dst holds the DynamicImage pixels for
 *                 the crate's own encoder; *out_format says which container was negotiated
 * Errors: FLGPU_ERR_PARSE where axum answers 400 (bad query, or the size gate of src/main.rs:134-138). */
typedef enum flgpu_input_format { FLGPU_IN_OTHER = 0, FLGPU_IN_JPEG = 1, FLGPU_IN_PNG = 2, FLGPU_IN_WEBP = 3, FLGPU_IN_GIF_FRAME = 4 } flgpu_input_format;
typedef enum flgpu_result_kind { FLGPU_RESULT_AS_IS = 0, FLGPU_RESULT_JPEG_STREAM = 1, FLGPU_RESULT_WEBP_PLANES = 2, FLGPU_RESULT_PIXELS = 3 } flgpu_result_kind;
int flgpu_process_image(flgpu_ctx *ctx, const flgpu_image *decoded, uint8_t exif_orientation, const char *query_string,
                        uint32_t accept_flags, int input_format, flgpu_image *dst, flgpu_plan *plan, int *result_kind,
                        int *out_format);
/* Room dst needs for that request (0 for AS_IS); same parsing and errors, no device work. */
int flgpu_process_image_plan(const flgpu_image *decoded, uint8_t exif_orientation, const char *query_string,
                             uint32_t accept_flags, int input_format, flgpu_plan *plan, int *result_kind);

/* ---- JPEG sources (src/handler.rs:205-220: zune-jpeg through the image crate) ----------------------------------------- */
typedef struct flgpu_jpeg_info {
    uint32_t width, height;
    uint32_t components;        /* as stored: 1 (decodes to Luma8), 3 (Rgb8), 4 (CMYK / YCCK) */
    uint32_t channels;          /* of the picture the pipeline sees: 1 or 3 (0 if unsupported).  Four-component files are decoded to
                                   their raw samples and converted to Rgb8 with the configured (or embedded) CMYK profile -- the whole
                                   of convert_jpeg_color_if_needed, src/handler.rs:398-466; without any profile: FLGPU_ERR_UNSUPPORTED */
    uint32_t progressive;       /* SOF2, or any process other than baseline / extended sequential Huffman */
    uint32_t restart_interval;  /* DRI, in MCUs (0 = none) */
    uint32_t h_max, v_max;      /* largest sampling factors: 1x1 = 4:4:4, 2x1 = 4:2:2, 2x2 = 4:2:0 */
    uint32_t exif_orientation;  /* 1..8 from the APP1 Exif segment (decoder.orientation(), src/handler.rs:206); 0 = no tag */
    uint32_t supported;         /* 1 = FLGPU_IMG_JPEG_SOURCE decodes it: 8-bit Huffman-coded baseline / extended sequential (one
                                   scan or several) or progressive (SOF2), 1, 3 or 4 components, every plane at full or half
                                   resolution per direction.  0: arithmetic coding, 12-bit, lossless, hierarchical */
    uint32_t adobe_transform;   /* APP14 transform byte + 1 (0 = no Adobe segment); 4 components: 1 = CMYK, 3 = YCCK */
    uint32_t has_icc_profile;   /* an embedded ICC profile (APP2) is present */
} flgpu_jpeg_info;
/* Header inspection only (no device needed).  FLGPU_ERR_PARSE if the bytes are not a JPEG. */
int flgpu_jpeg_info_of(const uint8_t *jpeg, uint64_t n, flgpu_jpeg_info *info);
/* Decodes a supported JPEG to interleaved pixels in HOST memory at dst->data (capacity >= width*height*channels). */
int flgpu_decode_jpeg(flgpu_ctx *ctx, const uint8_t *jpeg, uint64_t n, flgpu_image *dst);
/* State::process_image for a JPEG input from the file bytes on (src/handler.rs:198-308): header + EXIF orientation,
 * query parsing, size gate, as_is, container negotiation, then decode + pixel pipeline + encode in one device pass.
 * Same outcomes as flgpu_process_image; additionally FLGPU_ERR_UNSUPPORTED for streams the device decoder does not cover
 * (the host then decodes with its own decoder and calls flgpu_process_image). */
int flgpu_process_jpeg(flgpu_ctx *ctx, const uint8_t *jpeg, uint64_t n, const char *query_string, uint32_t accept_flags,
                       flgpu_image *dst, flgpu_plan *plan, int *result_kind, int *out_format);
int flgpu_process_jpeg_plan(const uint8_t *jpeg, uint64_t n, const char *query_string, uint32_t accept_flags, flgpu_plan *plan,
                            int *result_kind);

/* Page-locked host memory for sources / results of flgpu_transform (flag them FLGPU_IMG_PINNED): a decoder that
 * writes straight into such a buffer (zune-jpeg's decode_into) saves the 6 MB staging copy of a 1080p request. */
void *flgpu_host_alloc(flgpu_ctx *ctx, uint64_t bytes);
void flgpu_host_free(flgpu_ctx *ctx, void *p);

/* n images, host memory, blocking: staged through pinned buffers, one set of launches. */
int flgpu_transform_batch(flgpu_ctx *ctx, size_t n, const flgpu_image *srcs, const flgpu_params *ps,
                          flgpu_image *dsts);

/* n images already resident in device memory; dsts[i].data are device
 * pointers.  Work is enqueued on `hip_stream` (a hipStream_t; NULL = the
 * context's own stream) and the call returns once it is enqueued; the caller
 * synchronises the stream.  ps may have n entries or, with FLGPU_BATCH_SAME_PARAMS,
 * one entry shared by all images. */
#define FLGPU_BATCH_SAME_PARAMS 1u
int flgpu_transform_batch_device(flgpu_ctx *ctx, size_t n, const flgpu_image *srcs, const flgpu_params *ps,
                                 flgpu_image *dsts, void *hip_stream, uint32_t flags);
/* What only the device knows when flgpu_transform_batch_device returns -- the length of an encoded stream
 * (FLGPU_FE_JPEG: dsts[i].bytes, 0 + FLGPU_ERR_BUFFER_TOO_SMALL if it did not fit dsts[i].capacity) and
 * FLGPU_IMG_HAS_ALPHA of the WebP front end: waits for the most recent device batch of this context and completes
 * the same dsts[] array.  The host-memory entry points do this themselves.
 * It is ALSO where a device-side failure of the batch surfaces: the matrix-pipe resample kernel bounds its waits on LDS
 * hand-offs, and a wait that expired sets the batch's device error word, which this call returns as FLGPU_ERR_DEVICE
 * (the pixels of that batch are then not valid).  Call it once per device batch before using the results, whatever the
 * front end. */
int flgpu_batch_results(flgpu_ctx *ctx, size_t n, flgpu_image *dsts);

/* How a context of n_shards devices splits a batch (pure function, no device needed): shard_of[i] = the shard, hence the
 * device devices[shard_of[i]], that image i runs on -- contiguous runs of images, balanced by algorithmic bytes
 * (W*H*C + flgpu_plan.out_bytes per image).  With a multi-device context the images handed to
 * flgpu_transform_batch_device must be resident on (or peer-accessible from) exactly these devices, and the call then
 * returns only when every shard has finished (`hip_stream` is not used).  shard_bytes (optional, n_shards entries)
 * receives the weight of each shard.  ps has n entries, or one with FLGPU_BATCH_SAME_PARAMS. */
int flgpu_plan_shards(uint32_t n_shards, size_t n, const flgpu_image *srcs, const flgpu_params *ps, uint32_t flags,
                      uint32_t *shard_of, uint64_t *shard_bytes);
/* Devices of a context: returns n_devices (1 for a single-device context) and writes up to `cap` ordinals. */
uint32_t flgpu_devices(flgpu_ctx *ctx, int32_t *devices, uint32_t cap);

/* In-place YCCK -> "CMYK with inverted K" on n_pixels x 4 host bytes: the pointwise loop of
 * convert_jpeg_color_if_needed (src/handler.rs:423-438) that precedes the lcms2 transform.  Blocking. */
int flgpu_ycck_to_cmyk(flgpu_ctx *ctx, uint8_t *raw, uint64_t n_pixels);

/* Read-only device tables (weight tables, row schedules, gamma LUTs) as one blob.
 * Multi-GPU runs build them on rank 0, broadcast the blob over RCCL/xGMI and install
 * it on the other ranks, so every GPU resamples with byte-identical tables.
 *   export: device pointer + size of this context's blob (valid until the next transform call);
 *   copy:   device-to-device copy of the blob into a caller-owned device buffer;
 *   import: overwrite this context's blob with `bytes` from a device buffer; the layout must
 *           match what this context planned locally (same requests prepared in the same order),
 *           otherwise FLGPU_ERR_INVALID_ARG. */
int flgpu_export_tables(flgpu_ctx *ctx, void **device_ptr, uint64_t *bytes);
int flgpu_copy_tables(flgpu_ctx *ctx, void *dst_device, uint64_t capacity, uint64_t *bytes);
int flgpu_import_tables(flgpu_ctx *ctx, const void *src_device, uint64_t bytes);

int flgpu_get_stats(flgpu_ctx *ctx, flgpu_stats *out);
int flgpu_reset_stats(flgpu_ctx *ctx);

/* Test and experiment switches of a context (ABI 6).  The reference shares ONE Arc<State> between all its worker threads
 * (src/main.rs:108-112), so nothing in the library may depend on the process environment after start-up: getenv races setenv in a
 * multi-threaded server, and a stray variable must not change output bytes.  The environment is read exactly once, in flgpu_create
 * (FLGPU_<KEY IN CAPITALS> seeds the switch `key`); afterwards a switch changes only through this call, atomically, for the
 * context, its queue lanes and its device shards.  Keys (csrc/fl_context.h DebugKey): "no_mfma", "force_generic", "no_wtile",
 * "wtile_blur_always", "wtile_first", "mfma_arith" (0 full width, 1 packed), "force_bands", "no_tile", "no_place4", "host_huffman",
 * "device_huffman_always", "device_huffman_min_bytes", "mfma_spin_limit", "debug_mfma", "debug_jh"; "reset" restores every default.
 * Only "no_mfma", "force_generic", "no_wtile", "wtile_first" and "mfma_arith" can change a result, by at most 1 LSB (they pick
 * another resample kernel).  Unknown key: FLGPU_ERR_INVALID_ARG. */
int flgpu_debug_set(flgpu_ctx *ctx, const char *key, int64_t value);
int flgpu_debug_get(flgpu_ctx *ctx, const char *key, int64_t *value);

/* ---- diagnostics (host only; used by the CPU test-suite) ------------------- */

/* The weight table the runtime uploads for one axis (filter 0 = Lanczos3 as used by
 * resize, 1 = Gaussian of `sigma` with support 2*sigma as used by blur): left[o],
 * count[o] for o < out_size and the packed normalised weights (image 0.25.6
 * imageops/sample.rs maths).  *total receives the number of weights; returns
 * FLGPU_ERR_BUFFER_TOO_SMALL if it exceeds weights_cap. */
int flgpu_debug_axis_table(uint32_t in_size, uint32_t out_size, int filter, float sigma, uint32_t *left,
                           uint32_t *count, float *weights, uint64_t weights_cap, uint64_t *total);

/* Row schedule of the streaming kernel for output rows [y0,y1) of a Lanczos3 axis:
 * returns 1 if the fused kernel can run it (<= 8 rows alive per source row), 0 if the
 * generic two-pass kernels are used instead; *max_live receives the peak. */
/* The host half of the JPEG decode front end alone: the coefficient blob flgpu_transform uploads for a
 * FLGPU_IMG_JPEG_SOURCE (csrc/fl_jpegdec.h: header, one u32 word per block = first coefficient << 7 | count, i16
 * coefficients in zig-zag order up to the last non-zero one).  blob == NULL: *used = capacity to provide. */
int flgpu_debug_jpeg_blob(const uint8_t *jpeg, uint64_t n, uint8_t *blob, uint64_t capacity, uint64_t *used);

int flgpu_debug_stream_schedulable(uint32_t in_size, uint32_t out_size, uint32_t y0, uint32_t y1, uint32_t *max_live);
/* Builds the matrix-pipe kernel's tables (csrc/fl_mfma.h) for a source of sw x sh pixels with `channels` interleaved bytes,
 * resized to rw x rh, kept rows [cy, cy+ch) x columns [cx, cx+cw), and checks them on the host against the plain weight
 * tables: returns 1 if the geometry fits the kernel, 0 if not.  info[0..7] = tiles, K-blocks, strips, largest number of
 * distinct horizontal operands of a strip, log2 of the horizontal weight scale, 1 if the short last tile needs the extra
 * pass, weights of the horizontal tables that are missing / wrong / doubled (must be 0), rows of the vertical tables that
 * are wrong (must be 0).  err[0] = largest |f32 weight - (sum of its two f16 terms)| of the vertical tables, err[1] = largest
 * |weight - fixed-point weight| of the horizontal ones.  Needs no device. */
int flgpu_debug_mfma_plan(uint32_t sw, uint32_t sh, uint32_t channels, uint32_t rw, uint32_t rh, uint32_t cx, uint32_t cy,
                          uint32_t cw, uint32_t ch, uint32_t info[8], double err[2]);
/* The same for one of the kernel's two arithmetics (csrc/fl_mfma.h): arith 1 = full width (what the library uses: weights as
 * three f16 terms / three byte digits; flgpu_debug_mfma_plan is this one), 0 = packed (rounds 2-3: two terms / two digits). */
int flgpu_debug_mfma_plan_arith(uint32_t sw, uint32_t sh, uint32_t channels, uint32_t rw, uint32_t rh, uint32_t cx, uint32_t cy,
                                uint32_t cw, uint32_t ch, uint32_t arith, uint32_t info[8], double err[2]);

/* Which items each persistent workgroup of a uniform matrix-pipe launch walks (csrc/fl_batch.cpp assign_items): `pictures` pictures of
 * `strips` strips and `tiles` 16-row output tiles for a launch of `workgroups` workgroups.  job_of / strip_of / tile0_of / tile1_of
 * [capacity] receive picture, strip and tile range of every item in launch order (*nitems of them: pictures left over after the
 * whole rounds are cut into row bands), lists[2 b], lists[2 b + 1] the first item and item count of workgroup b.  Needs no device. */
int flgpu_debug_assign_items(uint32_t pictures, uint32_t strips, uint32_t tiles, uint32_t workgroups, uint32_t capacity, uint32_t *job_of, uint32_t *strip_of,
                             uint32_t *tile0_of, uint32_t *tile1_of, uint32_t *lists, uint32_t *nitems);

const char *flgpu_strerror(int status);
const char *flgpu_last_error(flgpu_ctx *ctx); /* detail of the last FLGPU_ERR_DEVICE on this context */
uint32_t flgpu_abi_version(void);
/* Build provenance: "sources <first 16 hex digits of the SHA-256 over every source and header of the library>; <hipcc version>;
 * arch ...; flags ..." -- compiled in by csrc/Makefile, which also writes fanlin-rs_amd/build_info.json. */
const char *flgpu_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* FANLIN_GPU_H */
