// fl_query.cpp -- host-only request model: query::Query (reference src/query.rs),
// content::Format (src/content.rs) and the geometry decisions of
// State::process_image (src/handler.rs:224-261).  No device code.
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/fanlin_gpu.h"
#include "fl_abi.h"
#include "fl_jpeg_tables.h"
#include "fl_jpegdec.h"
#include "fl_mfma.h"
#include "fl_tables.h"

namespace {

constexpr uint8_t kDefaultColor = 32;    // query.rs:17
constexpr uint8_t kDefaultQuality = 75;  // query.rs:18
constexpr float kDefaultBlurSigma = 0.0f; // query.rs:19
constexpr uint32_t kWidthMin = 20, kWidthMax = 2000;   // query.rs:20
constexpr uint32_t kHeightMin = 20, kHeightMax = 1000; // query.rs:21

int hexval(char c)
{
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

// application/x-www-form-urlencoded decoding as form_urlencoded::parse does it
std::string url_decode(const char *s, size_t n)
{
    std::string o;
    o.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (s[i] == '+') o.push_back(' ');
        else if (s[i] == '%' && i + 2 < n && hexval(s[i + 1]) >= 0 && hexval(s[i + 2]) >= 0) {
            o.push_back((char)(hexval(s[i + 1]) * 16 + hexval(s[i + 2])));
            i += 2;
        } else o.push_back(s[i]);
    }
    return o;
}

// str::parse::<u32>() / ::<u8>(): optional '+', then decimal digits only, no overflow
bool parse_uint(const std::string &v, uint64_t max, uint64_t *out)
{
    size_t i = 0;
    if (i < v.size() && v[i] == '+') ++i;
    if (i >= v.size()) return false;
    uint64_t acc = 0;
    for (; i < v.size(); ++i) {
        if (v[i] < '0' || v[i] > '9') return false;
        acc = acc * 10 + (uint64_t)(v[i] - '0');
        if (acc > max) return false;
    }
    *out = acc;
    return true;
}

bool parse_bool(const std::string &v, uint8_t *out)
{
    if (v == "true") { *out = 1; return true; }
    if (v == "false") { *out = 0; return true; }
    return false;
}

} // namespace

extern "C" {

int flgpu_query_parse(const char *query_string, flgpu_query *out)
try {
    if (!query_string || !out) return FLGPU_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    const char *q = query_string;
    const char *qm = strchr(q, '?');
    if (qm) q = qm + 1;
    else if (strstr(q, "://")) q = q + strlen(q); // a URI without a query part
    size_t len = strcspn(q, "#");
    size_t pos = 0;
    while (pos <= len) {
        size_t end = pos;
        while (end < len && q[end] != '&') ++end;
        if (end > pos) {
            size_t eq = pos;
            while (eq < end && q[eq] != '=') ++eq;
            const std::string key = url_decode(q + pos, eq - pos);
            const std::string val = eq < end ? url_decode(q + eq + 1, end - eq - 1) : std::string();
            uint64_t u = 0;
            // serde rejects a repeated known field ("duplicate field")
#define FL_DUP(flag) do { if (out->flag) return FLGPU_ERR_PARSE; } while (0)
            if (key == "w") { FL_DUP(has_w); if (!parse_uint(val, UINT32_MAX, &u)) return FLGPU_ERR_PARSE; out->w = (uint32_t)u; out->has_w = 1; }
            else if (key == "h") { FL_DUP(has_h); if (!parse_uint(val, UINT32_MAX, &u)) return FLGPU_ERR_PARSE; out->h = (uint32_t)u; out->has_h = 1; }
            else if (key == "rgb") {
                FL_DUP(has_rgb);
                out->has_rgb = 1;
                if (val.size() < sizeof(out->rgb)) memcpy(out->rgb, val.c_str(), val.size() + 1);
                else {
                    // query.rs:35-49 reads the WHOLE string (a u8 field may carry any number of leading zeros); a value that does
                    // not fit the fixed field is reduced here to the colour it means, spelled canonically, so nothing is cut off
                    flgpu_query tmp{};
                    tmp.has_rgb = 1;
                    uint8_t c3[3] = {kDefaultColor, kDefaultColor, kDefaultColor}, t3[3];
                    int nf = 0;
                    size_t a = 0;
                    while (nf < 3) {
                        const size_t e = val.find(',', a);
                        uint64_t v = 0;
                        t3[nf++] = parse_uint(val.substr(a, e == std::string::npos ? std::string::npos : e - a), 255, &v) ? (uint8_t)v : kDefaultColor;
                        if (e == std::string::npos) break;
                        a = e + 1;
                    }
                    if (nf == 3) { c3[0] = t3[0]; c3[1] = t3[1]; c3[2] = t3[2]; }
                    snprintf(out->rgb, sizeof(out->rgb), "%u,%u,%u", c3[0], c3[1], c3[2]);
                }
            }
            else if (key == "quality") { FL_DUP(has_quality); if (!parse_uint(val, 255, &u)) return FLGPU_ERR_PARSE; out->quality = (uint8_t)u; out->has_quality = 1; }
            else if (key == "crop") { FL_DUP(has_crop); if (!parse_bool(val, &out->crop)) return FLGPU_ERR_PARSE; out->has_crop = 1; }
            else if (key == "blur") { FL_DUP(has_blur); if (!parse_uint(val, 255, &u)) return FLGPU_ERR_PARSE; out->blur = (uint8_t)u; out->has_blur = 1; }
            else if (key == "grayscale") { FL_DUP(has_grayscale); if (!parse_bool(val, &out->grayscale)) return FLGPU_ERR_PARSE; out->has_grayscale = 1; }
            else if (key == "inverse") { FL_DUP(has_inverse); if (!parse_bool(val, &out->inverse)) return FLGPU_ERR_PARSE; out->has_inverse = 1; }
            else if (key == "avif") { FL_DUP(has_avif); if (!parse_bool(val, &out->avif)) return FLGPU_ERR_PARSE; out->has_avif = 1; }
            else if (key == "webp") { FL_DUP(has_webp); if (!parse_bool(val, &out->webp)) return FLGPU_ERR_PARSE; out->has_webp = 1; }
#undef FL_DUP
            // unknown keys are ignored (no deny_unknown_fields on query::Query)
        }
        pos = end + 1;
    }
    return FLGPU_OK;
} FL_ABI_CATCH

/* query.rs:28-33 */
int flgpu_query_dimensions(const flgpu_query *q, uint32_t *w, uint32_t *h)
{
    if (q->has_w && q->has_h) { if (w) *w = q->w; if (h) *h = q->h; return 1; }
    return 0;
}

/* query.rs:35-49: first three comma fields, each unparsable one becomes 32, fewer than three -> default */
void flgpu_query_fill_color(const flgpu_query *q, uint8_t *r, uint8_t *g, uint8_t *b)
{
    uint8_t c[3] = {kDefaultColor, kDefaultColor, kDefaultColor};
    if (q->has_rgb) {
        uint8_t t[3];
        int n = 0;
        const char *s = q->rgb;
        while (n < 3) {
            const char *e = strchr(s, ',');
            std::string field = e ? std::string(s, (size_t)(e - s)) : std::string(s);
            uint64_t u;
            t[n++] = parse_uint(field, 255, &u) ? (uint8_t)u : kDefaultColor;
            if (!e) break;
            s = e + 1;
        }
        if (n == 3) { c[0] = t[0]; c[1] = t[1]; c[2] = t[2]; }
    }
    *r = c[0]; *g = c[1]; *b = c[2];
}

uint8_t flgpu_query_quality(const flgpu_query *q) { return q->has_quality ? q->quality : kDefaultQuality; } /* query.rs:51-53 */
int flgpu_query_cropping(const flgpu_query *q) { return q->has_crop && q->crop; }                        /* query.rs:55-57 */
/* query.rs:59-62: any present value is clamped into 10..=20 */
float flgpu_query_blur(const flgpu_query *q)
{
    if (!q->has_blur) return kDefaultBlurSigma;
    float v = (float)q->blur;
    return v < 10.0f ? 10.0f : (v > 20.0f ? 20.0f : v);
}
int flgpu_query_grayscale(const flgpu_query *q) { return q->has_grayscale && q->grayscale; }
int flgpu_query_inverse(const flgpu_query *q) { return q->has_inverse && q->inverse; }
int flgpu_query_use_avif(const flgpu_query *q) { return q->has_avif && q->avif; }
int flgpu_query_use_webp(const flgpu_query *q) { return q->has_webp && q->webp; }
/* query.rs:80-87 */
int flgpu_query_as_is(const flgpu_query *q)
{
    return !flgpu_query_dimensions(q, nullptr, nullptr) && flgpu_query_blur(q) == kDefaultBlurSigma &&
           !flgpu_query_grayscale(q) && !flgpu_query_inverse(q) && !flgpu_query_use_avif(q) && !flgpu_query_use_webp(q);
}
/* query.rs:89-93 */
int flgpu_query_unsupported_scale_size(const flgpu_query *q)
{
    const uint32_t w = q->has_w ? q->w : 100, h = q->has_h ? q->h : 100;
    return !(w >= kWidthMin && w <= kWidthMax) || !(h >= kHeightMin && h <= kHeightMax);
}

int flgpu_params_from_query(const flgpu_query *q, uint32_t accept_flags, int input_is_jpeg, flgpu_params *p, int *out_format)
try {
    if (!q || !p) return FLGPU_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->has_dims = (uint32_t)flgpu_query_dimensions(q, &p->w, &p->h);
    flgpu_query_fill_color(q, &p->fill_r, &p->fill_g, &p->fill_b);
    p->crop = (uint8_t)flgpu_query_cropping(q);
    p->blur_sigma = flgpu_query_blur(q);
    p->grayscale = (uint8_t)flgpu_query_grayscale(q);
    p->inverse = (uint8_t)flgpu_query_inverse(q);
    p->quality = flgpu_query_quality(q);
    /* handler.rs:256-261: webp wins over avif, each only if the client accepts it */
    int fmt = FLGPU_OUT_KEEP;
    if (flgpu_query_use_webp(q) && (accept_flags & FLGPU_ACCEPT_WEBP)) fmt = FLGPU_OUT_WEBP;
    else if (flgpu_query_use_avif(q) && (accept_flags & FLGPU_ACCEPT_AVIF)) fmt = FLGPU_OUT_AVIF;
    if (out_format) *out_format = fmt;
    /* lossy WebP (quality < 100, handler.rs:288-297) and JPEG have a device colour front end;
       lossless WebP, AVIF, PNG, ... take interleaved pixels */
    uint8_t qc = p->quality < 1 ? 1 : (p->quality > 100 ? 100 : p->quality);
    if (fmt == FLGPU_OUT_WEBP) p->front_end = (qc == 100) ? FLGPU_FE_NONE : FLGPU_FE_WEBP420;
    else if (fmt == FLGPU_OUT_KEEP && input_is_jpeg) p->front_end = input_is_jpeg == 2 ? FLGPU_FE_JPEG : FLGPU_FE_JFIF444;
    else p->front_end = FLGPU_FE_NONE;
    return FLGPU_OK;
} FL_ABI_CATCH

/* Everything State::process_image decides before it touches pixels (handler.rs:198-261). */
static int plan_request(const flgpu_image *decoded, uint8_t exif_orientation, const char *query_string, uint32_t accept_flags,
                        int input_format, flgpu_params *p, flgpu_plan *plan, int *result_kind, int *out_format)
{
    if (!decoded || !query_string || !plan || !result_kind) return FLGPU_ERR_INVALID_ARG;
    if (input_format < FLGPU_IN_OTHER || input_format > FLGPU_IN_GIF_FRAME || exif_orientation > 8) return FLGPU_ERR_INVALID_ARG;
    flgpu_query q;
    int rc = flgpu_query_parse(query_string, &q);
    if (rc) return rc;
    if (flgpu_query_unsupported_scale_size(&q)) return FLGPU_ERR_PARSE;            /* main.rs:134-138: 400 before the handler runs */
    memset(plan, 0, sizeof(*plan));
    if (flgpu_query_as_is(&q)) { *result_kind = FLGPU_RESULT_AS_IS; if (out_format) *out_format = FLGPU_OUT_KEEP; return FLGPU_OK; }
    int fmt = FLGPU_OUT_KEEP;
    rc = flgpu_params_from_query(&q, accept_flags, input_format == FLGPU_IN_JPEG ? 2 : 0, p, &fmt);
    if (rc) return rc;
    if (input_format == FLGPU_IN_GIF_FRAME) {
        /* process_gif (handler.rs:311-366): Nearest, no blur, no orientation, always re-encoded as GIF by the host */
        p->filter = FLGPU_FILTER_NEAREST;
        p->blur_sigma = 0.0f;
        p->front_end = FLGPU_FE_NONE;
        fmt = FLGPU_OUT_KEEP;
    } else {
        p->orientation = exif_orientation;
        /* a WebP source that stays WebP goes through the same arm as a negotiated one (handler.rs:286-305) */
        if (fmt == FLGPU_OUT_KEEP && input_format == FLGPU_IN_WEBP) {
            const uint8_t qc = p->quality < 1 ? 1 : (p->quality > 100 ? 100 : p->quality);
            p->front_end = qc == 100 ? FLGPU_FE_NONE : FLGPU_FE_WEBP420;
        }
    }
    if (out_format) *out_format = fmt;
    *result_kind = p->front_end == FLGPU_FE_JPEG ? FLGPU_RESULT_JPEG_STREAM : p->front_end == FLGPU_FE_WEBP420 ? FLGPU_RESULT_WEBP_PLANES : FLGPU_RESULT_PIXELS;
    return flgpu_plan_output(p, decoded->width, decoded->height, decoded->channels, plan);
}

int flgpu_process_image_plan(const flgpu_image *decoded, uint8_t exif_orientation, const char *query_string, uint32_t accept_flags,
                             int input_format, flgpu_plan *plan, int *result_kind)
try {
    flgpu_params p;
    return plan_request(decoded, exif_orientation, query_string, accept_flags, input_format, &p, plan, result_kind, nullptr);
} FL_ABI_CATCH

int flgpu_process_image(flgpu_ctx *ctx, const flgpu_image *decoded, uint8_t exif_orientation, const char *query_string,
                        uint32_t accept_flags, int input_format, flgpu_image *dst, flgpu_plan *plan, int *result_kind, int *out_format)
try {
    if (!ctx || !dst) return FLGPU_ERR_INVALID_ARG;
    flgpu_params p;
    flgpu_plan local;
    int kind = 0;
    int rc = plan_request(decoded, exif_orientation, query_string, accept_flags, input_format, &p, plan ? plan : &local, &kind, out_format);
    if (result_kind) *result_kind = kind;
    if (rc || kind == FLGPU_RESULT_AS_IS) return rc;
    return flgpu_transform(ctx, decoded, &p, dst);
} FL_ABI_CATCH

int flgpu_jpeg_info_of(const uint8_t *jpeg, uint64_t n, flgpu_jpeg_info *info)
try {
    if (!jpeg || !info) return FLGPU_ERR_INVALID_ARG;
    fl::JpegInfo I;
    if (fl::jpeg_parse_info(jpeg, (size_t)n, I) != 0) return FLGPU_ERR_PARSE;
    memset(info, 0, sizeof(*info));
    info->width = I.width; info->height = I.height; info->components = I.components;
    info->channels = I.supported ? (I.components == 1 ? 1u : 3u) : 0u;
    info->adobe_transform = (uint32_t)(I.adobe_transform + 1);
    info->has_icc_profile = I.icc.empty() ? 0u : 1u;
    info->progressive = I.progressive; info->restart_interval = I.restart_interval;
    info->h_max = I.hmax; info->v_max = I.vmax;
    info->exif_orientation = I.exif_orientation; info->supported = I.supported;
    return FLGPU_OK;
} FL_ABI_CATCH

static int plan_jpeg(const uint8_t *jpeg, uint64_t n, const char *query_string, uint32_t accept_flags, flgpu_image *src, uint8_t *orientation,
                     flgpu_params *p, flgpu_plan *plan, int *kind, int *out_format)
{
    flgpu_jpeg_info info;
    int rc = flgpu_jpeg_info_of(jpeg, n, &info);
    if (rc) return rc;
    memset(src, 0, sizeof(*src));
    src->data = const_cast<uint8_t *>(jpeg);
    src->capacity = n;
    src->width = info.width; src->height = info.height; src->channels = info.components == 1 ? 1u : 3u;
    src->flags = FLGPU_IMG_JPEG_SOURCE;
    *orientation = (uint8_t)(info.exif_orientation ? info.exif_orientation : 1u);
    rc = plan_request(src, *orientation, query_string, accept_flags, FLGPU_IN_JPEG, p, plan, kind, out_format);
    if (rc) return rc;
    if (*kind != FLGPU_RESULT_AS_IS && !info.supported) return FLGPU_ERR_UNSUPPORTED; /* as_is never decodes (handler.rs:202-204) */
    return FLGPU_OK;
}

int flgpu_process_jpeg_plan(const uint8_t *jpeg, uint64_t n, const char *query_string, uint32_t accept_flags, flgpu_plan *plan, int *result_kind)
try {
    flgpu_image src;
    flgpu_params p;
    uint8_t o = 1;
    return plan_jpeg(jpeg, n, query_string, accept_flags, &src, &o, &p, plan, result_kind, nullptr);
} FL_ABI_CATCH

int flgpu_process_jpeg(flgpu_ctx *ctx, const uint8_t *jpeg, uint64_t n, const char *query_string, uint32_t accept_flags,
                       flgpu_image *dst, flgpu_plan *plan, int *result_kind, int *out_format)
try {
    if (!ctx || !dst) return FLGPU_ERR_INVALID_ARG;
    flgpu_image src;
    flgpu_params p;
    flgpu_plan local;
    uint8_t o = 1;
    int kind = 0;
    int rc = plan_jpeg(jpeg, n, query_string, accept_flags, &src, &o, &p, plan ? plan : &local, &kind, out_format);
    if (result_kind) *result_kind = kind;
    if (rc || kind == FLGPU_RESULT_AS_IS) return rc;
    return flgpu_transform(ctx, &src, &p, dst);
} FL_ABI_CATCH

int flgpu_debug_jpeg_blob(const uint8_t *jpeg, uint64_t n, uint8_t *blob, uint64_t capacity, uint64_t *used)
try {
    if (!jpeg || !used) return FLGPU_ERR_INVALID_ARG;
    fl::JpegInfo I;
    if (fl::jpeg_parse_info(jpeg, (size_t)n, I) != 0) return FLGPU_ERR_PARSE;
    if (!I.supported) return FLGPU_ERR_UNSUPPORTED;
    *used = fl::jpeg_blob_bound(I);
    if (!blob) return FLGPU_OK;
    if (capacity < *used) return FLGPU_ERR_BUFFER_TOO_SMALL;
    size_t u = 0;
    const int rc = fl::jpeg_entropy_decode(jpeg, (size_t)n, blob, (size_t)capacity, &u);
    *used = u;
    return rc == 0 ? FLGPU_OK : rc == -2 ? FLGPU_ERR_UNSUPPORTED : FLGPU_ERR_INVALID_ARG;
} FL_ABI_CATCH

int flgpu_decode_jpeg(flgpu_ctx *ctx, const uint8_t *jpeg, uint64_t n, flgpu_image *dst)
try {
    if (!ctx || !dst || !dst->data) return FLGPU_ERR_INVALID_ARG;
    flgpu_jpeg_info info;
    int rc = flgpu_jpeg_info_of(jpeg, n, &info);
    if (rc) return rc;
    if (!info.supported) return FLGPU_ERR_UNSUPPORTED;
    flgpu_image src;
    memset(&src, 0, sizeof(src));
    src.data = const_cast<uint8_t *>(jpeg); src.capacity = n;
    src.width = info.width; src.height = info.height; src.channels = info.channels; src.flags = FLGPU_IMG_JPEG_SOURCE;
    flgpu_params p;
    memset(&p, 0, sizeof(p)); /* no dimensions, no operation: the pipeline is the identity, the result the decoded picture */
    return flgpu_transform(ctx, &src, &p, dst);
} FL_ABI_CATCH

int flgpu_plan_output(const flgpu_params *p, uint32_t sw, uint32_t sh, uint32_t sc, flgpu_plan *plan)
{
    if (!p || !plan) return FLGPU_ERR_INVALID_ARG;
    if (sw == 0 || sh == 0 || sc < 1 || sc > 4) return FLGPU_ERR_INVALID_ARG;
    if ((uint64_t)sw * sh * sc >= (1ull << 31)) return FLGPU_ERR_UNSUPPORTED;
    if (p->front_end > FLGPU_FE_JPEG || p->orientation > 8 || p->filter > FLGPU_FILTER_NEAREST) return FLGPU_ERR_INVALID_ARG;
    memset(plan, 0, sizeof(*plan));
    /* handler.rs:221-223: EXIF orientations 5..8 contain a quarter turn: width and height swap */
    if (p->orientation >= 5) { const uint32_t t = sw; sw = sh; sh = t; }
    plan->src_w = sw;
    plan->src_h = sh;
    /* handler.rs:224-228 */
    plan->mid_c = p->grayscale ? (sc == 3 ? 1u : sc == 4 ? 2u : sc) : sc;
    uint32_t cw = sw, chh = sh; /* current image size */
    plan->resized_w = sw;
    plan->resized_h = sh;
    if (p->has_dims) {
        if (p->w == 0 || p->h == 0) return FLGPU_ERR_INVALID_ARG;
        if (p->w != sw || p->h != sh) {
            uint32_t w2, h2;
            fl::resize_dimensions(sw, sh, p->w, p->h, p->crop != 0, w2, h2);
            if ((uint64_t)w2 * h2 * 4 >= (1ull << 31)) return FLGPU_ERR_UNSUPPORTED;
            plan->resized_w = w2;
            plan->resized_h = h2;
            plan->resampled = (w2 != sw || h2 != sh) ? 1u : 0u; /* imageops::resize copies when the size is unchanged */
            cw = w2;
            chh = h2;
            if (p->crop) {
                /* DynamicImage::resize_to_fill + imageops::crop_dimms */
                uint32_t x = 0, y = 0;
                if ((uint64_t)p->w * h2 > (uint64_t)w2 * p->h) y = h2 >= p->h ? (h2 - p->h) / 2 : 0;
                else x = w2 >= p->w ? (w2 - p->w) / 2 : 0;
                x = x < w2 ? x : w2;
                y = y < h2 ? y : h2;
                plan->crop_x = x;
                plan->crop_y = y;
                cw = p->w < w2 - x ? p->w : w2 - x;
                chh = p->h < h2 - y ? p->h : h2 - y;
            }
        }
        if (p->w > cw || p->h > chh) {
            /* handler.rs:238-248 */
            plan->letterboxed = 1;
            plan->place_x = (p->w > cw ? p->w - cw : cw - p->w) / 2;
            plan->place_y = (p->h > chh ? p->h - chh : chh - p->h) / 2;
            plan->out_w = p->w;
            plan->out_h = p->h;
            plan->out_c = 4;
        }
    }
    if (!plan->letterboxed) {
        plan->out_w = cw;
        plan->out_h = chh;
        plan->out_c = plan->mid_c;
    }
    plan->pixel_bytes = (uint64_t)plan->out_w * plan->out_h * plan->out_c;
    switch (p->front_end) {
    case FLGPU_FE_JFIF444:
        plan->plane_w = (plan->out_w + 7u) & ~7u;
        plan->plane_h = (plan->out_h + 7u) & ~7u;
        plan->chroma_w = plan->plane_w;
        plan->chroma_h = plan->plane_h;
        plan->out_bytes = 3ull * plan->plane_w * plan->plane_h;
        break;
    case FLGPU_FE_JPEG:
        // out_bytes: a planning bound of the stream, not its length: header + one byte per sample (quantised photographs
        // stay far below).  max_out_bytes: the worst case of the format, 623 + 2 + 2 * 208 bytes per 8x8 block and
        // component: the library's own staging uses it, so a request fails with FLGPU_ERR_BUFFER_TOO_SMALL only if the
        // finished stream really exceeds the caller's dst->capacity.
        plan->plane_w = (plan->out_w + 7u) & ~7u;
        plan->plane_h = (plan->out_h + 7u) & ~7u;
        plan->chroma_w = plan->plane_w;
        plan->chroma_h = plan->plane_h;
        plan->out_bytes = 3ull * plan->plane_w * plan->plane_h + 1024ull;
        plan->max_out_bytes = 1024ull + 2ull * fl::kJpegMaxUnitBytes * 3ull * (plan->plane_w / 8u) * (plan->plane_h / 8u);
        break;
    case FLGPU_FE_WEBP420:
        plan->plane_w = plan->out_w;
        plan->plane_h = plan->out_h;
        plan->chroma_w = (plan->out_w + 1u) >> 1;
        plan->chroma_h = (plan->out_h + 1u) >> 1;
        /* Y | U | V | A: the alpha plane (w x h) is always written; it matters when FLGPU_IMG_HAS_ALPHA comes back */
        plan->out_bytes = 2ull * plan->plane_w * plan->plane_h + 2ull * plan->chroma_w * plan->chroma_h;
        break;
    default:
        plan->out_bytes = plan->pixel_bytes;
        break;
    }
    if (plan->max_out_bytes < plan->out_bytes) plan->max_out_bytes = plan->out_bytes;
    return FLGPU_OK;
}

int flgpu_debug_axis_table(uint32_t in_size, uint32_t out_size, int filter, float sigma, uint32_t *left, uint32_t *count,
                           float *weights, uint64_t weights_cap, uint64_t *total)
{
    if (!in_size || !out_size || !left || !count || !weights || !total || filter < 0 || filter > 1) return FLGPU_ERR_INVALID_ARG;
    fl::HostAxis a;
    fl::build_axis(in_size, out_size, filter ? fl::FILTER_GAUSSIAN : fl::FILTER_LANCZOS3, sigma, a);
    *total = a.weights.size();
    if (a.weights.size() > weights_cap) return FLGPU_ERR_BUFFER_TOO_SMALL;
    memcpy(left, a.left.data(), out_size * 4);
    memcpy(count, a.count.data(), out_size * 4);
    memcpy(weights, a.weights.data(), a.weights.size() * 4);
    return FLGPU_OK;
}

int flgpu_debug_stream_schedulable(uint32_t in_size, uint32_t out_size, uint32_t y0, uint32_t y1, uint32_t *max_live)
{
    if (!in_size || !out_size || y0 >= y1 || y1 > out_size) return 0;
    fl::HostAxis a;
    fl::build_axis(in_size, out_size, fl::FILTER_LANCZOS3, 0.0f, a);
    uint32_t r0, r1, peak = 0;
    std::vector<fl::RowSched> sched;
    const bool ok = fl::build_row_sched(a, y0, y1, fl::NACC, 1, r0, r1, sched);
    if (ok) {
        // every tap must appear exactly once and every output must be emitted exactly once
        uint64_t taps = 0, emits = 0, want = 0;
        for (auto &e : sched) { taps += (uint64_t)__builtin_popcount(e.live); emits += (uint64_t)__builtin_popcount(e.emit); peak = std::max<uint32_t>(peak, (uint32_t)__builtin_popcount(e.live)); }
        // (with block = 1 every row is its own block, so the folded emit masks are the per-row ones)
        for (uint32_t o = y0; o < y1; ++o) want += a.count[o];
        if (taps != want || emits != y1 - y0) return 0;
    } else {
        // report the true peak so callers can see why it was refused
        std::vector<uint32_t> live(in_size + 1, 0);
        for (uint32_t o = y0; o < y1; ++o) for (uint32_t i = 0; i < a.count[o]; ++i) peak = std::max(peak, ++live[a.left[o] + i]);
    }
    if (max_live) *max_live = peak;
    return ok ? 1 : 0;
}

const char *flgpu_strerror(int status)
{
    switch (status) {
    case FLGPU_OK: return "ok";
    case FLGPU_ERR_INVALID_ARG: return "invalid argument";
    case FLGPU_ERR_UNSUPPORTED: return "request not supported by the device path";
    case FLGPU_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case FLGPU_ERR_OOM: return "out of memory";
    case FLGPU_ERR_DEVICE: return "HIP runtime error";
    case FLGPU_ERR_PARSE: return "malformed query string";
    case FLGPU_ERR_BUFFER_TOO_SMALL: return "destination buffer too small";
    case FLGPU_ERR_SHUTDOWN: return "context is shutting down";
    }
    return "unknown status";
}

uint32_t flgpu_abi_version(void) { return FLGPU_ABI_VERSION; }

} // extern "C"

// Host-side self-check of the matrix-pipe kernel's tables: everything the kernel will read is decoded again here, the way the
// kernel uses it, and compared with the axis tables it was built from (tests/test_mfma_tables.py, no device needed).
int flgpu_debug_mfma_plan(uint32_t sw, uint32_t sh, uint32_t channels, uint32_t rw, uint32_t rh, uint32_t cx, uint32_t cy,
                          uint32_t cw, uint32_t ch, uint32_t info[8], double err[2])
{
    return flgpu_debug_mfma_plan_arith(sw, sh, channels, rw, rh, cx, cy, cw, ch, 1u, info, err);
}

int flgpu_debug_mfma_plan_arith(uint32_t sw, uint32_t sh, uint32_t channels, uint32_t rw, uint32_t rh, uint32_t cx, uint32_t cy,
                                uint32_t cw, uint32_t ch, uint32_t arith, uint32_t info[8], double err[2])
{
    using namespace fl;
    if (!sw || !sh || !rw || !rh || !info || !err || arith > 1u) return 0;
    const uint32_t NTERM = arith ? 3u : 2u, NDIG = arith ? 3u : 2u, CE = 1u + NDIG;
    const double vscale = ldexp(1.0, (int)(arith ? kMfmaVScaleLog2Full : kMfmaVScaleLog2));
    HostAxis v, h;
    build_axis(sh, rh, FILTER_LANCZOS3, 0.0f, v);
    build_axis(sw, rw, FILTER_LANCZOS3, 0.0f, h);
    HostMfmaPlan p;
    choose_mfma_plan(v, h, channels, cx, cy, cw, ch, p, arith ? MFMA_ARITH_FULL : MFMA_ARITH_PACKED);
    for (int k = 0; k < 8; ++k) info[k] = 0;
    err[0] = err[1] = 0.0;
    if (!p.ok) return 0;
    auto f16 = [](uint32_t hbits) -> double {
        const int sgn = (hbits & 0x8000u) ? -1 : 1, e = (hbits >> 10) & 31, m = hbits & 0x3ff;
        return e == 0 ? sgn * ldexp((double)m, -24) : sgn * ldexp((double)(m | 0x400), e - 25);
    };
    info[0] = p.ntiles; info[1] = p.nkb; info[2] = (uint32_t)p.strips.size(); info[5] = p.tail;
    // ---- vertical: walk the K-blocks as the kernel does (older live tile = set 0, the set moves down when a tile completes)
    uint32_t bad_rows = 0;
    {
        std::vector<std::vector<double>> got(p.rows, std::vector<double>(sh, 0.0)); // [kept output row][source row]
        int cur[2] = {-1, -1};                                                        // tile held by each accumulator set
        uint32_t next_tile = 0;
        for (uint32_t s = 0; s <= p.nkb; ++s) {                                       // the last pass is the all-zero K-block
            if (s == p.nkb && !p.tail) break;
            for (int set = 0; set < 2; ++set) {
                bool any = false;
                for (uint32_t k = 0; k < NTERM * 64u * 4u && !any; ++k) any = p.vw[(((size_t)s * 2 + set) * NTERM) * 64 * 4 + k] != 0u;
                if (!any) continue;
                if (cur[set] < 0) { cur[set] = (int)next_tile++; }
                for (uint32_t lane = 0; lane < 64; ++lane)
                    for (uint32_t jj = 0; jj < 8; ++jj) {
                        const size_t base = ((((size_t)s * 2 + set) * NTERM) * 64 + lane) * 4 + jj / 2;
                        double w = 0.0;
                        for (uint32_t t = NTERM; t-- > 0;) w += f16((p.vw[base + (size_t)t * 256] >> (16 * (jj & 1))) & 0xffffu); // (small terms first: exact in double either way)
                        w /= vscale;
                        const uint32_t r = kMfmaKRows * s + 8 * (lane >> 4) + jj, row = 16u * (uint32_t)cur[set] + (lane & 15u);
                        if (w != 0.0) { if (row >= p.rows || r >= sh) ++bad_rows; else got[row][r] += w; }
                    }
            }
            const uint32_t ft = p.vmeta[s] & 0xffffu;
            if (ft != 0xffffu) { if (cur[0] != (int)ft) ++bad_rows; cur[0] = cur[1]; cur[1] = -1; }
        }
        for (uint32_t row = 0; row < p.rows; ++row) {
            const uint32_t oy = cy + row;
            for (uint32_t r = 0; r < sh; ++r) {
                const bool in = r >= v.left[oy] && r < v.left[oy] + v.count[oy];
                const double want = in ? (double)v.weights[v.woff[oy] + r - v.left[oy]] : 0.0;
                if (!in && got[row][r] != 0.0) ++bad_rows;
                err[0] = std::max(err[0], fabs(got[row][r] - want));
            }
        }
    }
    info[7] = bad_rows;
    // ---- horizontal: decode every operand through the tile tables into a dense [output][strip byte column] matrix
    uint32_t bad_h = 0;
    for (auto &S : p.strips) {
        const uint32_t nout = S.hdr.nout;
        info[3] = std::max(info[3], S.hdr.n_ops); info[4] = S.hdr.hs;
        std::vector<int32_t> dense((size_t)nout * kMfmaStripBytes, 0);
        std::vector<uint8_t> hits((size_t)nout * kMfmaStripBytes, 0);
        for (uint32_t w = 0; w < kMfmaWaves; ++w)
            for (uint32_t c = 0; c < 4; ++c)
                for (uint32_t t = 0; t < 3; ++t) {
                    const int32_t *e = &S.ctab[((w * 4 + c) * 3 + t) * CE];
                    for (uint32_t lane = 0; lane < 64; ++lane)
                        for (uint32_t b = 0; b < 16; ++b) {
                            int32_t q = 0; // digits, high one first
                            for (uint32_t d = 0; d < NDIG; ++d) q = 256 * q + (int32_t)reinterpret_cast<const int8_t *>(S.ops.data() + (size_t)e[1 + d] * 256)[lane * 16 + b];
                            if (q == 0) continue;
                            const int64_t o = (int64_t)e[0] + (lane & 15u);
                            const uint32_t col = kMfmaWaveCols * w + 64u * c + 16u * (b >> 2) + 4u * (lane >> 4) + (b & 3u);
                            if (o < 0 || o >= (int64_t)nout) { ++bad_h; continue; }
                            dense[(size_t)o * kMfmaStripBytes + col] += q;
                            if (++hits[(size_t)o * kMfmaStripBytes + col] > 1) ++bad_h;
                        }
                }
        for (uint32_t o = 0; o < nout; ++o) {
            const uint32_t x = S.hdr.x0 + o / channels, chn = o % channels;
            int64_t sum = 0;
            for (uint32_t col = 0; col < kMfmaStripBytes; ++col) {
                const uint32_t ab = S.hdr.byte0 + col, px = ab / channels;
                const bool in = ab < channels * sw && ab % channels == chn && px >= h.left[x] && px < h.left[x] + h.count[x];
                const int32_t q = dense[(size_t)o * kMfmaStripBytes + col];
                sum += q;
                if (!in) { if (q != 0) ++bad_h; continue; }
                err[1] = std::max(err[1], fabs(ldexp((double)q, -(int)S.hdr.hs) - (double)h.weights[h.woff[x] + px - h.left[x]]));
            }
            if (sum != ((int64_t)1 << S.hdr.hs)) ++bad_h; // every tap present exactly once, and the fixed-point weights sum to exactly 1
        }
    }
    info[6] = bad_h;
    return 1;
}
