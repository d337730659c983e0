// fanlin_gpu.hpp -- C++ host-side mirror of the reference's interface for the hot path, over the C ABI of fanlin_gpu.h.
//
// The reference is Rust; a Rust toolchain does not exist where this was built, so the host layer a maintainer would
// write in `src/gpu.rs` is provided here in C++ with the reference's own names, argument meaning
// and error behaviour:
//
//   fanlin::query::Query      src/query.rs:3-94      parse + dimensions / fill_color / quality / cropping / blur /
//                                                    grayscale / inverse / use_avif / use_webp / as_is /
//                                                    unsupported_scale_size
//   fanlin::content::Format   src/content.rs:12-48   accept_webp / webp_accepted / accept_avif / avif_accepted
//   fanlin::handler::State    src/handler.rs:14-52,185-309   new (flgpu_create), process_image (post-decode half)
//
// Errors: the reference returns Result<_, Box<dyn Error>>; here every fallible call throws fanlin::Error carrying the
// flgpu_status and the library's message, which a caller maps onto the same fallback / 500 arm (src/main.rs:185-195).
#ifndef FANLIN_GPU_HPP
#define FANLIN_GPU_HPP

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "fanlin_gpu.h"

namespace fanlin {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};

inline void check(int st, flgpu_ctx *ctx = nullptr)
{
    if (st == FLGPU_OK) return;
    std::string msg = flgpu_strerror(st);
    if (ctx) { const char *d = flgpu_last_error(ctx); if (d && *d) msg += std::string(" (") + d + ")"; }
    throw Error(st, msg);
}

namespace query {
class Query {
public:
    // axum::extract::Query<query::Query>: throws Error{FLGPU_ERR_PARSE} where axum answers 400
    static Query parse(const std::string &query_string) { Query q; check(flgpu_query_parse(query_string.c_str(), &q.raw_)); q.text_ = query_string; return q; }
    std::optional<std::pair<uint32_t, uint32_t>> dimensions() const
    {
        uint32_t w = 0, h = 0;
        if (!flgpu_query_dimensions(&raw_, &w, &h)) return std::nullopt;
        return std::make_pair(w, h);
    }
    std::tuple<uint8_t, uint8_t, uint8_t> fill_color() const { uint8_t r, g, b; flgpu_query_fill_color(&raw_, &r, &g, &b); return {r, g, b}; }
    uint8_t quality() const { return flgpu_query_quality(&raw_); }
    bool cropping() const { return flgpu_query_cropping(&raw_) != 0; }
    float blur() const { return flgpu_query_blur(&raw_); }
    bool grayscale() const { return flgpu_query_grayscale(&raw_) != 0; }
    bool inverse() const { return flgpu_query_inverse(&raw_) != 0; }
    bool use_avif() const { return flgpu_query_use_avif(&raw_) != 0; }
    bool use_webp() const { return flgpu_query_use_webp(&raw_) != 0; }
    bool as_is() const { return flgpu_query_as_is(&raw_) != 0; }
    bool unsupported_scale_size() const { return flgpu_query_unsupported_scale_size(&raw_) != 0; }
    const std::string &text() const { return text_; }
    const flgpu_query &raw() const { return raw_; }

private:
    flgpu_query raw_{};
    std::string text_;
};
} // namespace query

namespace content {
class Format {
public:
    void accept_webp() { flags_ |= FLGPU_ACCEPT_WEBP; }
    bool webp_accepted() const { return (flags_ & FLGPU_ACCEPT_WEBP) == FLGPU_ACCEPT_WEBP; }
    void accept_avif() { flags_ |= FLGPU_ACCEPT_AVIF; }
    bool avif_accepted() const { return (flags_ & FLGPU_ACCEPT_AVIF) == FLGPU_ACCEPT_AVIF; }
    uint32_t flags() const { return flags_; }

private:
    uint32_t flags_ = 0;
};
} // namespace content

namespace handler {

// DynamicImage as the decoder leaves it: tightly packed rows, 1 = Luma8, 2 = LumaA8, 3 = Rgb8, 4 = Rgba8
struct Decoded {
    const uint8_t *pixels;
    uint32_t width, height, channels;
    uint8_t orientation = 1;                        // decoder.orientation() as an EXIF code
    flgpu_input_format format = FLGPU_IN_JPEG;      // what with_guessed_format() said
};

struct Processed {
    flgpu_result_kind kind;        // AS_IS: serve the original bytes; JPEG_STREAM: body is final; WEBP_PLANES / PIXELS: host encoder
    flgpu_out_format negotiated;   // container chosen at src/handler.rs:256-261
    flgpu_plan plan;               // geometry of what `data` holds
    uint32_t flags;                // FLGPU_IMG_*
    std::vector<uint8_t> data;
};

class State {
public:
    // handler::State::new: one context per process, shared by all workers (internally synchronised)
    explicit State(int device = -1, uint32_t max_batch = 0, uint32_t flush_timeout_us = 0, uint32_t queue_lanes = 0)
    {
        flgpu_config cfg{};
        cfg.device = device; cfg.max_batch = max_batch; cfg.flush_timeout_us = flush_timeout_us; cfg.queue_lanes = queue_lanes;
        int st = 0;
        ctx_ = flgpu_create(&cfg, &st);
        if (!ctx_) check(st ? st : FLGPU_ERR_NO_DEVICE);
    }
    ~State() { if (ctx_) flgpu_destroy(ctx_); }
    State(const State &) = delete;
    State &operator=(const State &) = delete;

    // create_cmyk_to_rgb_converter (src/main.rs:74-76)
    void set_cmyk_profile(const std::vector<uint8_t> &icc) { check(flgpu_set_cmyk_profile(ctx_, icc.data(), icc.size()), ctx_); }

    // convert_jpeg_color_if_needed's colour half (src/handler.rs:421-466): n x 4 bytes in, n x 3 out
    std::vector<uint8_t> cmyk_to_rgb(const std::vector<uint8_t> &cmyk, bool ycck, const std::vector<uint8_t> *embedded_icc = nullptr)
    {
        std::vector<uint8_t> rgb(cmyk.size() / 4 * 3);
        check(flgpu_cmyk_to_rgb(ctx_, cmyk.data(), cmyk.size() / 4, rgb.data(), embedded_icc ? embedded_icc->data() : nullptr,
                                embedded_icc ? embedded_icc->size() : 0, ycck ? FLGPU_CMYK_INPUT_YCCK : 0u), ctx_);
        return rgb;
    }

    // State::process_image after the decoder (src/handler.rs:198-308): blocking, safe to call from many threads at once
    Processed process_image(const Decoded &img, const query::Query &params, const content::Format &content)
    {
        const flgpu_image src{const_cast<uint8_t *>(img.pixels), (uint64_t)img.width * img.height * img.channels, img.width, img.height, img.channels, 0, 0};
        Processed out{};
        int kind = 0, fmt = 0;
        check(flgpu_process_image_plan(&src, img.orientation, params.text().c_str(), content.flags(), img.format, &out.plan, &kind));
        out.kind = static_cast<flgpu_result_kind>(kind);
        if (out.kind == FLGPU_RESULT_AS_IS) { out.negotiated = FLGPU_OUT_KEEP; return out; }
        out.data.resize(out.plan.out_bytes);
        flgpu_image dst{out.data.data(), out.data.size(), 0, 0, 0, 0, 0};
        check(flgpu_process_image(ctx_, &src, img.orientation, params.text().c_str(), content.flags(), img.format, &dst, &out.plan, &kind, &fmt), ctx_);
        out.negotiated = static_cast<flgpu_out_format>(fmt);
        out.flags = dst.flags;
        out.data.resize(dst.bytes);
        return out;
    }

    flgpu_ctx *raw() { return ctx_; }

private:
    flgpu_ctx *ctx_ = nullptr;
};

} // namespace handler
} // namespace fanlin

#endif
