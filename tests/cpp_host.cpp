// The C++ host-side mirror (include/fanlin_gpu.hpp) in use: written the way the reference's own tests read
// (src/query.rs test table, src/main.rs test_generic_handler).   cpp_host [gpu]
#include <cstdio>
#include <cstring>

#include "fanlin_gpu.hpp"

using namespace fanlin;

#define EXPECT(...) do { if (!(__VA_ARGS__)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #__VA_ARGS__); return 1; } } while (0)

int main(int argc, char **argv)
{
    // query.rs tests, a few rows
    auto q = query::Query::parse("w=300&h=200&rgb=1,2,3&quality=90&crop=true&blur=5&grayscale=true");
    EXPECT(q.dimensions() && q.dimensions()->first == 300 && q.dimensions()->second == 200);
    EXPECT(q.fill_color() == std::tuple<uint8_t, uint8_t, uint8_t>(1, 2, 3));
    EXPECT(q.quality() == 90 && q.cropping() && q.blur() == 10.0f && q.grayscale() && !q.inverse() && !q.as_is());
    EXPECT(query::Query::parse("rgb=9,9").fill_color() == std::tuple<uint8_t, uint8_t, uint8_t>(32, 32, 32));
    EXPECT(query::Query::parse("").as_is() && query::Query::parse("w=10&h=10").unsupported_scale_size());
    bool threw = false;
    try { query::Query::parse("w=abc"); } catch (const Error &e) { threw = e.status == FLGPU_ERR_PARSE; }
    EXPECT(threw);
    content::Format f;
    EXPECT(!f.webp_accepted());
    f.accept_webp(); f.accept_avif();
    EXPECT(f.webp_accepted() && f.avif_accepted());
    if (argc < 2 || std::strcmp(argv[1], "gpu") != 0) { std::puts("host ok"); return 0; }

    // test_generic_handler, image rows: status / Content-Type become result kind / negotiated container
    handler::State state;
    std::vector<uint8_t> px(512 * 512 * 3);
    uint32_t s = 7;
    for (auto &b : px) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    handler::Decoded img{px.data(), 512, 512, 3};
    auto r = state.process_image(img, query::Query::parse(""), f);
    EXPECT(r.kind == FLGPU_RESULT_AS_IS);
    r = state.process_image(img, query::Query::parse("w=300&h=200"), f);
    EXPECT(r.kind == FLGPU_RESULT_JPEG_STREAM && r.data.size() > 700 && r.data[0] == 0xFF && r.data[1] == 0xD8 &&
           r.data[r.data.size() - 2] == 0xFF && r.data[r.data.size() - 1] == 0xD9);
    r = state.process_image(img, query::Query::parse("w=300&h=200&webp=true"), f);
    EXPECT(r.kind == FLGPU_RESULT_WEBP_PLANES && r.negotiated == FLGPU_OUT_WEBP && r.data.size() == 2 * 300 * 200 + 2 * 150 * 100);
    r = state.process_image(img, query::Query::parse("w=300&h=200&avif=true"), f);
    EXPECT(r.kind == FLGPU_RESULT_PIXELS && r.negotiated == FLGPU_OUT_AVIF && r.data.size() == 300 * 200 * 4);
    threw = false;
    try { state.process_image(img, query::Query::parse("w=9999&h=9999"), f); } catch (const Error &e) { threw = e.status == FLGPU_ERR_PARSE; }
    EXPECT(threw);
    std::puts("gpu ok");
    return 0;
}
