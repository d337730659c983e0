// fl_cmyk.cpp -- host side of the CMYK -> sRGB path (reference src/handler.rs:31-34,469-493).
//
// The reference builds `lcms2::Transform<[u8;4],[u8;3]>` (CMYK_8 -> RGB_8, Intent::Perceptual, Flags::NO_CACHE)
// once at boot from `profiles/default.icc` (src/main.rs:74-76) or per request from an embedded profile
// (src/handler.rs:446-458).  Little CMS turns such a transform into a 17^4-node device-link table and then
// interpolates it per pixel; here the table is baked once on the host -- by asking the system's liblcms2 for the
// value of the un-optimised transform at every node, exactly what cmsopt.c does -- and the per-pixel half runs
// in cmyk_clut_kernel (fl_kernels.hip).  liblcms2 is loaded with dlopen so that the library itself has no
// link-time dependency on it; without it flgpu_set_cmyk_profile reports FLGPU_ERR_UNSUPPORTED and callers can
// still hand over a table of their own with flgpu_set_cmyk_clut.
#include "fl_cmyk.h"

#include <dlfcn.h>

#include <cmath>
#include <mutex>

namespace fl {

namespace {

// lcms2.h constants (pixel formats are COLORSPACE_SH | CHANNELS_SH | BYTES_SH)
constexpr uint32_t kTypeCmyk16 = (6u << 16) | (4u << 3) | 2u;
constexpr uint32_t kTypeRgb16 = (4u << 16) | (3u << 3) | 2u;
constexpr uint32_t kIntentPerceptual = 0;
constexpr uint32_t kFlagsNoCache = 0x0040, kFlagsNoOptimize = 0x0100;

struct Lcms {
    void *so = nullptr;
    void *(*open_mem)(const void *, uint32_t) = nullptr;
    void *(*srgb)() = nullptr;
    void *(*create)(void *, uint32_t, void *, uint32_t, uint32_t, uint32_t) = nullptr;
    void (*run)(void *, const void *, void *, uint32_t) = nullptr;
    void (*del_transform)(void *) = nullptr;
    int (*close_profile)(void *) = nullptr;
    bool ok = false;
};

Lcms &lcms()
{
    static Lcms L;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"liblcms2.so.2", "liblcms2.so"}) {
            L.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (L.so) break;
        }
        if (!L.so) return;
        L.open_mem = reinterpret_cast<decltype(L.open_mem)>(dlsym(L.so, "cmsOpenProfileFromMem"));
        L.srgb = reinterpret_cast<decltype(L.srgb)>(dlsym(L.so, "cmsCreate_sRGBProfile"));
        L.create = reinterpret_cast<decltype(L.create)>(dlsym(L.so, "cmsCreateTransform"));
        L.run = reinterpret_cast<decltype(L.run)>(dlsym(L.so, "cmsDoTransform"));
        L.del_transform = reinterpret_cast<decltype(L.del_transform)>(dlsym(L.so, "cmsDeleteTransform"));
        L.close_profile = reinterpret_cast<decltype(L.close_profile)>(dlsym(L.so, "cmsCloseProfile"));
        L.ok = L.open_mem && L.srgb && L.create && L.run && L.del_transform && L.close_profile;
    });
    return L;
}

std::mutex g_bake_mu;

} // namespace

bool cmyk_bake_available() { return lcms().ok; }

int bake_cmyk_clut(const uint8_t *icc, size_t n, std::vector<uint16_t> &nodes)
{
    Lcms &L = lcms();
    if (!L.ok) return -2;
    if (!icc || n < 128 || n > 0xffffffffull) return -1;
    std::lock_guard<std::mutex> g(g_bake_mu);
    void *src = L.open_mem(icc, (uint32_t)n);
    if (!src) return -1;
    void *dst = L.srgb();
    void *t = dst ? L.create(src, kTypeCmyk16, dst, kTypeRgb16, kIntentPerceptual, kFlagsNoCache | kFlagsNoOptimize) : nullptr;
    int rc = -1;
    if (t) {
        // cmslut.c _cmsQuantizeVal: node i of an n-point axis sits at floor(i * 65535 / (n - 1) + 0.5)
        constexpr uint32_t G = kCmykGrid;
        uint16_t q[G];
        for (uint32_t i = 0; i < G; ++i) q[i] = (uint16_t)std::floor((double)i * 65535.0 / (double)(G - 1) + 0.5);
        std::vector<uint16_t> in((size_t)G * G * G * 4), out((size_t)G * G * G * 3);
        nodes.assign((size_t)G * G * G * G * 4, 0);
        for (uint32_t c = 0; c < G; ++c) {
            size_t k = 0;
            for (uint32_t m = 0; m < G; ++m)
                for (uint32_t y = 0; y < G; ++y)
                    for (uint32_t b = 0; b < G; ++b) { in[k++] = q[c]; in[k++] = q[m]; in[k++] = q[y]; in[k++] = q[b]; }
            L.run(t, in.data(), out.data(), G * G * G);
            uint16_t *o = nodes.data() + (size_t)c * G * G * G * 4;
            for (size_t i = 0; i < (size_t)G * G * G; ++i) { o[i * 4] = out[i * 3]; o[i * 4 + 1] = out[i * 3 + 1]; o[i * 4 + 2] = out[i * 3 + 2]; }
        }
        L.del_transform(t);
        // cmsopt.c FixWhiteMisalignment (run by OptimizeByResampling unless NOWHITEONWHITEFIXUP): if paper white
        // (no ink, node 0) does not come out as RGB white, that node is patched to exact white -- unless a channel
        // is so far off (> 0xf000) that WhitesAreEqual refuses to touch it
        bool patch = false;
        for (int k = 0; k < 3; ++k) {
            const int d = 0xffff - (int)nodes[k];
            if (d > 0xf000) break;
            if (d != 0) { patch = true; break; }
        }
        if (patch) nodes[0] = nodes[1] = nodes[2] = 0xffff;
        rc = 0;
    }
    if (dst) L.close_profile(dst);
    L.close_profile(src);
    return rc;
}

uint64_t hash_bytes(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n; // FNV-1a over 8-byte words, tail byte-wise
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; __builtin_memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
    for (; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

} // namespace fl
