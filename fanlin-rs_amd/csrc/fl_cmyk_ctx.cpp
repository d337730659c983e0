// fl_cmyk_ctx.cpp -- CMYK / YCCK JPEG sources (reference src/handler.rs:398-493): the baked Little CMS device-link
// tables of a context, and their distribution to the devices of a multi-GPU context.
//
// the table is the one read-only LUT of the path.  It is baked ONCE on the host (40 ms of liblcms2 per
// profile, handler.rs:482), uploaded to the first device and handed to the others by one ncclBroadcast (RCCL over xGMI;
// librccl is loaded on demand so that single-GPU hosts do not need it).  Shards that share a physical GPU, or a host
// without RCCL, take a plain copy instead -- the table bytes are the same either way.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <set>

#include "fl_context.h"

using namespace fl;

namespace {

constexpr size_t kClutNodes = (size_t)kCmykGrid * kCmykGrid * kCmykGrid * kCmykGrid;
constexpr size_t kClutCacheEntries = 8; // baked tables of embedded profiles kept per context (653 KB each)

int upload_clut(flgpu_ctx *c, flgpu_ctx::Clut &t)
{
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    FL_HIP(c, t.dev.reserve(t.host.size() * sizeof(uint16_t)), "CLUT alloc");
    FL_HIP(c, hipMemcpyAsync(t.dev.p, t.host.data(), t.host.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream), "CLUT upload");
    FL_HIP(c, hipStreamSynchronize(c->stream), "CLUT sync");
    return FLGPU_OK;
}


// picks the table for one conversion: the embedded profile's if it can be baked (cached), else the default
int select_clut_impl(flgpu_ctx *c, const uint8_t *icc, uint64_t icc_len, const void **dev)
{
    if (icc && icc_len) {
        const uint64_t h = hash_bytes(icc, icc_len);
        auto it = c->cmyk_embedded.find(h);
        if (it == c->cmyk_embedded.end()) {
            std::vector<uint16_t> nodes;
            if (bake_cmyk_clut(icc, icc_len, nodes) == 0) {
                if (c->cmyk_embedded.size() >= kClutCacheEntries) {
                    // Evict the least recently used table -- but never one handed out since clut_batch_begin(): a batch selects
                    // the tables of ALL its pictures first and launches their conversions afterwards, so a table chosen for
                    // picture 1 must survive the selection for picture 9.  If every cached table is pinned that way the cache
                    // grows for this batch and is trimmed when the next one begins.
                    auto old = c->cmyk_embedded.end();
                    for (auto i2 = c->cmyk_embedded.begin(); i2 != c->cmyk_embedded.end(); ++i2)
                        if (i2->second.stamp <= c->cmyk_pin_floor && (old == c->cmyk_embedded.end() || i2->second.stamp < old->second.stamp)) old = i2;
                    if (old != c->cmyk_embedded.end()) {
                        FL_HIP(c, hipDeviceSynchronize(), "CLUT evict sync"); // earlier batches may have run on a caller's stream
                        old->second.dev.release();
                        c->cmyk_embedded.erase(old);
                    }
                }
                flgpu_ctx::Clut &t = c->cmyk_embedded[h];
                t.host.swap(nodes);
                int rc = upload_clut(c, t);
                if (rc) { c->cmyk_embedded.erase(h); return rc; }
                c->stats.cmyk_tables_baked++;
                it = c->cmyk_embedded.find(h);
            }
        }
        if (it != c->cmyk_embedded.end()) { it->second.stamp = ++c->cmyk_stamp; *dev = it->second.dev.p; return FLGPU_OK; }
        // handler.rs:449-455: an embedded profile that cannot be used falls back to the configured one
    }
    flgpu_ctx *o = c->clut_owner ? c->clut_owner : c; // a queue lane borrows the table of the context that owns it on this device
    if (!o->has_cmyk_default) { c->set_error("no CMYK profile configured"); return FLGPU_ERR_UNSUPPORTED; }
    *dev = o->cmyk_default.dev.p;
    return FLGPU_OK;
}

// ---- RCCL, loaded on demand ---------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    std::vector<void *> comms; // one per device of the context, in device-list order
};

// ncclDataType_t of one byte (nccl.h: ncclInt8 = ncclChar = 0, ncclUint8 = 1); flgpu_rccl_selftest checks on the box that a
// broadcast of n elements of this type moves exactly n bytes
constexpr int kNcclUint8 = 1;

// The five entry points the table distribution uses, resolved from the RCCL of the host (prototypes as in rccl's nccl.h:
// ncclResult_t == int, ncclComm_t == an opaque pointer).  false = no usable RCCL here.
bool rccl_load(Rccl *r)
{
    for (const char *name : {"librccl.so.1", "librccl.so"}) { r->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r->lib) break; }
    if (!r->lib) return false;
    r->CommInitAll = reinterpret_cast<int (*)(void **, int, const int *)>(dlsym(r->lib, "ncclCommInitAll"));
    r->CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(r->lib, "ncclCommDestroy"));
    r->GroupStart = reinterpret_cast<int (*)()>(dlsym(r->lib, "ncclGroupStart"));
    r->GroupEnd = reinterpret_cast<int (*)()>(dlsym(r->lib, "ncclGroupEnd"));
    r->Broadcast = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(r->lib, "ncclBroadcast"));
    if (!r->CommInitAll || !r->CommDestroy || !r->GroupStart || !r->GroupEnd || !r->Broadcast) { dlclose(r->lib); r->lib = nullptr; return false; }
    return true;
}

Rccl *rccl_for(flgpu_ctx *c)
{
    if (c->rccl) return static_cast<Rccl *>(c->rccl);
    if (c->rccl_failed) return nullptr; // (a failed dlopen / ncclCommInitAll is remembered: not retried on every table change)
    const size_t n = c->shard_ctx.size();
    std::set<int> distinct(c->devices.begin(), c->devices.end());
    if (n < 2 || distinct.size() != n) return nullptr; // RCCL wants one rank per physical GPU
    Rccl *r = new Rccl();
    if (!rccl_load(r)) { delete r; c->rccl_failed = true; return nullptr; }
    r->comms.assign(n, nullptr);
    if (r->CommInitAll(r->comms.data(), (int)n, c->devices.data()) != 0) { delete r; c->rccl_failed = true; return nullptr; } // (the library stays loaded)
    c->rccl = r;
    return r;
}

// Hands the parent's default table (host copy + device copy on devices[0]'s shard) to every shard context.
// Returns how it travelled: 2 = RCCL broadcast, 1 = copies, 0 = nothing to do.
int distribute_clut(flgpu_ctx *c, int *how)
{
    if (how) *how = 0;
    if (c->shard_ctx.empty()) return FLGPU_OK;
    const size_t bytes = c->cmyk_default.host.size() * sizeof(uint16_t);
    for (flgpu_ctx *s : c->shard_ctx) {
        std::lock_guard<std::mutex> g(s->mu);
        FL_HIP(s, hipSetDevice(s->device), "hipSetDevice");
        FL_HIP(s, hipStreamSynchronize(s->stream), "CLUT swap sync");
        s->cmyk_default.host = c->cmyk_default.host; // (host copy: flgpu_get_cmyk_clut and re-uploads)
        FL_HIP(s, s->cmyk_default.dev.reserve(bytes), "CLUT alloc");
    }
    flgpu_ctx *root = c->shard_ctx[0];
    FL_HIP(root, hipSetDevice(root->device), "hipSetDevice");
    FL_HIP(root, hipMemcpyAsync(root->cmyk_default.dev.p, c->cmyk_default.host.data(), bytes, hipMemcpyHostToDevice, root->stream), "CLUT upload");
    FL_HIP(root, hipStreamSynchronize(root->stream), "CLUT sync");
    bool sent = false;
    if (Rccl *r = rccl_for(c)) {
        bool ok = r->GroupStart() == 0;
        for (size_t k = 0; ok && k < c->shard_ctx.size(); ++k) {
            flgpu_ctx *s = c->shard_ctx[k];
            ok = hipSetDevice(s->device) == hipSuccess &&
                 r->Broadcast(s->cmyk_default.dev.p, s->cmyk_default.dev.p, bytes, kNcclUint8, /*root*/ 0, r->comms[k], s->stream) == 0;
        }
        ok = (r->GroupEnd() == 0) && ok;
        for (flgpu_ctx *s : c->shard_ctx) { (void)hipSetDevice(s->device); ok = (hipStreamSynchronize(s->stream) == hipSuccess) && ok; }
        sent = ok;
        if (sent && how) *how = 2;
    }
    if (!sent) {
        for (size_t k = 1; k < c->shard_ctx.size(); ++k) {
            flgpu_ctx *s = c->shard_ctx[k];
            FL_HIP(s, hipSetDevice(s->device), "hipSetDevice");
            hipError_t e = s->device == root->device
                               ? hipMemcpyAsync(s->cmyk_default.dev.p, root->cmyk_default.dev.p, bytes, hipMemcpyDeviceToDevice, s->stream)
                               : hipMemcpyPeerAsync(s->cmyk_default.dev.p, s->device, root->cmyk_default.dev.p, root->device, bytes, s->stream);
            if (e != hipSuccess) e = hipMemcpyAsync(s->cmyk_default.dev.p, c->cmyk_default.host.data(), bytes, hipMemcpyHostToDevice, s->stream);
            FL_HIP(s, e, "CLUT copy");
            FL_HIP(s, hipStreamSynchronize(s->stream), "CLUT sync");
        }
        if (how) *how = 1;
    }
    for (flgpu_ctx *s : c->shard_ctx) s->has_cmyk_default = true;
    return FLGPU_OK;
}

int install_default(flgpu_ctx *c)
{
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    FL_HIP(c, hipStreamSynchronize(c->stream), "CLUT swap sync");
    const int u = upload_clut(c, c->cmyk_default);
    c->has_cmyk_default = (u == FLGPU_OK);
    if (u) return u;
    int how = 0;
    const int d = distribute_clut(c, &how);
    if (d) { c->has_cmyk_default = false; return d; }
    c->cmyk_how = how;
    return FLGPU_OK;
}

} // namespace

namespace fl {

int select_clut(flgpu_ctx *c, const uint8_t *icc, uint64_t icc_len, const void **dev) { return select_clut_impl(c, icc, icc_len, dev); }

// Called before a batch (or a single conversion) selects its tables: everything selected from here on stays resident until
// the next call.  A cache that grew past its size because one batch used more distinct embedded profiles is trimmed here.
int clut_batch_begin(flgpu_ctx *c)
{
    if (c->cmyk_embedded.size() > kClutCacheEntries) {
        FL_HIP(c, hipDeviceSynchronize(), "CLUT trim sync");
        while (c->cmyk_embedded.size() > kClutCacheEntries) {
            auto old = c->cmyk_embedded.begin();
            for (auto i2 = c->cmyk_embedded.begin(); i2 != c->cmyk_embedded.end(); ++i2) if (i2->second.stamp < old->second.stamp) old = i2;
            old->second.dev.release();
            c->cmyk_embedded.erase(old);
        }
    }
    c->cmyk_pin_floor = c->cmyk_stamp;
    return FLGPU_OK;
}

void release_cmyk(flgpu_ctx *c)
{
    c->cmyk_default.dev.release();
    for (auto &kv : c->cmyk_embedded) kv.second.dev.release();
    if (c->rccl) {
        Rccl *r = static_cast<Rccl *>(c->rccl);
        for (void *comm : r->comms) if (comm) (void)r->CommDestroy(comm);
        // the library stays loaded: unloading RCCL while another context may still use it is not worth a dlclose
        delete r;
        c->rccl = nullptr;
    }
}

} // namespace fl

extern "C" {

int flgpu_set_cmyk_profile(flgpu_ctx *c, const uint8_t *icc, uint64_t n)
try {
    if (!c || !icc || !n) return FLGPU_ERR_INVALID_ARG;
    std::vector<uint16_t> nodes;
    const int rc = bake_cmyk_clut(icc, n, nodes);
    std::lock_guard<std::mutex> g(c->mu);
    if (rc == -2) { c->set_error("liblcms2.so.2 could not be loaded; bake the table elsewhere and use flgpu_set_cmyk_clut"); return FLGPU_ERR_UNSUPPORTED; }
    if (rc) { c->set_error("not a usable CMYK ICC profile"); return FLGPU_ERR_INVALID_ARG; }
    c->cmyk_default.host.swap(nodes);
    const int u = install_default(c);
    if (u == FLGPU_OK) c->stats.cmyk_tables_baked++;
    return u;
} FL_ABI_CATCH

int flgpu_set_cmyk_clut(flgpu_ctx *c, uint32_t grid, const uint16_t *rgb_nodes)
try {
    if (!c || !rgb_nodes) return FLGPU_ERR_INVALID_ARG;
    if (grid != kCmykGrid) return FLGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    c->cmyk_default.host.assign(kClutNodes * 4, 0);
    for (size_t i = 0; i < kClutNodes; ++i)
        for (int k = 0; k < 3; ++k) c->cmyk_default.host[i * 4 + k] = rgb_nodes[i * 3 + k];
    return install_default(c);
} FL_ABI_CATCH

int flgpu_get_cmyk_clut(flgpu_ctx *c, uint16_t *rgb_nodes, uint64_t capacity_entries, uint32_t *grid)
{
    if (!c || !grid) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->has_cmyk_default) { c->set_error("no CMYK profile configured"); return FLGPU_ERR_UNSUPPORTED; }
    *grid = kCmykGrid;
    if (!rgb_nodes) return FLGPU_OK;
    if (capacity_entries < kClutNodes * 3) return FLGPU_ERR_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < kClutNodes; ++i)
        for (int k = 0; k < 3; ++k) rgb_nodes[i * 3 + k] = c->cmyk_default.host[i * 4 + k];
    return FLGPU_OK;
}

int flgpu_cmyk_distribution(flgpu_ctx *c)
{
    if (!c) return 0;
    std::lock_guard<std::mutex> g(c->mu);
    return c->has_cmyk_default ? c->cmyk_how : 0;
}

int flgpu_cmyk_to_rgb_device(flgpu_ctx *c, const void *d_cmyk, void *d_rgb, uint64_t n_pixels, uint32_t flags, void *hip_stream)
try {
    if (!c || ((!d_cmyk || !d_rgb) && n_pixels)) return FLGPU_ERR_INVALID_ARG;
    if (n_pixels == 0) return FLGPU_OK;
    if (n_pixels >= (1ull << 32)) return FLGPU_ERR_UNSUPPORTED;
    if (((uintptr_t)d_cmyk & 15u) || ((uintptr_t)d_rgb & 3u)) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const void *clut = nullptr;
    const int rc = select_clut_impl(c, nullptr, 0, &clut);
    if (rc) return rc;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    FL_HIP(c, launch_cmyk_clut(d_cmyk, d_rgb, clut, kCmykGrid, n_pixels, (flags & FLGPU_CMYK_INPUT_YCCK) != 0, st), "CMYK kernel");
    c->stats.cmyk_pixels += n_pixels;
    return FLGPU_OK;
} FL_ABI_CATCH

// one contiguous range of pixels on one device context (its mutex held by the caller)
static int cmyk_range(flgpu_ctx *c, const void *clut, const uint8_t *cmyk, uint64_t n_pixels, uint8_t *rgb, uint32_t flags)
{
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const size_t n4 = (size_t)(n_pixels + 3) / 4;
    FL_HIP(c, c->d_in.reserve(n4 * 16), "device staging");
    FL_HIP(c, c->d_out.reserve(n4 * 12), "device staging");
    FL_HIP(c, hipMemcpyAsync(c->d_in.p, cmyk, (size_t)n_pixels * 4, hipMemcpyHostToDevice, c->stream), "H2D");
    FL_HIP(c, launch_cmyk_clut(c->d_in.p, c->d_out.p, clut, kCmykGrid, n_pixels, (flags & FLGPU_CMYK_INPUT_YCCK) != 0, c->stream), "CMYK kernel");
    FL_HIP(c, hipMemcpyAsync(rgb, c->d_out.p, (size_t)n_pixels * 3, hipMemcpyDeviceToHost, c->stream), "D2H");
    FL_HIP(c, hipStreamSynchronize(c->stream), "sync");
    c->stats.cmyk_pixels += n_pixels;
    return FLGPU_OK;
}

int flgpu_cmyk_to_rgb(flgpu_ctx *c, const uint8_t *cmyk, uint64_t n_pixels, uint8_t *rgb, const uint8_t *embedded_icc,
                      uint64_t icc_len, uint32_t flags)
try {
    if (!c || ((!cmyk || !rgb) && n_pixels)) return FLGPU_ERR_INVALID_ARG;
    if (n_pixels == 0) return FLGPU_OK;
    if (n_pixels >= (1ull << 30)) return FLGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    // With the configured profile on a multi-device context every device holds the broadcast table: pixels are
    // independent, so the picture is cut into one contiguous range per device (multiples of 4 pixels).
    const bool embedded = embedded_icc && icc_len;
    if (!embedded && c->shard_ctx.size() > 1 && c->has_cmyk_default && n_pixels >= (1u << 16) * c->shard_ctx.size()) {
        const size_t ns = c->shard_ctx.size();
        const uint64_t per = ((n_pixels + ns - 1) / ns + 3) & ~3ull;
        std::vector<int> rcs(ns, FLGPU_OK);
        std::vector<std::thread> ts;
        for (size_t k = 0; k < ns; ++k) {
            const uint64_t a = std::min<uint64_t>(per * k, n_pixels), b = std::min<uint64_t>(per * (k + 1), n_pixels);
            if (b == a) continue;
            ts.emplace_back([&, k, a, b] {
                flgpu_ctx *s = c->shard_ctx[k];
                std::lock_guard<std::mutex> gs(s->mu);
                rcs[k] = cmyk_range(s, s->cmyk_default.dev.p, cmyk + a * 4, b - a, rgb + a * 3, flags);
            });
        }
        for (auto &t : ts) t.join();
        for (size_t k = 0; k < ns; ++k) if (rcs[k]) { c->set_error(c->shard_ctx[k]->get_error()); return rcs[k]; }
        return FLGPU_OK;
    }
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const void *clut = nullptr;
    { const int brc = clut_batch_begin(c); if (brc) return brc; }
    const int rc = select_clut_impl(c, embedded_icc, icc_len, &clut);
    if (rc) return rc;
    return cmyk_range(c, clut, cmyk, n_pixels, rgb, flags);
} FL_ABI_CATCH

int flgpu_cmyk_bake_available(void) { return cmyk_bake_available() ? 1 : 0; }

// One GPU is enough to prove everything about the RCCL path except the wires: the library loads, the five symbols resolve,
// the hand-written prototypes and the ncclUint8 value are the ones this RCCL understands (a one-rank communicator from
// ncclCommInitAll, an out-of-place one-rank ncclBroadcast of n "uint8" elements between group calls must move exactly n bytes
// and not one more), and the communicator can be destroyed.
int flgpu_rccl_selftest(int device, uint32_t info[4])
try {
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    Rccl r;
    if (!rccl_load(&r)) return FLGPU_ERR_UNSUPPORTED; // no RCCL on this host: the distribution falls back to copies
    if (auto ver = reinterpret_cast<int (*)(int *)>(dlsym(r.lib, "ncclGetVersion"))) { int v = 0; if (ver(&v) == 0 && info) info[0] = (uint32_t)v; }
    // one way out: whatever was created below is released there (the library handle, the communicator, buffers, stream)
    void *comm = nullptr;
    uint8_t *src = nullptr, *dst = nullptr;
    hipStream_t st = nullptr;
    auto leave = [&](int code) {
        if (st) (void)hipStreamDestroy(st);
        if (src) (void)hipFree(src);
        if (dst) (void)hipFree(dst);
        if (comm) { if (r.CommDestroy(comm) != 0 && code == FLGPU_OK) code = FLGPU_ERR_DEVICE; if (info) info[3] = 1; } // created and destroyed
        if (r.lib) { dlclose(r.lib); r.lib = nullptr; }
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return leave(FLGPU_ERR_NO_DEVICE);
    const int devs[1] = {device};
    if (r.CommInitAll(&comm, 1, devs) != 0 || !comm) { comm = nullptr; return leave(FLGPU_ERR_DEVICE); }
    const size_t n = 250563, pad = 4096; // 17^4 x 3 bytes: an odd count, so a wider element type cannot pass by accident
    std::vector<uint8_t> h(n + pad), back(n + pad, 0);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 2654435761u >> 24);
    int rc = FLGPU_ERR_DEVICE;
    if (hipMalloc(&src, n + pad) == hipSuccess && hipMalloc(&dst, n + pad) == hipSuccess && hipStreamCreate(&st) == hipSuccess &&
        hipMemcpy(src, h.data(), n + pad, hipMemcpyHostToDevice) == hipSuccess && hipMemset(dst, 0xA5, n + pad) == hipSuccess) {
        bool ok = r.GroupStart() == 0;
        ok = ok && r.Broadcast(src, dst, n, kNcclUint8, /*root*/ 0, comm, st) == 0;
        ok = (r.GroupEnd() == 0) && ok;
        ok = ok && hipStreamSynchronize(st) == hipSuccess && hipMemcpy(back.data(), dst, n + pad, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) {
            size_t good = 0;
            while (good < n && back[good] == h[good]) ++good;
            bool tail_untouched = true;
            for (size_t i = n; i < n + pad; ++i) tail_untouched = tail_untouched && back[i] == 0xA5;
            if (info) { info[1] = (uint32_t)good; info[2] = tail_untouched ? 1u : 0u; }
            rc = (good == n && tail_untouched) ? FLGPU_OK : FLGPU_ERR_DEVICE;
        }
    }
    return leave(rc);
} FL_ABI_CATCH

} // extern "C"
