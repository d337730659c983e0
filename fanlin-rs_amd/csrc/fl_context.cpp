// fl_context.cpp -- host runtime behind the C ABI: device context, table
// arena + caches, batch planner/launcher, host-memory staging and the
// persistent request-batching queue.
//
// There is deliberately NO CPU fallback in this file: if HIP is unavailable
// or a launch fails the caller gets an error code (reference behaviour on any
// Err from process_image is the fallback image / 500, src/main.rs:185-195).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/fanlin_gpu.h"
#include "fl_cmyk.h"
#include "fl_jpeg_tables.h"
#include "fl_kernels.h"
#include "fl_tables.h"

using namespace fl;

namespace {

struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = std::max(bytes, (size_t)1 << 20);
        want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = std::max(bytes, (size_t)1 << 16);
        want = (want + 4095) & ~(size_t)4095;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// descriptor staging slot: pinned host copy + device copy, guarded by an event
struct DescSlot {
    PinnedBuf host;
    DeviceBuf dev;
    hipEvent_t done = nullptr;
    bool busy = false;
};

typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t> AxisKey; // in, out, filter, sigma bits

struct StreamPlanKey {
    AxisKey v, h;
    uint32_t cx, cy, cw, ch, nbands;
    uint32_t mono; // single-channel rows keep 4-byte partial sums in LDS: a different LDS footprint for the same geometry
    bool operator<(const StreamPlanKey &o) const
    {
        return std::tie(v, h, cx, cy, cw, ch, nbands, mono) < std::tie(o.v, o.h, o.cx, o.cy, o.cw, o.ch, o.nbands, o.mono);
    }
};

struct StreamPlan {
    bool ok = false;
    uint32_t nacc = NACC;
    std::vector<StreamItem> items; // job field unset
    size_t lds_bytes = 0;
};

struct PinBlock {
    void *p = nullptr;
    size_t cap = 0;
};

struct Request {
    const flgpu_image *src;
    const flgpu_params *p;
    flgpu_image *dst;
    PinBlock in, out;      // pinned staging filled / drained by the CALLER thread (parallel memcpy)
    uint64_t src_bytes = 0, out_bytes = 0;
    int status = 0;
    bool done = false;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

} // namespace

struct flgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    flgpu_config cfg{};
    std::mutex mu; // planning + launching is serialised per context

    // read-only table arena
    std::vector<uint32_t> h_arena;
    uint32_t *d_arena = nullptr;
    size_t arena_cap_words = 0, arena_uploaded = 0;
    std::map<AxisKey, uint32_t> axis_off;
    std::map<AxisKey, HostAxis> axis_host;
    std::map<StreamPlanKey, StreamPlan> stream_plans;
    std::map<std::tuple<AxisKey, AxisKey, uint32_t>, uint32_t> blur_plans; // blur kernel table blocks per (vertical, horizontal) Gaussian axis
    uint32_t gamma_off = 0;

    DescSlot slots[4];
    int next_slot = 0;
    DeviceBuf d_mid, d_tmp_a, d_tmp_b, d_tmp_o, d_status;
    DeviceBuf d_in, d_out;
    DeviceBuf d_jpeg_coef, d_jpeg_off, d_jpeg_raw; // JPEG encode scratch (fl_jpeg.hip): block meta words, bit offsets, AC bits
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> jpeg_tables; // (w, h, quality) -> arena offset of header + q tables
    // per-image result words of the most recent device batch: [2i] flags (bit 0: non-opaque alpha seen by the WebP front
    // end, FL_JPEG_RESULT_OVERFLOW), [2i + 1] bytes of an encoded stream
    size_t last_n = 0;
    bool last_has_results = false;
    std::vector<uint8_t> last_fe;
    PinnedBuf h_results;
    PinnedBuf h_stage_in, h_stage_out;
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr;

    flgpu_stats stats{};
    struct Pending { hipEvent_t a, b; int kind; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;

    std::string last_error;

    // CMYK -> sRGB device-link tables (kCmykGrid^4 nodes of 4 x u16): the boot-time default (main.rs:74-76) and a
    // small cache of tables baked from embedded profiles, keyed by a hash of the profile bytes (handler.rs:446-458
    // rebuilds the lcms2 transform, 40 ms, on every such request)
    struct Clut { DeviceBuf dev; std::vector<uint16_t> host; uint64_t stamp = 0; };
    Clut cmyk_default;
    bool has_cmyk_default = false;
    std::map<uint64_t, Clut> cmyk_embedded;
    uint64_t cmyk_stamp = 0;

    // pinned staging blocks recycled between requests (power-of-two size classes)
    std::mutex pin_mu;
    std::multimap<size_t, void *> pin_free;

    // request queue
    // Queued single-image requests are served by `lanes` worker threads, each driving its own child context (own
    // stream, scratch, table cache): while one lane's batch is on the PCIe link / in kernels, another lane is already
    // collecting and uploading the next batch.
    std::vector<std::thread> workers;
    std::vector<flgpu_ctx *> lanes;
    bool collecting = false; // a worker is gathering a batch (one collector at a time keeps batches large)
    std::atomic<int> staging{0}; // callers currently copying their source into pinned memory, i.e. about to enqueue
    // admission: callers beyond a few batches' worth wait BEFORE staging (a thousand threads each copying megabytes
    // into pinned memory only evict each other's buffers and starve the lane threads of CPU time)
    std::mutex adm_mu;
    std::condition_variable adm_cv;
    uint32_t admitted = 0;
    std::mutex qmu;
    std::condition_variable qcv, qdone;
    std::deque<Request *> queue;
    bool stop = false;
    bool worker_started = false;

    int fail(hipError_t e, const char *what)
    {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        last_error = buf;
        return e == hipErrorOutOfMemory ? FLGPU_ERR_OOM : FLGPU_ERR_DEVICE;
    }
};

#define FL_HIP(ctx, call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) return (ctx)->fail(e__, what); } while (0)

namespace {

constexpr size_t kArenaWords = (size_t)16 << 20; // 64 MiB of tables

// ---- arena ---------------------------------------------------------------

// Appends words, returns the word offset; 0 is never a valid offset (word 0 is a sentinel).
uint32_t arena_append(flgpu_ctx *c, const void *data, size_t words, size_t align_words = 4)
{
    size_t off = align_up(c->h_arena.size(), align_words);
    if (off + words > c->arena_cap_words) return 0;
    c->h_arena.resize(off + words);
    if (data) memcpy(c->h_arena.data() + off, data, words * 4);
    return (uint32_t)off;
}

void arena_reset(flgpu_ctx *c)
{
    c->h_arena.clear();
    c->h_arena.push_back(0xFA171200u); // sentinel so that no table sits at offset 0
    c->arena_uploaded = 0;
    c->axis_off.clear();
    c->axis_host.clear();
    c->stream_plans.clear();
    c->blur_plans.clear();
    c->jpeg_tables.clear();
    std::vector<uint32_t> g;
    build_webp_gamma(g);
    c->gamma_off = arena_append(c, g.data(), g.size());
}

int arena_flush(flgpu_ctx *c, hipStream_t st)
{
    if (c->arena_uploaded == c->h_arena.size()) return FLGPU_OK;
    const size_t from = c->arena_uploaded;
    // pageable source: the runtime stages it before returning, so h_arena may grow afterwards
    FL_HIP(c, hipMemcpyAsync(c->d_arena + from, c->h_arena.data() + from, (c->h_arena.size() - from) * 4, hipMemcpyHostToDevice, st),
           "table upload");
    c->arena_uploaded = c->h_arena.size();
    return FLGPU_OK;
}

// Returns the header offset of the axis table, building it on a miss (0 = arena full).
uint32_t get_axis(flgpu_ctx *c, uint32_t in, uint32_t out, Filter f, float sigma, AxisKey *key_out, const HostAxis **host_out)
{
    AxisKey key(in, out, (uint32_t)f, float_bits(sigma));
    if (key_out) *key_out = key;
    auto it = c->axis_off.find(key);
    if (it != c->axis_off.end()) {
        if (host_out) *host_out = &c->axis_host[key];
        return it->second;
    }
    HostAxis &ha = c->axis_host[key];
    build_axis(in, out, f, sigma, ha);
    c->stats.tables_built++;
    AxisTable hdr{};
    hdr.in_size = in; hdr.out_size = out; hdr.max_taps = ha.max_taps; hdr.total_taps = (uint32_t)ha.weights.size();
    const uint32_t hoff = arena_append(c, nullptr, sizeof(AxisTable) / 4);
    hdr.left_off = arena_append(c, ha.left.data(), out);
    hdr.count_off = arena_append(c, ha.count.data(), out);
    hdr.woff_off = arena_append(c, ha.woff.data(), out);
    hdr.weights_off = arena_append(c, ha.weights.data(), ha.weights.size());
    if (!hoff || !hdr.left_off || !hdr.count_off || !hdr.woff_off || !hdr.weights_off) {
        c->axis_host.erase(key);
        return 0;
    }
    memcpy(c->h_arena.data() + hoff, &hdr, sizeof(hdr));
    c->axis_off[key] = hoff;
    if (host_out) *host_out = &ha;
    return hoff;
}

const StreamPlan *get_stream_plan(flgpu_ctx *c, const AxisKey &vk, const HostAxis &va, const AxisKey &hk, const HostAxis &ha,
                                  uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t nbands, uint32_t cs, uint32_t pre)
{
    StreamPlanKey key{vk, hk, cx, cy, cw, ch, nbands, mid_channels(cs, pre) == 1 ? 1u : 0u};
    auto it = c->stream_plans.find(key);
    if (it != c->stream_plans.end()) return &it->second;
    StreamPlan plan;
    const uint32_t T = stream_lanes(), span_max = T * PXL;
    // strips: fewest equal-width strips such that each has <= T output columns and <= T*PXL source pixels
    std::vector<HostStrip> strips;
    for (uint32_t ns = std::max(1u, (cw + T - 1) / T); ns <= cw; ++ns) {
        strips.clear();
        bool fits = true;
        const uint32_t per = (cw + ns - 1) / ns;
        for (uint32_t x = cx; x < cx + cw && fits; x += per) {
            HostStrip s;
            build_strip(ha, x, std::min(x + per, cx + cw), T, PXL, s);
            if (s.sx1 - s.sx0 > span_max || s.x1 - s.x0 > T) fits = false;
            strips.push_back(std::move(s));
        }
        if (fits) break;
        strips.clear();
    }
    bool ok = !strips.empty();
    // fewest accumulator slots that can hold every output row alive on one source row (fewer slots = fewer VGPRs)
    // fewest accumulator slots whose schedule works for the whole kept range (fewer slots = fewer VGPRs and fewer idle FMAs)
    {
        uint32_t r0 = 0, r1 = 0;
        std::vector<RowSched> probe;
        plan.nacc = build_row_sched(va, cy, cy + ch, 7, stream_block_rows(), r0, r1, probe) ? 7 : NACC;
    }
    struct Band { uint32_t y0, y1, r0, r1, sched_off; };
    std::vector<Band> bands;
    if (ok) {
        const uint32_t per = (ch + nbands - 1) / nbands;
        for (uint32_t y = cy; y < cy + ch && ok; y += per) {
            Band b{y, std::min(y + per, cy + ch), 0, 0, 0};
            std::vector<RowSched> sched;
            if (!build_row_sched(va, b.y0, b.y1, plan.nacc, stream_block_rows(), b.r0, b.r1, sched)) { ok = false; break; }
            b.sched_off = arena_append(c, sched.data(), sched.size() * sizeof(RowSched) / 4);
            if (!b.sched_off) ok = false;
            bands.push_back(b);
        }
    }
    if (ok) {
        for (auto &s : strips) {
            const uint32_t wt_off = arena_append(c, s.wt.data(), s.wt.size());
            const uint32_t po_off = arena_append(c, s.po.data(), s.po.size());
            if (!wt_off || !po_off) { ok = false; break; }
            plan.lds_bytes = std::max(plan.lds_bytes, stream_lds_bytes(s.jmax, s.x1 - s.x0, s.ks, mid_channels(cs, pre)));
            for (size_t bi = 0; bi < bands.size(); ++bi) {
                const Band &b = bands[bi];
                StreamItem it2{};
                it2.y0 = b.y0; it2.y1 = b.y1; it2.x0 = s.x0; it2.x1 = s.x1; it2.r0 = b.r0; it2.r1 = b.r1;
                it2.sx0 = s.sx0; it2.sched_off = b.sched_off; it2.wt_off = wt_off; it2.po_off = po_off;
                it2.jmax = s.jmax; it2.kmax = s.kmax; it2.ks = s.ks;
                plan.items.push_back(it2);
            }
        }
    }
    if (ok && plan.lds_bytes > 150 * 1024) ok = false;
    plan.ok = ok;
    if (!ok) plan.items.clear();
    auto res = c->stream_plans.emplace(key, std::move(plan));
    return &res.first->second;
}

// ---- events / profiling ----------------------------------------------------

hipEvent_t get_event(flgpu_ctx *c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void resolve_pending(flgpu_ctx *c)
{
    for (auto &p : c->pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (p.kind == 0) c->stats.resample_ms += ms;
            else if (p.kind == 1) c->stats.blur_ms += ms;
            else c->stats.frontend_ms += ms;
        }
        c->event_pool.push_back(p.a);
        c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

struct ProfileScope {
    flgpu_ctx *c; hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr;
    ProfileScope(flgpu_ctx *c_, hipStream_t st_, int kind_) : c(c_), st(st_), kind(kind_)
    {
        if (!c->cfg.profile) return;
        if (c->pending.size() > 2048) resolve_pending(c);
        a = get_event(c); b = get_event(c);
        if (a && b) (void)hipEventRecord(a, st);
    }
    ~ProfileScope()
    {
        if (a && b) { (void)hipEventRecord(b, st); c->pending.push_back({a, b, kind}); }
    }
};

// ---- batch execution -------------------------------------------------------

enum Stage1Kind { S1_NONE = 0, S1_PLACE = 1, S1_GENERIC = 2, S1_STREAM = 3, S1_NEAREST = 4 };

struct Work {
    flgpu_plan plan;
    const flgpu_params *p;
    uint32_t cs, pre, sw, sh;
    Stage1Kind s1;
    const uint8_t *src;
    uint8_t *s1_dst;   // output of stage 1 (== src when S1_NONE)
    uint8_t *blur_dst; // output of the blur stage (or null)
    uint8_t *final_dst;
    AxisKey vk, hk;
    const HostAxis *va = nullptr, *ha = nullptr;
    uint32_t vtab = 0, htab = 0;
    const StreamPlan *splan = nullptr;
    bool unaligned = false;
    size_t jpeg_coef_off = 0, jpeg_off_off = 0, jpeg_raw_off = 0; // FE_JPEG scratch (bytes)
    uint32_t jpeg_tab = 0;
    uint32_t orient = 0, raw_w = 0, raw_h = 0; // EXIF orientation pre-pass (2..8), source size before it
    size_t orient_off = 0;
    const uint8_t *raw_src = nullptr;
};

struct GroupKey {
    uint32_t kind, cs, pre, lb;
    bool operator<(const GroupKey &o) const { return std::tie(kind, cs, pre, lb) < std::tie(o.kind, o.cs, o.pre, o.lb); }
};

// Channels the blur really has to filter: a letterboxed picture of an opaque source has alpha == 255 everywhere,
// and a grey one on a grey fill has R == G == B (see blur_tile_kernel).
uint32_t blur_channels(const Work &w)
{
    uint32_t ce = w.plan.out_c;
    if (w.plan.letterboxed && (w.cs == 1 || w.cs == 3)) {
        const bool grey = mid_channels(w.cs, w.pre) == 1 && w.p->fill_r == w.p->fill_g && w.p->fill_g == w.p->fill_b;
        ce = grey ? 1u : 3u;
    }
    return ce;
}

void fill_job(const Work &w, Job &j)
{
    memset(&j, 0, sizeof(j));
    const flgpu_plan &pl = w.plan;
    j.src = w.src;
    j.dst = w.s1_dst;
    j.src_bytes = w.sw * w.sh * w.cs;
    j.sw = w.sw; j.sh = w.sh;
    j.rw = pl.resized_w; j.rh = pl.resized_h;
    j.cx = pl.crop_x; j.cy = pl.crop_y;
    if (pl.letterboxed) {
        j.cw = std::min(pl.resized_w - pl.crop_x, pl.out_w - pl.place_x);
        j.ch = std::min(pl.resized_h - pl.crop_y, pl.out_h - pl.place_y);
    } else {
        j.cw = pl.out_w; j.ch = pl.out_h;
    }
    j.dw = pl.out_w; j.dh = pl.out_h;
    j.ox = pl.place_x; j.oy = pl.place_y;
    j.fill = (uint32_t)w.p->fill_r | ((uint32_t)w.p->fill_g << 8) | ((uint32_t)w.p->fill_b << 16) | (255u << 24);
    j.vtab = w.vtab; j.htab = w.htab;
    if (w.s1 == S1_NEAREST) {
        // sample.rs: ratio = in as f32 / out as f32, carried as bits where the Lanczos3 jobs carry table offsets
        const float ry = (float)w.sh / (float)pl.resized_h, rx = (float)w.sw / (float)pl.resized_w;
        memcpy(&j.vtab, &ry, 4); memcpy(&j.htab, &rx, 4);
    }
}

int run_batch_device(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, bool same_params,
                     flgpu_image *dsts, hipStream_t st)
{
    if (n == 0) return FLGPU_OK;
    if (!srcs || !ps || !dsts) return FLGPU_ERR_INVALID_ARG;
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    if (!st) st = c->stream;
    if (c->last_stream && c->last_stream != st && c->last_done) FL_HIP(c, hipStreamWaitEvent(st, c->last_done, 0), "stream handoff");

    // ---- plan every image ------------------------------------------------
    std::vector<Work> work(n);
    size_t tmp_a_bytes = 0, tmp_b_bytes = 0, tmp_o_bytes = 0, jpeg_coef_bytes = 0, jpeg_off_bytes = 0, jpeg_raw_bytes = 0;
    for (size_t i = 0; i < n; ++i) {
        Work &w = work[i];
        const flgpu_image &s = srcs[i];
        w.p = same_params ? &ps[0] : &ps[i];
        if (!s.data || !dsts[i].data) return FLGPU_ERR_INVALID_ARG;
        int rc = flgpu_plan_output(w.p, s.width, s.height, s.channels, &w.plan);
        if (rc) return rc;
        if (s.capacity < (uint64_t)s.width * s.height * s.channels) return FLGPU_ERR_INVALID_ARG;
        if (dsts[i].capacity < w.plan.out_bytes) return FLGPU_ERR_BUFFER_TOO_SMALL;
        w.cs = s.channels; w.sw = w.plan.src_w; w.sh = w.plan.src_h; // size after apply_orientation
        w.raw_w = s.width; w.raw_h = s.height;
        w.orient = w.p->orientation >= 2 ? w.p->orientation : 0;
        if (w.orient) { w.orient_off = tmp_o_bytes; tmp_o_bytes += align_up((size_t)s.width * s.height * s.channels, 256); }
        w.pre = w.p->grayscale ? PRE_GRAY : (w.p->inverse ? PRE_INVERT : PRE_NONE);
        w.src = s.data;
        w.final_dst = dsts[i].data;
        const flgpu_plan &pl = w.plan;
        const bool cropped = pl.crop_x || pl.crop_y || pl.out_w != pl.resized_w || pl.out_h != pl.resized_h;
        // grayscale of Luma/LumaA and "no-op" pre-ops change nothing
        const bool pre_changes = (w.pre == PRE_INVERT) || (w.pre == PRE_GRAY && w.cs >= 3);
        if (!pre_changes) w.pre = PRE_NONE;
        if (pl.resampled) w.s1 = w.p->filter == FLGPU_FILTER_NEAREST ? S1_NEAREST : S1_GENERIC;
        else if (pre_changes || pl.letterboxed || cropped) w.s1 = S1_PLACE;
        else w.s1 = S1_NONE;
        const bool blur = w.p->blur_sigma > 0.0f;
        const bool fe = w.p->front_end != FLGPU_FE_NONE;
        // buffer chain
        if (w.s1 == S1_NONE) w.s1_dst = const_cast<uint8_t *>(w.src);
        else if (!blur && !fe) w.s1_dst = w.final_dst;
        else { w.s1_dst = reinterpret_cast<uint8_t *>(tmp_a_bytes); tmp_a_bytes += align_up(pl.pixel_bytes, 256); }
        if (blur) {
            if (!fe) w.blur_dst = w.final_dst;
            else { w.blur_dst = reinterpret_cast<uint8_t *>(tmp_b_bytes); tmp_b_bytes += align_up(pl.pixel_bytes, 256); }
        } else w.blur_dst = nullptr;
        if (w.p->front_end == FLGPU_FE_JPEG) {
            if (pl.out_w > 65535u || pl.out_h > 65535u) return FLGPU_ERR_UNSUPPORTED; // SOF0 carries u16 dimensions
            const size_t units = (size_t)(pl.plane_w / 8u) * (pl.plane_h / 8u) * 3u;
            if (units * kJpegMaxUnitBytes * 8 >= ((size_t)1 << 32)) return FLGPU_ERR_UNSUPPORTED; // bit offsets are 32-bit
            w.jpeg_coef_off = jpeg_coef_bytes; jpeg_coef_bytes += align_up(units * sizeof(uint32_t), 256);
            w.jpeg_off_off = jpeg_off_bytes; jpeg_off_bytes += align_up((units + 1) * sizeof(uint32_t), 256);
            w.jpeg_raw_off = jpeg_raw_bytes; jpeg_raw_bytes += align_up(units * kAcWordsPerUnit * sizeof(uint32_t), 256);
        }
    }
    FL_HIP(c, c->d_jpeg_coef.reserve(jpeg_coef_bytes), "JPEG coefficient scratch");
    FL_HIP(c, c->d_jpeg_off.reserve(jpeg_off_bytes), "JPEG offset scratch");
    FL_HIP(c, c->d_jpeg_raw.reserve(jpeg_raw_bytes), "JPEG bit-stream scratch");
    FL_HIP(c, c->d_tmp_o.reserve(tmp_o_bytes), "orientation scratch");
    for (auto &w : work)
        if (w.orient) { w.raw_src = w.src; w.src = static_cast<uint8_t *>(c->d_tmp_o.p) + w.orient_off; }
    FL_HIP(c, c->d_tmp_a.reserve(tmp_a_bytes), "scratch A");
    FL_HIP(c, c->d_tmp_b.reserve(tmp_b_bytes), "scratch B");
    for (auto &w : work) {
        const bool blur = w.p->blur_sigma > 0.0f, fe = w.p->front_end != FLGPU_FE_NONE;
        if (w.s1 == S1_NONE) w.s1_dst = const_cast<uint8_t *>(w.src);
        if (w.s1 != S1_NONE && (blur || fe)) w.s1_dst = static_cast<uint8_t *>(c->d_tmp_a.p) + reinterpret_cast<size_t>(w.s1_dst);
        if (blur && fe) w.blur_dst = static_cast<uint8_t *>(c->d_tmp_b.p) + reinterpret_cast<size_t>(w.blur_dst);
    }

    // ---- tables ------------------------------------------------------------
    // first pass may overflow the arena: reset once and retry
    const char *env_generic = getenv("FLGPU_FORCE_GENERIC");
    const char *env_bands = getenv("FLGPU_FORCE_BANDS");
    const bool force_generic = env_generic && env_generic[0] == '1';
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool full = false;
        size_t n_resample = 0;
        for (auto &w : work) n_resample += (w.plan.resampled && w.s1 != S1_NEAREST) ? 1 : 0;
        for (auto &w : work) {
            if (!w.plan.resampled || w.s1 == S1_NEAREST) continue;
            w.vtab = get_axis(c, w.sh, w.plan.resized_h, FILTER_LANCZOS3, 0.0f, &w.vk, &w.va);
            w.htab = get_axis(c, w.sw, w.plan.resized_w, FILTER_LANCZOS3, 0.0f, &w.hk, &w.ha);
            if (!w.vtab || !w.htab) { full = true; break; }
            // fused streaming kernel if the geometry allows it
            w.s1 = S1_GENERIC;
            // rows that are not dword aligned: Rgb8 has a funnel-shift variant of the kernel, others use the generic path
            w.unaligned = ((w.sw * w.cs) % 4u != 0) || ((uintptr_t)w.src % 4u != 0);
            const bool aligned = (!w.unaligned || w.cs == 3) && (!w.plan.letterboxed || (uintptr_t)w.s1_dst % 4u == 0);
            if (stream_supported(w.cs, w.pre) && aligned && !force_generic) {
                Job jtmp; fill_job(w, jtmp);
                uint32_t nbands = 1;
                if (env_bands) nbands = (uint32_t)std::max(1, atoi(env_bands));
                else if (n_resample < 512) {
                    // small batches: split images into row bands so that the chip still gets >= ~1024 workgroups
                    const uint32_t want = (uint32_t)((1024 + n_resample * 2 - 1) / (n_resample * 2));
                    nbands = std::max(1u, std::min(want, jtmp.ch / 24u));
                }
                nbands = std::min(nbands, std::max(1u, jtmp.ch));
                const StreamPlan *sp = get_stream_plan(c, w.vk, *w.va, w.hk, *w.ha, jtmp.cx, jtmp.cy, jtmp.cw, jtmp.ch, nbands, w.cs, w.pre);
                if (c->h_arena.size() >= c->arena_cap_words - 1024) { full = true; break; }
                if (sp->ok) { w.s1 = S1_STREAM; w.splan = sp; }
            }
        }
        for (auto &w : work) {
            if (full) break;
            if (w.p->blur_sigma > 0.0f) {
                AxisKey k; const HostAxis *h;
                AxisKey kv; const HostAxis *hv;
                if (!get_axis(c, w.plan.out_h, w.plan.out_h, FILTER_GAUSSIAN, w.p->blur_sigma, &kv, &hv) ||
                    !get_axis(c, w.plan.out_w, w.plan.out_w, FILTER_GAUSSIAN, w.p->blur_sigma, &k, &h)) full = true;
                else {
                    const uint32_t ty = blur_band_rows(blur_channels(w));
                    if (blur_tile_supported(h->max_taps) && blur_tile_supported(hv->max_taps) && !c->blur_plans.count(std::make_tuple(kv, k, ty))) {
                        std::vector<uint32_t> blk;
                        build_blur_plan(*hv, *h, blur_tile_count(w.plan.out_w, h->max_taps), ty, blk);
                        const uint32_t off = arena_append(c, blk.data(), blk.size());
                        if (!off) full = true; else c->blur_plans[std::make_tuple(kv, k, ty)] = off;
                    }
                }
            }
        }
        for (auto &w : work) {
            if (full) break;
            if (w.p->front_end != FLGPU_FE_JPEG) continue;
            const uint32_t q = std::min<uint32_t>(std::max<uint32_t>(w.p->quality, 1u), 100u); // handler.rs:275 quality().clamp(1, 100)
            const auto key = std::make_tuple(w.plan.out_w, w.plan.out_h, q);
            auto it = c->jpeg_tables.find(key);
            if (it == c->jpeg_tables.end()) {
                std::vector<uint32_t> blk;
                build_jpeg_tables(w.plan.out_w, w.plan.out_h, q, blk);
                const uint32_t off = arena_append(c, blk.data(), blk.size());
                if (!off) { full = true; break; }
                it = c->jpeg_tables.emplace(key, off).first;
            }
            w.jpeg_tab = it->second;
        }
        if (!full) break;
        if (attempt == 1) return FLGPU_ERR_OOM;
        FL_HIP(c, hipStreamSynchronize(st), "arena reset sync");
        FL_HIP(c, hipDeviceSynchronize(), "arena reset sync");
        arena_reset(c);
    }
    { int rc = arena_flush(c, st); if (rc) return rc; }

    // ---- descriptors ---------------------------------------------------------
    std::map<GroupKey, std::vector<size_t>> s1_groups, blur_groups, fe_groups;
    for (size_t i = 0; i < n; ++i) {
        const Work &w = work[i];
        if (w.s1 != S1_NONE) s1_groups[{(uint32_t)w.s1 | (w.splan ? w.splan->nacc << 8 : 0u) | (w.s1 == S1_STREAM && w.unaligned ? 1u << 16 : 0u), w.cs, w.pre, w.plan.letterboxed}].push_back(i);
        if (w.p->blur_sigma > 0.0f) {
            const uint32_t ce = blur_channels(w);
            // pictures of one launch share the workgroup width the kernel is instantiated for
            AxisKey hk2; const HostAxis *hh2 = nullptr;
            const uint32_t lanes = (get_axis(c, w.plan.out_w, w.plan.out_w, FILTER_GAUSSIAN, w.p->blur_sigma, &hk2, &hh2) && hh2 &&
                                    blur_tile_supported(hh2->max_taps)) ? blur_lanes(w.plan.out_w, hh2->max_taps) : 256u;
            blur_groups[{lanes, w.plan.out_c, ce, 0}].push_back(i);
        }
        if (w.p->front_end != FLGPU_FE_NONE) fe_groups[{w.p->front_end, 0, 0, 0}].push_back(i);
    }
    std::vector<Job> jobs;
    std::vector<StreamItem> items;
    std::vector<FrontendJob> fjobs;
    // EXIF orientation pre-pass jobs, grouped by channel count
    struct OrientLaunch { uint32_t cs, base, n, mw, mh; };
    std::vector<OrientLaunch> orient_launches;
    for (uint32_t cs = 1; cs <= 4; ++cs) {
        OrientLaunch O{cs, (uint32_t)jobs.size(), 0, 0, 0};
        for (auto &w : work) {
            if (!w.orient || w.cs != cs) continue;
            Job j; memset(&j, 0, sizeof(j));
            j.src = w.raw_src; j.dst = const_cast<uint8_t *>(w.src);
            j.sw = w.raw_w; j.sh = w.raw_h; j.dw = w.sw; j.dh = w.sh; j.fill = w.orient;
            O.mw = std::max(O.mw, j.dw); O.mh = std::max(O.mh, j.dh);
            jobs.push_back(j);
            O.n++;
        }
        if (O.n) orient_launches.push_back(O);
    }
    struct S1Launch { GroupKey k; uint32_t job_base, njobs, item_base, nitems, nacc; LaunchGeneric g; size_t lds; size_t mid_floats; uint32_t blur_grid_x; bool blur_tiled; };
    std::vector<S1Launch> s1_launches, blur_launches;
    struct FeLaunch { uint32_t kind, base, n, mw, mh; bool rgba; };
    std::vector<FeLaunch> fe_launches;
    size_t mid_floats_max = 0;
    const size_t kMidCapFloats = (size_t)256 << 20; // 1 GiB of f32 intermediate per launch group

    auto new_launch = [&](const GroupKey &k) {
        S1Launch L{};
        L.k = k; L.job_base = (uint32_t)jobs.size(); L.item_base = (uint32_t)items.size();
        L.g.cs = k.cs; L.g.pre = k.pre; L.g.letterbox = k.lb; L.g.grouped = 1;
        return L;
    };
    for (auto &kv : s1_groups) {
        const GroupKey &k = kv.first;
        S1Launch L = new_launch(k);
        for (size_t idx : kv.second) {
            const Work &w = work[idx];
            Job j; fill_job(w, j);
            const size_t mid = ((k.kind & 255u) == S1_GENERIC) ? (size_t)w.sw * w.plan.resized_h * mid_channels(w.cs, w.pre) : 0;
            if ((k.kind & 255u) == S1_GENERIC && L.njobs && L.mid_floats + mid > kMidCapFloats) {
                s1_launches.push_back(L);
                L = new_launch(k);
            }
            j.mid_off = (uint32_t)L.mid_floats;
            L.mid_floats += mid;
            mid_floats_max = std::max(mid_floats_max, L.mid_floats);
            L.g.max_sw = std::max(L.g.max_sw, j.sw); L.g.max_rh = std::max(L.g.max_rh, j.rh);
            L.g.max_cw = std::max(L.g.max_cw, j.cw); L.g.max_ch = std::max(L.g.max_ch, j.ch);
            L.g.max_dw = std::max(L.g.max_dw, j.dw); L.g.max_dh = std::max(L.g.max_dh, j.dh);
            if ((k.kind & 255u) == S1_STREAM) {
                for (StreamItem it2 : w.splan->items) { it2.job = (uint32_t)jobs.size(); items.push_back(it2); }
                L.nitems += (uint32_t)w.splan->items.size();
                L.lds = std::max(L.lds, w.splan->lds_bytes);
                L.nacc = w.splan->nacc;
                c->stats.resample_src_bytes += (uint64_t)j.src_bytes;
                c->stats.resample_dst_bytes += w.plan.pixel_bytes;
            }
            jobs.push_back(j);
            L.njobs++;
        }
        if ((k.kind & 255u) == S1_STREAM && L.nitems > 1) {
            // longest workgroups first: in a mixed batch a 4K band walks four times the rows of a 1080p one, and the
            // hardware hands out workgroups in index order -- started last, the long ones would be the launch's tail
            auto first = items.begin() + L.item_base;
            std::stable_sort(first, first + L.nitems, [](const StreamItem &a, const StreamItem &b) { return a.r1 - a.r0 > b.r1 - b.r0; });
        }
        s1_launches.push_back(L);
    }
    for (auto &kv : blur_groups) {
        const GroupKey &k = kv.first; // cs = channel count of the blurred image
        S1Launch L = new_launch(k);
        for (size_t idx : kv.second) {
            const Work &w = work[idx];
            const flgpu_plan &pl = w.plan;
            Job j; memset(&j, 0, sizeof(j));
            j.src = w.s1_dst; j.dst = w.blur_dst; j.src_bytes = (uint32_t)pl.pixel_bytes;
            j.sw = pl.out_w; j.sh = pl.out_h; j.rw = pl.out_w; j.rh = pl.out_h; j.cw = pl.out_w; j.ch = pl.out_h;
            j.dw = pl.out_w; j.dh = pl.out_h;
            AxisKey vkey, hkey;
            j.vtab = get_axis(c, pl.out_h, pl.out_h, FILTER_GAUSSIAN, w.p->blur_sigma, &vkey, nullptr);
            j.htab = get_axis(c, pl.out_w, pl.out_w, FILTER_GAUSSIAN, w.p->blur_sigma, &hkey, nullptr);
            {
                auto bt = c->blur_plans.find(std::make_tuple(vkey, hkey, blur_band_rows(k.pre ? k.pre : pl.out_c)));
                j.pad0 = bt != c->blur_plans.end() ? bt->second : 0u; // table block of the blur kernel
            }
            const size_t mid = (size_t)pl.out_w * pl.out_h * pl.out_c;
            if (L.njobs && L.mid_floats + mid > kMidCapFloats) { blur_launches.push_back(L); L = new_launch(k); }
            {
                const AxisTable *vh = reinterpret_cast<const AxisTable *>(c->h_arena.data() + j.vtab);
                const AxisTable *hh = reinterpret_cast<const AxisTable *>(c->h_arena.data() + j.htab);
                if (L.njobs == 0) L.blur_tiled = true;
                const size_t lds = blur_lds_bytes(pl.out_w, k.pre ? k.pre : pl.out_c, vh->max_taps, hh->max_taps); // k.pre = channels filtered
                if (!blur_tile_supported(hh->max_taps) || !blur_tile_supported(vh->max_taps) || lds > 150 * 1024 || !j.pad0) L.blur_tiled = false;
                L.lds = std::max(L.lds, lds);
                L.blur_grid_x = std::max(L.blur_grid_x, blur_grid_x(pl.out_w, pl.out_h, hh->max_taps, k.pre ? k.pre : pl.out_c));
            }
            j.mid_off = (uint32_t)L.mid_floats;
            L.mid_floats += mid;
            mid_floats_max = std::max(mid_floats_max, L.mid_floats);
            L.g.max_sw = std::max(L.g.max_sw, j.sw); L.g.max_rh = std::max(L.g.max_rh, j.rh);
            L.g.max_cw = std::max(L.g.max_cw, j.cw); L.g.max_ch = std::max(L.g.max_ch, j.ch);
            jobs.push_back(j);
            L.njobs++;
        }
        blur_launches.push_back(L);
    }
    // result words: two per image of the batch, see flgpu_ctx::last_fe
    const bool has_results = !fe_groups.empty();
    if (has_results) {
        FL_HIP(c, c->d_status.reserve(n * 8), "result words");
        FL_HIP(c, hipMemsetAsync(c->d_status.p, 0, n * 8, st), "result clear");
    }
    std::vector<JpegJob> jjobs;
    uint32_t jpeg_max_blocks = 0;
    for (auto &kv : fe_groups) {
        if (kv.first.kind == FLGPU_FE_JPEG) {
            for (size_t idx : kv.second) {
                const Work &w = work[idx];
                const flgpu_plan &pl = w.plan;
                JpegJob j; memset(&j, 0, sizeof(j));
                j.src = w.blur_dst ? w.blur_dst : w.s1_dst;
                j.dst = w.final_dst;
                j.meta = reinterpret_cast<uint32_t *>(static_cast<char *>(c->d_jpeg_coef.p) + w.jpeg_coef_off);
                j.unit_off = reinterpret_cast<uint32_t *>(static_cast<char *>(c->d_jpeg_off.p) + w.jpeg_off_off);
                j.acbits = reinterpret_cast<uint32_t *>(static_cast<char *>(c->d_jpeg_raw.p) + w.jpeg_raw_off);
                j.result = static_cast<uint32_t *>(c->d_status.p) + 2 * idx;
                j.w = pl.out_w; j.h = pl.out_h; j.c = pl.out_c;
                j.bx = pl.plane_w / 8u; j.by = pl.plane_h / 8u;
                j.tab_off = w.jpeg_tab;
                j.dst_cap = (uint32_t)std::min<uint64_t>(dsts[idx].capacity, 0xffffffffull);
                jpeg_max_blocks = std::max(jpeg_max_blocks, j.bx * j.by);
                jjobs.push_back(j);
            }
            continue;
        }
        FeLaunch F{kv.first.kind, (uint32_t)fjobs.size(), 0, 0, 0, true};
        for (size_t idx : kv.second) {
            const Work &w = work[idx];
            const flgpu_plan &pl = w.plan;
            FrontendJob f; memset(&f, 0, sizeof(f));
            f.src = w.blur_dst ? w.blur_dst : w.s1_dst;
            f.dst = w.final_dst;
            f.status = static_cast<uint32_t *>(c->d_status.p) + 2 * idx;
            f.w = pl.out_w; f.h = pl.out_h; f.c = pl.out_c;
            f.plane_w = pl.plane_w; f.plane_h = pl.plane_h; f.chroma_w = pl.chroma_w; f.chroma_h = pl.chroma_h;
            if (f.c != 4 || ((uintptr_t)f.src & 3u) || ((uintptr_t)f.dst & 3u)) F.rgba = false;
            if (F.kind == FLGPU_FE_JFIF444) { F.mw = std::max(F.mw, f.plane_w); F.mh = std::max(F.mh, f.plane_h); }
            else { F.mw = std::max(F.mw, f.chroma_w); F.mh = std::max(F.mh, f.chroma_h); }
            fjobs.push_back(f);
            F.n++;
        }
        fe_launches.push_back(F);
    }
    FL_HIP(c, c->d_mid.reserve(mid_floats_max * 4), "f32 intermediate");

    // one staging slot: [jobs][items][fjobs][jjobs]
    const size_t jobs_b = align_up(jobs.size() * sizeof(Job), 256), items_b = align_up(items.size() * sizeof(StreamItem), 256),
                 fjobs_b = align_up(fjobs.size() * sizeof(FrontendJob), 256), jjobs_b = align_up(jjobs.size() * sizeof(JpegJob), 256);
    const size_t desc_b = jobs_b + items_b + fjobs_b + jjobs_b;
    const Job *d_jobs = nullptr; const StreamItem *d_items = nullptr; const FrontendJob *d_fjobs = nullptr; const JpegJob *d_jjobs = nullptr;
    DescSlot *slot = nullptr;
    if (desc_b) {
        slot = &c->slots[c->next_slot];
        c->next_slot = (c->next_slot + 1) % 4;
        if (slot->busy) { FL_HIP(c, hipEventSynchronize(slot->done), "descriptor slot wait"); slot->busy = false; }
        if (!slot->done) FL_HIP(c, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming), "event");
        FL_HIP(c, slot->host.reserve(desc_b), "pinned descriptors");
        FL_HIP(c, slot->dev.reserve(desc_b), "device descriptors");
        char *hp = static_cast<char *>(slot->host.p);
        if (!jobs.empty()) memcpy(hp, jobs.data(), jobs.size() * sizeof(Job));
        if (!items.empty()) memcpy(hp + jobs_b, items.data(), items.size() * sizeof(StreamItem));
        if (!fjobs.empty()) memcpy(hp + jobs_b + items_b, fjobs.data(), fjobs.size() * sizeof(FrontendJob));
        if (!jjobs.empty()) memcpy(hp + jobs_b + items_b + fjobs_b, jjobs.data(), jjobs.size() * sizeof(JpegJob));
        FL_HIP(c, hipMemcpyAsync(slot->dev.p, hp, desc_b, hipMemcpyHostToDevice, st), "descriptor upload");
        char *dp = static_cast<char *>(slot->dev.p);
        d_jobs = reinterpret_cast<const Job *>(dp);
        d_items = reinterpret_cast<const StreamItem *>(dp + jobs_b);
        d_fjobs = reinterpret_cast<const FrontendJob *>(dp + jobs_b + items_b);
        d_jjobs = reinterpret_cast<const JpegJob *>(dp + jobs_b + items_b + fjobs_b);
    }

    // ---- launches --------------------------------------------------------------
    for (auto &O : orient_launches) {
        LaunchGeneric g{};
        g.jobs = d_jobs; g.job_base = O.base; g.njobs = O.n; g.cs = O.cs; g.max_dw = O.mw; g.max_dh = O.mh;
        FL_HIP(c, launch_orient(g, st), "orientation kernel");
    }
    for (auto &L : s1_launches) {
        L.g.jobs = d_jobs; L.g.arena = c->d_arena; L.g.mid = static_cast<float *>(c->d_mid.p);
        L.g.job_base = L.job_base; L.g.njobs = L.njobs;
        if ((L.k.kind & 255u) == S1_NEAREST) {
            L.g.nearest = 1;
            FL_HIP(c, launch_place(L.g, false, st), "nearest kernel");
        } else if ((L.k.kind & 255u) == S1_PLACE) {
            FL_HIP(c, launch_place(L.g, false, st), "place kernel");
        } else if ((L.k.kind & 255u) == S1_GENERIC) {
            if (L.k.lb) FL_HIP(c, launch_place(L.g, true, st), "border fill");
            FL_HIP(c, launch_vpass_generic(L.g, st), "generic vertical pass");
            FL_HIP(c, launch_hpass_generic(L.g, st), "generic horizontal pass");
            c->stats.generic_launches++;
        } else {
            LaunchStream s{}; // (the streaming kernel paints the letterbox frame itself)
            s.jobs = d_jobs; s.items = d_items + L.item_base; s.arena = c->d_arena; s.nitems = L.nitems;
            s.cs = L.k.cs; s.pre = L.k.pre; s.letterbox = L.k.lb; s.lds_bytes = L.lds; s.nacc = L.nacc; s.unaligned = (L.k.kind >> 16) & 1u;
            {
                ProfileScope ps(c, st, 0);
                FL_HIP(c, launch_stream(s, st), "streaming resample kernel");
            }
            c->stats.resample_launches++;
        }
    }
    for (auto &L : blur_launches) {
        L.g.jobs = d_jobs; L.g.arena = c->d_arena; L.g.mid = static_cast<float *>(c->d_mid.p);
        L.g.job_base = L.job_base; L.g.njobs = L.njobs; L.g.letterbox = 0;
        L.g.grouped = 0;
        ProfileScope ps(c, st, 1);
        if (L.blur_tiled && !force_generic) {
            L.g.pre = L.k.pre; // channels to filter (group key), see blur_tile_kernel
            L.g.blur_lanes = L.k.kind;
            FL_HIP(c, launch_blur_tile(L.g, L.blur_grid_x, L.lds, st), "blur kernel");
        } else {
            L.g.pre = PRE_NONE;
            FL_HIP(c, launch_vpass_generic(L.g, st), "blur vertical pass");
            FL_HIP(c, launch_hpass_generic(L.g, st), "blur horizontal pass");
        }
        c->stats.blur_launches++;
    }
    for (auto &F : fe_launches) {
        ProfileScope ps(c, st, 2);
        if (F.kind == FLGPU_FE_JFIF444) FL_HIP(c, launch_jfif444(d_fjobs, F.base, F.n, F.mw, F.mh, F.rgba, st), "jfif front end");
        else FL_HIP(c, launch_webp420(d_fjobs, c->d_arena, c->gamma_off, F.base, F.n, F.mw, F.mh, F.rgba, st), "webp front end");
        c->stats.frontend_launches++;
    }
    if (!jjobs.empty()) {
        ProfileScope ps(c, st, 2);
        FL_HIP(c, launch_jpeg_encode(d_jjobs, c->d_arena, 0, (uint32_t)jjobs.size(), jpeg_max_blocks, st), "JPEG encode");
        c->stats.frontend_launches++;
    }
    // plain copies for requests that change nothing
    for (size_t i = 0; i < n; ++i) {
        const Work &w = work[i];
        if (w.s1 == S1_NONE && !(w.p->blur_sigma > 0.0f) && w.p->front_end == FLGPU_FE_NONE)
            FL_HIP(c, hipMemcpyAsync(w.final_dst, w.src, w.plan.pixel_bytes, hipMemcpyDeviceToDevice, st), "copy");
    }
    if (slot) { FL_HIP(c, hipEventRecord(slot->done, st), "event record"); slot->busy = true; }
    if (!c->last_done) FL_HIP(c, hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming), "event");
    FL_HIP(c, hipEventRecord(c->last_done, st), "event record");
    c->last_stream = st;

    for (size_t i = 0; i < n; ++i) {
        const flgpu_plan &pl = work[i].plan;
        dsts[i].width = pl.out_w; dsts[i].height = pl.out_h; dsts[i].channels = pl.out_c;
        const uint32_t fe = work[i].p->front_end;
        dsts[i].flags = fe == FLGPU_FE_JPEG ? FLGPU_IMG_ENCODED : (fe != FLGPU_FE_NONE ? FLGPU_IMG_FRONTEND_PLANES : 0u);
        dsts[i].bytes = fe == FLGPU_FE_JPEG ? 0 : pl.out_bytes; // an encoded stream's length is a result word: flgpu_batch_results
    }
    c->last_n = n;
    c->last_has_results = has_results;
    c->last_fe.resize(n);
    for (size_t i = 0; i < n; ++i) c->last_fe[i] = work[i].p->front_end;
    c->stats.images += n;
    c->stats.batches++;
    return FLGPU_OK;
}

// Reads the result words of the batch that was just enqueued on `st` (synchronises) and completes dsts[]:
// the alpha flag of the WebP front end, the length of an encoded stream.
int collect_results(flgpu_ctx *c, size_t n, flgpu_image *dsts, hipStream_t st)
{
    if (!c->last_has_results || n != c->last_n) { FL_HIP(c, hipStreamSynchronize(st), "batch sync"); return FLGPU_OK; }
    FL_HIP(c, c->h_results.reserve(n * 8), "pinned result words");
    FL_HIP(c, hipMemcpyAsync(c->h_results.p, c->d_status.p, n * 8, hipMemcpyDeviceToHost, st), "result words D2H");
    FL_HIP(c, hipStreamSynchronize(st), "batch sync");
    const uint32_t *r = static_cast<const uint32_t *>(c->h_results.p);
    int rc = FLGPU_OK;
    for (size_t i = 0; i < n; ++i) {
        if (c->last_fe[i] == FLGPU_FE_WEBP420 && (r[2 * i] & 1u)) dsts[i].flags |= FLGPU_IMG_HAS_ALPHA;
        if (c->last_fe[i] == FLGPU_FE_JPEG) {
            dsts[i].bytes = r[2 * i + 1];
            if (!r[2 * i + 1]) { c->last_error = "encoded stream does not fit the destination"; rc = FLGPU_ERR_BUFFER_TOO_SMALL; }
        }
    }
    return rc;
}

// Host-memory batch: stage in, run, stage out, wait.
int run_batch_host(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, flgpu_image *dsts)
{
    if (n == 0) return FLGPU_OK;
    if (!srcs || !ps || !dsts) return FLGPU_ERR_INVALID_ARG;
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    std::vector<flgpu_image> dsrc(n), ddst(n);
    std::vector<flgpu_plan> plans(n);
    size_t in_b = 0, out_b = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!srcs[i].data || !dsts[i].data) return FLGPU_ERR_INVALID_ARG;
        int rc = flgpu_plan_output(&ps[i], srcs[i].width, srcs[i].height, srcs[i].channels, &plans[i]);
        if (rc) return rc;
        const uint64_t sb = (uint64_t)srcs[i].width * srcs[i].height * srcs[i].channels;
        if (srcs[i].capacity < sb) return FLGPU_ERR_INVALID_ARG;
        if (dsts[i].capacity < plans[i].out_bytes) return FLGPU_ERR_BUFFER_TOO_SMALL;
        dsrc[i] = srcs[i]; ddst[i] = dsts[i];
        dsrc[i].data = reinterpret_cast<uint8_t *>(in_b); dsrc[i].capacity = sb; in_b += align_up(sb, 256);
        uint64_t ob = plans[i].out_bytes;
        if (ps[i].front_end == FLGPU_FE_JPEG) {
            const uint64_t worst = 1024ull + 2ull * kJpegMaxUnitBytes * 3ull * (plans[i].plane_w / 8u) * (plans[i].plane_h / 8u);
            ob = std::max<uint64_t>(ob, std::min<uint64_t>(dsts[i].capacity, worst));
        }
        ddst[i].data = reinterpret_cast<uint8_t *>(out_b); ddst[i].capacity = ob; out_b += align_up(ob, 256);
    }
    FL_HIP(c, c->d_in.reserve(in_b), "device input staging");
    FL_HIP(c, c->d_out.reserve(out_b), "device output staging");
    FL_HIP(c, c->h_stage_in.reserve(in_b), "pinned input staging");
    FL_HIP(c, c->h_stage_out.reserve(out_b), "pinned output staging");
    hipStream_t st = c->stream;
    for (size_t i = 0; i < n; ++i) {
        const size_t off = reinterpret_cast<size_t>(dsrc[i].data);
        memcpy(static_cast<char *>(c->h_stage_in.p) + off, srcs[i].data, dsrc[i].capacity);
        dsrc[i].data = static_cast<uint8_t *>(c->d_in.p) + off;
        ddst[i].data = static_cast<uint8_t *>(c->d_out.p) + reinterpret_cast<size_t>(ddst[i].data);
    }
    FL_HIP(c, hipMemcpyAsync(c->d_in.p, c->h_stage_in.p, in_b, hipMemcpyHostToDevice, st), "H2D");
    int rc = run_batch_device(c, n, dsrc.data(), ps, false, ddst.data(), st);
    if (rc) return rc;
    FL_HIP(c, hipMemcpyAsync(c->h_stage_out.p, c->d_out.p, out_b, hipMemcpyDeviceToHost, st), "D2H");
    rc = collect_results(c, n, ddst.data(), st);
    for (size_t i = 0; i < n; ++i) {
        const size_t off = static_cast<uint8_t *>(ddst[i].data) - static_cast<uint8_t *>(c->d_out.p);
        memcpy(dsts[i].data, static_cast<char *>(c->h_stage_out.p) + off, std::min<uint64_t>(ddst[i].bytes, ddst[i].capacity));
        dsts[i].width = ddst[i].width; dsts[i].height = ddst[i].height; dsts[i].channels = ddst[i].channels; dsts[i].flags = ddst[i].flags;
        dsts[i].bytes = ddst[i].bytes;
    }
    return rc;
}

// ---- request queue ---------------------------------------------------------

PinBlock pin_acquire(flgpu_ctx *c, size_t bytes)
{
    size_t cap = 64 * 1024;
    while (cap < bytes) cap <<= 1;
    {
        std::lock_guard<std::mutex> g(c->pin_mu);
        auto it = c->pin_free.find(cap);
        if (it != c->pin_free.end()) { PinBlock b{it->second, cap}; c->pin_free.erase(it); return b; }
    }
    PinBlock b;
    (void)hipSetDevice(c->device);
    if (hipHostMalloc(&b.p, cap, hipHostMallocDefault) != hipSuccess) { b.p = nullptr; return b; }
    b.cap = cap;
    return b;
}

void pin_release(flgpu_ctx *c, PinBlock &b)
{
    if (!b.p) return;
    std::lock_guard<std::mutex> g(c->pin_mu);
    c->pin_free.emplace(b.cap, b.p);
    b.p = nullptr;
}

// One flushed batch of queued requests: sources already sit in pinned blocks (copied there by the
// calling threads), results are left in pinned blocks for the callers to copy out.
int run_batch_queued(flgpu_ctx *c, std::vector<Request *> &batch)
{
    const size_t n = batch.size();
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    std::vector<flgpu_image> dsrc(n), ddst(n);
    std::vector<flgpu_params> ps(n);
    size_t in_b = 0, out_b = 0;
    for (size_t i = 0; i < n; ++i) {
        dsrc[i] = *batch[i]->src; ddst[i] = *batch[i]->dst; ps[i] = *batch[i]->p;
        dsrc[i].data = reinterpret_cast<uint8_t *>(in_b); dsrc[i].capacity = batch[i]->src_bytes; in_b += align_up(batch[i]->src_bytes, 256);
        ddst[i].data = reinterpret_cast<uint8_t *>(out_b); ddst[i].capacity = batch[i]->out_bytes; out_b += align_up(batch[i]->out_bytes, 256);
    }
    FL_HIP(c, c->d_in.reserve(in_b), "device input staging");
    FL_HIP(c, c->d_out.reserve(out_b), "device output staging");
    hipStream_t st = c->stream;
    for (size_t i = 0; i < n; ++i) {
        dsrc[i].data = static_cast<uint8_t *>(c->d_in.p) + reinterpret_cast<size_t>(dsrc[i].data);
        ddst[i].data = static_cast<uint8_t *>(c->d_out.p) + reinterpret_cast<size_t>(ddst[i].data);
        FL_HIP(c, hipMemcpyAsync(dsrc[i].data, batch[i]->in.p, batch[i]->src_bytes, hipMemcpyHostToDevice, st), "H2D");
    }
    int rc = run_batch_device(c, n, dsrc.data(), ps.data(), false, ddst.data(), st);
    if (rc) return rc;
    // encoded streams: learn their lengths first, then fetch exactly those bytes (a 300x200 JPEG is ~16 KB of a 183 KB bound)
    bool encoded = false;
    for (size_t i = 0; i < n; ++i) encoded |= ps[i].front_end == FLGPU_FE_JPEG;
    int rrc = FLGPU_OK;
    if (encoded) rrc = collect_results(c, n, ddst.data(), st);
    for (size_t i = 0; i < n; ++i) {
        const uint64_t nb = ps[i].front_end == FLGPU_FE_JPEG ? ddst[i].bytes : batch[i]->out_bytes;
        if (nb) FL_HIP(c, hipMemcpyAsync(batch[i]->out.p, ddst[i].data, nb, hipMemcpyDeviceToHost, st), "D2H");
    }
    if (!encoded) rrc = collect_results(c, n, ddst.data(), st);
    else FL_HIP(c, hipStreamSynchronize(st), "batch sync");
    for (size_t i = 0; i < n; ++i) {
        batch[i]->dst->width = ddst[i].width; batch[i]->dst->height = ddst[i].height;
        batch[i]->dst->channels = ddst[i].channels; batch[i]->dst->flags = ddst[i].flags;
        batch[i]->dst->bytes = ddst[i].bytes;
        if (ps[i].front_end == FLGPU_FE_JPEG && !ddst[i].bytes) batch[i]->status = FLGPU_ERR_BUFFER_TOO_SMALL;
    }
    (void)rrc; // per-request status above: one oversized stream must not fail its batch mates
    return FLGPU_OK;
}

void worker_main(flgpu_ctx *c, flgpu_ctx *lane)
{
    const size_t max_batch = c->cfg.max_batch ? c->cfg.max_batch : 32; // measured: 3 lanes x 32 keeps the PCIe link busiest
    const auto flush = std::chrono::microseconds(c->cfg.flush_timeout_us ? c->cfg.flush_timeout_us : 200);
    for (;;) {
        std::vector<Request *> batch;
        {
            std::unique_lock<std::mutex> lk(c->qmu);
            c->qcv.wait(lk, [&] { return c->stop || (!c->collecting && !c->queue.empty()); });
            if (c->queue.empty()) { if (c->stop) return; continue; }
            if (c->collecting) continue;
            c->collecting = true;
            // a first request arrived: wait for company -- but only while somebody is actually on the way (a caller
            // staging its source), and never beyond the flush timer or a full batch.  A lone caller is served at once.
            const auto deadline = std::chrono::steady_clock::now() + flush;
            while (c->queue.size() < max_batch && !c->stop && c->staging.load(std::memory_order_acquire) > 0) {
                if (c->qcv.wait_until(lk, deadline) == std::cv_status::timeout) break;
            }
            while (!c->queue.empty() && batch.size() < max_batch) { batch.push_back(c->queue.front()); c->queue.pop_front(); }
            c->collecting = false;
        }
        c->qcv.notify_all(); // the next batch may be collected while this one is in flight
        int rc;
        {
            std::lock_guard<std::mutex> g(lane->mu);
            rc = run_batch_queued(lane, batch);
            lane->stats.queue_flushes++;
            if (rc) { std::lock_guard<std::mutex> lk(c->qmu); c->last_error = lane->last_error; }
        }
        {
            std::lock_guard<std::mutex> lk(c->qmu);
            for (Request *r : batch) { if (rc) r->status = rc; r->done = true; }
        }
        c->qdone.notify_all();
    }
}

} // namespace

// ---- C ABI -------------------------------------------------------------------

extern "C" {

flgpu_ctx *flgpu_create(const flgpu_config *cfg, int *status)
{
    auto set = [&](int s) { if (status) *status = s; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set(FLGPU_ERR_NO_DEVICE); return nullptr; }
    flgpu_ctx *c = new (std::nothrow) flgpu_ctx();
    if (!c) { set(FLGPU_ERR_OOM); return nullptr; }
    if (cfg) c->cfg = *cfg;
    int dev = c->cfg.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete c; set(FLGPU_ERR_NO_DEVICE); return nullptr; }
    c->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c; set(FLGPU_ERR_NO_DEVICE); return nullptr;
    }
    c->arena_cap_words = kArenaWords;
    if (const char *aw = getenv("FLGPU_ARENA_WORDS")) { // tests: a small arena exercises the overflow -> reset path
        const long v = atol(aw);
        if (v >= 65536 && (size_t)v <= kArenaWords) c->arena_cap_words = (size_t)v;
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_arena), kArenaWords * 4) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; set(FLGPU_ERR_OOM); return nullptr;
    }
    arena_reset(c);
    set(FLGPU_OK);
    return c;
}

void flgpu_destroy(flgpu_ctx *c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->qmu);
        c->stop = true;
    }
    c->qcv.notify_all();
    for (auto &t : c->workers) t.join();
    for (flgpu_ctx *l : c->lanes) flgpu_destroy(l);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    resolve_pending(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto &s : c->slots) { s.host.release(); s.dev.release(); if (s.done) (void)hipEventDestroy(s.done); }
    if (c->last_done) (void)hipEventDestroy(c->last_done);
    c->d_mid.release(); c->d_tmp_a.release(); c->d_tmp_b.release(); c->d_tmp_o.release(); c->d_status.release(); c->d_in.release(); c->d_out.release();
    c->d_jpeg_coef.release(); c->d_jpeg_off.release(); c->d_jpeg_raw.release();
    c->cmyk_default.dev.release();
    for (auto &kv : c->cmyk_embedded) kv.second.dev.release();
    c->h_stage_in.release(); c->h_stage_out.release();
    for (auto &kv : c->pin_free) (void)hipHostFree(kv.second);
    if (c->d_arena) (void)hipFree(c->d_arena);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int flgpu_transform_batch_device(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, flgpu_image *dsts,
                                 void *hip_stream, uint32_t flags)
{
    if (!c) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    return run_batch_device(c, n, srcs, ps, (flags & FLGPU_BATCH_SAME_PARAMS) != 0, dsts, static_cast<hipStream_t>(hip_stream));
}

int flgpu_transform_batch(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, flgpu_image *dsts)
{
    if (!c) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    return run_batch_host(c, n, srcs, ps, dsts);
}

int flgpu_transform(flgpu_ctx *c, const flgpu_image *src, const flgpu_params *p, flgpu_image *dst)
{
    if (!c || !src || !p || !dst || !src->data || !dst->data) return FLGPU_ERR_INVALID_ARG;
    // validate on the caller's thread so that one bad request cannot fail a shared batch
    flgpu_plan plan;
    int rc = flgpu_plan_output(p, src->width, src->height, src->channels, &plan);
    if (rc) return rc;
    if (src->capacity < (uint64_t)src->width * src->height * src->channels) return FLGPU_ERR_INVALID_ARG;
    if (dst->capacity < plan.out_bytes) return FLGPU_ERR_BUFFER_TOO_SMALL;
    Request r{};
    r.src = src; r.p = p; r.dst = dst;
    r.src_bytes = (uint64_t)src->width * src->height * src->channels;
    r.out_bytes = plan.out_bytes;
    if (p->front_end == FLGPU_FE_JPEG) {
        // a caller that offers more room than the planning bound gets it, up to the worst case of the format
        const uint64_t worst = 1024ull + 2ull * kJpegMaxUnitBytes * 3ull * (plan.plane_w / 8u) * (plan.plane_h / 8u);
        r.out_bytes = std::max<uint64_t>(plan.out_bytes, std::min<uint64_t>(dst->capacity, worst));
    }
    {
        const uint32_t lanes = std::min<uint32_t>(std::max<uint32_t>(c->cfg.queue_lanes ? c->cfg.queue_lanes : 3u, 1u), 8u);
        const uint32_t limit = 4u * lanes * (c->cfg.max_batch ? c->cfg.max_batch : 32u);
        std::unique_lock<std::mutex> lk(c->adm_mu);
        c->adm_cv.wait(lk, [&] { return c->admitted < limit; });
        c->admitted++;
    }
    struct Admission {
        flgpu_ctx *c;
        ~Admission() { { std::lock_guard<std::mutex> lk(c->adm_mu); c->admitted--; } c->adm_cv.notify_one(); }
    } admission{c};
    c->staging.fetch_add(1, std::memory_order_acq_rel);
    // buffers from flgpu_host_alloc are page-locked already: the DMA engine reads / writes them directly, no staging copy
    const bool src_pinned = (src->flags & FLGPU_IMG_PINNED) != 0, dst_pinned = (dst->flags & FLGPU_IMG_PINNED) != 0 && dst->capacity >= r.out_bytes;
    if (src_pinned) r.in = PinBlock{src->data, 0}; else r.in = pin_acquire(c, r.src_bytes);
    if (dst_pinned) r.out = PinBlock{dst->data, 0}; else r.out = pin_acquire(c, r.out_bytes);
    if (!r.in.p || !r.out.p) {
        c->staging.fetch_sub(1, std::memory_order_acq_rel);
        if (!src_pinned) pin_release(c, r.in);
        if (!dst_pinned) pin_release(c, r.out);
        return FLGPU_ERR_OOM;
    }
    if (!src_pinned) memcpy(r.in.p, src->data, r.src_bytes); // on the caller's thread: concurrent callers stage in parallel
    {
        std::unique_lock<std::mutex> lk(c->qmu);
        c->staging.fetch_sub(1, std::memory_order_acq_rel);
        if (c->stop) { lk.unlock(); if (!src_pinned) pin_release(c, r.in); if (!dst_pinned) pin_release(c, r.out); return FLGPU_ERR_SHUTDOWN; }
        if (!c->worker_started) {
            // lanes: child contexts on the same device (cfg.queue_lanes, default 3)
            const uint32_t nl = std::min<uint32_t>(std::max<uint32_t>(c->cfg.queue_lanes ? c->cfg.queue_lanes : 3u, 1u), 8u);
            flgpu_config lc = c->cfg;
            lc.device = c->device;
            lc.queue_lanes = 1;
            for (uint32_t i = 0; i < nl; ++i) {
                int lst = 0;
                flgpu_ctx *l = flgpu_create(&lc, &lst);
                if (!l) break;
                c->lanes.push_back(l);
            }
            if (c->lanes.empty()) { lk.unlock(); if (!src_pinned) pin_release(c, r.in); if (!dst_pinned) pin_release(c, r.out); return FLGPU_ERR_OOM; }
            for (flgpu_ctx *l : c->lanes) c->workers.emplace_back(worker_main, c, l);
            c->worker_started = true;
        }
        c->queue.push_back(&r);
    }
    c->qcv.notify_all();
    {
        std::unique_lock<std::mutex> lk(c->qmu);
        c->qdone.wait(lk, [&] { return r.done; });
    }
    if (r.status == FLGPU_OK && !dst_pinned) memcpy(dst->data, r.out.p, std::min<uint64_t>(dst->bytes, r.out_bytes));
    if (!src_pinned) pin_release(c, r.in);
    if (!dst_pinned) pin_release(c, r.out);
    if (dst_pinned) dst->flags |= FLGPU_IMG_PINNED;
    return r.status;
}

void *flgpu_host_alloc(flgpu_ctx *c, uint64_t bytes)
{
    if (!c || !bytes) return nullptr;
    (void)hipSetDevice(c->device);
    void *p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void flgpu_host_free(flgpu_ctx *c, void *p)
{
    if (!c || !p) return;
    (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

int flgpu_batch_results(flgpu_ctx *c, size_t n, flgpu_image *dsts)
{
    if (!c || (!dsts && n)) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (n != c->last_n) { c->last_error = "flgpu_batch_results: n differs from the last device batch"; return FLGPU_ERR_INVALID_ARG; }
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    return collect_results(c, n, dsts, c->last_stream ? c->last_stream : c->stream);
}

int flgpu_ycck_to_cmyk(flgpu_ctx *c, uint8_t *raw, uint64_t n_pixels)
{
    if (!c || (!raw && n_pixels)) return FLGPU_ERR_INVALID_ARG;
    if (n_pixels == 0) return FLGPU_OK;
    if (n_pixels >= (1ull << 30)) return FLGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const size_t bytes = (size_t)n_pixels * 4;
    FL_HIP(c, c->d_in.reserve(bytes), "device staging");
    FL_HIP(c, hipMemcpyAsync(c->d_in.p, raw, bytes, hipMemcpyHostToDevice, c->stream), "H2D");
    FL_HIP(c, launch_ycck_to_cmyk(static_cast<uint32_t *>(c->d_in.p), n_pixels, c->stream), "ycck kernel");
    FL_HIP(c, hipMemcpyAsync(raw, c->d_in.p, bytes, hipMemcpyDeviceToHost, c->stream), "D2H");
    FL_HIP(c, hipStreamSynchronize(c->stream), "sync");
    return FLGPU_OK;
}

// ---- CMYK -> RGB (reference src/handler.rs:398-493) ----------------------------------------

namespace {

int upload_clut(flgpu_ctx *c, flgpu_ctx::Clut &t)
{
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    FL_HIP(c, t.dev.reserve(t.host.size() * sizeof(uint16_t)), "CLUT alloc");
    FL_HIP(c, hipMemcpyAsync(t.dev.p, t.host.data(), t.host.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream), "CLUT upload");
    FL_HIP(c, hipStreamSynchronize(c->stream), "CLUT sync");
    return FLGPU_OK;
}

constexpr size_t kClutNodes = (size_t)kCmykGrid * kCmykGrid * kCmykGrid * kCmykGrid;

// picks the table for one conversion: the embedded profile's if it can be baked (cached), else the default
int select_clut(flgpu_ctx *c, const uint8_t *icc, uint64_t icc_len, const void **dev)
{
    if (icc && icc_len) {
        const uint64_t h = hash_bytes(icc, icc_len);
        auto it = c->cmyk_embedded.find(h);
        if (it == c->cmyk_embedded.end()) {
            std::vector<uint16_t> nodes;
            if (bake_cmyk_clut(icc, icc_len, nodes) == 0) {
                if (c->cmyk_embedded.size() >= 8) { // evict the least recently used table
                    auto old = c->cmyk_embedded.begin();
                    for (auto i2 = c->cmyk_embedded.begin(); i2 != c->cmyk_embedded.end(); ++i2) if (i2->second.stamp < old->second.stamp) old = i2;
                    FL_HIP(c, hipStreamSynchronize(c->stream), "CLUT evict sync");
                    old->second.dev.release();
                    c->cmyk_embedded.erase(old);
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
    if (!c->has_cmyk_default) { c->last_error = "no CMYK profile configured"; return FLGPU_ERR_UNSUPPORTED; }
    *dev = c->cmyk_default.dev.p;
    return FLGPU_OK;
}

} // namespace

int flgpu_set_cmyk_profile(flgpu_ctx *c, const uint8_t *icc, uint64_t n)
{
    if (!c || !icc || !n) return FLGPU_ERR_INVALID_ARG;
    std::vector<uint16_t> nodes;
    const int rc = bake_cmyk_clut(icc, n, nodes);
    std::lock_guard<std::mutex> g(c->mu);
    if (rc == -2) { c->last_error = "liblcms2.so.2 could not be loaded; bake the table elsewhere and use flgpu_set_cmyk_clut"; return FLGPU_ERR_UNSUPPORTED; }
    if (rc) { c->last_error = "not a usable CMYK ICC profile"; return FLGPU_ERR_INVALID_ARG; }
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    FL_HIP(c, hipStreamSynchronize(c->stream), "CLUT swap sync");
    c->cmyk_default.host.swap(nodes);
    const int u = upload_clut(c, c->cmyk_default);
    c->has_cmyk_default = (u == FLGPU_OK);
    if (u == FLGPU_OK) c->stats.cmyk_tables_baked++;
    return u;
}

int flgpu_set_cmyk_clut(flgpu_ctx *c, uint32_t grid, const uint16_t *rgb_nodes)
{
    if (!c || !rgb_nodes) return FLGPU_ERR_INVALID_ARG;
    if (grid != kCmykGrid) return FLGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    FL_HIP(c, hipStreamSynchronize(c->stream), "CLUT swap sync");
    c->cmyk_default.host.assign(kClutNodes * 4, 0);
    for (size_t i = 0; i < kClutNodes; ++i)
        for (int k = 0; k < 3; ++k) c->cmyk_default.host[i * 4 + k] = rgb_nodes[i * 3 + k];
    const int u = upload_clut(c, c->cmyk_default);
    c->has_cmyk_default = (u == FLGPU_OK);
    return u;
}

int flgpu_get_cmyk_clut(flgpu_ctx *c, uint16_t *rgb_nodes, uint64_t capacity_entries, uint32_t *grid)
{
    if (!c || !grid) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->has_cmyk_default) { c->last_error = "no CMYK profile configured"; return FLGPU_ERR_UNSUPPORTED; }
    *grid = kCmykGrid;
    if (!rgb_nodes) return FLGPU_OK;
    if (capacity_entries < kClutNodes * 3) return FLGPU_ERR_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < kClutNodes; ++i)
        for (int k = 0; k < 3; ++k) rgb_nodes[i * 3 + k] = c->cmyk_default.host[i * 4 + k];
    return FLGPU_OK;
}

int flgpu_cmyk_to_rgb_device(flgpu_ctx *c, const void *d_cmyk, void *d_rgb, uint64_t n_pixels, uint32_t flags, void *hip_stream)
{
    if (!c || ((!d_cmyk || !d_rgb) && n_pixels)) return FLGPU_ERR_INVALID_ARG;
    if (n_pixels == 0) return FLGPU_OK;
    if (n_pixels >= (1ull << 32)) return FLGPU_ERR_UNSUPPORTED;
    if (((uintptr_t)d_cmyk & 15u) || ((uintptr_t)d_rgb & 3u)) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const void *clut = nullptr;
    const int rc = select_clut(c, nullptr, 0, &clut);
    if (rc) return rc;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    FL_HIP(c, launch_cmyk_clut(d_cmyk, d_rgb, clut, kCmykGrid, n_pixels, (flags & FLGPU_CMYK_INPUT_YCCK) != 0, st), "CMYK kernel");
    c->stats.cmyk_pixels += n_pixels;
    return FLGPU_OK;
}

int flgpu_cmyk_to_rgb(flgpu_ctx *c, const uint8_t *cmyk, uint64_t n_pixels, uint8_t *rgb, const uint8_t *embedded_icc,
                      uint64_t icc_len, uint32_t flags)
{
    if (!c || ((!cmyk || !rgb) && n_pixels)) return FLGPU_ERR_INVALID_ARG;
    if (n_pixels == 0) return FLGPU_OK;
    if (n_pixels >= (1ull << 30)) return FLGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const void *clut = nullptr;
    const int rc = select_clut(c, embedded_icc, icc_len, &clut);
    if (rc) return rc;
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

int flgpu_cmyk_bake_available(void) { return cmyk_bake_available() ? 1 : 0; }

int flgpu_export_tables(flgpu_ctx *c, void **device_ptr, uint64_t *bytes)
{
    if (!c || !device_ptr || !bytes) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    int rc = arena_flush(c, c->stream);
    if (rc) return rc;
    FL_HIP(c, hipStreamSynchronize(c->stream), "table sync");
    *device_ptr = c->d_arena;
    *bytes = (uint64_t)c->h_arena.size() * 4;
    return FLGPU_OK;
}

int flgpu_copy_tables(flgpu_ctx *c, void *dst_device, uint64_t capacity, uint64_t *bytes)
{
    if (!c || !dst_device || !bytes) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    int rc = arena_flush(c, c->stream);
    if (rc) return rc;
    *bytes = (uint64_t)c->h_arena.size() * 4;
    if (*bytes > capacity) return FLGPU_ERR_BUFFER_TOO_SMALL;
    FL_HIP(c, hipMemcpyAsync(dst_device, c->d_arena, *bytes, hipMemcpyDeviceToDevice, c->stream), "table copy");
    FL_HIP(c, hipStreamSynchronize(c->stream), "table sync");
    return FLGPU_OK;
}

int flgpu_import_tables(flgpu_ctx *c, const void *src_device, uint64_t bytes)
{
    if (!c || !src_device) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    int rc = arena_flush(c, c->stream);
    if (rc) return rc;
    if (bytes != (uint64_t)c->h_arena.size() * 4) return FLGPU_ERR_INVALID_ARG; // layouts differ: not the same plan
    FL_HIP(c, hipMemcpyAsync(c->d_arena, src_device, bytes, hipMemcpyDeviceToDevice, c->stream), "table import");
    FL_HIP(c, hipStreamSynchronize(c->stream), "table sync");
    return FLGPU_OK;
}

int flgpu_get_stats(flgpu_ctx *c, flgpu_stats *out)
{
    if (!c || !out) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    resolve_pending(c);
    *out = c->stats;
    for (flgpu_ctx *l : c->lanes) { // queued requests run on the lanes' child contexts
        flgpu_stats ls;
        if (flgpu_get_stats(l, &ls) != FLGPU_OK) continue;
        out->images += ls.images; out->batches += ls.batches; out->queue_flushes += ls.queue_flushes; out->tables_built += ls.tables_built;
        out->resample_launches += ls.resample_launches; out->resample_ms += ls.resample_ms;
        out->resample_src_bytes += ls.resample_src_bytes; out->resample_dst_bytes += ls.resample_dst_bytes;
        out->generic_launches += ls.generic_launches; out->blur_launches += ls.blur_launches; out->blur_ms += ls.blur_ms;
        out->frontend_launches += ls.frontend_launches; out->frontend_ms += ls.frontend_ms;
        out->cmyk_pixels += ls.cmyk_pixels; out->cmyk_tables_baked += ls.cmyk_tables_baked;
    }
    return FLGPU_OK;
}

int flgpu_reset_stats(flgpu_ctx *c)
{
    if (!c) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    resolve_pending(c);
    c->stats = flgpu_stats{};
    for (flgpu_ctx *l : c->lanes) (void)flgpu_reset_stats(l);
    return FLGPU_OK;
}

const char *flgpu_last_error(flgpu_ctx *c) { return c ? c->last_error.c_str() : ""; }

} // extern "C"
