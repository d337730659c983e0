// fl_context.cpp -- context lifetime, the read-only table arena and its caches, statistics and table export / import.
// The batch planner / launcher is fl_batch.cpp, the request queue and the sharding across devices fl_queue.cpp,
// the CMYK tables fl_cmyk_ctx.cpp (see fl_context.h).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "fl_context.h"

using namespace fl;

namespace fl {

hipError_t DeviceBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    // Twice what is asked for: hipFree + hipMalloc wait for the whole device (every lane's stream), 10-20 ms for everything in flight, and a
    // buffer that grew to exactly each new largest batch did so dozens of times in a server's first thousand requests
    // (tools/experiments/jh_tail.sh: the slow requests of a run were the batches that met a growth).  288 GB of HBM pay for the slack.
    size_t want = std::max(bytes > ((size_t)1 << 34) ? bytes : 2 * bytes, (size_t)1 << 20);
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && want > bytes + ((size_t)1 << 20)) { // no room for the slack: what is asked for
        (void)hipGetLastError();
        want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
}

hipError_t PinnedBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = std::max(bytes > ((size_t)1 << 32) ? bytes : bytes + bytes / 2, (size_t)1 << 16); // (page-locked memory is dearer: half as much again)
    want = (want + 4095) & ~(size_t)4095;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess && want > bytes + 4096) {
        (void)hipGetLastError();
        want = (bytes + 4095) & ~(size_t)4095;
        e = hipHostMalloc(&p, want, hipHostMallocDefault);
    }
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
}

// ---- arena ---------------------------------------------------------------

// Appends words, returns the word offset; 0 is never a valid offset (word 0 is a sentinel).
uint32_t arena_append(flgpu_ctx *c, const void *data, size_t words, size_t align_words)
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
    c->mfma_plans.clear();
    c->wtile_plans.clear();
    c->blur_plans.clear();
    c->tile_vplans.clear();
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

// Workgroups of one geometry for a given number of bands (runs of whole 16-row tiles), longest K-block ranges first is the
// launcher's business; the list is cached per band count.
const std::vector<MfmaItem> &MfmaPlan::items_for(uint32_t nbands)
{
    const uint32_t nt = (uint32_t)tiles.size(), nb = std::max(1u, std::min(nbands, nt));
    auto it = items_by_bands.find(nb);
    if (it != items_by_bands.end()) return it->second;
    std::vector<MfmaItem> items;
    for (uint32_t soff : strip_offs)
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t t0 = (uint32_t)((uint64_t)nt * b / nb), t1 = (uint32_t)((uint64_t)nt * (b + 1) / nb);
            if (t1 <= t0) continue;
            MfmaItem mi{};
            mi.vplan_off = vplan_off; mi.strip_off = soff; mi.tile0 = t0; mi.tile1 = t1;
            mi.kb0 = tiles[t0].kb_first; mi.kb1 = tiles[t1 - 1].kb_last + 1u;
            items.push_back(mi);
        }
    return items_by_bands.emplace(nb, std::move(items)).first->second;
}

static_assert(sizeof(WtItem) == sizeof(MfmaItem) && alignof(WtItem) == alignof(MfmaItem), "the two kernels' workgroup records share the batch's item array");

void WtPlan::items_for(uint32_t nbands, uint32_t job, std::vector<MfmaItem> &out) const
{
    const uint32_t nb = std::max(1u, std::min(nbands, n_mt));
    for (uint32_t s = 0; s < n_strips; ++s)
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t t0 = (uint32_t)((uint64_t)n_mt * b / nb), t1 = (uint32_t)((uint64_t)n_mt * (b + 1) / nb);
            if (t1 <= t0) continue;
            WtItem wi{};
            wi.job = job; wi.plan_off = off; wi.strip = s; wi.mt0 = t0; wi.mt1 = t1;
            MfmaItem mi;
            memcpy(&mi, &wi, sizeof(mi));
            out.push_back(mi);
        }
}

// Tables of the window-tile kernel for one geometry (resample or blur axes), or ok = false.
WtPlan *get_wtile_plan(flgpu_ctx *c, const AxisKey &vk, const HostAxis &va, const AxisKey &hk, const HostAxis &ha,
                       uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t cs)
{
    MfmaPlanKey key{vk, hk, cx, cy, cw, ch, cs, 0u};
    auto it = c->wtile_plans.find(key);
    if (it != c->wtile_plans.end()) return &it->second;
    WtPlan plan;
    HostWtPlan hp;
    build_wtile_plan(va, ha, cs, cx, cy, cw, ch, hp);
    if (hp.ok) {
        if (c->h_arena.size() + hp.blk.size() + 2048 > c->arena_cap_words) {
            static thread_local WtPlan full_plan;
            full_plan = WtPlan();
            full_plan.arena_full = hp.blk.size() + 4096 <= c->arena_cap_words; // (tables larger than the whole arena: never)
            if (full_plan.arena_full) return &full_plan;
            hp.ok = false;
        }
    }
    if (hp.ok) {
        plan.off = arena_append(c, hp.blk.data(), hp.blk.size());
        plan.ok = plan.off != 0;
        plan.nslot = hp.nslot; plan.nkmax = hp.nkmax; plan.n_mt = hp.n_mt; plan.n_strips = hp.n_strips; plan.lds_bytes = hp.lds_bytes;
    }
    if (c->dbg->on(DBG_DEBUG_MFMA))
        fprintf(stderr, "window-tile plan %ux%u -> rows [%u,+%u) cols [%u,+%u) x %u: ok %d, M-tiles %u, strips %u, registers %u x %u, LDS %u, %zu words\n",
                ha.in_size, va.in_size, cy, ch, cx, cw, cs, (int)plan.ok, hp.n_mt, hp.n_strips, hp.nslot, hp.nkmax, hp.lds_bytes, hp.blk.size());
    auto res = c->wtile_plans.emplace(key, plan);
    return &res.first->second;
}

// Tables of the matrix-pipe kernel for one geometry, or ok = false if the kernel cannot or should not take it.
MfmaPlan *get_mfma_plan(flgpu_ctx *c, const AxisKey &vk, const HostAxis &va, const AxisKey &hk, const HostAxis &ha,
                        uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t cs, MfmaArith arith)
{
    MfmaPlanKey key{vk, hk, cx, cy, cw, ch, cs, (uint32_t)arith};
    auto it = c->mfma_plans.find(key);
    if (it != c->mfma_plans.end()) return &it->second;
    MfmaPlan plan;
    HostMfmaPlan hp;
    choose_mfma_plan(va, ha, cs, cx, cy, cw, ch, hp, arith);
    bool ok = hp.ok;
    plan.full = arith == MFMA_ARITH_FULL;
    if (ok) {
        plan.wide = hp.wide;
        for (auto &S : hp.strips) plan.max_nout = std::max(plan.max_nout, S.hdr.nout);
        if (plan.full) {
            // full-width arithmetic: the LDS operand area is whatever the layout's output tiles leave (69 operands, 49 in the wide and 76 in the compact
            // layout for strips of at most 300 outputs), and each strip whose distinct operands fit reads them from there -- the
            // others from the L2.  (With 24-bit weights the low digit's operands repeat only where the f32 sample positions have
            // run out of fraction bits: the right-hand strips of 1080p -> 300 columns have 33 and 49 operands, the leftmost 69.)
            plan.compact = !plan.wide && plan.max_nout <= kMfmaMaxStripOutputsCompact;
            const uint32_t cap = mfma_lds_operand_capacity(plan.wide ? 1 : plan.compact ? 2 : 0);
            for (auto &S : hp.strips) S.hdr.lds_ops = S.hdr.n_ops <= cap ? 1u : 0u;
            plan.ops_in_lds = false;
        } else {
            plan.ops_in_lds = !hp.wide;
            for (auto &S : hp.strips)
                if (S.hdr.n_ops > kMfmaLdsOperands) plan.ops_in_lds = false;
            if (mfma_lds_bytes(plan.max_nout, plan.ops_in_lds, plan.wide) > 160 * 1024) ok = false;
        }
        // (Round 2 kept geometries with BOTH handicaps -- four strips where 2.8 would do, ~100 distinct operands read from the L2
        // instead of LDS -- on the streaming kernel, which was as fast there.  With the tile stage requesting its operands two
        // units ahead the matrix-pipe kernel wins on every one of them: 1080p -> 256x144 1.88 vs 1.99 ms, 512x288 2.33 vs 2.95,
        // 640x360 3.15 vs 4.82 (tools/experiments/sweep_mfma_always.py, profiles/r03_resample_sweep.txt); the rule is gone.)
    }
    if (ok) { // all or nothing: whether a geometry gets this kernel must not depend on how full the arena happens to be
        size_t need = sizeof(MfmaVPlan) / 4 + hp.vmeta.size() + hp.vw.size() + 64;
        for (auto &S : hp.strips) need += sizeof(MfmaStrip) / 4 + S.ctab.size() + S.ops.size() + 64;
        if (c->h_arena.size() + need + 2048 > c->arena_cap_words) {
            static thread_local MfmaPlan full_plan;
            full_plan = MfmaPlan();
            full_plan.arena_full = need + 4096 <= c->arena_cap_words; // (tables larger than the whole arena: the geometry never gets the kernel)
            if (full_plan.arena_full) return &full_plan;
            ok = false;
        }
    }
    if (ok) {
        MfmaVPlan vp{};
        vp.ntiles = hp.ntiles; vp.nkb = hp.nkb; vp.y0 = hp.y0; vp.rows = hp.rows; vp.tail = hp.tail; vp.nterms = plan.full ? 3u : 2u;
        plan.vplan_off = arena_append(c, nullptr, sizeof(MfmaVPlan) / 4);
        vp.meta_off = arena_append(c, hp.vmeta.data(), hp.vmeta.size());
        vp.w_off = arena_append(c, hp.vw.data(), hp.vw.size());
        if (!plan.vplan_off || !vp.meta_off || !vp.w_off) ok = false;
        else memcpy(c->h_arena.data() + plan.vplan_off, &vp, sizeof(vp));
    }
    if (ok) {
        plan.tiles = hp.tiles;
        for (auto &S : hp.strips) {
            MfmaStrip sh = S.hdr;
            const uint32_t soff = arena_append(c, nullptr, sizeof(MfmaStrip) / 4);
            sh.ctab_off = arena_append(c, S.ctab.data(), S.ctab.size());
            sh.ops_off = arena_append(c, S.ops.data(), S.ops.size());
            if (!soff || !sh.ctab_off || !sh.ops_off) { ok = false; break; }
            memcpy(c->h_arena.data() + soff, &sh, sizeof(sh));
            plan.strip_offs.push_back(soff);
        }
    }
    plan.ok = ok;
    if (c->dbg->on(DBG_DEBUG_MFMA)) {
        fprintf(stderr, "mfma plan %ux%u rows [%u,+%u) cols [%u,+%u): ok %d, tiles %u, K-blocks %u, max_nout %u, operands in LDS %d, wide %d, compact %d, full-width arithmetic %d;",
                ha.in_size, va.in_size, cy, ch, cx, cw, (int)ok, hp.ntiles, hp.nkb, plan.max_nout, (int)plan.ops_in_lds, (int)plan.wide, (int)plan.compact, (int)plan.full);
        for (auto &S : hp.strips) fprintf(stderr, " strip [%u,%u) byte0 %u hs %u ops %u%s slots %u", S.hdr.x0, S.hdr.x1, S.hdr.byte0, S.hdr.hs, S.hdr.n_ops, S.hdr.lds_ops ? " (LDS)" : "", S.hdr.slots);
        fprintf(stderr, "\n");
    }
    auto res = c->mfma_plans.emplace(key, std::move(plan));
    return &res.first->second;
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

const char *const kDebugKeyNames[DBG_COUNT] = {
    "no_mfma", "force_generic", "no_wtile", "wtile_blur_always", "wtile_first", "mfma_arith", "force_bands", "no_tile", "no_place4",
    "host_huffman", "device_huffman_always", "device_huffman_min_bytes", "mfma_spin_limit", "debug_mfma", "debug_jh",
};

DebugSwitches::DebugSwitches()
{
    for (auto &x : v) x.store(0, std::memory_order_relaxed);
    v[DBG_DEVICE_HUFFMAN_MIN_BYTES].store(16384, std::memory_order_relaxed); // small files are decoded faster by the thread that holds them than by six kernel launches
    v[DBG_MFMA_SPIN_LIMIT].store((int64_t)kMfmaDefaultSpinLimit, std::memory_order_relaxed);
    for (uint32_t k = 0; k < DBG_COUNT; ++k) initial[k] = v[k].load(std::memory_order_relaxed);
}

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("FLGPU_ROCTX");
        if (!e || e[0] != '1') return;
        void *lib = nullptr;
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx &roctx() { static Roctx r; return r; }
} // namespace

RoctxRange::RoctxRange(const char *name) { if (roctx().push) { roctx().push(name); on = true; } }
RoctxRange::~RoctxRange() { if (on) roctx().pop(); }

ProfileScope::ProfileScope(flgpu_ctx *c_, hipStream_t st_, int kind_) : c(c_), st(st_), kind(kind_)
{
    if (!c->cfg.profile) return;
    if (c->pending.size() > 2048) resolve_pending(c);
    a = get_event(c); b = get_event(c);
    if (a && b) (void)hipEventRecord(a, st);
}

ProfileScope::~ProfileScope()
{
    if (a && b) { (void)hipEventRecord(b, st); c->pending.push_back({a, b, kind}); }
}

// A context bound to one device with its own stream, arena and scratch: a queue lane or the shard of a device.
static flgpu_ctx *create_on_device(const flgpu_config &cfg, int dev, int *status, const std::shared_ptr<DebugSwitches> &dbg, size_t arena_words)
{
    auto set = [&](int s) { if (status) *status = s; };
    flgpu_ctx *c = new (std::nothrow) flgpu_ctx();
    if (!c) { set(FLGPU_ERR_OOM); return nullptr; }
    c->cfg = cfg;
    c->cfg.device = dev;
    c->cfg.n_devices = 0;
    c->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c; set(FLGPU_ERR_NO_DEVICE); return nullptr;
    }
    c->dbg = dbg;
    c->arena_cap_words = arena_words;
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) c->cu_count = (uint32_t)n; }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_arena), kArenaWords * 4) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; set(FLGPU_ERR_OOM); return nullptr;
    }
    arena_reset(c);
    c->lanes.reserve(FLGPU_MAX_DEVICES * 8); // published entries are never moved: readers walk [0, n_lanes) without the queue lock
    set(FLGPU_OK);
    return c;
}

flgpu_ctx *create_child(flgpu_ctx *parent, int device)
{
    flgpu_config lc = parent->cfg;
    lc.queue_lanes = 1;
    int st = 0;
    return create_on_device(lc, device, &st, parent->dbg, parent->arena_cap_words);
}

} // namespace fl

// ---- C ABI -------------------------------------------------------------------

extern "C" {

flgpu_ctx *flgpu_create(const flgpu_config *cfg, int *status)
{
    auto set = [&](int s) { if (status) *status = s; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set(FLGPU_ERR_NO_DEVICE); return nullptr; }
    flgpu_config c0{};
    if (cfg) c0 = *cfg;
    if (c0.n_devices > FLGPU_MAX_DEVICES) { set(FLGPU_ERR_INVALID_ARG); return nullptr; }
    std::vector<int> devs;
    if (c0.n_devices >= 2) {
        for (uint32_t k = 0; k < c0.n_devices; ++k) {
            if (c0.devices[k] < 0 || c0.devices[k] >= ndev) { set(FLGPU_ERR_NO_DEVICE); return nullptr; }
            devs.push_back(c0.devices[k]);
        }
    }
    int dev = devs.empty() ? (c0.n_devices == 1 ? c0.devices[0] : c0.device) : devs[0];
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { set(FLGPU_ERR_NO_DEVICE); return nullptr; }
    int st = 0;
    // The one place the library reads the process environment: the switches' start values (fl_context.h DebugSwitches) and the two
    // creation-time knobs (FLGPU_ARENA_WORDS: a small table arena, tests of the overflow -> reset path; FLGPU_ROCTX, read with the
    // first range).  Everything later goes through flgpu_debug_set.
    auto dbg = std::make_shared<DebugSwitches>();
    for (uint32_t k = 0; k < DBG_COUNT; ++k) {
        std::string name = "FLGPU_";
        for (const char *q = kDebugKeyNames[k]; *q; ++q) name.push_back((char)toupper((unsigned char)*q));
        if (const char *e = getenv(name.c_str())) dbg->v[k].store(k == DBG_MFMA_ARITH ? (e[0] == 'p' || e[0] == '1') : strtoll(e, nullptr, 10), std::memory_order_relaxed);
        dbg->initial[k] = dbg->get((DebugKey)k);
    }
    size_t arena_words = kArenaWords;
    if (const char *aw = getenv("FLGPU_ARENA_WORDS")) {
        const long v = atol(aw);
        if (v >= 65536 && (size_t)v <= kArenaWords) arena_words = (size_t)v;
    }
    (void)roctx(); // (FLGPU_ROCTX: read here, once, not with the first batch)
    flgpu_ctx *c = create_on_device(c0, dev, &st, dbg, arena_words);
    if (!c) { set(st); return nullptr; }
    c->cfg.n_devices = (uint32_t)devs.size();
    c->devices = devs;
    c->inbox.resize(std::max<size_t>(devs.size(), 1));
    for (int d : devs) { // one shard context per device of the node (batch entry points; the queue lanes come on demand)
        flgpu_ctx *s = create_child(c, d);
        if (!s) { flgpu_destroy(c); set(FLGPU_ERR_OOM); return nullptr; }
        c->shard_ctx.push_back(s);
    }
    set(FLGPU_OK);
    return c;
}

void flgpu_destroy(flgpu_ctx *c)
{
    if (!c) return;
    stop_queue(c);
    for (flgpu_ctx *s : c->shard_ctx) flgpu_destroy(s);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    resolve_pending(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto &s : c->slots) { s.host.release(); s.dev.release(); if (s.done) (void)hipEventDestroy(s.done); if (s.uploaded) (void)hipEventDestroy(s.uploaded); }
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->last_done) (void)hipEventDestroy(c->last_done);
    c->d_mid.release(); c->d_tmp_a.release(); c->d_tmp_b.release(); c->d_tmp_o.release(); c->d_tmp_al.release(); c->d_in.release(); c->d_out.release();
    c->d_jpeg_coef.release(); c->d_jpeg_off.release(); c->d_jpeg_raw.release();
    c->d_dec.release(); c->d_decjobs.release(); c->h_decjobs.release();
    release_cmyk(c);
    c->h_results.release();
    c->h_stage_in.release(); c->h_stage_out.release();
    for (auto &kv : c->pin_free) (void)hipHostFree(kv.second);
    if (c->d_arena) (void)hipFree(c->d_arena);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

uint32_t flgpu_devices(flgpu_ctx *c, int32_t *devices, uint32_t cap)
{
    if (!c) return 0;
    if (c->devices.empty()) { if (devices && cap) devices[0] = c->device; return 1; }
    for (uint32_t k = 0; k < c->devices.size() && k < cap && devices; ++k) devices[k] = c->devices[k];
    return (uint32_t)c->devices.size();
}

void *flgpu_host_alloc(flgpu_ctx *c, uint64_t bytes)
{
    if (!c || !bytes) return nullptr;
    (void)hipSetDevice(c->device);
    void *p = nullptr;
    // portable: a multi-device context hands the buffer to whichever device runs the request's shard
    return hipHostMalloc(&p, bytes, c->devices.size() > 1 ? hipHostMallocPortable : hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void flgpu_host_free(flgpu_ctx *c, void *p)
{
    if (!c || !p) return;
    (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

int flgpu_export_tables(flgpu_ctx *c, void **device_ptr, uint64_t *bytes)
try {
    if (!c || !device_ptr || !bytes) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    int rc = arena_flush(c, c->stream);
    if (rc) return rc;
    FL_HIP(c, hipStreamSynchronize(c->stream), "table sync");
    *device_ptr = c->d_arena;
    *bytes = (uint64_t)c->h_arena.size() * 4;
    return FLGPU_OK;
} FL_ABI_CATCH

int flgpu_copy_tables(flgpu_ctx *c, void *dst_device, uint64_t capacity, uint64_t *bytes)
try {
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
} FL_ABI_CATCH

int flgpu_import_tables(flgpu_ctx *c, const void *src_device, uint64_t bytes)
try {
    if (!c || !src_device) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    int rc = arena_flush(c, c->stream);
    if (rc) return rc;
    if (bytes != (uint64_t)c->h_arena.size() * 4) return FLGPU_ERR_INVALID_ARG; // layouts differ: not the same plan
    FL_HIP(c, hipMemcpyAsync(c->d_arena, src_device, bytes, hipMemcpyDeviceToDevice, c->stream), "table import");
    FL_HIP(c, hipStreamSynchronize(c->stream), "table sync");
    return FLGPU_OK;
} FL_ABI_CATCH

static void add_stats(flgpu_stats *out, const flgpu_stats &ls)
{
    out->images += ls.images; out->batches += ls.batches; out->queue_flushes += ls.queue_flushes; out->tables_built += ls.tables_built;
    out->resample_launches += ls.resample_launches; out->resample_ms += ls.resample_ms;
    out->resample_src_bytes += ls.resample_src_bytes; out->resample_dst_bytes += ls.resample_dst_bytes;
    out->generic_launches += ls.generic_launches; out->blur_launches += ls.blur_launches; out->blur_ms += ls.blur_ms;
    out->frontend_launches += ls.frontend_launches; out->frontend_ms += ls.frontend_ms;
    out->cmyk_pixels += ls.cmyk_pixels; out->cmyk_tables_baked += ls.cmyk_tables_baked;
    out->jpeg_sources += ls.jpeg_sources; out->jpeg_file_bytes += ls.jpeg_file_bytes; out->jpeg_upload_bytes += ls.jpeg_upload_bytes;
    out->mfma_launches += ls.mfma_launches;
    out->jpeg_device_huffman += ls.jpeg_device_huffman; out->jpeg_device_huffman_retries += ls.jpeg_device_huffman_retries;
    out->wtile_launches += ls.wtile_launches;
}

int flgpu_get_stats(flgpu_ctx *c, flgpu_stats *out)
{
    if (!c || !out) return FLGPU_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> g(c->mu);
        (void)hipSetDevice(c->device);
        resolve_pending(c);
        *out = c->stats;
    }
    // queued requests run on the lanes' child contexts, batch calls of a multi-device context on the shard contexts
    const size_t nl = c->n_lanes.load(std::memory_order_acquire);
    for (size_t i = 0; i < nl; ++i) { flgpu_stats ls; if (flgpu_get_stats(c->lanes[i], &ls) == FLGPU_OK) add_stats(out, ls); }
    for (flgpu_ctx *s : c->shard_ctx) { flgpu_stats ls; if (flgpu_get_stats(s, &ls) == FLGPU_OK) add_stats(out, ls); }
    return FLGPU_OK;
}

int flgpu_reset_stats(flgpu_ctx *c)
{
    if (!c) return FLGPU_ERR_INVALID_ARG;
    {
        std::lock_guard<std::mutex> g(c->mu);
        (void)hipSetDevice(c->device);
        resolve_pending(c);
        c->stats = flgpu_stats{};
    }
    const size_t nl = c->n_lanes.load(std::memory_order_acquire);
    for (size_t i = 0; i < nl; ++i) (void)flgpu_reset_stats(c->lanes[i]);
    for (flgpu_ctx *s : c->shard_ctx) (void)flgpu_reset_stats(s);
    return FLGPU_OK;
}

int flgpu_debug_set(flgpu_ctx *c, const char *key, int64_t value)
{
    if (!c || !key || !c->dbg) return FLGPU_ERR_INVALID_ARG;
    if (!strcmp(key, "reset")) {
        for (uint32_t k = 0; k < fl::DBG_COUNT; ++k) c->dbg->v[k].store(c->dbg->initial[k], std::memory_order_relaxed);
        return FLGPU_OK;
    }
    for (uint32_t k = 0; k < fl::DBG_COUNT; ++k)
        if (!strcmp(key, fl::kDebugKeyNames[k])) { c->dbg->v[k].store(value, std::memory_order_relaxed); return FLGPU_OK; }
    c->set_error(std::string("flgpu_debug_set: unknown key ") + key);
    return FLGPU_ERR_INVALID_ARG;
}

int flgpu_debug_get(flgpu_ctx *c, const char *key, int64_t *value)
{
    if (!c || !key || !value || !c->dbg) return FLGPU_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < fl::DBG_COUNT; ++k)
        if (!strcmp(key, fl::kDebugKeyNames[k])) { *value = c->dbg->get((fl::DebugKey)k); return FLGPU_OK; }
    return FLGPU_ERR_INVALID_ARG;
}

const char *flgpu_last_error(flgpu_ctx *c)
{
    static thread_local std::string copy; // the context's string may change under another thread: hand out a per-thread copy
    if (!c) return "";
    copy = c->get_error();
    return copy.c_str();
}

} // extern "C"
