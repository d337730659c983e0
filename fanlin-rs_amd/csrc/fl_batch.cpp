// fl_batch.cpp -- the batch planner and launcher: geometry, buffer chain, table lookups, descriptor staging and kernel
// launches of one batch on ONE device context (reference order of operations: src/handler.rs:221-278), the read-back of
// per-image result words, and the host-memory batch built on top of it.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <memory>

#include "fl_context.h"

using namespace fl;

namespace {

// ---- batch execution -------------------------------------------------------

enum Stage1Kind { S1_NONE = 0, S1_PLACE = 1, S1_GENERIC = 2, S1_STREAM = 3, S1_NEAREST = 4, S1_MFMA = 5, S1_TILE = 6, S1_WTILE = 7 };
constexpr uint32_t kBlurWtileKind = 0x1000u; // blur group key: the window-tile kernel instead of blur_tile_kernel (| its register split)

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
    const MfmaPlan *mplan = nullptr;
    const std::vector<MfmaItem> *mitems = nullptr; // its workgroups for the band count of this batch
    bool unaligned = false;
    size_t jpeg_coef_off = 0, jpeg_off_off = 0, jpeg_raw_off = 0; // FE_JPEG scratch (bytes)
    uint32_t jpeg_tab = 0;
    uint32_t orient = 0, raw_w = 0, raw_h = 0; // EXIF orientation pre-pass (2..8), source size before it
    size_t orient_off = 0;
    uint32_t tile_w = 0;        // S1_TILE: output columns per tile (power of two)
    uint32_t tile_vplan = 0;    // S1_TILE: arena offset of the vertical band tables (build_tile_vplan)
    size_t align_off = 0;       // misaligned device source of a matrix-pipe geometry: offset of its aligned copy in d_tmp_al
    bool align_copy = false;
    const uint8_t *raw_src = nullptr;
    const WtPlan *wplan = nullptr;   // S1_WTILE: the window-tile matrix-pipe kernel's tables
    const WtPlan *bwplan = nullptr;  // ... for the blur stage, where that kernel serves it
    bool luma_mid = false;           // grey picture on a grey frame + blur on that kernel: stage 1 leaves the UNFRAMED Luma8 picture, the blur reads it through a virtual frame and writes Rgba8
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

// Resamples the window-tile matrix-pipe kernel takes from the f32 vector kernels: every one its planner accepts.  (The first version
// measured equal to the tiled kernel on up-scales and the rule kept those there; after the kernel's tuning it is ahead on them as
// well -- 1080p -> 2000x1000 1.23 vs 1.48 ms per 128, 720p -> 1600x900 1.64 vs 1.97 per 256, 1080p -> 1600x900 1.02 vs 1.77 per 128,
// thumbnails 3.4 vs 3.55 per 8,192; profiles/r04_wtile_experiments.txt.)
bool wtile_resample_wanted(const Work &)
{
    return true;
}

// Blurs the window-tile matrix-pipe kernel takes: every one its planner accepts.  (Round 4 left the one-channel shortcut -- a grey
// picture on a grey frame, config 2 -- to the vector kernel, which filters one byte column in four there: the matrix kernel would have
// filtered all four.  Round 5: stage 1 leaves the unframed Luma8 picture, the matrix kernel filters ONE channel, reading the frame's
// rows and columns as the fill value, and expands to Rgba8 in its store -- Work::luma_mid.)
bool wtile_blur_wanted(const DebugSwitches &, const Work &)
{
    return true;
}

// Row bands per picture for the window-tile kernel: small batches are cut so that the chip still sees a few hundred workgroups
// (the result does not depend on the cut: every M-tile is computed from the same rows by the same instructions).
uint32_t wtile_bands(const DebugSwitches &dbg, const WtPlan &p, size_t pictures)
{
    if (const int64_t b = dbg.get(DBG_FORCE_BANDS)) return (uint32_t)std::max<int64_t>(1, b);
    const size_t wgs = std::max<size_t>(1, pictures * p.n_strips);
    return (uint32_t)std::min<size_t>(p.n_mt, std::max<size_t>(1, (512 + wgs - 1) / wgs));
}

// Which items each persistent workgroup of a matrix-pipe launch walks (full-width arithmetic, fl_mfma.hip): `items` come in the classic
// order (pictures x strips [x bands], longest first, strips of a picture on one XCD) and leave in WORKGROUP-MAJOR order; lists[2 b],
// lists[2 b + 1] = first item and item count of workgroup b.
//
// General launch: workgroup b takes items b, b + G, b + 2 G, ... of the classic order.
//
// Uniform launch (every picture the same plan, cut into the same S strips, whole pictures): a workgroup keeps ONE strip, so that its walk
// goes from one picture's rows into the next one's without a new set-up (fl_mfma.hip: a "light" transition -- no operand copy, no tile
// table, no barrier, the last tile converted on the way), and the S strips of a picture run on one XCD at the same time (hardware
// deals workgroup b to XCD (b + const) mod 8; the strips share their halo columns through that XCD's L2): per XCD floor(slots / S)
// triples of neighbouring slots, the slots left over form cross-XCD triples, and G mod S workgroups stay free.  Every triple takes
// R = floor(pictures / triples) pictures.  The pictures left over are cut into row bands of `plan` (an item each: a band computes
// exactly what the full walk computes for its rows) and dealt to the workgroups with the least work -- the free ones first --, so
// that the launch ends within one band of its average instead of one picture.
void assign_items(std::vector<MfmaItem> &mitems, size_t base, uint32_t &nitems, uint32_t G, MfmaPlan *plan, std::vector<uint32_t> &lists)
{
    MfmaItem *items = mitems.data() + base;
    const uint32_t n = nitems;
    std::vector<std::vector<MfmaItem>> per_wg(G);
    // General launch: the classic order has the strips of a picture at positions that agree modulo 8 (the XCD they should run on);
    // item k goes to the least-loaded workgroup of ITS class b = k (mod 8) -- round-robin would do for equal items, but a mixed
    // launch (config 4: 4K items walk twice the K-blocks of 1080p ones) needs the balance a dynamic dispatch used to give
    auto classic = [&]() {
        std::vector<uint64_t> load(G, 0);
        for (uint32_t k = 0; k < n; ++k) {
            uint32_t best = k % 8u < G ? k % 8u : 0u;
            for (uint32_t b = best; b < G; b += 8u) if (load[b] < load[best]) best = b;
            per_wg[best].push_back(items[k]);
            load[best] += items[k].kb1 - items[k].kb0 + 2u;
        }
    };
    // uniform?  (the classic order of a uniform launch: 8 pictures interleaved strip by strip; recover pictures x strips from the jobs)
    bool uniform = plan && n >= 2 && G >= 16;
    std::vector<uint32_t> strips, jobs_in_order;
    if (uniform) {
        std::map<uint32_t, std::vector<MfmaItem>> by_job;
        for (uint32_t k = 0; k < n; ++k) {
            if (!by_job.count(items[k].job)) jobs_in_order.push_back(items[k].job);
            by_job[items[k].job].push_back(items[k]);
        }
        for (const MfmaItem &a : by_job[jobs_in_order[0]]) strips.push_back(a.strip_off);
        const uint32_t S = (uint32_t)strips.size();
        uniform = S >= 2 && S <= 8 && G >= 8u * S && n == S * jobs_in_order.size();
        for (uint32_t j : jobs_in_order) {
            if (!uniform) break;
            const auto &v = by_job[j];
            if (v.size() != S) { uniform = false; break; }
            for (uint32_t s2 = 0; s2 < S; ++s2)
                if (v[s2].strip_off != strips[s2] || v[s2].vplan_off != items[0].vplan_off || v[s2].kb0 != items[0].kb0 || v[s2].kb1 != items[0].kb1 ||
                    v[s2].tile0 != items[0].tile0 || v[s2].tile1 != items[0].tile1 || v[s2].tile0 != 0u || v[s2].tile1 != (uint32_t)plan->tiles.size())
                    uniform = false;
        }
        if (uniform) {
            const uint32_t P = (uint32_t)jobs_in_order.size();
            // triples of workgroups: inside an XCD first, then across
            std::vector<std::vector<uint32_t>> triples;
            std::vector<uint32_t> leftover;
            for (uint32_t x = 0; x < 8; ++x) {
                const uint32_t slots = (G - x + 7u) / 8u, t_x = slots / S;
                for (uint32_t j = 0; j < slots; ++j) {
                    const uint32_t b = 8u * j + x;
                    if (j < t_x * S) { if (j % S == 0) triples.emplace_back(); triples.back().push_back(b); }
                    else leftover.push_back(b);
                }
            }
            std::sort(leftover.begin(), leftover.end());
            for (uint32_t q = 0; q + S <= leftover.size(); q += S) triples.emplace_back(leftover.begin() + q, leftover.begin() + q + S);
            const uint32_t T = (uint32_t)triples.size(), R = T ? P / T : 0u;
            if (!R) uniform = false;
            else {
                for (uint32_t r = 0; r < R; ++r)
                    for (uint32_t t = 0; t < T; ++t) {
                        const auto &v = by_job[jobs_in_order[r * T + t]];
                        for (uint32_t s2 = 0; s2 < S; ++s2) per_wg[triples[t][s2]].push_back(v[s2]);
                    }
                // the pictures left over, in bands
                const uint32_t Lp = P - R * T;
                if (Lp) {
                    const uint32_t nt = (uint32_t)plan->tiles.size();
                    const uint32_t nb = std::max(1u, std::min(nt, (2u * G + Lp * S - 1u) / (Lp * S)));
                    const std::vector<MfmaItem> &bands = plan->items_for(nb);
                    std::vector<MfmaItem> units;
                    for (uint32_t p2 = R * T; p2 < P; ++p2)
                        for (MfmaItem u : bands) { u.job = jobs_in_order[p2]; units.push_back(u); }
                    std::stable_sort(units.begin(), units.end(), [](const MfmaItem &x, const MfmaItem &y) { return x.kb1 - x.kb0 > y.kb1 - y.kb0; });
                    // least-loaded workgroup first (load in K-blocks; a transition counts like two)
                    std::vector<uint64_t> load(G, 0);
                    for (uint32_t b = 0; b < G; ++b) for (const MfmaItem &a : per_wg[b]) load[b] += a.kb1 - a.kb0;
                    for (const MfmaItem &u : units) {
                        uint32_t best = 0;
                        for (uint32_t b = 1; b < G; ++b) if (load[b] < load[best]) best = b;
                        per_wg[best].push_back(u);
                        load[best] += u.kb1 - u.kb0 + 2u;
                    }
                }
            }
        }
    }
    if (!uniform) { for (auto &v : per_wg) v.clear(); classic(); }
    // flatten, workgroup-major
    std::vector<MfmaItem> flat;
    lists.assign(2u * G, 0u);
    for (uint32_t b = 0; b < G; ++b) {
        lists[2 * b] = (uint32_t)flat.size();
        lists[2 * b + 1] = (uint32_t)per_wg[b].size();
        flat.insert(flat.end(), per_wg[b].begin(), per_wg[b].end());
    }
    mitems.resize(base);
    mitems.insert(mitems.end(), flat.begin(), flat.end());
    nitems = (uint32_t)flat.size();
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
    if (w.luma_mid) { j.dw = j.cw; j.dh = j.ch; j.ox = 0; j.oy = 0; } // (the picture alone, tightly packed: the blur supplies the frame)
    j.fill = (uint32_t)w.p->fill_r | ((uint32_t)w.p->fill_g << 8) | ((uint32_t)w.p->fill_b << 16) | (255u << 24);
    j.vtab = w.vtab; j.htab = w.htab;
    j.pad1 = w.tile_w;
    j.pad0 = w.tile_vplan;
    if (w.s1 == S1_NEAREST) {
        // sample.rs: ratio = in as f32 / out as f32, carried as bits where the Lanczos3 jobs carry table offsets
        const float ry = (float)w.sh / (float)pl.resized_h, rx = (float)w.sw / (float)pl.resized_w;
        memcpy(&j.vtab, &ry, 4); memcpy(&j.htab, &rx, 4);
    }
}


// Tile width of the tiled two-pass kernel for output columns [cx, cx + cw) of axis h: the largest power of two (256 .. 16) whose
// tiles all have a source column window that fits the kernel's LDS tile (kTileLdsFloats / (8 rows x channels)); 0 = none does
// (ratios beyond ~7 with four channels: those geometries belong to the other kernels anyway), or a band of 8 output rows [cy + 8 k, ...)
// of axis v touches more source rows than the kernel's table of vertical weights holds.
uint32_t tile_width_for(const HostAxis &h, const HostAxis &v, uint32_t cx, uint32_t cw, uint32_t cy, uint32_t ch, uint32_t mc)
{
    for (uint32_t y0 = cy; y0 < cy + ch; y0 += 8u) {
        const uint32_t y1 = std::min(y0 + 8u, cy + ch);
        if (v.left[y1 - 1] + v.count[y1 - 1] - v.left[y0] > kTileVRows) return 0;
    }
    const uint32_t nc_max = kTileLdsFloats / (8u * mc); // (8 = the kernel's rows per tile)
    for (uint32_t tw = 256; tw >= 16; tw >>= 1) {
        bool fits = true;
        for (uint32_t x0 = cx; x0 < cx + cw && fits; x0 += tw) {
            const uint32_t x1 = std::min(x0 + tw, cx + cw);
            if (h.left[x1 - 1] + h.count[x1 - 1] - h.left[x0] > nc_max) fits = false;
            if (h.woff[x1 - 1] + h.count[x1 - 1] - h.woff[x0] > kTileWeightFloats) fits = false;
        }
        if (fits) return tw;
    }
    return 0;
}

// The tiled two-pass kernel's vertical tables for output rows [cy, cy + ch) of axis v: every band of 8 rows gets its source row
// range and a dense [source row][output row] weight block, so that the kernel's band loop starts with one coalesced copy instead
// of three dependent table look-ups per weight.
void build_tile_vplan(const HostAxis &v, uint32_t cy, uint32_t ch, std::vector<uint32_t> &blk)
{
    const uint32_t nb = (ch + 7u) / 8u;
    uint32_t stride = 1;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t y0 = cy + 8u * b, y1 = std::min(y0 + 8u, cy + ch);
        stride = std::max(stride, v.left[y1 - 1] + v.count[y1 - 1] - v.left[y0]);
    }
    TileVPlanHeader hd{nb, stride, (uint32_t)(sizeof(TileVPlanHeader) / 4), (uint32_t)(sizeof(TileVPlanHeader) / 4) + 2u * nb};
    blk.assign((size_t)hd.dense_off + (size_t)nb * stride * 8u, 0u);
    memcpy(blk.data(), &hd, sizeof(hd));
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t y0 = cy + 8u * b, y1 = std::min(y0 + 8u, cy + ch);
        const uint32_t top = v.left[y0], rv = v.left[y1 - 1] + v.count[y1 - 1] - top;
        blk[hd.bands_off + 2u * b] = top;
        blk[hd.bands_off + 2u * b + 1u] = rv;
        float *dense = reinterpret_cast<float *>(blk.data() + hd.dense_off) + (size_t)b * stride * 8u;
        for (uint32_t oy = y0; oy < y1; ++oy)
            for (uint32_t k = 0; k < v.count[oy]; ++k) dense[(size_t)(v.left[oy] + k - top) * 8u + (oy - y0)] = v.weights[v.woff[oy] + k];
    }
}

// XCD-aware numbering: workgroups are dealt to the 8 XCDs round-robin in launch order and each XCD has its own L2.
// The strips of one picture re-read each other's halo columns, so inside every run of equally long workgroups the
// items of 8 pictures are interleaved: picture p's strips get launch indices congruent modulo 8 (same XCD, close
// in time) and the halo is served by that XCD's L2 instead of HBM.  Speed only; any placement is correct.
template <class Item, class Len>
void xcd_interleave(Item *first, uint32_t nitems, Len len_of)
{
#ifdef FL_EXPERIMENT
    static const bool no_xcd_order = [] { const char *e = getenv("FLGPU_NO_XCD_ORDER"); return e && e[0] == '1'; }(); // A/B experiments (experiment builds only)
#else
    constexpr bool no_xcd_order = false;
#endif
    std::vector<Item> tmp;
    for (uint32_t a = 0; a < nitems && !no_xcd_order;) {
        uint32_t b = a;
        const uint32_t len = len_of(first[a]);
        while (b < nitems && len_of(first[b]) == len) ++b;
        // items of one job are consecutive inside the run (stable sort): collect up to 8 jobs at a time
        for (uint32_t g0 = a; g0 < b;) {
            uint32_t g1 = g0, njob = 0, prev = 0xffffffffu, per = 0, cur = 0;
            bool uniform = true;
            while (g1 < b) {
                if (first[g1].job != prev) {
                    if (njob == 8) break;
                    if (njob >= 1) { if (per == 0) per = cur; else if (cur != per) uniform = false; }
                    ++njob; prev = first[g1].job; cur = 0;
                }
                ++cur; ++g1;
            }
            if (per == 0) per = cur; else if (cur != per) uniform = false;
            if (uniform && njob > 1 && per > 1) {
                tmp.assign(first + g0, first + g1);
                for (uint32_t s2 = 0; s2 < per; ++s2)
                    for (uint32_t j2 = 0; j2 < njob; ++j2) first[g0 + s2 * njob + j2] = tmp[j2 * per + s2];
            }
            g0 = g1;
        }
        a = b;
    }
}

} // namespace

namespace fl {

// What a JPEG file's own header may make this library allocate is decided here, before anything is allocated: the header
// must describe the picture the caller announced, the decoded picture must respect the limit the reference's decoder runs
// under (image::Limits::default(): max_alloc 512 MiB -- ImageReader rejects larger pictures, handler.rs:205-220), and the file
// must be long enough to hold that many blocks at all (a block costs at least a DC and an end-of-block code, two bits), so a
// few hundred hostile bytes cannot reserve gigabytes of pinned or host memory.
int jpeg_source_precheck(flgpu_ctx *c, const flgpu_image *src, const JpegInfo &info)
{
    if (info.width != src->width || info.height != src->height) {
        c->set_error("FLGPU_IMG_JPEG_SOURCE: width / height do not match the file (see flgpu_jpeg_info_of)");
        return FLGPU_ERR_INVALID_ARG;
    }
    const uint64_t decoded = (uint64_t)info.width * info.height * std::max<uint32_t>(info.components, 1u);
    if (decoded > (512ull << 20)) {
        c->set_error("JPEG source: the decoded picture exceeds the 512 MiB the reference's decoder allows (image::Limits)");
        return FLGPU_ERR_UNSUPPORTED;
    }
    const uint64_t blocks = ((uint64_t)(info.width + 7u) / 8u) * ((info.height + 7u) / 8u) * std::max<uint32_t>(info.components, 1u);
    // How short can a file be for that many blocks?  Sequential coding spends at least a DC and an end-of-block code on a block
    // (two bits); a progressive file needs only its first DC scan (one bit per block), every later scan may end 32767 blocks
    // with one end-of-band run.  Below that the header lies about the picture.  (Chroma sub-sampling only lowers the count:
    // the test is repeated on the smallest count the frame could mean.)
    const uint64_t bits_per_block = info.progressive ? 1u : 2u;
    const uint64_t fewest = ((uint64_t)(info.width + 15u) / 16u) * ((info.height + 15u) / 16u) * 3u;
    if (std::min(blocks, std::max<uint64_t>(fewest, 1u)) * bits_per_block > 8ull * src->capacity + 512ull) {
        c->set_error("malformed JPEG stream: shorter than its header's picture needs");
        return FLGPU_ERR_INVALID_ARG;
    }
    // multi-scan files are assembled in a full coefficient array first (128 bytes per block, fl_jpeghuff.cpp): it counts
    // against the same 512 MiB
    if (info.progressive && blocks * 128ull > (512ull << 20)) {
        c->set_error("JPEG source: the coefficient array of this progressive picture exceeds 512 MiB");
        return FLGPU_ERR_UNSUPPORTED;
    }
    return FLGPU_OK;
}

thread_local bool tl_force_host_huffman = false;

// Whether JPEG sources are entropy-decoded on the device where the file allows it.
// 0 = this file is Huffman-decoded on the host, 1 = on the device if the caller has no idle CPU for it, 2 = on the device in any case.
// Switches (flgpu_debug_set; tests and A/B runs): host_huffman = never on the device, device_huffman_always = always,
// device_huffman_min_bytes = files below it are decoded faster by the thread that holds them than by six kernel launches.
int device_huffman_policy(const flgpu_ctx *c, uint64_t file_bytes)
{
    const DebugSwitches &dbg = *c->dbg;
    if (tl_force_host_huffman || dbg.on(DBG_HOST_HUFFMAN)) return 0;
    if (file_bytes < (uint64_t)std::max<int64_t>(0, dbg.get(DBG_DEVICE_HUFFMAN_MIN_BYTES))) return 0;
    return dbg.on(DBG_DEVICE_HUFFMAN_ALWAYS) ? 2 : 1;
}

int jpeg_source_to_blob(flgpu_ctx *c, const flgpu_image *src, uint8_t *blob, size_t cap, JpegBlobHeader *hdr, size_t *used, bool host_huffman)
{
    int rc = -2;
    // files the device entropy decoder takes (sequential, one interleaved scan) are only STAGED here: header,
    // code tables, the segment without its stuffing -- tens of microseconds instead of ~2 ms of Huffman decoding on this thread
    if (!host_huffman && device_huffman_policy(c, src->capacity) != 0) rc = jpeg_entropy_stage(src->data, (size_t)src->capacity, blob, cap, used);
    if (rc == -2) rc = jpeg_entropy_decode(src->data, (size_t)src->capacity, blob, cap, used);
    if (rc == -2) { c->set_error("JPEG stream not covered by the device decoder (arithmetic coding, 12-bit samples, lossless or hierarchical processes)"); return FLGPU_ERR_UNSUPPORTED; }
    if (rc) { c->set_error("malformed JPEG stream"); return FLGPU_ERR_INVALID_ARG; }
    memcpy(hdr, blob, sizeof(*hdr));
    if (hdr->width != src->width || hdr->height != src->height || (hdr->nc == 4 ? 3u : hdr->nc) != src->channels) {
        c->set_error("FLGPU_IMG_JPEG_SOURCE: width / height / channels do not match the file (see flgpu_jpeg_info_of)");
        return FLGPU_ERR_INVALID_ARG;
    }
    return FLGPU_OK;
}

int decode_jpeg_sources(flgpu_ctx *c, size_t n, flgpu_image *dsrc, const JpegSrc *srcs, hipStream_t st)
{
    size_t nj = 0, scratch = 0;
    for (size_t i = 0; i < n; ++i) {
        const JpegBlobHeader *H = srcs[i].hdr;
        if (!H) continue;
        ++nj;
        const size_t px = (size_t)H->width * H->height;
        scratch += align_up(H->plane_bytes, 256) + align_up(px * H->nc + 64, 256);
        if (H->nc == 4) scratch += align_up((px + 3) / 4 * 12, 256); // the Rgb8 picture after the CMYK table
    }
    c->last_jh_slot.assign(n, -1);
    c->last_jh_n = 0;
    if (!nj) return FLGPU_OK;
    { const int brc = clut_batch_begin(c); if (brc) return brc; } // tables selected below stay resident until the batch's kernels are launched
    // ---- staged sources (kJhMagic): the entropy-coded segment is decoded on the device first, into a blob of its own ----
    {
        size_t njh = 0, bytes = 0, nitems = 0;
        for (size_t i = 0; i < n; ++i) {
            const JpegBlobHeader *H = srcs[i].hdr;
            if (!H || H->magic != kJhMagic) continue;
            ++njh;
        }
        if (njh) {
            // per picture: the blob, states (nsub + 1 x 8), counts and prefix (nsub x 16 each)
            std::vector<JhJob> jobs;
            std::vector<JhItem> items;
            std::vector<size_t> blob_off, scratch_off;
            uint32_t max_blocks = 0;
            for (size_t i = 0; i < n; ++i) {
                const JpegBlobHeader *H = srcs[i].hdr;
                if (!H || H->magic != kJhMagic) continue;
                const JpegHuffStage &S = srcs[i].stage;
                const uint32_t nsub = jh_subsequences(S);
                blob_off.push_back(bytes); bytes += align_up(jh_blob_bytes(*H), 256);
                scratch_off.push_back(bytes); bytes += align_up((size_t)(nsub + 2) * 8 + (size_t)nsub * 8 + (size_t)nsub * 32 + 32, 256);
                JhJob j{};
                j.stage = static_cast<const uint8_t *>(dsrc[i].data);
                j.nsub = nsub;
                for (uint32_t f = 0; f < nsub; f += kJhSubsPerItem) items.push_back({(uint32_t)jobs.size(), f});
                c->last_jh_slot[i] = (int32_t)jobs.size();
                jobs.push_back(j);
                max_blocks = std::max(max_blocks, H->nblocks);
            }
            nitems = items.size();
            FL_HIP(c, c->d_jh.reserve(bytes), "device entropy decode scratch");
            FL_HIP(c, c->d_jherr.reserve(njh * 4), "device entropy decode error words");
            FL_HIP(c, c->h_jherr.reserve(njh * 4), "device entropy decode error words");
            const size_t jobs_b = align_up(njh * sizeof(JhJob), 256);
            FL_HIP(c, c->h_jhjobs.reserve(jobs_b + nitems * sizeof(JhItem)), "device entropy decode descriptors");
            FL_HIP(c, c->d_jhjobs.reserve(jobs_b + nitems * sizeof(JhItem)), "device entropy decode descriptors");
            size_t k = 0;
            for (size_t i = 0; i < n; ++i) {
                if (c->last_jh_slot[i] < 0) continue;
                JhJob &j = jobs[k];
                uint8_t *base = static_cast<uint8_t *>(c->d_jh.p);
                j.blob = base + blob_off[k];
                j.states = reinterpret_cast<uint64_t *>(base + scratch_off[k]);
                j.used = j.states + (j.nsub + 2u);
                j.counts = reinterpret_cast<int32_t *>(base + scratch_off[k] + align_up((size_t)(2u * j.nsub + 2u) * 8, 16));
                j.prefix = j.counts + (size_t)j.nsub * 4;
                j.err = static_cast<uint32_t *>(c->d_jherr.p) + k;
                dsrc[i].data = j.blob; // what the IDCT kernel reads from here on
                c->stats.jpeg_device_huffman++;
                ++k;
            }
            memcpy(c->h_jhjobs.p, jobs.data(), njh * sizeof(JhJob));
            memcpy(static_cast<char *>(c->h_jhjobs.p) + jobs_b, items.data(), nitems * sizeof(JhItem));
            FL_HIP(c, hipMemcpyAsync(c->d_jhjobs.p, c->h_jhjobs.p, jobs_b + nitems * sizeof(JhItem), hipMemcpyHostToDevice, st), "device entropy decode descriptors");
            FL_HIP(c, launch_jpeg_huff(static_cast<const JhJob *>(c->d_jhjobs.p), jobs.data(), (uint32_t)njh,
                                       reinterpret_cast<const JhItem *>(static_cast<const char *>(c->d_jhjobs.p) + jobs_b), (uint32_t)nitems, max_blocks, st),
                   "device entropy decode kernels");
            c->last_jh_n = (uint32_t)njh;
        }
    }
    FL_HIP(c, c->d_dec.reserve(scratch), "JPEG decode scratch");
    FL_HIP(c, c->h_decjobs.reserve(nj * sizeof(JpegDecJob)), "JPEG decode descriptors");
    FL_HIP(c, c->d_decjobs.reserve(nj * sizeof(JpegDecJob)), "JPEG decode descriptors");
    JpegDecJob *jobs = static_cast<JpegDecJob *>(c->h_decjobs.p);
    size_t off = 0, k = 0;
    uint32_t max_blocks = 0, max_w = 0, max_h = 0;
    struct Cmyk { const void *raw; void *rgb; const void *clut; uint64_t px; bool ycck; };
    std::vector<Cmyk> cmyk;
    for (size_t i = 0; i < n; ++i) {
        if (!srcs[i].hdr) continue;
        const JpegBlobHeader &H = *srcs[i].hdr;
        const size_t px = (size_t)H.width * H.height;
        JpegDecJob &j = jobs[k++];
        j.blob = dsrc[i].data;
        j.planes = static_cast<uint8_t *>(c->d_dec.p) + off; off += align_up(H.plane_bytes, 256);
        j.dst = static_cast<uint8_t *>(c->d_dec.p) + off; off += align_up(px * H.nc + 64, 256);
        jpeg_color_job(H, j);
        max_blocks = std::max(max_blocks, H.nblocks); max_w = std::max(max_w, H.width); max_h = std::max(max_h, H.height);
        c->stats.jpeg_sources++;
        c->stats.jpeg_upload_bytes += H.total_bytes;
        uint8_t *pixels = j.dst;
        uint32_t channels = H.nc;
        if (H.nc == 4) {
            // convert_jpeg_color_if_needed (handler.rs:398-466): raw CMYK / YCCK samples -> (YCCK loop) -> the profile's table
            const void *clut = nullptr;
            const int rc = select_clut(c, srcs[i].icc, srcs[i].icc_len, &clut);
            if (rc) return rc;
            uint8_t *rgb = static_cast<uint8_t *>(c->d_dec.p) + off; off += align_up((px + 3) / 4 * 12, 256);
            cmyk.push_back({j.dst, rgb, clut, px, H.adobe_transform == 3u});
            pixels = rgb;
            channels = 3;
            c->stats.cmyk_pixels += px;
        }
        dsrc[i].data = pixels;
        dsrc[i].channels = channels;
        dsrc[i].capacity = (uint64_t)px * channels;
        dsrc[i].flags &= ~FLGPU_IMG_JPEG_SOURCE;
    }
    FL_HIP(c, hipMemcpyAsync(c->d_decjobs.p, jobs, nj * sizeof(JpegDecJob), hipMemcpyHostToDevice, st), "JPEG decode descriptors");
    for (size_t base = 0; base < nj; base += 32768) { // grid.y / .z limits
        const uint32_t cnt = (uint32_t)std::min<size_t>(32768, nj - base);
        FL_HIP(c, launch_jpeg_decode(static_cast<const JpegDecJob *>(c->d_decjobs.p) + base, cnt, max_blocks, max_w, max_h, st), "JPEG decode kernels");
    }
    for (const Cmyk &m : cmyk) FL_HIP(c, launch_cmyk_clut(m.raw, m.rgb, m.clut, kCmykGrid, m.px, m.ycck, st), "CMYK kernel");
    return FLGPU_OK;
}

// Enqueues the copy of the device entropy decoder's error words (final once its kernels have run): a caller that waits for the stream anyway
// asks for them in front of that wait and passes fetched = true below.
int entropy_failures_fetch(flgpu_ctx *c, size_t n, hipStream_t st)
{
    if (!c->last_jh_n || c->last_jh_slot.size() != n) return FLGPU_OK;
    FL_HIP(c, hipMemcpyAsync(c->h_jherr.p, c->d_jherr.p, (size_t)c->last_jh_n * 4, hipMemcpyDeviceToHost, st), "device entropy decode: error words D2H");
    return FLGPU_OK;
}

int entropy_failures(flgpu_ctx *c, size_t n, std::vector<uint8_t> &bad, hipStream_t st, bool fetched)
{
    bad.assign(n, 0);
    if (!c->last_jh_n || c->last_jh_slot.size() != n) return 0;
    if (!fetched && (hipMemcpyAsync(c->h_jherr.p, c->d_jherr.p, (size_t)c->last_jh_n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
        c->set_error("device entropy decode: error words D2H");
        return -FLGPU_ERR_DEVICE;
    }
    int nbad = 0;
    const uint32_t *e = static_cast<const uint32_t *>(c->h_jherr.p);
    if (c->dbg->on(DBG_DEBUG_JH)) for (uint32_t k = 0; k < c->last_jh_n; ++k) fprintf(stderr, "device entropy decode: picture %u error word %u\n", k, e[k]);
    for (size_t i = 0; i < n; ++i)
        if (c->last_jh_slot[i] >= 0 && e[c->last_jh_slot[i]]) { bad[i] = 1; ++nbad; }
    c->stats.jpeg_device_huffman_retries += (uint64_t)nbad;
    return nbad;
}

uint64_t staged_out_bytes(const flgpu_params &p, const flgpu_plan &plan, uint64_t)
{
    return p.front_end == FLGPU_FE_JPEG ? plan.max_out_bytes : plan.out_bytes;
}

int run_batch_device(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, bool same_params,
                     flgpu_image *dsts, hipStream_t st)
{
    if (n == 0) return FLGPU_OK;
    if (!srcs || !ps || !dsts) return FLGPU_ERR_INVALID_ARG;
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    if (!st) st = c->stream;
    if (c->last_stream && c->last_stream != st && c->last_done) FL_HIP(c, hipStreamWaitEvent(st, c->last_done, 0), "stream handoff");

    RoctxRange range_batch("flgpu batch");
    // ---- plan every image ------------------------------------------------
    std::unique_ptr<RoctxRange> range_plan(new RoctxRange("flgpu plan + tables"));
    std::vector<Work> work(n);
    size_t tmp_a_bytes = 0, tmp_b_bytes = 0, tmp_o_bytes = 0, tmp_al_bytes = 0, jpeg_coef_bytes = 0, jpeg_off_bytes = 0, jpeg_raw_bytes = 0;
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
        // Which resample kernel serves a request is a function of the REQUEST (geometry, channels, pre-op), never of where the
        // caller's buffer happens to start: the matrix-pipe kernel moves 16-byte pieces of a row, so a source whose rows qualify
        // but whose base is not 16-byte aligned (only possible through the device-batch entry point; staged and decoded sources
        // are 256-byte aligned) is copied to aligned scratch first.  The two kernels may differ by 1 LSB; an HTTP cache in front of
        // the service must not see that difference come and go with an address.
        // (only where that kernel can be the one: its own gate below -- no pre-op, rows of at least 64 bytes, not switched off -- and a
        // ratio above the window-tile kernel's range, which takes any alignment; everything else is served by kernels that do not care)
        // (from ratio 2 up whether or not the window-tile kernel is on: that kernel takes any alignment, but where ITS planner refuses
        // a geometry below ratio 2.5 the request falls through to the matrix-pipe branch -- which must not then depend on the address)
        const bool mfma_candidate = !c->dbg->on(DBG_NO_MFMA) && !c->dbg->on(DBG_FORCE_GENERIC) && (size_t)w.sw * w.cs >= 64u &&
                                    (uint64_t)w.sh >= 2u * (uint64_t)pl.resized_h;
        if (pl.resampled && w.s1 == S1_GENERIC && !pre_changes && !w.orient && mfma_candidate && ((size_t)w.sw * w.cs) % 16u == 0 && (uintptr_t)s.data % 16u != 0) {
            w.align_off = tmp_al_bytes; w.align_copy = true;
            tmp_al_bytes += align_up((size_t)s.width * s.height * s.channels, 256);
        }
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
    FL_HIP(c, c->d_tmp_al.reserve(tmp_al_bytes), "alignment scratch");
    for (auto &w : work)
        if (w.align_copy) {
            uint8_t *al = static_cast<uint8_t *>(c->d_tmp_al.p) + w.align_off;
            FL_HIP(c, hipMemcpyAsync(al, w.src, (size_t)w.sw * w.sh * w.cs, hipMemcpyDeviceToDevice, st), "alignment copy");
            w.src = al;
        }
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
    // the context's switches (flgpu_debug_set; tests and A/B runs), read once per batch
    const DebugSwitches &dbg = *c->dbg;
    const bool force_generic = dbg.on(DBG_FORCE_GENERIC);
    const uint32_t forced_bands = (uint32_t)std::max<int64_t>(0, dbg.get(DBG_FORCE_BANDS));
    const bool no_mfma = dbg.on(DBG_NO_MFMA); // keep the streaming kernel
    // which arithmetic the matrix-pipe kernel computes in (fl_mfma.h): the full-width one unless mfma_arith = 1 asks for
    // rounds 2-3's (A/B runs and the tests that keep the packed kernel's bars)
    const MfmaArith mfma_arith = dbg.on(DBG_MFMA_ARITH) ? MFMA_ARITH_PACKED : MFMA_ARITH_FULL;
    const bool no_tile = dbg.on(DBG_NO_TILE);
    // the window-tile matrix-pipe kernel (fl_wtile.h) for mild ratios, up-scales and blurs: full-width arithmetic only; no_wtile
    // keeps the f32 vector kernels
    const bool use_wtile = !dbg.on(DBG_NO_WTILE) && !no_mfma && !force_generic && mfma_arith == MFMA_ARITH_FULL;
    const bool wt_first = dbg.on(DBG_WTILE_FIRST); // experiments: the window-tile kernel before the streaming matrix-pipe kernel
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
            // The matrix-pipe kernel takes down-scales (any channel count, no pre-op) whose rows are 16-byte aligned (it moves 16-byte pieces of a row
            // straight into LDS).  The choice depends on the request's geometry only, never on the batch around it.
            // Ratios below 2.5 (up-scales included) go to the window-tile kernel BEFORE the fused ones: measured 1.00 vs 1.03 ms per 256 at ratio 2.4 and -- against the
            // streaming f32 kernel, which serves what the matrix-pipe planner refuses down there -- 1.16 vs 2.10 at 2.13.  From 2.67 up the
            // streaming matrix-pipe kernel wins since its wide layout keeps operands in LDS (0.78 vs 0.92 at 2.67, 0.72 vs 0.80 at 3;
            // profiles/r04_wtile_experiments.txt); where ITS planner refuses a geometry below ratio 3.4, the window-tile kernel is asked again.
            const bool wt_range = 2u * w.sh < 5u * w.plan.resized_h;
            if ((wt_first || wt_range) && use_wtile && (w.pre == PRE_NONE || w.pre == PRE_INVERT) && (!w.plan.letterboxed || (uintptr_t)w.s1_dst % 4u == 0)) {
                Job jtmp; fill_job(w, jtmp);
                WtPlan *wp = get_wtile_plan(c, w.vk, *w.va, w.hk, *w.ha, jtmp.cx, jtmp.cy, jtmp.cw, jtmp.ch, w.cs);
                if (wp->arena_full || c->h_arena.size() >= c->arena_cap_words - 1024) { full = true; break; }
                if (wp->ok) { w.s1 = S1_WTILE; w.wplan = wp; continue; }
            }
            if (w.pre == PRE_NONE && !force_generic && !no_mfma && (w.sw * w.cs) % 16u == 0 && (uintptr_t)w.src % 16u == 0 &&
                (!w.plan.letterboxed || (uintptr_t)w.s1_dst % 4u == 0) && w.sw * w.cs >= 64u) {
                Job jtmp; fill_job(w, jtmp);
                MfmaPlan *mp = get_mfma_plan(c, w.vk, *w.va, w.hk, *w.ha, jtmp.cx, jtmp.cy, jtmp.cw, jtmp.ch, w.cs, mfma_arith);
                if (mp->arena_full || c->h_arena.size() >= c->arena_cap_words - 1024) { full = true; break; }
                if (mp->ok) {
                    uint32_t nbands = 1;
                    if (forced_bands) nbands = forced_bands;
                    else if (n_resample < 128) {
                        // A small launch: bands of rows so that every CU has an item -- and so that the items come out in whole rounds of the
                        // workgroups.  (Until round 5: ceil(256 / (3 n)) bands; 13 files x 3 strips x 7 bands = 273 items on 256 workgroups,
                        // i.e. two rounds of items a seventh of a strip long where one round of sixths does: 84 us per batch of file requests.)
                        // The cost of a band count: rounds x (the longest item's K-blocks -- the bands' halos are in there -- + a transition's two).
                        const uint64_t G = std::max(1u, c->cu_count);
                        uint64_t best = ~0ull;
                        for (uint32_t b = 1; b <= std::min<uint32_t>(16u, (uint32_t)mp->tiles.size()); ++b) {
                            const std::vector<MfmaItem> &cand = mp->items_for(b);
                            uint32_t longest = 0;
                            for (const MfmaItem &mi : cand) longest = std::max(longest, mi.kb1 - mi.kb0);
                            const uint64_t cost = (((uint64_t)n_resample * cand.size() + G - 1) / G) * (longest + 2u);
                            if (cost < best) { best = cost; nbands = b; }
                        }
                    }
                    w.s1 = S1_MFMA; w.mplan = mp; w.mitems = &mp->items_for(nbands);
                    continue;
                }
            }
            if (use_wtile && (w.pre == PRE_NONE || w.pre == PRE_INVERT) && 2u * w.sh >= 5u * w.plan.resized_h && 10u * w.sh < 34u * w.plan.resized_h &&
                (!w.plan.letterboxed || (uintptr_t)w.s1_dst % 4u == 0)) { // (ratio 2.5 .. 3.4 and no streaming matrix-pipe plan: unaligned rows, a pre-op, a refused geometry)
                Job jtmp; fill_job(w, jtmp);
                WtPlan *wp = get_wtile_plan(c, w.vk, *w.va, w.hk, *w.ha, jtmp.cx, jtmp.cy, jtmp.cw, jtmp.ch, w.cs);
                if (wp->arena_full || c->h_arena.size() >= c->arena_cap_words - 1024) { full = true; break; }
                if (wp->ok) { w.s1 = S1_WTILE; w.wplan = wp; continue; }
            }
            if (stream_supported(w.cs, w.pre) && aligned && !force_generic) {
                Job jtmp; fill_job(w, jtmp);
                uint32_t nbands = 1;
                if (forced_bands) nbands = forced_bands;
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
            // what neither fused kernel takes and no pre-op precedes: the window-tile matrix-pipe kernel (any pitch and alignment)
            if (w.s1 == S1_GENERIC && use_wtile && (w.pre == PRE_NONE || w.pre == PRE_INVERT) && wtile_resample_wanted(w) && (!w.plan.letterboxed || (uintptr_t)w.s1_dst % 4u == 0)) {
                Job jtmp; fill_job(w, jtmp);
                WtPlan *wp = get_wtile_plan(c, w.vk, *w.va, w.hk, *w.ha, jtmp.cx, jtmp.cy, jtmp.cw, jtmp.ch, w.cs);
                if (wp->arena_full || c->h_arena.size() >= c->arena_cap_words - 1024) { full = true; break; }
                if (wp->ok) { w.s1 = S1_WTILE; w.wplan = wp; }
            }
            // what neither fused kernel takes (up-scales, mild down-scales, odd pitches, forced generic): the two passes through an
            // LDS tile instead of an f32 intermediate in HBM, if a tile width fits (FLGPU_NO_TILE=1 keeps the HBM form: tests, A/B)
            if (w.s1 == S1_GENERIC && !no_tile) {
                Job jtmp; fill_job(w, jtmp);
                const uint32_t tw = tile_width_for(*w.ha, *w.va, jtmp.cx, jtmp.cw, jtmp.cy, jtmp.ch, mid_channels(w.cs, w.pre));
                if (tw) {
                    const auto key = std::make_tuple(w.vk, jtmp.cy, jtmp.ch);
                    auto it = c->tile_vplans.find(key);
                    if (it == c->tile_vplans.end()) {
                        std::vector<uint32_t> blk;
                        build_tile_vplan(*w.va, jtmp.cy, jtmp.ch, blk);
                        const uint32_t off = arena_append(c, blk.data(), blk.size());
                        if (!off) { full = true; break; }
                        it = c->tile_vplans.emplace(key, off).first;
                    }
                    w.s1 = S1_TILE; w.tile_w = tw; w.tile_vplan = it->second;
                }
            }
        }
        for (auto &w : work) {
            if (full) break;
            if (w.p->blur_sigma > 0.0f) {
                AxisKey k; const HostAxis *h;
                AxisKey kv; const HostAxis *hv;
                w.bwplan = nullptr; w.luma_mid = false; // (a second attempt after an arena reset plans again)
                if (!get_axis(c, w.plan.out_h, w.plan.out_h, FILTER_GAUSSIAN, w.p->blur_sigma, &kv, &hv) ||
                    !get_axis(c, w.plan.out_w, w.plan.out_w, FILTER_GAUSSIAN, w.p->blur_sigma, &k, &h)) full = true;
                else if (use_wtile && wtile_blur_wanted(dbg, w)) {
                    // a grey picture on a grey frame (R == G == B everywhere, alpha 255): one channel is filtered, if that plan is one of
                    // the single-register-set kind (the kernel's framed source exists in that instantiation only)
                    const bool one = w.plan.letterboxed && w.plan.out_c == 4u && blur_channels(w) == 1u && w.s1 != S1_NONE && w.s1 != S1_NEAREST;
                    WtPlan *wp = one ? get_wtile_plan(c, kv, *hv, k, *h, 0, 0, w.plan.out_w, w.plan.out_h, 1u) : nullptr;
                    if (wp && !wp->arena_full && wp->ok && wp->nslot == 1u) { w.bwplan = wp; w.luma_mid = true; }
                    else {
                        if (wp && wp->arena_full) full = true;
                        wp = full ? nullptr : get_wtile_plan(c, kv, *hv, k, *h, 0, 0, w.plan.out_w, w.plan.out_h, w.plan.out_c);
                        if (wp && (wp->arena_full || c->h_arena.size() >= c->arena_cap_words - 1024)) full = true;
                        else if (wp && wp->ok) w.bwplan = wp;
                    }
                    if (c->h_arena.size() >= c->arena_cap_words - 1024) full = true;
                }
                if (!full && !w.bwplan) {
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
        if (w.s1 == S1_WTILE) s1_groups[{(uint32_t)S1_WTILE | (w.wplan->nslot << 8), w.cs, w.pre, w.plan.letterboxed && !w.luma_mid}].push_back(i);
        else if (w.s1 == S1_MFMA) s1_groups[{(uint32_t)S1_MFMA | (w.mplan->ops_in_lds ? 1u << 8 : 0u) | (w.mplan->wide ? 1u << 9 : 0u) | (w.mplan->full ? 1u << 10 : 0u) | (w.mplan->compact ? 1u << 11 : 0u), w.cs, w.pre, w.plan.letterboxed && !w.luma_mid}].push_back(i);
        else if (w.s1 != S1_NONE) s1_groups[{(uint32_t)w.s1 | (w.splan ? w.splan->nacc << 8 : 0u) | (w.s1 == S1_STREAM && w.unaligned ? 1u << 16 : 0u), w.cs, w.pre, w.plan.letterboxed && !w.luma_mid}].push_back(i);
        if (w.p->blur_sigma > 0.0f && w.bwplan) blur_groups[{kBlurWtileKind | w.bwplan->nslot, w.luma_mid ? 1u : w.plan.out_c, 0, w.luma_mid ? 1u : 0u}].push_back(i);
        else if (w.p->blur_sigma > 0.0f) {
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
    std::vector<MfmaItem> mitems;
    std::vector<uint32_t> mwg; // matrix-pipe launches with persistent workgroups: {first item, items} of every workgroup
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
    struct S1Launch { GroupKey k; uint32_t job_base, njobs, item_base, nitems, nacc, max_nout, grid = 0, wg_base = 0; LaunchGeneric g; size_t lds; size_t mid_floats; uint32_t blur_grid_x; bool blur_tiled; };
    std::vector<S1Launch> s1_launches, blur_launches;
    struct FeLaunch { uint32_t kind, base, n, mw, mh; bool rgba; };
    std::vector<FeLaunch> fe_launches;
    size_t mid_floats_max = 0;
    const size_t kMidCapFloats = (size_t)256 << 20; // 1 GiB of f32 intermediate per launch group

    auto new_launch = [&](const GroupKey &k) {
        S1Launch L{};
        L.k = k; L.job_base = (uint32_t)jobs.size(); L.item_base = (uint32_t)(((k.kind & 255u) == S1_MFMA || (k.kind & 255u) == S1_WTILE || (k.kind & kBlurWtileKind)) ? mitems.size() : items.size());
        L.g.cs = k.cs; L.g.pre = k.pre; L.g.letterbox = k.lb; L.g.grouped = 1;
        return L;
    };
    for (auto &kv : s1_groups) {
        const GroupKey &k = kv.first;
        S1Launch L = new_launch(k);
        MfmaPlan *launch_plan = nullptr; // the matrix-pipe plan every picture of the launch shares, if they all do
        bool launch_plan_set = false;
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
            if ((k.kind & 255u) == S1_TILE) {
                L.g.tile_w_min = L.g.tile_w_min ? std::min(L.g.tile_w_min, w.tile_w) : w.tile_w;
                c->stats.resample_src_bytes += (uint64_t)j.src_bytes;
                c->stats.resample_dst_bytes += w.plan.pixel_bytes;
            }
            if ((k.kind & 255u) == S1_WTILE) {
                const size_t before = mitems.size();
                w.wplan->items_for(wtile_bands(dbg, *w.wplan, kv.second.size()), (uint32_t)jobs.size(), mitems);
                L.nitems += (uint32_t)(mitems.size() - before);
                L.lds = std::max(L.lds, (size_t)w.wplan->lds_bytes);
                c->stats.resample_src_bytes += (uint64_t)j.src_bytes;
                c->stats.resample_dst_bytes += w.plan.pixel_bytes;
            }
            if ((k.kind & 255u) == S1_MFMA) {
                launch_plan = (!launch_plan_set || launch_plan == w.mplan) ? const_cast<MfmaPlan *>(w.mplan) : nullptr; launch_plan_set = true; // (one plan for the whole launch, or none)
                for (MfmaItem it2 : *w.mitems) { it2.job = (uint32_t)jobs.size(); mitems.push_back(it2); }
                L.nitems += (uint32_t)w.mitems->size();
                L.max_nout = std::max(L.max_nout, w.mplan->max_nout);
                c->stats.resample_src_bytes += (uint64_t)j.src_bytes;
                c->stats.resample_dst_bytes += w.plan.pixel_bytes;
            }
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
        if ((k.kind & 255u) == S1_MFMA) {
            if (L.nitems > 1) {
                // longest workgroups first and strips of a picture on one XCD, as for the streaming kernel below
                auto first = mitems.begin() + L.item_base;
                std::stable_sort(first, first + L.nitems, [](const MfmaItem &a, const MfmaItem &b) { return a.kb1 - a.kb0 > b.kb1 - b.kb0; });
                xcd_interleave(&*first, L.nitems, [](const MfmaItem &x) { return x.kb1 - x.kb0; });
            }
            if ((k.kind >> 10) & 1u) {
                // full-width arithmetic: persistent workgroups, each with its own list of items
                L.grid = std::max(1u, std::min(L.nitems, c->cu_count));
                L.wg_base = (uint32_t)mwg.size();
                std::vector<uint32_t> lists;
                assign_items(mitems, L.item_base, L.nitems, L.grid, launch_plan, lists);
                mwg.insert(mwg.end(), lists.begin(), lists.end());
            }
        }
        if ((k.kind & 255u) == S1_WTILE && L.nitems > 1) {
            // strips and bands of a picture on one XCD (their source windows overlap: the halo then comes from that XCD's L2)
            auto first = mitems.begin() + L.item_base;
            xcd_interleave(&*first, L.nitems, [](const MfmaItem &) { return 1u; });
        }
        if ((k.kind & 255u) == S1_STREAM && L.nitems > 1) {
            // longest workgroups first: in a mixed batch a 4K band walks four times the rows of a 1080p one, and the
            // hardware hands out workgroups in index order -- started last, the long ones would be the launch's tail
            auto first = items.begin() + L.item_base;
            std::stable_sort(first, first + L.nitems, [](const StreamItem &a, const StreamItem &b) { return a.r1 - a.r0 > b.r1 - b.r0; });
            xcd_interleave(&*first, L.nitems, [](const StreamItem &x) { return x.r1 - x.r0; });
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
            if (k.kind & kBlurWtileKind) {
                if (w.luma_mid) {
                    // the source is stage 1's unframed Luma8 picture (rw x rh) at (cx, cy) of a virtual sw x sh frame of value `fill`
                    j.rw = std::min(pl.resized_w - pl.crop_x, pl.out_w - pl.place_x); j.rh = std::min(pl.resized_h - pl.crop_y, pl.out_h - pl.place_y);
                    j.cx = pl.place_x; j.cy = pl.place_y;
                    j.src_bytes = j.rw * j.rh;
                    j.fill = (uint32_t)w.p->fill_r * 0x01010101u;
                }
                const size_t before = mitems.size();
                w.bwplan->items_for(wtile_bands(dbg, *w.bwplan, kv.second.size()), (uint32_t)jobs.size(), mitems);
                L.nitems += (uint32_t)(mitems.size() - before);
                L.lds = std::max(L.lds, (size_t)w.bwplan->lds_bytes);
                jobs.push_back(j);
                L.njobs++;
                continue;
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
        if ((k.kind & kBlurWtileKind) && L.nitems > 1) {
            auto first = mitems.begin() + L.item_base;
            xcd_interleave(&*first, L.nitems, [](const MfmaItem &) { return 1u; });
        }
        blur_launches.push_back(L);
    }
    // result words: two per image of the batch, see flgpu_ctx::last_fe
    // ... followed by the batch's device error word (fl_mfma.h FLGPU_DEVERR_*): kernels that wait on one another inside a
    // workgroup bound their waits and report an expired one here instead of delivering pixels that were never synchronised
    const bool has_results = !fe_groups.empty();
    bool has_err_word = false;
    for (auto &L : s1_launches) has_err_word |= (L.k.kind & 255u) == S1_MFMA;
    // (they live at the end of the batch's descriptor block and arrive zeroed with it: a clear of their own was two fill kernels
    // and two engine switches between one batch's last kernel and the next one's first)
    std::vector<size_t> jjob_idx, fjob_idx; // image of every encoder / front-end job: its result words are addressed once the block's place is known
    const uint32_t mfma_spin_limit = (uint32_t)std::max<int64_t>(0, dbg.get(DBG_MFMA_SPIN_LIMIT)); // tests: 0 = every bounded wait expires
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
                j.result = nullptr; jjob_idx.push_back(idx);
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
            f.status = nullptr; fjob_idx.push_back(idx);
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

    // The matrix-pipe kernel's persistent workgroups request the first K-block of their NEXT item in the last pass of the current
    // one: what that request needs (source, pitch, last row, the strip's first byte, the first K-block), one record per item in item
    // order, so that it is ONE scalar load at that point and nothing of the next item occupies registers before (fl_mfma.h MfmaReq).
    std::vector<MfmaReq> mreqs;
    for (auto &L : s1_launches) {
        if ((L.k.kind & 255u) != S1_MFMA) continue;
        mreqs.resize(mitems.size());
        for (uint32_t k = L.item_base; k < L.item_base + L.nitems; ++k) {
            const MfmaItem &mi = mitems[k];
            const Job &j = jobs[mi.job];
            MfmaReq &r = mreqs[k];
            r.src = j.src; r.pitch = j.sw * L.k.cs; r.last_row = j.sh - 1u; r.kb0 = mi.kb0; r.kb1 = mi.kb1; r.job = mi.job;
            r.strip_off = mi.strip_off; r.vplan_off = mi.vplan_off; r.pad[0] = r.pad[1] = 0;
            r.byte0 = reinterpret_cast<const MfmaStrip *>(c->h_arena.data() + mi.strip_off)->byte0;
        }
    }
    // one staging slot: [jobs][items][fjobs][jjobs][mitems][mreqs][mwg]
    const size_t jobs_b = align_up(jobs.size() * sizeof(Job), 256), items_b = align_up(items.size() * sizeof(StreamItem), 256),
                 fjobs_b = align_up(fjobs.size() * sizeof(FrontendJob), 256), jjobs_b = align_up(jjobs.size() * sizeof(JpegJob), 256),
                 mitems_b = align_up(mitems.size() * sizeof(MfmaItem), 256), mreqs_b = align_up(mreqs.size() * sizeof(MfmaReq), 256), mwg_b = align_up(mwg.size() * sizeof(uint32_t), 256);
    const size_t stat_off = jobs_b + items_b + fjobs_b + jjobs_b + mitems_b + mreqs_b + mwg_b, stat_b = (has_results || has_err_word) ? align_up(n * 8 + 8, 256) : 0;
    const size_t desc_b = stat_off + stat_b;
    uint32_t *status_dev = nullptr;
    const Job *d_jobs = nullptr; const StreamItem *d_items = nullptr; const FrontendJob *d_fjobs = nullptr; const JpegJob *d_jjobs = nullptr;
    const MfmaItem *d_mitems = nullptr;
    const MfmaReq *d_mreqs = nullptr;
    const uint32_t *d_mwg = nullptr;
    DescSlot *slot = nullptr;
    if (desc_b) {
        slot = &c->slots[c->next_slot];
        c->next_slot = (c->next_slot + 1) % 4;
        if (slot->busy) { FL_HIP(c, hipEventSynchronize(slot->done), "descriptor slot wait"); slot->busy = false; }
        if (!slot->done) FL_HIP(c, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming), "event");
        FL_HIP(c, slot->host.reserve(desc_b), "pinned descriptors");
        FL_HIP(c, slot->dev.reserve(desc_b), "device descriptors");
        char *hp = static_cast<char *>(slot->host.p);
        if (stat_b) {
            status_dev = reinterpret_cast<uint32_t *>(static_cast<char *>(slot->dev.p) + stat_off);
            memset(hp + stat_off, 0, stat_b);
            for (size_t k = 0; k < jjobs.size(); ++k) jjobs[k].result = status_dev + 2 * jjob_idx[k];
            for (size_t k = 0; k < fjobs.size(); ++k) fjobs[k].status = status_dev + 2 * fjob_idx[k];
        }
        if (!jobs.empty()) memcpy(hp, jobs.data(), jobs.size() * sizeof(Job));
        if (!items.empty()) memcpy(hp + jobs_b, items.data(), items.size() * sizeof(StreamItem));
        if (!fjobs.empty()) memcpy(hp + jobs_b + items_b, fjobs.data(), fjobs.size() * sizeof(FrontendJob));
        if (!jjobs.empty()) memcpy(hp + jobs_b + items_b + fjobs_b, jjobs.data(), jjobs.size() * sizeof(JpegJob));
        if (!mitems.empty()) memcpy(hp + jobs_b + items_b + fjobs_b + jjobs_b, mitems.data(), mitems.size() * sizeof(MfmaItem));
        if (!mreqs.empty()) memcpy(hp + jobs_b + items_b + fjobs_b + jjobs_b + mitems_b, mreqs.data(), mreqs.size() * sizeof(MfmaReq));
        if (!mwg.empty()) memcpy(hp + jobs_b + items_b + fjobs_b + jjobs_b + mitems_b + mreqs_b, mwg.data(), mwg.size() * sizeof(uint32_t));
        // While a previous batch is still running, the block goes up on the context's upload stream: the slot is free (its last
        // batch has ended, see above), so the copy runs under that batch's kernels, and this batch's first kernel follows its
        // last one without a copy engine in between.  A lone request on an idle device sends the block down its own stream (no
        // second stream, no wait: the 15 us would be 3 % of its latency).
        if (c->last_done && hipEventQuery(c->last_done) == hipErrorNotReady) {
            if (!c->up_stream) FL_HIP(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking), "upload stream");
            if (!slot->uploaded) FL_HIP(c, hipEventCreateWithFlags(&slot->uploaded, hipEventDisableTiming), "event");
            FL_HIP(c, hipMemcpyAsync(slot->dev.p, hp, desc_b, hipMemcpyHostToDevice, c->up_stream), "descriptor upload");
            FL_HIP(c, hipEventRecord(slot->uploaded, c->up_stream), "event record");
            // (the HOST waits the ~15 us the 130 KB take: a device-side wait on the event is a barrier packet between the previous
            // batch's last kernel and this one's first, 7 us of idle chip per batch; the host has the previous batch's 2 ms to spare)
            FL_HIP(c, hipEventSynchronize(slot->uploaded), "descriptor upload wait");
        } else {
            FL_HIP(c, hipMemcpyAsync(slot->dev.p, hp, desc_b, hipMemcpyHostToDevice, st), "descriptor upload");
        }
        char *dp = static_cast<char *>(slot->dev.p);
        d_jobs = reinterpret_cast<const Job *>(dp);
        d_items = reinterpret_cast<const StreamItem *>(dp + jobs_b);
        d_fjobs = reinterpret_cast<const FrontendJob *>(dp + jobs_b + items_b);
        d_jjobs = reinterpret_cast<const JpegJob *>(dp + jobs_b + items_b + fjobs_b);
        d_mitems = reinterpret_cast<const MfmaItem *>(dp + jobs_b + items_b + fjobs_b + jjobs_b);
        d_mreqs = reinterpret_cast<const MfmaReq *>(dp + jobs_b + items_b + fjobs_b + jjobs_b + mitems_b);
        d_mwg = reinterpret_cast<const uint32_t *>(dp + jobs_b + items_b + fjobs_b + jjobs_b + mitems_b + mreqs_b);
    }

    range_plan.reset();
    // ---- launches --------------------------------------------------------------
    RoctxRange range_launch("flgpu launches");
    for (auto &O : orient_launches) {
        LaunchGeneric g{};
        g.jobs = d_jobs; g.job_base = O.base; g.njobs = O.n; g.cs = O.cs; g.max_dw = O.mw; g.max_dh = O.mh;
        FL_HIP(c, launch_orient(g, st), "orientation kernel");
    }
    for (auto &L : s1_launches) {
        L.g.jobs = d_jobs; L.g.arena = c->d_arena; L.g.mid = static_cast<float *>(c->d_mid.p);
        L.g.job_base = L.job_base; L.g.njobs = L.njobs;
        L.g.no_place4 = dbg.on(DBG_NO_PLACE4) ? 1u : 0u;
        if ((L.k.kind & 255u) == S1_NEAREST) {
            L.g.nearest = 1;
            FL_HIP(c, launch_place(L.g, false, st), "nearest kernel");
        } else if ((L.k.kind & 255u) == S1_PLACE) {
            FL_HIP(c, launch_place(L.g, false, st), "place kernel");
        } else if ((L.k.kind & 255u) == S1_TILE) {
            if (L.k.lb) FL_HIP(c, launch_place(L.g, true, st), "border fill");
            L.g.grouped = 1;
            {
                ProfileScope ps(c, st, 0);
                FL_HIP(c, launch_tile_resample(L.g, st), "tiled two-pass resample kernel");
            }
            c->stats.resample_launches++;
            c->stats.generic_launches++; // (the two-pass generic resample, LDS form)
        } else if ((L.k.kind & 255u) == S1_WTILE) {
            if (L.k.lb) FL_HIP(c, launch_place(L.g, true, st), "border fill");
            LaunchWtile m{};
            m.jobs = d_jobs; m.items = reinterpret_cast<const WtItem *>(d_mitems + L.item_base); m.arena = c->d_arena; m.nitems = L.nitems;
            m.nslot = (L.k.kind >> 8) & 255u; m.nkmax = kWtOperandRegs / m.nslot; m.letterbox = L.k.lb; m.lds_bytes = (uint32_t)L.lds; m.invert = L.k.pre == PRE_INVERT;
            {
                ProfileScope ps(c, st, 0);
                FL_HIP(c, launch_wtile(m, st), "window-tile matrix-pipe kernel");
            }
            c->stats.resample_launches++;
            c->stats.mfma_launches++;
            c->stats.wtile_launches++;
        } else if ((L.k.kind & 255u) == S1_GENERIC) {
            if (L.k.lb) FL_HIP(c, launch_place(L.g, true, st), "border fill");
            FL_HIP(c, launch_vpass_generic(L.g, st), "generic vertical pass");
            FL_HIP(c, launch_hpass_generic(L.g, st), "generic horizontal pass");
            c->stats.generic_launches++;
        } else if ((L.k.kind & 255u) == S1_MFMA) {
            LaunchMfma m{}; // (paints the letterbox frame itself, like the streaming kernel)
            m.jobs = d_jobs; m.items = d_mitems + L.item_base; m.reqs = d_mreqs + L.item_base; m.arena = c->d_arena; m.nitems = L.nitems;
            m.grid = L.grid; m.wg_lists = L.grid ? d_mwg + L.wg_base : nullptr;
            m.cs = L.k.cs; m.letterbox = L.k.lb; m.ops_in_lds = (L.k.kind >> 8) & 1u; m.wide = (L.k.kind >> 9) & 1u; m.full = (L.k.kind >> 10) & 1u; m.compact = (L.k.kind >> 11) & 1u; m.max_nout = L.max_nout;
            m.spin_limit = mfma_spin_limit; m.err_word = status_dev + 2 * n;
            {
                ProfileScope ps(c, st, 0);
                FL_HIP(c, launch_mfma(m, st), "matrix-pipe resample kernel");
            }
            c->stats.resample_launches++;
            c->stats.mfma_launches++;
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
        if (L.k.kind & kBlurWtileKind) {
            LaunchWtile m{};
            m.jobs = d_jobs; m.items = reinterpret_cast<const WtItem *>(d_mitems + L.item_base); m.arena = c->d_arena; m.nitems = L.nitems;
            m.nslot = L.k.kind & 255u; m.nkmax = kWtOperandRegs / m.nslot; m.letterbox = L.k.lb; m.framed = L.k.lb; m.lds_bytes = (uint32_t)L.lds; m.half_waves = L.k.cs == 1u; // (one-channel pictures: little work per step, two 4-wave workgroups per CU hide each other's barriers; Rgba8 blurs measured 2 % slower that way)
            FL_HIP(c, launch_wtile(m, st), "window-tile matrix-pipe kernel (blur)");
            c->stats.mfma_launches++;
            c->stats.wtile_launches++;
        } else if (L.blur_tiled && !force_generic) {
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
    c->last_status_dev = status_dev;
    c->last_has_results = has_results;
    c->last_has_err_word = has_err_word;
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
    if ((!c->last_has_results && !c->last_has_err_word) || n != c->last_n) { FL_HIP(c, hipStreamSynchronize(st), "batch sync"); return FLGPU_OK; }
    FL_HIP(c, c->h_results.reserve(n * 8 + 8), "pinned result words");
    if (c->last_has_results) FL_HIP(c, hipMemcpyAsync(c->h_results.p, c->last_status_dev, n * 8 + 8, hipMemcpyDeviceToHost, st), "result words D2H");
    else FL_HIP(c, hipMemcpyAsync(static_cast<char *>(c->h_results.p) + n * 8, reinterpret_cast<char *>(c->last_status_dev) + n * 8, 8, hipMemcpyDeviceToHost, st), "error word D2H");
    FL_HIP(c, hipStreamSynchronize(st), "batch sync");
    const uint32_t *r = static_cast<const uint32_t *>(c->h_results.p);
    if (c->last_has_err_word && r[2 * n]) {
        c->set_error((r[2 * n] & FLGPU_DEVERR_MFMA_WAIT) ? "matrix-pipe resample kernel: a bounded wait on an LDS hand-off expired; the batch's pixels are not valid"
                                                         : "device error word set");
        return FLGPU_ERR_DEVICE;
    }
    if (!c->last_has_results) return FLGPU_OK;
    int rc = FLGPU_OK;
    for (size_t i = 0; i < n; ++i) {
        if (c->last_fe[i] == FLGPU_FE_WEBP420 && (r[2 * i] & 1u)) dsts[i].flags |= FLGPU_IMG_HAS_ALPHA;
        if (c->last_fe[i] == FLGPU_FE_JPEG) {
            dsts[i].bytes = r[2 * i + 1];
            if (!r[2 * i + 1]) { c->set_error("encoded stream does not fit the destination"); rc = FLGPU_ERR_BUFFER_TOO_SMALL; }
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
    // JPEG sources: the serial half (parsing + Huffman decoding) runs here, on the host; what is staged is the blob
    std::vector<std::vector<uint8_t>> blobs(n);
    std::vector<JpegBlobHeader> jh(n);
    std::vector<JpegSrc> jhp(n);
    std::vector<std::vector<uint8_t>> iccs(n);
    size_t in_b = 0, out_b = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!srcs[i].data || !dsts[i].data) return FLGPU_ERR_INVALID_ARG;
        int rc = flgpu_plan_output(&ps[i], srcs[i].width, srcs[i].height, srcs[i].channels, &plans[i]);
        if (rc) return rc;
        uint64_t sb = (uint64_t)srcs[i].width * srcs[i].height * srcs[i].channels;
        if (srcs[i].flags & FLGPU_IMG_JPEG_SOURCE) {
            JpegInfo info;
            if (jpeg_parse_info(srcs[i].data, (size_t)srcs[i].capacity, info) != 0) return FLGPU_ERR_INVALID_ARG;
            if (!info.supported) return FLGPU_ERR_UNSUPPORTED;
            rc = jpeg_source_precheck(c, &srcs[i], info);
            if (rc) return rc;
            blobs[i].resize(jpeg_blob_bound(info));
            size_t used = 0;
            rc = jpeg_source_to_blob(c, &srcs[i], blobs[i].data(), blobs[i].size(), &jh[i], &used);
            if (rc) return rc;
            stage_of(blobs[i].data(), jh[i], jhp[i].stage);
            blobs[i].resize(used);
            jhp[i].hdr = &jh[i];
            if (jh[i].nc == 4 && c->cfg.use_embedded_profile && !info.icc.empty()) { iccs[i].swap(info.icc); jhp[i].icc = iccs[i].data(); jhp[i].icc_len = iccs[i].size(); }
            c->stats.jpeg_file_bytes += srcs[i].capacity;
            sb = used;
        } else
        if (srcs[i].capacity < sb) return FLGPU_ERR_INVALID_ARG;
        if (dsts[i].capacity < plans[i].out_bytes && ps[i].front_end != FLGPU_FE_JPEG) return FLGPU_ERR_BUFFER_TOO_SMALL;
        dsrc[i] = srcs[i]; ddst[i] = dsts[i];
        dsrc[i].data = reinterpret_cast<uint8_t *>(in_b); dsrc[i].capacity = sb; in_b += align_up(sb, 256);
        const uint64_t ob = plans[i].max_out_bytes; // JPEG: the format's worst case, so the device side never overflows
        ddst[i].data = reinterpret_cast<uint8_t *>(out_b); ddst[i].capacity = ob; out_b += align_up(ob, 256);
    }
    FL_HIP(c, c->d_in.reserve(in_b), "device input staging");
    FL_HIP(c, c->d_out.reserve(out_b), "device output staging");
    FL_HIP(c, c->h_stage_in.reserve(in_b), "pinned input staging");
    FL_HIP(c, c->h_stage_out.reserve(out_b), "pinned output staging");
    hipStream_t st = c->stream;
    for (size_t i = 0; i < n; ++i) {
        const size_t off = reinterpret_cast<size_t>(dsrc[i].data);
        memcpy(static_cast<char *>(c->h_stage_in.p) + off, jhp[i].hdr ? blobs[i].data() : srcs[i].data, dsrc[i].capacity);
        dsrc[i].data = static_cast<uint8_t *>(c->d_in.p) + off;
        ddst[i].data = static_cast<uint8_t *>(c->d_out.p) + reinterpret_cast<size_t>(ddst[i].data);
    }
    FL_HIP(c, hipMemcpyAsync(c->d_in.p, c->h_stage_in.p, in_b, hipMemcpyHostToDevice, st), "H2D");
    { int drc = decode_jpeg_sources(c, n, dsrc.data(), jhp.data(), st); if (drc) return drc; }
    int rc = run_batch_device(c, n, dsrc.data(), ps, false, ddst.data(), st);
    if (rc) return rc;
    FL_HIP(c, hipMemcpyAsync(c->h_stage_out.p, c->d_out.p, out_b, hipMemcpyDeviceToHost, st), "D2H");
    rc = collect_results(c, n, ddst.data(), st);
    {   // pictures the device entropy decoder gave up on (states that did not settle, an invalid code word): the whole batch once
        // more with the host decoder -- which either decodes them or says what is wrong with the file
        std::vector<uint8_t> bad;
        const int nbad = entropy_failures(c, n, bad, st);
        if (nbad < 0) return -nbad;
        if (nbad > 0 && !tl_force_host_huffman) {
            struct Force { Force() { tl_force_host_huffman = true; } ~Force() { tl_force_host_huffman = false; } } force;
            return run_batch_host(c, n, srcs, ps, dsts);
        }
    }
    for (size_t i = 0; i < n; ++i) {
        const size_t off = static_cast<uint8_t *>(ddst[i].data) - static_cast<uint8_t *>(c->d_out.p);
        if (ddst[i].bytes > dsts[i].capacity) { // only now is the length of an encoded stream known
            c->set_error("encoded stream does not fit the destination");
            if (rc == FLGPU_OK) rc = FLGPU_ERR_BUFFER_TOO_SMALL;
            dsts[i].bytes = 0; dsts[i].flags = ddst[i].flags;
            continue;
        }
        memcpy(dsts[i].data, static_cast<char *>(c->h_stage_out.p) + off, std::min<uint64_t>(ddst[i].bytes, ddst[i].capacity));
        dsts[i].width = ddst[i].width; dsts[i].height = ddst[i].height; dsts[i].channels = ddst[i].channels; dsts[i].flags = ddst[i].flags;
        dsts[i].bytes = ddst[i].bytes;
    }
    return rc;
}

} // namespace fl

// ---- diagnostics -----------------------------------------------------------------------------------------------------------------
extern "C" int flgpu_debug_assign_items(uint32_t pictures, uint32_t strips, uint32_t tiles, uint32_t workgroups, uint32_t capacity, uint32_t *job_of, uint32_t *strip_of,
                                        uint32_t *tile0_of, uint32_t *tile1_of, uint32_t *lists, uint32_t *nitems)
{
    if (!pictures || !strips || !tiles || !workgroups || !job_of || !strip_of || !tile0_of || !tile1_of || !lists || !nitems) return FLGPU_ERR_INVALID_ARG;
    MfmaPlan plan;
    plan.ok = true; plan.vplan_off = 7u;
    for (uint32_t s = 0; s < strips; ++s) plan.strip_offs.push_back(1000u + s);
    for (uint32_t t = 0; t < tiles; ++t) plan.tiles.push_back({3u * t, 3u * t + 4u}); // (a tile needs five K-blocks, a new one ends every third)
    std::vector<MfmaItem> v;
    for (uint32_t p = 0; p < pictures; ++p)
        for (MfmaItem m : plan.items_for(1)) { m.job = p; v.push_back(m); }
    uint32_t n = (uint32_t)v.size();
    std::stable_sort(v.begin(), v.end(), [](const MfmaItem &a, const MfmaItem &b) { return a.kb1 - a.kb0 > b.kb1 - b.kb0; });
    xcd_interleave(v.data(), n, [](const MfmaItem &x) { return x.kb1 - x.kb0; });
    const uint32_t G = std::max(1u, std::min(n, workgroups));
    std::vector<uint32_t> l;
    assign_items(v, 0, n, G, &plan, l);
    *nitems = n;
    if (n > capacity) return FLGPU_ERR_BUFFER_TOO_SMALL;
    for (uint32_t k = 0; k < n; ++k) { job_of[k] = v[k].job; strip_of[k] = v[k].strip_off - 1000u; tile0_of[k] = v[k].tile0; tile1_of[k] = v[k].tile1; }
    for (uint32_t k = 0; k < 2u * G; ++k) lists[k] = l[k];
    return FLGPU_OK;
}
