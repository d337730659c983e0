// fl_context.h -- internals of the host runtime behind the C ABI, shared by its translation units:
//   fl_context.cpp   context lifetime, table arena + caches, stats, table export / import
//   fl_batch.cpp     batch planner + launcher (device batches, host batches)
//   fl_queue.cpp     persistent request-batching queue, lanes, sharding of flushed batches across the devices of a node
//   fl_cmyk_ctx.cpp  CMYK device-link tables and their distribution to the devices (RCCL broadcast)
//
// There is deliberately NO CPU fallback anywhere in the runtime: if HIP is unavailable or a launch fails the
// caller gets an error code (reference behaviour on any Err from process_image is the fallback image / 500,
// src/main.rs:185-195).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/fanlin_gpu.h"
#include "fl_abi.h"
#include "fl_cmyk.h"
#include "fl_jpeg_tables.h"
#include "fl_jpegdec.h"
#include "fl_kernels.h"
#include "fl_mfma.h"
#include "fl_tables.h"
#include "fl_wtile.h"

namespace fl {

struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes);
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes);
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// descriptor staging slot: pinned host copy + device copy, guarded by an event
struct DescSlot {
    PinnedBuf host;
    DeviceBuf dev;
    hipEvent_t done = nullptr;
    hipEvent_t uploaded = nullptr; // the block has reached the device (recorded on the context's upload stream)
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

struct MfmaPlanKey {
    AxisKey v, h;
    uint32_t cx, cy, cw, ch, cs;
    uint32_t arith; // MfmaArith: the two arithmetics of the kernel have different tables
    bool operator<(const MfmaPlanKey &o) const { return std::tie(v, h, cx, cy, cw, ch, cs, arith) < std::tie(o.v, o.h, o.cx, o.cy, o.cw, o.ch, o.cs, o.arith); }
};

// Tables of the matrix-pipe resample kernel for one geometry (fl_mfma.h) and its workgroup lists (job field unset), one per
// band count: the tables do not depend on how a picture is cut into bands, so they are built and stored once.
struct MfmaPlan {
    bool ok = false;
    uint32_t vplan_off = 0;
    std::vector<uint32_t> strip_offs;
    std::vector<HostMfmaPlan::Tile> tiles;
    std::map<uint32_t, std::vector<MfmaItem>> items_by_bands;
    const std::vector<MfmaItem> &items_for(uint32_t nbands);
    uint32_t max_nout = 0;
    bool ops_in_lds = false;
    bool wide = false;       // wide layout (fl_mfma.h): more outputs per strip, operands from the L2
    bool full = true;        // built for the full-width arithmetic (MFMA_ARITH_FULL)
    bool compact = false;    // full-width arithmetic: strips of at most 300 outputs, the compact LDS layout (56 operands fit)
    bool arena_full = false; // the tables did not fit what is left of the arena: not cached, the caller resets the arena and plans again
};

// Tables of the window-tile matrix-pipe kernel for one geometry (fl_wtile.h): one block in the arena.  Keyed like the other plans;
// a Gaussian blur is the same thing with Gaussian axes of equal in and out size.
struct WtPlan {
    bool ok = false;
    bool arena_full = false; // the tables did not fit what is left of the arena: not cached, the caller resets the arena and plans again
    uint32_t off = 0;        // arena word offset of the WtHeader
    uint32_t nslot = 0, nkmax = 0, n_mt = 0, n_strips = 0, lds_bytes = 0;
    void items_for(uint32_t nbands, uint32_t job, std::vector<MfmaItem> &out) const; // (WtItem and MfmaItem share the batch's item array: same size)
};

struct PinBlock {
    void *p = nullptr;
    size_t cap = 0;
};

// One queued single-image request (flgpu_transform): lives on its caller's stack until `done`.
struct Request {
    const flgpu_image *src;
    const flgpu_params *p;
    flgpu_image *dst;
    PinBlock in, out;      // pinned staging filled / drained by the CALLER thread (parallel memcpy)
    uint64_t src_bytes = 0, out_bytes = 0;
    uint64_t weight = 0;   // algorithmic bytes: W*H*C + out_bytes (shard balancing)
    bool jpeg = false;     // FLGPU_IMG_JPEG_SOURCE: `in` holds the coefficient blob the caller's thread decoded, jhdr its header
    JpegBlobHeader jhdr;
    JpegHuffStage jstage{}; // jhdr.magic == kJhMagic: `in` holds the staged entropy-coded segment, decoded on the device
    std::vector<uint8_t> icc; // four-component source + use_embedded_profile: the file's own ICC profile
    uint64_t file_bytes = 0;
    int status = 0;
    bool done = false;
    std::condition_variable cv; // the caller waits here (qmu): a lane wakes the callers of ITS batch, not every waiting caller
};

// Test and experiment switches of a context (flgpu_debug_set / flgpu_debug_get, include/fanlin_gpu.h).  One block per ROOT context, shared
// with its queue lanes and device shards; every field is an atomic read with relaxed ordering, so a caller may flip a switch
// while other threads plan batches (a batch sees the old or the new value, never a torn one).  The process environment is read
// exactly once, in flgpu_create (FLGPU_<KEY IN CAPITALS> seeds the switch of that name): a server that calls setenv() on another
// thread cannot race the library, and a stray variable set after start-up cannot change output bytes.
enum DebugKey : uint32_t {
    DBG_NO_MFMA = 0,              // the f32 streaming kernel also where the matrix-pipe kernel would run
    DBG_FORCE_GENERIC,            // the two-pass kernels instead of every fused one
    DBG_NO_WTILE,                 // mild ratios, up-scales and blurs on the f32 vector kernels
    DBG_WTILE_BLUR_ALWAYS,        // one-channel blurs on the window-tile kernel too
    DBG_WTILE_FIRST,              // the window-tile kernel is asked before the streaming matrix-pipe kernel at every ratio
    DBG_MFMA_ARITH,               // 0 = full-width arithmetic, 1 = the packed arithmetic of rounds 2-3
    DBG_FORCE_BANDS,              // row bands per picture (0 = the planner's choice)
    DBG_NO_TILE,                  // the two-pass resample through an f32 picture in HBM instead of the LDS tile
    DBG_NO_PLACE4,                // placement one pixel per thread
    DBG_HOST_HUFFMAN,             // JPEG sources are entropy-decoded on the caller's thread
    DBG_DEVICE_HUFFMAN_ALWAYS,    // ... on the device whenever the file allows it
    DBG_DEVICE_HUFFMAN_MIN_BYTES, // files below this size are not worth staging (default 16384)
    DBG_MFMA_SPIN_LIMIT,          // bound of the matrix-pipe kernel's LDS-counter waits (default kMfmaDefaultSpinLimit; 0 = every wait expires)
    DBG_DEBUG_MFMA,               // print every matrix-pipe plan
    DBG_DEBUG_JH,                 // print why a staged file went back to the host decoder
    DBG_COUNT
};
struct DebugSwitches {
    std::atomic<int64_t> v[DBG_COUNT];
    int64_t initial[DBG_COUNT]; // what flgpu_create left (defaults, or the environment's values): "reset" goes back to these
    DebugSwitches();
    int64_t get(DebugKey k) const { return v[k].load(std::memory_order_relaxed); }
    bool on(DebugKey k) const { return get(k) != 0; }
};
extern const char *const kDebugKeyNames[DBG_COUNT];

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// Worst case of a baseline JPEG stream of the encoder in fl_jpeg.hip: header + 2 x 208 bytes per 8x8 block and
// component (every byte stuffed).  No picture can exceed it, so a destination of this size never overflows.
inline uint64_t jpeg_worst_bytes(uint32_t plane_w, uint32_t plane_h)
{
    return 1024ull + 2ull * kJpegMaxUnitBytes * 3ull * (plane_w / 8u) * (plane_h / 8u);
}

} // namespace fl

struct flgpu_ctx {
    int device = 0;
    uint32_t cu_count = 256; // compute units of the device: the persistent matrix-pipe launch has one workgroup per CU
    hipStream_t stream = nullptr;
    flgpu_config cfg{};
    std::shared_ptr<fl::DebugSwitches> dbg; // the root context's switch block, shared with its lanes and shards
    std::mutex mu; // planning + launching on THIS context's stream is serialised (lanes and device shards have their own)

    // read-only table arena
    std::vector<uint32_t> h_arena;
    uint32_t *d_arena = nullptr;
    size_t arena_cap_words = 0, arena_uploaded = 0;
    std::map<fl::AxisKey, uint32_t> axis_off;
    std::map<fl::AxisKey, fl::HostAxis> axis_host;
    std::map<fl::StreamPlanKey, fl::StreamPlan> stream_plans;
    std::map<fl::MfmaPlanKey, fl::MfmaPlan> mfma_plans;
    std::map<fl::MfmaPlanKey, fl::WtPlan> wtile_plans;   // (arith field unused)
    std::map<std::tuple<fl::AxisKey, uint32_t, uint32_t>, uint32_t> tile_vplans;     // tiled two-pass kernel: dense vertical weights per band of 8 output rows, per (axis, first kept row, rows)
    std::map<std::tuple<fl::AxisKey, fl::AxisKey, uint32_t>, uint32_t> blur_plans; // blur kernel table blocks per (vertical, horizontal) Gaussian axis
    uint32_t gamma_off = 0;

    fl::DescSlot slots[4];
    int next_slot = 0;
    fl::DeviceBuf d_mid, d_tmp_a, d_tmp_b, d_tmp_o, d_tmp_al;
    uint32_t *last_status_dev = nullptr; // result words + device error word of the batch enqueued last: they live at the end of its descriptor block
    hipStream_t up_stream = nullptr;     // descriptor blocks travel on a stream of their own, so that batch i + 1's block is on the device before batch i's kernels end
    fl::DeviceBuf d_in, d_out;
    fl::DeviceBuf d_dec, d_decjobs;                    // JPEG decode: planes + decoded pixels of a batch, job descriptors
    fl::PinnedBuf h_decjobs;
    // entropy decoding on the device (fl_jpeghuff_dev.hip): blobs + per-subsequence scratch of a batch, descriptors, per-picture error words
    fl::DeviceBuf d_jh, d_jhjobs, d_jherr;
    fl::PinnedBuf h_jhjobs, h_jherr;
    std::vector<int32_t> last_jh_slot; // per image of the batch decoded last: index of its error word, -1 = not entropy-decoded on the device
    uint32_t last_jh_n = 0;
    fl::DeviceBuf d_jpeg_coef, d_jpeg_off, d_jpeg_raw; // JPEG encode scratch (fl_jpeg.hip): block meta words, bit offsets, AC bits
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> jpeg_tables; // (w, h, quality) -> arena offset of header + q tables
    // per-image result words of the most recent device batch: [2i] flags (bit 0: non-opaque alpha seen by the WebP front
    // end, FL_JPEG_RESULT_OVERFLOW), [2i + 1] bytes of an encoded stream
    size_t last_n = 0;
    bool last_has_results = false;
    bool last_has_err_word = false; // a kernel of the batch reports through the device error word that follows the result words
    std::vector<uint8_t> last_fe;
    fl::PinnedBuf h_results;
    fl::PinnedBuf h_stage_in, h_stage_out;
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr;

    flgpu_stats stats{};
    struct Pending { hipEvent_t a, b; int kind; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;

    std::mutex err_mu;
    std::string last_error; // guarded by err_mu

    // CMYK -> sRGB device-link tables (kCmykGrid^4 nodes of 4 x u16): the boot-time default (main.rs:74-76) and a
    // small cache of tables baked from embedded profiles, keyed by a hash of the profile bytes (handler.rs:446-458
    // rebuilds the lcms2 transform, 40 ms, on every such request)
    struct Clut { fl::DeviceBuf dev; std::vector<uint16_t> host; uint64_t stamp = 0; };
    Clut cmyk_default;
    bool has_cmyk_default = false;
    std::map<uint64_t, Clut> cmyk_embedded;
    uint64_t cmyk_stamp = 0;
    uint64_t cmyk_pin_floor = 0;   // tables stamped later than this were handed out in the current batch: not evictable

    // pinned staging blocks recycled between requests (power-of-two size classes)
    std::mutex pin_mu;
    std::multimap<size_t, void *> pin_free;

    // ---- devices of the node (flgpu_config::n_devices > 1) ------------------------------------------------------
    // One context, several GPUs (reference analogue: one Arc<State> shared by all workers, src/main.rs:108-112): the
    // parent owns the request queue; `shard_ctx[k]` is a child context bound to devices[k] (own stream, arena, scratch)
    // that runs shard k of every batch entry point.  Empty for a single-device context (the parent is the device context).
    flgpu_ctx *clut_owner = nullptr; // a queue lane: the context on ITS device that holds the configured CMYK table (the parent, or the parent's shard context of that device)
    std::vector<int> devices;
    std::vector<flgpu_ctx *> shard_ctx;
    std::vector<std::pair<size_t, size_t>> last_shards; // [first, last) image range of each shard in the last device batch
    void *rccl = nullptr;                               // fl_cmyk_ctx.cpp: RCCL communicators of the node (lazy)
    bool rccl_failed = false;                           // ... or the fact that librccl could not be loaded / initialised (not retried)
    int cmyk_how = 0;                                   // how the default CMYK table last reached the shards: 2 = RCCL broadcast, 1 = copies

    // ---- request queue ------------------------------------------------------------------------------------------
    // Queued single-image requests are served by worker threads, each driving its own child context ("lane": own
    // stream, scratch, table cache) on one of the context's devices: while one lane's batch is on the PCIe link / in
    // kernels, another lane is already collecting and uploading the next batch.  A worker that collects a flushed batch
    // splits it into one contiguous shard per device, balanced by algorithmic bytes, keeps its own device's shard and
    // hands the others to the inboxes of the other devices' lanes.
    std::vector<std::thread> workers;
    std::vector<flgpu_ctx *> lanes;    // capacity reserved at creation; entries published through n_lanes
    std::atomic<size_t> n_lanes{0};
    std::vector<std::deque<std::vector<fl::Request *>>> inbox; // per device slot: shards waiting for a lane of that device (qmu)
    bool collecting = false; // a worker is gathering a batch (one collector at a time keeps batches large)
    std::atomic<int> staging{0}; // callers currently copying their source into pinned memory, i.e. about to enqueue
    // admission: callers beyond a few batches' worth wait BEFORE staging (a thousand threads each copying megabytes
    // into pinned memory only evict each other's buffers and starve the lane threads of CPU time)
    std::mutex adm_mu;
    std::condition_variable adm_cv;
    uint32_t admitted = 0;
    std::mutex dec_mu;                 // host-side JPEG decoding: at most dec_limit callers at a time (flgpu_config.decode_threads)
    std::condition_variable dec_cv;
    uint32_t decoding = 0, dec_limit = 0;
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<fl::Request *> queue;
    bool stop = false;
    bool worker_started = false;

    void set_error(const std::string &s) { std::lock_guard<std::mutex> g(err_mu); last_error = s; }
    std::string get_error() { std::lock_guard<std::mutex> g(err_mu); return last_error; }
    int fail(hipError_t e, const char *what)
    {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        set_error(buf);
        return e == hipErrorOutOfMemory ? FLGPU_ERR_OOM : FLGPU_ERR_DEVICE;
    }
    uint32_t n_dev() const { return devices.empty() ? 1u : (uint32_t)devices.size(); }
};

#define FL_HIP(ctx, call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) return (ctx)->fail(e__, what); } while (0)

namespace fl {

constexpr size_t kArenaWords = (size_t)16 << 20; // 64 MiB of tables

// ---- fl_context.cpp: arena + caches -------------------------------------------------------------------------------
uint32_t arena_append(flgpu_ctx *c, const void *data, size_t words, size_t align_words = 4);
void arena_reset(flgpu_ctx *c);
int arena_flush(flgpu_ctx *c, hipStream_t st);
uint32_t get_axis(flgpu_ctx *c, uint32_t in, uint32_t out, Filter f, float sigma, AxisKey *key_out, const HostAxis **host_out);
MfmaPlan *get_mfma_plan(flgpu_ctx *c, const AxisKey &vk, const HostAxis &va, const AxisKey &hk, const HostAxis &ha,
                              uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t cs, MfmaArith arith = MFMA_ARITH_FULL);
WtPlan *get_wtile_plan(flgpu_ctx *c, const AxisKey &vk, const HostAxis &va, const AxisKey &hk, const HostAxis &ha,
                       uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t cs);
const StreamPlan *get_stream_plan(flgpu_ctx *c, const AxisKey &vk, const HostAxis &va, const AxisKey &hk, const HostAxis &ha,
                                  uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t nbands, uint32_t cs, uint32_t pre);
hipEvent_t get_event(flgpu_ctx *c);
void resolve_pending(flgpu_ctx *c);
flgpu_ctx *create_child(flgpu_ctx *parent, int device); // a lane / device shard: same config, one device, no queue of its own

// roctx ranges around the phases of a batch (tracing): visible to `rocprofv3 --marker-trace`; libroctx64 is
// loaded on demand and only when FLGPU_ROCTX=1, so production runs pay nothing.
struct RoctxRange {
    explicit RoctxRange(const char *name);
    ~RoctxRange();
    bool on = false;
};

struct ProfileScope {
    flgpu_ctx *c; hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr;
    ProfileScope(flgpu_ctx *c_, hipStream_t st_, int kind_);
    ~ProfileScope();
};

// ---- fl_batch.cpp ----------------------------------------------------------------------------------------------------
int run_batch_device(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, bool same_params,
                     flgpu_image *dsts, hipStream_t st);
int collect_results(flgpu_ctx *c, size_t n, flgpu_image *dsts, hipStream_t st);
int run_batch_host(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, flgpu_image *dsts);
// room to give an encoded result on the device: the planning bound, or the format's worst case if the caller offers it
uint64_t staged_out_bytes(const flgpu_params &p, const flgpu_plan &plan, uint64_t dst_capacity);
// JPEG sources of a batch: dsrc[i].data = DEVICE copy of the coefficient blob whose header (host copy) is hdrs[i], or
// hdrs[i] == nullptr for ordinary pixel sources.  Runs the decode kernels into scratch and points dsrc[i] at the pixels.
struct JpegSrc {
    const JpegBlobHeader *hdr = nullptr;
    const uint8_t *icc = nullptr;
    size_t icc_len = 0;
    JpegHuffStage stage{}; // hdr->magic == kJhMagic: the staged segment's description (a host copy: the blob itself is on its way to the device)
};
int decode_jpeg_sources(flgpu_ctx *c, size_t n, flgpu_image *dsrc, const JpegSrc *srcs, hipStream_t st);
// After the batch decode_jpeg_sources fed has completed on `st`: bad[i] = 1 where the device entropy decoder gave up on picture i
// (its states did not settle, or the stream holds an invalid code word): the caller decodes that file on the host instead.
// Returns the number of such pictures, or a negative FLGPU_ERR_* .
int entropy_failures(flgpu_ctx *c, size_t n, std::vector<uint8_t> &bad, hipStream_t st, bool fetched = false);
int entropy_failures_fetch(flgpu_ctx *c, size_t n, hipStream_t st);
// internal status of a queued request: run it again with the host entropy decoder
constexpr int FL_STATUS_RETRY_HOST_HUFFMAN = 1000;
// Host half for one source: parses + Huffman-decodes `src` (a JPEG file) into `blob`; validates the declared size.
int jpeg_source_to_blob(flgpu_ctx *c, const flgpu_image *src, uint8_t *blob, size_t cap, JpegBlobHeader *hdr, size_t *used, bool host_huffman = false);
// set while a request is run again after the device entropy decoder gave up on its file (this thread's JPEG sources are then decoded on the host)
extern thread_local bool tl_force_host_huffman;
int device_huffman_policy(const flgpu_ctx *c, uint64_t file_bytes); // fl_batch.cpp
inline void stage_of(const uint8_t *staged_blob, const JpegBlobHeader &hdr, JpegHuffStage &out)
{
    if (hdr.magic == kJhMagic) memcpy(&out, staged_blob + sizeof(JpegBlobHeader), sizeof(out));
}
// fl_cmyk_ctx.cpp: the device-link table for one conversion on context c: the embedded profile's if given and usable
// (baked once, cached on c), else the configured one (c's own, or its clut_owner's)
int select_clut(flgpu_ctx *c, const uint8_t *icc, uint64_t icc_len, const void **dev);
int clut_batch_begin(flgpu_ctx *c);
int jpeg_source_precheck(flgpu_ctx *c, const flgpu_image *src, const fl::JpegInfo &info);

// ---- fl_queue.cpp ----------------------------------------------------------------------------------------------------
// contiguous split of n weighted items into n_shards shards of about equal weight: shard_of[i] is non-decreasing
void split_by_weight(const uint64_t *weight, size_t n, uint32_t n_shards, uint32_t *shard_of);
PinBlock pin_acquire(flgpu_ctx *c, size_t bytes);
void pin_release(flgpu_ctx *c, PinBlock &b);
void stop_queue(flgpu_ctx *c); // joins the workers and destroys the lanes

// ---- fl_cmyk_ctx.cpp ---------------------------------------------------------------------------------------------------
void release_cmyk(flgpu_ctx *c);

} // namespace fl
