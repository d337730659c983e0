// fl_mfma.hip -- the matrix-pipe resample kernel for gfx950 (design and arithmetic: fl_mfma.h).
// Reference: image 0.25.6 imageops/sample.rs vertical_sample + horizontal_sample behind DynamicImage::resize_exact
// (src/handler.rs:229-247 of the reference calls resize / resize_to_fill).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <map>
#include <vector>

#include "fl_mfma.h"
#include "fl_pixel.h"

namespace fl {

namespace {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Pointers read from a descriptor have no address space the compiler could know, so stores through them become flat_store --
// which counts on lgkmcnt as well as vmcnt and completes out of order with LDS traffic: one pending pixel store turns every LDS
// wait behind it into lgkmcnt(0).  The destination is device memory by contract; saying so makes them global_store.
typedef __attribute__((address_space(1))) uint32_t *gptr32;
typedef __attribute__((address_space(1))) uint8_t *gptr8;

constexpr uint32_t THREADS = kMfmaWaves * 64;
constexpr uint32_t RING_BYTES = kMfmaKRows * kMfmaWaveCols; // one K-block of one wave
constexpr uint32_t CNT_BYTES = kMfmaCntBytes;               // add_cnt[2], conv_cnt[2], then two conversion contexts of 8 words (the item a pending tile belongs to: destination, row pitch, first pixel, fill)

__shared__ __attribute__((aligned(16))) uint8_t mfma_ring[kMfmaWaves * RING_BYTES]; // the rows' landing zone (static), everything else is dynamic
extern __shared__ __attribute__((aligned(16))) uint8_t mfma_lds[];

// (a << 8) + b, opaque to the optimiser: left alone hipcc re-associates the digit sums' Horner form into two shifts and a
// three-operand add.  (An empty asm, not an asm instruction: the operands come straight out of the matrix unit and only
// instructions the compiler knows get the wait states that requires.)
__device__ __forceinline__ uint32_t shl8_add(uint32_t a, uint32_t b)
{
    uint32_t r = (a << 8) + b;
    asm("" : "+v"(r)); // (not volatile: the scheduler may interleave the digit sums of different tiles)
    return r;
}

__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0f70); }   // vmcnt(0)
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); } // lgkmcnt(0)

// Two counters per LDS output tile order the waves' traffic on it (add_cnt: waves that have added their sums, conv_cnt: waves
// that have converted their rows).  Everything they guard lives in LDS, so the fences name the local address space only: a
// plain workgroup-scope acquire or release also drains vmcnt, i.e. waits for the K-block that has just been requested from
// HBM -- the one latency this kernel is built to hide.  Every wait is bounded (spin_limit): by construction it is short or
// none (the slowest wave never waits), but a wave that gives up sets the launch's error word and the host returns
// FLGPU_ERR_DEVICE for the batch instead of pixels that were never synchronised.

// CS: interleaved 8-bit channels of the source (the vertical pass does not care; the horizontal tables carry the channel
// structure; only the last step, bytes -> destination pixel, is written per channel count).
// LB: the destination is Rgba8 with the picture placed on a fill frame.  HLDS: the strip's horizontal operands sit in LDS.
// LAYOUT: 0 = output tiles of up to 408 outputs per row, 1 (never with HLDS) = wide: the LDS the operands would take goes to
// output tiles of up to 728 outputs, 2 = compact: up to 300 outputs, which leaves room for 56 operands (fl_mfma.h).
// FW: full-width arithmetic (fl_mfma.h MFMA_ARITH_FULL) -- bytes as f16 subnormals x three-term weights, a 23-bit intermediate in
// three byte planes x three weight digits; otherwise the packed arithmetic of rounds 2-3.  With FW, HLDS only says that the
// launch's LDS has an operand area: whether a strip's operands live there is the strip's own flag (MfmaStrip::lds_ops).
// Persistent workgroups (round 5, full-width arithmetic): the launch has one workgroup per CU and each walks its own list of items
// (consecutive in the item array; the host deals them out, fl_batch.cpp assign_items) -- with 160 KB of LDS and 512 x 256 registers a CU holds ONE workgroup, so with one item per
// workgroup every item paid a dispatch, a prologue, the latency of its first K-block and the drain of its last tile with nothing in
// flight for that CU (15-17 us of ~130).  Now the first K-block of item n + 1 is requested in the LAST pass of item n (the wave's
// 8 KB of LDS are free once the transposed reads have returned).  Two kinds of transition:
//   light   the next item has the same strip, plan and K-blocks (another picture of the same geometry: the host orders a uniform
//           launch so that a workgroup's items share their strip, fl_batch.cpp persistent_order): NOTHING is waited for -- the walk
//           goes on into the next picture's rows, the last tile's rows are converted in the passes that follow like any other
//           tile's (what the conversion needs of the old item -- destination, pitch, first pixel -- sits in an LDS context, two of
//           them, by item parity), the frame is painted and the new context written behind the first request;
//   heavy   anything else: this wave's rows of the last tile once all eight waves have added (a bounded spin), then descriptors,
//           tile table, operands, a barrier -- the first K-block flies under all of it.
// The LDS counters run on across items (a tile's global number = tiles of earlier items + its number in this one); the packed
// arithmetic keeps one item per workgroup (its launch has gridDim.x = nitems).
template <int CS, bool LB, bool HLDS, int LAYOUT, bool FW>
__global__ __launch_bounds__(THREADS, 1) void resample_mfma_kernel(const Job *__restrict__ jobs, const MfmaItem *__restrict__ items, const MfmaReq *__restrict__ reqs, const uint32_t *__restrict__ wg_lists,
                                                                   const uint32_t *__restrict__ arena, uint32_t nitems, uint32_t ot_words, uint32_t spin_limit,
                                                                   uint32_t *__restrict__ err_word
#ifdef FL_MFMA_STAMPS
                                                                   , unsigned long long *__restrict__ stamps
#endif
)
{
#ifdef FL_MFMA_STAMPS // development aid (-DFL_MFMA_STAMPS): 100 MHz timestamps of wave 0 and wave 5 of workgroups 0, 100, 200: [wg][wave][item][event]
    unsigned long long *my_stamps = nullptr;
    uint32_t stamp_item = 0;
#define FL_STAMP(ev_) do { if (my_stamps && stamp_item < 16u && lane == 0) my_stamps[stamp_item * 8u + (ev_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FL_STAMP(ev_) do { } while (0)
#endif
    // experiments only (-DFL_ABLATE=mask, tools/build_ablate.sh): 1 = no horizontal MFMAs / LDS adds, 2 = no conversion of
    // finished tiles, 32 = conversion without its global stores, 4 = no horizontal stage at all, 8 = no vertical MFMAs,
    // 128 = no LDS adds, 8192 (with 4) = the vertical pass alone, its sums kept alive
#ifdef FL_ABLATE
    constexpr uint32_t ablate = FL_ABLATE;
#else
    constexpr uint32_t ablate = 0;
#endif
    // experiments (-DFL_VARIANT=mask): 16 = a chunk's planes are made beside the matrix instructions of the chunk before (spills),
    // 8 (launcher) = one item per workgroup; -DFL_NINE_PRODUCTS: the ninth digit product, plane 0 x digit 0, is computed too (round 4)
    // (tried and dropped, profiles/r05_kernel_experiments.txt: waves 4-7 started late so that their stages run beside the other waves'
    // vertical passes -- paid for its own delay when the delay was per item, nothing once it was per workgroup)
    // wave priorities on the SIMD (s_setprio): 3 from a K-block's transposed reads to the next request (round 3), then FL_PRIO_VERT through the
    // vertical pass and FL_PRIO_STAGE through a tile stage.  Round 5: 2 / 1 instead of 0 / 0 -- the stage's dependent chains go ahead of the
    // other wave's vertical pass: -0.6 % (1.5405 -> 1.5302 ms; 3 / 1, 3 / 2, 3 / 0 and 1 / 0 measured the same, profiles/r05_kernel_experiments.txt)
#ifndef FL_PRIO_STAGE
#define FL_PRIO_STAGE 2
#endif
#ifndef FL_PRIO_VERT
#define FL_PRIO_VERT 1
#endif
#ifdef FL_VARIANT
    constexpr uint32_t variant = FL_VARIANT;
#else
    constexpr uint32_t variant = 0;
#endif
#ifdef FL_NINE_PRODUCTS
    constexpr bool EIGHT = false;
#else
    constexpr bool EIGHT = true; // fl_mfma.h: the full-width arithmetic's horizontal pass
#endif
    const uint32_t tid = threadIdx.x, lane = tid & 63u, g = lane >> 4, i = lane & 15u;
    uint32_t wg_error = 0u; // a bounded wait expired in this lane's wave
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The item this workgroup is on: the descriptors are read again at the top of every trip of the item loop below (nothing but the
    // item's number is carried around the loop: the kernel has no registers to spare), everything derived from them is set in `enter_item`.
    // (persistent launch: this workgroup's list of items, consecutive in the launch's item array; else one item)
    uint32_t item = wg_lists ? wg_lists[2u * blockIdx.x] : blockIdx.x;
    uint32_t items_left = wg_lists ? wg_lists[2u * blockIdx.x + 1u] : 1u;
    if (!items_left) return;
#ifdef FL_MFMA_STAMPS
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    if ((blockIdx.x == 0u || blockIdx.x == 8u || blockIdx.x == 16u) && (wave == 0u || wave == 5u))
        my_stamps = stamps + ((blockIdx.x / 8u) * 2u + (wave ? 1u : 0u)) * 16u * 8u;
#endif
    MfmaItem it;
    Job jb;
    MfmaVPlan vp;
    MfmaStrip sp;
    auto load_item = [&]() __attribute__((always_inline)) {
        it = items[item];
        jb = jobs[it.job];
        vp = *reinterpret_cast<const MfmaVPlan *>(arena + it.vplan_off);
        sp = *reinterpret_cast<const MfmaStrip *>(arena + it.strip_off);
    };
    constexpr uint32_t np = mfma_out_pitch(LAYOUT); // words per output-tile row (compile-time: the row offsets of the LDS adds become immediates); column sp.nout is a dummy

    // (packed arithmetic: two output tiles that alternate; full width: ONE -- a wave converts its rows of a tile in the pass after
    // the tile's stage, see try_convert -- and the 20-27 KB go to the operand area, where the flagship's 69 operands then fit)
    constexpr uint32_t NBUF = FW ? 1u : 2u;
    uint32_t *otile = reinterpret_cast<uint32_t *>(mfma_lds);
    uint32_t *add_cnt = reinterpret_cast<uint32_t *>(mfma_lds + NBUF * ot_words * 4u);
    uint32_t *conv_cnt = add_cnt + 2;
    const u32x4 *ops_lds = reinterpret_cast<const u32x4 *>(mfma_lds + NBUF * ot_words * 4u + CNT_BYTES);
    uint8_t *ring = mfma_ring + wave * RING_BYTES;
    const u32x4 *ops_glb = nullptr; // (set per item)

    // once per workgroup: the output tile(s) and the counters start at zero (afterwards the conversions leave the tile zeroed and
    // the counters run on)
    for (uint32_t k = tid; k < NBUF * ot_words + CNT_BYTES / 4u; k += THREADS) otile[k] = 0u;

    // Letterbox frame: every workgroup paints the part next to its own band and strip (same split as the streaming kernel).
    auto paint_frame = [&]() __attribute__((always_inline)) {
        const uint32_t y_first = vp.y0 + 16u * it.tile0, y_end = min(vp.y0 + 16u * it.tile1, vp.y0 + vp.rows);
        const uint32_t dx0 = sp.x0 == jb.cx ? 0u : jb.ox + sp.x0 - jb.cx;
        const uint32_t dx1 = sp.x1 == jb.cx + jb.cw ? jb.dw : jb.ox + sp.x1 - jb.cx;
        const uint32_t dy0 = y_first == jb.cy ? 0u : jb.oy + y_first - jb.cy;
        const uint32_t dy1 = y_end == jb.cy + jb.ch ? jb.dh : jb.oy + y_end - jb.cy;
        gptr32 d32 = (gptr32)(uintptr_t)jb.dst;
        const uint32_t wcols = dx1 - dx0;
        const uint32_t top_rows = dy0 < jb.oy ? min(dy1, jb.oy) - dy0 : 0u;
        for (uint32_t k = tid; k < top_rows * wcols; k += THREADS) d32[(dy0 + k / wcols) * jb.dw + dx0 + k % wcols] = jb.fill;
        const uint32_t by0 = max(dy0, jb.oy + jb.ch);
        const uint32_t bot_rows = dy1 > by0 ? dy1 - by0 : 0u;
        for (uint32_t k = tid; k < bot_rows * wcols; k += THREADS) d32[(by0 + k / wcols) * jb.dw + dx0 + k % wcols] = jb.fill;
        const uint32_t my0 = max(dy0, jb.oy), my1 = min(dy1, jb.oy + jb.ch);
        const uint32_t mrows = my1 > my0 ? my1 - my0 : 0u;
        const uint32_t lcols = dx0 < jb.ox ? min(dx1, jb.ox) - dx0 : 0u;
        for (uint32_t k = tid; k < mrows * lcols; k += THREADS) d32[(my0 + k / lcols) * jb.dw + dx0 + k % lcols] = jb.fill;
        const uint32_t rx0 = max(dx0, jb.ox + jb.cw);
        const uint32_t rcols = dx1 > rx0 ? dx1 - rx0 : 0u;
        for (uint32_t k = tid; k < mrows * rcols; k += THREADS) d32[(my0 + k / rcols) * jb.dw + rx0 + k % rcols] = jb.fill;
    };

    // ---- source rows -> LDS ---------------------------------------------------------------------------------------
    // Load instruction u of a K-block (u = 0..7): row octet u >> 1, column half u & 1.  Lane: row q = lane & 7 of the
    // octet, 16-byte column tile t = lane >> 3 of the half; the data lands lane-linear, i.e. as eight [8 rows][16 bytes]
    // tiles of 128 bytes, which is the block ds_read_b64_tr_b8 transposes.  Odd octets swap neighbouring tiles so that
    // the two 16-lane groups of a transposed read hit different banks.
    const uint32_t lq = lane & 7u, lt = lane >> 3;
    // Addresses: an SGPR base per row octet of the K-block + a 32-bit byte offset per lane that changes only with the item (the lane's
    // row of the octet times the pitch + its 16-byte column tile; descriptors promise < 4 GiB per picture): the eight requests of a
    // K-block cost scalar adds only.  (Round 3 computed eight per-lane offsets per K-block: ~30 vector instructions and eight
    // registers in every pass.)  The picture's last K-block, whose rows past the end are clamped to the last row -- their
    // weights are zero, the bytes must merely be readable --, computes its offsets per lane.
    uint32_t voff[2][2]; // [column half][octet parity]
    auto set_voff = [&](uint32_t pitch_, uint32_t byte0_) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t hh = 0; hh < 2; ++hh)
#pragma unroll
            for (uint32_t par = 0; par < 2; ++par)
                voff[hh][par] = lq * pitch_ + min(byte0_ + wave * kMfmaWaveCols + hh * 128u + ((lt ^ par) * 16u), pitch_ - 16u); // (clamped so that 16 bytes stay inside the row)
    };
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)ring);
    // (the requests go out the moment the transposed reads have returned: between "the rows have landed" and "the next rows are
    // requested" the wave's 8 KB of LDS are not in flight, and with one K-block in flight per wave every such cycle is missing
    // bandwidth.)
    // K-block s of the picture at `src_` (rows of `pitch_` bytes, the last one `last_row_`); voff must be set for that picture
    auto request_from = [&](const void *src_, uint32_t pitch_, uint32_t last_row_, uint32_t s) __attribute__((always_inline)) {
        // (in slot order: slots 2 ro and 2 ro + 1 are the two 128-byte halves of the same eight 256-byte row pieces, and memory
        // serves them best back to back -- requesting the even slots as soon as "their" transposed reads had returned and the odd
        // ones later measured 6 % SLOWER)
        const uint8_t *kb_base = static_cast<const uint8_t *>(src_) + (size_t)(s * kMfmaKRows) * pitch_;
        if (s * kMfmaKRows + kMfmaKRows - 1u <= last_row_) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                // (inline asm: hipcc orders EVERY later LDS access behind a global_load_lds it knows about -- s_waitcnt vmcnt(0) in
                // front of the first counter or operand read -- which would park the whole horizontal stage behind the K-block just
                // requested.  The transfers are waited for by hand, wait_vm0() in front of the transposed reads.)
                const uint8_t *ob = kb_base + (size_t)((u >> 1) * 8u) * pitch_;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[u & 1u][(u >> 1) & 1u]), "s"(ob), "s"(ring_lds + u * 1024u) : "memory");
            }
        } else {
            const uint32_t rows_left = last_row_ - s * kMfmaKRows; // < 31
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t o = voff[u & 1u][(u >> 1) & 1u] + (min((u >> 1) * 8u + lq, rows_left) - lq) * pitch_;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(o), "s"(kb_base), "s"(ring_lds + u * 1024u) : "memory");
            }
        }
    };
    // what the K loop needs of the picture it walks: the item's request record (src, pitch, last row), and the next item's once its
    // first K-block has been requested
    // (single fields, not the records: only these four are alive through the K loop, and of the next item only what is named here)
    const void *rq_src, *nx_src = nullptr;
    uint32_t rq_pitch, rq_last_row, rq_job, nx_pitch = 0, nx_last_row = 0, nx_job = 0;
    bool light_next = false; // the next item continues this walk: same strip, plan and K-blocks
    auto request = [&](uint32_t s) __attribute__((always_inline)) { request_from(rq_src, rq_pitch, rq_last_row, s); };
    {   // The first item's first K-block goes out before anything else: it flies under the set-up below.
        const MfmaReq r = reqs[item];
        rq_src = r.src; rq_pitch = r.pitch; rq_last_row = r.last_row; rq_job = r.job;
        set_voff(r.pitch, r.byte0);
        request(r.kb0);
    }

    f32x4 acc[2][16];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) acc[s2][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const uint32_t traddr = (i >> 1) * 16u + (i & 1u) * 8u;
    const u32x4 *vw = nullptr; // (set per item)
    // this wave's 4 chunks x 3 tiles x { first output, operand of each weight digit, high digit first }, kept in SGPRs
    constexpr int CE = FW ? 4 : 3, NT = FW ? 3 : 2; // words per tile slot; f16 terms per vertical weight
    // (full width: two words per unit -- { first output (signed) | high digit's operand << 16 }, { middle digit's operand | low
    // digit's << 16 }: 24 scalar registers instead of 48, in a kernel that has none to spare; unpacking is scalar work)
    int32_t ctab[FW ? 24 : 12 * CE];
    uint32_t hs = 0;
    // packed: sums are (value - 128) * 2^(hs + 6) with the low plane offset by 128; full: value * 2^20 + (2^22 - 0x8080) * 2^6 - 2^27
    // (the byte planes carry 2^22 + x - 0x8080 for x = (value - 128) * 2^14, the weights of an output sum to exactly 2^hs, and every
    // wave shifts its part of the sum down to 2^-20 steps before it adds it to the tile, see the stage)
    int32_t round_add = 0;
    uint32_t out_shift = 0; // (packed arithmetic)
    uint32_t npx = 0;
    uint32_t ops_resident = 0xffffffffu; // arena offset of the operands the LDS operand area holds (a workgroup's consecutive items often share a strip)
    // Everything that hangs on the item: called once per trip of the item loop, with the item's first K-block already in flight.
    // The LDS operand area may be written here: every wave has finished the previous item's last stage (its epilogue waited for all
    // eight waves' adds of the last tile).
    auto enter_item = [&]() __attribute__((always_inline)) {
        vw = reinterpret_cast<const u32x4 *>(arena + vp.w_off);
        ops_glb = reinterpret_cast<const u32x4 *>(arena + sp.ops_off);
        hs = sp.hs;
        round_add = FW ? (int32_t)(-132112384 + (1 << (kMfmaOutFracBitsFull - 1u))) : (int32_t)((128u << hs) + (1u << (hs + kMfmaXFracBits - 1u)));
        out_shift = hs + kMfmaXFracBits;
        npx = sp.x1 - sp.x0;
        {
            const int32_t *cp = reinterpret_cast<const int32_t *>(arena + sp.ctab_off) + wave * (12u * CE);
            if constexpr (FW) {
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int32_t o = __builtin_amdgcn_readfirstlane(cp[u * 4]), i2 = __builtin_amdgcn_readfirstlane(cp[u * 4 + 1]),
                                  i1 = __builtin_amdgcn_readfirstlane(cp[u * 4 + 2]), i0 = __builtin_amdgcn_readfirstlane(cp[u * 4 + 3]);
                    // (a slot not in use has first output 2^30: any value that puts every lane past the strip's outputs does)
                    const int32_t oc = (o > 32767 || o < -32768) ? -32768 : o;
                    ctab[2 * u] = (int32_t)(((uint32_t)oc & 0xffffu) | ((uint32_t)i2 << 16));
                    ctab[2 * u + 1] = (int32_t)(((uint32_t)i1 & 0xffffu) | ((uint32_t)i0 << 16));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 12 * CE; ++k) ctab[k] = __builtin_amdgcn_readfirstlane(cp[k]);
            }
        }
        if (HLDS && (!FW || sp.lds_ops) && sp.ops_off != ops_resident) {
            for (uint32_t k = tid; k < sp.n_ops * 64u; k += THREADS) reinterpret_cast<u32x4 *>(mfma_lds + NBUF * ot_words * 4u + CNT_BYTES)[k] = ops_glb[k];
            ops_resident = sp.ops_off;
        }
    };
    // The conversion context of the item with parity `par` (seq & 1): what convert_rows needs of the item a tile belongs to.  One
    // thread writes it; a wave reads it only for a tile all eight waves have added to, i.e. after the writer's own add (release /
    // acquire on add_cnt) -- or after the barrier of a heavy transition.
    uint32_t *ctxbuf = conv_cnt + 2;
    auto write_ctx = [&](uint32_t par) __attribute__((always_inline)) {
        if (tid == 0) {
            uint32_t *cx = ctxbuf + 8u * par;
            cx[0] = (uint32_t)(uintptr_t)jb.dst; cx[1] = (uint32_t)((uintptr_t)jb.dst >> 32);
            cx[2] = jb.dw;
            cx[3] = (jb.oy - jb.cy) * jb.dw + jb.ox + (sp.x0 - jb.cx); // first pixel of the strip's row y0
            cx[4] = jb.fill;
        }
    };

    constexpr uint32_t CONV_G = (mfma_max_outputs(LAYOUT) / (uint32_t)CS + 63u) / 64u;
    // One wave's share (2 of the 16 rows) of a finished output tile: i32 sums -> bytes -> destination.  A strip row has
    // at most kMfmaMaxStripOutputs[Wide] / CS pixels = CONV_G groups of 64 lanes.  All LDS reads of the share are issued before the first of
    // them is used; lanes past the row's end repeat its last pixel (same words read, same value stored to the same address), so
    // nothing here switches lanes off; rows past the picture's end (short last tile) are skipped as a whole.  `whole_rows` (the
    // item's last tile, full-width arithmetic): the wave's two rows of the LDS tile are zeroed from end to end afterwards -- the
    // skipped rows and the dummy columns too: the next item of this workgroup may have a wider strip.
    auto convert_rows = [&](uint32_t tile, uint32_t buf, bool whole_rows, uint32_t par) __attribute__((always_inline)) {
        uint32_t *ot = otile + buf * ot_words;
        // (the item's context as per-lane values: every lane reads the same LDS words -- no scalar registers are held for it)
        const u32x4 cx = *reinterpret_cast<const u32x4 *>(ctxbuf + 8u * par);
        const uint32_t cx_fill = ctxbuf[8u * par + 4u];
        const uint64_t cx_dst = (uint64_t)cx[0] | ((uint64_t)cx[1] << 32);
        const uint32_t ngrp = (npx + 63u) >> 6; // lane groups in use (wave-uniform)
        // (both rows' sums are fetched up front where the registers allow it -- Rgb8, the flagship: 12 of them; the other
        // instantiations fetch one lane group at a time: 16-24 registers of sums at this point pushed them into scratch, whose reloads wait on vmcnt,
        // i.e. for the K-block in flight)
        constexpr bool BOTH = CS == 3 && CONV_G <= 2;
        uint32_t sums[BOTH ? 2 : 1][BOTH ? CONV_G : 1][CS];
        if (BOTH) {
#pragma unroll
            for (uint32_t rr = 0; rr < 2; ++rr)
#pragma unroll
                for (uint32_t k = 0; k < CONV_G; ++k) {
                    const uint32_t xo = min(lane + 64u * k, npx - 1u);
                    const uint32_t *o = ot + (2u * wave + rr) * np + (uint32_t)CS * xo;
#pragma unroll
                    for (int c = 0; c < CS; ++c) sums[BOTH ? rr : 0u][BOTH ? k : 0u][c] = o[c];
                }
        }
#pragma unroll
        for (uint32_t rr = 0; rr < 2; ++rr) {
            const uint32_t row = 2u * wave + rr;
            const bool live = 16u * tile + row < vp.rows;
#pragma unroll
            for (uint32_t k = 0; k < CONV_G; ++k) {
                if (k >= ngrp) continue;
                const uint32_t xo = min(lane + 64u * k, npx - 1u);
                uint32_t *o = ot + row * np + (uint32_t)CS * xo;
                if (!live) { // (a row past the picture's end: its sums are of zero weights, and the LDS tile goes on to the next tile -- of this item or the next)
#pragma unroll
                    for (int c = 0; c < CS; ++c) o[c] = 0u;
                    continue;
                }
                if (!BOTH) { // (one lane group at a time: CS registers of sums)
#pragma unroll
                    for (int c = 0; c < CS; ++c) sums[0][0][c] = o[c];
                }
                uint32_t c8[CS];
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    if constexpr (FW) {
                        // clamp-then-shift: clamp(x >> 20, 0, 255) is the pattern hipcc (ROCm 7.2) fuses into gfx950's v_ashr_pk_u8_i32,
                        // whose destination keeps stale bits above bit 15 (fl_jpegdec.hip sat17 met the same bug)
                        const int32_t x = (int32_t)sums[BOTH ? rr : 0u][BOTH ? k : 0u][c] + round_add;
                        c8[c] = (uint32_t)min(max(x, 0), (256 << kMfmaOutFracBitsFull) - 1) >> kMfmaOutFracBitsFull;
                    } else {
                        const int32_t q = ((int32_t)sums[BOTH ? rr : 0u][BOTH ? k : 0u][c] + round_add) >> out_shift;
                        c8[c] = (uint32_t)min(max(q + 128, 0), 255);
                    }
                    o[c] = 0u;
                }
                uint32_t v;
                if (LB) { // DynamicImage -> Rgba8 (to_rgba8) and imageops::overlay onto the fill colour, as in the streaming kernel
                    if (CS == 1) v = c8[0] | (c8[0] << 8) | (c8[0] << 16) | 0xff000000u;
                    else if (CS == 2) v = blend_over_fill(cx_fill, c8[0], c8[0], c8[0], c8[CS > 1 ? 1 : 0]);
                    else if (CS == 3) v = c8[0] | (c8[CS > 1 ? 1 : 0] << 8) | (c8[CS > 2 ? 2 : 0] << 16) | 0xff000000u;
                    else v = blend_over_fill(cx_fill, c8[0], c8[CS > 1 ? 1 : 0], c8[CS > 2 ? 2 : 0], c8[CS > 3 ? 3 : 0]);
                } else {
                    v = c8[0] | (c8[CS > 1 ? 1 : 0] << 8) | (c8[CS > 2 ? 2 : 0] << 16) | (c8[CS > 3 ? 3 : 0] << 24);
                }
                if (ablate & 32u) { asm volatile("" : : "v"(v)); continue; } // (experiment: everything but the store itself)
                // (a pointer read from a descriptor has no address space the compiler could know; the destination is device
                // memory by contract, see gptr32)
                const uint32_t pix = cx[3] + (vp.y0 + 16u * tile + row) * cx[2] + xo;
                if (LB) ((gptr32)(uintptr_t)cx_dst)[pix] = v;
                else {
                    gptr8 p = (gptr8)(uintptr_t)cx_dst + (size_t)pix * CS;
#pragma unroll
                    for (int c = 0; c < CS; ++c) p[c] = (uint8_t)(v >> (8 * c));
                }
            }
        }
        if (whole_rows) {
            for (uint32_t k = lane; k < 2u * np; k += 64u) ot[2u * wave * np + k] = 0u;
        }
    };

    // A K-block's weights (64 or 96 bytes per lane, served by the L2) and its meta word are fetched one K-block ahead: the vmcnt(0)
    // in front of the transposed reads covers them for free, while fetched at the top of their own K-block they would add one L2
    // round trip to every pass of the loop.  They are requested right BEHIND the pass's vertical matrix instructions, into the
    // registers those have just read (round 3 requested them in front, into a second set: 16-24 registers more at the kernel's
    // tightest point and a register-to-register copy of the set in every pass).
    // (the meta word comes through the VECTOR memory path like the weights, its index made opaque to the compiler: a scalar load
    // left in flight across the tile stage shares lgkmcnt with the stage's LDS traffic, and since scalar loads return out of
    // order every LDS wait of the stage then becomes lgkmcnt(0) -- behind all the LDS adds issued so far)
    uint32_t vzero = 0u;
    asm("" : "+v"(vzero));
    u32x4 wv[2 * NT];
    uint32_t meta_n = 0u;
    // Full-width arithmetic, one LDS output tile: a wave converts its two rows of a tile not inside the NEXT tile's stage (the packed
    // form, two tiles) but as soon as every wave has added its sums -- in the passes right after the tile's own stage, which wait
    // for rows anyway.  pend_li: the tile this wave has added to but not converted yet, numbered through the workgroup's whole
    // walk (gl_base = tiles of the items before this one: the counters run on across items).
    uint32_t pend_li = 0xffffffffu, gl_base = 0u;
    uint32_t pend_tile = 0u, pend_par = 0u; // ... its number inside its item, and that item's parity (conversion context)
    uint32_t seq = 0u;                      // items this workgroup has finished
    auto try_convert = [&](bool must, bool whole_rows) __attribute__((always_inline)) {
        const uint32_t need = kMfmaWaves * (pend_li + 1u);
        uint32_t spin = 0;
        for (;;) {
            const uint32_t ad = __hip_atomic_load(&add_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (ad >= need) break;
            if (!must) return;          // not every wave has added yet: look again in the next pass
            if (++spin >= spin_limit) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (spin >= spin_limit) wg_error = 1u; // (as in the packed form's wait: a limit of 0 -- the tests' -- reports every wait)
        if (!(ablate & 2u)) convert_rows(pend_tile, 0u, whole_rows, pend_par);
        pend_li = 0xffffffffu;
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __hip_atomic_fetch_add(&conv_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    bool heavy = true;  // how this item is entered: the first one with a full set-up
    for (;;) { // ---- the items of this workgroup ----------------------------------------------------------------------------------
    const uint32_t next_item = item + 1u;
    const bool has_next_item = FW && items_left > 1u;
    FL_STAMP(0);
    if (heavy) {
        load_item();
        enter_item();
    } else {
        // entered lightly: strip, plan, K-blocks, tile table, operands and weights stay; only the picture is another one
        jb = jobs[rq_job];
    }
    // the item's frame and its conversion context (no tile of this item is converted before every wave -- the writer too -- has added
    // to it, several passes from here)
    if (LB) paint_frame();
    write_ctx(seq & 1u);
    if (heavy) {
        meta_n = arena[vp.meta_off + it.kb0 + vzero];
#pragma unroll
        for (int k = 0; k < 2 * NT; ++k) wv[k] = vw[(it.kb0 * (2u * NT) + k) * 64u + lane];
        // the output tile is zero (first item: the clear above; later ones: every wave zeroed its rows in the epilogue and the first
        // stage waits for all of them, conv_cnt), the operands and the conversion context are in place for every wave
        __syncthreads();
    }
    FL_STAMP(1);
    // (kb_end: one pass more than the band has K-blocks when the picture's short last tile ends together with the tile before
    // it -- that pass runs on the table's all-zero K-block, index vp.nkb, whose meta word names the last tile: the matrix unit
    // adds zeros to whatever the transposed reads deliver, nothing is requested, and the loop body stays as it is)
    const uint32_t kb_end = it.kb1 + ((vp.tail && it.tile1 == vp.ntiles) ? 1u : 0u);
    for (uint32_t s = it.kb0; s < kb_end; ++s) {
        const bool have_next = s + 1u < it.kb1;
        __builtin_amdgcn_s_setprio(3); // (from here to the next request this wave's instructions go first on its SIMD: its LDS is not in flight)
        wait_vm0();
        if (s == it.kb0) FL_STAMP(2);
        const uint32_t meta = __builtin_amdgcn_readfirstlane(meta_n);
        v2i raw[16];
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const uint32_t off = (2u * g + (ct >> 3)) * 1024u + (((ct & 7) ^ (g & 1u)) * 128u) + traddr;
            raw[ct] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(ring + off));
        }
        wait_lgkm0();
        if (have_next) request(s + 1u);
        else if (has_next_item && s + 1u == kb_end) {
            // this item's last pass: the wave's 8 KB are free for good -- the NEXT item's first K-block flies under the last tile's
            // stage, the epilogue and the next item's set-up
            const MfmaReq r = reqs[next_item];
            set_voff(r.pitch, r.byte0);
            request_from(r.src, r.pitch, r.last_row, r.kb0);
            nx_src = r.src; nx_pitch = r.pitch; nx_last_row = r.last_row; nx_job = r.job;
            light_next = r.strip_off == it.strip_off && r.vplan_off == it.vplan_off && r.kb0 == it.kb0 && r.kb1 == it.kb1;
            FL_STAMP(3);
        }
        __builtin_amdgcn_s_setprio(FL_PRIO_VERT);
        if constexpr (FW) {
            if (pend_li != 0xffffffffu) try_convert(false, false);
        }
        // (the meta word says whether the K-block has weights for a second, younger tile (set 1) at all: about a third of the
        // K-blocks touch one tile only, and a matrix instruction on zeros costs the same time and nearly the same power --
        // this kernel runs at the socket's power limit, so what it does not compute is what makes it faster)
        const bool both_sets = ((meta >> 17) & 1u) != 0u;
        // (the test stays INSIDE the loop over the column tiles: hipcc keeps it as sixteen branches, i.e. sixteen small basic blocks, and
        // that is what holds the register pressure down -- hoisted out, as two straight-line copies of the loop, the scheduler converts
        // all sixteen column tiles up front and spills ~300 registers (measured, round 4).  Nothing is lost: the conversions of tile
        // ct + 1 issue while the last matrix instruction of tile ct executes.)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            // bytes -> f16 pairs.  Packed arithmetic: 0x6400 | byte = 1024 + byte.  Full width: the byte zero-extended IS the f16
            // subnormal byte * 2^-24, which gfx950's matrix unit multiplies exactly (tools/microbench/f16_denorm_probe.hip) -- no
            // bias in the f32 sums, which are then value * 2^-9 (weights are stored times 2^15)
            constexpr uint32_t fill = FW ? 0u : 0x64646464u, sel_lo = FW ? 0x0c010c00u : 0x04010400u, sel_hi = FW ? 0x0c030c02u : 0x04030402u;
            u32x4 a;
            a[0] = __builtin_amdgcn_perm(fill, (uint32_t)raw[ct][0], sel_lo);
            a[1] = __builtin_amdgcn_perm(fill, (uint32_t)raw[ct][0], sel_hi);
            a[2] = __builtin_amdgcn_perm(fill, (uint32_t)raw[ct][1], sel_lo);
            a[3] = __builtin_amdgcn_perm(fill, (uint32_t)raw[ct][1], sel_hi);
            const f16x8 av = __builtin_bit_cast(f16x8, a);
            if (ablate & 8u) { acc[0][ct][0] += (float)a[0]; acc[1][ct][1] += (float)a[1]; acc[0][ct][2] += (float)a[2]; acc[1][ct][3] += (float)a[3]; continue; }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[t]), acc[0][ct], 0, 0, 0);
            if (both_sets) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, wv[NT + t]), acc[1][ct], 0, 0, 0);
            }
        }
        // (the item's last pass: if the next item continues this walk -- same strip, plan and K-blocks -- the weights of ITS first
        // K-block, which are this item's first again)
        if (s + 1u < kb_end || light_next) {
            const uint32_t sn = s + 1u < it.kb1 ? s + 1u : (s + 1u < kb_end ? vp.nkb : it.kb0);
            meta_n = arena[vp.meta_off + sn + vzero];
#pragma unroll
            for (int k = 0; k < 2 * NT; ++k) wv[k] = vw[(sn * (2u * NT) + k) * 64u + lane];
        }
        const uint32_t ft = meta & 0xffffu;
        if (ft != 0xffffu) { // output tile ft is complete
            const bool mine = ft >= it.tile0 && ft < it.tile1 && !(ablate & 4u);
            if (ablate & 8192u) { // (experiment, with 4: the vertical pass alone -- its sums are kept alive, nothing is done with them)
#pragma unroll
                for (int ct = 0; ct < 16; ++ct) asm volatile("" : : "v"(acc[0][ct][0]), "v"(acc[0][ct][1]), "v"(acc[0][ct][2]), "v"(acc[0][ct][3]));
            }
            if (mine) {
                const uint32_t li = ft - it.tile0, buf = FW ? 0u : (li & 1u);
                uint32_t *ot = otile + buf * ot_words;
                // The stage is written for instruction-level parallelism -- two waves per SIMD run it at the same time, so nothing else
                // hides its latencies: (1) all A operands first (vector work only), (2) a chunk's B operands are requested one chunk
                // ahead, in front of the LDS adds of the chunk before (LDS executes a wave's instructions in order: behind 12 adds
                // they would return ~12 adds late), (3) both counters are read in ONE round trip at the top, (4) the previous tile's
                // rows are read before this tile's adds are issued and converted while its matrix instructions run.
                // the tile that used this buffer two tiles ago must have been converted by every wave, and every wave must have added
                // its sums of the previous tile (both long since true in practice: the slowest wave never waits)
                auto wait_for_the_tiles = [&]() __attribute__((always_inline)) {
                    const uint32_t need_conv = kMfmaWaves * (li >> 1), need_add = li ? kMfmaWaves * (((li - 1u) >> 1) + 1u) : 0u;
                    uint32_t spin = 0;
                    for (;;) {
                        const uint32_t cv = __hip_atomic_load(&conv_cnt[buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const uint32_t ad = __hip_atomic_load(&add_cnt[buf ^ 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                        if ((cv >= need_conv && ad >= need_add) || ++spin >= spin_limit) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (spin >= spin_limit) wg_error = 1u;
                };
                if constexpr (FW) {
                    // ---- full-width arithmetic ------------------------------------------------------------------------------------
                    // sums = value * 2^-9.  One fused multiply-add with the magic constant 1.5 * 2^23 - 2^21 leaves
                    // 2^22 + round((value - 128) * 2^14) in the 23 mantissa bits (|value - 128| < 256: Lanczos overshoot stays far
                    // inside); its three bytes are the planes: the top one (7 bits) as it is, the lower two offset by 128 so that they are
                    // signed (the offsets and the 2^22 come out again as constants, round_add above).
                    u32x4 p2[4], p1[4], p0[4];
                    auto make_planes = [&](int c) __attribute__((always_inline)) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const f32x4 v = acc[0][4 * c + a];
                            const uint32_t x0 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[0], 8388608.0f, 10485760.0f)), x1 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[1], 8388608.0f, 10485760.0f)),
                                           x2 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[2], 8388608.0f, 10485760.0f)), x3 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[3], 8388608.0f, 10485760.0f));
                            const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x3, x2, 0x05010400u); // (x0.b0, x1.b0, x0.b1, x1.b1)
                            p0[c][a] = __builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u;
                            p1[c][a] = __builtin_amdgcn_perm(t1, t0, 0x07060302u) ^ 0x80808080u;
                            p2[c][a] = __builtin_amdgcn_perm(x1, x0, 0x0c0c0602u) | __builtin_amdgcn_perm(x3, x2, 0x06020c0cu);       // (x0.b2, x1.b2, x2.b2, x3.b2)
                        }
                    };
                    make_planes(0);
                    if (!(variant & 16u)) { make_planes(1); make_planes(2); make_planes(3); }
                    // Units (chunk c of 64 byte columns, tile slot t of 16 outputs) as in the packed form; per unit three operands (weight
                    // digits 2, 1, 0) and all nine digit products, summed by the matrix unit into five scales: L[k] = sum over i + j = k of
                    // plane i x digit j, exact in i32 (|L| < 2^22: 64 products of at most 2^14, three of them per term at most).
                    // The units of slots 0 and 1 of every chunk run first, as one software pipeline (operands of the unit after next
                    // requested, the next unit's matrix instructions issued, in front of this unit's recombination and LDS adds); slot 2
                    // is rare -- a 64-byte chunk reaches a third tile only for ratios below ~4.5 -- and runs behind them if the strip has one.
                    u32x4 hb[2][3];
                    i32x4 L[2][5];
                    // sum of the unit = L4 2^32 + L3 2^24 + L2 2^16 + L1 2^8 + L0 in units of 2^-(14 + hs) of a pixel step; the LDS tile
                    // takes it in units of 2^-20 (a wave's part of an output rounded once, to a millionth of a step)
                    // (hs = 24 for every geometry the planner lets through: the shifts are immediates, and the rounding constant of the one
                    // right shift enters as the C operand of the first L1 product)
                    constexpr uint32_t sh = 24u + kMfmaXFracBitsFull - kMfmaOutFracBitsFull; // 18
                    constexpr uint32_t s4 = 32u - sh, s3 = 24u - sh, slo = sh - 8u;
                    constexpr int32_t rnd = 1 << (slo - 1u);
                    auto stage = [&](auto in_lds) __attribute__((always_inline)) {
                        constexpr bool OL = decltype(in_lds)::value;
                        auto load_ops = [&](int k, int u) __attribute__((always_inline)) { // operands of unit u = 3 c + t into buffer k & 1
                            uint32_t i2 = (uint32_t)ctab[2 * u] >> 16, i1 = (uint32_t)ctab[2 * u + 1] & 0xffffu, i0 = (uint32_t)ctab[2 * u + 1] >> 16;
                            asm volatile("" : "+s"(i2), "+s"(i1), "+s"(i0)); // (see the packed form)
                            hb[k & 1][0] = OL ? ops_lds[i2 * 64u + lane] : ops_glb[i2 * 64u + lane];
                            hb[k & 1][1] = OL ? ops_lds[i1 * 64u + lane] : ops_glb[i1 * 64u + lane];
                            hb[k & 1][2] = OL ? ops_lds[i0 * 64u + lane] : ops_glb[i0 * 64u + lane];
                        };
                        auto unit_mfma = [&](int k, int u) __attribute__((always_inline)) {
                            const i32x4 b2 = __builtin_bit_cast(i32x4, hb[k & 1][0]), b1 = __builtin_bit_cast(i32x4, hb[k & 1][1]), b0 = __builtin_bit_cast(i32x4, hb[k & 1][2]);
                            const i32x4 a2 = __builtin_bit_cast(i32x4, p2[u / 3]), a1 = __builtin_bit_cast(i32x4, p1[u / 3]), a0 = __builtin_bit_cast(i32x4, p0[u / 3]);
                            const i32x4 z = {0, 0, 0, 0};
                            i32x4 *l = L[k & 1]; // (five independent chains, interleaved: no instruction waits for the one before it)
                            l[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, z, 0, 0, 0);
                            l[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b1, z, 0, 0, 0);
                            l[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b0, z, 0, 0, 0);
                            l[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b0, i32x4{rnd, rnd, rnd, rnd}, 0, 0, 0);
                            if (!EIGHT) l[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, z, 0, 0, 0);
                            l[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b2, l[3], 0, 0, 0);
                            l[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, l[2], 0, 0, 0);
                            l[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b1, l[1], 0, 0, 0);
                            l[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b2, l[2], 0, 0, 0);
                        };
                        auto recombine = [&](int k, int u) __attribute__((always_inline)) {
                            // lanes outside the strip's outputs (and every lane of a slot not in use: operand 0, first output 2^30) add
                            // into dummy columns of their own, nout + i: no lane is switched off, no two lanes share an address
                            const uint32_t o = (uint32_t)((int32_t)(int16_t)(ctab[2 * u] & 0xffff) + (int32_t)i);
                            const uint32_t col = o < sp.nout ? o : sp.nout + i;
                            const i32x4 *l = L[k & 1];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                uint32_t p;
                                if (EIGHT) {
                                    // (plane 0 x digit 0 -- at most 2^20 in units of 2^-38 of a pixel step, i.e. below 2^-18 of a step --
                                    // is not computed: eight products, and the recombination is two shift-adds, a shift and a shift-add)
                                    const int32_t low = (int32_t)shl8_add((uint32_t)l[2][r], (uint32_t)l[1][r]) >> slo;
                                    p = ((uint32_t)l[4][r] << s4) + (((uint32_t)l[3][r] << s3) + (uint32_t)low);
                                } else {
                                    const int32_t low = (((l[2][r] << 8) + l[1][r]) + (l[0][r] >> 8)) >> slo;
                                    p = ((uint32_t)l[4][r] << s4) + ((uint32_t)l[3][r] << s3) + (uint32_t)low;
                                }
                                if (ablate & 128u) { asm volatile("" : : "v"(p), "v"(col)); continue; }
                                __hip_atomic_fetch_add(&ot[(4u * g + r) * np + col], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        };
                        // the k-th unit of the first pipeline is (chunk k / 2, slot k % 2), of the second (chunk k, slot 2)
#define FL_U01(k_) (3 * ((k_) / 2) + (k_) % 2)
#define FL_U2(k_) (3 * (k_) + 2)
                        if (!(ablate & 1u)) load_ops(0, FL_U01(0));
                        // one tile: this wave's rows of the previous tile, if no pass since has found them complete, and then every
                        // other wave's -- the tile must be all zeros again before the first add
                        (void)wait_for_the_tiles;
                        if (pend_li != 0xffffffffu) try_convert(true, false);
                        uint32_t spin = 0;
                        for (;;) {
                            const uint32_t cv = __hip_atomic_load(&conv_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                            if (cv >= kMfmaWaves * (gl_base + li) || ++spin >= spin_limit) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (spin >= spin_limit) wg_error = 1u;
                        if (!(ablate & 1u)) {
                            unit_mfma(0, FL_U01(0));
                            load_ops(1, FL_U01(1));
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                if (k + 1 < 8) unit_mfma(k + 1, FL_U01(k + 1));
                                if (k + 2 < 8) load_ops(k + 2, FL_U01(k + 2)); // (into the registers unit k's matrix instructions have just read)
                                if ((variant & 16u) && k % 2 == 0 && k < 6) make_planes(k / 2 + 1); // (the next chunk's planes beside this chunk's matrix instructions)
                                recombine(k, FL_U01(k));
                            }
                            if (sp.slots > 2u) {
                                load_ops(0, FL_U2(0));
                                unit_mfma(0, FL_U2(0));
                                load_ops(1, FL_U2(1));
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    if (k + 1 < 4) unit_mfma(k + 1, FL_U2(k + 1));
                                    if (k + 2 < 4) load_ops(k + 2, FL_U2(k + 2));
                                    recombine(k, FL_U2(k));
                                }
                            }
                        }
#undef FL_U01
#undef FL_U2
                    };
                    __builtin_amdgcn_s_setprio(FL_PRIO_STAGE);
                    if (HLDS && sp.lds_ops) stage(std::true_type{});
                    else stage(std::false_type{});
                    __builtin_amdgcn_s_setprio(FL_PRIO_VERT);
                } else {
                    // ---- packed arithmetic (rounds 2-3) ---------------------------------------------------------------------------
                    u32x4 ahi[4], alo[4];
    #pragma unroll
                    for (int c = 0; c < 4; ++c)
    #pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const f32x4 v = acc[0][4 * c + a]; // (the older of the two live tiles always sits in set 0, see below)
                            // acc = 256 * (1024 + value); 1.5 * 2^23 - 64 * (1024 + 128) = 12509184: the sum's low 16 bits are
                            // round((value - 128) * 64) in two's complement
                            const uint32_t x0 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[0], 0.25f, 12509184.0f)), x1 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[1], 0.25f, 12509184.0f)),
                                           x2 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[2], 0.25f, 12509184.0f)), x3 = __builtin_bit_cast(uint32_t, __builtin_fmaf(v[3], 0.25f, 12509184.0f));
                            const uint32_t t0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t1 = __builtin_amdgcn_perm(x3, x2, 0x05010400u); // (x0.b0, x1.b0, x0.b1, x1.b1)
                            alo[c][a] = __builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u; // low bytes, as signed value - 128
                            ahi[c][a] = __builtin_amdgcn_perm(t1, t0, 0x07060302u);               // high bytes (signed)
                        }
                    // units u = 3 c + t (chunk c, tile slot t): operands of unit u + 2 are requested, and the matrix instructions of
                    // unit u + 1 issued, in front of the digit sums and LDS adds of unit u
                    u32x4 h1[3], h0[3];
                    i32x4 t2[2], t1[2], t0[2];
                    auto load_ops = [&](int u) __attribute__((always_inline)) {
                        uint32_t i1 = (uint32_t)ctab[u * 3 + 1], i0 = (uint32_t)ctab[u * 3 + 2];
                        asm volatile("" : "+s"(i1), "+s"(i0)); // (keeps hipcc from hoisting the 24 operand addresses out of the row loop into 24 VGPRs)
                        h1[u % 3] = HLDS ? ops_lds[i1 * 64u + lane] : ops_glb[i1 * 64u + lane];
                        h0[u % 3] = HLDS ? ops_lds[i0 * 64u + lane] : ops_glb[i0 * 64u + lane];
                    };
                    auto unit_mfma = [&](int u) __attribute__((always_inline)) {
                        const i32x4 bh = __builtin_bit_cast(i32x4, h1[u % 3]), bl = __builtin_bit_cast(i32x4, h0[u % 3]);
                        const i32x4 ah = __builtin_bit_cast(i32x4, ahi[u / 3]), al = __builtin_bit_cast(i32x4, alo[u / 3]);
                        t2[u & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, bh, i32x4{0, 0, 0, 0}, 0, 0, 0);
                        t1[u & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah, bl, i32x4{0, 0, 0, 0}, 0, 0, 0);
                        t0[u & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al, bl, i32x4{0, 0, 0, 0}, 0, 0, 0);
                        t1[u & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al, bh, t1[u & 1], 0, 0, 0);
                    };
                    if (!(ablate & 1u)) { load_ops(0); load_ops(1); }
                    wait_for_the_tiles();
                    if (li >= 1u && !(ablate & 2u)) convert_rows(ft - 1u, buf ^ 1u, false, 0u); // this wave's two rows of the previous tile
                    if (!(ablate & 1u)) {
                        unit_mfma(0);
    #pragma unroll
                        for (int u = 0; u < 12; ++u) {
                            if (u + 2 < 12) load_ops(u + 2);
                            if (u + 1 < 12) unit_mfma(u + 1);
                            // lanes outside the strip's outputs (and every lane of a slot not in use: operand 0, first output 2^30) add
                            // into dummy columns of their own, nout + i: no lane is switched off, no two lanes share an address
                            const uint32_t o = (uint32_t)(ctab[u * 3] + (int32_t)i);
                            const uint32_t col = o < sp.nout ? o : sp.nout + i;
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const uint32_t p = shl8_add(shl8_add((uint32_t)t2[u & 1][r], (uint32_t)t1[u & 1][r]), (uint32_t)t0[u & 1][r]);
                                if (ablate & 128u) { asm volatile("" : : "v"(p), "v"(col)); continue; }
                                __hip_atomic_fetch_add(&ot[(4u * g + r) * np + col], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                if (lane == 0) { // one release for both counters
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __hip_atomic_fetch_add(&add_cnt[buf], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!FW && li >= 1u) __hip_atomic_fetch_add(&conv_cnt[buf ^ 1u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if constexpr (FW) { pend_li = gl_base + li; pend_tile = ft; pend_par = seq & 1u; }
            }
            // The younger tile becomes the older one: set 0 <- set 1, set 1 <- 0, so that the finished tile is always read from
            // compile-time registers (set 0).  Plain moves: at the power limit 128 moves are cheaper than the 32 matrix
            // instructions (0 x 0 + C) that did this before.
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                acc[0][ct] = acc[1][ct];
                acc[1][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    }
    FL_STAMP(4);
    // ---- on to the next item ------------------------------------------------------------------------------------------------------
    const bool light = light_next;
    if constexpr (FW) {
        if (!light) {
            // heavy transition, or the end: this wave's two rows of the last tile, once every wave has added its sums (a bounded
            // spin; the next item's first K-block is in flight meanwhile).  The rows are zeroed from end to end: the next item's
            // strip may be wider.  (Every wave is then past its last stage: the LDS operand area may be rewritten.)
            if (pend_li != 0xffffffffu) try_convert(true, true);
        }
    } else {
        __syncthreads(); // every wave has added its sums of the last tile
        if (!(ablate & 6u)) convert_rows(it.tile1 - 1u, (it.tile1 - 1u - it.tile0) & 1u, false, 0u);
    }
    FL_STAMP(5);
#ifdef FL_MFMA_STAMPS
    ++stamp_item;
#endif
    if (!has_next_item) break;
    gl_base += it.tile1 - it.tile0;
    item = next_item;
    --items_left;
    rq_src = nx_src; rq_pitch = nx_pitch; rq_last_row = nx_last_row; rq_job = nx_job;
    ++seq;
    heavy = !light;
    light_next = false;
    // (a band's younger accumulator set may hold rows of the tile after the band: every item starts from zero)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) acc[s2][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    } // ---- items ---------------------------------------------------------------------------------------------------------------
#ifdef FL_MFMA_STAMPS
    if (tid == 0u && blockIdx.x < 512u) { // every workgroup: start, end, XCC_ID (hardware register 20, bits 3:0), items done
        unsigned long long *d = stamps + 3 * 2 * 16 * 8 + (size_t)blockIdx.x * 4u;
        d[0] = wg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20); d[3] = seq + 1u;
    }
#endif
    if (wg_error && lane == 0) atomicOr(err_word, FLGPU_DEVERR_MFMA_WAIT);
}

} // namespace

size_t mfma_lds_bytes(uint32_t max_nout, bool ops_in_lds, bool wide)
{
    return (size_t)2 * 16 * (wide ? kMfmaOutPitchWide : kMfmaOutPitch) * 4 + CNT_BYTES + (ops_in_lds ? kMfmaLdsOperands * 1024u : 0u); // dynamic part; the rows' 64 KB are static
}

size_t mfma_lds_bytes_full(int layout)
{
    return (size_t)16 * mfma_out_pitch(layout) * 4 + CNT_BYTES + (size_t)mfma_lds_operand_capacity(layout) * 1024u; // ONE output tile
}

template <int CS, bool LB, bool HLDS, int LAYOUT, bool FW>
static hipError_t launch_mfma_t(const LaunchMfma &m, hipStream_t st)
{
    constexpr bool WIDE = LAYOUT == 1;
    const size_t lds = FW ? mfma_lds_bytes_full(LAYOUT) : mfma_lds_bytes(m.max_nout, HLDS, WIDE);
    // the attribute is per function and device (the size is a constant of the instantiation): set once per (instantiation, device)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&resample_mfma_kernel<CS, LB, HLDS, LAYOUT, FW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    // full-width arithmetic: persistent workgroups, one per CU, each walking items blockIdx.x, blockIdx.x + gridDim.x, ... (the
    // kernel's header says why); the packed arithmetic keeps one item per workgroup
    uint32_t grid = m.nitems;
#ifdef FL_VARIANT
    constexpr bool persistent = !(FL_VARIANT & 8); // (experiment: 8 = one item per workgroup, as the packed arithmetic)
#else
    constexpr bool persistent = true;
#endif
    if (FW && persistent && m.wg_lists && m.grid) grid = m.grid;
#ifdef FL_MFMA_STAMPS
    static unsigned long long *stamps = nullptr;
    static int launches = 0;
    constexpr size_t kStampWords = 3 * 2 * 16 * 8 + 512 * 4;
    if (!stamps) { (void)hipMalloc(&stamps, kStampWords * 8); (void)hipMemset(stamps, 0, kStampWords * 8); }
    resample_mfma_kernel<CS, LB, HLDS, LAYOUT, FW><<<grid, THREADS, lds, st>>>(m.jobs, m.items, m.reqs, (FW && persistent) ? m.wg_lists : nullptr, m.arena, m.nitems, 16u * mfma_out_pitch(LAYOUT), m.spin_limit, m.err_word, stamps);
    if (++launches == 50 && m.nitems > 2900) {
        unsigned long long h[kStampWords];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 6; ++w)
            for (int k = 0; k < 13; ++k) {
                const unsigned long long *d = h + (w * 16 + k) * 8;
                if (!d[0]) continue;
                fprintf(stderr, "mfma stamps wg %d wave %d item %d: set-up %.2f us, first K-block ready +%.2f, K loop to the last request %.2f, last pass + stage %.2f, epilogue %.2f; item %.2f us%s\n",
                        (w / 2) * 8, (w % 2) ? 5 : 0, k, (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01, (d[3] - d[2]) * 0.01, (d[4] - d[3]) * 0.01, (d[5] - d[4]) * 0.01, (d[5] - d[0]) * 0.01,
                        k ? "" : " (first)");
            }
        const unsigned long long *wt = h + 3 * 2 * 16 * 8;
        unsigned long long t0 = ~0ull;
        for (uint32_t b = 0; b < grid && b < 512u; ++b) t0 = std::min(t0, wt[4 * b]);
        for (uint32_t b = 0; b < grid && b < 512u; ++b)
            fprintf(stderr, "mfma wg %u: xcc %llu, start %.2f us, end %.2f us, %llu items\n", b, wt[4 * b + 2] & 15ull, (wt[4 * b] - t0) * 0.01, (wt[4 * b + 1] - t0) * 0.01, wt[4 * b + 3]);
    }
#else
    resample_mfma_kernel<CS, LB, HLDS, LAYOUT, FW><<<grid, THREADS, lds, st>>>(m.jobs, m.items, m.reqs, (FW && persistent) ? m.wg_lists : nullptr, m.arena, m.nitems, 16u * mfma_out_pitch(LAYOUT), m.spin_limit, m.err_word);
#endif
    return hipGetLastError();
}

template <int CS, bool FW>
static hipError_t launch_mfma_c(const LaunchMfma &m, hipStream_t st)
{
    if constexpr (FW) { // the operand area takes whatever LDS the layout leaves; each strip says whether it uses it
        if constexpr (CS >= 3) {
            if (m.wide) return m.letterbox ? launch_mfma_t<CS, true, true, 1, true>(m, st) : launch_mfma_t<CS, false, true, 1, true>(m, st);
        } else if (m.wide) return hipErrorInvalidValue;
        if (m.compact) return m.letterbox ? launch_mfma_t<CS, true, true, 2, true>(m, st) : launch_mfma_t<CS, false, true, 2, true>(m, st);
        return m.letterbox ? launch_mfma_t<CS, true, true, 0, true>(m, st) : launch_mfma_t<CS, false, true, 0, true>(m, st);
    } else {
        if constexpr (CS >= 3) { // (the planner keeps 1- and 2-channel sources on the narrow layout, fl_mfma_tables.cpp choose_mfma_plan)
            if (m.wide) return m.letterbox ? launch_mfma_t<CS, true, false, 1, false>(m, st) : launch_mfma_t<CS, false, false, 1, false>(m, st);
        } else if (m.wide) return hipErrorInvalidValue;
        if (m.letterbox) return m.ops_in_lds ? launch_mfma_t<CS, true, true, 0, false>(m, st) : launch_mfma_t<CS, true, false, 0, false>(m, st);
        return m.ops_in_lds ? launch_mfma_t<CS, false, true, 0, false>(m, st) : launch_mfma_t<CS, false, false, 0, false>(m, st);
    }
}

hipError_t launch_mfma(const LaunchMfma &m, hipStream_t st)
{
    if (m.nitems == 0) return hipSuccess;
    switch (m.cs) {
    case 1: return m.full ? launch_mfma_c<1, true>(m, st) : launch_mfma_c<1, false>(m, st);
    case 2: return m.full ? launch_mfma_c<2, true>(m, st) : launch_mfma_c<2, false>(m, st);
    case 3: return m.full ? launch_mfma_c<3, true>(m, st) : launch_mfma_c<3, false>(m, st);
    case 4: return m.full ? launch_mfma_c<4, true>(m, st) : launch_mfma_c<4, false>(m, st);
    default: return hipErrorInvalidValue;
    }
}

} // namespace fl
