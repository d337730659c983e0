// fl_mfma.h -- the streaming matrix-pipe resample kernel (fl_mfma.hip): descriptors, table layout and the host-side builders.
//
// Fused vertical + horizontal Lanczos3 down-scale (ratio >= 2) of an 8-bit picture (reference: image 0.25.6 imageops/sample.rs
// vertical_sample + horizontal_sample, called by resize_exact from src/handler.rs:229-255), arranged for gfx950's MFMA units.
//
// THE SHIPPED KERNEL is the full-width form (template parameter FW, MFMA_ARITH_FULL, the default since round 4):
//
//   work item        one picture x one strip of <= 2048 source bytes per row (strips start on 128-byte lines where the picture's
//                    rows do) x a band of 16-row output tiles.  Since round 5 the launch is PERSISTENT: min(items, CUs) workgroups
//                    of 8 waves, each walking its own item list (LaunchMfma::wg_lists, fl_batch.cpp assign_items).  A workgroup
//                    keeps one strip where the launch allows: an item whose strip, vertical plan and K-blocks equal its
//                    predecessor's ("light" transition) reuses the operands in LDS and the packed tile table in SGPRs, and its
//                    first K-block is requested during the predecessor's last pass (MfmaReq).
//   vertical pass    the source rows stream once, in K-blocks of 32 rows, from HBM straight into LDS (global_load_lds_dwordx4,
//                    a wave-private 8 KB image per K-block, no VGPRs, no barrier); ds_read_b64_tr_b8 hands every lane the 8 rows
//                    of one byte column; the bytes enter v_mfma_f32_16x16x32_f16 as exact f16 subnormals (b * 2^-24) against the
//                    reference's f32 weight, split into three f16 terms of 2^15 w.  At most two 16-row output tiles are alive
//                    per K-block; their f32 sums (= value * 2^-9) stay in registers.
//   horizontal pass  a finished tile is rounded once to 2^-14 of a pixel step, 2^22 + round((value - 128) * 2^14), and split
//                    into three byte planes -- already the A operand of v_mfma_i32_16x16x64_i8.  The B operands hold the
//                    horizontal weights round(w * 2^hs) (sums forced to exactly 2^hs) as three balanced signed byte digits; eight
//                    of the nine plane x digit products (all but lowest x lowest, < 2^-18 of a step) are summed exactly in i32 and
//                    recombined to 2^-20 of a step.  Waves add their partial sums into ONE shared [16][outputs] LDS tile
//                    (integer adds commute: the result does not depend on the order; add_cnt / conv_cnt hand the tile over
//                    without a workgroup barrier); one rounding to the byte, clamp, store (letterbox placement, Rgba8 expansion).
//   LDS              rows 64 KB | one output tile (layout-dependent pitch, below) | counters + two conversion contexts | the
//                    strip's B operands (1 KB each) when they fit, else they are read through the L2 (MfmaStrip::lds_ops).
//
// Error budget: no operand is narrower than the reference's f32; the two roundings (2^-14 between the passes, 2^-20 before the
// byte) leave every output byte within 1 of the reference's, and ~3e-5 of them off by one (tests/test_mfma_resample.py measures
// the rate; tests/test_mfma_tables.py bounds the tables).
//
// THE PACKED FORM (rounds 2-3; FW = false, MFMA_ARITH_PACKED, selected by flgpu_config / the mfma_arith switch) is kept as the
// measured comparison: bytes as f16 1024 + b, weights as two f16 terms, the intermediate as (value - 128) * 64 in i16 (two byte
// planes), horizontal weights of 14..17 bits as two digits, three digit products; one item per workgroup, two output tiles in LDS.
// ~0.1 % of its bytes are off by one.  Constants marked "packed" below belong to it.
#pragma once
#include <stdint.h>

#include <vector>

#include "fl_tables.h"
#include "fl_types.h"

namespace fl {

constexpr uint32_t kMfmaWaves = 8;          // waves per workgroup
constexpr uint32_t kMfmaKRows = 32;         // source rows per K-block
constexpr uint32_t kMfmaWaveCols = 256;     // byte columns per wave
constexpr uint32_t kMfmaStripBytes = kMfmaWaves * kMfmaWaveCols;
constexpr uint32_t kMfmaMaxStripOutputs = 408; // outputs (pixels x channels) per strip: 136 Rgb8 pixels (bounds the LDS output tiles: 2 x 16 x 428 words = 53.5 KB)
// Words per row of an LDS output tile: 408 outputs + 16 dummy columns (one per lane of a 16-lane group, for lanes whose output lies
// outside the strip), rounded up to 4 (mod 8): an LDS add goes to row 4g + r, column base + i of lane (g, i), and with
// 4 x pitch = 16 (mod 32) the two 16-lane groups that share an LDS cycle fall on disjoint halves of the 32 banks
// (pitch 409 put them 4 banks apart: 12 of 16 banks hit twice).
constexpr uint32_t kMfmaOutPitch = 428;
static_assert(kMfmaOutPitch >= kMfmaMaxStripOutputs + 16 && (4 * kMfmaOutPitch) % 32 == 16, "LDS output tile pitch");
// The wide layout: no horizontal operands in LDS (they come from the L2), their 40 KB go to the output tiles instead:
// 2 x 16 x 748 words = 93.5 KB, 242 Rgb8 pixels per strip.
constexpr uint32_t kMfmaMaxStripOutputsWide = 728;
constexpr uint32_t kMfmaOutPitchWide = 748;
static_assert(kMfmaOutPitchWide >= kMfmaMaxStripOutputsWide + 16 && (4 * kMfmaOutPitchWide) % 32 == 16, "LDS output tile pitch (wide)");
// The three LDS layouts of the kernel (template parameter LAYOUT): words per output-tile row / outputs per strip / operands
// (1 KB each) that fit next to the two output tiles and the rows' 64 KB.
constexpr uint32_t kMfmaMaxStripOutputsCompact = 312; // (104 Rgb8 pixels: what three strips of 1080p -> 300 columns need when every strip starts on a 128-byte line)
constexpr uint32_t kMfmaOutPitchCompact = 332;
static_assert(kMfmaOutPitchCompact >= kMfmaMaxStripOutputsCompact + 16 && (4 * kMfmaOutPitchCompact) % 32 == 16, "LDS output tile pitch (compact)");
constexpr uint32_t mfma_out_pitch(int layout) { return layout == 1 ? kMfmaOutPitchWide : layout == 2 ? kMfmaOutPitchCompact : kMfmaOutPitch; }
constexpr uint32_t mfma_max_outputs(int layout) { return layout == 1 ? kMfmaMaxStripOutputsWide : layout == 2 ? kMfmaMaxStripOutputsCompact : kMfmaMaxStripOutputs; }
// full-width arithmetic: operands the LDS operand area of a layout holds (160 KB - the rows' 64 KB - ONE output tile - counters)
constexpr uint32_t kMfmaCntBytes = 16u + 64u; // LDS counters add_cnt[2], conv_cnt[2] + two conversion contexts of 8 words (fl_mfma.hip)
constexpr uint32_t mfma_lds_operand_capacity(int layout) { return (160u * 1024u - 64u * 1024u - 16u * mfma_out_pitch(layout) * 4u - kMfmaCntBytes) / 1024u; } // 69 / 49 (wide) / 75 (compact)
constexpr uint32_t kMfmaDefaultSpinLimit = 1u << 22; // polls of an LDS counter before a wave gives up and reports FLGPU_DEVERR_MFMA_WAIT
constexpr uint32_t FLGPU_DEVERR_MFMA_WAIT = 1u;      // bit of the batch's device error word
constexpr uint32_t kMfmaVScaleLog2 = 8;     // packed arithmetic: vertical weights are stored times 2^8 (keeps the low f16 term normal)
constexpr uint32_t kMfmaXFracBits = 6;      // packed arithmetic: intermediate rows are (value - 128) * 64 as i16
constexpr uint32_t kMfmaLdsOperands = 40;   // horizontal operands (1 KB each) kept in LDS when a strip has no more distinct ones
// Full-width arithmetic (the default since round 4; fl_mfma.hip, template parameter FW):
constexpr uint32_t kMfmaVScaleLog2Full = 15; // vertical weights times 2^15 as three f16 terms; the bytes enter as f16 subnormals (b * 2^-24): sums = value * 2^-9
constexpr uint32_t kMfmaXFracBitsFull = 14;  // intermediate rows: round((value - 128) * 2^14), 23 bits, as three byte planes
constexpr uint32_t kMfmaOutFracBitsFull = 20; // the horizontal sums reach the LDS output tile in units of 2^-20 of a pixel step

// The two arithmetics of the kernel (which one a context uses: flgpu_config / FLGPU_MFMA_ARITH, fl_batch.cpp).
//   PACKED (rounds 2-3): u8 as f16 (1024 + b) x weights as two f16 terms -> f32; intermediate 1/64 steps (i16) x 14..17-bit
//                        fixed-point weights -> exact i32.  Within 1 LSB of the reference on every byte, ~0.1 % of them off by one.
//   FULL   (round 4):    no operand narrower than the reference's f32: u8 (exact, f16 subnormal) x the f32 weight itself (three f16
//                        terms) -> f32 sums in the matrix unit; intermediate rounded to 2^-14 of a pixel step (23 bits + sign,
//                        three byte planes) x weights rounded to 2^-24 (three byte digits, sums forced to exactly 1) -> the digit
//                        products, exact in i32, recombined to 2^-20 of a pixel step.  Round 5: EIGHT of the nine products -- the
//                        lowest plane x the lowest digit is at most 2^20 in units of 2^-38 of a pixel step per 64 columns, i.e. less
//                        than 2^-18 of a step (below half an f32 ulp of any value above 64), and is not computed; every operand
//                        still enters with all of its 24 bits.
enum MfmaArith : uint32_t { MFMA_ARITH_PACKED = 0, MFMA_ARITH_FULL = 1 };

// One workgroup.
struct alignas(16) MfmaItem {
    uint32_t job;
    uint32_t vplan_off;    // arena word offset of MfmaVPlan
    uint32_t strip_off;    // arena word offset of MfmaStrip
    uint32_t tile0, tile1; // output tiles [tile0, tile1) of this band
    uint32_t kb0, kb1;     // K-blocks walked [kb0, kb1)
    uint32_t flags;        // ITEM_* letterbox duties
};

// What the request of an item's FIRST K-block needs, one record per item in item order (the persistent workgroups of the full-width
// kernel issue that request in the last pass of the item before: one scalar load there, nothing of the next item in registers earlier).
struct alignas(16) MfmaReq {
    const void *src;       // the job's source picture
    uint32_t pitch;        // bytes per source row
    uint32_t last_row;     // source rows - 1
    uint32_t byte0;        // MfmaStrip::byte0 of the item's strip
    uint32_t kb0, kb1;     // the item's K-blocks
    uint32_t job;          // MfmaItem::job
    uint32_t strip_off, vplan_off; // MfmaItem's: an item whose strip, plan and K-blocks equal its predecessor's needs no new set-up
    uint32_t pad[2];
};
static_assert(sizeof(MfmaReq) == 48, "MfmaReq layout");

// Vertical plan of one (axis, kept rows) pair.  All offsets are arena word offsets.
//   kb_meta[nkb + 1] per K-block: bits 0-15 = tile that is complete after it (0xffff: none), bit 16/17 = set 0/1 has weights in it;
//                  entry nkb = the all-zero K-block the kernel appends when `tail` is set (it completes the last tile)
//   kb_w[nkb][2 sets][nterms][64 lanes] x 16 bytes: B operand of v_mfma_f32_16x16x32_f16: lane 16g + n holds the
//                  weights of source rows 32 s + 8 g .. + 7 towards output row 16 tile + n as 8 f16 (nterms = 2: packed, 3: full arithmetic)
struct MfmaVPlan {
    uint32_t ntiles, nkb;
    uint32_t y0, rows;     // first kept output row (resized coordinates) and how many
    uint32_t meta_off, w_off;
    uint32_t tail;         // the last tile ends in the same K-block as the one before it: one more pass, on the all-zero K-block, completes it
    uint32_t nterms;       // f16 terms per weight (2 or 3)
};

// Horizontal plan of one strip.
//   ctab[8 waves][4 chunks][3 tile slots][1 + digits]: { first output index of the tile (may be negative or past nout: lanes outside go to the
//                  dummy column), operand index of each weight digit, high digit first (2 digits: packed, 3: full arithmetic) }; unused tile slots: operand 0 (all zeros), first output 2^30
//   ops[n_ops][64 lanes] x 16 bytes: B operands of v_mfma_i32_16x16x64_i8, deduplicated
struct MfmaStrip {
    uint32_t x0, x1;       // output columns [x0, x1) in resized coordinates
    uint32_t byte0;        // first source byte of the strip inside a row (multiple of 16)
    uint32_t nout;         // (x1 - x0) * channels
    uint32_t hs;           // horizontal weights are scaled by 2^hs
    uint32_t n_ops;
    uint32_t ctab_off, ops_off;
    uint32_t lds_ops;      // full-width arithmetic: the strip's operands fit the launch's LDS operand area and are read from there (else from the L2)
    uint32_t slots;        // tile slots a 64-byte chunk of this strip uses at most (2 or 3)
};

struct HostMfmaPlan {
    bool ok = false;
    bool wide = false;                   // strips of up to kMfmaMaxStripOutputsWide outputs (choose_mfma_plan)
    MfmaArith arith = MFMA_ARITH_FULL;
    std::vector<uint32_t> vmeta, vw;     // MfmaVPlan tables
    uint32_t ntiles = 0, nkb = 0, y0 = 0, rows = 0, tail = 0;
    struct Tile { uint32_t kb_first, kb_last; };
    std::vector<Tile> tiles;
    struct Strip { MfmaStrip hdr{}; std::vector<int32_t> ctab; std::vector<uint32_t> ops; };
    std::vector<Strip> strips;
};

// Builds the tables for output rows [cy, cy+ch) x columns [cx, cx+cw) of a picture with cs interleaved 8-bit channels.
// ok = false when the geometry does not fit the kernel (more than two tiles alive in a K-block, horizontal windows
// that touch more than three 16-output tiles per 64-byte chunk, weights too large for the digit planes).
void build_mfma_plan(const HostAxis &v, const HostAxis &h, uint32_t cs, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, HostMfmaPlan &out,
                     uint32_t max_outputs = kMfmaMaxStripOutputs, MfmaArith arith = MFMA_ARITH_FULL);
// The plan the library uses for the geometry: the narrow layout, or the wide one where that saves strips.
void choose_mfma_plan(const HostAxis &v, const HostAxis &h, uint32_t cs, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, HostMfmaPlan &out,
                      MfmaArith arith = MFMA_ARITH_FULL);

struct LaunchMfma {
    const Job *jobs;
    const MfmaItem *items;
    const MfmaReq *reqs;   // one per item (full-width arithmetic)
    const uint32_t *arena;
    uint32_t nitems;
    uint32_t grid;         // full-width arithmetic: workgroups of the persistent launch (min(nitems, CUs))
    const uint32_t *wg_lists; // ... and, per workgroup, {first item, items} (fl_batch.cpp assign_items); packed arithmetic: null, one item per workgroup
    uint32_t cs;           // channels of the source (1..4)
    uint32_t letterbox;
    uint32_t ops_in_lds;   // every strip of the launch has <= kMfmaLdsOperands distinct operands
    uint32_t wide;         // the launch's plans use the wide layout (never together with ops_in_lds)
    uint32_t compact;      // ... the compact layout (full-width arithmetic only)
    uint32_t full;         // the launch's plans were built for the full-width arithmetic (MFMA_ARITH_FULL)
    uint32_t max_nout;
    uint32_t spin_limit;   // bound of the kernel's LDS counter waits (kMfmaDefaultSpinLimit; tests force 0 = every wait expires)
    uint32_t *err_word;    // device word of the batch: the kernel ORs FLGPU_DEVERR_MFMA_WAIT into it when a wait expired
};
size_t mfma_lds_bytes(uint32_t max_nout, bool ops_in_lds, bool wide = false); // packed arithmetic
size_t mfma_lds_bytes_full(int layout);                                       // full-width arithmetic: the operand area takes what is left
hipError_t launch_mfma(const LaunchMfma &m, hipStream_t st);

} // namespace fl
