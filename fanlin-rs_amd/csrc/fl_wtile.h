// fl_wtile.h -- the window-tile matrix-pipe kernel (round 4): separable resampling through an LDS window of the source, for what
// the streaming matrix-pipe kernel (fl_mfma.h) cannot take because more than two 16-row output tiles are alive per K-block --
// mild down-scales (ratio < ~3), up-scales (thumbnails), and the Gaussian blur (ratio 1, 41..81 taps).  Same arithmetic as that
// kernel's full-width form, operand for operand (reference: image 0.25.6 imageops/sample.rs vertical_sample + horizontal_sample,
// called by resize_exact and blur, src/handler.rs:229-255):
//   vertical    bytes as exact f16 subnormals (b * 2^-24) x the f32 weight as three f16 terms of 2^15 w -> f32 sums
//               (v_mfma_f32_16x16x32_f16), = value * 2^-9
//   between     one rounding to 2^-14 of a pixel step: 2^22 + round((value - 128) * 2^14) as three byte planes
//   horizontal  weights round(w * 2^hs) with sums forced to exactly 2^hs, three balanced signed byte digits; all nine plane x
//               digit products exact in i32 (v_mfma_i32_16x16x64_i8), recombined to 2^-20 of a step, rounded once to the byte.
//               hs = 24 while every weight is below 0.498 (every down-scale and every blur of sigma >= 1), 23 below 0.996, else 22:
//               a three-digit balanced number ends at 127.5 * 2^16.
//
// Work split: one workgroup (8 waves) = one picture x one STRIP of 16-output-byte column tiles (N-tiles) x a band of 16-row
// output tiles (M-tiles), walked top to bottom.  The source rows an M-tile needs live in an LDS ring (row r in slot r mod
// ring_rows, row-major, pitch sp bytes); each step loads only the rows the previous M-tile did not have.  Per M-tile:
//   1. the rows and the M-tile's vertical operands requested during the previous step are written to LDS      (barrier)
//   2. vertical: every wave takes 16-byte column tiles of the strip's window; ds_read_b64_tr_b8 hands a lane 8 rows of its
//      byte column, three f16 MFMAs per 32 rows; the f32 sums become byte planes in LDS, row-major                  (barrier)
//   3. the next step's rows and operands are requested (they fly during 4)
//   4. horizontal: every wave takes N-tiles; A = 16 rows x 64 plane bytes straight from LDS (ds_read_b128), B = the N-tile's
//      weight digits -- kept in REGISTERS for the whole walk (a wave owns the same N-tiles in every M-tile) -- nine MFMAs per
//      64 source bytes into five accumulators, one recombination per N-tile, bytes into an LDS output tile        (barrier)
//   5. the output tile goes to the destination (letterbox placement / Rgba8 expansion as in the other kernels).
#pragma once
#include <stdint.h>

#include <vector>

#include "fl_tables.h"
#include "fl_types.h"

namespace fl {

constexpr uint32_t kWtWaves = 8;
constexpr uint32_t kWtThreads = kWtWaves * 64;
constexpr uint32_t kWtMaxKV = 4;          // K-steps of 32 source rows per M-tile (<= 128 rows: sigma 20 needs 96 + alignment)
constexpr uint32_t kWtMaxKH = 8;          // K-steps of 64 source bytes per N-tile (sigma 20 on Rgba8: 16 + 2 * 160 bytes)
constexpr uint32_t kWtOperandRegs = 6;    // (N-tile, K-step) operand triples a wave keeps in registers (3 x 4 VGPRs each: 72 of its 256)
constexpr uint32_t kWtLdsBudget = 150 * 1024;
constexpr uint32_t kWtPrefetch = 4;       // 16-byte pieces a thread keeps in flight for the next step's rows

// LDS layout of a workgroup: ring | three planes of 16 rows | the M-tile's vertical operands | output tile | the strip's N-tile records
// (= 16 mod 32 bytes, i.e. an odd multiple of 4 dwords: lane (row i, g) of the horizontal pass writes dword i * pitch / 4 + 4 tile + g, and the
// sixteen rows x four dwords of a wave's write land on the 64 banks once each)
__host__ __device__ constexpr uint32_t wt_out_pitch(uint32_t tn) { return 16u * tn + ((tn & 1u) ? 32u : 16u); }
__host__ __device__ constexpr uint32_t wt_lds_bytes(uint32_t ring_rows, uint32_t sp, uint32_t nkv_max, uint32_t tn)
{
    return ring_rows * sp + 48u * sp + nkv_max * 3072u + 16u * wt_out_pitch(tn) + 16u * tn;
}

// Header of a plan block in the arena; all offsets are words relative to the header.
struct WtHeader {
    uint32_t n_mt, n_nt, n_strips, cs;
    uint32_t rows, nout;            // kept output rows, kept output bytes per row (columns x channels)
    uint32_t src_rows, src_rowbytes;
    uint32_t mt_off, nt_off, strip_off, hs;
    uint32_t ring_rows, ring_magic; // rows of the LDS ring (multiple of 8); ceil(2^32 / ring_rows)
    uint32_t nkv_max, nkh_max;
};
struct WtMTile { uint32_t kr0, nk, ops, pad; };   // first source row of the K window (multiple of 8), K-steps, operands [nk][3 terms][64 lanes][4 words]
struct WtNTile { uint32_t kc0, nk, ops, pad; };   // first source byte of the K window (multiple of 16), K-steps, operands [nk][3 digits][64 lanes][4 words]
struct WtStrip {
    uint32_t n0, n1;       // N-tiles [n0, n1)
    uint32_t col0;         // first source byte of the strip's LDS window (multiple of 16)
    uint32_t sp;           // LDS row pitch in bytes: a multiple of 16 with sp / 16 odd (transposed and 16-byte reads of 8 / 16 rows then touch every bank once)
    uint32_t common_ops;   // operands most N-tiles of the strip share (a blur's interior columns), or 0xffffffff
    uint32_t lds_bytes;
    uint32_t common_nk;    // K-steps of that block
    uint32_t pad;
};

// One workgroup.
struct alignas(16) WtItem {
    uint32_t job;
    uint32_t plan_off;     // arena word offset of the WtHeader
    uint32_t strip;
    uint32_t mt0, mt1;     // M-tiles [mt0, mt1)
    uint32_t pad[3];
};

struct HostWtPlan {
    bool ok = false;
    uint32_t nslot = 0, nkmax = 0;     // the kernel instantiation the plan needs: nslot x nkmax = kWtOperandRegs
    uint32_t n_mt = 0, n_strips = 0;
    uint32_t lds_bytes = 0;
    std::vector<uint32_t> blk;         // header, tables, operands
};

// Tables for output rows [cy, cy + ch) x columns [cx, cx + cw) of a picture with cs interleaved 8-bit channels.
// ok = false when the geometry does not fit (windows beyond kWtMaxKV / kWtMaxKH, weights of 2 or more).
void build_wtile_plan(const HostAxis &v, const HostAxis &h, uint32_t cs, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, HostWtPlan &out);

struct LaunchWtile {
    const Job *jobs;
    const WtItem *items;
    const uint32_t *arena;
    uint32_t nitems;
    uint32_t nslot, nkmax;
    uint32_t letterbox;
    uint32_t lds_bytes;
    uint32_t invert;       // PRE_INVERT: the colour channels enter as 255 - c
    uint32_t half_waves;   // 1 = four waves per workgroup where the plan allows (fl_wtile.hip launch_wtile)
    uint32_t framed;       // one-channel blur of a letterboxed grey picture: the source is the UNFRAMED Luma8 picture (Job rw x rh at (cx, cy) of the sw x sh
                           // image the plan was built for); everything around it reads as Job::fill, and with `letterbox` the store expands to Rgba8
};
hipError_t launch_wtile(const LaunchWtile &m, hipStream_t st);

} // namespace fl
