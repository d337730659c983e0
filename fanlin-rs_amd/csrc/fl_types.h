// fl_types.h -- descriptors shared between the host runtime and the HIP kernels.
// All device-visible tables live in one "arena" of 32-bit words; descriptors
// refer to them by word offset so a whole arena can be copied (or broadcast
// to another GPU) as one blob.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fl {

// Pointwise operation applied to the SOURCE-resolution image before
// resampling (reference src/handler.rs:224-228: grayscale wins over inverse).
enum PreOp : uint32_t { PRE_NONE = 0, PRE_GRAY = 1, PRE_INVERT = 2 };

// Channels after the pre-op: grayscale turns Rgb8 into Luma8 and Rgba8 into LumaA8.
__host__ __device__ constexpr uint32_t mid_channels(uint32_t cs, uint32_t pre)
{
    return pre == PRE_GRAY ? (cs == 3 ? 1u : cs == 4 ? 2u : cs) : cs;
}

// One image of a batch.  Geometry follows flgpu_plan.
struct alignas(16) Job {
    const uint8_t *src;     // source pixels (device)
    uint8_t *dst;           // stage-1 destination: final pixels, or the blur input
    uint32_t src_bytes;     // sw*sh*cs, for hardware range checking
    uint32_t sw, sh;        // source size
    uint32_t rw, rh;        // resize_exact target (== sw,sh when nothing is resampled)
    uint32_t cx, cy;        // centre-crop origin inside the resized image
    uint32_t cw, ch;        // size kept after the crop (== rw,rh without crop)
    uint32_t dw, dh;        // destination size
    uint32_t ox, oy;        // where the kept image lands inside the destination
    uint32_t fill;          // r | g<<8 | b<<16 | 255<<24
    uint32_t vtab, htab;    // arena word offsets of the AxisTable headers (vertical, horizontal)
    uint32_t mid_off;       // generic path: float offset of this job's f32 intermediate
    uint32_t pad0;          // blur jobs: arena word offset of the horizontal weight tiles
    uint32_t pad1;
};

// Output-major weight table of one axis (image 0.25.6 sample.rs index maths).
// Header followed by left[out], count[out], woff[out] and the packed weights.
struct AxisTable {
    uint32_t in_size, out_size;
    uint32_t max_taps, total_taps;
    uint32_t left_off, count_off, woff_off, weights_off; // word offsets from the arena base
};

// Streaming kernel: max output rows alive for one source row.
constexpr int NACC = 8;
// Pixels of one source row owned by one lane.
constexpr int PXL = 4;
// Rows of the schedule staged in LDS at a time (double buffered); schedules are padded to whole chunks.
constexpr int SCHED_CHUNK = 32;

// Per source row of a band: the weight each live accumulator slot applies to
// this row, which slots are live, and which complete (emit) after it.
struct RowSched {
    float w[NACC];      // weight of this source row in each slot; 0 for slots that are not alive
    uint32_t live;      // bit s: slot s accumulates this row (diagnostics; the kernel relies on w == 0)
    uint32_t emit;      // first row of a block: slots that complete inside the block (flushed at its end)
    uint32_t first_out; // first row of a block: output row index of the first slot that completes in it
    uint32_t pad;
};

// One workgroup of the streaming resample kernel: a band of output rows x a strip of output columns.
struct alignas(16) StreamItem {
    uint32_t job;
    uint32_t y0, y1;     // resized-image rows produced [y0,y1)
    uint32_t x0, x1;     // resized-image columns produced [x0,x1)
    uint32_t r0, r1;     // source rows walked [r0,r1)
    uint32_t sx0;        // source pixel column of lane 0
    uint32_t sched_off;  // arena word offset: RowSched[r1-r0]
    uint32_t wt_off;     // arena word offset: float4 WT[jmax][256] - lane t's 4 pixel weights towards its j-th output column
    uint32_t po_off;     // arena word offset: uint32 PO[jmax][256] - LDS byte offset of that partial sum (dummy slot if unused)
    uint32_t jmax;       // output columns one lane contributes to (max over lanes)
    uint32_t kmax;       // lanes contributing to one output column (max over columns)
    uint32_t ks;         // odd slot stride per output column in the partial-sum buffer (>= kmax)
    uint32_t flags;      // ITEM_* letterbox duties
    uint32_t pad0;
};
// Header of the blur kernel's table block (fl_tables.h build_blur_plan); offsets are words relative to the header.
// htiles: per tile tw_full x {hleft, row id}; hrows: the distinct horizontal weight vectors, tap-major [htaps][nrows_h]
// (all interior columns of a blurred picture share one vector; only the columns within 2 sigma of a border differ)
struct BlurPlanHeader { uint32_t nt, nb, tw_full, htaps, rv, tiles_off, bands_off, vdense_off, htiles_off, hrows_off, nrows_h, pad; };

enum : uint32_t { ITEM_FIRST_BAND = 1, ITEM_LAST_BAND = 2, ITEM_FIRST_STRIP = 4, ITEM_LAST_STRIP = 8 };

} // namespace fl
