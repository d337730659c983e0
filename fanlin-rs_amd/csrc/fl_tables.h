// fl_tables.h -- host-side construction of the read-only device tables:
// per-axis resample weights (image 0.25.6 imageops/sample.rs index + weight
// maths, evaluated in f32 exactly as the reference does), the per-row
// accumulator schedule of the streaming kernel, per-strip horizontal weight
// tiles, and libwebp's gamma tables.
#pragma once
#include <stdint.h>

#include <vector>

#include "fl_types.h"

namespace fl {

enum Filter : uint32_t { FILTER_LANCZOS3 = 0, FILTER_GAUSSIAN = 1 };

struct HostAxis {
    uint32_t in_size = 0, out_size = 0, max_taps = 0;
    std::vector<uint32_t> left, count, woff; // per output sample; woff = offset into weights
    std::vector<float> weights;              // packed, normalised
};

// sample.rs vertical_sample / horizontal_sample window + weights for every output sample of one axis.
void build_axis(uint32_t in_size, uint32_t out_size, Filter filter, float sigma, HostAxis &out);

// image::math::utils::resize_dimensions
void resize_dimensions(uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, bool fill, uint32_t &ow, uint32_t &oh);

// Row schedule of output rows [y0,y1): one RowSched per source row in [r0,r1).
// Output row o uses accumulator slot o % nacc (nacc <= NACC).
// Returns false if more than nacc output rows are alive on some source row or
// if the windows are not monotone (the streaming kernel then cannot be used).
// `block`: the kernel walks the rows in blocks of `block` rows (counted from r0) and flushes completed
// output rows only at block ends, so a slot must not be re-armed inside the block in which it completed;
// the schedule is padded with empty rows to a whole number of blocks.
bool build_row_sched(const HostAxis &v, uint32_t y0, uint32_t y1, uint32_t nacc, uint32_t block, uint32_t &r0,
                     uint32_t &r1, std::vector<RowSched> &out);

// Horizontal tile of output columns [x0,x1) for the streaming kernel.  The kernel's horizontal pass is
// input-stationary: lane t owns source pixels sx0 + 4t .. +3 of the finished f32 row (they are still in
// its registers), multiplies them into one partial sum per output column whose window they touch, and a
// second step adds each column's partial sums in ascending pixel order.  So the tables are per lane:
//   wt[j][t] = the 4 weights of lane t's pixels towards its j-th column (0 outside the window),
//   po[j][t] = LDS byte offset of that partial sum: ((x - x0) * ks + k) * 16 with k = ordinal of lane t
//              among the lanes contributing to column x; unused entries point at a dummy slot.
struct HostStrip {
    uint32_t x0 = 0, x1 = 0, sx0 = 0, sx1 = 0;
    uint32_t jmax = 0, kmax = 0, ks = 0;
    std::vector<float> wt;     // [jmax][lanes][4]
    std::vector<uint32_t> po;  // [jmax][lanes]
};
void build_strip(const HostAxis &h, uint32_t x0, uint32_t x1, uint32_t lanes, uint32_t px_per_lane, HostStrip &out);

// Everything the blur kernel needs for one (width, height, sigma), as one arena block so that a workgroup
// reaches its tables with two dependent loads instead of a chain through the generic axis tables:
//   header  : nt, nb, tw_full, htaps, rv (rows of a dense vertical table), tiles_off, bands_off, vdense_off, htiles_off (relative words)
//   tiles   : per column tile  { first source column, number of source columns }
//   bands   : per row band     { first source row, number of source rows }
//   vdense  : per band, rv x ty floats: weight of source row (top + r) in output row (y0 + o), 0 outside the window
//   htiles  : per tile, hleft[tw_full] then weights tap-major [htaps][tw_full]
void build_blur_plan(const HostAxis &v, const HostAxis &h, uint32_t nt, uint32_t ty, std::vector<uint32_t> &out);

// libwebp picture_csp_enc.c InitGammaTables: kGammaToLinearTab[256] then kLinearToGammaTab[33], as int32.
void build_webp_gamma(std::vector<uint32_t> &out);

// JPEG: everything in front of the entropy-coded data (SOI, APP0, SOF0, DQT x2, DHT x4, SOS = 623 bytes, padded to
// 624) followed by the two quality-scaled quantisation tables in natural order (64 + 64 bytes) and ceil(2^32 / 2q) of each.
// image 0.25.6 codecs/jpeg/encoder.rs: JpegEncoder::new_with_quality (table scaling), encode_image (segment order),
// build_jfif_header / build_frame_header / build_quantization_segment / build_huffman_segment / build_scan_header.
void build_jpeg_tables(uint32_t width, uint32_t height, uint32_t quality, std::vector<uint32_t> &out);

} // namespace fl
