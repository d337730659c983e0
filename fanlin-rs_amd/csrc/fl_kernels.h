// fl_kernels.h -- launch interface between the host runtime and fl_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fl_types.h"

namespace fl {

// One image of an encoder-front-end launch.
struct alignas(16) FrontendJob {
    const uint8_t *src;   // interleaved pixels, w x h x c
    uint8_t *dst;         // planes
    uint32_t *status;     // optional: bit 0 set if any alpha != 255 (WEBP420)
    uint32_t w, h, c;
    uint32_t plane_w, plane_h;
    uint32_t chroma_w, chroma_h;
    uint32_t pad;
};

// One image of a JPEG-encode launch (fl_jpeg.hip).
#define FL_JPEG_RESULT_OVERFLOW 4u
constexpr uint32_t kAcWordsPerUnit = 52; // 63 coefficients x (16-bit code + 10 value bits) = 1638 bits
struct alignas(16) JpegJob {
    const uint8_t *src;   // interleaved pixels, w x h x c
    uint8_t *dst;         // the JFIF stream
    uint32_t *meta;       // scratch: [unit] quantised DC (low 16 bits, i16) | AC code bits << 16; unit = block * 3 + component
    uint32_t *unit_off;   // scratch: units + 1 bit offsets
    uint32_t *acbits;     // scratch: [unit][kAcWordsPerUnit] the block's AC code from bit 0, most significant bit first
    uint32_t *result;     // [0] |= flags (FL_JPEG_RESULT_OVERFLOW), [1] = stream bytes (0 if it did not fit dst_cap)
    uint32_t w, h, c;
    uint32_t bx, by;      // blocks per row / column
    uint32_t tab_off;     // arena word offset of the header + quantisation tables block (fl_jpeg_tables.h)
    uint32_t pad0;
    uint32_t dst_cap;     // bytes available at dst
};

// A group of jobs sharing source channels / pre-op / letterbox flag.
struct LaunchGeneric {
    const Job *jobs;        // device array
    const uint32_t *arena;  // device table arena
    float *mid;             // device f32 intermediate (generic resample only)
    uint32_t job_base;      // first job of the group inside `jobs`
    uint32_t njobs;
    uint32_t cs, pre;       // source channels, PreOp
    uint32_t letterbox;     // destination is Rgba8 over the fill colour
    uint32_t grouped;       // horizontal pass sums by aligned 4-pixel blocks (Lanczos3 resize) instead of tap by tap (blur)
    uint32_t max_sw, max_rh;  // vertical pass grid
    uint32_t max_cw, max_ch;  // horizontal pass grid
    uint32_t max_dw, max_dh;  // placement grid
    uint32_t nearest;         // place kernel: FilterType::Nearest gather (jobs carry the f32 ratios in vtab/htab)
    uint32_t blur_lanes;      // blur kernel: lanes per workgroup of this group (blur_lanes() of its pictures)
    uint32_t tile_w_min;      // tiled two-pass kernel: the narrowest tile width (Job::pad1) among the group's pictures
    uint32_t no_place4;       // place kernel: one pixel per thread (the context's no_place4 switch: identical bytes)
};

struct LaunchStream {
    const Job *jobs;
    const StreamItem *items; // device array, one per workgroup
    const uint32_t *arena;
    uint32_t nitems;
    uint32_t cs, pre, letterbox;
    uint32_t nacc;           // accumulator slots the row schedules were built for (7 or 8)
    uint32_t unaligned;      // Rgb8 sources whose rows are not dword aligned (odd pitch or base): funnel-shift variant
    size_t lds_bytes;
};

hipError_t launch_vpass_generic(const LaunchGeneric &g, hipStream_t st);
hipError_t launch_hpass_generic(const LaunchGeneric &g, hipStream_t st);
hipError_t launch_tile_resample(const LaunchGeneric &g, hipStream_t st); // the same two passes through an LDS tile (no f32 intermediate in HBM)
constexpr uint32_t kTileLdsFloats = 8192;    // LDS of the tiled two-pass kernel: 32 KB = 8 rows x 1024 (source columns x channels) f32: four bytes per thread ...
constexpr uint32_t kTileWeightFloats = 2048;  // ... + the horizontal weights of a tile's columns ...
constexpr uint32_t kTileVRows = 64;           // ... + the vertical weights of a band: source rows one band of 8 output rows may touch (the host checks all three)
// Vertical plan of the tiled two-pass kernel (arena block, built by the host per (axis, kept rows)): header, then per band of 8
// output rows {first source row, source rows touched}, then the dense weights [band][rv_stride][8 output rows] (0 outside a window).
struct TileVPlanHeader { uint32_t nbands, rv_stride, bands_off, dense_off; }; // offsets in words from the header
hipError_t launch_place(const LaunchGeneric &g, bool border_only, hipStream_t st);

// fused LDS-tiled Gaussian blur (g.cs = channels of the blurred image; g.jobs[i].vtab/htab = Gaussian tables)
bool blur_tile_supported(uint32_t htaps);
size_t blur_lds_bytes(uint32_t w, uint32_t channels, uint32_t vtaps, uint32_t htaps);
uint32_t blur_grid_x(uint32_t w, uint32_t h, uint32_t htaps, uint32_t channels_filtered);
uint32_t blur_tile_count(uint32_t w, uint32_t htaps);
uint32_t blur_lanes(uint32_t w, uint32_t htaps); // lanes per workgroup the blur kernel uses for pictures of this width
uint32_t blur_band_rows(uint32_t channels_filtered); // output rows per workgroup of the blur kernel (see BLUR_TY / BLUR_TY_MONO)
hipError_t launch_blur_tile(const LaunchGeneric &g, uint32_t grid_x, size_t lds, hipStream_t st);

bool stream_supported(uint32_t cs, uint32_t pre);
uint32_t stream_block_rows(); // source rows per block of the streaming kernel (emits are deferred to block ends)
size_t stream_lds_bytes(uint32_t jmax, uint32_t nxs, uint32_t ks, uint32_t mid_channels);
uint32_t stream_lanes(); // lanes (threads) per workgroup of the streaming kernel
hipError_t launch_stream(const LaunchStream &s, hipStream_t st);

// EXIF orientation pre-pass: g.cs = channels, jobs[i].fill = EXIF code, sw/sh source size, dw/dh oriented size
hipError_t launch_orient(const LaunchGeneric &g, hipStream_t st);
// CMYK_8 (or YCCK, converted first) -> RGB_8 through a baked lcms2 device-link table; clut = grid^4 nodes of 4 x u16,
// src / dst padded to a multiple of 4 pixels
hipError_t launch_cmyk_clut(const void *src, void *dst, const void *clut, uint32_t grid, uint64_t n_pixels, bool ycck, hipStream_t st);
hipError_t launch_ycck_to_cmyk(uint32_t *px, uint64_t n_pixels, hipStream_t st);

// all_rgba_aligned: every job of the group is Rgba8 with dword-aligned source and destination (4 pixels per thread)
hipError_t launch_jfif444(const FrontendJob *fjobs, uint32_t job_base, uint32_t njobs, uint32_t max_pw, uint32_t max_ph,
                          bool all_rgba_aligned, hipStream_t st);
hipError_t launch_webp420(const FrontendJob *fjobs, const uint32_t *arena, uint32_t gamma_off, uint32_t job_base,
                          uint32_t njobs, uint32_t max_cw, uint32_t max_ch, bool all_rgba_aligned, hipStream_t st);

// JPEG encode of njobs pictures: colour + FDCT + quantise (one wave per block), then entropy coding (one workgroup per picture)
hipError_t launch_jpeg_encode(const JpegJob *jobs, const uint32_t *arena, uint32_t job_base, uint32_t njobs, uint32_t max_blocks,
                              hipStream_t st);

} // namespace fl
