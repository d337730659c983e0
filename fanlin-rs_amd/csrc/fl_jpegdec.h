// fl_jpegdec.h -- JPEG decode front end (reference src/handler.rs:205-220: image 0.25.6 -> zune-jpeg 0.4.14).
//
// Split as SURVEY 8 f2 plans it: the serial part of a baseline JPEG -- marker parsing and Huffman decoding -- runs on the
// host (on the CALLER's thread of flgpu_transform, so concurrent requests decode in parallel), and produces a compact
// blob of quantised coefficients (typically 1/5 of the decoded picture); the data-parallel part -- dequantisation, the
// integer IDCT, chroma up-sampling and YCbCr -> RGB -- runs on the device and feeds the resample kernel directly.
// A 1080p request then uploads ~1 MB instead of 6.2 MB.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace fl {

struct JpegComponent {
    uint32_t h, v;          // sampling factors
    uint32_t bw, bh;        // blocks per row / column (padded to whole MCUs)
    uint32_t w, hpx;        // real plane size in samples: ceil(W * h / hmax), ceil(H * v / vmax)
    uint32_t block_base;    // index of this component's first block word
    uint32_t plane_off;     // byte offset of its plane (bw*8 x bh*8 samples) inside the per-picture plane scratch
    uint32_t tq, pad;
};

// Header of the coefficient blob (host -> device), followed by the block words and the coefficients.
struct alignas(16) JpegBlobHeader {
    uint32_t magic;             // 'FJD1'
    uint32_t width, height, nc; // stored components: 1 (decodes to Luma8), 3 (Rgb8) or 4 (raw CMYK / YCCK samples, 4 bytes per pixel)
    uint32_t hmax, vmax;
    uint32_t is_rgb;            // three components with Adobe transform 0: they are R, G, B already
    uint32_t nblocks;           // over all components
    uint32_t blocks_off;        // byte offset of u32 block words [nblocks]: (data offset in halfwords << 7) | (count - 1) << 1 | wide
    uint32_t coef_off;          // byte offset of the block data: quantised coefficients in zig-zag order up to the last non-zero
                                // one; wide blocks store all of them as i16, narrow blocks the first kJpegWideHead as i16 and
                                // the rest as i8 (typical files: ~1.2 bytes per coded coefficient)
    uint32_t plane_bytes;       // scratch the planes need
    uint32_t total_bytes;       // size of the whole blob
    uint32_t adobe_transform;   // APP14 transform byte + 1 (0 = no Adobe segment): four components with 2 + 1 are YCCK
    uint32_t pad[3];
    JpegComponent comp[4];
    uint16_t qt[4][64];         // per COMPONENT, zig-zag order
};

constexpr uint32_t kJpegWideHead = 4; // coefficients of a narrow block kept as i16 (DC and the largest ACs)

struct JpegInfo {
    uint32_t width = 0, height = 0, components = 0;
    uint32_t progressive = 0, precision = 0, restart_interval = 0, hmax = 0, vmax = 0;
    uint32_t sof = 0;              // marker byte of the frame header (0xC0 baseline, 0xC1 extended sequential, 0xC2 progressive)
    uint32_t exif_orientation = 0; // 1..8, 0 = no tag
    int adobe_transform = -1;
    uint32_t supported = 0;        // 1 = the device path decodes it
    std::vector<uint8_t> icc;      // embedded ICC profile (APP2 "ICC_PROFILE" chunks in order), empty if none / inconsistent
};

// 0 = ok, -1 = not a JPEG / malformed header
int jpeg_parse_info(const uint8_t *data, size_t n, JpegInfo &info);

// Size bound of the blob for `info` (worst case: every coefficient present).
size_t jpeg_blob_bound(const JpegInfo &info);

// Entropy-decodes a baseline stream into `blob` (capacity >= jpeg_blob_bound): 0 = ok, -1 = malformed, -2 = unsupported.
int jpeg_entropy_decode(const uint8_t *data, size_t n, uint8_t *blob, size_t cap, size_t *used);

// One picture of a decode launch.
struct alignas(16) JpegDecJob {
    const uint8_t *blob;   // device copy of the blob
    uint8_t *planes;       // device scratch, header.plane_bytes
    uint8_t *dst;          // decoded pixels, width*height*nc, tightly packed
};

hipError_t launch_jpeg_decode(const JpegDecJob *jobs, uint32_t njobs, uint32_t max_blocks, uint32_t max_w, uint32_t max_h, hipStream_t st);

} // namespace fl
