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
    // what the colour kernel's fast form needs of the header, worked out once on the host (round 5: the kernel spent 435 scalar instructions
    // and 48 scalar loads per wave on deriving it -- divisions of sampling factors among them -- for 8 pixels per thread)
    uint32_t mode;         // 0 = pixel-wise from the header; 1 / 2 / 3 = YCbCr with chroma 2x2 / 2x1 / 1x1 and luma at full resolution
    uint32_t width, height;
    uint32_t y_off, y_pitch;            // luma plane: byte offset in `planes`, bytes per row
    uint32_t cb_off, cr_off, c_pitch;   // chroma planes
    uint32_t c_rows;                    // chroma rows that hold samples (JpegComponent::hpx)
    uint32_t c_w;                       // chroma samples per row (JpegComponent::w): the row's last pixel takes sample c_w - 1 as it is
    uint32_t pad_[2];
};
// fills the fields behind `dst` from the header
void jpeg_color_job(const JpegBlobHeader &H, JpegDecJob &j);

hipError_t launch_jpeg_decode(const JpegDecJob *jobs, uint32_t njobs, uint32_t max_blocks, uint32_t max_w, uint32_t max_h, hipStream_t st);

} // namespace fl

// ---- entropy decoding on the device (round 4; fl_jpeghuff_dev.hip) -------------------------------------------------------------
// For sequential Huffman files in one interleaved scan (round 5: with or without restart intervals) the host no longer decodes anything: it parses
// the header, copies the entropy-coded segment with the 0xFF00 stuffing removed, and hands over the four code tables.  The
// device then decodes the segment in parallel -- subsequences of kJhSubBits bits decoded speculatively, then re-decoded from
// their predecessor's end state until the states stop changing (a Huffman decoder that starts at a wrong bit re-synchronises
// after a few code words), a prefix sum over the subsequences' block counts and DC sums, and a last pass that writes the
// coefficients -- into the SAME blob the host decoder writes (every block "wide", 64 coefficients), so the IDCT / colour
// kernels do not know the difference.
namespace fl {

constexpr uint32_t kJhSubBits = 1024;          // bits per subsequence (round 5 tried 512: twice the walks at half the length -- the first kernel took 961 us against 907, profiles/r05_jpeg_decode_kernels.txt)
constexpr uint32_t kJhLookBits = 12;           // lookahead of the device's code table: every code of up to 12 bits in one LDS read (Annex K: all but the 15- and 16-bit ones)
constexpr uint32_t kJhTableWords = 2148;       // per code table: look[4096 x u16] = code length | symbol << 8 (0: a longer code), maxcode[18], valoff[17], vals[256 x u8]
constexpr uint32_t kJhSyncRounds = 2;          // launches of the re-synchronisation kernel after the speculative one (each iterates inside its workgroups)
constexpr uint32_t kJhMagic = 0x32444a46u;     // "FJD2": a staged entropy-coded segment instead of coefficients

// Behind the JpegBlobHeader of a staged source (magic kJhMagic; the header's blocks_off / coef_off describe the blob the DEVICE builds).
struct alignas(16) JpegHuffStage {
    uint32_t stream_off;      // byte offset (from the start of the staged blob) of the unstuffed segment, 16-byte aligned, padded with 16 bytes
    uint32_t stream_bits;     // its length in bits (whole bytes)
    uint32_t tables_off;      // byte offset of 4 tables x kJhTableWords words: DC 0, DC 1, AC 0, AC 1
    uint32_t bpm;             // blocks per MCU (1..10)
    uint32_t mcux, mcuy;      // MCUs per row / column
    uint32_t total_blocks;    // = header.nblocks
    uint32_t staged_bytes;    // size of the staged blob
    uint8_t blk_comp[12];     // per block of an MCU: component, and its position inside the MCU's h x v group
    uint8_t blk_h[12], blk_v[12];
    uint8_t dc_tab[4], ac_tab[4]; // per component: table index 0..3 into the four tables above (AC: 2..3)
    // restart intervals (round 5): the RSTn markers are taken out of the staged segment like the stuffing; what is left of a marker is the
    // byte offset at which the next interval starts (every interval starts on a byte: the bits in front of it are padding, F.1.2.3)
    uint32_t rst_mcus;        // MCUs per restart interval (DRI), 0 = none
    uint32_t n_rst;           // interval starts recorded (= intervals - 1)
    uint32_t rst_off;         // byte offset (from the start of the staged blob) of n_rst ascending u32 byte offsets into the segment
    uint32_t pad_[1];
};
constexpr uint32_t kJhMaxRestarts = 16384; // files with more restart intervals stay with the host decoder

// One staged picture of a device entropy-decode launch.
struct alignas(16) JhJob {
    const uint8_t *stage;  // device copy of the staged blob (JpegBlobHeader, JpegHuffStage, tables, segment)
    uint8_t *blob;         // device blob to build: header, block words, 64 x i16 per block
    uint64_t *used;        // [nsub]: the start state every subsequence was last decoded from
    uint64_t *states;      // [nsub + 1]: end state of every subsequence (bit position | block of the MCU << 32 | coefficient index << 40); [0] = the start
    int32_t *counts;       // [nsub][4]: blocks completed, DC difference sums of components 0..2
    int32_t *prefix;       // [nsub][4]: exclusive prefix sums of the same
    uint32_t *err;         // device error word of the picture (1: the state chain did not settle, 2: invalid code / DC out of range)
    uint32_t nsub;
    uint32_t pad;
};
constexpr uint32_t kJhSubsPerItem = 240;       // subsequences a workgroup owns (its 256 threads also decode kJhWarm = 16 in front of them in the speculative kernel)
struct JhItem { uint32_t job, first_sub; }; // one workgroup of the per-subsequence kernels: kJhSubsPerItem consecutive subsequences of one picture

// Host half: parses `data`, and if the file is one the device entropy decoder takes (sequential, one interleaved scan, with or without restart
// interval, 1 or 3 components) writes the staged blob to `out`: 0 ok (*used = its size), -1 malformed, -2 not for this path.
int jpeg_entropy_stage(const uint8_t *data, size_t n, uint8_t *out, size_t cap, size_t *used);
size_t jpeg_stage_bound(size_t file_bytes);
// Bytes of the device blob / per-subsequence scratch a staged picture needs.
size_t jh_blob_bytes(const JpegBlobHeader &H);
uint32_t jh_subsequences(const JpegHuffStage &S);
hipError_t launch_jpeg_huff(const JhJob *d_jobs, const JhJob *h_jobs, uint32_t njobs, const JhItem *d_items, uint32_t nitems, uint32_t max_blocks, hipStream_t st);

} // namespace fl
