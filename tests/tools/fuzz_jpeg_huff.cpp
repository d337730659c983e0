// Fuzz harness for the host half of the JPEG decode front end (csrc/fl_jpeghuff.cpp), built with AddressSanitizer +
// UBSan by tests/test_jpeg_decode.py: the decoder parses bytes that come from the network (src/handler.rs:192-220), so any
// input must end in "ok" or an error code, never in an out-of-bounds access.
//   fuzz_jpeg_huff <seed file>... : every file is decoded as it is and under `rounds` random mutations
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "fl_jpegdec.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static int decode(const std::vector<uint8_t> &d, long *ok, long *bad)
{
    fl::JpegInfo info;
    if (fl::jpeg_parse_info(d.data(), d.size(), info) != 0) { ++*bad; return 0; }
    if (!info.supported) { ++*bad; return 0; }
    if ((uint64_t)info.width * info.height > (1u << 22)) { ++*bad; return 0; } // mutated dimensions: keep the harness's memory bounded
    const size_t cap = fl::jpeg_blob_bound(info);
    std::vector<uint8_t> blob(cap);
    size_t used = 0;
    const int rc = fl::jpeg_entropy_decode(d.data(), d.size(), blob.data(), cap, &used);
    if (rc == 0) {
        if (used > cap) { fprintf(stderr, "used %zu > capacity %zu\n", used, cap); return 1; }
        ++*ok;
    } else ++*bad;
    // the staging step of the DEVICE entropy decoder reads the same bytes (header, code tables laid out for the kernels, the segment
    // with its stuffing removed): whatever it answers, it must stay inside its buffers, and what it stages must describe itself truthfully
    const size_t scap = fl::jpeg_stage_bound(d.size());
    std::vector<uint8_t> stage(scap);
    size_t sused = 0;
    if (fl::jpeg_entropy_stage(d.data(), d.size(), stage.data(), scap, &sused) == 0) {
        fl::JpegHuffStage S;
        memcpy(&S, stage.data() + sizeof(fl::JpegBlobHeader), sizeof(S));
        if (sused > scap || S.staged_bytes != sused || (size_t)S.stream_off + S.stream_bits / 8u + 16u > sused || S.tables_off + 4u * fl::kJhTableWords * 4u > S.stream_off || S.bpm == 0 || S.bpm > 10) {
            fprintf(stderr, "staged blob inconsistent: used %zu of %zu, stream at %u + %u bits\n", sused, scap, S.stream_off, S.stream_bits);
            return 1;
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const int rounds = 400;
    long ok = 0, bad = 0;
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 2;
        std::vector<uint8_t> seed;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) seed.insert(seed.end(), buf, buf + n);
        fclose(f);
        if (decode(seed, &ok, &bad)) return 1;
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> d = seed;
            const int kind = (int)(rnd() % 5u);
            if (kind == 0) d.resize(rnd() % (d.size() + 1));                                   // truncation
            else if (kind == 1) { for (int k = 0; k < 1 + (int)(rnd() % 8u); ++k) d[rnd() % d.size()] = (uint8_t)rnd(); } // byte flips anywhere
            else if (kind == 2) { const size_t hdr = d.size() < 700 ? d.size() : 700; for (int k = 0; k < 1 + (int)(rnd() % 4u); ++k) d[rnd() % hdr] = (uint8_t)rnd(); } // headers
            else if (kind == 3) { const size_t p = rnd() % d.size(); d.insert(d.begin() + (long)p, (uint8_t)0xFF); d.insert(d.begin() + (long)p + 1, (uint8_t)(0xD0 + rnd() % 16u)); } // stray markers
            else { const size_t p = rnd() % d.size(), q = rnd() % d.size(); const size_t l = rnd() % 64u; for (size_t k = 0; k < l && p + k < d.size() && q + k < d.size(); ++k) d[p + k] = d[q + k]; } // splices
            if (d.empty()) continue;
            if (decode(d, &ok, &bad)) return 1;
        }
    }
    printf("fuzz: %ld decoded, %ld rejected\n", ok, bad);
    return 0;
}
