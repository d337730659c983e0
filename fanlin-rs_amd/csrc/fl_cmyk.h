// fl_cmyk.h -- baking of the CMYK -> sRGB device-link table (see fl_cmyk.cpp)
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fl {

constexpr uint32_t kCmykGrid = 17; // cmspcs.c _cmsReasonableGridpointsByColorspace: 4 input channels, default flags

bool cmyk_bake_available();
// nodes: kCmykGrid^4 entries of 4 x u16 (R, G, B, 0), index ((c * G + m) * G + y) * G + k.
// 0 ok, -1 not a usable CMYK profile (the reference's `.ok()?` -> None), -2 liblcms2 not loadable
int bake_cmyk_clut(const uint8_t *icc, size_t n, std::vector<uint16_t> &nodes);
uint64_t hash_bytes(const uint8_t *p, size_t n);

} // namespace fl
