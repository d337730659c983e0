// fl_pixel.h -- small device helpers shared by the encoder front ends (fl_kernels.hip) and the JPEG back half (fl_jpeg.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fl {

// DynamicImage as GenericImageView<Pixel = Rgba<u8>>: Luma -> (l, l, l, 255), LumaA -> (l, l, l, a), Rgb -> a = 255
__device__ __forceinline__ void load_rgba(const uint8_t *p, uint32_t c, uint32_t &r, uint32_t &g, uint32_t &b, uint32_t &a)
{
    if (c == 1) { r = g = b = p[0]; a = 255u; }
    else if (c == 2) { r = g = b = p[0]; a = p[1]; }
    else if (c == 3) { r = p[0]; g = p[1]; b = p[2]; a = 255u; }
    else { r = p[0]; g = p[1]; b = p[2]; a = p[3]; }
}

__device__ __forceinline__ uint8_t sat_u8(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

// image 0.25.6 codecs/jpeg/encoder.rs rgb_to_ycbcr: f32, truncating casts (this tree is built with -ffp-contract=off)
__device__ __forceinline__ void jfif_px(uint32_t d, uint32_t &y, uint32_t &cb, uint32_t &cr)
{
    const float max = 255.0f;
    const float r = (float)(d & 255u), g = (float)((d >> 8) & 255u), b = (float)((d >> 16) & 255u);
    y = sat_u8(76.245f / max * r + 149.685f / max * g + 29.07f / max * b);
    cb = sat_u8(-43.0185f / max * r - 84.4815f / max * g + 127.5f / max * b + 128.0f);
    cr = sat_u8(127.5f / max * r - 106.7685f / max * g - 20.7315f / max * b + 128.0f);
}

} // namespace fl
