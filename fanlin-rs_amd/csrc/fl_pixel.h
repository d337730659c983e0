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

// Rust's `f32 as u8`: truncates, saturates, NaN -> 0.  v_cvt_u32_f32 truncates toward zero and gives 0 for negative values and
// NaN, so one conversion and one minimum do it (three compares and selects before: these run once per pixel and component in
// kernels whose time is their vector instructions).
__device__ __forceinline__ uint8_t sat_u8(float v)
{
    return (uint8_t)min(__float2uint_rz(v), 255u);
}

// image::color `impl Blend for Rgba<u8>` onto an opaque background (the letterbox
// fill), f32 src-over with truncating casts; alpha 0 keeps the background, 255 replaces it.
__device__ __forceinline__ uint32_t blend_over_fill(uint32_t fill, uint32_t r, uint32_t g, uint32_t b, uint32_t a)
{
    if (a == 0u) return fill;
    if (a == 255u) return r | (g << 8) | (b << 16) | (255u << 24);
    const float max_t = 255.0f;
    float bg_r = (float)(fill & 255u) / max_t, bg_g = (float)((fill >> 8) & 255u) / max_t,
          bg_b = (float)((fill >> 16) & 255u) / max_t, bg_a = (float)(fill >> 24) / max_t;
    float fg_r = (float)r / max_t, fg_g = (float)g / max_t, fg_b = (float)b / max_t, fg_a = (float)a / max_t;
    float alpha_final = bg_a + fg_a - bg_a * fg_a;
    if (alpha_final == 0.0f) return fill;
    float bg_r_a = bg_r * bg_a, bg_g_a = bg_g * bg_a, bg_b_a = bg_b * bg_a;
    float fg_r_a = fg_r * fg_a, fg_g_a = fg_g * fg_a, fg_b_a = fg_b * fg_a;
    float out_r_a = fg_r_a + bg_r_a * (1.0f - fg_a);
    float out_g_a = fg_g_a + bg_g_a * (1.0f - fg_a);
    float out_b_a = fg_b_a + bg_b_a * (1.0f - fg_a);
    float out_r = out_r_a / alpha_final, out_g = out_g_a / alpha_final, out_b = out_b_a / alpha_final;
    uint32_t o_r = (uint32_t)(max_t * out_r), o_g = (uint32_t)(max_t * out_g), o_b = (uint32_t)(max_t * out_b),
             o_a = (uint32_t)(max_t * alpha_final);
    return (o_r & 255u) | ((o_g & 255u) << 8) | ((o_b & 255u) << 16) | ((o_a & 255u) << 24);
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
