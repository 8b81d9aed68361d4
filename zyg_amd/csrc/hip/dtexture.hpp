// 2-D image textures on the device: fetch, address modes, nearest / stochastic-bilinear filter
// (reference src/core/texture/texture_sampler.zig:84-168, sampler_mode.zig:20-88, texture.zig:188-219)
#pragma once

#include "dscene.hpp"

namespace zd {

// ---- 2-D image textures: texture_sampler.zig:84-168 (Nearest2D / LinearStochastic2D), sampler_mode.zig:20-88, texture.zig:188-219
ZD float texAddress(bool repeat, float x) { return repeat ? x - floorf(x) : clampf(x, 0.f, 1.f); }
ZD int   texCoord(bool repeat, int c, int end) {
    if (repeat) {
        const int m = c % end;  // @mod: the result has the sign of the divisor
        return m < 0 ? m + end : m;
    }
    return max(min(c, end - 1), 0);
}
ZD f3 imageFetch3(const DScene& s, const DImage& img, int x, int y) {
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE3_SRGB == img.format) {
        const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + 3 * i;
        const float    r = s.srgb_to_float[p[0]], g = s.srgb_to_float[p[1]], b = s.srgb_to_float[p[2]];
        // aces.zig:9-17 sRGBtoAP1
        return mk3(0.61309732f, 0.07019422f, 0.02061560f) * S3(r) + mk3(0.33952285f, 0.91635557f, 0.10956983f) * S3(g) +
               mk3(0.04737928f, 0.01345259f, 0.86981512f) * S3(b);
    }
    const float* p = static_cast<const float*>(img.pixels) + 3 * i;
    return mk3(p[0], p[1], p[2]);
}
// ts.sample2D_3 for an image texture; r = rs.stochastic_r picks one of the four texels of the bilinear footprint
ZD f3 sampleImageMap(const DScene& s, uint32_t image, uint32_t sampler, const float* scale, float u, float v, float r) {
    const DImage img = s.images[image];
    const bool   ru = 0 != (sampler & ZYGPU_TEX_REPEAT_U), rv = 0 != (sampler & ZYGPU_TEX_REPEAT_V);
    const float  su = scale[0] * u, sv = scale[1] * v;
    const float  dw = float(img.width), dh = float(img.height);
    if (!(sampler & ZYGPU_TEX_LINEAR)) {
        const int x = min(int(texAddress(ru, su) * dw), img.width - 1), y = min(int(texAddress(rv, sv) * dh), img.height - 1);
        return imageFetch3(s, img, x, y);
    }
    const float mx_ = texAddress(ru, su) * dw - 0.5f, my_ = texAddress(rv, sv) * dh - 0.5f;
    const float fx = floorf(mx_), fy = floorf(my_);
    const float wx = mx_ - fx, wy = my_ - fy;
    const float ox = 1.f - wx, oy = 1.f - wy;
    int         x = int(fx), y = int(fy), index = 0;
    float       threshold = ox * oy;
    index += r > threshold ? 1 : 0;
    threshold = __fmaf_rn(wx, oy, threshold);
    index += r > threshold ? 1 : 0;
    threshold = __fmaf_rn(ox, wy, threshold);
    index += r > threshold ? 1 : 0;
    x += index & 1;
    y += (index & 2) >> 1;
    return imageFetch3(s, img, texCoord(ru, x, img.width), texCoord(rv, y, img.height));
}

}  // namespace zd
