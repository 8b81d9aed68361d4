// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp).
//
// 2-D image textures and emission-map importance sampling, restated from the reference:
//   texture fetch + filters     src/core/texture/texture_sampler.zig:84-168, sampler_mode.zig:20-88, texture.zig:188-219
//   Distribution1D / 2D         src/base/math/distribution_1d.zig:49-102, 209-217, distribution_2d.zig:67-90
//   ImageImpl.sample / pdf      src/core/scene/shape/shape_sampler.zig:128-152
#pragma once

#include "scene.hpp"

namespace orc {

// ---- 3-channel fetches (texAddress / texCoord / texelPick: scene.hpp)
inline V4 imageFetch3(const ZygpuSceneDesc& scene, const ZygpuImage& img, int x, int y) {
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE3_SRGB == img.format) {
        const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + 3 * i;
        const float    r = scene.srgb_to_float[p[0]], g = scene.srgb_to_float[p[1]], b = scene.srgb_to_float[p[2]];
        return mk(0.61309732f, 0.07019422f, 0.02061560f) * S(r) + mk(0.33952285f, 0.91635557f, 0.10956983f) * S(g) +
               mk(0.04737928f, 0.01345259f, 0.86981512f) * S(b);  // aces.zig:9-17
    }
    const float* p = static_cast<const float*>(img.pixels) + 3 * i;
    return mk(p[0], p[1], p[2]);
}
// ts.sample2D_3 for an image texture (rs.uv(), rs.stochastic_r)
inline V4 sampleImageMap(const ZygpuSceneDesc& scene, uint32_t image, uint32_t sampler, const float scale[2], float u, float v, float r) {
    const ZygpuImage& img = scene.images[image];
    int               x, y;
    texelPick(img, sampler, scale, u, v, r, x, y);
    return imageFetch3(scene, img, x, y);
}
inline V4 sampleColorMap(const ZygpuSceneDesc& scene, const ZygpuMaterial& m, float u, float v, float r) {
    return sampleImageMap(scene, m.color_image, m.color_sampler, m.color_scale, u, v, r);
}

// ---- Distribution1D over the flattened rows of a ZygpuImageLight (row = conditional index, num_rows = the marginal)
struct Dist1D {
    const float*    cdf;
    const uint32_t* lut;
    uint32_t        size;
    float           lut_range;
    static Dist1D row(const ZygpuImageLight& il, uint32_t r) {
        const uint32_t* w = il.rows + 4 * size_t(r);
        float range;
        std::memcpy(&range, &w[3], 4);
        return {il.cdf + w[0], il.lut + w[2], w[1], range};
    }
    uint32_t sample(float r) const {  // distribution_1d.zig:49-53 + search 209-217
        const uint32_t bucket = uint32_t(r * lut_range);
        uint32_t       i = lut[bucket];
        const uint32_t end = size - 1;
        for (; i < end; ++i)
            if (cdf[i] >= r) break;
        return i - 1;
    }
    void sampleContinuous(float r, float& offset_out, float& pdf) const {  // 61-75
        const uint32_t offset = sample(r);
        const float    c = cdf[offset + 1];
        const float    v = c - cdf[offset];
        if (0.f == v) {
            offset_out = 0.f;
            pdf = 0.f;
            return;
        }
        const float t = (c - r) / v;
        offset_out = (float(offset) + t) / float(size - 1);
        pdf = v;
    }
    float pdfF(float u) const {  // 81-86
        const uint32_t o = std::min(uint32_t(u * float(size - 1)), size - 2);
        return cdf[o + 1] - cdf[o];
    }
};

// ImageImpl.sample: uv and pdf * total_weight
inline float imageLightSample(const ZygpuImageLight& il, float r0, float r1, float& u, float& v) {
    float v_pdf, u_pdf;
    Dist1D::row(il, il.num_rows).sampleContinuous(r1, v, v_pdf);
    const uint32_t c = std::min(uint32_t(v * float(il.num_rows)), il.num_rows - 1);
    Dist1D::row(il, c).sampleContinuous(r0, u, u_pdf);
    return (u_pdf * v_pdf) * il.total_weight;
}
// ImageImpl.pdf(uv): mode.address2 first
inline float imageLightPdf(const ZygpuImageLight& il, float u, float v) {
    const float au = texAddress(0 != (il.sampler & ZYGPU_TEX_REPEAT_U), u), av = texAddress(0 != (il.sampler & ZYGPU_TEX_REPEAT_V), v);
    const float v_pdf = Dist1D::row(il, il.num_rows).pdfF(av);
    const uint32_t c = std::min(uint32_t(av * float(il.num_rows)), il.num_rows - 1);
    const float u_pdf = Dist1D::row(il, c).pdfF(au);
    return (u_pdf * v_pdf) * il.total_weight;
}

}  // namespace orc
