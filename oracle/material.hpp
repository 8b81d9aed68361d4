// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp).
//
// Material closures restated from the reference (uniform textures only):
//   Material.sample dispatch    src/core/scene/material/material.zig:184-194
//   sample Base / bxdf types    material/sample_base.zig:11-108, material/bxdf.zig:5-80
//   Debug (Lambert)             material/debug/debug_material.zig:20-28, debug_sample.zig:15-47
//   Light (no scattering)       material/light/light_sample.zig:7-13, material_sample.zig:47-60
//   Substitute (opaque base)    material/substitute/substitute_material.zig:114-221,
//                               substitute_sample.zig:38-86 (init), 88-145 (evaluate), 147-234 (sample),
//                               236-302 (baseEvaluate/baseSample), 338-410 (diffuseSample/glossSample)
//   diffuse.Micro               material/diffuse.zig:42-110
//   ggx Iso/Aniso, dspbrMicroEc material/ggx.zig:30-46, 60-130, 228-449
//   fresnel.Schlick             material/fresnel.zig:5-29
//   LUT evaluation              base/math/interpolated_function.zig:105-271, base/math/math.zig:172-191
//   Glass (thick, specular)     material/glass/glass_material.zig:46-80, glass_sample.zig:33-66 (init), 68-80 (evaluate: empty),
//                               165-254 (specularSample), 425-435 (reflect), 519-537 (thickSpecularRefract);
//                               fresnel.dielectric material/fresnel.zig:31-43
#pragma once

#include "procedural.hpp"
#include "sampler.hpp"
#include "scene.hpp"
#include "texture.hpp"

namespace orc {

enum class Scattering : uint8_t { Diffuse, Glossy, Specular, None };
enum class PathEvent : uint8_t { Reflection, Transmission, Straight };

struct PathInfo {  // bxdf.zig:34-70
    float      reg_alpha;
    Scattering scattering;
    PathEvent  event;
    bool       singular() const { return 0.f == reg_alpha; }
    static PathInfo diffuseReflection() { return {1.f, Scattering::Diffuse, PathEvent::Reflection}; }
    static PathInfo singularReflection() { return {0.f, Scattering::Specular, PathEvent::Reflection}; }
    static PathInfo singularTransmission() { return {0.f, Scattering::Specular, PathEvent::Transmission}; }
    static PathInfo straight() { return {1.f, Scattering::None, PathEvent::Straight}; }
    static PathInfo reflection(float alpha, float specular_threshold) {
        return {alpha, alpha <= specular_threshold ? Scattering::Specular : Scattering::Glossy, PathEvent::Reflection};
    }
};

struct BxdfResult {
    V4    reflection;
    float pdf;
};

struct BxdfSample {
    V4       reflection, wi;
    float    pdf, split_weight, wavelength;
    PathInfo path;
};

struct Renderstate {  // renderstate.zig:10-45 (subset in scope)
    Trafo    trafo;
    V4       p, geo_n, t, b, n, origin, uvw;
    float    stochastic_r, ior;
    uint32_t prop, part;
    uint32_t primitive = 0;                     // frag.isec.primitive (vertex.zig:172): a mesh's surface differentials
    Sampler* sampler = nullptr;                 // procedural textures may draw (Mix, procedural_mix.zig:17-26)
    const ProceduralContext* pctx = nullptr;    // context.camera at rs.time
    bool     caustics;
    int32_t  highest_priority = -128;  // mediums.highestPriority() (prop/medium.zig:75-83)
    float    wavelength = 0.f;                   // probe.wavelength (vertex.zig:167): 0 until a dispersive Glass picks one
    float    reg_weight = 0.f, reg_alpha = 0.f;  // PathtracerMIS' regularize_roughness and Vertex.reg_alpha (vertex.zig:168-169)

    // the coordinates a texture is read at (texture_sampler.zig:22-35): rs.uv(), or rs.triplanarSt() (renderstate.zig:39-49)
    void st(uint32_t sampler, float out[2]) const;

    // renderstate.zig:60-69: a path that has scattered diffusely sees later surfaces rougher
    void regularizeAlpha(float alpha[2], float specular_threshold) const {
        if (0.f == reg_weight || (alpha[0] <= specular_threshold && !caustics)) return;
        const float k = 1.f - reg_weight * reg_alpha;
        alpha[0] = 1.f - ((1.f - alpha[0]) * k);
        alpha[1] = 1.f - ((1.f - alpha[1]) * k);
    }
};

struct Luts {
    const ZygpuLuts* l;
    static float bilinear(const float c[4], float s, float t) {  // math.zig:172-179
        const float _s = 1.f - s, _t = 1.f - t;
        return _t * (_s * c[0] + s * c[1]) + t * (_s * c[2] + s * c[3]);
    }
    static float eval1(const float* samples, int N, float x) {  // InterpolatedFunction1DN.fromArray/eval
        const float    cx = mn(x, 1.f);
        const float    o = cx * float(N - 1);
        const uint32_t offset = uint32_t(o);
        const float    t = o - float(offset);
        return lerpf(samples[offset], samples[std::min<uint32_t>(offset + 1, uint32_t(N - 1))], t);
    }
    static float eval2(const float* samples, int X, int Y, float x, float y) {
        const float   ox = mn(x, 1.f) * float(X - 1), oy = mn(y, 1.f) * float(Y - 1);
        const int32_t o0 = int32_t(ox), o1 = int32_t(oy);
        const float   t0 = ox - float(o0), t1 = oy - float(o1);
        const int32_t col1 = std::min(o0 + 1, X - 1);
        const int32_t row0 = o1 * X, row1 = std::min(o1 + 1, Y - 1) * X;
        const float   c[4] = {samples[o0 + row0], samples[col1 + row0], samples[o0 + row1], samples[col1 + row1]};
        return bilinear(c, t0, t1);
    }
    static float eval3(const float* samples, int X, int Y, int Z, float x, float y, float z) {
        const float   ox = mn(x, 1.f) * float(X - 1), oy = mn(y, 1.f) * float(Y - 1), oz = mn(z, 1.f) * float(Z - 1);
        const int32_t o0 = int32_t(ox), o1 = int32_t(oy), o2 = int32_t(oz);
        const float   t0 = ox - float(o0), t1 = oy - float(o1), t2 = oz - float(o2);
        const int32_t col1 = std::min(o0 + 1, X - 1);
        const int32_t row0 = o1 * X, row1 = std::min(o1 + 1, Y - 1) * X;
        const int32_t area = X * Y;
        const int32_t s0 = o2 * area, s1 = std::min(o2 + 1, Z - 1) * area;
        const float   ca[4] = {samples[o0 + row0 + s0], samples[col1 + row0 + s0], samples[o0 + row1 + s0],
                               samples[col1 + row1 + s0]};
        const float   cb[4] = {samples[o0 + row0 + s1], samples[col1 + row0 + s1], samples[o0 + row1 + s1],
                               samples[col1 + row1 + s1]};
        return lerpf(bilinear(ca, t0, t1), bilinear(cb, t0, t1), t2);
    }
    float E_m(float n_dot, float alpha) const { return eval2(l->e_m, 32, 32, n_dot, alpha); }
    float E_m_avg(float alpha) const { return eval1(l->e_m_avg, 32, alpha); }
    float E(float n_dot, float alpha, float f0) const { return eval3(l->e, 16, 16, 16, n_dot, alpha, f0); }
    float E_avg(float alpha, float f0) const { return eval2(l->e_avg, 16, 16, alpha, f0); }
    // ggx.ilmEpDielectric (ggx.zig:30-32): E_s over f0 in [0, 0.25] (ggx_integral.zig:1045 E_s_inverse_max_f0 = 4)
    float ilmEpDielectric(float n_dot_wo, float alpha, float f0) const { return 1.f / eval3(l->e_s, 16, 16, 16, n_dot_wo, alpha, f0 * 4.f); }
};

namespace fresnel {
inline float pow5(float x) {
    const float x2 = x * x, x4 = x2 * x2;
    return x4 * x;
}
inline V4    schlick(V4 f0, float wo_dot_h) { return fma4(S(pow5(1.f - wo_dot_h)), S(1.f) - f0, f0); }
inline float schlick1(float wo_dot_h, float f0) { return std::fmaf(pow5(1.f - wo_dot_h), 1.f - f0, f0); }  // fresnel.zig:5-7
inline float iorToF0(float n0, float n1) {
    const float t = (n0 - n1) / (n0 + n1);
    return t * t;
}
inline float dielectric(float cos_theta_i, float cos_theta_t, float eta_i, float eta_t) {  // fresnel.zig:31-43
    const float t0 = eta_t * cos_theta_i, t1 = eta_i * cos_theta_t;
    const float r_p = (t0 - t1) / (t0 + t1);
    const float t2 = eta_i * cos_theta_i, t3 = eta_t * cos_theta_t;
    const float r_o = (t2 - t3) / (t2 + t3);
    return 0.5f * (r_p * r_p + r_o * r_o);
}
}  // namespace fresnel

namespace ggx {

constexpr float kMinRoughness = 0.01314f;
inline float    clampRoughness(float r) { return mx(r, kMinRoughness); }

struct Micro {
    V4    h;
    float n_dot_wi, h_dot_wi;
};

inline V4 dspbrMicroEc(const Luts& L, V4 f0, float n_dot_wi, float n_dot_wo, float alpha) {  // ggx.zig:34-46
    const float e_wo = L.E_m(n_dot_wo, alpha), e_wi = L.E_m(n_dot_wi, alpha), e_avg = L.E_m_avg(alpha);
    const float m = ((1.f - e_wo) * (1.f - e_wi)) / (kPi * (1.f - e_avg));
    const V4    f_avg = fma4(S(20.f / 21.f), f0, S(1.f / 21.f));
    const V4    f = ((f_avg * f_avg) * S(e_avg)) / fma4(-f_avg, S(1.f - e_avg), S(1.f));
    return S(m) * f;
}

inline float pdfVisible(float d, float g1_wo) { return (0.5f * d) / g1_wo; }

inline float isoDistribution(float n_dot_h, float a2) {  // ggx.zig:228-231
    const float d = std::fmaf(n_dot_h * n_dot_h, a2 - 1.f, 1.f);
    return a2 / (kPi * d * d);
}
inline void isoVisibilityAndG1Wo(float n_dot_wi, float n_dot_wo, float alpha2, float g[2]) {  // :233-243
    const float t_wi = std::sqrt(std::fmaf(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));
    const float t_wo = std::sqrt(std::fmaf(1.f - alpha2, n_dot_wo * n_dot_wo, alpha2));
    g[0] = 0.5f / (n_dot_wi * t_wo + n_dot_wo * t_wi);
    g[1] = t_wo + n_dot_wo;
}
inline float anisoDistribution(float n_dot_h, float x_dot_h, float y_dot_h, const float a[2]) {  // :416-424
    const float a20 = a[0] * a[0], a21 = a[1] * a[1];
    const float x = (x_dot_h * x_dot_h) / a20, y = (y_dot_h * y_dot_h) / a21;
    const float d = (x + y) + (n_dot_h * n_dot_h);
    return 1.f / (kPi * (a[0] * a[1]) * (d * d));
}
inline void anisoVisibilityAndG1Wo(float t_dot_wi, float t_dot_wo, float b_dot_wi, float b_dot_wo, float n_dot_wi,
                                   float n_dot_wo, const float a[2], float g[2]) {  // :426-440
    const float t_wo = length3(mk(a[0] * t_dot_wo, a[1] * b_dot_wo, n_dot_wo));
    const float t_wi = length3(mk(a[0] * t_dot_wi, a[1] * b_dot_wi, n_dot_wi));
    g[0] = 0.5f / (n_dot_wi * t_wo + n_dot_wo * t_wi);
    g[1] = t_wo + n_dot_wo;
}

// ggx.zig:396-414 (VNDF, spherical caps)
inline V4 anisoSample(V4 wo, const float alpha[2], float xi0, float xi1, const Frame& frame, float& n_dot_h) {
    const V4    wo_l = frame.worldToFrame(wo);
    const V4    v = normalize3(mk(alpha[0] * wo_l[0], alpha[1] * wo_l[1], wo_l[2]));
    const float phi = (2.f * kPi) * xi0;
    const float z = std::fmaf(1.f - xi1, 1.f + v[2], -v[2]);
    const float sin_theta = std::sqrt(saturate(1.f - z * z));
    const float x = sin_theta * cosF(phi), y = sin_theta * sinF(phi);
    const V4    h = mk(x, y, z) + v;
    const V4    m = normalize3(mk(alpha[0] * h[0], alpha[1] * h[1], h[2]));
    n_dot_h = clampf(m[2], kDotMin, 1.f);
    return frame.frameToWorld(m);
}

// Aniso.reflection (ggx.zig:250-300), falling to Iso.reflectionF (74-97) when alpha is isotropic
inline BxdfResult reflection(V4 wi, V4 wo, V4 h, float n_dot_wi, float n_dot_wo, float wo_dot_h, const float alpha[2], V4 f0,
                             const Frame& frame) {
    float d, g[2];
    if (alpha[0] == alpha[1]) {
        const float alpha2 = alpha[0] * alpha[0];
        const float n_dot_h = saturate(dot3(frame.z, h));
        d = isoDistribution(n_dot_h, alpha2);
        isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g);
    } else {
        const float n_dot_h = saturate(dot3(frame.z, h));
        d = anisoDistribution(n_dot_h, dot3(frame.x, h), dot3(frame.y, h), alpha);
        anisoVisibilityAndG1Wo(dot3(frame.x, wi), dot3(frame.x, wo), dot3(frame.y, wi), dot3(frame.y, wo), n_dot_wi, n_dot_wo,
                               alpha, g);
    }
    const V4 f = fresnel::schlick(f0, wo_dot_h);
    return {S(d * g[0]) * f, pdfVisible(d, g[1])};
}

// Aniso.reflect (ggx.zig:302-345) / Iso.reflect (99-130)
inline Micro reflect(V4 wo, float n_dot_wo, const float alpha[2], float specular_threshold, float xi0, float xi1, V4 f0,
                     const Frame& frame, BxdfSample& result) {
    float       n_dot_h;
    const V4    h = anisoSample(wo, alpha, xi0, xi1, frame, n_dot_h);
    const float wo_dot_h = clampDot(wo, h);
    const V4    wi = normalize3(fma4(S(2.f * wo_dot_h), h, -wo));
    const float n_dot_wi = frame.clampNdot(wi);
    float       d, g[2];
    if (alpha[0] == alpha[1]) {
        const float alpha2 = alpha[0] * alpha[0];
        d = isoDistribution(n_dot_h, alpha2);
        isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g);
    } else {
        d = anisoDistribution(n_dot_h, dot3(frame.x, h), dot3(frame.y, h), alpha);
        anisoVisibilityAndG1Wo(dot3(frame.x, wi), dot3(frame.x, wo), dot3(frame.y, wi), dot3(frame.y, wo), n_dot_wi, n_dot_wo,
                               alpha, g);
    }
    const V4 f = fresnel::schlick(f0, wo_dot_h);
    result.reflection = S(d * g[0]) * f;
    result.wi = wi;
    result.pdf = pdfVisible(d, g[1]);
    result.path = PathInfo::reflection(alpha[1], specular_threshold);
    return {h, n_dot_wi, wo_dot_h};
}

// ---- rough dielectrics: ggx.Iso.reflectNoFresnel / refractNoFresnel / reflectionF / refractionF (ggx.zig:73-95, 127-227)
inline float gSmithCorrelated(float n_dot_wi, float n_dot_wo, float alpha2) {  // :246-251
    const float a = n_dot_wo * std::sqrt(std::fmaf(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));
    const float b = n_dot_wi * std::sqrt(std::fmaf(1.f - alpha2, n_dot_wo * n_dot_wo, alpha2));
    return (2.f * n_dot_wi * n_dot_wo) / (a + b);
}
inline float pdfVisibleRefract(float n_dot_wo, float wo_dot_h, float d, float alpha2) {  // :441-449
    const float g1 = (2.f * n_dot_wo) / (n_dot_wo + std::sqrt(alpha2 + (1.f - alpha2) * (n_dot_wo * n_dot_wo)));
    return g1 * wo_dot_h * d / n_dot_wo;
}
inline float clampAbs(float x) { return clampf(std::fabs(x), kDotMin, 1.f); }  // safe.zig:16-18

inline float reflectNoFresnel(V4 wo, V4 h, float n_dot_wo, float n_dot_h, float wo_dot_h, float alpha, float specular_threshold,
                              const Frame& frame, BxdfSample& result) {
    const V4    wi = normalize3(fma4(S(2.f * wo_dot_h), h, -wo));
    const float n_dot_wi = frame.clampNdot(wi);
    const float alpha2 = alpha * alpha;
    const float d = isoDistribution(n_dot_h, alpha2);
    float       g[2];
    isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g);
    result.reflection = S(d * g[0]);
    result.wi = wi;
    result.pdf = pdfVisible(d, g[1]);
    result.path = PathInfo::reflection(alpha, specular_threshold);
    return n_dot_wi;
}
inline float refractNoFresnel(V4 wo, V4 h, float n_dot_wo, float n_dot_h, float wi_dot_h, float wo_dot_h, float alpha,
                              float specular_threshold, float eta_i, float eta_t, const Frame& frame, BxdfSample& result) {
    const float eta = eta_i / eta_t;
    const float abs_wi_dot_h = clampAbs(wi_dot_h), abs_wo_dot_h = clampAbs(wo_dot_h);
    const V4    wi = normalize3(S(std::fmaf(eta, abs_wo_dot_h, -abs_wi_dot_h)) * h - S(eta) * wo);
    const float n_dot_wi = frame.clampAbsNdot(wi);
    const float alpha2 = alpha * alpha;
    const float d = isoDistribution(n_dot_h, alpha2);
    const float g = gSmithCorrelated(n_dot_wi, n_dot_wo, alpha2);
    const float refr = d * g;
    const float factor = (abs_wi_dot_h * abs_wo_dot_h) / (n_dot_wi * n_dot_wo);
    const float dn = eta_i * wo_dot_h + eta_t * wi_dot_h, denom = dn * dn;
    const float sqr_eta_t = eta_t * eta_t;
    const float pdf = pdfVisibleRefract(n_dot_wo, abs_wo_dot_h, d, alpha2);
    result.reflection = S((factor * sqr_eta_t / denom) * refr);
    result.wi = wi;
    result.pdf = pdf * (abs_wi_dot_h * sqr_eta_t / denom);
    result.path = {alpha, alpha <= specular_threshold ? Scattering::Specular : Scattering::Glossy, PathEvent::Transmission};
    return n_dot_wi;
}
struct ResultF {
    BxdfResult r;
    float      f;
};
inline ResultF reflectionF(V4 h, V4 n, float n_dot_wi, float n_dot_wo, float wo_dot_h, float alpha, float f0) {
    const float alpha2 = alpha * alpha;
    const float n_dot_h = saturate(dot3(n, h));
    const float d = isoDistribution(n_dot_h, alpha2);
    float       g[2];
    isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g);
    const V4 f = fresnel::schlick(S(f0), wo_dot_h);
    return {{S(d * g[0]) * f, pdfVisible(d, g[1])}, f[0]};
}
inline ResultF refractionF(float n_dot_wi, float n_dot_wo, float wi_dot_h, float wo_dot_h, float n_dot_h, float alpha, float eta_i,
                           float eta_t, float f0) {
    const float alpha2 = alpha * alpha;
    const float abs_wi_dot_h = clampAbs(wi_dot_h), abs_wo_dot_h = clampAbs(wo_dot_h);
    const float d = isoDistribution(n_dot_h, alpha2);
    const float g = gSmithCorrelated(n_dot_wi, n_dot_wo, alpha2);
    const float cos_x = eta_i > eta_t ? abs_wi_dot_h : abs_wo_dot_h;
    const float f = 1.f - fresnel::schlick(S(f0), cos_x)[0];
    const float sqr_eta_t = eta_t * eta_t;
    const float factor = (abs_wi_dot_h * abs_wo_dot_h) / (n_dot_wi * n_dot_wo);
    const float dn = eta_i * wo_dot_h + eta_t * wi_dot_h, denom = dn * dn;
    const float refr = d * g * f;
    const float refl = (factor * sqr_eta_t / denom) * refr;
    const float pdf = pdfVisibleRefract(n_dot_wo, abs_wo_dot_h, d, alpha2);
    return {{S(refl), pdf * (abs_wi_dot_h * sqr_eta_t / denom)}, f};
}

}  // namespace ggx

namespace diffuse_micro {  // diffuse.zig:42-110

inline float estimateContribution(const Luts& L, float alpha, float f0, float albedo) {
    const float e_avg = L.E_avg(alpha, f0);
    const float a = e_avg;
    const float b = 1.f / (kPi * (1.f - e_avg)) * albedo;
    return b / (a + b);
}
inline V4 evaluate(const Luts& L, V4 color, float n_dot_wi, float n_dot_wo, float alpha, float f0) {
    const float e_wo = L.E(n_dot_wo, alpha, f0), e_wi = L.E(n_dot_wi, alpha, f0), e_avg = L.E_avg(alpha, f0);
    return S(((1.f - e_wo) * (1.f - e_wi)) / (kPi * (1.f - e_avg))) * color;
}
inline BxdfResult reflection(const Luts& L, V4 color, float f0, float n_dot_wi, float n_dot_wo, float alpha) {
    return {evaluate(L, color, n_dot_wi, n_dot_wo, alpha, f0), n_dot_wi * kPiInv};
}
inline ggx::Micro reflect(const Luts& L, V4 color, float f0, V4 wo, float n_dot_wo, const Frame& frame, float alpha, float xi0,
                          float xi1, BxdfSample& result) {
    const V4    is = hemisphereCosine(xi0, xi1);
    const V4    wi = normalize3(frame.frameToWorld(is));
    const V4    h = normalize3(wo + wi);
    const float h_dot_wi = clampDot(h, wi);
    const float n_dot_wi = frame.clampNdot(wi);
    result.reflection = evaluate(L, color, n_dot_wi, n_dot_wo, alpha, f0);
    result.wi = wi;
    result.pdf = n_dot_wi * kPiInv;
    result.path = PathInfo::diffuseReflection();
    return {h, n_dot_wi, h_dot_wi};
}

}  // namespace diffuse_micro

// The spectrum a dispersive Glass weights its wavelength draw with: rainbow_integral.zig:3-41 (a generated table: 32 bands
// between 380 and 720 nm; tests/golden/reference_constants.json holds the reference's numbers), material_base.zig:56-73
constexpr float kRainbowStart = 380.f, kRainbowEnd = 720.f;
constexpr int   kRainbowBands = 32;
static const float kRainbow[kRainbowBands][3] = {
    {0.00118749f, 0.00000000f, 0.01277892f}, {0.00428908f, 0.00000000f, 0.04665186f}, {0.01393313f, 0.00000000f, 0.15397894f},
    {0.04654151f, 0.00000000f, 0.52832049f}, {0.10802665f, 0.00000000f, 1.00000000f}, {0.13396028f, 0.00000000f, 1.00000000f},
    {0.11002098f, 0.00000000f, 1.00000000f}, {0.05638582f, 0.00000000f, 1.00000000f}, {0.00323226f, 0.04778787f, 1.00000000f},
    {0.00000000f, 0.19377914f, 0.82263476f}, {0.00000000f, 0.36241144f, 0.45633313f}, {0.00000000f, 0.60088211f, 0.25896370f},
    {0.00000000f, 0.92129856f, 0.13798122f}, {0.00000000f, 1.00000000f, 0.06571960f}, {0.06015893f, 1.00000000f, 0.03276091f},
    {0.25364995f, 1.00000000f, 0.01465140f}, {0.49281576f, 1.00000000f, 0.00740940f}, {0.76601928f, 1.00000000f, 0.00603952f},
    {1.00000000f, 0.90572321f, 0.00713901f}, {1.00000000f, 0.63148290f, 0.00846385f}, {1.00000000f, 0.37303036f, 0.00921571f},
    {1.00000000f, 0.17757514f, 0.00884298f}, {1.00000000f, 0.06006211f, 0.00760580f}, {0.90220016f, 0.00659840f, 0.00571195f},
    {0.61788863f, 0.00000000f, 0.00391464f}, {0.38115984f, 0.00000000f, 0.00241280f}, {0.21279863f, 0.00000000f, 0.00134878f},
    {0.10870021f, 0.00000000f, 0.00068967f}, {0.05491111f, 0.00000000f, 0.00034864f}, {0.02550786f, 0.00000000f, 0.00016202f},
    {0.01237659f, 0.00000000f, 0.00007862f}, {0.00601408f, 0.00000000f, 0.00003820f}};
inline V4 spectrumAtWavelength(float lambda, float value) {
    const float    u = ((lambda - kRainbowStart) / (kRainbowEnd - kRainbowStart)) * float(kRainbowBands);
    const uint32_t id = uint32_t(u);
    const float    frac = u - float(id);
    if (id >= uint32_t(kRainbowBands - 1)) return mk(kRainbow[kRainbowBands - 1][0], kRainbow[kRainbowBands - 1][1], kRainbow[kRainbowBands - 1][2]);
    return S(value) * lerp4(mk(kRainbow[id][0], kRainbow[id][1], kRainbow[id][2]), mk(kRainbow[id + 1][0], kRainbow[id + 1][1], kRainbow[id + 1][2]), S(frac));
}

struct MaterialSample {
    uint32_t kind;
    Frame    frame;
    V4       geo_n, n, wo;
    float    alpha[2];
    bool     can_evaluate, translucent, avoid_caustics;
    // Debug
    V4 albedo;
    // Substitute
    V4    f0;
    float metallic, specular, specular_threshold, opacity;
    Luts  luts;
    // Glass
    V4    absorption;
    float ior, ior_outside, glass_f0, thickness;
    float abbe = 0.f, wavelength = 0.f;  // dispersion: the material's Abbe number, the path's wavelength (rs.wavelength)
    bool  lower_priority;
    // Substitute: clearcoat (substitute_coating.zig:15-113) and flakes (sample_base.zig:17)
    struct Coating {
        V4    n, absorption_coef;
        float thickness = 0.f, f0, alpha, weight;
        struct Result {
            V4    reflection, attenuation;
            float f, pdf;
        };
        V4 singleAttenuation(float n_dot_wo) const { return attenuation3(absorption_coef, thickness * (1.f / n_dot_wo)); }  // :97-105
        V4 attenuation(float n_dot_wi, float n_dot_wo) const {  // :107-113
            const float f = weight * fresnel::schlick1(mn(n_dot_wi, n_dot_wo), f0);
            const float d = thickness * (1.f / n_dot_wi + 1.f / n_dot_wo);
            return S(1.f - f) * attenuation3(absorption_coef, d);
        }
        Result evaluate(const Luts& L, V4 wi, V4 wo, V4 h, float wo_dot_h, float specular_threshold, bool avoid_caustics) const {  // :35-56
            const float n_dot_wi = clampDot(n, wi), n_dot_wo = clampAbsDot(n, wo);
            const V4    att = attenuation(n_dot_wi, n_dot_wo);
            if (avoid_caustics && alpha <= specular_threshold) return {S(0.f), att, 0.f, 0.f};
            const ggx::ResultF gg = ggx::reflectionF(h, n, n_dot_wi, n_dot_wo, wo_dot_h, alpha, f0);
            const float        ep = L.ilmEpDielectric(n_dot_wo, alpha, f0);
            return {S(ep * weight * n_dot_wi) * gg.r.reflection, att, gg.f, gg.r.pdf};
        }
        V4 reflect(const Luts& L, V4 wo, V4 h, float n_dot_wo, float n_dot_h, float wo_dot_h, float specular_threshold,
                   BxdfSample& result) const {  // :58-85
            const float n_dot_wi = ggx::reflectNoFresnel(wo, h, n_dot_wo, n_dot_h, wo_dot_h, alpha, specular_threshold, Frame::init(n), result);
            const float ep = L.ilmEpDielectric(n_dot_wo, alpha, f0);
            result.reflection *= S(ep * weight * n_dot_wi);
            return attenuation(n_dot_wi, n_dot_wo);
        }
        ggx::Micro sample(V4 wo, float xi0, float xi1, float& n_dot_h) const {  // :87-95: n_dot_wi carries the Fresnel term
            const float a[2] = {alpha, alpha};
            const V4    h = ggx::anisoSample(wo, a, xi0, xi1, Frame::init(n), n_dot_h);
            const float wo_dot_h = clampDot(wo, h);
            return {h, fresnel::schlick1(wo_dot_h, f0), wo_dot_h};
        }
    } coating;
    bool flakes = false;

    bool sameHemisphere(V4 v) const { return dot3(geo_n, v) > 0.f; }

    // ---- Substitute ----
    BxdfResult baseEvaluate(V4 wi, V4 woo, V4 h, float wo_dot_h, bool force_disable_caustics) const {
        const float n_dot_wi = frame.clampNdot(wi);
        const float n_dot_wo = frame.clampAbsNdot(woo);
        BxdfResult  d{S(0.f), 0.f};
        float       dw = 0.f;
        if (1.f != metallic) {
            const V4    alb = S(opacity) * albedo;
            const float f0m = hmax3(f0);
            d = diffuse_micro::reflection(luts, alb, f0m, n_dot_wi, n_dot_wo, alpha[1]);
            const float am = hmax3(albedo);
            dw = diffuse_micro::estimateContribution(luts, alpha[1], f0m, am);
        }
        if ((force_disable_caustics || avoid_caustics) && alpha[1] <= specular_threshold) {
            return {S(n_dot_wi) * d.reflection, dw * d.pdf};
        }
        const float      s = specular;
        if (flakes) {  // substitute_sample.zig:265-271: a flake reflects into a cone around the mirror direction
            const float cos_cone = alpha[0];
            const V4    r = reflect3(frame.z, woo);
            const float f = dot3(wi, r) > cos_cone ? 1.f / solidAngleOfCone(cos_cone) : 0.f;
            return {S(n_dot_wi * f * s) * f0, f};
        }
        const BxdfResult gg = ggx::reflection(wi, woo, h, n_dot_wi, n_dot_wo, wo_dot_h, alpha, f0, frame);
        const V4         mms = ggx::dspbrMicroEc(luts, f0, n_dot_wi, n_dot_wo, alpha[1]);
        const float      pdf = dw * d.pdf + (1.f - dw) * gg.pdf;
        return {S(n_dot_wi) * (d.reflection + S(s) * (gg.reflection + mms)), pdf};
    }

    static float diffuseFresnelHack(float n_dot_wi, float n_dot_wo, float f0_) {  // substitute_sample.zig:440-442
        return fresnel::schlick1(mx(n_dot_wi, n_dot_wo), f0_);
    }

    float diffuseWeight() const {  // the estimate shared by baseSample / coatingSample (substitute_sample.zig:284-294, 316-326)
        if (1.f == metallic) return 0.f;
        return diffuse_micro::estimateContribution(luts, alpha[1], hmax3(f0), hmax3(albedo));
    }

    ggx::Micro diffuseSample(float diffuse_weight, float xi0, float xi1, BxdfSample& result) const {
        const float      n_dot_wo = frame.clampAbsNdot(wo);
        const V4         alb = S(opacity) * albedo;
        const float      f0m = hmax3(f0);
        const ggx::Micro micro = diffuse_micro::reflect(luts, alb, f0m, wo, n_dot_wo, frame, alpha[1], xi0, xi1, result);
        const BxdfResult gg = ggx::reflection(result.wi, wo, micro.h, micro.n_dot_wi, n_dot_wo, micro.h_dot_wi, alpha, f0, frame);
        const V4         mms = ggx::dspbrMicroEc(luts, f0, micro.n_dot_wi, frame.clampNdot(wo), alpha[1]);
        const float      s = specular;
        result.reflection = S(micro.n_dot_wi) * (result.reflection + S(s) * (gg.reflection + mms));
        result.pdf = diffuse_weight * result.pdf + (1.f - diffuse_weight) * gg.pdf;
        return micro;
    }

    ggx::Micro glossSample(float diffuse_weight, float xi0, float xi1, BxdfSample& result) const {
        const float      s = specular;
        if (flakes) {  // substitute_sample.zig:376-396
            const float cos_cone = alpha[0];
            const V4    wi_l = coneUniform(xi0, xi1, cos_cone);
            const V4    h = reflect3(frame.z, wo);
            const V4    wi = Frame::init(h).frameToWorld(wi_l);
            const float wi_dot_h = clampDot(wi, h);
            const float f = wi_dot_h > cos_cone ? 1.f / solidAngleOfCone(cos_cone) : 0.f;
            const float n_dot_wi = frame.clampNdot(wi);
            result.reflection = S(n_dot_wi * f * s) * f0;
            result.wi = wi;
            result.pdf = f;
            result.path = PathInfo::reflection(alpha[1], specular_threshold);
            return {h, n_dot_wi, wi_dot_h};
        }
        const float      n_dot_wo = frame.clampAbsNdot(wo);
        const ggx::Micro micro = ggx::reflect(wo, n_dot_wo, alpha, specular_threshold, xi0, xi1, f0, frame, result);
        const V4         mms = ggx::dspbrMicroEc(luts, f0, micro.n_dot_wi, frame.clampNdot(wo), alpha[1]);
        BxdfResult       d{S(0.f), 0.f};
        if (diffuse_weight > 0.f) {
            const V4    alb = S(opacity) * albedo;
            const float f0m = hmax3(f0);
            d = diffuse_micro::reflection(luts, alb, f0m, micro.n_dot_wi, n_dot_wo, alpha[1]);
        }
        result.reflection = S(micro.n_dot_wi) * (S(s) * (result.reflection + mms) + d.reflection);
        result.pdf = (1.f - diffuse_weight) * result.pdf + diffuse_weight * d.pdf;
        return micro;
    }

    void baseSample(Sampler& sampler, BxdfSample& result) const {  // substitute_sample.zig:281-302
        const float dw = diffuseWeight();
        float       s3[3];
        sampler.sample3D(s3);
        if (s3[0] < dw) diffuseSample(dw, s3[1], s3[2], result);
        else glossSample(dw, s3[1], s3[2], result);
    }

    // substitute_sample.zig:304-336, coatingReflect 412-422, coatingBaseSample 426-433
    void coatingSample(Sampler& sampler, BxdfSample& result) const {
        float s2[2];
        sampler.sample2D(s2);
        float            coat_n_dot_h;
        const ggx::Micro micro = coating.sample(wo, s2[0], s2[1], coat_n_dot_h);
        const float      f = micro.n_dot_wi;
        float            s3[3];
        sampler.sample3D(s3);
        const float p = s3[0];
        if (p <= f) {
            const float n_dot_wo = clampAbsDot(coating.n, wo);
            const V4    coating_attenuation = coating.reflect(luts, wo, micro.h, n_dot_wo, coat_n_dot_h, micro.h_dot_wi, specular_threshold, result);
            const BxdfResult base_result = baseEvaluate(result.wi, wo, micro.h, micro.h_dot_wi, false);
            result.reflection = (result.reflection * S(f)) + coating_attenuation * base_result.reflection;
            result.pdf = f * result.pdf + (1.f - f) * base_result.pdf;
        } else {
            const float      dw = diffuseWeight();
            const float      p1 = (p - f) / (1.f - f);
            const ggx::Micro bm = p1 < dw ? diffuseSample(dw, s3[1], s3[2], result) : glossSample(dw, s3[1], s3[2], result);
            const Coating::Result c = coating.evaluate(luts, result.wi, wo, bm.h, bm.h_dot_wi, specular_threshold, avoid_caustics);
            result.reflection = c.attenuation * result.reflection + c.reflection;
            result.pdf = (1.f - f) * result.pdf + f * c.pdf;
        }
    }

    // ---- dispatch: material_sample.zig:47-60 ----
    // Glass.Sample.evaluate (glass_sample.zig:68-146): rough interfaces only; `split`: max_splits > 1
    BxdfResult glassEvaluate(V4 wi, bool split, bool force_disable_caustics) const {
        const float a = alpha[0];
        if (ior == ior_outside || !(a > 0.f) || lower_priority || ((force_disable_caustics || avoid_caustics) && a <= specular_threshold))
            return {S(0.f), 0.f};
        const float s = specular;
        if (!sameHemisphere(wo)) {  // thick refraction from inside
            const float eta_i = ior, eta_t = ior_outside;
            const V4    h = -normalize3(S(eta_t) * wi + S(eta_i) * wo);
            const float wi_dot_h = dot3(wi, h);
            if (wi_dot_h <= 0.f) return {S(0.f), 0.f};
            const float wo_dot_h = dot3(wo, h);
            const float eta = eta_i / eta_t;
            const float sint2 = (eta * eta) * (1.f - wo_dot_h * wo_dot_h);
            if (sint2 >= 1.f) return {S(0.f), 0.f};
            const float n_dot_wi = frame.clampNdot(wi), n_dot_wo = frame.clampAbsNdot(wo);
            const float n_dot_h = saturate(dot3(frame.z, h));
            const ggx::ResultF gg = ggx::refractionF(n_dot_wi, n_dot_wo, wi_dot_h, wo_dot_h, n_dot_h, a, eta_i, eta_t, glass_f0);
            const float comp = luts.ilmEpDielectric(n_dot_wo, a, glass_f0);
            const float split_pdf = split ? 1.f : gg.f;
            return {S(mn(n_dot_wi, n_dot_wo) * comp * s) * gg.r.reflection, split_pdf * gg.r.pdf};
        }
        if (sameHemisphere(wi)) {
            const float n_dot_wi = frame.clampNdot(wi), n_dot_wo = frame.clampAbsNdot(wo);
            const V4    h = normalize3(wo + wi);
            const float wo_dot_h = clampDot(wo, h);
            const ggx::ResultF gg = ggx::reflectionF(h, frame.z, n_dot_wi, n_dot_wo, wo_dot_h, a, glass_f0);
            const float comp = luts.ilmEpDielectric(n_dot_wo, a, glass_f0);
            const float split_pdf = split ? 1.f : gg.f;
            return {S(n_dot_wi * comp * s) * gg.r.reflection, split_pdf * gg.r.pdf};
        }
        return {S(0.f), 0.f};
    }

    // Glass.roughSample, thick (glass_sample.zig:294-422)
    // weight / ior_t / wl: the dispersion draw of Glass.sample (1, the material's ior and 0 without dispersion)
    uint32_t glassRoughSample(Sampler& sampler, bool split, BxdfSample out[2], V4 weight, float ior_t, float wl) const {
        float eta_i = ior_outside, eta_t = ior_t;
        if (std::fabs(eta_i - eta_t) <= 2.e-7f || lower_priority) {
            out[0].reflection = weight;
            out[0].wi = -wo;
            out[0].pdf = 1.f;
            out[0].split_weight = 1.f;
            out[0].wavelength = wl;
            out[0].path = PathInfo::singularTransmission();
            return 1;
        }
        const float a = alpha[0];
        const bool  same_side = sameHemisphere(wo);
        Frame       fr = frame;
        if (!same_side) {
            fr.z = -fr.z;
            std::swap(eta_i, eta_t);
        }
        float s3[3];
        sampler.sample3D(s3);
        float       n_dot_h;
        const V4    h = ggx::anisoSample(wo, alpha, s3[1], s3[2], fr, n_dot_h);
        const float n_dot_wo = fr.clampAbsNdot(wo);
        const float wo_dot_h = clampDot(wo, h);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - wo_dot_h * wo_dot_h);
        const float s = specular;
        float       wi_dot_h, f;
        if (sint2 >= 1.f) {
            wi_dot_h = 0.f;
            f = 1.f;
        } else {
            wi_dot_h = std::sqrt(1.f - sint2);
            const float cos_x = eta_i > eta_t ? wi_dot_h : wo_dot_h;
            f = fresnel::schlick(S(glass_f0), cos_x)[0];  // fresnel.schlick1
        }
        const float ep = luts.ilmEpDielectric(n_dot_wo, a, glass_f0);
        const float r_wo_dot_h = same_side ? -wo_dot_h : wo_dot_h;
        // roughRefract (glass_sample.zig:437-503): the thick branch refracts through the micro-facet; the thin branch
        // (a slab without volume) reflects about the facet re-expressed in a frame around -wo, i.e. scatters the straight-through
        // direction by the roughness, attenuates over the slab and calls the event .straight; < 0: no sample
        auto roughRefract = [&](BxdfSample& result) {
            if (thickness > 0.f) {
                const Frame thin_frame = Frame::init(-wo);
                const V4    tangent_h = fr.worldToFrame(h);
                const V4    thin_h = thin_frame.frameToWorld(tangent_h);
                const float thin_n_dot_wo = tangent_h[2], thin_n_dot_h = saturate(tangent_h[2]), thin_wo_dot_h = 1.f;
                const float n_dot_wi = ggx::reflectNoFresnel(thin_h, thin_h, thin_n_dot_wo, thin_n_dot_h, thin_wo_dot_h, a,
                                                             specular_threshold, thin_frame, result);
                if (sameHemisphere(result.wi)) return -1.f;
                const float approx_dist = thickness / n_dot_wo;
                result.reflection *= attenuation3(absorption, approx_dist);
                result.path = PathInfo::straight();
                return n_dot_wi;
            }
            return ggx::refractNoFresnel(wo, h, n_dot_wo, n_dot_h, -wi_dot_h, r_wo_dot_h, a, specular_threshold, eta_i, eta_t, fr, result);
        };
        if (split) {
            {
                const float n_dot_wi = ggx::reflectNoFresnel(wo, h, n_dot_wo, n_dot_h, wo_dot_h, a, specular_threshold, fr, out[0]);
                out[0].reflection *= S(n_dot_wi * ep * s) * weight;
                out[0].split_weight = f;
                out[0].wavelength = wl;
            }
            if (1.f == f) return 1;
            const float n_dot_wi = roughRefract(out[1]);
            if (n_dot_wi < 0.f) return 1;
            out[1].reflection *= S(n_dot_wi * ep) * weight;
            out[1].split_weight = 1.f - f;
            out[1].wavelength = wl;
            return 2;
        }
        const float p = s3[0];
        if (p <= f) {
            const float n_dot_wi = ggx::reflectNoFresnel(wo, h, n_dot_wo, n_dot_h, wo_dot_h, a, specular_threshold, fr, out[0]);
            out[0].reflection *= S(f * n_dot_wi * ep * s) * weight;
            out[0].pdf *= f;
        } else {
            const float n_dot_wi = roughRefract(out[0]);
            if (n_dot_wi < 0.f) return 0;
            const float omf = 1.f - f;
            out[0].reflection *= S(omf * n_dot_wi * ep) * weight;
            out[0].pdf *= omf;
        }
        out[0].split_weight = 1.f;
        out[0].wavelength = wl;
        return 1;
    }

    // max_splits: the integrator's split budget (pathtracer_mis.zig:107-124); only rough Glass looks at it here
    BxdfResult evaluate(V4 wi, bool force_disable_caustics, uint32_t max_splits = 1) const {
        if (ZYGPU_MAT_GLASS == kind) return glassEvaluate(wi, max_splits > 1, force_disable_caustics);
        if (ZYGPU_MAT_LIGHT == kind) return {S(0.f), 0.f};
        if (ZYGPU_MAT_DEBUG == kind) {  // debug_sample.zig:19-26
            const float n_dot_wi = frame.clampNdot(wi);
            const float pdf = n_dot_wi * kPiInv;
            return {S(pdf) * albedo, pdf};
        }
        // Substitute.evaluate: substitute_sample.zig:88-145 (exit_sss / volumetric: subsurface scattering, out of scope)
        const float op = opacity;
        const bool  thin = op < 1.f;  // `translucent` there
        if (thin) {
            if (!sameHemisphere(wi)) {
                const float n_dot_wi = frame.clampAbsNdot(wi), n_dot_wo = frame.clampAbsNdot(wo);
                const float f = diffuseFresnelHack(n_dot_wi, n_dot_wo, f0[0]);
                const float pdf = n_dot_wi * ((1.f - op) * kPiInv);
                return {S(pdf * (1.f - f)) * albedo, pdf};
            }
        } else if (!sameHemisphere(wo)) {
            return {S(0.f), 0.f};
        }
        const V4    h = normalize3(wo + wi);
        const float wo_dot_h = clampDot(wo, h);
        BxdfResult  base_result = baseEvaluate(wi, wo, h, wo_dot_h, force_disable_caustics);
        if (thin) base_result.pdf *= op;
        if (coating.thickness > 0.f) {
            const Coating::Result c = coating.evaluate(luts, wi, wo, h, wo_dot_h, specular_threshold, force_disable_caustics || avoid_caustics);
            const float           pdf = c.f * c.pdf + (1.f - c.f) * base_result.pdf;
            return {c.reflection + c.attenuation * base_result.reflection, pdf};
        }
        return base_result;
    }

    // Glass.specularSample, thick (glass_sample.zig:165-254); `split`: both the reflected and the refracted direction,
    // weighted by the Fresnel term, instead of a choice between them
    uint32_t glassSample(Sampler& sampler, bool split, BxdfSample out[2]) const {
        // Glass.sample (glass_sample.zig:165-199): a thick dispersive glass (abbe != 0) first settles the path's wavelength (a
        // uniform draw over the visible range, weighted by the Rainbow spectrum x 3; a path that has one keeps it, weight 1) and
        // takes its index of refraction at that wavelength (Cauchy's equation through the Abbe number)
        V4    weight = S(1.f);
        float ior_t = ior, wl = 0.f;
        if (!(thickness > 0.f) && 0.f != abbe) {
            wl = wavelength;
            const float r1 = sampler.sample1D();
            if (0.f == wl) {  // wavelengthSpectrumWeight (glass_sample.zig:154-163)
                wl = kRainbowStart + (kRainbowEnd - kRainbowStart) * r1;
                weight = spectrumAtWavelength(wl, 1.f) * S(3.f);
            }
            const float sqr_wl = wl * wl;
            ior_t = ior + ((ior - 1.f) / abbe) * (523655.f / sqr_wl - 1.5168f);
        }
        if (alpha[0] > 0.f) return glassRoughSample(sampler, split, out, weight, ior_t, wl);
        float eta_i = ior_outside, eta_t = ior_t;
        auto  fill = [&](BxdfSample& r, V4 wi, V4 reflection, float split_weight, PathInfo path) {
            r.reflection = reflection;
            r.wi = wi;
            r.pdf = 1.f;
            r.split_weight = split_weight;
            r.wavelength = PathEvent::Straight == path.event ? 0.f : wl;  // thinSpecularRefract: 0 (glass_sample.zig:514)
            r.path = path;
        };
        if (eta_i == eta_t || lower_priority) {
            fill(out[0], -wo, weight, 1.f, PathInfo::singularTransmission());
            return 1;
        }
        const bool thin = thickness > 0.f;
        V4         nn = frame.z;
        if (!thin && !sameHemisphere(wo)) {  // thin material is always double sided (glass_sample.zig:203-209)
            nn = -nn;
            std::swap(eta_i, eta_t);
        }
        const float n_dot_wo = mn(std::fabs(dot3(nn, wo)), 1.f);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - n_dot_wo * n_dot_wo);
        float       n_dot_t, f;
        if (sint2 >= 1.f) {
            n_dot_t = 0.f;
            f = 1.f;
        } else {
            n_dot_t = std::sqrt(1.f - sint2);
            f = fresnel::dielectric(n_dot_wo, n_dot_t, eta_i, eta_t);
        }
        const V4 reflected = normalize3(S(2.f * n_dot_wo) * nn - wo);                       // reflect, 425-435
        const V4 refracted = normalize3(S(eta * n_dot_wo - n_dot_t) * nn - S(eta) * wo);  // thickSpecularRefract, 519-537
        // thinSpecularRefract (glass_sample.zig:505-517): straight through, attenuated over the slab
        const V4 thin_attenuation = thin ? attenuation3(absorption, thickness / clampf(n_dot_wo, kDotMin, 1.f)) : S(1.f);
        if (split) {
            fill(out[0], reflected, S(specular) * weight, f, PathInfo::singularReflection());
            if (1.f == f) return 1;
            if (thin) fill(out[1], -wo, thin_attenuation, 1.f - f, PathInfo::straight());
            else fill(out[1], refracted, weight, 1.f - f, PathInfo::singularTransmission());
            return 2;
        }
        const float p = sampler.sample1D();
        if (p <= f) fill(out[0], reflected, S(specular) * weight, 1.f, PathInfo::singularReflection());
        else if (thin) fill(out[0], -wo, thin_attenuation, 1.f, PathInfo::straight());
        else fill(out[0], refracted, weight, 1.f, PathInfo::singularTransmission());
        return 1;
    }

    // MaterialSample.sample with the integrator's split budget (pathtracer_mis.zig:123-124): up to two samples
    uint32_t sampleSplit(Sampler& sampler, uint32_t max_splits, BxdfSample out[2]) const {
        if (ZYGPU_MAT_GLASS == kind) return glassSample(sampler, max_splits > 1, out);
        return sample(sampler, out[0]);
    }

    // returns number of samples (0 or 1)
    uint32_t sample(Sampler& sampler, BxdfSample& result) const {
        if (ZYGPU_MAT_GLASS == kind) {
            BxdfSample two[2];
            const uint32_t n = glassSample(sampler, false, two);
            result = two[0];
            return n;
        }
        if (ZYGPU_MAT_LIGHT == kind) return 0;
        if (ZYGPU_MAT_DEBUG == kind) {  // debug_sample.zig:28-47
            float s2d[2];
            sampler.sample2D(s2d);
            const V4    is = hemisphereCosine(s2d[0], s2d[1]);
            const V4    wi = normalize3(frame.frameToWorld(is));
            const float n_dot_wi = frame.clampNdot(wi);
            const float pdf = n_dot_wi * kPiInv;
            result.reflection = S(pdf) * albedo;
            result.wi = wi;
            result.pdf = pdf;
            result.split_weight = 1.f;
            result.wavelength = 0.f;
            result.path = PathInfo::diffuseReflection();
            return 1;
        }
        // Substitute.sample: substitute_sample.zig:147-234 (exit_sss / volumetric: subsurface scattering, out of scope)
        result.split_weight = 1.f;
        result.wavelength = 0.f;
        const float op = opacity;
        if (op < 1.f) {  // translucent: diffuse transmission with probability 1 - opacity (:184-208)
            const float tr = 1.f - op;
            float       s3[3];
            sampler.sample3D(s3);
            if (s3[0] <= tr) {
                float s2d[2];
                sampler.sample2D(s2d);  // diffuse.Lambert.reflect (diffuse.zig:14-29)
                const V4    is = hemisphereCosine(s2d[0], s2d[1]);
                const V4    wi = normalize3(frame.frameToWorld(is));
                const float n_dot_wi = frame.clampNdot(wi);
                const float n_dot_wo = frame.clampAbsNdot(wo);
                const float f = diffuseFresnelHack(n_dot_wi, n_dot_wo, f0[0]);
                result.wi = -wi;
                result.reflection = (S(kPiInv) * albedo) * S(tr * n_dot_wi * (1.f - f));
                result.pdf = (n_dot_wi * kPiInv) * tr;
                result.path = PathInfo::diffuseReflection();
            } else {
                baseSample(sampler, result);
                result.pdf *= op;
            }
            return 1;
        }
        if (!sameHemisphere(wo)) return 0;
        if (coating.thickness > 0.f) coatingSample(sampler, result);
        else baseSample(sampler, result);
        if (0.f == result.pdf) return 0;
        return 1;
    }
};

// hlp.triplanarMapping (material_helper.zig:81-93); p and n in object space
inline void triplanarMapping(V4 p, V4 n, float uv[2]) {
    const float an[3] = {std::fabs(n[0]), std::fabs(n[1]), std::fabs(n[2])};
    if (an[0] > an[1] && an[0] > an[2]) {
        const float sign = std::copysign(1.f, n[0]);
        uv[0] = sign * p[2] + 0.5f;
        uv[1] = -p[1] + 0.5f;
    } else if (an[1] > an[0] && an[1] > an[2]) {
        const float sign = std::copysign(1.f, n[1]);
        uv[0] = sign * p[0] + 0.5f;
        uv[1] = -p[2] + 0.5f;
    } else {
        const float sign = std::copysign(1.f, n[2]);
        uv[0] = -sign * p[0] + 0.5f;
        uv[1] = -p[1] + 0.5f;
    }
}

inline void Renderstate::st(uint32_t sampler, float out[2]) const {
    if (sampler & ZYGPU_TEX_TRIPLANAR) {
        triplanarMapping(trafo.worldToObjectPoint(p), trafo.worldToObjectNormal(geo_n), out);
    } else {
        out[0] = uvw[0];
        out[1] = uvw[1];
    }
}

// ---- ts.sample2D_1 / _2 / _3 (texture_sampler.zig:22-76) over ZygpuTexture: uniform, image or procedural
inline float texture1(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs);
inline void  texture2(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs, float out[2]);
inline V4    texture3(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs);

// Context.screenspaceDifferentials (context.zig:139-149)
inline V4 screenspaceDifferentials(const ZygpuSceneDesc& scene, const Renderstate& rs, uint32_t mode) {
    const RayDifferential     rd = cameraRayDifferential(*rs.pctx, rs.p);
    const bool                uv0 = 0 == (mode & (ZYGPU_TEX_TRIPLANAR | ZYGPU_TEX_OBJECTPOS));
    const DifferentialSurface ds = uv0 ? surfaceDifferentials(scene, rs.prop, rs.primitive, rs.trafo) : triplanarDifferentials(rs.geo_n, rs.trafo);
    return calculateScreenspaceDifferentials(rs.p, rs.geo_n, rd, ds.dpdu, ds.dpdv);
}

// Checker.evaluate (procedural_checker.zig:46-84)
inline V4 checkerEvaluate(const ZygpuSceneDesc& scene, const ZygpuProcedural& c, uint32_t mode, const Renderstate& rs) {
    const V4 dd = screenspaceDifferentials(scene, rs, mode);
    float    st[2];
    rs.st(mode, st);  // Triplanar: rs.triplanarSt(), else rs.uv()
    const float scale = c.checker_scale;
    const float uv[2] = {scale * texAddress(0 != (mode & ZYGPU_TEX_REPEAT_U), st[0]), scale * texAddress(0 != (mode & ZYGPU_TEX_REPEAT_V), st[1])};
    const float ddx[2] = {scale * dd[0], scale * dd[1]}, ddy[2] = {scale * dd[2], scale * dd[3]};
    float       i[2];
    for (int k = 0; k < 2; ++k) {  // checkersGrad: the triangle-filtered integral
        const float w = mx(std::fabs(ddx[k]), std::fabs(ddy[k])) + 0.0001f;
        auto        pf = [](float x) {
            const float h = ((x * 0.5f) - std::floor(x * 0.5f)) - 0.5f;
            return x * 0.5f + h * (1.f - 2.f * std::fabs(h));
        };
        i[k] = (pf(uv[k] + w) - 2.f * pf(uv[k]) + pf(uv[k] - w)) / (w * w);
    }
    const float t = 0.5f - 0.5f * i[0] * i[1];
    return lerp4(ld3(c.value[0]), ld3(c.value[1]), S(t));
}

// Noise.evaluate1 (procedural_noise.zig:44-101)
inline float noiseEvaluate1(const ZygpuProcedural& n, uint32_t mode, const Renderstate& rs, V4 offset) {
    const bool  cellular = 0 != (n.flags & ZYGPU_NOISE_CELLULAR);
    const float att = n.attenuation;
    float       weight = 0.f, amplitude = 1.f, value = 0.f;
    if (mode & ZYGPU_TEX_OBJECTPOS) {
        V4       scale = mk(n.scale[0], n.scale[1], n.scale[2], n.scale[3]);
        const V4 uvw = rs.trafo.worldToObjectPoint(rs.p - offset);
        for (uint32_t l = 0; l < n.levels; ++l) {
            const float local_weight = noise::powZ(amplitude, att);
            const float local = cellular ? noise::worley3D(uvw * scale, 1.f) : noise::perlin3D(uvw * scale);
            value += local * local_weight;
            weight += local_weight;
            amplitude *= 0.5f;
            scale = scale * S(2.f);
        }
    } else {
        float scale[2] = {n.scale[0], n.scale[1]}, period[2] = {n.period[0], n.period[1]};
        float st[2];
        rs.st(mode, st);
        const float uv[2] = {st[0] - offset[0], st[1] - offset[1]};
        for (uint32_t l = 0; l < n.levels; ++l) {
            const float local_weight = noise::powZ(amplitude, att);
            const float local = cellular ? noise::worley2D(uv[0] * scale[0], uv[1] * scale[1], 1.f)
                                         : noise::perlin2D(uv[0] * scale[0], uv[1] * scale[1], period[0], period[1]);
            value += local * local_weight;
            weight += local_weight;
            amplitude *= 0.5f;
            scale[0] *= 2.f, scale[1] *= 2.f;
            period[0] *= 2.f, period[1] *= 2.f;
        }
    }
    value /= weight;
    const float unsigned_ = cellular ? value : ((n.flags & ZYGPU_NOISE_ABSOLUTE) ? std::fabs(value) : (value + 1.f) * 0.5f);
    const float a = n.ratio - n.transition, b = n.ratio + n.transition;
    const float result = saturate((unsigned_ - a) / (b - a));
    return (n.flags & ZYGPU_NOISE_INVERT) ? 1.f - result : result;
}

// Noise.evaluateNormalmap (procedural_noise.zig:103-135)
inline void noiseNormalmap(const ZygpuSceneDesc& scene, const ZygpuProcedural& n, uint32_t mode, const Renderstate& rs, float out[2]) {
    float nx, ny, nz;
    if (mode & ZYGPU_TEX_OBJECTPOS) {
        V4 dpdx, dpdy;
        approximateDpDxy(*rs.pctx, rs.p, rs.geo_n, dpdx, dpdy);
        const float center = noiseEvaluate1(n, mode, rs, S(0.f));
        const float left = noiseEvaluate1(n, mode, rs, dpdx);
        const float bottom = noiseEvaluate1(n, mode, rs, dpdy);
        nx = left - center;
        ny = bottom - center;
        nz = length3(dpdx + dpdy);
    } else {
        const V4    dd = abs4(screenspaceDifferentials(scene, rs, mode));
        const float shift_x = dd[0] + dd[2], shift_y = dd[1] + dd[3];
        const float center = noiseEvaluate1(n, mode, rs, S(0.f));
        const float left = noiseEvaluate1(n, mode, rs, mk(shift_x, 0.f, 0.f));
        const float top = noiseEvaluate1(n, mode, rs, mk(0.f, shift_y, 0.f));
        nx = left - center;
        ny = top - center;
        nz = std::sqrt(shift_x * shift_x + shift_y * shift_y);  // math.length2
    }
    const V4 nn = normalize3(mk(nx, ny, nz));
    out[0] = nn[0];
    out[1] = nn[1];
}

// Procedural.sample2D_1 / _2 / _3 (procedural.zig:82-130)
inline float procedural1(const ZygpuSceneDesc& scene, uint32_t index, uint32_t mode, const Renderstate& rs);
inline void  procedural2(const ZygpuSceneDesc& scene, uint32_t index, uint32_t mode, const Renderstate& rs, float out[2]);
inline V4    procedural3(const ZygpuSceneDesc& scene, uint32_t index, uint32_t mode, const Renderstate& rs);

inline float texture1(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs) {
    if (kNull == t.image) return uniform[0];
    if (t.sampler & ZYGPU_TEX_PROCEDURAL) return procedural1(scene, t.image, t.sampler, rs);
    const uint32_t format = scene.images[t.image].format;
    if (ZYGPU_IMAGE_BYTE1_UNORM != format && ZYGPU_IMAGE_FLOAT1 != format) return 0.f;  // image2D_1 of another width (texture.zig:166-175)
    float st[2];
    rs.st(t.sampler, st);
    return sampleImageMap1(scene, t.image, t.sampler, t.scale, st[0], st[1], rs.stochastic_r);
}
inline void texture2(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs, float out[2]) {
    if (kNull == t.image) {
        out[0] = uniform[0], out[1] = uniform[1];
        return;
    }
    if (t.sampler & ZYGPU_TEX_PROCEDURAL) return procedural2(scene, t.image, t.sampler, rs, out);
    const uint32_t format = scene.images[t.image].format;
    if (ZYGPU_IMAGE_BYTE2 != format && ZYGPU_IMAGE_FLOAT2 != format) {  // image2D_2 (texture.zig:177-186)
        out[0] = out[1] = 0.f;
        return;
    }
    float st[2];
    rs.st(t.sampler, st);
    sampleImageMap2(scene, t.image, t.sampler, t.scale, st[0], st[1], rs.stochastic_r, out);
}
inline V4 texture3(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs) {
    if (kNull == t.image) return mk(uniform[0], uniform[1], uniform[2]);
    if (t.sampler & ZYGPU_TEX_PROCEDURAL) return procedural3(scene, t.image, t.sampler, rs);
    float st[2];
    rs.st(t.sampler, st);
    const uint32_t format = scene.images[t.image].format;
    if (ZYGPU_IMAGE_BYTE1_UNORM == format || ZYGPU_IMAGE_FLOAT1 == format)  // image2D_3 of a one-channel image (texture.zig:192-195)
        return mk(sampleImageMap1(scene, t.image, t.sampler, t.scale, st[0], st[1], rs.stochastic_r), 0.f, 0.f);
    if (ZYGPU_IMAGE_BYTE3_SRGB != format && ZYGPU_IMAGE_FLOAT3 != format) return S(0.f);
    return sampleImageMap(scene, t.image, t.sampler, t.scale, st[0], st[1], rs.stochastic_r);
}

inline V4 detailReconstruct(const ZygpuSceneDesc& scene, const ZygpuTexture& t, const float* uniform, const Renderstate& rs) {  // procedural_detail_normal.zig:31-35
    float nm[2];
    texture2(scene, t, uniform, rs, nm);
    const float nmz = std::sqrt(mx(1.f - (nm[0] * nm[0] + nm[1] * nm[1]), 0.01f));
    return normalize3(mk(nm[0], nm[1], nmz));
}

inline float procedural1(const ZygpuSceneDesc& scene, uint32_t index, uint32_t mode, const Renderstate& rs) {
    const ZygpuProcedural& p = scene.procedurals[index];
    switch (p.kind) {
        case ZYGPU_PROC_CHANNEL_MIX: {
            const V4 c = texture3(scene, p.child[0], p.child_uniform[0], rs);
            return clampf(dot3(c, ld3(p.value[0])), 0.f, 1.f);
        }
        case ZYGPU_PROC_CHECKER: return checkerEvaluate(scene, p, mode, rs)[0];
        case ZYGPU_PROC_MAX: {
            const float a = texture1(scene, p.child[0], p.child_uniform[0], rs), b = texture1(scene, p.child[1], p.child_uniform[1], rs);
            return mx(a, b);
        }
        case ZYGPU_PROC_MIX: {
            const float weight = texture1(scene, p.child[2], p.child_uniform[2], rs);
            const float r = rs.sampler->sample1D();
            return weight < r ? texture1(scene, p.child[0], p.child_uniform[0], rs) : texture1(scene, p.child[1], p.child_uniform[1], rs);
        }
        case ZYGPU_PROC_MUL: {
            const float a = texture1(scene, p.child[0], p.child_uniform[0], rs), b = texture1(scene, p.child[1], p.child_uniform[1], rs);
            return a * b;
        }
        case ZYGPU_PROC_NOISE: return noiseEvaluate1(p, mode, rs, S(0.f));
        default: return 0.f;  // DetailNormal
    }
}

inline void procedural2(const ZygpuSceneDesc& scene, uint32_t index, uint32_t mode, const Renderstate& rs, float out[2]) {
    const ZygpuProcedural& p = scene.procedurals[index];
    float                  a[2], b[2];
    switch (p.kind) {
        case ZYGPU_PROC_CHANNEL_MIX: {
            const V4 c = texture3(scene, p.child[0], p.child_uniform[0], rs);
            out[0] = clampf(dot3(c, ld3(p.value[0])), 0.f, 1.f);
            out[1] = clampf(dot3(c, ld3(p.value[1])), 0.f, 1.f);
            return;
        }
        case ZYGPU_PROC_CHECKER: {
            const V4 c = checkerEvaluate(scene, p, mode, rs);
            out[0] = c[0], out[1] = c[1];
            return;
        }
        case ZYGPU_PROC_DETAIL_NORMAL: {  // procedural_detail_normal.zig:19-29
            const V4 n1 = detailReconstruct(scene, p.child[0], p.child_uniform[0], rs);
            const V4 n2 = detailReconstruct(scene, p.child[1], p.child_uniform[1], rs);
            V4       x, y;
            orthonormalBasis3(n1, x, y);
            const V4 r = S(n2[0]) * x + S(n2[1]) * y + S(n2[2]) * n1;
            out[0] = r[0], out[1] = r[1];
            return;
        }
        case ZYGPU_PROC_MAX:
            texture2(scene, p.child[0], p.child_uniform[0], rs, a);
            texture2(scene, p.child[1], p.child_uniform[1], rs, b);
            out[0] = mx(a[0], b[0]), out[1] = mx(a[1], b[1]);
            return;
        case ZYGPU_PROC_MIX: {
            const float weight = texture1(scene, p.child[2], p.child_uniform[2], rs);
            const float r = rs.sampler->sample1D();
            if (weight < r) texture2(scene, p.child[0], p.child_uniform[0], rs, out);
            else texture2(scene, p.child[1], p.child_uniform[1], rs, out);
            return;
        }
        case ZYGPU_PROC_MUL:
            texture2(scene, p.child[0], p.child_uniform[0], rs, a);
            texture2(scene, p.child[1], p.child_uniform[1], rs, b);
            out[0] = a[0] * b[0], out[1] = a[1] * b[1];
            return;
        default: noiseNormalmap(scene, p, mode, rs, out); return;
    }
}

inline V4 procedural3(const ZygpuSceneDesc& scene, uint32_t index, uint32_t mode, const Renderstate& rs) {
    const ZygpuProcedural& p = scene.procedurals[index];
    switch (p.kind) {
        case ZYGPU_PROC_CHANNEL_MIX: {
            const V4 c = texture3(scene, p.child[0], p.child_uniform[0], rs);
            return mk(clampf(dot3(c, ld3(p.value[0])), 0.f, 1.f), clampf(dot3(c, ld3(p.value[1])), 0.f, 1.f), clampf(dot3(c, ld3(p.value[2])), 0.f, 1.f));
        }
        case ZYGPU_PROC_CHECKER: return checkerEvaluate(scene, p, mode, rs);
        case ZYGPU_PROC_MAX: {
            const V4 a = texture3(scene, p.child[0], p.child_uniform[0], rs), b = texture3(scene, p.child[1], p.child_uniform[1], rs);
            return max4(a, b);
        }
        case ZYGPU_PROC_MIX: {
            const float weight = texture1(scene, p.child[2], p.child_uniform[2], rs);
            const float r = rs.sampler->sample1D();
            return weight < r ? texture3(scene, p.child[0], p.child_uniform[0], rs) : texture3(scene, p.child[1], p.child_uniform[1], rs);
        }
        case ZYGPU_PROC_MUL: {
            const V4 a = texture3(scene, p.child[0], p.child_uniform[0], rs), b = texture3(scene, p.child[1], p.child_uniform[1], rs);
            return a * b;
        }
        case ZYGPU_PROC_NOISE: return S(noiseEvaluate1(p, mode, rs, S(0.f)));
        default: return S(0.f);  // DetailNormal
    }
}

// hlp.sampleNormal (material_helper.zig:16-78), tex_coord UV0: the normal map's xy in the tangent frame, z reconstructed, then the
// "iray" adaption that keeps the reflection of wo above the geometric surface
inline V4 sampleNormal(const ZygpuSceneDesc& scene, const ZygpuTexture& map, V4 wo, const Renderstate& rs) {
    float nmxy[2];
    texture2(scene, map, nullptr, rs, nmxy);
    const float nmz = std::sqrt(mx(1.f - (nmxy[0] * nmxy[0] + nmxy[1] * nmxy[1]), 0.01f));
    const V4    nm = mk(nmxy[0], nmxy[1], nmz);
    V4          n;
    if (map.sampler & ZYGPU_TEX_OBJECTPOS) {  // material_helper.zig:24-29
        n = normalize3(Frame::init(rs.n).frameToWorld(nm));
    } else if (map.sampler & ZYGPU_TEX_TRIPLANAR) {  // material_helper.zig:30-38, triplanarTangent 125-139
        const V4    on = rs.trafo.worldToObjectNormal(rs.n);
        const float an[3] = {std::fabs(on[0]), std::fabs(on[1]), std::fabs(on[2])};
        V4          wt;
        if (an[0] > an[1] && an[0] > an[2]) wt = S(std::copysign(1.f, on[0])) * rs.trafo.r[2];
        else if (an[1] > an[0] && an[1] > an[2]) wt = S(std::copysign(1.f, on[1])) * rs.trafo.r[0];
        else wt = S(-std::copysign(1.f, on[2])) * rs.trafo.r[0];
        wt[3] = 0.f;
        const V4    bt = cross3(rs.n, wt);
        const V4    nt = cross3(bt, rs.n);
        const Frame frame{nt, bt, rs.n};
        n = normalize3(frame.frameToWorld(nm));
    } else {
        // rs.tangentToWorld (renderstate.zig:51-58)
        n = normalize3(mk(nmxy[0] * rs.t[0] + nmxy[1] * rs.b[0] + nmz * rs.n[0], nmxy[0] * rs.t[1] + nmxy[1] * rs.b[1] + nmz * rs.n[1],
                          nmxy[0] * rs.t[2] + nmxy[1] * rs.b[2] + nmz * rs.n[2]));
    }
    const V4    ng = rs.geo_n;
    const V4    r = reflect3(n, wo);
    const float a = dot3(ng, r);
    if (a >= 0.f) return n;
    const float cos_threshold = 0.0017453f;  // cos(89.9 degrees)
    if (dot3(ng, wo) < cos_threshold) return ng;
    const float b = dot3(ng, n);
    const float epsilon = 1e-4f;
    V4          tangent;
    if (b > epsilon) {
        const float distance_to_surface_along_normal = std::fabs(a) / b;
        tangent = normalize3(r + S(distance_to_surface_along_normal) * n);
    } else {
        tangent = n;
    }
    tangent += S(epsilon) * ng;
    return normalize3(wo + tangent);
}

// Material.sampleFlake (substitute_material.zig:235-273): four flakes per cell of a res x res grid, each with its own PCG stream
// (base/random/generator.zig:49-93 SingleGenerator: increment 1); the nearest flake centre owns the point, and it exists with
// probability `coverage`.  Returns the flake's 2-D number (which fixes its orientation).
inline bool sampleFlake(const float uv[2], float res, float coverage, float xi_out[2]) {
    const int32_t i = int32_t(res * (uv[0] - std::floor(uv[0]))), j = int32_t(res * (uv[1] - std::floor(uv[1])));  // gridCell: @mod(uv, 1)
    const float   suv[2] = {res * uv[0], res * uv[1]};
    float         nearest_d = std::numeric_limits<float>::max(), nearest_r = 0.f, nearest_xi[2] = {0.f, 0.f};
    for (int32_t ii = i - 1; ii <= i + 1; ++ii) {
        for (int32_t jj = j - 1; jj <= j + 1; ++jj) {
            uint64_t state = 0;  // SingleGenerator.start(fuse(ii, jj))
            auto     advance = [&]() {
                const uint64_t old = state;
                state = old * 6364136223846793005ull + 1ull;
                const uint32_t xrs = uint32_t(((old >> 18) ^ old) >> 27);
                const uint32_t rot = uint32_t(old >> 59);
                return (xrs >> rot) | (xrs << ((0u - rot) & 31u));
            };
            auto randomFloat = [&]() {
                uint32_t bits = (advance() & 0x007FFFFFu) | 0x3F800000u;
                float    f;
                std::memcpy(&f, &bits, 4);
                return f - 1.f;
            };
            advance();
            state += (uint64_t(uint32_t(ii)) << 32) | uint64_t(uint32_t(jj));
            advance();
            for (uint32_t fl = 0; fl < 4; ++fl) {
                const float px = float(ii) + randomFloat(), py = float(jj) + randomFloat();
                const float x0 = randomFloat(), x1 = randomFloat();
                const float r = randomFloat();
                const float dx = suv[0] - px, dy = suv[1] - py;
                const float vcd = dx * dx + dy * dy;  // squaredLength2
                if (vcd < nearest_d) {
                    nearest_d = vcd;
                    nearest_r = r;
                    nearest_xi[0] = x0;
                    nearest_xi[1] = x1;
                }
            }
        }
    }
    if (nearest_r < coverage) {
        xi_out[0] = nearest_xi[0];
        xi_out[1] = nearest_xi[1];
        return true;
    }
    return false;
}

// Material.sample (material.zig:184-194) for the kinds in scope
inline MaterialSample materialSample(const ZygpuMaterial& m, const ZygpuSceneDesc& scene, V4 wo, const Renderstate& rs) {
    MaterialSample s{};
    s.kind = m.kind;
    s.geo_n = rs.geo_n;
    s.n = rs.n;
    s.wo = wo;
    s.translucent = false;
    s.avoid_caustics = !rs.caustics;
    s.luts.l = &scene.luts;
    if (ZYGPU_MAT_DEBUG == m.kind) {
        // debug_material.zig:20-28 + Base.initTBN (sample_base.zig:49-62)
        const V4   n = cross3(rs.t, rs.b);
        const bool same_side = dot3(n, rs.n) > 0.f;
        s.albedo = same_side ? ld3(m.color_a) : ld3(m.color_b);
        s.frame = {rs.t, rs.b, rs.n};
        s.alpha[0] = s.alpha[1] = 1.f;
        s.can_evaluate = true;
    } else if (ZYGPU_MAT_LIGHT == m.kind) {
        s.frame = {rs.t, rs.b, rs.n};  // light_sample.zig:9-11
        s.alpha[0] = s.alpha[1] = 0.f;
        s.can_evaluate = false;
    } else if (ZYGPU_MAT_GLASS == m.kind) {
        // Glass.sample / Sample.init (glass_material.zig:49-79, glass_sample.zig:33-66)
        auto weight = [&](uint32_t slot, float uniform) { return texture1(scene, m.tex[slot], &uniform, rs); };  // ts.sample2D_1
        const float raw_r = weight(ZYGPU_STEX_ROUGHNESS, m.roughness);
        const float r = 0.f == raw_r ? 0.f : ggx::clampRoughness(raw_r);
        const float glass_specular = weight(ZYGPU_STEX_SPECULAR, m.specular);  // (before the normal map: a Mix draws in this order)
        if (kNull != m.tex[ZYGPU_STEX_NORMAL].image) s.frame = Frame::init(sampleNormal(scene, m.tex[ZYGPU_STEX_NORMAL], wo, rs));
        else s.frame = {rs.t, rs.b, rs.n};
        s.alpha[0] = s.alpha[1] = r * r;
        rs.regularizeAlpha(s.alpha, scene.specular_threshold);
        const bool rough = s.alpha[0] > 0.f;
        s.can_evaluate = rough && m.ior != rs.ior;
        s.absorption = ld3(m.absorption);
        s.ior = m.ior;
        s.ior_outside = rs.ior;
        s.glass_f0 = rough ? fresnel::iorToF0(m.ior, rs.ior) : 0.f;
        s.thickness = m.thickness;
        s.abbe = m.abbe;
        s.wavelength = rs.wavelength;
        s.translucent = m.thickness > 0.f;
        s.specular_threshold = scene.specular_threshold;
        s.specular = glass_specular;
        s.lower_priority = m.priority < rs.highest_priority;  // sample_base.zig:44
    } else {
        // Substitute.sample (substitute_material.zig:114-221) without the volumetric branches
        auto weight = [&](uint32_t slot, float uniform) { return texture1(scene, m.tex[slot], &uniform, rs); };  // ts.sample2D_1
        const ZygpuTexture color_map{m.color_image, m.color_sampler, {m.color_scale[0], m.color_scale[1]}};
        const V4           color = texture3(scene, color_map, m.color_a, rs);
        const float roughness = ggx::clampRoughness(weight(ZYGPU_STEX_ROUGHNESS, m.roughness));
        const float metallic = weight(ZYGPU_STEX_METALLIC, m.metallic);
        const float specular = weight(ZYGPU_STEX_SPECULAR, m.specular);
        float       alpha[2];
        if (m.anisotropy > 0.f) {  // anisotropicAlpha, substitute_material.zig:298-305
            const float rv = ggx::clampRoughness(roughness * (1.f - m.anisotropy));
            alpha[0] = roughness * roughness;
            alpha[1] = rv * rv;
        } else {
            alpha[0] = alpha[1] = roughness * roughness;
        }
        const float coating_scale = weight(ZYGPU_STEX_COATING_SCALE, m.coating_scale);
        const float coating_thickness = coating_scale * m.coating_thickness;
        const float coating_weight = coating_scale > 0.1f ? 1.f : coating_scale;
        const float coating_ior = lerpf(rs.ior, m.coating_ior, coating_weight);
        const float ior = m.ior, ior_outer = coating_thickness > 0.f ? coating_ior : rs.ior, ior_medium = rs.ior;
        const float translucency = weight(ZYGPU_STEX_TRANSLUCENCY, m.translucency);
        // Surface.init (substitute_sample.zig:38-86)
        rs.regularizeAlpha(alpha, scene.specular_threshold);
        s.albedo = S(1.f - metallic) * color;
        s.alpha[0] = alpha[0];
        s.alpha[1] = alpha[1];
        s.can_evaluate = ior != ior_medium;
        s.translucent = translucency > 0.f;
        const float f0 = fresnel::iorToF0(ior, ior_outer);
        s.f0 = lerp4(S(f0), color, S(metallic));
        s.metallic = metallic;
        s.specular = specular;
        s.specular_threshold = scene.specular_threshold;
        s.opacity = 1.f - (0.5f * translucency);
        if (kNull != m.tex[ZYGPU_STEX_NORMAL].image) s.frame = Frame::init(sampleNormal(scene, m.tex[ZYGPU_STEX_NORMAL], wo, rs));
        else s.frame = {rs.t, rs.b, rs.n};
        if (coating_thickness > 0.f) {  // substitute_material.zig:164-181
            const ZygpuTexture &nm = m.tex[ZYGPU_STEX_NORMAL], &cnm = m.tex[ZYGPU_STEX_COATING_NORMAL];
            // Texture.equal: the same description (two uniform maps are equal, too)
            const bool same = nm.image == cnm.image && (kNull == nm.image || (nm.sampler == cnm.sampler && nm.scale[0] == cnm.scale[0] && nm.scale[1] == cnm.scale[1]));
            if (same) s.coating.n = s.frame.z;
            else if (kNull != cnm.image) s.coating.n = sampleNormal(scene, cnm, wo, rs);
            else s.coating.n = rs.n;
            const float r = ggx::clampRoughness(weight(ZYGPU_STEX_COATING_ROUGHNESS, m.coating_roughness));
            s.coating.absorption_coef = ld3(m.coating_absorption);
            s.coating.thickness = coating_thickness;
            s.coating.f0 = fresnel::iorToF0(coating_ior, rs.ior);
            float ca[2] = {r * r, r * r};
            rs.regularizeAlpha(ca, scene.specular_threshold);
            s.coating.alpha = ca[0];
            s.coating.weight = coating_weight;
        }
        // the rotation applies to the base frame only, after the coating took its normal
        const float rotation = weight(ZYGPU_STEX_ROTATION, m.rotation) * (2.f * kPi);
        if (rotation > 0.f) {  // Frame.rotateTangenFrame (frame.zig:64-73)
            const V4 t = s.frame.x, b = s.frame.y;
            const V4 sin_a = S(sinF(rotation)), cos_a = S(cosF(rotation));
            s.frame.x = cos_a * t + sin_a * b;
            s.frame.y = -sin_a * t + cos_a * b;
        }
        const float flakes_coverage = weight(ZYGPU_STEX_FLAKES_COVERAGE, m.flakes_coverage);
        if (flakes_coverage > 0.f) {  // substitute_material.zig:190-215
            const V4 op = rs.trafo.worldToObjectNormal(rs.p - rs.trafo.position);
            const V4 on = rs.trafo.worldToObjectNormal(s.frame.z);
            float    uv[2];
            triplanarMapping(op, on, uv);
            uv[0] -= std::floor(uv[0]);  // math.frac
            uv[1] -= std::floor(uv[1]);
            float xi[2];
            if (sampleFlake(uv, m.flakes_res, flakes_coverage, xi)) {
                const float fa = m.flakes_alpha;
                // flakesA2cone (substitute_material.zig:222-227): the cone of 7 degrees bounds the flake's lobe
                const float target_angle = solidAngleOfCone(0.992546151641322f);  // cos(7 degrees)
                const float limit = target_angle / ((4.f * kPi) - target_angle);
                const float a2_cone = mn(limit, 0.5f * fa);
                const float fa2 = fa - a2_cone;
                const float cos_cone = 1.f - (2.f * a2_cone) / (1.f + a2_cone);
                const float a[2] = {fa2, fa2};
                float       n_dot_h;
                const V4    mh = ggx::anisoSample(wo, a, xi[0], xi[1], s.frame, n_dot_h);
                s.metallic = 1.f;
                s.f0 = ld3(m.flakes_color);
                s.alpha[0] = cos_cone;
                s.alpha[1] = a2_cone;
                s.frame = Frame::init(mh);
                s.flakes = true;
            }
        }
    }
    return s;
}

}  // namespace orc
