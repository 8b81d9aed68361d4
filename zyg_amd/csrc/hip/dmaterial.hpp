// Device material closures (uniform textures).  Behaviour follows the reference:
//   Debug (Lambert)          material/debug/debug_material.zig:20-28, debug_sample.zig:15-47
//   Light                    material/light/light_sample.zig:7-13
//   Substitute (opaque base) material/substitute/substitute_material.zig:114-221, substitute_sample.zig:38-410
//   diffuse.Micro            material/diffuse.zig:42-110;  ggx: material/ggx.zig:30-46,60-130,228-449
//   fresnel.Schlick          material/fresnel.zig:5-29;   LUT eval: base/math/interpolated_function.zig:105-271
#pragma once

#include "dlight.hpp"

namespace zd {

enum : uint32_t { kScatterDiffuse = 0, kScatterGlossy = 1, kScatterSpecular = 2, kScatterNone = 3 };

struct BxdfResult {
    f3    reflection;
    float pdf;
};

struct BxdfSample {
    f3       reflection, wi;
    float    pdf;
    float    reg_alpha;   // bxdf.zig Path.reg_alpha
    uint32_t scattering;
    // set by Glass only (every other sample in scope is a reflection with split weight 1)
    bool     transmission;
    bool     straight;  // bxdf.Path.straight: thin glass passes the ray on unbent (no scattering event for the vertex state)
    float    split_weight;
    float    wavelength;  // set by Glass only: the wavelength a dispersive glass settled for the path (0: none)
};

// The spectrum a dispersive Glass weights its wavelength draw with: rainbow_integral.zig:3-41 (a generated table: 32 bands
// between 380 and 720 nm; tests/golden/reference_constants.json holds the reference's numbers), material_base.zig:56-73
constexpr float kRainbowStart = 380.f, kRainbowEnd = 720.f;
constexpr int   kRainbowBands = 32;
static __device__ const float kRainbow[kRainbowBands][3] = {
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
ZD f3 spectrumAtWavelength(float lambda, float value) {
    const float    u = ((lambda - kRainbowStart) / (kRainbowEnd - kRainbowStart)) * float(kRainbowBands);
    const uint32_t id = uint32_t(u);
    const float    frac = u - float(id);
    const uint32_t a = min(id, uint32_t(kRainbowBands - 1)), b = min(id + 1, uint32_t(kRainbowBands - 1));
    const f3       ca = mk3(kRainbow[a][0], kRainbow[a][1], kRainbow[a][2]), cb = mk3(kRainbow[b][0], kRainbow[b][1], kRainbow[b][2]);
    if (id >= uint32_t(kRainbowBands - 1)) return ca;
    return S3(value) * lerp3(ca, cb, S3(frac));
}

ZD float fresnelDielectric(float cos_theta_i, float cos_theta_t, float eta_i, float eta_t) {  // fresnel.zig:31-43
    const float t0 = eta_t * cos_theta_i, t1 = eta_i * cos_theta_t;
    const float r_p = (t0 - t1) / (t0 + t1);
    const float t2 = eta_i * cos_theta_i, t3 = eta_t * cos_theta_t;
    const float r_o = (t2 - t3) / (t2 + t3);
    return 0.5f * (r_p * r_p + r_o * r_o);
}

struct Luts {
    const DScene& s;
    ZD static float bilinear(float c0, float c1, float c2, float c3, float sx, float t) {  // math.zig:172-179
        const float _s = 1.f - sx, _t = 1.f - t;
        return _t * (_s * c0 + sx * c1) + t * (_s * c2 + sx * c3);
    }
    ZD static float eval1(const float* smp, int N, float x) {
        const float    o = mn(x, 1.f) * float(N - 1);
        const uint32_t offset = uint32_t(o);
        const float    t = o - float(offset);
        return lerpf(smp[offset], smp[min(offset + 1, uint32_t(N - 1))], t);
    }
    ZD static float eval2(const float* smp, int X, int Y, float x, float y) {
        const float ox = mn(x, 1.f) * float(X - 1), oy = mn(y, 1.f) * float(Y - 1);
        const int   o0 = int(ox), o1 = int(oy);
        const float t0 = ox - float(o0), t1 = oy - float(o1);
        const int   col1 = min(o0 + 1, X - 1);
        const int   row0 = o1 * X, row1 = min(o1 + 1, Y - 1) * X;
        return bilinear(smp[o0 + row0], smp[col1 + row0], smp[o0 + row1], smp[col1 + row1], t0, t1);
    }
    ZD static float eval3(const float* smp, int X, int Y, int Z, float x, float y, float z) {
        const float ox = mn(x, 1.f) * float(X - 1), oy = mn(y, 1.f) * float(Y - 1), oz = mn(z, 1.f) * float(Z - 1);
        const int   o0 = int(ox), o1 = int(oy), o2 = int(oz);
        const float t0 = ox - float(o0), t1 = oy - float(o1), t2 = oz - float(o2);
        const int   col1 = min(o0 + 1, X - 1);
        const int   row0 = o1 * X, row1 = min(o1 + 1, Y - 1) * X;
        const int   area = X * Y;
        const int   s0 = o2 * area, s1 = min(o2 + 1, Z - 1) * area;
        const float c0 = bilinear(smp[o0 + row0 + s0], smp[col1 + row0 + s0], smp[o0 + row1 + s0], smp[col1 + row1 + s0], t0, t1);
        const float c1 = bilinear(smp[o0 + row0 + s1], smp[col1 + row0 + s1], smp[o0 + row1 + s1], smp[col1 + row1 + s1], t0, t1);
        return lerpf(c0, c1, t2);
    }
    ZD float E_m(float n_dot, float alpha) const { return eval2(s.lut_e_m, 32, 32, n_dot, alpha); }
    ZD float E_m_avg(float alpha) const { return eval1(s.lut_e_m_avg, 32, alpha); }
    ZD float E(float n_dot, float alpha, float f0) const { return eval3(s.lut_e, 16, 16, 16, n_dot, alpha, f0); }
    ZD float E_avg(float alpha, float f0) const { return eval2(s.lut_e_avg, 16, 16, alpha, f0); }
    // ggx.ilmEpDielectric (ggx.zig:30-32): E_s over f0 in [0, 0.25]
    ZD float ilmEpDielectric(float n_dot_wo, float alpha, float f0) const { return 1.f / eval3(s.lut_e_s, 16, 16, 16, n_dot_wo, alpha, f0 * 4.f); }
};

ZD float pow5(float x) {
    const float x2 = x * x, x4 = x2 * x2;
    return x4 * x;
}
ZD f3 schlick(f3 f0, float wo_dot_h) { return fma3(S3(pow5(1.f - wo_dot_h)), S3(1.f) - f0, f0); }

constexpr float kMinRoughness = 0.01314f;  // ggx.zig:14

struct Micro {
    f3    h;
    float n_dot_wi, h_dot_wi;
};

ZD f3 dspbrMicroEc(const Luts& L, f3 f0, float n_dot_wi, float n_dot_wo, float alpha) {  // ggx.zig:34-46
    const float e_wo = L.E_m(n_dot_wo, alpha), e_wi = L.E_m(n_dot_wi, alpha), e_avg = L.E_m_avg(alpha);
    const float m = ((1.f - e_wo) * (1.f - e_wi)) / (kPi * (1.f - e_avg));
    const f3    f_avg = fma3(S3(20.f / 21.f), f0, S3(1.f / 21.f));
    const f3    f = ((f_avg * f_avg) * S3(e_avg)) / fma3(-f_avg, S3(1.f - e_avg), S3(1.f));
    return S3(m) * f;
}

ZD float pdfVisible(float d, float g1_wo) { return (0.5f * d) / g1_wo; }
ZD float isoDistribution(float n_dot_h, float a2) {
    const float d = __fmaf_rn(n_dot_h * n_dot_h, a2 - 1.f, 1.f);
    return a2 / (kPi * d * d);
}
ZD void isoVisibilityAndG1Wo(float n_dot_wi, float n_dot_wo, float alpha2, float& g0, float& g1) {
    const float t_wi = sqrtf(__fmaf_rn(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));
    const float t_wo = sqrtf(__fmaf_rn(1.f - alpha2, n_dot_wo * n_dot_wo, alpha2));
    g0 = 0.5f / (n_dot_wi * t_wo + n_dot_wo * t_wi);
    g1 = t_wo + n_dot_wo;
}
ZD float anisoDistribution(float n_dot_h, float x_dot_h, float y_dot_h, float a0, float a1) {
    const float a20 = a0 * a0, a21 = a1 * a1;
    const float x = (x_dot_h * x_dot_h) / a20, y = (y_dot_h * y_dot_h) / a21;
    const float d = (x + y) + (n_dot_h * n_dot_h);
    return 1.f / (kPi * (a0 * a1) * (d * d));
}
ZD void anisoVisibilityAndG1Wo(float t_dot_wi, float t_dot_wo, float b_dot_wi, float b_dot_wo, float n_dot_wi, float n_dot_wo,
                               float a0, float a1, float& g0, float& g1) {
    const float t_wo = length3(mk3(a0 * t_dot_wo, a1 * b_dot_wo, n_dot_wo));
    const float t_wi = length3(mk3(a0 * t_dot_wi, a1 * b_dot_wi, n_dot_wi));
    g0 = 0.5f / (n_dot_wi * t_wo + n_dot_wo * t_wi);
    g1 = t_wo + n_dot_wo;
}

ZD f3 ggxSample(f3 wo, float a0, float a1, float xi0, float xi1, const Frame& frame, float& n_dot_h) {  // ggx.zig:396-414
    const f3    wo_l = frame.worldToFrame(wo);
    const f3    v = normalize3(mk3(a0 * wo_l.x, a1 * wo_l.y, wo_l.z));
    const float phi = (2.f * kPi) * xi0;
    const float z = __fmaf_rn(1.f - xi1, 1.f + v.z, -v.z);
    const float sin_theta = sqrtf(saturate(1.f - z * z));
    const float x = sin_theta * cosF(phi), y = sin_theta * sinF(phi);
    const f3    h = mk3(x, y, z) + v;
    const f3    m = normalize3(mk3(a0 * h.x, a1 * h.y, h.z));
    n_dot_h = clampf(m.z, kDotMin, 1.f);
    return frame.frameToWorld(m);
}

ZD BxdfResult ggxReflection(f3 wi, f3 wo, f3 h, float n_dot_wi, float n_dot_wo, float wo_dot_h, float a0, float a1, f3 f0,
                            const Frame& frame) {
    float       d, g0, g1;
    const float n_dot_h = saturate(dot3(frame.z, h));
    if (a0 == a1) {
        const float alpha2 = a0 * a0;
        d = isoDistribution(n_dot_h, alpha2);
        isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g0, g1);
    } else {
        d = anisoDistribution(n_dot_h, dot3(frame.x, h), dot3(frame.y, h), a0, a1);
        anisoVisibilityAndG1Wo(dot3(frame.x, wi), dot3(frame.x, wo), dot3(frame.y, wi), dot3(frame.y, wo), n_dot_wi, n_dot_wo, a0, a1,
                               g0, g1);
    }
    const f3 f = schlick(f0, wo_dot_h);
    return {S3(d * g0) * f, pdfVisible(d, g1)};
}

ZD Micro ggxReflect(f3 wo, float n_dot_wo, float a0, float a1, float specular_threshold, float xi0, float xi1, f3 f0,
                    const Frame& frame, BxdfSample& result) {
    float       n_dot_h;
    const f3    h = ggxSample(wo, a0, a1, xi0, xi1, frame, n_dot_h);
    const float wo_dot_h = clampDot(wo, h);
    const f3    wi = normalize3(fma3(S3(2.f * wo_dot_h), h, -wo));
    const float n_dot_wi = frame.clampNdot(wi);
    float       d, g0, g1;
    if (a0 == a1) {
        const float alpha2 = a0 * a0;
        d = isoDistribution(n_dot_h, alpha2);
        isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g0, g1);
    } else {
        d = anisoDistribution(n_dot_h, dot3(frame.x, h), dot3(frame.y, h), a0, a1);
        anisoVisibilityAndG1Wo(dot3(frame.x, wi), dot3(frame.x, wo), dot3(frame.y, wi), dot3(frame.y, wo), n_dot_wi, n_dot_wo, a0, a1,
                               g0, g1);
    }
    const f3 f = schlick(f0, wo_dot_h);
    result.reflection = S3(d * g0) * f;
    result.wi = wi;
    result.pdf = pdfVisible(d, g1);
    result.reg_alpha = a1;
    result.scattering = a1 <= specular_threshold ? kScatterSpecular : kScatterGlossy;
    return {h, n_dot_wi, wo_dot_h};
}

// ---- rough dielectrics: ggx.Iso.reflectNoFresnel / refractNoFresnel / reflectionF / refractionF (ggx.zig:73-95, 127-227)
ZD float gSmithCorrelated(float n_dot_wi, float n_dot_wo, float alpha2) {
    const float a = n_dot_wo * sqrtf(__fmaf_rn(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));
    const float b = n_dot_wi * sqrtf(__fmaf_rn(1.f - alpha2, n_dot_wo * n_dot_wo, alpha2));
    return (2.f * n_dot_wi * n_dot_wo) / (a + b);
}
ZD float pdfVisibleRefract(float n_dot_wo, float wo_dot_h, float d, float alpha2) {
    const float g1 = (2.f * n_dot_wo) / (n_dot_wo + sqrtf(alpha2 + (1.f - alpha2) * (n_dot_wo * n_dot_wo)));
    return g1 * wo_dot_h * d / n_dot_wo;
}
ZD float clampAbs(float x) { return clampf(fabsf(x), kDotMin, 1.f); }
ZD float schlick1(float wo_dot_h, float f0) {  // fresnel.zig:5-7
    const float om = 1.f - wo_dot_h, om2 = om * om;
    return __fmaf_rn((om2 * om2) * om, 1.f - f0, f0);
}
ZD float reflectNoFresnel(f3 wo, f3 h, float n_dot_wo, float n_dot_h, float wo_dot_h, float alpha, float specular_threshold,
                          const Frame& frame, BxdfSample& result) {
    const f3    wi = normalize3(fma3(S3(2.f * wo_dot_h), h, -wo));
    const float n_dot_wi = frame.clampNdot(wi);
    const float alpha2 = alpha * alpha;
    const float d = isoDistribution(n_dot_h, alpha2);
    float       g0, g1;
    isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g0, g1);
    result.reflection = S3(d * g0);
    result.wi = wi;
    result.pdf = pdfVisible(d, g1);
    result.reg_alpha = alpha;
    result.scattering = alpha <= specular_threshold ? kScatterSpecular : kScatterGlossy;
    result.transmission = false;
    result.straight = false;
    return n_dot_wi;
}
ZD float refractNoFresnel(f3 wo, f3 h, float n_dot_wo, float n_dot_h, float wi_dot_h, float wo_dot_h, float alpha, float specular_threshold,
                          float eta_i, float eta_t, const Frame& frame, BxdfSample& result) {
    const float eta = eta_i / eta_t;
    const float abs_wi_dot_h = clampAbs(wi_dot_h), abs_wo_dot_h = clampAbs(wo_dot_h);
    const f3    wi = normalize3(S3(__fmaf_rn(eta, abs_wo_dot_h, -abs_wi_dot_h)) * h - S3(eta) * wo);
    const float n_dot_wi = frame.clampAbsNdot(wi);
    const float alpha2 = alpha * alpha;
    const float d = isoDistribution(n_dot_h, alpha2);
    const float g = gSmithCorrelated(n_dot_wi, n_dot_wo, alpha2);
    const float refr = d * g;
    const float factor = (abs_wi_dot_h * abs_wo_dot_h) / (n_dot_wi * n_dot_wo);
    const float dn = eta_i * wo_dot_h + eta_t * wi_dot_h, denom = dn * dn;
    const float sqr_eta_t = eta_t * eta_t;
    const float pdf = pdfVisibleRefract(n_dot_wo, abs_wo_dot_h, d, alpha2);
    result.reflection = S3((factor * sqr_eta_t / denom) * refr);
    result.wi = wi;
    result.pdf = pdf * (abs_wi_dot_h * sqr_eta_t / denom);
    result.reg_alpha = alpha;
    result.scattering = alpha <= specular_threshold ? kScatterSpecular : kScatterGlossy;
    result.transmission = true;
    result.straight = false;
    return n_dot_wi;
}

struct MaterialSample {
    uint32_t kind;
    Frame    frame;
    f3       geo_n, n, wo;
    float    a0, a1;
    bool     can_evaluate, translucent, avoid_caustics;
    f3       albedo;
    f3       f0;
    float    metallic, specular, specular_threshold, opacity;
    float    ior, ior_outside, glass_f0, thickness;  // Glass
    float    abbe;  // dispersion: the material's Abbe number (the path's wavelength is an argument of glassSample)
    f3       absorption;
    bool     lower_priority;
    // Substitute: clearcoat (substitute_coating.zig:15-113) and flakes (sample_base.zig:17); only the EXT instantiations
    // (the kernels that rebuild the whole fragment) look at them
    struct Coating {
        f3    n, absorption_coef;
        float thickness, f0, alpha, weight;
        struct Result {
            f3    reflection, attenuation;
            float f, pdf;
        };
        ZD f3 singleAttenuation(float n_dot_wo) const { return attenuation3(absorption_coef, thickness * (1.f / n_dot_wo)); }
        ZD f3 attenuation(float n_dot_wi, float n_dot_wo) const {  // :107-113
            const float f = weight * schlick1(mn(n_dot_wi, n_dot_wo), f0);
            const float d = thickness * (1.f / n_dot_wi + 1.f / n_dot_wo);
            return S3(1.f - f) * attenuation3(absorption_coef, d);
        }
        ZD Result evaluate(const Luts& L, f3 wi, f3 wo, f3 h, float wo_dot_h, float specular_threshold, bool avoid_caustics) const {  // :35-56
            const float n_dot_wi = clampDot(n, wi), n_dot_wo = clampAbsDot(n, wo);
            const f3    att = attenuation(n_dot_wi, n_dot_wo);
            if (avoid_caustics && alpha <= specular_threshold) return {S3(0.f), att, 0.f, 0.f};
            // Iso.reflectionF (ggx.zig:74-97)
            const float alpha2 = alpha * alpha;
            const float n_dot_h = saturate(dot3(n, h));
            const float d = isoDistribution(n_dot_h, alpha2);
            float       g0, g1;
            isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g0, g1);
            const f3    f = schlick(S3(f0), wo_dot_h);
            const float ep = L.ilmEpDielectric(n_dot_wo, alpha, f0);
            return {S3(ep * weight * n_dot_wi) * (S3(d * g0) * f), att, f.x, pdfVisible(d, g1)};
        }
        ZD f3 reflect(const Luts& L, f3 wo, f3 h, float n_dot_wo, float n_dot_h, float wo_dot_h, float specular_threshold,
                      BxdfSample& result) const {  // :58-85
            const float n_dot_wi = reflectNoFresnel(wo, h, n_dot_wo, n_dot_h, wo_dot_h, alpha, specular_threshold, Frame::init(n), result);
            const float ep = L.ilmEpDielectric(n_dot_wo, alpha, f0);
            result.reflection = result.reflection * S3(ep * weight * n_dot_wi);
            return attenuation(n_dot_wi, n_dot_wo);
        }
        ZD Micro sample(f3 wo, float xi0, float xi1, float& n_dot_h) const {  // :87-95: n_dot_wi carries the Fresnel term
            const f3    h = ggxSample(wo, alpha, alpha, xi0, xi1, Frame::init(n), n_dot_h);
            const float wo_dot_h = clampDot(wo, h);
            return {h, schlick1(wo_dot_h, f0), wo_dot_h};
        }
    } coating;
    bool flakes;

    // Glass.Sample.evaluate (glass_sample.zig:68-146): rough interfaces only; `split`: max_splits > 1
    ZD BxdfResult glassEvaluate(const DScene& s, f3 wi, bool split) const {
        const float a = a0;
        if (ior == ior_outside || !(a > 0.f) || lower_priority || (avoid_caustics && a <= specular_threshold)) return {S3(0.f), 0.f};
        const Luts L{s};
        if (!sameHemisphere(wo)) {  // thick refraction from inside
            const float eta_i = ior, eta_t = ior_outside;
            const f3    h = -normalize3(S3(eta_t) * wi + S3(eta_i) * wo);
            const float wi_dot_h = dot3(wi, h);
            if (wi_dot_h <= 0.f) return {S3(0.f), 0.f};
            const float wo_dot_h = dot3(wo, h);
            const float eta = eta_i / eta_t;
            const float sint2 = (eta * eta) * (1.f - wo_dot_h * wo_dot_h);
            if (sint2 >= 1.f) return {S3(0.f), 0.f};
            const float n_dot_wi = frame.clampNdot(wi), n_dot_wo = frame.clampAbsNdot(wo);
            const float n_dot_h = saturate(dot3(frame.z, h));
            // Iso.refractionF
            const float alpha2 = a * a;
            const float abs_wi_dot_h = clampAbs(wi_dot_h), abs_wo_dot_h = clampAbs(wo_dot_h);
            const float d = isoDistribution(n_dot_h, alpha2);
            const float g = gSmithCorrelated(n_dot_wi, n_dot_wo, alpha2);
            const float cos_x = eta_i > eta_t ? abs_wi_dot_h : abs_wo_dot_h;
            const float f = 1.f - schlick(S3(glass_f0), cos_x).x;
            const float sqr_eta_t = eta_t * eta_t;
            const float factor = (abs_wi_dot_h * abs_wo_dot_h) / (n_dot_wi * n_dot_wo);
            const float dn = eta_i * wo_dot_h + eta_t * wi_dot_h, denom = dn * dn;
            const float refr = d * g * f;
            const float refl = (factor * sqr_eta_t / denom) * refr;
            const float pdf = pdfVisibleRefract(n_dot_wo, abs_wo_dot_h, d, alpha2) * (abs_wi_dot_h * sqr_eta_t / denom);
            const float comp = L.ilmEpDielectric(n_dot_wo, a, glass_f0);
            const float split_pdf = split ? 1.f : f;
            return {S3(mn(n_dot_wi, n_dot_wo) * comp * specular) * S3(refl), split_pdf * pdf};
        }
        if (sameHemisphere(wi)) {
            const float n_dot_wi = frame.clampNdot(wi), n_dot_wo = frame.clampAbsNdot(wo);
            const f3    h = normalize3(wo + wi);
            const float wo_dot_h = clampDot(wo, h);
            // Iso.reflectionF
            const float alpha2 = a * a;
            const float n_dot_h = saturate(dot3(frame.z, h));
            const float d = isoDistribution(n_dot_h, alpha2);
            float       g0, g1;
            isoVisibilityAndG1Wo(n_dot_wi, n_dot_wo, alpha2, g0, g1);
            const f3    f = schlick(S3(glass_f0), wo_dot_h);
            const float comp = L.ilmEpDielectric(n_dot_wo, a, glass_f0);
            const float split_pdf = split ? 1.f : f.x;
            return {S3(n_dot_wi * comp * specular) * (S3(d * g0) * f), split_pdf * pdfVisible(d, g1)};
        }
        return {S3(0.f), 0.f};
    }

    // Glass.roughSample, thick (glass_sample.zig:294-422)
    // weight / ior_t / wl: the dispersion draw of Glass.sample (1, the material's ior and 0 without dispersion)
    ZD uint32_t glassRoughSample(const DScene& s, Sampler& sampler, bool split, BxdfSample& r0, BxdfSample& r1, f3 weight, float ior_t,
                                 float wl) const {
        float eta_i = ior_outside, eta_t = ior_t;
        r0.wavelength = r1.wavelength = wl;
        if (fabsf(eta_i - eta_t) <= 2.e-7f || lower_priority) {
            r0.reflection = weight;
            r0.wi = -wo;
            r0.pdf = 1.f;
            r0.reg_alpha = 0.f;
            r0.scattering = kScatterSpecular;
            r0.transmission = true;
            r0.straight = false;
            r0.split_weight = 1.f;
            return 1;
        }
        const Luts  L{s};
        const float a = a0;
        const bool  same_side = sameHemisphere(wo);
        Frame       fr = frame;
        if (!same_side) {
            fr.z = -fr.z;
            const float e = eta_i;
            eta_i = eta_t;
            eta_t = e;
        }
        float p, xi0, xi1;
        sampler.sample3D(p, xi0, xi1);
        float       n_dot_h;
        const f3    h = ggxSample(wo, a0, a1, xi0, xi1, fr, n_dot_h);
        const float n_dot_wo = fr.clampAbsNdot(wo);
        const float wo_dot_h = clampDot(wo, h);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - wo_dot_h * wo_dot_h);
        float       wi_dot_h, f;
        if (sint2 >= 1.f) {
            wi_dot_h = 0.f;
            f = 1.f;
        } else {
            wi_dot_h = sqrtf(1.f - sint2);
            f = schlick1(eta_i > eta_t ? wi_dot_h : wo_dot_h, glass_f0);
        }
        const float ep = L.ilmEpDielectric(n_dot_wo, a, glass_f0);
        const float r_wo_dot_h = same_side ? -wo_dot_h : wo_dot_h;
        // roughRefract (glass_sample.zig:437-503): thick = refraction through the micro-facet; thin (a slab without volume) =
        // the straight-through direction scattered by the roughness (a reflection about the facet re-expressed in a frame around
        // -wo), attenuated over the slab, event .straight; < 0: no sample
        auto roughRefract = [&](BxdfSample& result) {
            if (thickness > 0.f) {
                const Frame thin_frame = Frame::init(-wo);
                const f3    tangent_h = fr.worldToFrame(h);
                const f3    thin_h = thin_frame.frameToWorld(tangent_h);
                const float n_dot_wi = reflectNoFresnel(thin_h, thin_h, tangent_h.z, saturate(tangent_h.z), 1.f, a, specular_threshold, thin_frame, result);
                if (sameHemisphere(result.wi)) return -1.f;
                result.reflection = result.reflection * attenuation3(absorption, thickness / n_dot_wo);
                result.reg_alpha = 1.f;  // bxdf.Path.straight (bxdf.zig:34-70): no scattering class, the vertex state stays
                result.scattering = kScatterNone;
                result.transmission = false;
                result.straight = true;
                return n_dot_wi;
            }
            return refractNoFresnel(wo, h, n_dot_wo, n_dot_h, -wi_dot_h, r_wo_dot_h, a, specular_threshold, eta_i, eta_t, fr, result);
        };
        if (split) {
            const float n_dot_wi = reflectNoFresnel(wo, h, n_dot_wo, n_dot_h, wo_dot_h, a, specular_threshold, fr, r0);
            r0.reflection = r0.reflection * (S3(n_dot_wi * ep * specular) * weight);
            r0.split_weight = f;
            if (1.f == f) return 1;
            const float n_dot_wt = roughRefract(r1);
            if (n_dot_wt < 0.f) return 1;
            r1.reflection = r1.reflection * (S3(n_dot_wt * ep) * weight);
            r1.split_weight = 1.f - f;
            return 2;
        }
        if (p <= f) {
            const float n_dot_wi = reflectNoFresnel(wo, h, n_dot_wo, n_dot_h, wo_dot_h, a, specular_threshold, fr, r0);
            r0.reflection = r0.reflection * (S3(f * n_dot_wi * ep * specular) * weight);
            r0.pdf *= f;
        } else {
            const float n_dot_wt = roughRefract(r0);
            if (n_dot_wt < 0.f) return 0;
            const float omf = 1.f - f;
            r0.reflection = r0.reflection * (S3(omf * n_dot_wt * ep) * weight);
            r0.pdf *= omf;
        }
        r0.split_weight = 1.f;
        return 1;
    }

    ZD bool sameHemisphere(f3 v) const { return dot3(geo_n, v) > 0.f; }

    // Glass.specularSample, thick (glass_sample.zig:165-254, reflect 425-435, thickSpecularRefract 519-537); `split`: both
    // directions, weighted by the Fresnel term, instead of a choice between them.  Returns the number of samples.
    // path_wavelength: rs.wavelength, the wavelength the path already has (0: none yet)
    ZD uint32_t glassSample(const DScene& s, Sampler& sampler, bool split, BxdfSample& r0, BxdfSample& r1, float path_wavelength = 0.f) const {
        // Glass.sample (glass_sample.zig:165-199): a thick dispersive glass (abbe != 0) first settles the path's wavelength (a
        // uniform draw over the visible range, weighted by the Rainbow spectrum x 3; a path that has one keeps it, weight 1) and
        // takes its index of refraction at that wavelength (Cauchy's equation through the Abbe number)
        f3    weight = S3(1.f);
        float ior_t = ior, wl = 0.f;
        if (!(thickness > 0.f) && 0.f != abbe) {
            wl = path_wavelength;
            const float r1d = sampler.sample1D();
            if (0.f == wl) {  // wavelengthSpectrumWeight (glass_sample.zig:154-163)
                wl = kRainbowStart + (kRainbowEnd - kRainbowStart) * r1d;
                weight = spectrumAtWavelength(wl, 1.f) * S3(3.f);
            }
            const float sqr_wl = wl * wl;
            ior_t = ior + ((ior - 1.f) / abbe) * (523655.f / sqr_wl - 1.5168f);
        }
        if (a0 > 0.f) return glassRoughSample(s, sampler, split, r0, r1, weight, ior_t, wl);
        float eta_i = ior_outside, eta_t = ior_t;
        auto  fill = [&](BxdfSample& r, f3 wi, f3 reflection, float split_weight, bool transmission, bool straight = false) {
            r.wavelength = straight ? 0.f : wl;  // thinSpecularRefract: 0 (glass_sample.zig:514)
            r.reflection = reflection;
            r.wi = wi;
            r.pdf = 1.f;
            r.reg_alpha = straight ? 1.f : 0.f;
            r.scattering = straight ? kScatterNone : kScatterSpecular;
            r.transmission = transmission;
            r.straight = straight;
            r.split_weight = split_weight;
        };
        if (eta_i == eta_t || lower_priority) {
            fill(r0, -wo, weight, 1.f, true);
            return 1;
        }
        const bool thin = thickness > 0.f;
        f3         nn = frame.z;
        if (!thin && !sameHemisphere(wo)) {  // thin material is always double sided (glass_sample.zig:203-209)
            nn = -nn;
            const float e = eta_i;
            eta_i = eta_t;
            eta_t = e;
        }
        const float n_dot_wo = mn(fabsf(dot3(nn, wo)), 1.f);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - n_dot_wo * n_dot_wo);
        float       n_dot_t, f;
        if (sint2 >= 1.f) {
            n_dot_t = 0.f;
            f = 1.f;
        } else {
            n_dot_t = sqrtf(1.f - sint2);
            f = fresnelDielectric(n_dot_wo, n_dot_t, eta_i, eta_t);
        }
        const f3 reflected = normalize3(S3(2.f * n_dot_wo) * nn - wo);
        const f3 refracted = normalize3(S3(eta * n_dot_wo - n_dot_t) * nn - S3(eta) * wo);
        // thinSpecularRefract (glass_sample.zig:505-517): straight through, attenuated over the slab
        const f3 thin_attenuation = thin ? attenuation3(absorption, thickness / clampf(n_dot_wo, kDotMin, 1.f)) : S3(1.f);
        if (split) {
            fill(r0, reflected, S3(specular) * weight, f, false);
            if (1.f == f) return 1;
            if (thin) fill(r1, -wo, thin_attenuation, 1.f - f, false, true);
            else fill(r1, refracted, weight, 1.f - f, true);
            return 2;
        }
        const float p = sampler.sample1D();
        if (p <= f) fill(r0, reflected, S3(specular) * weight, 1.f, false);
        else if (thin) fill(r0, -wo, thin_attenuation, 1.f, false, true);
        else fill(r0, refracted, weight, 1.f, true);
        return 1;
    }

    ZD float diffuseWeight(const Luts& L) const {  // diffuse.zig:50-73 estimateContribution
        const float e_avg = L.E_avg(a1, hmax3(f0));
        const float a = e_avg;
        const float b = 1.f / (kPi * (1.f - e_avg)) * hmax3(albedo);
        return b / (a + b);
    }
    ZD f3 microEvaluate(const Luts& L, f3 color, float n_dot_wi, float n_dot_wo, float f0m) const {  // diffuse.zig:103-110
        const float e_wo = L.E(n_dot_wo, a1, f0m), e_wi = L.E(n_dot_wi, a1, f0m), e_avg = L.E_avg(a1, f0m);
        return S3(((1.f - e_wo) * (1.f - e_wi)) / (kPi * (1.f - e_avg))) * color;
    }

    template <bool EXT = false>
    ZD BxdfResult baseEvaluate(const Luts& L, f3 wi, f3 h, float wo_dot_h, bool force_disable_caustics) const {
        const float n_dot_wi = frame.clampNdot(wi);
        const float n_dot_wo = frame.clampAbsNdot(wo);
        BxdfResult  d{S3(0.f), 0.f};
        float       dw = 0.f;
        if (1.f != metallic) {
            d.reflection = microEvaluate(L, S3(opacity) * albedo, n_dot_wi, n_dot_wo, hmax3(f0));
            d.pdf = n_dot_wi * kPiInv;
            dw = diffuseWeight(L);
        }
        if ((force_disable_caustics || avoid_caustics) && a1 <= specular_threshold) return {S3(n_dot_wi) * d.reflection, dw * d.pdf};
        if (EXT && flakes) {  // substitute_sample.zig:265-271: a flake reflects into a cone around the mirror direction
            const float cos_cone = a0;
            const f3    r = reflect3(frame.z, wo);
            const float f = dot3(wi, r) > cos_cone ? 1.f / solidAngleOfCone(cos_cone) : 0.f;
            return {S3(n_dot_wi * f * specular) * f0, f};
        }
        const BxdfResult gg = ggxReflection(wi, wo, h, n_dot_wi, n_dot_wo, wo_dot_h, a0, a1, f0, frame);
        const f3         mms = dspbrMicroEc(L, f0, n_dot_wi, n_dot_wo, a1);
        const float      pdf = dw * d.pdf + (1.f - dw) * gg.pdf;
        return {S3(n_dot_wi) * (d.reflection + S3(specular) * (gg.reflection + mms)), pdf};
    }

    ZD static float diffuseFresnelHack(float n_dot_wi, float n_dot_wo, float f0_) {  // substitute_sample.zig:440-442
        return schlick1(mx(n_dot_wi, n_dot_wo), f0_);
    }

    // substitute_sample.zig:338-361
    ZD Micro diffuseSample(const Luts& L, float dw, float xi0, float xi1, BxdfSample& result) const {
        const float n_dot_wo = frame.clampAbsNdot(wo);
        const f3    is = hemisphereCosine(xi0, xi1);
        const f3    wi = normalize3(frame.frameToWorld(is));
        const f3    h = normalize3(wo + wi);
        const float h_dot_wi = clampDot(h, wi);
        const float n_dot_wi = frame.clampNdot(wi);
        const f3    refl = microEvaluate(L, S3(opacity) * albedo, n_dot_wi, n_dot_wo, hmax3(f0));
        const float dpdf = n_dot_wi * kPiInv;
        const BxdfResult gg = ggxReflection(wi, wo, h, n_dot_wi, n_dot_wo, h_dot_wi, a0, a1, f0, frame);
        const f3         mms = dspbrMicroEc(L, f0, n_dot_wi, frame.clampNdot(wo), a1);
        result.reflection = S3(n_dot_wi) * (refl + S3(specular) * (gg.reflection + mms));
        result.wi = wi;
        result.pdf = dw * dpdf + (1.f - dw) * gg.pdf;
        result.reg_alpha = 1.f;
        result.scattering = kScatterDiffuse;
        return {h, n_dot_wi, h_dot_wi};
    }

    // substitute_sample.zig:363-410
    template <bool EXT = false>
    ZD Micro glossSample(const Luts& L, float dw, float xi0, float xi1, BxdfSample& result) const {
        if (EXT && flakes) {
            const float cos_cone = a0;
            const f3    wi_l = coneUniform(xi0, xi1, cos_cone);
            const f3    h = reflect3(frame.z, wo);
            const f3    wi = Frame::init(h).frameToWorld(wi_l);
            const float wi_dot_h = clampDot(wi, h);
            const float f = wi_dot_h > cos_cone ? 1.f / solidAngleOfCone(cos_cone) : 0.f;
            const float n_dot_wi = frame.clampNdot(wi);
            result.reflection = S3(n_dot_wi * f * specular) * f0;
            result.wi = wi;
            result.pdf = f;
            result.reg_alpha = a1;
            result.scattering = a1 <= specular_threshold ? kScatterSpecular : kScatterGlossy;
            return {h, n_dot_wi, wi_dot_h};
        }
        const float n_dot_wo = frame.clampAbsNdot(wo);
        const Micro micro = ggxReflect(wo, n_dot_wo, a0, a1, specular_threshold, xi0, xi1, f0, frame, result);
        const f3    mms = dspbrMicroEc(L, f0, micro.n_dot_wi, frame.clampNdot(wo), a1);
        BxdfResult  d{S3(0.f), 0.f};
        if (dw > 0.f) {
            d.reflection = microEvaluate(L, S3(opacity) * albedo, micro.n_dot_wi, n_dot_wo, hmax3(f0));
            d.pdf = micro.n_dot_wi * kPiInv;
        }
        result.reflection = S3(micro.n_dot_wi) * (S3(specular) * (result.reflection + mms) + d.reflection);
        result.pdf = (1.f - dw) * result.pdf + dw * d.pdf;
        return micro;
    }

    template <bool EXT = false>
    ZD void baseSample(const Luts& L, Sampler& sampler, BxdfSample& result) const {  // substitute_sample.zig:281-302
        float dw = 0.f;
        if (1.f != metallic) dw = diffuseWeight(L);
        float p, xi0, xi1;
        sampler.sample3D(p, xi0, xi1);
        if (p < dw) diffuseSample(L, dw, xi0, xi1, result);
        else glossSample<EXT>(L, dw, xi0, xi1, result);
    }

    // substitute_sample.zig:304-336, coatingReflect 412-422, coatingBaseSample 426-433
    ZD void coatingSample(const Luts& L, Sampler& sampler, BxdfSample& result) const {
        float c0, c1;
        sampler.sample2D(c0, c1);
        float       coat_n_dot_h;
        const Micro micro = coating.sample(wo, c0, c1, coat_n_dot_h);
        const float f = micro.n_dot_wi;
        float       p, xi0, xi1;
        sampler.sample3D(p, xi0, xi1);
        if (p <= f) {
            const float n_dot_wo = clampAbsDot(coating.n, wo);
            const f3    coating_attenuation = coating.reflect(L, wo, micro.h, n_dot_wo, coat_n_dot_h, micro.h_dot_wi, specular_threshold, result);
            const BxdfResult base_result = baseEvaluate<true>(L, result.wi, micro.h, micro.h_dot_wi, false);
            result.reflection = (result.reflection * S3(f)) + coating_attenuation * base_result.reflection;
            result.pdf = f * result.pdf + (1.f - f) * base_result.pdf;
        } else {
            float dw = 0.f;
            if (1.f != metallic) dw = diffuseWeight(L);
            const float p1 = (p - f) / (1.f - f);
            const Micro bm = p1 < dw ? diffuseSample(L, dw, xi0, xi1, result) : glossSample<true>(L, dw, xi0, xi1, result);
            const Coating::Result c = coating.evaluate(L, result.wi, wo, bm.h, bm.h_dot_wi, specular_threshold, avoid_caustics);
            result.reflection = c.attenuation * result.reflection + c.reflection;
            result.pdf = (1.f - f) * result.pdf + f * c.pdf;
        }
    }

    // SUBST = false compiles the Substitute branch out (kernels specialised for scenes without that material); GLASS: the
    // pool kernels, the only ones that see Glass; EXT: Substitute's coating / translucency / flakes (the kernels that rebuild the
    // whole fragment: a scene with a ZYGPU_MATF_EXTENDED material runs those)
    template <bool SUBST = true, bool GLASS = false, bool EXT = false>
    ZD BxdfResult evaluate(const DScene& s, f3 wi, bool split = false) const {  // material_sample.zig:47-53
        if (GLASS && ZYGPU_MAT_GLASS == kind) return glassEvaluate(s, wi, split);
        if (ZYGPU_MAT_LIGHT == kind || ZYGPU_MAT_GLASS == kind) return {S3(0.f), 0.f};
        if (ZYGPU_MAT_DEBUG == kind) {
            const float n_dot_wi = frame.clampNdot(wi);
            const float pdf = n_dot_wi * kPiInv;
            return {S3(pdf) * albedo, pdf};
        }
        if (!SUBST) return {S3(0.f), 0.f};
        // Substitute.evaluate (substitute_sample.zig:88-145; exit_sss / volumetric: subsurface scattering, out of scope)
        const Luts L{s};
        if (EXT) {
            const float op = opacity;
            const bool  thin = op < 1.f;  // `translucent` there
            if (thin) {
                if (!sameHemisphere(wi)) {
                    const float n_dot_wi = frame.clampAbsNdot(wi), n_dot_wo = frame.clampAbsNdot(wo);
                    const float f = diffuseFresnelHack(n_dot_wi, n_dot_wo, f0.x);
                    const float pdf = n_dot_wi * ((1.f - op) * kPiInv);
                    return {S3(pdf * (1.f - f)) * albedo, pdf};
                }
            } else if (!sameHemisphere(wo)) {
                return {S3(0.f), 0.f};
            }
            const f3    h = normalize3(wo + wi);
            const float wo_dot_h = clampDot(wo, h);
            BxdfResult  base_result = baseEvaluate<true>(L, wi, h, wo_dot_h, false);
            if (thin) base_result.pdf *= op;
            if (coating.thickness > 0.f) {
                const Coating::Result c = coating.evaluate(L, wi, wo, h, wo_dot_h, specular_threshold, avoid_caustics);
                const float           pdf = c.f * c.pdf + (1.f - c.f) * base_result.pdf;
                return {c.reflection + c.attenuation * base_result.reflection, pdf};
            }
            return base_result;
        }
        if (!sameHemisphere(wo)) return {S3(0.f), 0.f};
        const f3    h = normalize3(wo + wi);
        const float wo_dot_h = clampDot(wo, h);
        return baseEvaluate(L, wi, h, wo_dot_h, false);
    }

    template <bool SUBST = true, bool EXT = false>
    ZD bool sample(const DScene& s, Sampler& sampler, BxdfSample& result) const {  // material_sample.zig:55-70
        result.transmission = false;
        result.straight = false;
        result.split_weight = 1.f;
        if (ZYGPU_MAT_LIGHT == kind || ZYGPU_MAT_GLASS == kind) return false;  // Glass: glassSample (pool kernels only)
        if (ZYGPU_MAT_DEBUG == kind) {  // debug_sample.zig:28-47
            float u0, u1;
            sampler.sample2D(u0, u1);
            const f3    is = hemisphereCosine(u0, u1);
            const f3    wi = normalize3(frame.frameToWorld(is));
            const float n_dot_wi = frame.clampNdot(wi);
            const float pdf = n_dot_wi * kPiInv;
            result.reflection = S3(pdf) * albedo;
            result.wi = wi;
            result.pdf = pdf;
            result.reg_alpha = 1.f;
            result.scattering = kScatterDiffuse;
            return true;
        }
        if (!SUBST) return false;
        // Substitute.sample (substitute_sample.zig:147-234; exit_sss / volumetric: subsurface scattering, out of scope)
        const Luts L{s};
        if (EXT && opacity < 1.f) {  // translucent: diffuse transmission with probability 1 - opacity (:184-208)
            const float op = opacity, tr = 1.f - op;
            float       p, unused0, unused1;
            sampler.sample3D(p, unused0, unused1);
            if (p <= tr) {
                float u0, u1;
                sampler.sample2D(u0, u1);  // diffuse.Lambert.reflect (diffuse.zig:14-29)
                const f3    is = hemisphereCosine(u0, u1);
                const f3    wi = normalize3(frame.frameToWorld(is));
                const float n_dot_wi = frame.clampNdot(wi);
                const float n_dot_wo = frame.clampAbsNdot(wo);
                const float f = diffuseFresnelHack(n_dot_wi, n_dot_wo, f0.x);
                result.wi = -wi;
                result.reflection = (S3(kPiInv) * albedo) * S3(tr * n_dot_wi * (1.f - f));
                result.pdf = (n_dot_wi * kPiInv) * tr;
                result.reg_alpha = 1.f;
                result.scattering = kScatterDiffuse;
            } else {
                baseSample<true>(L, sampler, result);
                result.pdf *= op;
            }
            return true;
        }
        if (!sameHemisphere(wo)) return false;
        if (EXT && coating.thickness > 0.f) coatingSample(L, sampler, result);
        else baseSample<EXT>(L, sampler, result);
        return 0.f != result.pdf;
    }
};

// hlp.triplanarMapping (material_helper.zig:81-93); p and n in object space
ZD void triplanarMapping(f3 p, f3 n, float& u, float& v) {
    const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax > ay && ax > az) {
        const float sign = copysignf(1.f, n.x);
        u = sign * p.z + 0.5f;
        v = -p.y + 0.5f;
    } else if (ay > ax && ay > az) {
        const float sign = copysignf(1.f, n.y);
        u = sign * p.x + 0.5f;
        v = -p.z + 0.5f;
    } else {
        const float sign = copysignf(1.f, n.z);
        u = -sign * p.x + 0.5f;
        v = -p.y + 0.5f;
    }
}

// The coordinates a texture is read at (texture_sampler.zig:22-35): the fragment's uv, or for tex_coord Triplanar rs.triplanarSt()
// (renderstate.zig:44-49): the object-space position projected along the dominant axis of the geometric normal
ZD void textureSt(uint32_t sampler, float u, float v, f3 p, f3 geo_n, const Trafo& trafo, float& su, float& sv) {
    if (sampler & ZYGPU_TEX_TRIPLANAR) {
        triplanarMapping(trafo.worldToObjectPoint(p), trafo.worldToObjectNormal(geo_n), su, sv);
    } else {
        su = u;
        sv = v;
    }
}

ZD f3 sampleColorMap(const DScene& s, const ZygpuMaterial& m, float u, float v, float r) {
    return sampleImageMap(s, m.color_image, m.color_sampler, m.color_scale, u, v, r);
}
// geo_n: after Vertex.sample's two-sided flip (rs.geo_n)
ZD f3 sampleColorMapAt(const DScene& s, const ZygpuMaterial& m, float u, float v, float r, f3 p, f3 geo_n, const Trafo& trafo) {
    float su, sv;
    textureSt(m.color_sampler, u, v, p, geo_n, trafo, su, sv);
    return sampleImageMap(s, m.color_image, m.color_sampler, m.color_scale, su, sv, r);
}

// Where wf_textures (wavefront.hpp, dprocedural.hpp) leaves the values of a vertex' procedural textures: the floats of the vertex' block
// of Params::tex_values
enum : uint32_t {
    kTexColor = 0,                                // 3 planes
    kTexWeight = 3,                               // + ZYGPU_STEX_ROUGHNESS .. ZYGPU_STEX_FLAKES_COVERAGE
    kTexNormal = kTexWeight + ZYGPU_STEX_NORMAL,  // 2 planes
    kTexCoatingNormal = kTexNormal + 2,           // 2 planes
    kTexDraws = kTexCoatingNormal + 2,            // the sampler dimensions the vertex' Mix textures drew (an integer's bits)
    kTexPlanes = kTexDraws + 1
};

// What the complete Substitute closure reads of the fragment beyond its frame (Renderstate: uv, stochastic_r, p, trafo)
// `tex`: the vertex' block of its procedural textures' values (null: the scene has none)
struct FragmentExtra {
    float u, v, stochastic_r;
    f3    p;
    Trafo trafo;
    const float* tex = nullptr;
    ZD float texValue(uint32_t plane) const { return tex[plane]; }
};

// hlp.sampleNormal (material_helper.zig:16-78), tex_coord UV0: the normal map's xy in the tangent frame, z reconstructed, then the
// "iray" adaption that keeps the reflection of wo above the geometric surface
ZD f3 sampleNormal(const DScene& s, const ZygpuTexture& map, f3 wo, f3 geo_n, f3 t, f3 b, f3 n, const FragmentExtra& fx, uint32_t plane = kTexNormal) {
    float nx, ny, su, sv;
    if (map.sampler & ZYGPU_TEX_PROCEDURAL) {  // evaluated by wf_textures
        nx = fx.texValue(plane);
        ny = fx.texValue(plane + 1u);
    } else {
        textureSt(map.sampler, fx.u, fx.v, fx.p, geo_n, fx.trafo, su, sv);
        sampleImageMap2(s, map.image, map.sampler, map.scale, su, sv, fx.stochastic_r, nx, ny);
    }
    const float nz = sqrtf(mx(1.f - (nx * nx + ny * ny), 0.01f));
    f3          nn;
    if (map.sampler & ZYGPU_TEX_OBJECTPOS) {  // a frame around the interpolated normal (material_helper.zig:24-29)
        nn = normalize3(Frame::init(n).frameToWorld(mk3(nx, ny, nz)));
    } else if (map.sampler & ZYGPU_TEX_TRIPLANAR) {  // triplanarTangent (material_helper.zig:30-38, 125-139)
        const f3    on = fx.trafo.worldToObjectNormal(n);
        const float ax = fabsf(on.x), ay = fabsf(on.y), az = fabsf(on.z);
        const f3    wt = (ax > ay && ax > az)   ? S3(copysignf(1.f, on.x)) * fx.trafo.r2
                         : (ay > ax && ay > az) ? S3(copysignf(1.f, on.y)) * fx.trafo.r0
                                                : S3(-copysignf(1.f, on.z)) * fx.trafo.r0;
        const f3    bt = cross3(n, wt);
        const f3    nt = cross3(bt, n);
        const Frame fr{nt, bt, n};
        nn = normalize3(fr.frameToWorld(mk3(nx, ny, nz)));
    } else {
        // rs.tangentToWorld (renderstate.zig:51-58)
        nn = normalize3(mk3(nx * t.x + ny * b.x + nz * n.x, nx * t.y + ny * b.y + nz * n.y, nx * t.z + ny * b.z + nz * n.z));
    }
    const f3    ng = geo_n;
    const f3    r = reflect3(nn, wo);
    const float a = dot3(ng, r);
    if (a >= 0.f) return nn;
    const float cos_threshold = 0.0017453f;  // cos(89.9 degrees)
    if (dot3(ng, wo) < cos_threshold) return ng;
    const float bb = dot3(ng, nn);
    const float epsilon = 1e-4f;
    f3          tangent;
    if (bb > epsilon) {
        const float distance_to_surface_along_normal = fabsf(a) / bb;
        tangent = normalize3(r + S3(distance_to_surface_along_normal) * nn);
    } else {
        tangent = nn;
    }
    tangent = tangent + S3(epsilon) * ng;
    return normalize3(wo + tangent);
}

// Material.sampleFlake (substitute_material.zig:235-273): four flakes per cell of a res x res grid, each cell with its own PCG
// stream (base/random/generator.zig:49-93 SingleGenerator: increment 1); the nearest flake centre owns the point, and the flake
// exists with probability `coverage`.  Returns the flake's 2-D number (which fixes its orientation).
ZD bool sampleFlake(float u, float v, float res, float coverage, float& xi0, float& xi1) {
    const int32_t i = int32_t(res * (u - floorf(u))), j = int32_t(res * (v - floorf(v)));  // gridCell: @mod(uv, 1)
    const float   su = res * u, sv = res * v;
    float         nearest_d = 3.40282347e+38f, nearest_r = 0.f, n0 = 0.f, n1 = 0.f;
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
            auto randomFloat = [&]() { return __uint_as_float((advance() & 0x007FFFFFu) | 0x3F800000u) - 1.f; };
            advance();
            state += (uint64_t(uint32_t(ii)) << 32) | uint64_t(uint32_t(jj));
            advance();
            for (uint32_t fl = 0; fl < 4; ++fl) {
                const float px = float(ii) + randomFloat(), py = float(jj) + randomFloat();
                const float x0 = randomFloat(), x1 = randomFloat();
                const float r = randomFloat();
                const float dx = su - px, dy = sv - py;
                const float vcd = dx * dx + dy * dy;  // squaredLength2
                if (vcd < nearest_d) {
                    nearest_d = vcd;
                    nearest_r = r;
                    n0 = x0;
                    n1 = x1;
                }
            }
        }
    }
    xi0 = n0;
    xi1 = n1;
    return nearest_r < coverage;
}

// Material.sample (material.zig:184-194) from a fragment; `two-sided flip` of vertex.zig:155-161 is done by the caller
// `color`: the Substitute colour at the hit (uniform value or colour map sample)
// `ior_outside`: Vertex.iorOutside (vertex.zig:87-92), 1 outside every dielectric; `lower_priority`: sample_base.zig:44
// EXT with `fx`: a ZYGPU_MATF_EXTENDED Substitute gets its complete closure (substitute_material.zig:114-221)
// renderstate.zig:60-69: a path that has scattered diffusely sees later surfaces rougher (PathtracerMIS' regularize_roughness)
struct Regularize {
    float weight = 0.f, alpha = 0.f;  // rs.reg_weight, rs.reg_alpha (vertex.zig:168-169)
    ZD void apply(float& a0, float& a1, float specular_threshold, bool caustics) const {
        if (0.f == weight || (a0 <= specular_threshold && !caustics)) return;
        const float k = 1.f - weight * alpha;
        a0 = 1.f - ((1.f - a0) * k);
        a1 = 1.f - ((1.f - a1) * k);
    }
};

template <bool EXT = false>
ZD MaterialSample makeMaterialSample(const DScene& s, const ZygpuMaterial& m, f3 wo, f3 geo_n, f3 t, f3 b, f3 n, bool caustics, f3 color,
                                     float ior_outside = 1.f, bool lower_priority = false, const FragmentExtra* fx = nullptr,
                                     Regularize reg = Regularize{}) {
    MaterialSample ms;
    ms.ior = m.ior;
    ms.ior_outside = ior_outside;
    ms.glass_f0 = 0.f;
    ms.thickness = 0.f;
    ms.abbe = 0.f;
    ms.absorption = S3(0.f);
    ms.lower_priority = lower_priority;
    ms.kind = m.kind;
    ms.geo_n = geo_n;
    ms.n = n;
    ms.wo = wo;
    ms.translucent = false;
    ms.avoid_caustics = !caustics;
    ms.frame = {t, b, n};
    ms.albedo = S3(0.f);
    ms.f0 = S3(0.f);
    ms.metallic = 0.f;
    ms.specular = 1.f;
    ms.specular_threshold = s.specular_threshold;
    ms.opacity = 1.f;
    ms.coating.thickness = 0.f;
    ms.flakes = false;
    if (ZYGPU_MAT_DEBUG == m.kind) {
        const f3   nn = cross3(t, b);
        const bool same_side = dot3(nn, n) > 0.f;
        ms.albedo = same_side ? mk3(m.color_a[0], m.color_a[1], m.color_a[2]) : mk3(m.color_b[0], m.color_b[1], m.color_b[2]);
        ms.a0 = ms.a1 = 1.f;
        ms.can_evaluate = true;
    } else if (ZYGPU_MAT_LIGHT == m.kind) {
        ms.a0 = ms.a1 = 0.f;
        ms.can_evaluate = false;
    } else if (ZYGPU_MAT_GLASS == m.kind) {
        // Glass.sample / Sample.init (glass_material.zig:49-79, glass_sample.zig:33-66); roughness, specular and the normal may
        // come from maps (EXT, like the extended Substitute)
        float raw_r = m.roughness, specular = m.specular;
        if (EXT && (m.flags & ZYGPU_MATF_EXTENDED)) {
            const FragmentExtra& f = *fx;
            const ZygpuTexture & tr = m.tex[ZYGPU_STEX_ROUGHNESS], &tsp = m.tex[ZYGPU_STEX_SPECULAR], &tn = m.tex[ZYGPU_STEX_NORMAL];
            float su, sv;
            if (kNull != tr.image) {
                if (tr.sampler & ZYGPU_TEX_PROCEDURAL) {
                    raw_r = f.texValue(kTexWeight + ZYGPU_STEX_ROUGHNESS);
                } else {
                    textureSt(tr.sampler, f.u, f.v, f.p, geo_n, f.trafo, su, sv);
                    raw_r = sampleImageMap1(s, tr.image, tr.sampler, tr.scale, su, sv, f.stochastic_r);
                }
            }
            if (kNull != tsp.image) {
                if (tsp.sampler & ZYGPU_TEX_PROCEDURAL) {
                    specular = f.texValue(kTexWeight + ZYGPU_STEX_SPECULAR);
                } else {
                    textureSt(tsp.sampler, f.u, f.v, f.p, geo_n, f.trafo, su, sv);
                    specular = sampleImageMap1(s, tsp.image, tsp.sampler, tsp.scale, su, sv, f.stochastic_r);
                }
            }
            if (kNull != tn.image) ms.frame = Frame::init(sampleNormal(s, tn, wo, geo_n, t, b, n, f));
        }
        const float r = 0.f == raw_r ? 0.f : mx(raw_r, kMinRoughness);
        ms.a0 = ms.a1 = r * r;
        reg.apply(ms.a0, ms.a1, s.specular_threshold, caustics);
        const bool rough = ms.a0 > 0.f;
        ms.can_evaluate = rough && m.ior != ior_outside;
        const float t_ = (m.ior - ior_outside) / (m.ior + ior_outside);
        ms.glass_f0 = rough ? t_ * t_ : 0.f;
        ms.specular = specular;
        ms.thickness = m.thickness;
        ms.abbe = m.abbe;
        ms.absorption = mk3(m.absorption[0], m.absorption[1], m.absorption[2]);
        ms.translucent = m.thickness > 0.f;
    } else if (EXT && (m.flags & ZYGPU_MATF_EXTENDED)) {
        const FragmentExtra& f = *fx;
        auto weight = [&](uint32_t slot, float uniform) {  // ts.sample2D_1
            const ZygpuTexture& tx = m.tex[slot];
            if (kNull == tx.image) return uniform;
            if (tx.sampler & ZYGPU_TEX_PROCEDURAL) return f.texValue(kTexWeight + slot);  // evaluated by wf_textures
            float su, sv;
            textureSt(tx.sampler, f.u, f.v, f.p, geo_n, f.trafo, su, sv);
            return sampleImageMap1(s, tx.image, tx.sampler, tx.scale, su, sv, f.stochastic_r);
        };
        const float roughness = mx(weight(ZYGPU_STEX_ROUGHNESS, m.roughness), kMinRoughness);
        const float metallic = weight(ZYGPU_STEX_METALLIC, m.metallic);
        const float specular = weight(ZYGPU_STEX_SPECULAR, m.specular);
        if (m.anisotropy > 0.f) {
            const float rv = mx(roughness * (1.f - m.anisotropy), kMinRoughness);
            ms.a0 = roughness * roughness;
            ms.a1 = rv * rv;
        } else {
            ms.a0 = ms.a1 = roughness * roughness;
        }
        reg.apply(ms.a0, ms.a1, s.specular_threshold, caustics);
        const float coating_scale = weight(ZYGPU_STEX_COATING_SCALE, m.coating_scale);
        const float coating_thickness = coating_scale * m.coating_thickness;
        const float coating_weight = coating_scale > 0.1f ? 1.f : coating_scale;
        const float coating_ior = lerpf(ior_outside, m.coating_ior, coating_weight);
        const float ior = m.ior, ior_outer = coating_thickness > 0.f ? coating_ior : ior_outside, ior_medium = ior_outside;
        const float translucency = weight(ZYGPU_STEX_TRANSLUCENCY, m.translucency);
        ms.albedo = S3(1.f - metallic) * color;
        ms.can_evaluate = ior != ior_medium;
        ms.translucent = translucency > 0.f;
        const float t_ = (ior - ior_outer) / (ior + ior_outer);
        ms.f0 = lerp3(S3(t_ * t_), color, S3(metallic));
        ms.metallic = metallic;
        ms.specular = specular;
        ms.opacity = 1.f - (0.5f * translucency);
        if (kNull != m.tex[ZYGPU_STEX_NORMAL].image) ms.frame = Frame::init(sampleNormal(s, m.tex[ZYGPU_STEX_NORMAL], wo, geo_n, t, b, n, f));
        if (coating_thickness > 0.f) {  // substitute_material.zig:164-181
            const ZygpuTexture &nm = m.tex[ZYGPU_STEX_NORMAL], &cnm = m.tex[ZYGPU_STEX_COATING_NORMAL];
            // Texture.equal: the same description (two uniform maps are equal, too)
            const bool same = nm.image == cnm.image && (kNull == nm.image || (nm.sampler == cnm.sampler && nm.scale[0] == cnm.scale[0] && nm.scale[1] == cnm.scale[1]));
            if (same) ms.coating.n = ms.frame.z;
            else if (kNull != cnm.image) ms.coating.n = sampleNormal(s, cnm, wo, geo_n, t, b, n, f, kTexCoatingNormal);
            else ms.coating.n = n;
            const float r = mx(weight(ZYGPU_STEX_COATING_ROUGHNESS, m.coating_roughness), kMinRoughness);
            ms.coating.absorption_coef = mk3(m.coating_absorption[0], m.coating_absorption[1], m.coating_absorption[2]);
            ms.coating.thickness = coating_thickness;
            const float ct = (coating_ior - ior_outside) / (coating_ior + ior_outside);
            ms.coating.f0 = ct * ct;
            float ca0 = r * r, ca1 = r * r;
            reg.apply(ca0, ca1, s.specular_threshold, caustics);
            ms.coating.alpha = ca0;
            ms.coating.weight = coating_weight;
        }
        // the rotation applies to the base frame only, after the coating took its normal
        const float rotation = weight(ZYGPU_STEX_ROTATION, m.rotation) * (2.f * kPi);
        if (rotation > 0.f) {  // Frame.rotateTangenFrame (frame.zig:64-73)
            const f3 ft = ms.frame.x, fb = ms.frame.y;
            const f3 sin_a = S3(sinF(rotation)), cos_a = S3(cosF(rotation));
            ms.frame.x = cos_a * ft + sin_a * fb;
            ms.frame.y = -sin_a * ft + cos_a * fb;
        }
        const float flakes_coverage = weight(ZYGPU_STEX_FLAKES_COVERAGE, m.flakes_coverage);
        if (flakes_coverage > 0.f) {  // substitute_material.zig:190-215
            const f3 op = f.trafo.worldToObjectNormal(f.p - f.trafo.position);
            const f3 on = f.trafo.worldToObjectNormal(ms.frame.z);
            float    fu, fv;
            triplanarMapping(op, on, fu, fv);
            fu -= floorf(fu);  // math.frac
            fv -= floorf(fv);
            float xi0, xi1;
            if (sampleFlake(fu, fv, m.flakes_res, flakes_coverage, xi0, xi1)) {
                const float fa = m.flakes_alpha;
                // flakesA2cone (substitute_material.zig:222-227): the cone of 7 degrees bounds the flake's lobe
                const float target_angle = solidAngleOfCone(0.992546151641322f);  // cos(7 degrees)
                const float limit = target_angle / ((4.f * kPi) - target_angle);
                const float a2_cone = mn(limit, 0.5f * fa);
                const float fa2 = fa - a2_cone;
                const float cos_cone = 1.f - (2.f * a2_cone) / (1.f + a2_cone);
                float       n_dot_h;
                const f3    mh = ggxSample(wo, fa2, fa2, xi0, xi1, ms.frame, n_dot_h);
                ms.metallic = 1.f;
                ms.f0 = mk3(m.flakes_color[0], m.flakes_color[1], m.flakes_color[2]);
                ms.a0 = cos_cone;
                ms.a1 = a2_cone;
                ms.frame = Frame::init(mh);
                ms.flakes = true;
            }
        }
    } else {
        const float roughness = mx(m.roughness, kMinRoughness);
        if (m.anisotropy > 0.f) {
            const float rv = mx(roughness * (1.f - m.anisotropy), kMinRoughness);
            ms.a0 = roughness * roughness;
            ms.a1 = rv * rv;
        } else {
            ms.a0 = ms.a1 = roughness * roughness;
        }
        reg.apply(ms.a0, ms.a1, s.specular_threshold, caustics);
        const float ior = m.ior, ior_outer = ior_outside, ior_medium = ior_outside;
        ms.albedo = S3(1.f - m.metallic) * color;
        ms.can_evaluate = ior != ior_medium;
        const float t_ = (ior - ior_outer) / (ior + ior_outer);
        const float f0 = t_ * t_;
        ms.f0 = lerp3(S3(f0), color, S3(m.metallic));
        ms.metallic = m.metallic;
        ms.specular = m.specular;
        ms.opacity = 1.f - (0.5f * 0.f);
    }
    return ms;
}

}  // namespace zd
