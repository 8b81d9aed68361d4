// Albedo tables of the GGX energy compensation, computed at scene-compile time.
//
// The reference ships these numbers as a generated source file (material/ggx_integral.zig) written by its own
// integrator (material/ggx_integrate.zig).  This file re-runs that procedure instead of carrying the tables:
//   E_m      [32 alpha][32 n.wo]          integrate_micro_directional_albedo   ggx_integrate.zig:27-59, 208-257
//   E_m_avg  [32 alpha]                   integrate_micro_average_albedo       ggx_integrate.zig:61-76, 259-299
//   E        [16 f0][16 alpha][16 n.wo]   integrate_directional_albedo         ggx_integrate.zig:92-120, 301-364
//   E_avg    [16 f0][16 alpha]            integrate_average_albedo (<= 0.9997) ggx_integrate.zig:122-137, 366-414
//   E_s      [16 f0 <= 0.25][16 alpha][16 n.wo]  integrate_f_s_ss (rough dielectrics) ggx_integrate.zig:134-205, 418-486
// 1024 Hammersley points per entry (base/math/sample_distribution.zig:3-18), shading frame = identity, wo = (sin, 0, cos),
// isotropic GGX reflection sampled by visible normals (ggx.zig:96-125, 396-414) with Schlick Fresnel (fresnel.zig:5-29).
// The reference prints each entry with 8 decimals; tests/golden/ggx_lut_golden.json holds a sample of those entries and
// test_host.py checks this generator against them.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace zyg {
namespace ggxlut {

constexpr int   kEmSize = 32, kESize = 16, kSamples = 1024;
constexpr float kPiF = 3.14159265358979323846f;
constexpr float kDotMin = 0.00001f;                // base/math/safe.zig:5
constexpr float kMinAlpha = 0.01314f * 0.01314f;   // ggx.zig:14-15

inline float mnf(float a, float b) { return a < b ? a : b; }
inline float mxf(float a, float b) { return b < a ? a : b; }
inline float clampf(float x, float lo, float hi) { return mnf(mxf(x, lo), hi); }
inline float lerpf(float a, float b, float t) { return std::fmaf(1.f - t, a, t * b); }  // base/math/util.zig:3-16

inline void hammersley(uint32_t i, uint32_t n, float& x, float& y) {
    uint32_t b = (i << 16) | (i >> 16);
    b = ((b & 0x55555555u) << 1) | ((b & 0xAAAAAAAAu) >> 1);
    b = ((b & 0x33333333u) << 2) | ((b & 0xCCCCCCCCu) >> 2);
    b = ((b & 0x0F0F0F0Fu) << 4) | ((b & 0xF0F0F0F0u) >> 4);
    b = ((b & 0x00FF00FFu) << 8) | ((b & 0xFF00FF00u) >> 8);
    x = float(i) / float(n);
    y = float(b) * 2.3283064365386963e-10f;
}

// smpl.hemisphereCosine(uv).z (sampling.zig:8-54)
inline float hemisphereCosineZ(float u0, float u1) {
    const float s0 = (u0 * 2.f) - 1.f, s1 = (u1 * 2.f) - 1.f;
    float       x = 0.f, y = 0.f;
    if (!(0.f == s0 && 0.f == s1)) {
        float r, theta;
        if (std::fabs(s0) > std::fabs(s1)) {
            r = s0;
            theta = (kPiF / 4.f) * (s1 / s0);
        } else {
            r = s1;
            theta = (kPiF / 2.f) - (kPiF / 4.f) * (s0 / s1);
        }
        x = std::cos(theta) * r;
        y = std::sin(theta) * r;
    }
    return std::sqrt(mxf(0.f, 1.f - x * x - y * y));
}

inline float bilinear(float c0, float c1, float c2, float c3, float s, float t) {  // math.zig:172-179
    const float _s = 1.f - s, _t = 1.f - t;
    return _t * (_s * c0 + s * c1) + t * (_s * c2 + s * c3);
}

// base/math/interpolated_function.zig:105-271
inline float eval1(const float* smp, int N, float x) {
    const float    o = mnf(x, 1.f) * float(N - 1);
    const uint32_t offset = uint32_t(o);
    const float    t = o - float(offset);
    return lerpf(smp[offset], smp[std::min(offset + 1, uint32_t(N - 1))], t);
}
inline float eval2(const float* smp, int X, int Y, float x, float y) {
    const float ox = mnf(x, 1.f) * float(X - 1), oy = mnf(y, 1.f) * float(Y - 1);
    const int   o0 = int(ox), o1 = int(oy);
    const float t0 = ox - float(o0), t1 = oy - float(o1);
    const int   col1 = std::min(o0 + 1, X - 1);
    const int   row0 = o1 * X, row1 = std::min(o1 + 1, Y - 1) * X;
    return bilinear(smp[o0 + row0], smp[col1 + row0], smp[o0 + row1], smp[col1 + row1], t0, t1);
}
inline float eval3(const float* smp, int X, int Y, int Z, float x, float y, float z) {
    const float ox = mnf(x, 1.f) * float(X - 1), oy = mnf(y, 1.f) * float(Y - 1), oz = mnf(z, 1.f) * float(Z - 1);
    const int   o0 = int(ox), o1 = int(oy), o2 = int(oz);
    const float t0 = ox - float(o0), t1 = oy - float(o1), t2 = oz - float(o2);
    const int   col1 = std::min(o0 + 1, X - 1);
    const int   row0 = o1 * X, row1 = std::min(o1 + 1, Y - 1) * X;
    const int   area = X * Y;
    const int   s0 = o2 * area, s1 = std::min(o2 + 1, Z - 1) * area;
    const float c0 = bilinear(smp[o0 + row0 + s0], smp[col1 + row0 + s0], smp[o0 + row1 + s0], smp[col1 + row1 + s0], t0, t1);
    const float c1 = bilinear(smp[o0 + row0 + s1], smp[col1 + row0 + s1], smp[o0 + row1 + s1], smp[col1 + row1 + s1], t0, t1);
    return lerpf(c0, c1, t2);
}

struct ReflectSample {
    float reflection, pdf, n_dot_wi;
};

// ggx.Iso.reflect with the identity frame, wo = (wo_x, 0, wo_z), scalar Schlick f0
inline ReflectSample isoReflect(float wo_x, float wo_z, float n_dot_wo, float alpha, float xi0, float xi1, float f0) {
    // Aniso.sample (ggx.zig:396-414)
    const float vx = alpha * wo_x, vy = alpha * 0.f, vz = wo_z;
    const float vl = std::sqrt(vx * vx + vy * vy + vz * vz);
    const float nvx = vx / vl, nvy = vy / vl, nvz = vz / vl;
    const float phi = (2.f * kPiF) * xi0;
    const float z = std::fmaf(1.f - xi1, 1.f + nvz, -nvz);
    const float sin_theta = std::sqrt(clampf(1.f - z * z, 0.f, 1.f));
    const float x = sin_theta * std::cos(phi), y = sin_theta * std::sin(phi);
    const float hx = x + nvx, hy = y + nvy, hz = z + nvz;
    const float mx_ = alpha * hx, my_ = alpha * hy, mz_ = hz;
    const float ml = std::sqrt(mx_ * mx_ + my_ * my_ + mz_ * mz_);
    const float h0 = mx_ / ml, h1 = my_ / ml, h2 = mz_ / ml;
    const float n_dot_h = clampf(h2, kDotMin, 1.f);

    const float wo_dot_h = clampf(wo_x * h0 + 0.f * h1 + wo_z * h2, kDotMin, 1.f);
    const float s2 = 2.f * wo_dot_h;
    const float w0 = std::fmaf(s2, h0, -wo_x), w1 = std::fmaf(s2, h1, -0.f), w2 = std::fmaf(s2, h2, -wo_z);
    const float wl = std::sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const float n_dot_wi = clampf(w2 / wl, kDotMin, 1.f);

    const float alpha2 = alpha * alpha;
    const float dd = std::fmaf(n_dot_h * n_dot_h, alpha2 - 1.f, 1.f);  // ggx.zig distribution
    const float d = alpha2 / (kPiF * dd * dd);
    const float t_wi = std::sqrt(std::fmaf(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));  // visibilityAndG1Wo
    const float t_wo = std::sqrt(std::fmaf(1.f - alpha2, n_dot_wo * n_dot_wo, alpha2));
    const float g0 = 0.5f / (n_dot_wi * t_wo + n_dot_wo * t_wi);
    const float g1 = t_wo + n_dot_wo;
    const float om = 1.f - wo_dot_h, om2 = om * om;
    const float f = std::fmaf((om2 * om2) * om, 1.f - f0, f0);  // fresnel.zig schlick
    return {(d * g0) * f, (0.5f * d) / g1, n_dot_wi};
}

struct Tables {
    std::vector<float> e_m, e_m_avg, e, e_avg, e_s;
};

// integrate_f_s_ss (ggx_integrate.zig:134-205): reflected + refracted energy of a rough dielectric interface seen from
// outside (eta_i = 1), with ggx.Iso.reflectNoFresnel / refractNoFresnel (ggx.zig:161-227) and fresnel.schlick1
inline float fsSs(float alpha, float f0, float ior_t, float n_dot_wo) {
    if (alpha < kMinAlpha || ior_t <= 1.f) return 1.f;
    const float cn_dot_wo = clampf(n_dot_wo, kDotMin, 1.f);
    const float wo_x = std::sqrt(1.f - cn_dot_wo * cn_dot_wo), wo_z = cn_dot_wo;
    const float eta_i = 1.f, eta_t = ior_t;
    const float alpha2 = alpha * alpha;
    float       accum = 0.f;
    for (uint32_t i = 0; i < uint32_t(kSamples); ++i) {
        float xi0, xi1;
        hammersley(i, kSamples, xi0, xi1);
        // Aniso.sample (ggx.zig:393-409)
        const float vx = alpha * wo_x, vz = wo_z;
        const float vl = std::sqrt(vx * vx + 0.f + vz * vz);
        const float nvx = vx / vl, nvy = 0.f / vl, nvz = vz / vl;
        const float phi = (2.f * kPiF) * xi0;
        const float z = std::fmaf(1.f - xi1, 1.f + nvz, -nvz);
        const float sin_theta = std::sqrt(clampf(1.f - z * z, 0.f, 1.f));
        const float x = sin_theta * std::cos(phi), y = sin_theta * std::sin(phi);
        const float hx = x + nvx, hy = y + nvy, hz = z + nvz;
        const float mx_ = alpha * hx, my_ = alpha * hy, mz_ = hz;
        const float ml = std::sqrt(mx_ * mx_ + my_ * my_ + mz_ * mz_);
        const float h0 = mx_ / ml, h1 = my_ / ml, h2 = mz_ / ml;
        const float n_dot_h = clampf(h2, kDotMin, 1.f);

        const float wo_dot_h = clampf(wo_x * h0 + 0.f * h1 + wo_z * h2, kDotMin, 1.f);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - wo_dot_h * wo_dot_h);
        const float wi_dot_h = std::sqrt(1.f - sint2);
        const float cos_x = eta_i > eta_t ? wi_dot_h : wo_dot_h;
        const float om = 1.f - cos_x, om2 = om * om;
        const float f = std::fmaf((om2 * om2) * om, 1.f - f0, f0);

        const float dd = std::fmaf(n_dot_h * n_dot_h, alpha2 - 1.f, 1.f);
        const float d = alpha2 / (kPiF * dd * dd);
        {  // reflectNoFresnel
            const float s2 = 2.f * wo_dot_h;
            const float w0 = std::fmaf(s2, h0, -wo_x), w1 = std::fmaf(s2, h1, -0.f), w2 = std::fmaf(s2, h2, -wo_z);
            const float wl = std::sqrt(w0 * w0 + w1 * w1 + w2 * w2);
            const float n_dot_wi = clampf(w2 / wl, kDotMin, 1.f);
            const float t_wi = std::sqrt(std::fmaf(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));
            const float t_wo = std::sqrt(std::fmaf(1.f - alpha2, cn_dot_wo * cn_dot_wo, alpha2));
            const float g0 = 0.5f / (n_dot_wi * t_wo + cn_dot_wo * t_wi), g1 = t_wo + cn_dot_wo;
            accum += (mnf(n_dot_wi, n_dot_wo) * f * (d * g0)) / ((0.5f * d) / g1);
        }
        {  // refractNoFresnel(wo, h, cn_dot_wo, n_dot_h, -wi_dot_h, -wo_dot_h, alpha, ior)
            const float wi_dot_h_r = -wi_dot_h, wo_dot_h_r = -wo_dot_h;
            const float abs_wi = clampf(std::fabs(wi_dot_h_r), kDotMin, 1.f), abs_wo = clampf(std::fabs(wo_dot_h_r), kDotMin, 1.f);
            const float k = std::fmaf(eta, abs_wo, -abs_wi);
            const float w0 = k * h0 - eta * wo_x, w1 = k * h1 - eta * 0.f, w2 = k * h2 - eta * wo_z;
            const float wl = std::sqrt(w0 * w0 + w1 * w1 + w2 * w2);
            const float n_dot_wi = clampf(std::fabs(w2 / wl), kDotMin, 1.f);
            const float ga = cn_dot_wo * std::sqrt(std::fmaf(1.f - alpha2, n_dot_wi * n_dot_wi, alpha2));
            const float gb = n_dot_wi * std::sqrt(std::fmaf(1.f - alpha2, cn_dot_wo * cn_dot_wo, alpha2));
            const float g = (2.f * n_dot_wi * cn_dot_wo) / (ga + gb);
            const float refr = d * g;
            const float factor = (abs_wi * abs_wo) / (n_dot_wi * cn_dot_wo);
            const float dn = eta_i * wo_dot_h_r + eta_t * wi_dot_h_r, denom = dn * dn;
            const float sqr_eta_t = eta_t * eta_t;
            const float g1 = (2.f * cn_dot_wo) / (cn_dot_wo + std::sqrt(alpha2 + (1.f - alpha2) * (cn_dot_wo * cn_dot_wo)));
            const float pdf_vr = g1 * abs_wo * d / cn_dot_wo;
            const float reflection = (factor * sqr_eta_t / denom) * refr;
            const float pdf = pdf_vr * (abs_wi * sqr_eta_t / denom);
            accum += (n_dot_wi * (1.f - f) * reflection) / pdf;
        }
    }
    return accum / float(kSamples);
}

inline float microDirectionalAlbedo(float alpha, float n_dot_wo) {
    if (0.f == alpha) return 1.f;
    const float calpha = mxf(alpha, kMinAlpha);
    const float wo_x = std::sqrt(1.f - n_dot_wo * n_dot_wo);
    float       accum = 0.f;
    for (uint32_t i = 0; i < uint32_t(kSamples); ++i) {
        float xi0, xi1;
        hammersley(i, kSamples, xi0, xi1);
        const ReflectSample r = isoReflect(wo_x, n_dot_wo, n_dot_wo, calpha, xi0, xi1, 1.f);
        accum += ((r.n_dot_wi * r.reflection) / r.pdf) / float(kSamples);
    }
    return accum;
}

inline float dspbrMicroEc(float f0, float n_dot_wi, float n_dot_wo, float alpha, const Tables& t) {  // ggx_integrate.zig:78-90
    const float e_wo = eval2(t.e_m.data(), kEmSize, kEmSize, n_dot_wo, alpha);
    const float e_wi = eval2(t.e_m.data(), kEmSize, kEmSize, n_dot_wi, alpha);
    const float e_avg = eval1(t.e_m_avg.data(), kEmSize, alpha);
    const float m = ((1.f - e_wo) * (1.f - e_wi)) / (kPiF * (1.f - e_avg));
    const float f_avg = (1.f / 21.f) + (20.f / 21.f) * f0;
    const float f = (f_avg * f_avg * e_avg) / (1.f - (f_avg * (1.f - e_avg)));
    return m * f;
}

inline float directionalAlbedo(float alpha, float f0, float n_dot_wo, const Tables& t) {
    const float calpha = mxf(alpha, kMinAlpha);
    const float wo_x = std::sqrt(1.f - n_dot_wo * n_dot_wo);
    float       accum = 0.f;
    for (uint32_t i = 0; i < uint32_t(kSamples); ++i) {
        float xi0, xi1;
        hammersley(i, kSamples, xi0, xi1);
        const ReflectSample r = isoReflect(wo_x, n_dot_wo, n_dot_wo, calpha, xi0, xi1, f0);
        const float         mms = dspbrMicroEc(f0, r.n_dot_wi, n_dot_wo, calpha, t);
        accum += ((r.n_dot_wi * (r.reflection + mms)) / r.pdf) / float(kSamples);
    }
    return mnf(accum, 1.f);
}

inline const Tables& tables() {
    static const Tables cached = [] {
        Tables      t;
        const float step_m = 1.f / float(kEmSize - 1), step_e = 1.f / float(kESize - 1);

        t.e_m.resize(size_t(kEmSize) * kEmSize);
        float alpha = 0.f;
        for (int a = 0; a < kEmSize; ++a, alpha += step_m) {
            float n_dot_wo = 0.f;
            for (int i = 0; i < kEmSize; ++i, n_dot_wo += step_m) t.e_m[size_t(a) * kEmSize + i] = microDirectionalAlbedo(alpha, n_dot_wo);
        }

        t.e_m_avg.resize(kEmSize);
        alpha = 0.f;
        for (int a = 0; a < kEmSize; ++a, alpha += step_m) {
            float accum = 0.f;
            for (uint32_t i = 0; i < uint32_t(kSamples); ++i) {
                float xi0, xi1;
                hammersley(i, kSamples, xi0, xi1);
                accum += eval2(t.e_m.data(), kEmSize, kEmSize, hemisphereCosineZ(xi0, xi1), alpha) / float(kSamples);
            }
            t.e_m_avg[a] = accum;
        }

        t.e.resize(size_t(kESize) * kESize * kESize);
        float  f0 = 0.f;
        size_t count = 0;
        for (int z = 0; z < kESize; ++z, f0 += step_e) {
            alpha = 0.f;
            for (int a = 0; a < kESize; ++a, alpha += step_e) {
                float n_dot_wo = 0.f;
                for (int i = 0; i < kESize; ++i, n_dot_wo += step_e) t.e[count++] = directionalAlbedo(alpha, f0, n_dot_wo, t);
            }
        }

        t.e_avg.resize(size_t(kESize) * kESize);
        f0 = 0.f;
        for (int z = 0; z < kESize; ++z, f0 += step_e) {
            alpha = 0.f;
            for (int a = 0; a < kESize; ++a, alpha += step_e) {
                float accum = 0.f;
                for (uint32_t i = 0; i < uint32_t(kSamples); ++i) {
                    float xi0, xi1;
                    hammersley(i, kSamples, xi0, xi1);
                    accum += eval3(t.e.data(), kESize, kESize, kESize, hemisphereCosineZ(xi0, xi1), alpha, f0) / float(kSamples);
                }
                t.e_avg[size_t(z) * kESize + a] = mnf(accum, 0.9997f);
            }
        }

        // make_f_s_ss_table (ggx_integrate.zig:418-486): f0 in [0, 0.25], ior = Schlick.F0ToIor(f0) (fresnel.zig:26-29)
        t.e_s.resize(size_t(kESize) * kESize * kESize);
        f0 = 0.f;
        count = 0;
        for (int z = 0; z < kESize; ++z, f0 += 0.25f * step_e) {
            const float r = std::sqrt(f0);
            const float ior = ((-f0 - 1.f) / (f0 - 1.f)) - ((2.f * r) / ((r - 1.f) * (r + 1.f)));
            alpha = 0.f;
            for (int a = 0; a < kESize; ++a, alpha += step_e) {
                float n_dot_wo = 0.f;
                for (int i = 0; i < kESize; ++i, n_dot_wo += step_e) t.e_s[count++] = fsSs(alpha, f0, ior, n_dot_wo);
            }
        }
        return t;
    }();
    return cached;
}

}  // namespace ggxlut
}  // namespace zyg
