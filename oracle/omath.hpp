// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in zyg_amd/ (the product) may include, link or call this.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
//
// CPU restatement of the reference's base/math subset used by the integration loop.
// Lane-wise f32, no implicit contraction (build with -ffp-contract=off); fused multiply-add only
// where the reference writes @mulAdd.  min/max are the x86 `a < b ? a : b` forms
// (reference src/base/math/util.zig:17-29).
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};

inline V4 S(float s) { return {{s, s, s, s}}; }
inline V4 mk(float x, float y, float z, float w = 0.f) { return {{x, y, z, w}}; }
inline V4 ld3(const float* p) { return {{p[0], p[1], p[2], 0.f}}; }
inline V4 ld4(const float* p) { return {{p[0], p[1], p[2], p[3]}}; }

inline V4 operator+(V4 a, V4 b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}}; }
inline V4 operator-(V4 a, V4 b) { return {{a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}}; }
inline V4 operator*(V4 a, V4 b) { return {{a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]}}; }
inline V4 operator/(V4 a, V4 b) { return {{a[0] / b[0], a[1] / b[1], a[2] / b[2], a[3] / b[3]}}; }
inline V4 operator-(V4 a) { return {{-a[0], -a[1], -a[2], -a[3]}}; }
inline V4& operator+=(V4& a, V4 b) { return a = a + b; }
inline V4& operator*=(V4& a, V4 b) { return a = a * b; }
inline V4& operator/=(V4& a, V4 b) { return a = a / b; }

inline float mn(float x, float y) { return x < y ? x : y; }
inline float mx(float x, float y) { return y < x ? x : y; }
inline float clampf(float x, float lo, float hi) { return mn(mx(x, lo), hi); }
inline float saturate(float x) { return clampf(x, 0.f, 1.f); }
// std.math.clamp (used by rectangle.zig SphQuad): max(lower, min(val, upper)) via @max/@min
inline float stdClamp(float val, float lo, float hi) { return std::fmax(lo, std::fmin(val, hi)); }

inline V4 min4(V4 a, V4 b) { return {{mn(a[0], b[0]), mn(a[1], b[1]), mn(a[2], b[2]), mn(a[3], b[3])}}; }
inline V4 max4(V4 a, V4 b) { return {{mx(a[0], b[0]), mx(a[1], b[1]), mx(a[2], b[2]), mx(a[3], b[3])}}; }
inline V4 fma4(V4 a, V4 b, V4 c) {
    return {{std::fmaf(a[0], b[0], c[0]), std::fmaf(a[1], b[1], c[1]), std::fmaf(a[2], b[2], c[2]), std::fmaf(a[3], b[3], c[3])}};
}
inline V4 sqrt4(V4 a) { return {{std::sqrt(a[0]), std::sqrt(a[1]), std::sqrt(a[2]), std::sqrt(a[3])}}; }

// vector4.zig:37-60
inline float dot3(V4 a, V4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline float squaredLength3(V4 a) { return dot3(a, a); }
inline float length3(V4 a) { return std::sqrt(dot3(a, a)); }
inline float squaredDistance3(V4 a, V4 b) { return squaredLength3(a - b); }
inline float distance3(V4 a, V4 b) { return length3(a - b); }
inline V4    normalize3(V4 a) { return a / S(length3(a)); }
inline float hmax3(V4 a) { return mx(a[0], mx(a[1], a[2])); }
inline float hmin3(V4 a) { return mn(a[0], mn(a[1], a[2])); }
inline float hmax4(V4 a) { return mx(a[0], mx(a[1], mx(a[2], a[3]))); }
inline float hmin4(V4 a) { return mn(a[0], mn(a[1], mn(a[2], a[3]))); }
inline float average3(V4 a) { return (a[0] + a[1] + a[2]) / 3.f; }

// vector4.zig:73-92
inline V4 cross3(V4 a, V4 b) {
    const V4 t0 = {{b[1], b[2], b[0], b[3]}};
    const V4 t1 = {{a[1], a[2], a[0], a[3]}};
    const V4 t2 = fma4(t0, a, -(t1 * b));
    return {{t2[1], t2[2], t2[0], t2[3]}};
}

// vector4.zig:121 gramSchmidt; util.zig:3-16 lerp
inline V4    gramSchmidt(V4 v, V4 w) { return fma4(S(-dot3(v, w)), w, v); }
inline float lerpf(float a, float b, float t) { return std::fmaf(1.f - t, a, t * b); }
inline V4    lerp4(V4 a, V4 b, V4 t) { return fma4(S(1.f) - t, a, t * b); }

// vector4.zig:98-110 (Duff et al.)
inline void orthonormalBasis3(V4 n, V4& t, V4& b) {
    const float sign = std::copysign(1.f, n[2]);
    const float c = -1.f / (sign + n[2]);
    const float d = n[0] * n[1] * c;
    t = mk(1.f + sign * n[0] * n[0] * c, sign * d, -sign * n[0]);
    b = mk(d, sign + n[1] * n[1] * c, -n[1]);
}


// sin / cos / acos of the reference, restated.  rectangle.zig calls std.math.acos, whose f32 path in Zig's std
// (lib/std/math/acos.zig, not vendored in /root/reference; Zig toolchain of build.zig, unpinned) is the musl /
// FreeBSD msun e_acosf.c algorithm; @sin/@cos lower to compiler-rt's sinf/cosf, ports of musl's sinf.c/cosf.c
// (k_sinf/k_cosf double-precision kernels).  Arguments in the integration loop stay within |x| <= 9*pi/4
// (diskConcentric: [-pi/4, 3pi/4]; SphQuad and GGX: [0, 2pi]); larger ones fall back to libm.
namespace m32 {
inline uint32_t bits(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return u;
}
inline float fromBits(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline float sindf(double x) {  // musl __sindf
    const double S1 = -0x1.5555554cbac77p-3, S2 = 0x1.11110896efbb2p-7, S3 = -0x1.a00f9e2cae774p-13, S4 = 0x1.6cd878c3b46a7p-19;
    const double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return float((x + s * (S1 + z * S2)) + s * w * r);
}
inline float cosdf(double x) {  // musl __cosdf
    const double C0 = -0x1.ffffffd0c5e81p-2, C1 = 0x1.55553e1053a42p-5, C2 = -0x1.6c087e80f1e27p-10, C3 = 0x1.99342e0ee5069p-16;
    const double z = x * x, w = z * z, r = C2 + z * C3;
    return float(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
constexpr double kPio2 = 1.57079632679489661923;
// musl __rem_pio2f, medium size (|x| < 2^28 * pi/2): x = n * pi/2 + y, |y| <= pi/4, in double precision
inline int remPio2f(float x, double& y) {
    const double toint = 1.5 / 2.220446049250313e-16, pio4 = 0x1.921fb6p-1, invpio2 = 6.36619772367581382433e-01,
                 pio2_1 = 1.57079631090164184570e+00, pio2_1t = 1.58932547735281966916e-08;
    double  fn = double(x) * invpio2 + toint - toint;
    int32_t n = int32_t(fn);
    y = x - fn * pio2_1 - fn * pio2_1t;
    if (y < -pio4) {
        n--;
        fn--;
        y = x - fn * pio2_1 - fn * pio2_1t;
    } else if (y > pio4) {
        n++;
        fn++;
        y = x - fn * pio2_1 - fn * pio2_1t;
    }
    return n;
}
}  // namespace m32

inline float sinF(float x) {
    const uint32_t hx = m32::bits(x), ix = hx & 0x7fffffffu;
    const bool     sign = 0 != (hx >> 31);
    const double   xd = double(x);
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? x : m32::sindf(xd);
    if (ix <= 0x407b53d1u) {
        if (ix <= 0x4016cbe3u) return sign ? -m32::cosdf(xd + m32::kPio2) : m32::cosdf(xd - m32::kPio2);
        return m32::sindf(sign ? -(xd + 2 * m32::kPio2) : -(xd - 2 * m32::kPio2));
    }
    if (ix <= 0x40e231d5u) {
        if (ix <= 0x40afeddfu) return sign ? m32::cosdf(xd + 3 * m32::kPio2) : -m32::cosdf(xd - 3 * m32::kPio2);
        return m32::sindf(sign ? xd + 4 * m32::kPio2 : xd - 4 * m32::kPio2);
    }
    if (ix < 0x4dc90fdbu) {  // sinf.c's general case with __rem_pio2f's medium-size reduction
        double    y;
        const int n = m32::remPio2f(x, y);
        switch (n & 3) {
            case 0: return m32::sindf(y);
            case 1: return m32::cosdf(y);
            case 2: return m32::sindf(-y);
            default: return -m32::cosdf(y);
        }
    }
    return std::sin(x);
}
inline float cosF(float x) {
    const uint32_t hx = m32::bits(x), ix = hx & 0x7fffffffu;
    const bool     sign = 0 != (hx >> 31);
    const double   xd = double(x);
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? 1.f : m32::cosdf(xd);
    if (ix <= 0x407b53d1u) {
        if (ix > 0x4016cbe3u) return -m32::cosdf(sign ? xd + 2 * m32::kPio2 : xd - 2 * m32::kPio2);
        return sign ? m32::sindf(xd + m32::kPio2) : m32::sindf(m32::kPio2 - xd);
    }
    if (ix <= 0x40e231d5u) {
        if (ix > 0x40afeddfu) return m32::cosdf(sign ? xd + 4 * m32::kPio2 : xd - 4 * m32::kPio2);
        return sign ? m32::sindf(-xd - 3 * m32::kPio2) : m32::sindf(xd - 3 * m32::kPio2);
    }
    if (ix < 0x4dc90fdbu) {  // cosf.c's general case
        double    y;
        const int n = m32::remPio2f(x, y);
        switch (n & 3) {
            case 0: return m32::cosdf(y);
            case 1: return m32::sindf(-y);
            case 2: return -m32::cosdf(y);
            default: return m32::sindf(y);
        }
    }
    return std::cos(x);
}
// @tan lowers to compiler-rt's tanf, a port of musl's tanf.c (__tandf in double precision); the disk light's equi-angular
// sampling calls it with angles from atan, |x| < pi/2 (disk.zig:229-234); larger arguments fall back to libm
inline float tandf(double x, bool odd) {
    const double T0 = 0x15554d3418c99f.0p-54, T1 = 0x1112fd38999f72.0p-55, T2 = 0x1b54c91d865afe.0p-57, T3 = 0x191df3908c33ce.0p-58,
                 T4 = 0x185dadfcecf44e.0p-61, T5 = 0x1362b9bf971bcd.0p-59;
    const double z = x * x;
    const double r = T4 + z * T5;
    const double t = T2 + z * T3;
    const double w = z * z;
    const double s = z * x;
    const double u = T0 + z * T1;
    const double r0 = (x + s * u) + (s * w) * (t + w * r);
    return float(odd ? -1.0 / r0 : r0);
}
inline float tanF(float x) {
    const uint32_t hx = m32::bits(x), ix = hx & 0x7fffffffu;
    const bool     sign = 0 != (hx >> 31);
    const double   xd = double(x);
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? x : tandf(xd, false);
    if (ix <= 0x407b53d1u) {
        if (ix <= 0x4016cbe3u) return tandf(sign ? xd + m32::kPio2 : xd - m32::kPio2, true);
        return tandf(sign ? xd + 2 * m32::kPio2 : xd - 2 * m32::kPio2, false);
    }
    return std::tan(x);
}
inline float acosR(float z) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const float p = z * (pS0 + z * (pS1 + z * pS2)), q = 1.f + z * qS1;
    return p / q;
}
inline float acosF(float x) {
    const float    pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const uint32_t hx = m32::bits(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3f800000u) {
        if (ix == 0x3f800000u) return (hx >> 31) ? 2.f * pio2_hi + 0x1p-120f : 0.f;
        return std::nanf("");
    }
    if (ix < 0x3f000000u) {
        if (ix <= 0x32800000u) return pio2_hi + 0x1p-120f;
        return pio2_hi - (x - (pio2_lo - x * acosR(x * x)));
    }
    if (hx >> 31) {
        const float z = (1.f + x) * 0.5f, s = std::sqrt(z), w = acosR(z) * s - pio2_lo;
        return 2.f * (pio2_hi - (s + w));
    }
    const float z = (1.f - x) * 0.5f, s = std::sqrt(z);
    const float df = m32::fromBits(m32::bits(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df), w = acosR(z) * s + c;
    return 2.f * (df + w);
}

// atanf / atan2f with the msun algorithms (Zig std.math.atan / atan2 are ports of musl's s_atanf.c / e_atan2f.c)
inline float atanF(float x) {
    const float    atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float    atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float    aT[5] = {3.3333328366e-01f, -1.9999158382e-01f, 1.4253635705e-01f, -1.0648017377e-01f, 6.1687607318e-02f};
    uint32_t       ix = m32::bits(x);
    const uint32_t sign = ix >> 31;
    ix &= 0x7fffffffu;
    int id;
    if (ix >= 0x4c800000u) {  // |x| >= 2^26
        if (x != x) return x;
        const float z = atanhi[3] + 0x1p-120f;
        return sign ? -z : z;
    }
    if (ix < 0x3ee00000u) {  // |x| < 0.4375
        if (ix < 0x39800000u) return x;  // |x| < 2^-12
        id = -1;
    } else {
        x = std::fabs(x);
        if (ix < 0x3f980000u) {  // |x| < 1.1875
            if (ix < 0x3f300000u) {
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000u) {  // |x| < 2.4375
            id = 2;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * aT[4]));
    const float s2 = w * (aT[1] + w * aT[3]);
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return sign ? -r : r;
}
inline float atan2F(float y, float x) {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    if (x != x || y != y) return x + y;
    uint32_t ix = m32::bits(x), iy = m32::bits(y);
    if (ix == 0x3f800000u) return atanF(y);
    const uint32_t m = ((iy >> 31) & 1u) | ((ix >> 30) & 2u);
    ix &= 0x7fffffffu;
    iy &= 0x7fffffffu;
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi : -pi);
    if (ix == 0) return (m & 1u) ? -pi / 2 : pi / 2;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) return m == 0 ? pi / 4 : m == 1 ? -pi / 4 : m == 2 ? 3 * pi / 4 : -3 * pi / 4;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi : -pi;
    }
    if (ix + (26u << 23) < iy || iy == 0x7f800000u) return (m & 1u) ? -pi / 2 : pi / 2;
    float z;
    if ((m & 2u) && iy + (26u << 23) < ix) z = 0.0f;
    else z = atanF(std::fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// expf with the msun e_expf.c algorithm (Zig's @exp on f32 lowers to compiler-rt's expf, a port of musl's older expf.c):
// argument reduction by k*ln2 in two pieces, degree-2 rational kernel, scaling by 2^k
inline float expF(float x) {
    const float    half[2] = {0.5f, -0.5f};
    const float    ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    const float    P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t       hx = m32::bits(x);
    const uint32_t sign = hx >> 31;
    hx &= 0x7fffffffu;
    if (hx >= 0x42aeac50u) {  // |x| >= 87.33655 or NaN
        if (hx > 0x7f800000u) return x;
        if (hx >= 0x42b17218u && !sign) return x * 0x1p127f;
        if (sign && hx >= 0x42cff1b5u) return 0.f;
    }
    int   k;
    float hi, lo;
    if (hx > 0x3eb17218u) {  // |x| > 0.5 ln2
        if (hx > 0x3f851592u) k = int(invln2 * x + half[sign]);  // |x| > 1.5 ln2
        else k = 1 - int(sign) - int(sign);
        hi = x - float(k) * ln2hi;
        lo = float(k) * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000u) {  // |x| > 2^-14
        k = 0;
        hi = x;
        lo = 0.f;
    } else {
        return 1.f + x;
    }
    const float xx = x * x;
    const float c = x - xx * (P1 + xx * P2);
    const float y = 1.f + (x * c / (2.f - c) - lo + hi);
    if (0 == k) return y;
    return std::ldexp(y, k);
}

constexpr float kPi = 3.14159265358979323846f;
constexpr float kPiInv = float(1.0 / 3.14159265358979323846);  // math.zig:112
constexpr float kDotMin = 0.00001f;                            // safe.zig:5
inline float    clampDot(V4 a, V4 b) { return clampf(dot3(a, b), kDotMin, 1.f); }
inline float    clampAbsDot(V4 a, V4 b) { return clampf(std::fabs(dot3(a, b)), kDotMin, 1.f); }

struct Frame {  // frame.zig:5-69
    V4 x, y, z;
    static Frame init(V4 n) {  // frame.zig:10-13 -> orthonormalBasis3 (vector4.zig:98-110)
        const float sign = std::copysign(1.f, n[2]);
        const float c = -1.f / (sign + n[2]);
        const float d = n[0] * n[1] * c;
        return {mk(1.f + sign * n[0] * n[0] * c, sign * d, -sign * n[0]), mk(d, sign + n[1] * n[1] * c, -n[1]), n};
    }
    V4 frameToWorld(V4 v) const {
        V4 r = S(v[0]) * x;
        r = fma4(S(v[1]), y, r);
        return fma4(S(v[2]), z, r);
    }
    V4 worldToFrame(V4 v) const {
        const V4 t = v * x, b = v * y, n = v * z;
        return mk(t[0] + t[1] + t[2], b[0] + b[1] + b[2], n[0] + n[1] + n[2]);
    }
    float clampNdot(V4 v) const { return clampDot(z, v); }
    float clampAbsNdot(V4 v) const { return clampAbsDot(z, v); }
    float nDot(V4 v) const { return dot3(z, v); }
};

inline V4    reflect3(V4 n, V4 v) { return S(2.f * dot3(v, n)) * n - v; }      // vector4.zig:94-96
inline float solidAngleOfCone(float c) { return (2.f * kPi) * (1.f - c); }  // math.zig:218-220
inline V4    coneUniform(float u0, float u1, float cos_theta_max) {         // sampling.zig:85-94
    const float cos_theta = (1.f - u0) + (u0 * cos_theta_max);
    const float sin_theta = std::sqrt(mx(0.f, 1.f - cos_theta * cos_theta));
    const float phi = u1 * (2.f * kPi);
    const float sin_phi = sinF(phi), cos_phi = cosF(phi);
    return mk(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
}

// sampling.zig:8-54
inline void diskConcentric(float u0, float u1, float& ox, float& oy) {
    const float s0 = (u0 * 2.f) - 1.f, s1 = (u1 * 2.f) - 1.f;
    if (0.f == s0 && 0.f == s1) {
        ox = oy = 0.f;
        return;
    }
    float r, theta;
    if (std::fabs(s0) > std::fabs(s1)) {
        r = s0;
        theta = (kPi / 4.f) * (s1 / s0);
    } else {
        r = s1;
        theta = (kPi / 2.f) - (kPi / 4.f) * (s0 / s1);
    }
    const float sin_theta = sinF(theta), cos_theta = cosF(theta);
    ox = cos_theta * r;
    oy = sin_theta * r;
}
inline V4 hemisphereCosine(float u0, float u1) {
    float x, y;
    diskConcentric(u0, u1, x, y);
    const float z = std::sqrt(mx(0.f, 1.f - x * x - y * y));
    return mk(x, y, z);
}

// encoding.zig:56-108
inline float unorm16ToFloat(uint16_t n) { return float(n) * (1.f / 65535.f); }
inline float snorm16ToFloat(uint16_t n) { return std::fmaf(float(n), 1.f / 32768.f, -1.f); }
inline V4    octDecode(float o0, float o1) {
    V4          v = mk(o0, o1, -1.f + std::fabs(o0) + std::fabs(o1));
    const float t = mx(v[2], 0.f);
    v[0] += v[0] > 0.f ? -t : t;
    v[1] += v[1] > 0.f ? -t : t;
    return normalize3(v);
}

}  // namespace orc
