// Device math for the gfx950 kernels.  3-lane vectors in registers (the reference's 4th SIMD lane never
// carries information through dot3/cross3/normalize3); arithmetic order and explicit-FMA placement follow
// the reference's base/math (file:line cited per function).  Built with -ffp-contract=off: only the
// __fmaf_rn sites below fuse, exactly where the reference writes @mulAdd.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#define ZD __device__ __forceinline__
#ifndef ZYG_FAST
#define ZYG_FAST 0  // 1 in the relaxed-arithmetic kernel units (zyg_amd/build.py)
#endif

namespace zd {

struct f3 {
    float x, y, z;
};

ZD f3 mk3(float x, float y, float z) { return {x, y, z}; }
ZD f3 S3(float s) { return {s, s, s}; }
ZD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
ZD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
ZD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
ZD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
ZD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
ZD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
ZD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
ZD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

// `c ? a : b` between two lvalues is itself an lvalue: the compiler selects an ADDRESS and loads through it, and an aggregate whose
// members are addressed by a variable is never split into registers (Sampler::buf kept every PathHead in scratch that way, the
// medium stack's entries their record, the cube's axis the whole Fragment).  pick() selects between values.
template <typename T>
ZD T pick(bool c, T a, T b) {
    return c ? T(a) : T(b);
}

// base/math/util.zig:17-29 (x86 forms)
// The relaxed units take the hardware's v_min_f32 / v_max_f32 (one instruction, and v_min3 / v_max3 for nested ones) where the x86
// form compiles to v_cmp + v_cndmask with the wait states gfx950 wants between a VALU write of vcc and its use: the two differ
// only where the second operand is a NaN.  -DZYG_FAST_MINMAX=0: the x86 form in the relaxed units too.
#ifndef ZYG_FAST_MINMAX
#define ZYG_FAST_MINMAX ZYG_FAST
#endif
#if ZYG_FAST_MINMAX
ZD float mn(float x, float y) { return __builtin_fminf(x, y); }
ZD float mx(float x, float y) { return __builtin_fmaxf(x, y); }
#else
ZD float mn(float x, float y) { return x < y ? x : y; }
ZD float mx(float x, float y) { return y < x ? x : y; }
#endif
ZD float clampf(float x, float lo, float hi) { return mn(mx(x, lo), hi); }
ZD float saturate(float x) { return clampf(x, 0.f, 1.f); }
ZD float stdClamp(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); }  // std.math.clamp
ZD f3    min3v(f3 a, f3 b) { return {mn(a.x, b.x), mn(a.y, b.y), mn(a.z, b.z)}; }
ZD f3    max3v(f3 a, f3 b) { return {mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z)}; }
ZD f3    fma3(f3 a, f3 b, f3 c) { return {__fmaf_rn(a.x, b.x, c.x), __fmaf_rn(a.y, b.y, c.y), __fmaf_rn(a.z, b.z, c.z)}; }

// base/math/vector4.zig:37-60
ZD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ZD float squaredLength3(f3 a) { return dot3(a, a); }
ZD float length3(f3 a) { return sqrtf(dot3(a, a)); }
ZD float distance3(f3 a, f3 b) { return length3(a - b); }
ZD f3    normalize3(f3 a) { return a / S3(length3(a)); }
ZD float hmax3(f3 a) { return mx(a.x, mx(a.y, a.z)); }
ZD float hmin3(f3 a) { return mn(a.x, mn(a.y, a.z)); }
ZD float average3(f3 a) { return (a.x + a.y + a.z) / 3.f; }

// base/math/vector4.zig:73-92: tmp2 = fma(b.yzx, a, -(a.yzx * b)); result = tmp2.yzx
ZD f3 cross3(f3 a, f3 b) {
    const float t0 = __fmaf_rn(b.y, a.x, -(a.y * b.x));
    const float t1 = __fmaf_rn(b.z, a.y, -(a.z * b.y));
    const float t2 = __fmaf_rn(b.x, a.z, -(a.x * b.z));
    return {t1, t2, t0};
}

ZD f3    gramSchmidt(f3 v, f3 w) { return fma3(S3(-dot3(v, w)), w, v); }  // vector4.zig:121
ZD float lerpf(float a, float b, float t) { return __fmaf_rn(1.f - t, a, t * b); }  // util.zig:3-16
ZD f3    lerp3(f3 a, f3 b, f3 t) { return fma3(S3(1.f) - t, a, t * b); }


// ZYG_FAST (the relaxed-arithmetic units, kernel_table.hpp): the contract is RMS <= 1e-3 against the CPU reference, not
// bit-identity, so these units take OCML's f32 functions (no f64 kernels), and the build gives them native divide / sqrt
// (-fno-hip-fp32-correctly-rounded-divide-sqrt) and fused multiply-adds (-ffp-contract=fast).  The precise units below
// stay the verification mode.
#if ZYG_FAST
ZD float sinF(float x) { return sinf(x); }
ZD float cosF(float x) { return cosf(x); }
// normalised vectors are unit length only to an ulp or two here (native rsqrt / rcp): a dot product of 1 + 1e-7 must not
// turn into a NaN solid angle
ZD float acosF(float x) { return acosf(fminf(fmaxf(x, -1.f), 1.f)); }
ZD float tanF(float x) { return tanf(x); }
ZD float atanF(float x) { return atanf(x); }
ZD float atan2F(float y, float x) { return atan2f(y, x); }
ZD float expF(float x) { return expf(x); }
ZD float acosR(float z) {  // only the unit-test hook calls it directly
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    return (z * (pS0 + z * (pS1 + z * pS2))) / (1.f + z * qS1);
}
#else
// The restated functions are expanded at every call site by default (ZD).  -DZYG_LIBM_CALLS=1 makes them functions that are called:
// the kernels that know every shape, light and closure hold several hundred expansions of them (wf_scatter<.., MESHL, POOL>: 10 K
// f64 instructions and as many literal moves of 98 K), far past the instruction cache.
#ifndef ZYG_LIBM_CALLS
#define ZYG_LIBM_CALLS 0
#endif
#if ZYG_LIBM_CALLS
#define ZD_LIBM __device__ __noinline__
#else
#define ZD_LIBM ZD
#endif
// sin / cos / acos with the reference's algorithms (Zig std.math.acos = msun e_acosf.c; compiler-rt sinf/cosf =
// musl sinf.c/cosf.c with the double-precision k_sinf/k_cosf kernels) so that device results are reproducible
// bit for bit on the host.  Beyond 9*pi/4 musl's general case with the medium-size argument reduction (up to 2^28 * pi/2).
ZD float sindf(double x) {
    const double S1 = -0x1.5555554cbac77p-3, S2 = 0x1.11110896efbb2p-7, S3 = -0x1.a00f9e2cae774p-13, S4 = 0x1.6cd878c3b46a7p-19;
    const double z = x * x, w = z * z, r = S3 + z * S4, s = z * x;
    return float((x + s * (S1 + z * S2)) + s * w * r);
}
ZD float cosdf(double x) {
    const double C0 = -0x1.ffffffd0c5e81p-2, C1 = 0x1.55553e1053a42p-5, C2 = -0x1.6c087e80f1e27p-10, C3 = 0x1.99342e0ee5069p-16;
    const double z = x * x, w = z * z, r = C2 + z * C3;
    return float(((1.0 + z * C0) + w * C1) + (w * z) * r);
}
constexpr double kPio2d = 1.57079632679489661923;
// musl __rem_pio2f, medium size (|x| < 2^28 * pi/2): x = n * pi/2 + y, |y| <= pi/4, in double precision
ZD int remPio2f(float x, double& y) {
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
ZD_LIBM float sinF(float x) {
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    const bool     sign = 0 != (hx >> 31);
    const double   xd = double(x);
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? x : sindf(xd);
    if (ix <= 0x407b53d1u) {
        if (ix <= 0x4016cbe3u) return sign ? -cosdf(xd + kPio2d) : cosdf(xd - kPio2d);
        return sindf(sign ? -(xd + 2 * kPio2d) : -(xd - 2 * kPio2d));
    }
    if (ix <= 0x40e231d5u) {
        if (ix <= 0x40afeddfu) return sign ? cosdf(xd + 3 * kPio2d) : -cosdf(xd - 3 * kPio2d);
        return sindf(sign ? xd + 4 * kPio2d : xd - 4 * kPio2d);
    }
    if (ix < 0x4dc90fdbu) {  // sinf.c's general case with __rem_pio2f's medium-size reduction
        double    y;
        const int n = remPio2f(x, y);
        switch (n & 3) {
            case 0: return sindf(y);
            case 1: return cosdf(y);
            case 2: return sindf(-y);
            default: return -cosdf(y);
        }
    }
    return sinf(x);
}
ZD_LIBM float cosF(float x) {
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    const bool     sign = 0 != (hx >> 31);
    const double   xd = double(x);
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? 1.f : cosdf(xd);
    if (ix <= 0x407b53d1u) {
        if (ix > 0x4016cbe3u) return -cosdf(sign ? xd + 2 * kPio2d : xd - 2 * kPio2d);
        return sign ? sindf(xd + kPio2d) : sindf(kPio2d - xd);
    }
    if (ix <= 0x40e231d5u) {
        if (ix > 0x40afeddfu) return cosdf(sign ? xd + 4 * kPio2d : xd - 4 * kPio2d);
        return sign ? sindf(-xd - 3 * kPio2d) : sindf(xd - 3 * kPio2d);
    }
    if (ix < 0x4dc90fdbu) {  // cosf.c's general case
        double    y;
        const int n = remPio2f(x, y);
        switch (n & 3) {
            case 0: return cosdf(y);
            case 1: return sindf(-y);
            case 2: return -cosdf(y);
            default: return sindf(y);
        }
    }
    return cosf(x);
}
ZD float acosR(float z) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const float p = z * (pS0 + z * (pS1 + z * pS2)), q = 1.f + z * qS1;
    return p / q;
}
ZD_LIBM float acosF(float x) {
    const float    pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3f800000u) {
        if (ix == 0x3f800000u) return (hx >> 31) ? 2.f * pio2_hi + 0x1p-120f : 0.f;
        return __uint_as_float(0x7fc00000u);
    }
    if (ix < 0x3f000000u) {
        if (ix <= 0x32800000u) return pio2_hi + 0x1p-120f;
        return pio2_hi - (x - (pio2_lo - x * acosR(x * x)));
    }
    if (hx >> 31) {
        const float z = (1.f + x) * 0.5f, s = sqrtf(z), w = acosR(z) * s - pio2_lo;
        return 2.f * (pio2_hi - (s + w));
    }
    const float z = (1.f - x) * 0.5f, s = sqrtf(z);
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df), w = acosR(z) * s + c;
    return 2.f * (df + w);
}

// @tan -> compiler-rt tanf = musl tanf.c (__tandf in double precision); called with angles from atan, |x| < pi/2
// (disk.zig:229-234); beyond 5 pi / 4 the device has no libm fallback and returns tanf of the reduced argument's class
ZD float tandf(double x, bool odd) {
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
ZD_LIBM float tanF(float x) {
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    const bool     sign = 0 != (hx >> 31);
    const double   xd = double(x);
    const double   pio2 = 1.57079632679489661923;
    if (ix <= 0x3f490fdau) return ix < 0x39800000u ? x : tandf(xd, false);
    if (ix <= 0x4016cbe3u) return tandf(sign ? xd + pio2 : xd - pio2, true);
    return tandf(sign ? xd + 2 * pio2 : xd - 2 * pio2, false);
}

// atanf / atan2f with the msun algorithms (Zig std.math.atan / atan2 are ports of musl's s_atanf.c / e_atan2f.c)
ZD_LIBM float atanF(float x) {
    const float    atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float    atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float    aT[5] = {3.3333328366e-01f, -1.9999158382e-01f, 1.4253635705e-01f, -1.0648017377e-01f, 6.1687607318e-02f};
    uint32_t       ix = __float_as_uint(x);
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
        x = fabsf(x);
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
ZD_LIBM float atan2F(float y, float x) {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    if (x != x || y != y) return x + y;
    uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
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
    else z = atanF(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// expf with the msun e_expf.c algorithm (Zig's @exp on f32 lowers to compiler-rt's expf, a port of musl's older expf.c):
// argument reduction by k*ln2 in two pieces, degree-2 rational kernel, scaling by 2^k
ZD_LIBM float expF(float x) {
    const float    ln2hi = 6.9314575195e-1f, ln2lo = 1.4286067653e-6f, invln2 = 1.4426950216e+0f;
    const float    P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t       hx = __float_as_uint(x);
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
        if (hx > 0x3f851592u) k = int(invln2 * x + (sign ? -0.5f : 0.5f));  // |x| > 1.5 ln2
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
    return ldexpf(y, k);
}

#endif  // ZYG_FAST

// The square root of a difference that exact arithmetic keeps >= 0 (1 - cos^2, r^2 - y^2 ...).  With native reciprocals and
// fused multiply-adds the difference can come out a few ulps below zero where the precise units' stays at or above it; the
// relaxed units clamp instead of returning NaN (one NaN sample blackens its pixel for good).
#if ZYG_FAST
ZD float sqrtD(float x) { return sqrtf(fmaxf(x, 0.f)); }
#else
ZD float sqrtD(float x) { return sqrtf(x); }
#endif

constexpr float kPi = 3.14159265358979323846f;
constexpr float kPiInv = float(1.0 / 3.14159265358979323846);
constexpr float kDotMin = 0.00001f;  // safe.zig:5
ZD float clampDot(f3 a, f3 b) { return clampf(dot3(a, b), kDotMin, 1.f); }
ZD float clampAbsDot(f3 a, f3 b) { return clampf(fabsf(dot3(a, b)), kDotMin, 1.f); }

struct Frame {  // frame.zig:5-69
    f3 x, y, z;
    ZD static Frame init(f3 n) {  // frame.zig:10-13 -> orthonormalBasis3 (vector4.zig:98-110)
        const float sign = copysignf(1.f, n.z);
        const float c = -1.f / (sign + n.z);
        const float d = n.x * n.y * c;
        return {{1.f + sign * n.x * n.x * c, sign * d, -sign * n.x}, {d, sign + n.y * n.y * c, -n.y}, n};
    }
    ZD f3 frameToWorld(f3 v) const {
        f3 r = S3(v.x) * x;
        r = fma3(S3(v.y), y, r);
        return fma3(S3(v.z), z, r);
    }
    ZD f3 worldToFrame(f3 v) const {
        const f3 t = v * x, b = v * y, n = v * z;
        return {t.x + t.y + t.z, b.x + b.y + b.z, n.x + n.y + n.z};
    }
    ZD float clampNdot(f3 v) const { return clampDot(z, v); }
    ZD float clampAbsNdot(f3 v) const { return clampAbsDot(z, v); }
};

// sampling.zig:8-54
ZD void diskConcentric(float u0, float u1, float& ox, float& oy) {
    const float s0 = (u0 * 2.f) - 1.f, s1 = (u1 * 2.f) - 1.f;
    if (0.f == s0 && 0.f == s1) {
        ox = oy = 0.f;
        return;
    }
    float r, theta;
    if (fabsf(s0) > fabsf(s1)) {
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
ZD f3 hemisphereCosine(float u0, float u1) {
    float x, y;
    diskConcentric(u0, u1, x, y);
    const float z = sqrtf(mx(0.f, 1.f - x * x - y * y));
    return {x, y, z};
}

ZD f3    reflect3(f3 n, f3 v) { return S3(2.f * dot3(v, n)) * n - v; }        // vector4.zig:94-96
ZD float solidAngleOfCone(float c) { return (2.f * kPi) * (1.f - c); }  // math.zig:218-220
ZD f3    coneUniform(float u0, float u1, float cos_theta_max) {         // sampling.zig:85-94
    const float cos_theta = (1.f - u0) + (u0 * cos_theta_max);
    const float sin_theta = sqrtf(mx(0.f, 1.f - cos_theta * cos_theta));
    const float phi = u1 * (2.f * kPi);
    const float sin_phi = sinF(phi), cos_phi = cosF(phi);
    return {cos_phi * sin_theta, sin_phi * sin_theta, cos_theta};
}

// encoding.zig:56-108
ZD float unorm16ToFloat(uint32_t n) { return float(n) * (1.f / 65535.f); }
ZD float snorm16ToFloat(uint32_t n) { return __fmaf_rn(float(n), 1.f / 32768.f, -1.f); }
ZD f3    octDecode(float o0, float o1) {
    f3          v = {o0, o1, -1.f + fabsf(o0) + fabsf(o1)};
    const float t = mx(v.z, 0.f);
    v.x += v.x > 0.f ? -t : t;
    v.y += v.y > 0.f ? -t : t;
    return normalize3(v);
}

// rotation rows + scale + position of a prop (composed_transformation.zig:11-130)
struct Trafo {
    f3 r0, r1, r2;
    f3 scale;
    f3 position;

    ZD f3 transformVector(f3 v) const {  // matrix3x3.zig:113-126
        f3 res = S3(v.x) * r0;
        res = fma3(S3(v.y), r1, res);
        return fma3(S3(v.z), r2, res);
    }
    ZD f3 transformVectorTransposed(f3 v) const {  // matrix3x3.zig:128-139
        const f3 x = v * r0, y = v * r1, z = v * r2;
        return {x.x + x.y + x.z, y.x + y.y + y.z, z.x + z.y + z.z};
    }
    ZD f3 objectToWorldVector(f3 v) const {
        const f3 a = r0 * S3(scale.x), b = r1 * S3(scale.y), c = r2 * S3(scale.z);
        f3       res = S3(v.x) * a;
        res = fma3(S3(v.y), b, res);
        return fma3(S3(v.z), c, res);
    }
    ZD f3 objectToWorldPoint(f3 p) const { return objectToWorldVector(p) + position; }
    ZD f3 objectToWorldNormal(f3 n) const { return transformVector(n); }
    ZD f3 frameToWorldPoint(f3 p) const { return transformVector(p) + position; }
    ZD f3 worldToObjectVector(f3 v) const { return transformVectorTransposed(v) / scale; }
    ZD f3 worldToObjectPoint(f3 p) const { return worldToObjectVector(p - position); }
    ZD f3 worldToObjectNormal(f3 n) const { return transformVectorTransposed(n); }
    ZD f3 worldToFramePoint(f3 p) const { return transformVectorTransposed(p - position); }
};

struct Ray {
    f3       origin, direction, inv_direction;
    float    min_t, max_t;
    uint32_t time;  // Probe.time as ticks since the frame's start: selects an animated prop's transformation (0 elsewhere)
    ZD static Ray init(f3 o, f3 d, float mint, float maxt) {  // ray.zig:11-20
        Ray r;
        r.time = 0;
        r.origin = o;
        r.direction = d;
        r.inv_direction = S3(1.f) / d;
        r.min_t = mint;
        r.max_t = maxt;
        return r;
    }
    ZD f3 point(float t) const { return fma3(S3(t), direction, origin); }  // ray.zig:27-29
};

constexpr float kRayMaxT = 2.14748313e+09f;  // ray_offset.zig:5

// ray_offset.zig:14-31
ZD float offsetRay1(float p, float n) {
    const float   origin = 1.f / 32.f, float_scale = 1.f / 65536.f, int_scale = 256.f;
    const int32_t of_i = int32_t(int_scale * n);
    const int32_t p_ii = __float_as_int(p);
    const float   p_in = __int_as_float(int32_t(uint32_t(p_ii) - uint32_t(of_i)));
    const float   p_ip = __int_as_float(int32_t(uint32_t(p_ii) + uint32_t(of_i)));
    const float   p_i = p < 0.f ? p_in : p_ip;
    const float   mad = __fmaf_rn(float_scale, n, p);
    return fabsf(p) < origin ? mad : p_i;
}
ZD f3 offsetRay(f3 p, f3 n) { return {offsetRay1(p.x, n.x), offsetRay1(p.y, n.y), offsetRay1(p.z, n.z)}; }
ZD float offsetF(float t) {  // ray_offset.zig:29-31
    const float origin = 1.f / 32.f, float_scale = 1.f / 65536.f;
    return t < origin ? t + float_scale : __uint_as_float(__float_as_uint(t) + 256u);
}

}  // namespace zd
