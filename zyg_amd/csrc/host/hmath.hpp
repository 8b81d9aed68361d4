// Host-side math used while building the compiled scene (BVHs, light tree, transforms).
// Mirrors the arithmetic order of the reference's base/math (explicit FMA only where the
// reference writes @mulAdd); compile with -ffp-contract=off.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace zyg {

struct Vec4f {
    float v[4];
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};

inline Vec4f splat(float s) { return {{s, s, s, s}}; }
inline Vec4f vec(float x, float y, float z, float w = 0.f) { return {{x, y, z, w}}; }

#define ZYG_VOP(op)                                                                              \
    inline Vec4f operator op(const Vec4f& a, const Vec4f& b) {                                   \
        return {{a.v[0] op b.v[0], a.v[1] op b.v[1], a.v[2] op b.v[2], a.v[3] op b.v[3]}};       \
    }
ZYG_VOP(+) ZYG_VOP(-) ZYG_VOP(*) ZYG_VOP(/)
#undef ZYG_VOP
inline Vec4f operator-(const Vec4f& a) { return {{-a.v[0], -a.v[1], -a.v[2], -a.v[3]}}; }
inline Vec4f operator*(float s, const Vec4f& a) { return splat(s) * a; }

// base/math/util.zig:17-29
inline float fmin_(float x, float y) { return x < y ? x : y; }
inline float fmax_(float x, float y) { return y < x ? x : y; }
inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }

inline Vec4f min4(const Vec4f& a, const Vec4f& b) {
    return {{fmin_(a[0], b[0]), fmin_(a[1], b[1]), fmin_(a[2], b[2]), fmin_(a[3], b[3])}};
}
inline Vec4f max4(const Vec4f& a, const Vec4f& b) {
    return {{fmax_(a[0], b[0]), fmax_(a[1], b[1]), fmax_(a[2], b[2]), fmax_(a[3], b[3])}};
}
inline Vec4f fma4(const Vec4f& a, const Vec4f& b, const Vec4f& c) {
    return {{std::fmaf(a[0], b[0], c[0]), std::fmaf(a[1], b[1], c[1]), std::fmaf(a[2], b[2], c[2]),
             std::fmaf(a[3], b[3], c[3])}};
}

// base/math/vector4.zig:37-60
inline float dot3(const Vec4f& a, const Vec4f& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline float squaredLength3(const Vec4f& a) { return dot3(a, a); }
inline float length3(const Vec4f& a) { return std::sqrt(dot3(a, a)); }
inline Vec4f normalize3(const Vec4f& a) { return a / splat(length3(a)); }
inline float hmax3(const Vec4f& a) { return fmax_(a[0], fmax_(a[1], a[2])); }
inline float hmin3(const Vec4f& a) { return fmin_(a[0], fmin_(a[1], a[2])); }
inline bool equal4(const Vec4f& a, const Vec4f& b) {
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}

// base/math/vector4.zig:73-92: tmp2 = fma(b.yzx, a, -(a.yzx * b)); result = tmp2.yzx
inline Vec4f cross3(const Vec4f& a, const Vec4f& b) {
    const Vec4f t0 = {{b[1], b[2], b[0], b[3]}};
    const Vec4f t1 = {{a[1], a[2], a[0], a[3]}};
    const Vec4f t2 = fma4(t0, a, -(t1 * b));
    return {{t2[1], t2[2], t2[0], t2[3]}};
}

inline uint32_t indexMaxComponent3(const Vec4f& v) {
    if (v[0] > v[1]) return v[0] > v[2] ? 0 : 2;
    return v[1] > v[2] ? 1 : 2;
}

// base/math/vector4.zig:98-110
inline void orthonormalBasis3(const Vec4f& n, Vec4f& t, Vec4f& b) {
    const float sign = std::copysign(1.f, n[2]);
    const float c = -1.f / (sign + n[2]);
    const float d = n[0] * n[1] * c;
    t = vec(1.f + sign * n[0] * n[0] * c, sign * d, -sign * n[0]);
    b = vec(d, sign + n[1] * n[1] * c, -n[1]);
}

struct Mat3x3 {
    Vec4f r[3];
};

// base/math/matrix3x3.zig:113-139
inline Vec4f transformVector(const Mat3x3& m, const Vec4f& v) {
    Vec4f result = splat(v[0]) * m.r[0];
    result = fma4(splat(v[1]), m.r[1], result);
    return fma4(splat(v[2]), m.r[2], result);
}
inline Vec4f transformVectorTransposed(const Mat3x3& m, const Vec4f& v) {
    const Vec4f x = v * m.r[0], y = v * m.r[1], z = v * m.r[2];
    return vec(x[0] + x[1] + x[2], y[0] + y[1] + y[2], z[0] + z[1] + z[2]);
}

// base/math/matrix3x3.zig:53-79 (the [0][0] term really is v0*v1*t in the reference)
inline Mat3x3 initRotation(const Vec4f& v, float a) {
    const float c = std::cos(a), s = std::sin(a), t = 1.f - c;
    const float at0 = v[0] * v[1] * t, at1 = v[2] * s;
    const float bt0 = v[0] * v[2] * t, bt1 = v[1] * s;
    const float ct0 = v[1] * v[2] * t, ct1 = v[0] * s;
    Mat3x3 m;
    m.r[0] = vec(c + v[0] * v[1] * t, at0 - at1, bt0 + bt1);
    m.r[1] = vec(at0 + at1, c + v[1] * v[1] * t, ct0 - ct1);
    m.r[2] = vec(bt0 - bt1, ct0 + ct1, c + v[2] * v[2] * t);
    return m;
}

struct Mat4x4 {
    Vec4f r[4];
};

// base/math/matrix4x4.zig:88-121
inline Mat4x4 compose(const Mat3x3& b, const Vec4f& scale, const Vec4f& origin) {
    Mat4x4 m;
    m.r[0] = vec(b.r[0][0] * scale[0], b.r[0][1] * scale[0], b.r[0][2] * scale[0], 0.f);
    m.r[1] = vec(b.r[1][0] * scale[1], b.r[1][1] * scale[1], b.r[1][2] * scale[1], 0.f);
    m.r[2] = vec(b.r[2][0] * scale[2], b.r[2][1] * scale[2], b.r[2][2] * scale[2], 0.f);
    m.r[3] = vec(origin[0], origin[1], origin[2], 1.f);
    return m;
}
inline void decompose(const Mat4x4& m, Mat3x3& basis, Vec4f& scale, Vec4f& origin) {
    const float sx = length3(m.r[0]), sy = length3(m.r[1]), sz = length3(m.r[2]);
    basis.r[0] = m.r[0] / splat(sx);
    basis.r[1] = m.r[1] / splat(sy);
    basis.r[2] = m.r[2] / splat(sz);
    scale = vec(sx, sy, sz, 0.f);
    origin = m.r[3];
}

// base/math/quaternion.zig:11-34
inline Vec4f quaternionFromMat3x3(const Mat3x3& m) {
    float t;
    Vec4f q;
    if (m.r[2][2] < 0.f) {
        if (m.r[0][0] > m.r[1][1]) {
            t = 1.f + m.r[0][0] - m.r[1][1] - m.r[2][2];
            q = vec(t, m.r[0][1] + m.r[1][0], m.r[2][0] + m.r[0][2], m.r[2][1] - m.r[1][2]);
        } else {
            t = 1.f - m.r[0][0] + m.r[1][1] - m.r[2][2];
            q = vec(m.r[0][1] + m.r[1][0], t, m.r[1][2] + m.r[2][1], m.r[0][2] - m.r[2][0]);
        }
    } else {
        if (m.r[0][0] < -m.r[1][1]) {
            t = 1.f - m.r[0][0] - m.r[1][1] + m.r[2][2];
            q = vec(m.r[2][0] + m.r[0][2], m.r[1][2] + m.r[2][1], t, m.r[1][0] - m.r[0][1]);
        } else {
            t = 1.f + m.r[0][0] + m.r[1][1] + m.r[2][2];
            q = vec(m.r[2][1] - m.r[1][2], m.r[0][2] - m.r[2][0], m.r[1][0] - m.r[0][1], t);
        }
    }
    return q * splat(0.5f / std::sqrt(t));
}

// base/math/quaternion.zig:72-116
inline Mat3x3 quaternionToMat3x3(const Vec4f& q) {
    const Vec4f tq = q + q;
    const float xy = tq[1] * q[0], xz = tq[2] * q[0], yz = tq[1] * q[2];
    const Vec4f w = tq * splat(q[3]);
    const Vec4f a = {{q[3], q[0], q[3], q[1]}};
    const Vec4f b = {{q[1], q[2], q[0], q[2]}};
    const Vec4f t = (a + b) * (a - b);
    Mat3x3 m;
    m.r[0] = vec(t[0] + t[1], xy - w[2], xz + w[1]);
    m.r[1] = vec(xy + w[2], t[2] + t[3], yz - w[0]);
    m.r[2] = vec(xz - w[1], yz + w[0], t[2] - t[3]);
    return m;
}

struct Transformation {  // base/math/transformation.zig: position, scale, rotation quaternion
    Vec4f position, scale, rotation;
};

struct AABB {
    Vec4f b[2];
};

inline Vec4f quaternionMul(const Vec4f& a, const Vec4f& b) {  // quaternion.zig:208-215
    return vec((a[3] * b[0] + a[0] * b[3]) + (a[1] * b[2] - a[2] * b[1]), (a[3] * b[1] + a[1] * b[3]) + (a[2] * b[0] - a[0] * b[2]),
               (a[3] * b[2] + a[2] * b[3]) + (a[0] * b[1] - a[1] * b[0]), (a[3] * b[3] - a[0] * b[0]) - (a[1] * b[1] + a[2] * b[2]));
}
// Transformation.transform (transformation.zig:30-36): `child` in the space of `parent`
inline Transformation transformBy(const Transformation& parent, const Transformation& child) {
    const Mat4x4 m = compose(quaternionToMat3x3(parent.rotation), parent.scale, parent.position);
    Vec4f        p = splat(child.position[0]) * m.r[0] + splat(child.position[1]) * m.r[1] + splat(child.position[2]) * m.r[2] + m.r[3];
    p[3] = 0.f;
    return {p, parent.scale * child.scale, quaternionMul(child.rotation, parent.rotation)};
}

inline AABB emptyAabb() { return {{splat(FLT_MAX), splat(-FLT_MAX)}}; }
inline Vec4f aabbExtent(const AABB& a) { return a.b[1] - a.b[0]; }
inline Vec4f aabbPosition(const AABB& a) { return splat(0.5f) * (a.b[0] + a.b[1]); }
inline float aabbSurfaceArea(const AABB& a) {
    const Vec4f d = a.b[1] - a.b[0];
    return 2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
}
inline void aabbMerge(AABB& a, const AABB& o) {
    a.b[0] = min4(a.b[0], o.b[0]);
    a.b[1] = max4(a.b[1], o.b[1]);
}
inline AABB aabbIntersection(const AABB& a, const AABB& o) { return {{max4(a.b[0], o.b[0]), min4(a.b[1], o.b[1])}}; }
inline bool aabbCovers(const AABB& a, const AABB& o) {
    return a.b[0][0] <= o.b[0][0] && a.b[0][1] <= o.b[0][1] && a.b[0][2] <= o.b[0][2] && a.b[1][0] >= o.b[1][0] &&
           a.b[1][1] >= o.b[1][1] && a.b[1][2] >= o.b[1][2];
}
inline void aabbCacheRadius(AABB& a) {  // aabb.zig:141-144
    a.b[0][3] = 0.f;
    a.b[1][3] = 0.5f * length3(aabbExtent(a));
}
// aabb.zig:146-165
inline AABB aabbTransform(const AABB& a, const Mat4x4& m) {
    const Vec4f xa = m.r[0] * splat(a.b[0][0]), xb = m.r[0] * splat(a.b[1][0]);
    const Vec4f ya = m.r[1] * splat(a.b[0][1]), yb = m.r[1] * splat(a.b[1][1]);
    const Vec4f za = m.r[2] * splat(a.b[0][2]), zb = m.r[2] * splat(a.b[1][2]);
    return {{min4(xa, xb) + min4(ya, yb) + min4(za, zb) + m.r[3], max4(xa, xb) + max4(ya, yb) + max4(za, zb) + m.r[3]}};
}

// base/math/cone.zig:8-44
inline Vec4f coneMerge(Vec4f a, Vec4f b) {
    if (equal4(splat(1.f), a)) return b;
    if (equal4(a, b)) return a;
    float a_angle = std::acos(a[3]);
    float b_angle = std::acos(b[3]);
    if (b_angle > a_angle) {
        std::swap(a, b);
        std::swap(a_angle, b_angle);
    }
    const float d_angle = std::acos(clamp_(dot3(a, b), -1.f, 1.f));
    if (fmin_(d_angle + b_angle, float(M_PI)) <= a_angle) return a;
    const float o_angle = (a_angle + d_angle + b_angle) / 2.f;
    if (o_angle >= float(M_PI)) return vec(a[0], a[1], a[2], -1.f);
    const float r_angle = o_angle - a_angle;
    const Mat3x3 rot = initRotation(normalize3(cross3(a, b)), r_angle);
    const Vec4f axis = normalize3(transformVector(rot, a));
    return vec(axis[0], axis[1], axis[2], std::cos(o_angle));
}

// base/encoding.zig:30-108
inline uint16_t floatToUnorm16(float x) { return uint16_t(std::fmaf(x, 65535.f, 0.5f)); }
inline uint16_t floatToSnorm16(float x) { return uint16_t((x + 1.f) * (x > 0.f ? 32767.5f : 32768.f)); }
inline void octEncode(const Vec4f& v, float o[2]) {
    const float inorm = 1.f / (std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]));
    const float t = fmax_(v[2], 0.f);
    o[0] = (v[0] + (v[0] > 0.f ? t : -t)) * inorm;
    o[1] = (v[1] + (v[1] > 0.f ? t : -t)) * inorm;
}

// base/spectrum/aces.zig:9-27
inline Vec4f sRGBtoAP1(const Vec4f& c) {
    return vec(0.61309732f, 0.07019422f, 0.02061560f) * splat(c[0]) + vec(0.33952285f, 0.91635557f, 0.10956983f) * splat(c[1]) +
           vec(0.04737928f, 0.01345259f, 0.86981512f) * splat(c[2]);
}

}  // namespace zyg
