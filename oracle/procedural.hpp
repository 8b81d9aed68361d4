// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp).
//
// Procedural textures, restated from the reference:
//   Procedural.sample2D_1 / _2 / _3      src/core/texture/procedural.zig:82-130
//   Checker                              src/core/texture/procedural_checker.zig:46-84
//   Noise                                src/core/texture/procedural_noise.zig:44-142
//   gradient / cellular noise            src/core/texture/noise/perlin.zig:11-219, worley.zig:10-168
//   ChannelMix / Max / Mul / Mix / DetailNormal   src/core/texture/procedural_*.zig
//   the texture-space footprint of a pixel        src/core/scene/context.zig:139-239, camera/camera_perspective.zig:223-259,
//                                                 shape/shape.zig:565-575, rectangle.zig:613-622, triangle_mesh.zig:751-760,
//                                                 triangle.zig:102-131, material_helper.zig:96-122
// Parity unpinned: the reference ships no vectors for these and cannot be run here (no Zig toolchain).
#pragma once

#include "scene.hpp"

namespace orc {

inline V4 xyz0(V4 a) { return mk(a[0], a[1], a[2], 0.f); }
inline V4 abs4(V4 a) { return mk(std::fabs(a[0]), std::fabs(a[1]), std::fabs(a[2]), std::fabs(a[3])); }

// ---- noise/perlin.zig
namespace noise {

inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

inline uint32_t bjfinal(uint32_t a, uint32_t b, uint32_t c) {  // perlin.zig:186-207
    c ^= b;
    c -= rotl(b, 14);
    a ^= c;
    a -= rotl(c, 11);
    b ^= a;
    b -= rotl(a, 25);
    c ^= b;
    c -= rotl(b, 16);
    a ^= c;
    a -= rotl(c, 4);
    b ^= a;
    b -= rotl(a, 14);
    c ^= b;
    c -= rotl(b, 24);
    return c;
}

inline float fade(float t) { return t * t * t * std::fma(t, std::fma(t, 6.f, -15.f), 10.f); }  // perlin.zig:64-70

inline float gradient2(uint32_t hash, float x, float y) {  // gradient2v, perlin.zig:86-93
    const uint32_t h = hash & 7u;
    const float    u = h < 4u ? x : y;
    const float    v = 2.f * (h < 4u ? y : x);
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}

inline float gradient3(uint32_t hash, float x, float y, float z) {  // gradient3v, perlin.zig:104-116
    const uint32_t h = hash & 15u;
    const float    u = h < 8u ? x : y;
    const float    t = (12u == h || 14u == h) ? x : z;
    const float    v = h < 4u ? y : t;
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}

inline int32_t zigMod(int32_t a, int32_t n) {  // @mod: the result has the divisor's sign
    const int32_t r = a % n;
    return (r != 0 && ((r < 0) != (n < 0))) ? r + n : r;
}

inline float perlin2D(float px, float py, float period_x, float period_y) {  // perlin.zig:11-39
    const float   flx = std::floor(px), fly = std::floor(py);
    const float   fx = px - flx, fy = py - fly;
    const int32_t X = int32_t(flx), Y = int32_t(fly);
    const float   u = fade(fx), v = fade(fy);
    int32_t       x0 = X, x1 = int32_t(uint32_t(X) + 1u), y0 = Y, y1 = int32_t(uint32_t(Y) + 1u);
    const int32_t perx = int32_t(period_x), pery = int32_t(period_y);
    if (perx > 0) x0 = zigMod(x0, perx), x1 = zigMod(x1, perx);
    if (pery > 0) y0 = zigMod(y0, pery), y1 = zigMod(y1, pery);
    const uint32_t start = 0xdeadbeefu + (2u << 2) + 13u;  // hash2v, perlin.zig:136-142
    auto hash = [&](int32_t x, int32_t y) { return bjfinal(start + uint32_t(x), start + uint32_t(y), start); };
    const float c[4] = {gradient2(hash(x0, y0), fx, fy), gradient2(hash(x1, y0), fx - 1.f, fy), gradient2(hash(x0, y1), fx, fy - 1.f),
                        gradient2(hash(x1, y1), fx - 1.f, fy - 1.f)};
    const float _s = 1.f - u, _t = 1.f - v;  // math.bilinear (math.zig:172-179)
    return 0.6616f * (_t * (_s * c[0] + u * c[1]) + v * (_s * c[2] + u * c[3]));
}

inline float perlin3D(V4 p) {  // perlin.zig:41-61
    const float   fl[3] = {std::floor(p[0]), std::floor(p[1]), std::floor(p[2])};
    const float   f[3] = {p[0] - fl[0], p[1] - fl[1], p[2] - fl[2]};
    const int32_t P[3] = {int32_t(fl[0]), int32_t(fl[1]), int32_t(fl[2])};
    const float   u = fade(f[0]), v = fade(f[1]), w = fade(f[2]);
    const uint32_t start = 0xdeadbeefu + (3u << 2) + 13u;  // hash3v, perlin.zig:154-161
    auto hash = [&](int32_t x, int32_t y, uint32_t dz) {
        return bjfinal(start + uint32_t(x), start + uint32_t(y), start + uint32_t(P[2]) + dz);
    };
    float c0[4], c1[4];
    for (int k = 0; k < 4; ++k) {
        const int32_t x = P[0] + (k & 1), y = P[1] + (k >> 1);
        const float   gx = f[0] - float(k & 1), gy = f[1] - float(k >> 1);
        c0[k] = gradient3(hash(x, y, 0u), gx, gy, f[2]);
        c1[k] = gradient3(hash(x, y, 1u), gx, gy, f[2] - 1.f);
    }
    const float _s = 1.f - u, _t = 1.f - v;
    const float r0 = _t * (_s * c0[0] + u * c0[1]) + v * (_s * c0[2] + u * c0[3]);
    const float r1 = _t * (_s * c1[0] + u * c1[1]) + v * (_s * c1[2] + u * c1[3]);
    return 0.9820f * lerpf(r0, r1, w);
}

// ---- noise/worley.zig
inline float to01(uint32_t bits) {  // vecTo01, worley.zig:157-168
    bits = (bits & 0x007FFFFFu) | 0x3F800000u;
    float f;
    std::memcpy(&f, &bits, 4);
    return f - 1.f;
}

inline float worley2D(float px, float py, float jitter) {  // worley.zig:10-38, 79-84, 93-99, 109-117, 129-143
    const float   flx = std::floor(px), fly = std::floor(py);
    const float   lx = px - flx, ly = py - fly;
    const int32_t X = int32_t(flx), Y = int32_t(fly);
    float         min_dist = 1.0e6f;
    for (int32_t x = -1; x <= 1; ++x)
        for (int32_t y = -1; y <= 1; ++y) {
            // cellNoise2(@floatFromInt(xy + offset)): through f32 and back
            const float cx = float(x + X), cy = float(y + Y);
            uint32_t    v0 = uint32_t(int32_t(std::floor(cx))) * 1664525u + 1013904223u;
            uint32_t    v1 = uint32_t(int32_t(std::floor(cy))) * 1664525u + 1013904223u;
            v0 += v1 * 1664525u;
            v1 += v0 * 1664525u;
            v0 ^= v0 >> 16;
            v1 ^= v1 >> 16;
            v0 += v1 * 1664525u;
            v1 += v0 * 1664525u;
            v0 ^= v0 >> 16;
            v1 ^= v1 >> 16;
            const float ox = std::fma(to01(v0) - 0.5f, jitter, 0.5f), oy = std::fma(to01(v1) - 0.5f, jitter, 0.5f);
            const float dx = (float(x) + ox) - lx, dy = (float(y) + oy) - ly;
            const float dist = dx * dx + dy * dy;
            if (dist < min_dist) min_dist = dist;
        }
    return min_dist;
}

inline float worley3D(V4 p, float jitter) {  // worley.zig:40-64, 86-91, 101-107, 119-127, 145-155
    const float   fl[3] = {std::floor(p[0]), std::floor(p[1]), std::floor(p[2])};
    const float   l[3] = {p[0] - fl[0], p[1] - fl[1], p[2] - fl[2]};
    const int32_t P[3] = {int32_t(fl[0]), int32_t(fl[1]), int32_t(fl[2])};
    float         min_dist = 1.0e6f;
    for (int32_t x = -1; x <= 1; ++x)
        for (int32_t y = -1; y <= 1; ++y)
            for (int32_t z = -1; z <= 1; ++z) {
                const int32_t c[3] = {x, y, z};
                uint32_t      v[3];
                for (int k = 0; k < 3; ++k) v[k] = uint32_t(int32_t(std::floor(float(c[k] + P[k])))) * 1664525u + 1013904223u;
                v[0] += v[1] * v[2];
                v[1] += v[2] * v[0];
                v[2] += v[0] * v[1];
                for (int k = 0; k < 3; ++k) v[k] ^= v[k] >> 16;
                v[0] += v[1] * v[2];
                v[1] += v[2] * v[0];
                v[2] += v[0] * v[1];
                float dist = 0.f, d[3];
                for (int k = 0; k < 3; ++k) d[k] = (float(c[k]) + std::fma(to01(v[k]) - 0.5f, jitter, 0.5f)) - l[k];
                dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];  // math.dot3
                if (dist < min_dist) min_dist = dist;
            }
    return min_dist;
}

// logf as compiler-rt has it (lib/compiler_rt/log.zig logf = musl's logf.c before its table version: fdlibm e_logf.c)
inline float logZ(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    uint32_t    ix = m32::bits(x);
    int         k = 0;
    if (ix < 0x00800000u || (ix >> 31)) {
        if (0 == (ix << 1)) return -1.f / (x * x);
        if (ix >> 31) return (x - x) / 0.f;
        k -= 25;
        x *= 0x1p25f;
        ix = m32::bits(x);
    } else if (ix >= 0x7f800000u) {
        return x;
    } else if (0x3f800000u == ix) {
        return 0.f;
    }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += int(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    std::memcpy(&x, &ix, 4);
    const float f = x - 1.f;
    const float s = f / (2.f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (Lg2 + w * Lg4);
    const float t2 = z * (Lg1 + w * Lg3);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = float(k);
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// std.math.pow(f32, x, y) (lib/std/math/pow.zig) for the finite positive x and finite y the octave weights are made of
inline float powZ(float x, float y) {
    if (0.f == y || 1.f == x) return 1.f;
    if (1.f == y) return x;
    if (0.5f == y) return std::sqrt(x);
    if (-0.5f == y) return 1.f / std::sqrt(x);
    float yi = std::trunc(std::fabs(y)), yf = std::fabs(y) - yi;
    if (yi >= 2147483648.f) return expF(y * logZ(x));
    float   a1 = 1.f;
    int32_t ae = 0;
    if (0.f != yf) {
        if (yf > 0.5f) {
            yf -= 1.f;
            yi += 1.f;
        }
        a1 = expF(yf * logZ(x));
    }
    int     xe;
    float   x1 = std::frexp(x, &xe);
    for (int32_t i = int32_t(yi); 0 != i; i >>= 1) {
        if (xe < -(1 << 9) || (1 << 9) < xe) {  // overflow_shift = exponent bits + 1
            ae += xe;
            break;
        }
        if (i & 1) {
            a1 *= x1;
            ae += xe;
        }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) {
            x1 += x1;
            xe -= 1;
        }
    }
    if (y < 0.f) {
        a1 = 1.f / a1;
        ae = -ae;
    }
    return std::ldexp(a1, ae);
}

}  // namespace noise

// What a procedural reads beyond the Renderstate: the camera (context.camera at rs.time, context.zig:140, 211-213)
struct ProceduralContext {
    const ZygpuSceneDesc* scene = nullptr;
    const ZygpuViewDesc*  view = nullptr;
    Trafo                 camera;  // scene.propTransformationAt(camera.entity, rs.time)
};

struct DifferentialSurface {
    V4 dpdu, dpdv;
};

// Shape.surfaceDifferentials (shape.zig:565-575)
inline DifferentialSurface surfaceDifferentials(const ZygpuSceneDesc& scene, uint32_t prop, uint32_t primitive, const Trafo& trafo) {
    const ZygpuProp& pr = scene.props[prop];
    if (ZYGPU_SHAPE_RECTANGLE == pr.shape && trafo.scaleZ() < 0.f) return {-xyz0(trafo.r[0]), -xyz0(trafo.r[1])};  // rectangle.zig:613-622
    if (ZYGPU_SHAPE_TRIANGLE_MESH == pr.shape) {  // triangle_mesh.zig:751-760 + triangle.zig:102-131
        const ZygpuMesh& mesh = scene.meshes[pr.mesh];
        const uint32_t*  tri = mesh.triangles + 3 * size_t(primitive);
        V4               p[3];
        float            uv[3][2];
        for (int k = 0; k < 3; ++k) {
            p[k] = mk(mesh.positions[3 * size_t(tri[k])], mesh.positions[3 * size_t(tri[k]) + 1], mesh.positions[3 * size_t(tri[k]) + 2]);
            uv[k][0] = mesh.uvs[2 * size_t(tri[k])];
            uv[k][1] = mesh.uvs[2 * size_t(tri[k]) + 1];
        }
        const float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
        const float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        const V4    dp02 = p[0] - p[2], dp12 = p[1] - p[2];
        V4          dpdu, dpdv;
        if (0.f == std::fabs(determinant)) {
            const V4 ng = normalize3(cross3(p[2] - p[0], p[1] - p[0]));
            if (std::fabs(ng[0]) > std::fabs(ng[1])) dpdu = mk(-ng[2], 0.f, ng[0]) / S(std::sqrt(ng[0] * ng[0] + ng[2] * ng[2]));
            else dpdu = mk(0.f, ng[2], -ng[1]) / S(std::sqrt(ng[1] * ng[1] + ng[2] * ng[2]));
            dpdv = cross3(ng, dpdu);
        } else {
            const float invdet = 1.f / determinant;
            dpdu = S(invdet) * fma4(S(duv12[1]), dp02, S(-duv02[1]) * dp12);
            dpdv = S(invdet) * fma4(S(-duv12[0]), dp02, S(duv02[0]) * dp12);
        }
        return {trafo.objectToWorldVector(dpdu), trafo.objectToWorldVector(dpdv)};
    }
    return {S(-trafo.scaleX()) * xyz0(trafo.r[0]), S(-trafo.scaleY()) * xyz0(trafo.r[1])};
}

// hlp.triplanarDifferentials (material_helper.zig:96-122)
inline DifferentialSurface triplanarDifferentials(V4 wn, const Trafo& trafo) {
    const V4    n = trafo.worldToObjectNormal(wn);
    const float an[3] = {std::fabs(n[0]), std::fabs(n[1]), std::fabs(n[2])};
    const V4    r0 = xyz0(trafo.r[0]), r1 = xyz0(trafo.r[1]), r2 = xyz0(trafo.r[2]);
    if (an[0] > an[1] && an[0] > an[2]) {
        const float sign = std::copysign(1.f, n[0]);
        return {S(sign * trafo.scaleZ()) * r2, S(-1.f * trafo.scaleY()) * abs4(r1)};
    }
    if (an[1] > an[0] && an[1] > an[2]) {
        const float sign = std::copysign(1.f, n[1]);
        return {S(sign * trafo.scaleX()) * r0, S(-1.f * trafo.scaleZ()) * abs4(r2)};
    }
    const float sign = std::copysign(1.f, n[2]);
    return {S(-sign * trafo.scaleX()) * r0, S(-1.f * trafo.scaleY()) * abs4(r1)};
}

inline float sampleSpacing(const ZygpuViewDesc& view) {  // View.configure (take.zig:97-104)
    return 1.f / std::sqrt(float(view.num_expected_samples));
}

struct RayDifferential {  // camera/camera_base.zig RayDif
    V4 x_origin, x_direction, y_origin, y_direction;
};

// Perspective.calculateRayDifferential (camera_perspective.zig:223-249; no stereo: the eye is the camera's position)
inline RayDifferential cameraRayDifferential(const ProceduralContext& ctx, V4 p) {
    const Trafo& trafo = ctx.camera;
    const V4     p_w = xyz0(trafo.position);
    const V4     dir_w = normalize3(p - p_w);
    const V4     d_x_w = trafo.objectToWorldVector(ld3(ctx.view->d_x)), d_y_w = trafo.objectToWorldVector(ld3(ctx.view->d_y));
    const V4     ss = S(sampleSpacing(*ctx.view));
    return {p_w, normalize3(fma4(ss, d_x_w, dir_w)), p_w, normalize3(fma4(ss, d_y_w, dir_w))};
}

// Context.calculateScreenspaceDifferentials (context.zig:152-200): (dudx, dvdx, dudy, dvdy)
inline V4 calculateScreenspaceDifferentials(V4 p, V4 n, const RayDifferential& rd, V4 dpdu, V4 dpdv) {
    const float d = dot3(n, p);
    const float tx = -(dot3(n, rd.x_origin) - d) / dot3(n, rd.x_direction);
    const float ty = -(dot3(n, rd.y_origin) - d) / dot3(n, rd.y_direction);
    const V4    px = fma4(S(tx), rd.x_direction, rd.x_origin), py = fma4(S(ty), rd.y_direction, rd.y_origin);
    int         dim0, dim1;
    if (std::fabs(n[0]) > std::fabs(n[1]) && std::fabs(n[0]) > std::fabs(n[2])) dim0 = 1, dim1 = 2;
    else if (std::fabs(n[1]) > std::fabs(n[2])) dim0 = 0, dim1 = 2;
    else dim0 = 0, dim1 = 1;
    const float a[2][2] = {{dpdu[dim0], dpdv[dim0]}, {dpdu[dim1], dpdv[dim1]}};
    const float bx[2] = {px[dim0] - p[dim0], px[dim1] - p[dim1]}, by[2] = {py[dim0] - p[dim0], py[dim1] - p[dim1]};
    const float det = std::fma(a[0][0], a[1][1], -a[0][1] * a[1][0]);
    if (std::fabs(det) < 1.0e-10f) return S(0.f);
    const float dudx = std::fma(a[1][1], bx[0], -a[0][1] * bx[1]) / det;
    const float dvdx = std::fma(a[0][0], bx[1], -a[1][0] * bx[0]) / det;
    const float dudy = std::fma(a[1][1], by[0], -a[0][1] * by[1]) / det;
    const float dvdy = std::fma(a[0][0], by[1], -a[1][0] * by[0]) / det;
    return mk(dudx, dvdx, dudy, dvdy);
}

// Context.approximateDpDxy (context.zig:204-239) with Perspective.minDirDifferential (camera_perspective.zig:251-259)
inline void approximateDpDxy(const ProceduralContext& ctx, V4 p, V4 geo_n, V4& dpdx, V4& dpdy) {
    const V4     Z = mk(0.f, 0.f, 1.f);
    const V4     ss = S(sampleSpacing(*ctx.view));
    const V4     min_dir_x = ss * ld3(ctx.view->d_x), min_dir_y = ss * ld3(ctx.view->d_y);
    const Trafo& trafo = ctx.camera;
    const V4     p_o = trafo.worldToObjectPoint(p), n_o = trafo.worldToObjectNormal(geo_n);
    // Mat3x3.initRotationAlign(normalize(p_o), Z) (matrix3x3.zig:81-97)
    const V4     dn = normalize3(p_o);
    const V4     v = cross3(Z, dn);
    const float  c = dot3(Z, dn);
    const float  k = 1.f / (1.f + c);
    Trafo        m;
    m.r[0] = mk(v[0] * v[0] * k + c, v[1] * v[0] * k - v[2], v[2] * v[0] * k + v[1]);
    m.r[1] = mk(v[0] * v[1] * k + v[2], v[1] * v[1] * k + c, v[2] * v[1] * k - v[0]);
    m.r[2] = mk(v[0] * v[2] * k - v[1], v[1] * v[2] * k + v[0], v[2] * v[2] * k + c);
    const V4    p_down_z = m.transformVector(p_o), n_down_z = m.transformVector(n_o);
    const float d = n_down_z[2] * p_down_z[2];
    const V4    x_dir = Z + min_dir_x, y_dir = Z + min_dir_y;  // Ray.init(Origin, Z + min_dir_differential, ..)
    const float tx = -(dot3(n_down_z, S(0.f)) - d) / dot3(n_down_z, x_dir);
    const float ty = -(dot3(n_down_z, S(0.f)) - d) / dot3(n_down_z, y_dir);
    const V4    px = fma4(S(tx), x_dir, S(0.f)), py = fma4(S(ty), y_dir, S(0.f));  // Ray.point
    dpdx = trafo.objectToWorldVector(m.transformVectorTransposed(px - p_down_z));
    dpdy = trafo.objectToWorldVector(m.transformVectorTransposed(py - p_down_z));
}

}  // namespace orc
