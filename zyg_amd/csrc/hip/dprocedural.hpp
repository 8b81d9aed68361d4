// Procedural textures on the device (the material description's last refusal).
//
// The reference evaluates a procedural texture wherever a closure samples its maps (ts.sample2D_*, texture_sampler.zig:22-76 ->
// Procedural.sample2D_*, texture/procedural.zig:82-130).  Here one kernel, wf_textures (wavefront.hpp), evaluates the procedural
// textures of a vertex' material once, between the shading pass and the kernels that build the closure, in the order Material.sample
// reads them (substitute_material.zig:114-221, glass_material.zig:49-79), and leaves the values in Params::tex_values: the closure
// kernels read a value where they would have fetched a texel.  That keeps the noise and checker code, its registers and its calls out
// of every other kernel, and lets a Mix draw its sampler dimension exactly once per vertex (the closure kernels skip the draws the
// vertex made: `draws`).
//
//   Checker      texture/procedural_checker.zig:46-84 (the triangle-filtered integral over the pixel's footprint in texture space)
//   Noise        texture/procedural_noise.zig:44-142, noise/perlin.zig:11-219, noise/worley.zig:10-168
//   ChannelMix / Max / Mul / Mix / DetailNormal    texture/procedural_*.zig
//   the footprint: Context.screenspaceDifferentials / approximateDpDxy (scene/context.zig:139-239), Perspective.calculateRayDifferential /
//   minDirDifferential (camera_perspective.zig:223-259), Shape.surfaceDifferentials (shape.zig:565-575, rectangle.zig:613-622,
//   triangle_mesh.zig:751-760, triangle.zig:102-131), hlp.triplanarDifferentials (material_helper.zig:96-122)
//
// A procedural's children may be procedurals: the nesting is bounded (ZYGPU_PROCEDURAL_MAX_DEPTH, checked by the host) and unrolled by
// a template depth instead of recursion.
#pragma once

namespace zd {

namespace noise {

ZD uint32_t rotl(uint32_t x, uint32_t k) { return (x << k) | (x >> (32u - k)); }

ZD uint32_t bjfinal(uint32_t a, uint32_t b, uint32_t c) {  // perlin.zig:186-207
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

ZD float fade(float t) { return t * t * t * __fmaf_rn(t, __fmaf_rn(t, 6.f, -15.f), 10.f); }  // perlin.zig:64-70

ZD float gradient2(uint32_t hash, float x, float y) {  // perlin.zig:86-93
    const uint32_t h = hash & 7u;
    const float    u = h < 4u ? x : y;
    const float    v = 2.f * (h < 4u ? y : x);
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}

ZD float gradient3(uint32_t hash, float x, float y, float z) {  // perlin.zig:104-116
    const uint32_t h = hash & 15u;
    const float    u = h < 8u ? x : y;
    const float    t = (12u == h || 14u == h) ? x : z;
    const float    v = h < 4u ? y : t;
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}

ZD int32_t zigMod(int32_t a, int32_t n) {  // @mod: the result has the divisor's sign
    const int32_t r = a % n;
    return (r != 0 && ((r < 0) != (n < 0))) ? r + n : r;
}

ZD float bilinear(const float c[4], float s, float t) {  // math.zig:172-179
    const float _s = 1.f - s, _t = 1.f - t;
    return _t * (_s * c[0] + s * c[1]) + t * (_s * c[2] + s * c[3]);
}

__device__ __noinline__ float perlin2D(float px, float py, float period_x, float period_y) {  // perlin.zig:11-39
    const float   flx = floorf(px), fly = floorf(py);
    const float   fx = px - flx, fy = py - fly;
    const int32_t X = int32_t(flx), Y = int32_t(fly);
    const float   u = fade(fx), v = fade(fy);
    int32_t       x0 = X, x1 = int32_t(uint32_t(X) + 1u), y0 = Y, y1 = int32_t(uint32_t(Y) + 1u);
    const int32_t perx = int32_t(period_x), pery = int32_t(period_y);
    if (perx > 0) x0 = zigMod(x0, perx), x1 = zigMod(x1, perx);
    if (pery > 0) y0 = zigMod(y0, pery), y1 = zigMod(y1, pery);
    const uint32_t start = 0xdeadbeefu + (2u << 2) + 13u;  // hash2v, perlin.zig:136-142
    const float    c[4] = {gradient2(bjfinal(start + uint32_t(x0), start + uint32_t(y0), start), fx, fy),
                           gradient2(bjfinal(start + uint32_t(x1), start + uint32_t(y0), start), fx - 1.f, fy),
                           gradient2(bjfinal(start + uint32_t(x0), start + uint32_t(y1), start), fx, fy - 1.f),
                           gradient2(bjfinal(start + uint32_t(x1), start + uint32_t(y1), start), fx - 1.f, fy - 1.f)};
    return 0.6616f * bilinear(c, u, v);
}

__device__ __noinline__ float perlin3D(float px, float py, float pz) {  // perlin.zig:41-61
    const float    fl[3] = {floorf(px), floorf(py), floorf(pz)};
    const float    f[3] = {px - fl[0], py - fl[1], pz - fl[2]};
    const int32_t  P[3] = {int32_t(fl[0]), int32_t(fl[1]), int32_t(fl[2])};
    const float    u = fade(f[0]), v = fade(f[1]), w = fade(f[2]);
    const uint32_t start = 0xdeadbeefu + (3u << 2) + 13u;  // hash3v, perlin.zig:154-161
    float          c0[4], c1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int32_t  x = P[0] + (k & 1), y = P[1] + (k >> 1);
        const float    gx = f[0] - float(k & 1), gy = f[1] - float(k >> 1);
        const uint32_t a = start + uint32_t(x), b = start + uint32_t(y), c = start + uint32_t(P[2]);
        c0[k] = gradient3(bjfinal(a, b, c), gx, gy, f[2]);
        c1[k] = gradient3(bjfinal(a, b, c + 1u), gx, gy, f[2] - 1.f);
    }
    return 0.9820f * lerpf(bilinear(c0, u, v), bilinear(c1, u, v), w);
}

ZD float to01(uint32_t bits) { return __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u) - 1.f; }  // vecTo01, worley.zig:157-168

__device__ __noinline__ float worley2D(float px, float py, float jitter) {  // worley.zig:10-38, 79-84, 93-99, 109-117, 129-143
    const float   flx = floorf(px), fly = floorf(py);
    const float   lx = px - flx, ly = py - fly;
    const int32_t X = int32_t(flx), Y = int32_t(fly);
    float         min_dist = 1.0e6f;
    for (int32_t x = -1; x <= 1; ++x)
        for (int32_t y = -1; y <= 1; ++y) {
            uint32_t v0 = uint32_t(int32_t(floorf(float(x + X)))) * 1664525u + 1013904223u;  // cellNoise2 of the cell as an f32
            uint32_t v1 = uint32_t(int32_t(floorf(float(y + Y)))) * 1664525u + 1013904223u;
            v0 += v1 * 1664525u;
            v1 += v0 * 1664525u;
            v0 ^= v0 >> 16;
            v1 ^= v1 >> 16;
            v0 += v1 * 1664525u;
            v1 += v0 * 1664525u;
            v0 ^= v0 >> 16;
            v1 ^= v1 >> 16;
            const float ox = __fmaf_rn(to01(v0) - 0.5f, jitter, 0.5f), oy = __fmaf_rn(to01(v1) - 0.5f, jitter, 0.5f);
            const float dx = (float(x) + ox) - lx, dy = (float(y) + oy) - ly;
            const float dist = dx * dx + dy * dy;
            if (dist < min_dist) min_dist = dist;
        }
    return min_dist;
}

__device__ __noinline__ float worley3D(float px, float py, float pz, float jitter) {  // worley.zig:40-64, 86-91, 101-107, 119-127, 145-155
    const float   fl[3] = {floorf(px), floorf(py), floorf(pz)};
    const float   l[3] = {px - fl[0], py - fl[1], pz - fl[2]};
    const int32_t P[3] = {int32_t(fl[0]), int32_t(fl[1]), int32_t(fl[2])};
    float         min_dist = 1.0e6f;
    for (int32_t x = -1; x <= 1; ++x)
        for (int32_t y = -1; y <= 1; ++y)
            for (int32_t z = -1; z <= 1; ++z) {
                uint32_t v0 = uint32_t(int32_t(floorf(float(x + P[0])))) * 1664525u + 1013904223u;
                uint32_t v1 = uint32_t(int32_t(floorf(float(y + P[1])))) * 1664525u + 1013904223u;
                uint32_t v2 = uint32_t(int32_t(floorf(float(z + P[2])))) * 1664525u + 1013904223u;
                v0 += v1 * v2;
                v1 += v2 * v0;
                v2 += v0 * v1;
                v0 ^= v0 >> 16;
                v1 ^= v1 >> 16;
                v2 ^= v2 >> 16;
                v0 += v1 * v2;
                v1 += v2 * v0;
                v2 += v0 * v1;
                const float d0 = (float(x) + __fmaf_rn(to01(v0) - 0.5f, jitter, 0.5f)) - l[0];
                const float d1 = (float(y) + __fmaf_rn(to01(v1) - 0.5f, jitter, 0.5f)) - l[1];
                const float d2 = (float(z) + __fmaf_rn(to01(v2) - 0.5f, jitter, 0.5f)) - l[2];
                const float dist = d0 * d0 + d1 * d1 + d2 * d2;
                if (dist < min_dist) min_dist = dist;
            }
    return min_dist;
}

#if ZYG_FAST
ZD float powZ(float x, float y) { return powf(x, y); }
#else
// logf as compiler-rt has it (lib/compiler_rt/log.zig logf: fdlibm's e_logf.c) and std.math.pow(f32, ..) (lib/std/math/pow.zig), for the
// finite positive x and finite y the octave weights are made of: the precise units restate the reference's libm (dmath.hpp)
ZD float logZ(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0xaaaaaa.0p-24f, Lg2 = 0xccce13.0p-25f, Lg3 = 0x91e9ee.0p-25f, Lg4 = 0xf89e26.0p-26f;
    uint32_t    ix = __float_as_uint(x);
    int         k = 0;
    if (ix < 0x00800000u || (ix >> 31)) {
        if (0 == (ix << 1)) return -1.f / (x * x);
        if (ix >> 31) return (x - x) / 0.f;
        k -= 25;
        x *= 0x1p25f;
        ix = __float_as_uint(x);
    } else if (ix >= 0x7f800000u) {
        return x;
    } else if (0x3f800000u == ix) {
        return 0.f;
    }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += int(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    x = __uint_as_float(ix);
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
ZD float powZ(float x, float y) {
    if (0.f == y || 1.f == x) return 1.f;
    if (1.f == y) return x;
    if (0.5f == y) return sqrtf(x);
    if (-0.5f == y) return 1.f / sqrtf(x);
    float yi = truncf(fabsf(y)), yf = fabsf(y) - yi;
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
    int   xe;
    float x1 = frexpf(x, &xe);
    for (int32_t i = int32_t(yi); 0 != i; i >>= 1) {
        if (xe < -(1 << 9) || (1 << 9) < xe) {
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
    return ldexpf(a1, ae);
}
#endif

}  // namespace noise

// The Renderstate a procedural reads (renderstate.zig:10-45) with what Context adds: the camera at the vertex' time and the sampler
struct TexState {
    f3       p, geo_n;  // geo_n: after Vertex.sample's two-sided flip
    Trafo    trafo;
    float    u, v, stochastic_r;
    uint32_t prop, primitive;
    Trafo    camera;
    Sampler* smp;
    uint32_t draws;  // sampler dimensions the vertex' procedurals have drawn (Mix)
};

ZD void textureStOf(uint32_t mode, const TexState& rs, float& su, float& sv) { textureSt(mode, rs.u, rs.v, rs.p, rs.geo_n, rs.trafo, su, sv); }

ZD float sampleSpacing() { return 1.f / sqrtf(float(c_view.v.num_expected_samples)); }  // View.configure (take.zig:97-104)

// Shape.surfaceDifferentials (shape.zig:565-575)
ZD void surfaceDifferentials(const DScene& s, const TexState& rs, f3& dpdu, f3& dpdv) {
    const ZygpuProp& pr = s.props[rs.prop];
    const Trafo&     trafo = rs.trafo;
    if (ZYGPU_SHAPE_RECTANGLE == pr.shape && trafo.scale.z < 0.f) {  // rectangle.zig:613-622
        dpdu = -trafo.r0;
        dpdv = -trafo.r1;
        return;
    }
    if (ZYGPU_SHAPE_TRIANGLE_MESH == pr.shape) {  // triangle_mesh.zig:751-760 + triangle.zig:102-131
        const DMesh&    mesh = s.meshes[pr.mesh];
        const uint32_t* tri = mesh.triangles + 3 * size_t(rs.primitive);
        f3              p[3];
        float           uv[3][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t vi = tri[k];
            p[k] = mk3(mesh.positions[3 * size_t(vi)], mesh.positions[3 * size_t(vi) + 1], mesh.positions[3 * size_t(vi) + 2]);
            uv[k][0] = mesh.uvs[2 * size_t(vi)];
            uv[k][1] = mesh.uvs[2 * size_t(vi) + 1];
        }
        const float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
        const float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        const f3    dp02 = p[0] - p[2], dp12 = p[1] - p[2];
        f3          du, dv;
        if (0.f == fabsf(determinant)) {
            const f3 ng = normalize3(cross3(p[2] - p[0], p[1] - p[0]));
            if (fabsf(ng.x) > fabsf(ng.y)) du = mk3(-ng.z, 0.f, ng.x) / S3(sqrtf(ng.x * ng.x + ng.z * ng.z));
            else du = mk3(0.f, ng.z, -ng.y) / S3(sqrtf(ng.y * ng.y + ng.z * ng.z));
            dv = cross3(ng, du);
        } else {
            const float invdet = 1.f / determinant;
            du = S3(invdet) * fma3(S3(duv12[1]), dp02, S3(-duv02[1]) * dp12);
            dv = S3(invdet) * fma3(S3(-duv12[0]), dp02, S3(duv02[0]) * dp12);
        }
        dpdu = trafo.objectToWorldVector(du);
        dpdv = trafo.objectToWorldVector(dv);
        return;
    }
    dpdu = S3(-trafo.scale.x) * trafo.r0;
    dpdv = S3(-trafo.scale.y) * trafo.r1;
}

ZD f3 abs3(f3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }

// hlp.triplanarDifferentials (material_helper.zig:96-122)
ZD void triplanarDifferentials(f3 wn, const Trafo& trafo, f3& dpdu, f3& dpdv) {
    const f3    n = trafo.worldToObjectNormal(wn);
    const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax > ay && ax > az) {
        dpdu = S3(copysignf(1.f, n.x) * trafo.scale.z) * trafo.r2;
        dpdv = S3(-1.f * trafo.scale.y) * abs3(trafo.r1);
    } else if (ay > ax && ay > az) {
        dpdu = S3(copysignf(1.f, n.y) * trafo.scale.x) * trafo.r0;
        dpdv = S3(-1.f * trafo.scale.z) * abs3(trafo.r2);
    } else {
        dpdu = S3(-copysignf(1.f, n.z) * trafo.scale.x) * trafo.r0;
        dpdv = S3(-1.f * trafo.scale.y) * abs3(trafo.r1);
    }
}

ZD float comp(f3 a, int k) { return 0 == k ? a.x : 1 == k ? a.y : a.z; }

// Context.screenspaceDifferentials (context.zig:139-200): (dudx, dvdx, dudy, dvdy)
__device__ __noinline__ float4 screenspaceDifferentials(const DScene& s, const TexState& rs, uint32_t mode) {
    // Perspective.calculateRayDifferential (camera_perspective.zig:223-249; no stereo: the eye is the camera's position)
    const ZygpuViewDesc& v = c_view.v;
    const Trafo&         cam = rs.camera;
    const f3             p_w = cam.position;
    const f3             dir_w = normalize3(rs.p - p_w);
    const f3             d_x_w = cam.objectToWorldVector(mk3(v.d_x[0], v.d_x[1], v.d_x[2])), d_y_w = cam.objectToWorldVector(mk3(v.d_y[0], v.d_y[1], v.d_y[2]));
    const f3             ss = S3(sampleSpacing());
    const f3             x_dir = normalize3(fma3(ss, d_x_w, dir_w)), y_dir = normalize3(fma3(ss, d_y_w, dir_w));
    f3                   dpdu, dpdv;
    if (0 == (mode & (ZYGPU_TEX_TRIPLANAR | ZYGPU_TEX_OBJECTPOS))) surfaceDifferentials(s, rs, dpdu, dpdv);
    else triplanarDifferentials(rs.geo_n, rs.trafo, dpdu, dpdv);
    const f3    n = rs.geo_n, p = rs.p;
    const float d = dot3(n, p);
    const float tx = -(dot3(n, p_w) - d) / dot3(n, x_dir);
    const float ty = -(dot3(n, p_w) - d) / dot3(n, y_dir);
    const f3    px = fma3(S3(tx), x_dir, p_w), py = fma3(S3(ty), y_dir, p_w);
    int         dim0, dim1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) dim0 = 1, dim1 = 2;
    else if (fabsf(n.y) > fabsf(n.z)) dim0 = 0, dim1 = 2;
    else dim0 = 0, dim1 = 1;
    const float a00 = comp(dpdu, dim0), a01 = comp(dpdv, dim0), a10 = comp(dpdu, dim1), a11 = comp(dpdv, dim1);
    const float bx0 = comp(px, dim0) - comp(p, dim0), bx1 = comp(px, dim1) - comp(p, dim1);
    const float by0 = comp(py, dim0) - comp(p, dim0), by1 = comp(py, dim1) - comp(p, dim1);
    const float det = __fmaf_rn(a00, a11, -a01 * a10);
    if (fabsf(det) < 1.0e-10f) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float dudx = __fmaf_rn(a11, bx0, -a01 * bx1) / det;
    const float dvdx = __fmaf_rn(a00, bx1, -a10 * bx0) / det;
    const float dudy = __fmaf_rn(a11, by0, -a01 * by1) / det;
    const float dvdy = __fmaf_rn(a00, by1, -a10 * by0) / det;
    return make_float4(dudx, dvdx, dudy, dvdy);
}

// Context.approximateDpDxy (context.zig:204-239) with Perspective.minDirDifferential (camera_perspective.zig:251-259)
ZD void approximateDpDxy(const TexState& rs, f3& dpdx, f3& dpdy) {
    const ZygpuViewDesc& v = c_view.v;
    const f3             Z = mk3(0.f, 0.f, 1.f);
    const f3             ss = S3(sampleSpacing());
    const f3             min_dir_x = ss * mk3(v.d_x[0], v.d_x[1], v.d_x[2]), min_dir_y = ss * mk3(v.d_y[0], v.d_y[1], v.d_y[2]);
    const Trafo&         trafo = rs.camera;
    const f3             p_o = trafo.worldToObjectPoint(rs.p), n_o = trafo.worldToObjectNormal(rs.geo_n);
    const f3             dn = normalize3(p_o);  // Mat3x3.initRotationAlign(normalize(p_o), Z) (matrix3x3.zig:81-97)
    const f3             w = cross3(Z, dn);
    const float          c = dot3(Z, dn);
    const float          k = 1.f / (1.f + c);
    Trafo                m;
    m.r0 = mk3(w.x * w.x * k + c, w.y * w.x * k - w.z, w.z * w.x * k + w.y);
    m.r1 = mk3(w.x * w.y * k + w.z, w.y * w.y * k + c, w.z * w.y * k - w.x);
    m.r2 = mk3(w.x * w.z * k - w.y, w.y * w.z * k + w.x, w.z * w.z * k + c);
    const f3    p_down_z = m.transformVector(p_o), n_down_z = m.transformVector(n_o);
    const float d = n_down_z.z * p_down_z.z;
    const f3    x_dir = Z + min_dir_x, y_dir = Z + min_dir_y;
    const float tx = -(dot3(n_down_z, S3(0.f)) - d) / dot3(n_down_z, x_dir);
    const float ty = -(dot3(n_down_z, S3(0.f)) - d) / dot3(n_down_z, y_dir);
    const f3    px = fma3(S3(tx), x_dir, S3(0.f)), py = fma3(S3(ty), y_dir, S3(0.f));
    dpdx = trafo.objectToWorldVector(m.transformVectorTransposed(px - p_down_z));
    dpdy = trafo.objectToWorldVector(m.transformVectorTransposed(py - p_down_z));
}

ZD float texAddressF(bool repeat, float x) { return repeat ? x - floorf(x) : mn(mx(x, 0.f), 1.f); }  // sampler_mode.zig:25-30

// Checker.evaluate (procedural_checker.zig:46-84)
__device__ __noinline__ f3 checkerEvaluate(const DScene& s, const ZygpuProcedural& c, uint32_t mode, const TexState& rs) {
    const float4 dd = screenspaceDifferentials(s, rs, mode);
    float        su, sv;
    textureStOf(mode, rs, su, sv);
    const float scale = c.checker_scale;
    const float uv[2] = {scale * texAddressF(0 != (mode & ZYGPU_TEX_REPEAT_U), su), scale * texAddressF(0 != (mode & ZYGPU_TEX_REPEAT_V), sv)};
    const float ddx[2] = {scale * dd.x, scale * dd.y}, ddy[2] = {scale * dd.z, scale * dd.w};
    float       i[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // checkersGrad: the triangle-filtered integral
        const float w = mx(fabsf(ddx[k]), fabsf(ddy[k])) + 0.0001f;
        auto        pf = [](float x) {
            const float h = ((x * 0.5f) - floorf(x * 0.5f)) - 0.5f;
            return x * 0.5f + h * (1.f - 2.f * fabsf(h));
        };
        i[k] = (pf(uv[k] + w) - 2.f * pf(uv[k]) + pf(uv[k] - w)) / (w * w);
    }
    const float t = 0.5f - 0.5f * i[0] * i[1];
    return lerp3(mk3(c.value[0][0], c.value[0][1], c.value[0][2]), mk3(c.value[1][0], c.value[1][1], c.value[1][2]), S3(t));
}

// Noise.evaluate1 (procedural_noise.zig:44-101)
__device__ __noinline__ float noiseEvaluate1(const ZygpuProcedural& n, uint32_t mode, const TexState& rs, f3 offset) {
    const bool  cellular = 0 != (n.flags & ZYGPU_NOISE_CELLULAR);
    const float att = n.attenuation;
    float       weight = 0.f, amplitude = 1.f, value = 0.f;
    if (mode & ZYGPU_TEX_OBJECTPOS) {
        f3       scale = mk3(n.scale[0], n.scale[1], n.scale[2]);
        const f3 uvw = rs.trafo.worldToObjectPoint(rs.p - offset);
        for (uint32_t l = 0; l < n.levels; ++l) {
            const float local_weight = noise::powZ(amplitude, att);
            const f3    q = uvw * scale;
            const float local = cellular ? noise::worley3D(q.x, q.y, q.z, 1.f) : noise::perlin3D(q.x, q.y, q.z);
            value += local * local_weight;
            weight += local_weight;
            amplitude *= 0.5f;
            scale = scale * S3(2.f);
        }
    } else {
        float scale0 = n.scale[0], scale1 = n.scale[1], period0 = n.period[0], period1 = n.period[1];
        float su, sv;
        textureStOf(mode, rs, su, sv);
        const float u = su - offset.x, v = sv - offset.y;
        for (uint32_t l = 0; l < n.levels; ++l) {
            const float local_weight = noise::powZ(amplitude, att);
            const float local = cellular ? noise::worley2D(u * scale0, v * scale1, 1.f) : noise::perlin2D(u * scale0, v * scale1, period0, period1);
            value += local * local_weight;
            weight += local_weight;
            amplitude *= 0.5f;
            scale0 *= 2.f, scale1 *= 2.f;
            period0 *= 2.f, period1 *= 2.f;
        }
    }
    value /= weight;
    const float unsigned_ = cellular ? value : ((n.flags & ZYGPU_NOISE_ABSOLUTE) ? fabsf(value) : (value + 1.f) * 0.5f);
    const float a = n.ratio - n.transition, b = n.ratio + n.transition;
    const float result = mn(mx((unsigned_ - a) / (b - a), 0.f), 1.f);
    return (n.flags & ZYGPU_NOISE_INVERT) ? 1.f - result : result;
}

// Noise.evaluateNormalmap (procedural_noise.zig:103-135)
__device__ __noinline__ float2 noiseNormalmap(const DScene& s, const ZygpuProcedural& n, uint32_t mode, const TexState& rs) {
    float nx, ny, nz;
    if (mode & ZYGPU_TEX_OBJECTPOS) {
        f3 dpdx, dpdy;
        approximateDpDxy(rs, dpdx, dpdy);
        const float center = noiseEvaluate1(n, mode, rs, S3(0.f));
        const float left = noiseEvaluate1(n, mode, rs, dpdx);
        const float bottom = noiseEvaluate1(n, mode, rs, dpdy);
        nx = left - center;
        ny = bottom - center;
        nz = length3(dpdx + dpdy);
    } else {
        const float4 dd = screenspaceDifferentials(s, rs, mode);
        const float  shift_x = fabsf(dd.x) + fabsf(dd.z), shift_y = fabsf(dd.y) + fabsf(dd.w);
        const float  center = noiseEvaluate1(n, mode, rs, S3(0.f));
        const float  left = noiseEvaluate1(n, mode, rs, mk3(shift_x, 0.f, 0.f));
        const float  top = noiseEvaluate1(n, mode, rs, mk3(0.f, shift_y, 0.f));
        nx = left - center;
        ny = top - center;
        nz = sqrtf(shift_x * shift_x + shift_y * shift_y);  // math.length2
    }
    const f3 nn = normalize3(mk3(nx, ny, nz));
    return make_float2(nn.x, nn.y);
}

// ---- ts.sample2D_1 / _2 / _3 (texture_sampler.zig:22-76) over a ZygpuTexture, and Procedural.sample2D_* (procedural.zig:82-130);
// DEPTH: the levels of procedurals that may still follow
template <int DEPTH> __device__ __noinline__ float  procedural1(const DScene& s, uint32_t index, uint32_t mode, TexState& rs);
template <int DEPTH> __device__ __noinline__ float2 procedural2(const DScene& s, uint32_t index, uint32_t mode, TexState& rs);
template <int DEPTH> __device__ __noinline__ f3     procedural3(const DScene& s, uint32_t index, uint32_t mode, TexState& rs);

template <int DEPTH>
ZD float texture1(const DScene& s, const ZygpuTexture& t, const float* uniform, TexState& rs) {
    if (kNull == t.image) return uniform[0];
    if (t.sampler & ZYGPU_TEX_PROCEDURAL) {
        if constexpr (DEPTH > 0) return procedural1<DEPTH - 1>(s, t.image, t.sampler, rs);
        else return 0.f;
    }
    const uint32_t format = s.images[t.image].format;
    if (ZYGPU_IMAGE_BYTE1_UNORM != format && ZYGPU_IMAGE_FLOAT1 != format) return 0.f;  // image2D_1 of another width (texture.zig:166-175)
    float su, sv;
    textureStOf(t.sampler, rs, su, sv);
    return sampleImageMap1(s, t.image, t.sampler, t.scale, su, sv, rs.stochastic_r);
}
template <int DEPTH>
ZD float2 texture2(const DScene& s, const ZygpuTexture& t, const float* uniform, TexState& rs) {
    if (kNull == t.image) return make_float2(uniform[0], uniform[1]);
    if (t.sampler & ZYGPU_TEX_PROCEDURAL) {
        if constexpr (DEPTH > 0) return procedural2<DEPTH - 1>(s, t.image, t.sampler, rs);
        else return make_float2(0.f, 0.f);
    }
    const uint32_t format = s.images[t.image].format;
    if (ZYGPU_IMAGE_BYTE2 != format && ZYGPU_IMAGE_FLOAT2 != format) return make_float2(0.f, 0.f);  // image2D_2 (texture.zig:177-186)
    float su, sv, o0, o1;
    textureStOf(t.sampler, rs, su, sv);
    sampleImageMap2(s, t.image, t.sampler, t.scale, su, sv, rs.stochastic_r, o0, o1);
    return make_float2(o0, o1);
}
template <int DEPTH>
ZD f3 texture3(const DScene& s, const ZygpuTexture& t, const float* uniform, TexState& rs) {
    if (kNull == t.image) return mk3(uniform[0], uniform[1], uniform[2]);
    if (t.sampler & ZYGPU_TEX_PROCEDURAL) {
        if constexpr (DEPTH > 0) return procedural3<DEPTH - 1>(s, t.image, t.sampler, rs);
        else return S3(0.f);
    }
    float su, sv;
    textureStOf(t.sampler, rs, su, sv);
    const uint32_t format = s.images[t.image].format;
    if (ZYGPU_IMAGE_BYTE1_UNORM == format || ZYGPU_IMAGE_FLOAT1 == format)  // image2D_3 of a one-channel image (texture.zig:192-195)
        return mk3(sampleImageMap1(s, t.image, t.sampler, t.scale, su, sv, rs.stochastic_r), 0.f, 0.f);
    if (ZYGPU_IMAGE_BYTE3_SRGB != format && ZYGPU_IMAGE_FLOAT3 != format) return S3(0.f);
    return sampleImageMap(s, t.image, t.sampler, t.scale, su, sv, rs.stochastic_r);
}

ZD f3 row3(const float* v) { return mk3(v[0], v[1], v[2]); }
ZD float clamp01(float x) { return mn(mx(x, 0.f), 1.f); }

template <int DEPTH>
ZD f3 detailReconstruct(const DScene& s, const ZygpuTexture& t, const float* uniform, TexState& rs) {  // procedural_detail_normal.zig:31-35
    const float2 nm = texture2<DEPTH>(s, t, uniform, rs);
    const float  nmz = sqrtf(mx(1.f - (nm.x * nm.x + nm.y * nm.y), 0.01f));
    return normalize3(mk3(nm.x, nm.y, nmz));
}

template <int DEPTH>
__device__ __noinline__ float procedural1(const DScene& s, uint32_t index, uint32_t mode, TexState& rs) {
    const ZygpuProcedural& p = s.procedurals[index];
    switch (p.kind) {
        case ZYGPU_PROC_CHANNEL_MIX: return clamp01(dot3(texture3<DEPTH>(s, p.child[0], p.child_uniform[0], rs), row3(p.value[0])));
        case ZYGPU_PROC_CHECKER: return checkerEvaluate(s, p, mode, rs).x;
        case ZYGPU_PROC_MAX: {
            const float a = texture1<DEPTH>(s, p.child[0], p.child_uniform[0], rs), b = texture1<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            return mx(a, b);
        }
        case ZYGPU_PROC_MIX: {
            const float weight = texture1<DEPTH>(s, p.child[2], p.child_uniform[2], rs);
            const float r = rs.smp->sample1D();
            rs.draws += 1;
            const uint32_t k = weight < r ? 0u : 1u;
            return texture1<DEPTH>(s, p.child[k], p.child_uniform[k], rs);
        }
        case ZYGPU_PROC_MUL: {
            const float a = texture1<DEPTH>(s, p.child[0], p.child_uniform[0], rs), b = texture1<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            return a * b;
        }
        case ZYGPU_PROC_NOISE: return noiseEvaluate1(p, mode, rs, S3(0.f));
        default: return 0.f;  // DetailNormal
    }
}

template <int DEPTH>
__device__ __noinline__ float2 procedural2(const DScene& s, uint32_t index, uint32_t mode, TexState& rs) {
    const ZygpuProcedural& p = s.procedurals[index];
    switch (p.kind) {
        case ZYGPU_PROC_CHANNEL_MIX: {
            const f3 c = texture3<DEPTH>(s, p.child[0], p.child_uniform[0], rs);
            return make_float2(clamp01(dot3(c, row3(p.value[0]))), clamp01(dot3(c, row3(p.value[1]))));
        }
        case ZYGPU_PROC_CHECKER: {
            const f3 c = checkerEvaluate(s, p, mode, rs);
            return make_float2(c.x, c.y);
        }
        case ZYGPU_PROC_DETAIL_NORMAL: {  // procedural_detail_normal.zig:19-29
            const f3    n1 = detailReconstruct<DEPTH>(s, p.child[0], p.child_uniform[0], rs);
            const f3    n2 = detailReconstruct<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            const Frame fr = Frame::init(n1);  // math.orthonormalBasis3
            const f3    r = S3(n2.x) * fr.x + S3(n2.y) * fr.y + S3(n2.z) * n1;
            return make_float2(r.x, r.y);
        }
        case ZYGPU_PROC_MAX: {
            const float2 a = texture2<DEPTH>(s, p.child[0], p.child_uniform[0], rs), b = texture2<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            return make_float2(mx(a.x, b.x), mx(a.y, b.y));
        }
        case ZYGPU_PROC_MIX: {
            const float weight = texture1<DEPTH>(s, p.child[2], p.child_uniform[2], rs);
            const float r = rs.smp->sample1D();
            rs.draws += 1;
            const uint32_t k = weight < r ? 0u : 1u;
            return texture2<DEPTH>(s, p.child[k], p.child_uniform[k], rs);
        }
        case ZYGPU_PROC_MUL: {
            const float2 a = texture2<DEPTH>(s, p.child[0], p.child_uniform[0], rs), b = texture2<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            return make_float2(a.x * b.x, a.y * b.y);
        }
        default: return noiseNormalmap(s, p, mode, rs);
    }
}

template <int DEPTH>
__device__ __noinline__ f3 procedural3(const DScene& s, uint32_t index, uint32_t mode, TexState& rs) {
    const ZygpuProcedural& p = s.procedurals[index];
    switch (p.kind) {
        case ZYGPU_PROC_CHANNEL_MIX: {
            const f3 c = texture3<DEPTH>(s, p.child[0], p.child_uniform[0], rs);
            return mk3(clamp01(dot3(c, row3(p.value[0]))), clamp01(dot3(c, row3(p.value[1]))), clamp01(dot3(c, row3(p.value[2]))));
        }
        case ZYGPU_PROC_CHECKER: return checkerEvaluate(s, p, mode, rs);
        case ZYGPU_PROC_MAX: {
            const f3 a = texture3<DEPTH>(s, p.child[0], p.child_uniform[0], rs), b = texture3<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            return max3v(a, b);
        }
        case ZYGPU_PROC_MIX: {
            const float weight = texture1<DEPTH>(s, p.child[2], p.child_uniform[2], rs);
            const float r = rs.smp->sample1D();
            rs.draws += 1;
            const uint32_t k = weight < r ? 0u : 1u;
            return texture3<DEPTH>(s, p.child[k], p.child_uniform[k], rs);
        }
        case ZYGPU_PROC_MUL: {
            const f3 a = texture3<DEPTH>(s, p.child[0], p.child_uniform[0], rs), b = texture3<DEPTH>(s, p.child[1], p.child_uniform[1], rs);
            return a * b;
        }
        case ZYGPU_PROC_NOISE: return S3(noiseEvaluate1(p, mode, rs, S3(0.f)));
        default: return S3(0.f);  // DetailNormal
    }
}

}  // namespace zd
