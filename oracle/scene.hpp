// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp).
//
// Scene access + ray casting over the compiled scene (include/zygpu.h), restating:
//   Ray                         src/base/math/ray.zig:11-29
//   ComposedTransformation      src/core/scene/composed_transformation.zig:70-130
//   bvh.Node.intersect          src/core/scene/bvh/node.zig:73-87,  NodeStack node_stack.zig:1-30
//   AABB.intersect              src/base/math/aabb.zig:46-60
//   PropBvh intersect/visibility src/core/scene/prop/prop_tree.zig:56-116,185-240
//   Prop intersect/visibility   src/core/scene/prop/prop.zig:163-237
//   TriangleTree                src/core/scene/shape/triangle/triangle_tree.zig:46-109,197-242
//   triangle.intersect          src/core/scene/shape/triangle/triangle.zig:26-80
//   TriangleMesh.fragment       src/core/scene/shape/triangle/triangle_mesh.zig:310-335, triangle_data.zig:106-130
//   Rectangle                   src/core/scene/shape/rectangle.zig:30-152
//   ray offset                  src/core/scene/ray_offset.zig:5-31, shape/intersection.zig:115-123
#pragma once

#include "../include/zygpu.h"
#include "omath.hpp"
#include "sampler.hpp"

namespace orc {

constexpr float    kRayMaxT = 2.14748313e+09f;  // ray_offset.zig:5
constexpr uint32_t kNull = 0xFFFFFFFFu;

struct Ray {
    V4       origin, direction, inv_direction;
    float    min_t, max_t;
    uint32_t time = 0;  // Probe.time (probe.zig) as ticks since the frame's start: selects an animated prop's transformation
    static Ray init(V4 o, V4 d, float mint, float maxt) {
        Ray r;
        r.origin = o;
        r.direction = d;
        r.inv_direction = S(1.f) / mk(d[0], d[1], d[2], 1.f);
        r.min_t = mint;
        r.max_t = maxt;
        return r;
    }
    V4 point(float t) const { return fma4(S(t), direction, origin); }
};

struct Trafo {  // rotation rows carry the scale in w
    V4 r[3];
    V4 position;
    static Trafo from(const ZygpuTrafo& t) { return {{ld4(t.r[0]), ld4(t.r[1]), ld4(t.r[2])}, ld4(t.position)}; }
    float scaleX() const { return r[0][3]; }
    float scaleY() const { return r[1][3]; }
    float scaleZ() const { return r[2][3]; }
    V4    scale() const { return mk(r[0][3], r[1][3], r[2][3], 1.f); }
    V4    transformVector(V4 v) const {  // matrix3x3.zig:113-126
        V4 res = S(v[0]) * r[0];
        res = fma4(S(v[1]), r[1], res);
        return fma4(S(v[2]), r[2], res);
    }
    V4 transformVectorTransposed(V4 v) const {  // matrix3x3.zig:128-139
        const V4 x = v * r[0], y = v * r[1], z = v * r[2];
        return mk(x[0] + x[1] + x[2], y[0] + y[1] + y[2], z[0] + z[1] + z[2]);
    }
    V4 objectToWorldVector(V4 v) const {
        const V4 s = scale();
        const V4 a = r[0] * S(s[0]), b = r[1] * S(s[1]), c = r[2] * S(s[2]);
        V4       res = S(v[0]) * a;
        res = fma4(S(v[1]), b, res);
        return fma4(S(v[2]), c, res);
    }
    V4  objectToWorldPoint(V4 p) const { return objectToWorldVector(p) + position; }
    V4  objectToWorldNormal(V4 n) const { return transformVector(n); }
    V4  frameToWorldPoint(V4 p) const { return objectToWorldNormal(p) + position; }
    V4  worldToObjectNormal(V4 n) const { return transformVectorTransposed(n); }
    V4  worldToObjectVector(V4 v) const { return transformVectorTransposed(v) / scale(); }
    V4  worldToObjectPoint(V4 p) const { return worldToObjectVector(p - position); }
    V4  worldToFramePoint(V4 p) const { return worldToObjectNormal(p - position); }
    Ray worldToObjectRay(const Ray& ray) const {
        return Ray::init(worldToObjectPoint(ray.origin), worldToObjectVector(ray.direction), ray.min_t, ray.max_t);
    }
};

struct Intersection {  // shape/intersection.zig:46-62
    float    t, u, v;
    uint32_t primitive;
    Trafo    trafo;
};

struct Fragment {  // shape/intersection.zig:64-124 (event is always .Pass: no volumes in scope)
    Intersection isec;
    uint32_t     prop = kNull, part = 0;
    V4           p, geo_n, t, b, n, uvw;

    bool hit() const { return kNull != prop; }
    bool sameHemisphere(V4 v) const { return dot3(geo_n, v) > 0.f; }
};

// ray_offset.zig:14-31
inline V4 offsetRay(V4 p, V4 n) {
    const float origin = 1.f / 32.f, float_scale = 1.f / 65536.f, int_scale = 256.f;
    V4          r;
    for (int i = 0; i < 4; ++i) {
        const int32_t of_i = int32_t(int_scale * n[i]);
        int32_t       p_ii;
        std::memcpy(&p_ii, &p.v[i], 4);
        const int32_t in = int32_t(uint32_t(p_ii) - uint32_t(of_i)), ip = int32_t(uint32_t(p_ii) + uint32_t(of_i));
        float         p_in, p_ip;
        std::memcpy(&p_in, &in, 4);
        std::memcpy(&p_ip, &ip, 4);
        const float p_i = p[i] < 0.f ? p_in : p_ip;
        const float mad = std::fmaf(float_scale, n[i], p[i]);
        r[i] = std::fabs(p[i]) < origin ? mad : p_i;
    }
    r[3] = 0.f;
    return r;
}

inline V4 fragOffsetP(const Fragment& f, V4 v) {  // intersection.zig:115-119
    const V4 n = f.sameHemisphere(v) ? f.geo_n : -f.geo_n;
    return offsetRay(fma4(S(f.uvw[3]), n, f.p), n);
}
inline Ray fragOffsetRay(const Fragment& f, V4 dir, uint32_t time = 0) {
    Ray r = Ray::init(fragOffsetP(f, dir), dir, 0.f, kRayMaxT);
    r.time = time;
    return r;
}

struct Counters {  // what B_trav is computed from
    uint64_t camera_samples = 0, extension_rays = 0, shadow_rays = 0, nodes_fetched = 0, triangles_tested = 0,
             instances_tested = 0;
    void add(const Counters& o) {
        camera_samples += o.camera_samples;
        extension_rays += o.extension_rays;
        shadow_rays += o.shadow_rays;
        nodes_fetched += o.nodes_fetched;
        triangles_tested += o.triangles_tested;
        instances_tested += o.instances_tested;
    }
};

struct NodeStack {  // node_stack.zig
    uint32_t end = 0;
    uint32_t stack[127];
    void     push(uint32_t v) { stack[end++] = v; }
    uint32_t pop() { return 0 == end ? kNull : stack[--end]; }
};

// node.zig:73-87
inline float nodeIntersect(const ZygpuBvhNode& n, const Ray& ray) {
    const V4    lower = (mk(n.min[0], n.min[1], n.min[2], 0.f) - ray.origin) * ray.inv_direction;
    const V4    upper = (mk(n.max[0], n.max[1], n.max[2], 0.f) - ray.origin) * ray.inv_direction;
    const V4    t0 = min4(lower, upper), t1 = max4(lower, upper);
    const float tboxmin = hmax4(mk(t0[0], t0[1], t0[2], ray.min_t));
    const float tboxmax = hmin4(mk(t1[0], t1[1], t1[2], ray.max_t));
    return tboxmin <= tboxmax ? tboxmin : FLT_MAX;
}

// aabb.zig:46-60
inline bool aabbIntersect(const ZygpuAabb& a, const Ray& ray) {
    const V4    lower = (ld4(a.min) - ray.origin) * ray.inv_direction;
    const V4    upper = (ld4(a.max) - ray.origin) * ray.inv_direction;
    const V4    t0 = min4(lower, upper), t1 = max4(lower, upper);
    const float tboxmin = hmax4(mk(t0[0], t0[1], t0[2], ray.min_t));
    const float tboxmax = hmin4(mk(t1[0], t1[1], t1[2], ray.max_t));
    return tboxmin <= tboxmax;
}

// AABB.intersectP (base/math/aabb.zig:62-83): the entry distance, or the exit distance when the origin is inside
inline float aabbIntersectP(const ZygpuAabb& a, const Ray& ray) {
    const V4    lower = (ld4(a.min) - ray.origin) * ray.inv_direction;
    const V4    upper = (ld4(a.max) - ray.origin) * ray.inv_direction;
    const V4    t0 = min4(lower, upper), t1 = max4(lower, upper);
    const float imin = mx(mx(t0[0], t0[1]), t0[2]);
    const float imax = mn(mn(t1[0], t1[1]), t1[2]);
    const float tboxmin = mx(imin, ray.min_t);
    const float tboxmax = mn(imax, ray.max_t);
    if (tboxmin <= tboxmax) return imin < ray.min_t ? imax : imin;
    return FLT_MAX;
}

// ray_offset.zig:29-31
inline float offsetF(float t) {
    const float origin = 1.f / 32.f, float_scale = 1.f / 65536.f, int_scale = 256.f;
    if (t < origin) return t + float_scale;
    int32_t bits;
    std::memcpy(&bits, &t, 4);
    bits = int32_t(uint32_t(bits) + uint32_t(int32_t(int_scale)));
    float r;
    std::memcpy(&r, &bits, 4);
    return r;
}

// collision_coefficients.zig:67-69
inline V4 attenuation3(V4 c, float distance) {
    const V4 x = S(-distance) * c;
    return mk(expF(x[0]), expF(x[1]), expF(x[2]), 1.f);
}

// triangle.zig:26-52; returns hit flag
inline bool triangleIntersect(const Ray& ray, V4 a, V4 b, V4 c, float& ht, float& hu, float& hv) {
    const V4    e1 = b - a, e2 = c - a;
    const V4    tvec = ray.origin - a;
    const V4    pvec = cross3(ray.direction, e2);
    const V4    qvec = cross3(tvec, e1);
    const float e1_d_pv = dot3(e1, pvec), tv_d_pv = dot3(tvec, pvec);
    const float di_d_qv = dot3(ray.direction, qvec), e2_d_qv = dot3(e2, qvec);
    const float inv_det = 1.f / e1_d_pv;
    const float u = tv_d_pv * inv_det, v = di_d_qv * inv_det, hit_t = e2_d_qv * inv_det;
    const float uv = u + v;
    if (u >= 0.f && 1.f >= u && v >= 0.f && 1.f >= uv && hit_t >= ray.min_t && ray.max_t >= hit_t) {
        ht = hit_t;
        hu = u;
        hv = v;
        return true;
    }
    return false;
}

// ---- 2-D image textures: texture_sampler.zig:84-168 (Nearest2D / LinearStochastic2D), sampler_mode.zig:20-88, texture.zig:166-219
inline float texAddress(bool repeat, float x) { return repeat ? x - std::floor(x) : clampf(x, 0.f, 1.f); }
inline int   texCoord(bool repeat, int c, int end) {
    if (repeat) {
        const int m = c % end;  // @mod: the result has the sign of the divisor
        return m < 0 ? m + end : m;
    }
    return std::max(std::min(c, end - 1), 0);
}
// the texel a lookup reads: the nearest one, or the one of the four of the bilinear footprint that r picks
inline void texelPick(const ZygpuImage& img, uint32_t sampler, const float scale[2], float u, float v, float r, int& x, int& y) {
    const bool  ru = 0 != (sampler & ZYGPU_TEX_REPEAT_U), rv = 0 != (sampler & ZYGPU_TEX_REPEAT_V);
    const float su = scale[0] * u, sv = scale[1] * v;
    const float dw = float(img.width), dh = float(img.height);
    if (!(sampler & ZYGPU_TEX_LINEAR)) {
        x = std::min(int(texAddress(ru, su) * dw), img.width - 1);
        y = std::min(int(texAddress(rv, sv) * dh), img.height - 1);
        return;
    }
    const float mx_ = texAddress(ru, su) * dw - 0.5f, my_ = texAddress(rv, sv) * dh - 0.5f;
    const float fx = std::floor(mx_), fy = std::floor(my_);
    const float wx = mx_ - fx, wy = my_ - fy;
    const float ox = 1.f - wx, oy = 1.f - wy;
    int         index = 0;
    float       threshold = ox * oy;
    index += r > threshold ? 1 : 0;
    threshold = std::fmaf(wx, oy, threshold);
    index += r > threshold ? 1 : 0;
    threshold = std::fmaf(ox, wy, threshold);
    index += r > threshold ? 1 : 0;
    x = texCoord(ru, int(fx) + (index & 1), img.width);
    y = texCoord(rv, int(fy) + ((index & 2) >> 1), img.height);
}
inline float imageFetch1(const ZygpuImage& img, int x, int y) {  // texture.zig:166-175; unorm8ToFloat (base/encoding.zig:10-12)
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE1_UNORM == img.format) return float(static_cast<const uint8_t*>(img.pixels)[i]) * (1.f / 255.f);
    return static_cast<const float*>(img.pixels)[i];
}
// ts.sampleImage2D_1 (texture_sampler.zig:15-20)
inline float sampleImageMap1(const ZygpuSceneDesc& scene, uint32_t image, uint32_t sampler, const float scale[2], float u, float v, float r) {
    const ZygpuImage& img = scene.images[image];
    int               x, y;
    texelPick(img, sampler, scale, u, v, r, x, y);
    return imageFetch1(img, x, y);
}
// ts.sample2D_2 for an image texture (texture.zig:177-186): Byte2 as unorm or, for usage Normal, snorm (base/encoding.zig:10-20)
inline void sampleImageMap2(const ZygpuSceneDesc& scene, uint32_t image, uint32_t sampler, const float scale[2], float u, float v, float r,
                            float out[2]) {
    const ZygpuImage& img = scene.images[image];
    int               x, y;
    texelPick(img, sampler, scale, u, v, r, x, y);
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE2 == img.format) {
        const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + 2 * i;
        for (int c = 0; c < 2; ++c) out[c] = (sampler & ZYGPU_TEX_SNORM) ? std::fmaf(float(p[c]), 1.f / 128.f, -1.f) : float(p[c]) * (1.f / 255.f);
    } else {
        const float* p = static_cast<const float*>(img.pixels) + 2 * i;
        out[0] = p[0];
        out[1] = p[1];
    }
}
// the traversal's mask draws come from the path's sampler; the trace-only test entry points have none
inline float maskDraw(Sampler* sampler) { return sampler ? sampler->sample1D() : 0.f; }
// Base.stochasticOpacity (material_base.zig:44-55): does the closest-hit traversal accept this hit?
inline bool stochasticOpacity(const ZygpuSceneDesc& scene, const ZygpuMaterial& m, float u, float v, Sampler* sampler) {
    if (kNull == m.mask_image) return true;
    const float o = sampleImageMap1(scene, m.mask_image, m.mask_sampler, m.mask_scale, u, v, maskDraw(sampler));
    if (0.f == o || (o < 1.f && o <= maskDraw(sampler))) return false;
    return true;
}
// Base.opacity (material_base.zig:35-41)
inline float maskOpacity(const ZygpuSceneDesc& scene, const ZygpuMaterial& m, float u, float v, Sampler* sampler) {
    if (kNull == m.mask_image) return 1.f;
    return sampleImageMap1(scene, m.mask_image, m.mask_sampler, m.mask_scale, u, v, maskDraw(sampler));
}

struct MeshView {
    const ZygpuMesh* m;
    // triangle_data.zig:62-64: an unaligned 16-byte read; lane w is the next float in the array
    V4 position(uint32_t index) const { return ld4(m->positions + size_t(index) * 3); }
    V4 shadingNormal(uint32_t index) const {
        return octDecode(snorm16ToFloat(m->normals[2 * index]), snorm16ToFloat(m->normals[2 * index + 1]));
    }
};

// triangle_tree.zig:46-109 (closest hit; ray already in object space); accept(triangle, u, v) = intersectOpacity's
// stochastic mask test (triangle_tree.zig:111-195)
template <typename Accept>
inline bool meshIntersect(const ZygpuMesh& mesh, Ray local_ray, Intersection& isec, Counters& cnt, Accept&& accept) {
    const MeshView mv{&mesh};
    NodeStack      stack;
    uint32_t       n = 0;
    float          ht = 0.f, hu = 0.f, hv = 0.f;
    uint32_t       primitive = kNull;
    cnt.nodes_fetched += 1;  // root
    while (kNull != n) {
        const ZygpuBvhNode& node = mesh.nodes[n];
        if (0 != node.count) {
            for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                cnt.triangles_tested += 1;
                const uint32_t* tri = mesh.triangles + size_t(i) * 3;
                float           t, u, v;
                if (triangleIntersect(local_ray, mv.position(tri[0]), mv.position(tri[1]), mv.position(tri[2]), t, u, v) && accept(i, u, v)) {
                    local_ray.max_t = t;
                    ht = t;
                    hu = u;
                    hv = v;
                    primitive = i;
                }
            }
            n = stack.pop();
            continue;
        }
        uint32_t a = node.children_or_first, b = a + 1;
        cnt.nodes_fetched += 2;
        float dista = nodeIntersect(mesh.nodes[a], local_ray);
        float distb = nodeIntersect(mesh.nodes[b], local_ray);
        if (dista > distb) {
            std::swap(a, b);
            std::swap(dista, distb);
        }
        if (FLT_MAX == dista) {
            n = stack.pop();
        } else {
            n = a;
            if (FLT_MAX != distb) stack.push(b);
        }
    }
    if (kNull == primitive) return false;
    isec.t = ht;
    isec.u = hu;
    isec.v = hv;
    isec.primitive = primitive;
    return true;
}

inline bool meshIntersect(const ZygpuMesh& mesh, Ray local_ray, Intersection& isec, Counters& cnt) {
    return meshIntersect(mesh, local_ray, isec, cnt, [](uint32_t, float, float) { return true; });
}
inline void meshInterpolateUv(const ZygpuMesh& mesh, uint32_t prim, float u, float v, float& uv0, float& uv1) {  // triangle_data.zig:137-149
    const uint32_t* tri = mesh.triangles + size_t(prim) * 3;
    const float     w = 1.f - u - v;
    uv0 = std::fmaf(mesh.uvs[2 * tri[0]], w, std::fmaf(mesh.uvs[2 * tri[1]], u, mesh.uvs[2 * tri[2]] * v));
    uv1 = std::fmaf(mesh.uvs[2 * tri[0] + 1], w, std::fmaf(mesh.uvs[2 * tri[1] + 1], u, mesh.uvs[2 * tri[2] + 1] * v));
}

// triangle_tree.zig:197-242 (any hit)
inline bool meshIntersectP(const ZygpuMesh& mesh, const Ray& ray, Counters& cnt) {
    const MeshView mv{&mesh};
    NodeStack      stack;
    uint32_t       n = 0;
    cnt.nodes_fetched += 1;
    while (kNull != n) {
        const ZygpuBvhNode& node = mesh.nodes[n];
        if (0 != node.count) {
            for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                cnt.triangles_tested += 1;
                const uint32_t* tri = mesh.triangles + size_t(i) * 3;
                float           t, u, v;
                if (triangleIntersect(ray, mv.position(tri[0]), mv.position(tri[1]), mv.position(tri[2]), t, u, v)) return true;
            }
            n = stack.pop();
            continue;
        }
        uint32_t a = node.children_or_first, b = a + 1;
        cnt.nodes_fetched += 2;
        float dista = nodeIntersect(mesh.nodes[a], ray);
        float distb = nodeIntersect(mesh.nodes[b], ray);
        if (dista > distb) {
            std::swap(a, b);
            std::swap(dista, distb);
        }
        if (FLT_MAX == dista) {
            n = stack.pop();
        } else {
            n = a;
            if (FLT_MAX != distb) stack.push(b);
        }
    }
    return false;
}

// Glass.visibility (glass_material.zig:69-115): thin glass = Fresnel transmission x absorption over the slab, let through
// further by the mask's openings; o = the mask's opacity at the hit; false = blocked
inline bool glassVisibility(const ZygpuMaterial& m, V4 wi, V4 geo_n, float o, V4& tr) {
    if (m.thickness > 0.f) {
        const float eta_i = 1.f, eta_t = m.ior;
        const float n_dot_wo = mn(std::fabs(dot3(geo_n, wi)), 1.f);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - n_dot_wo * n_dot_wo);
        if (sint2 >= 1.f) {
            if (o < 1.f) {
                tr *= S(1.f - o);
                return true;
            }
            return false;
        }
        const float n_dot_t = std::sqrt(1.f - sint2);
        const float t0 = eta_t * n_dot_wo, t1 = eta_i * n_dot_t;  // fresnel.dielectric (fresnel.zig:31-43)
        const float r_p = (t0 - t1) / (t0 + t1);
        const float t2 = eta_i * n_dot_wo, t3 = eta_t * n_dot_t;
        const float r_o = (t2 - t3) / (t2 + t3);
        const float f = 0.5f * (r_p * r_p + r_o * r_o);
        const float n_dot_wi = clampf(n_dot_wo, 0.00001f, 1.f);
        const float approx_dist = m.thickness / n_dot_wi;
        const V4    attenuation = attenuation3(ld3(m.absorption), approx_dist);
        const V4    ta = min4(S(1.f - o) + attenuation, S(1.f));
        tr *= S(1.f - f) * ta;
        return true;
    }
    if (o < 1.f) {
        tr *= S(1.f - o);
        return true;
    }
    return false;
}
// Material.visibility (material.zig:227-248): the mask's opacity, drawn with the path's sampler, scales the transmittance
inline bool materialVisibility(const ZygpuSceneDesc& scene, const ZygpuMaterial& m, V4 wi, V4 geo_n, float u, float v, Sampler* sampler, V4& tr) {
    const float o = maskOpacity(scene, m, u, v, sampler);
    if (ZYGPU_MAT_GLASS == m.kind) return glassVisibility(m, wi, geo_n, o, tr);
    if (o < 1.f) {
        tr *= S(1.f - o);
        return true;
    }
    return false;
}

// TriangleTree.visibility (triangle_tree.zig:244-306): every triangle the ray crosses multiplies the transmittance
template <typename Material>
inline bool meshVisibility(const ZygpuSceneDesc& scene, const ZygpuMesh& mesh, const Ray& ray, Counters& cnt, V4& tr, Sampler* sampler,
                           Material&& materialOfPart) {
    const MeshView mv{&mesh};
    NodeStack      stack;
    uint32_t       n = 0;
    cnt.nodes_fetched += 1;
    while (kNull != n) {
        const ZygpuBvhNode& node = mesh.nodes[n];
        if (0 != node.count) {
            for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                cnt.triangles_tested += 1;
                const uint32_t* tri = mesh.triangles + size_t(i) * 3;
                const V4        a = mv.position(tri[0]), b = mv.position(tri[1]), c = mv.position(tri[2]);
                float           t, u, v;
                if (!triangleIntersect(ray, a, b, c, t, u, v)) continue;
                const ZygpuMaterial& m = materialOfPart(mesh.triangle_parts[i]);
                if (!(m.flags & ZYGPU_MATF_EVALUATE_VISIBILITY)) return false;
                const V4 geo_n = normalize3(cross3(b - a, c - a));  // data.normal (triangle_data.zig:129-135), object space
                float    uv0 = 0.f, uv1 = 0.f;
                if (kNull != m.mask_image) meshInterpolateUv(mesh, i, u, v, uv0, uv1);
                if (!materialVisibility(scene, m, ray.direction, geo_n, uv0, uv1, sampler, tr)) return false;
            }
            n = stack.pop();
            continue;
        }
        uint32_t a = node.children_or_first, b = a + 1;
        cnt.nodes_fetched += 2;
        float dista = nodeIntersect(mesh.nodes[a], ray);
        float distb = nodeIntersect(mesh.nodes[b], ray);
        if (dista > distb) {
            std::swap(a, b);
            std::swap(dista, distb);
        }
        if (FLT_MAX == dista) {
            n = stack.pop();
        } else {
            n = a;
            if (FLT_MAX != distb) stack.push(b);
        }
    }
    return true;
}

// triangle_tree.zig:405-483 (emission): every triangle the ray crosses inside [min_t, max_t], in traversal order
template <typename Visit>
inline void meshEmissionWalk(const ZygpuMesh& mesh, const Ray& ray, Counters& cnt, Visit&& visit) {
    const MeshView mv{&mesh};
    NodeStack      stack;
    uint32_t       n = 0;
    cnt.nodes_fetched += 1;
    while (kNull != n) {
        const ZygpuBvhNode& node = mesh.nodes[n];
        if (0 != node.count) {
            for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                cnt.triangles_tested += 1;
                const uint32_t* tri = mesh.triangles + size_t(i) * 3;
                float           t, u, v;
                if (triangleIntersect(ray, mv.position(tri[0]), mv.position(tri[1]), mv.position(tri[2]), t, u, v)) visit(i, t, u, v);
            }
            n = stack.pop();
            continue;
        }
        uint32_t a = node.children_or_first, b = a + 1;
        cnt.nodes_fetched += 2;
        float dista = nodeIntersect(mesh.nodes[a], ray);
        float distb = nodeIntersect(mesh.nodes[b], ray);
        if (dista > distb) {
            std::swap(a, b);
            std::swap(dista, distb);
        }
        if (FLT_MAX == dista) {
            n = stack.pop();
        } else {
            n = a;
            if (FLT_MAX != distb) stack.push(b);
        }
    }
}

// rectangle.zig:30-62 / 126-152 (intersect and intersectP share the test)
inline bool rectangleIntersect(const Ray& ray, const Trafo& trafo, float& out_t, float& out_u, float& out_v) {
    const V4    n = trafo.r[2];
    const float d = dot3(n, trafo.position);
    const float hit_t = -(dot3(n, ray.origin) - d) / dot3(n, ray.direction);
    if (hit_t >= ray.min_t && ray.max_t >= hit_t) {
        const V4    p = ray.point(hit_t);
        const V4    k = p - trafo.position;
        const V4    t = -trafo.r[0];
        const float u = dot3(t, k) / (0.5f * trafo.scaleX());
        if (u > 1.f || u < -1.f) return false;
        const V4    b = -trafo.r[1];
        const float v = dot3(b, k) / (0.5f * trafo.scaleY());
        if (v > 1.f || v < -1.f) return false;
        out_t = hit_t;
        out_u = u;
        out_v = v;
        return true;
    }
    return false;
}

// sphere.zig:28-60 (nearer root first); unit-diameter sphere scaled by trafo.scaleX()
inline bool sphereIntersect(const Ray& ray, const Trafo& trafo, float& out_t) {
    const float idl = 1.f / length3(ray.direction);
    const V4    nd = ray.direction * S(idl);
    const V4    v = trafo.position - ray.origin;
    const float b = dot3(nd, v);
    const V4    remedy_term = v - S(b) * nd;
    const float radius = 0.5f * trafo.scaleX();
    const float discriminant = radius * radius - dot3(remedy_term, remedy_term);
    if (discriminant > 0.f) {
        const float dist = std::sqrt(discriminant);
        const float t0 = (b - dist) * idl;
        if (t0 >= ray.min_t && ray.max_t >= t0) {
            out_t = t0;
            return true;
        }
        const float t1 = (b + dist) * idl;
        if (t1 >= ray.min_t && ray.max_t >= t1) {
            out_t = t1;
            return true;
        }
    }
    return false;
}

// sphere.zig:62-90
// the uv of a point on the sphere with world normal n (sphere.zig:110-115)
inline void sphereUv(const Trafo& trafo, V4 n, float& u, float& v) {
    const V4    xyz = normalize3(trafo.worldToObjectNormal(n));
    const float phi = -atan2F(xyz[0], xyz[2]) + kPi;
    const float theta = acosF(xyz[1]);
    u = phi * (0.5f * kPiInv);
    v = theta * kPiInv;
}
// sphere.zig:94-146: the nearer root the mask lets through
inline bool sphereIntersectOpacity(const ZygpuSceneDesc& scene, const ZygpuMaterial& m, const Ray& ray, const Trafo& trafo, float& out_t,
                                   Sampler* sampler) {
    const float idl = 1.f / length3(ray.direction);
    const V4    nd = ray.direction * S(idl);
    const V4    v = trafo.position - ray.origin;
    const float b = dot3(nd, v);
    const V4    remedy_term = v - S(b) * nd;
    const float radius = 0.5f * trafo.scaleX();
    const float discriminant = radius * radius - dot3(remedy_term, remedy_term);
    if (discriminant > 0.f) {
        const float dist = std::sqrt(discriminant);
        for (const float t : {(b - dist) * idl, (b + dist) * idl}) {
            if (t >= ray.min_t && ray.max_t >= t) {
                float uu, vv;
                sphereUv(trafo, normalize3(ray.point(t) - trafo.position), uu, vv);
                if (stochasticOpacity(scene, m, uu, vv, sampler)) {
                    out_t = t;
                    return true;
                }
            }
        }
    }
    return false;
}

inline void sphereFragment(const Ray& ray, Fragment& frag) {
    const Trafo& tr = frag.isec.trafo;
    const V4     p = ray.point(frag.isec.t);
    const V4     n = normalize3(p - tr.position);
    frag.p = p;
    frag.geo_n = n;
    frag.n = n;
    frag.part = 0;
    const V4    xyz = normalize3(tr.worldToObjectNormal(n));
    const float phi = -atan2F(xyz[0], xyz[2]) + kPi;
    const float theta = acosF(xyz[1]);
    const float sin_phi = sinF(phi), cos_phi = cosF(phi);
    const float sin_theta = mx(sinF(theta), 0.00001f);  // avoid the singularity at the poles
    const V4    t = normalize3(tr.objectToWorldNormal(mk(sin_theta * cos_phi, 0.f, sin_theta * sin_phi)));
    frag.t = t;
    frag.b = -cross3(t, n);
    frag.uvw = mk(phi * (0.5f * kPiInv), theta * kPiInv, 0.f, 0.f);
}

// rectangle.zig:102-124
inline void rectangleFragment(const Ray& ray, Fragment& frag) {
    const Trafo& tr = frag.isec.trafo;
    const V4     p = ray.point(frag.isec.t);
    const V4     n = tr.r[2], t = -tr.r[0], b = -tr.r[1];
    frag.p = p;
    frag.t = t;
    frag.b = b;
    frag.n = n;
    frag.geo_n = n;
    if (tr.scaleZ() < 0.f) {
        const V4    k = p - tr.position;
        const float u = dot3(t, k) * 2.f, v = dot3(b, k) * 2.f;
        frag.uvw = mk(0.5f * (u + 1.f), 0.5f * (v + 1.f), 0.f, 0.f);
    } else {
        frag.uvw = mk(0.5f * (frag.isec.u + 1.f), 0.5f * (frag.isec.v + 1.f), 0.f, 0.f);
    }
    frag.part = 0;
}

// ---- Disk (disk.zig:28-60, 100-116) and Cube (cube.zig:23-67): geometry only
inline bool diskIntersect(const Ray& ray, const Trafo& trafo, float& out_t, float& out_u, float& out_v) {
    const V4    normal = trafo.r[2];
    const float d = dot3(normal, trafo.position);
    const float denom = -dot3(normal, ray.direction);
    const float numer = dot3(normal, ray.origin) - d;
    const float hit_t = numer / denom;
    if (hit_t >= ray.min_t && ray.max_t >= hit_t) {
        const V4    k = ray.point(hit_t) - trafo.position;
        const float l = dot3(k, k);
        const float radius = 0.5f * trafo.scaleX();
        if (l <= radius * radius) {
            const V4 sk = k / S(radius);
            out_u = -dot3(trafo.r[0], sk);
            out_v = -dot3(trafo.r[1], sk);
            out_t = hit_t;
            return true;
        }
    }
    return false;
}
inline void diskFragment(const Ray& ray, Fragment& frag) {
    const Trafo& tr = frag.isec.trafo;
    frag.p = ray.point(frag.isec.t);
    frag.t = -tr.r[0];
    frag.b = -tr.r[1];
    frag.n = frag.geo_n = tr.r[2];
    frag.uvw = mk(0.5f * (frag.isec.u + 1.f), 0.5f * (frag.isec.v + 1.f), 0.f, 0.f);
    frag.part = 0;
}
inline ZygpuAabb unitCube() {
    ZygpuAabb a;
    for (int i = 0; i < 4; ++i) {
        a.min[i] = -0.5f;
        a.max[i] = 0.5f;
    }
    return a;
}
inline bool cubeIntersect(const Ray& ray, const Trafo& trafo, float& out_t) {
    const float hit_t = aabbIntersectP(unitCube(), trafo.worldToObjectRay(ray));
    if (hit_t < ray.max_t) {
        out_t = hit_t;
        return true;
    }
    return false;
}
inline bool cubeIntersectP(const Ray& ray, const Trafo& trafo) { return aabbIntersect(unitCube(), trafo.worldToObjectRay(ray)); }
inline void cubeFragment(const Ray& ray, Fragment& frag) {
    const Trafo& tr = frag.isec.trafo;
    const float  hit_t = frag.isec.t;
    frag.p = ray.point(hit_t);
    const V4 local_p = tr.worldToObjectRay(ray).point(hit_t);
    float    distance[3];
    for (int i = 0; i < 3; ++i) distance[i] = std::fabs(0.5f - std::fabs(local_p[i]));
    const int   i = distance[0] < distance[1] ? (distance[0] < distance[2] ? 0 : 2) : (distance[1] < distance[2] ? 1 : 2);  // indexMinComponent3
    const float sgn = std::copysign(1.f, local_p[i]);
    const V4    n = S(sgn) * tr.r[i];
    frag.part = 0;
    frag.geo_n = frag.n = n;
    frag.uvw = S(0.f);
    orthonormalBasis3(n, frag.t, frag.b);
}

// triangle.zig:102-149
inline void interpolate2(const float* a, const float* b, const float* c, float u, float v, float out[2]) {
    const float w = 1.f - u - v;
    for (int i = 0; i < 2; ++i) out[i] = std::fmaf(a[i], w, std::fmaf(b[i], u, c[i] * v));
}
inline V4 interpolate3(V4 a, V4 b, V4 c, float u, float v) {
    const float w = 1.f - u - v;
    return fma4(a, S(w), fma4(b, S(u), c * S(v)));
}
inline void positionDifferentials(V4 pa, V4 pb, V4 pc, const float* uva, const float* uvb, const float* uvc, V4& dpdu, V4& dpdv) {
    const float duv02[2] = {uva[0] - uvc[0], uva[1] - uvc[1]};
    const float duv12[2] = {uvb[0] - uvc[0], uvb[1] - uvc[1]};
    const float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    const V4    dp02 = pa - pc, dp12 = pb - pc;
    if (0.f == std::fabs(determinant)) {
        const V4 ng = normalize3(cross3(pc - pa, pb - pa));
        if (std::fabs(ng[0]) > std::fabs(ng[1])) {
            dpdu = mk(-ng[2], 0.f, ng[0], 0.f) / S(std::sqrt(ng[0] * ng[0] + ng[2] * ng[2]));
        } else {
            dpdu = mk(0.f, ng[2], -ng[1], 0.f) / S(std::sqrt(ng[1] * ng[1] + ng[2] * ng[2]));
        }
        dpdv = cross3(ng, dpdu);
    } else {
        const float invdet = 1.f / determinant;
        dpdu = S(invdet) * fma4(S(duv12[1]), dp02, S(-duv02[1]) * dp12);
        dpdv = S(invdet) * fma4(S(-duv12[0]), dp02, S(duv02[0]) * dp12);
    }
}

// triangle_mesh.zig:310-335 + triangle_data.zig:106-149
inline void meshFragment(const ZygpuMesh& mesh, Fragment& frag) {
    const MeshView mv{&mesh};
    const uint32_t prim = frag.isec.primitive;
    frag.part = mesh.triangle_parts[prim];
    const float     hit_u = frag.isec.u, hit_v = frag.isec.v;
    const uint32_t* tri = mesh.triangles + size_t(prim) * 3;
    const V4        pa = mv.position(tri[0]), pb = mv.position(tri[1]), pc = mv.position(tri[2]);

    const V4 geo_n = normalize3(cross3(pb - pa, pc - pa));
    frag.geo_n = frag.isec.trafo.objectToWorldNormal(geo_n);

    const V4     p = interpolate3(pa, pb, pc, hit_u, hit_v);
    const float* uva = mesh.uvs + size_t(tri[0]) * 2;
    const float* uvb = mesh.uvs + size_t(tri[1]) * 2;
    const float* uvc = mesh.uvs + size_t(tri[2]) * 2;
    float        uv[2];
    interpolate2(uva, uvb, uvc, hit_u, hit_v, uv);

    const V4 nb = mv.shadingNormal(tri[1]), na = mv.shadingNormal(tri[0]), nc = mv.shadingNormal(tri[2]);
    const V4 ni = normalize3(interpolate3(na, nb, nc, hit_u, hit_v));

    V4 dpdu, dpdv;
    positionDifferentials(pa, pb, pc, uva, uvb, uvc, dpdu, dpdv);
    const V4 t = normalize3(gramSchmidt(dpdu, ni));
    const V4 b = normalize3(gramSchmidt(dpdv, ni));

    const Trafo& tr = frag.isec.trafo;
    frag.p = tr.objectToWorldPoint(p);
    frag.t = tr.objectToWorldNormal(t);
    frag.b = tr.objectToWorldNormal(b);
    frag.n = tr.objectToWorldNormal(ni);
    frag.uvw = mk(uv[0], uv[1], 0.f, 0.f);
}

// math.Transformation.lerp (transformation.zig:36-44: lerp, lerp, quaternion.slerp 217-245), origin subtracted after the
// interpolation (space.zig:121-124), ComposedTransformation.init with quaternion.toMat3x3 (quaternion.zig:72-116)
inline Trafo interpolateKeyframes(const ZygpuKeyframe& ka, const ZygpuKeyframe& kb, float wgt, V4 origin) {
    const V4 t4 = S(wgt);
    const V4 position = lerp4(ld4(ka.position), ld4(kb.position), t4) - origin;
    const V4 scale = lerp4(ld4(ka.scale), ld4(kb.scale), t4);
    const V4 qa = ld4(ka.rotation);
    V4       end = ld4(kb.rotation);
    const V4 ab = qa * end;
    float    cosom = ab[0] + ab[1] + ab[2] + ab[3];
    if (cosom < 0.f) {
        cosom = -cosom;
        end = -end;
    }
    float sclp, sclq;
    if (1.f - cosom > 0.00001f) {
        const float omega = acosF(cosom), sinom = sinF(omega);
        sclp = sinF((1.f - wgt) * omega) / sinom;
        sclq = sinF(wgt * omega) / sinom;
    } else {
        sclp = 1.f - wgt;
        sclq = wgt;
    }
    const V4    q = S(sclp) * qa + S(sclq) * end;
    const V4    tq = q + q;
    const float xy = tq[1] * q[0], xz = tq[2] * q[0], yz = tq[1] * q[2];
    const V4    wq = tq * S(q[3]);
    const V4    a = mk(q[3], q[0], q[3], q[1]), b = mk(q[1], q[2], q[0], q[2]);
    const V4    t = (a + b) * (a - b);
    Trafo       out;
    out.r[0] = mk(t[0] + t[1], xy - wq[2], xz + wq[1], scale[0]);
    out.r[1] = mk(xy + wq[2], t[2] + t[3], yz - wq[0], scale[1]);
    out.r[2] = mk(xz - wq[1], yz + wq[0], t[2] - t[3], scale[2]);
    out.position = position;
    return out;
}

struct SceneView {
    const ZygpuSceneDesc* d;

    Trafo propTrafo(uint32_t p) const { return Trafo::from(d->trafos[p]); }
    // Scene.propTransformationAtMaybeStatic (scene.zig:382-385 -> space.zig:99-137): a static prop's transformation, or an
    // animated prop's keyframes interpolated at `time` (ticks since the frame's start): frameAt, Transformation.lerp with
    // quaternion.slerp, the scene origin subtracted, ComposedTransformation.init
    Trafo propTrafo(uint32_t p, uint32_t time) const {
        if (!d->prop_frames || ZYGPU_NULL == d->prop_frames[p]) return propTrafo(p);
        const uint32_t tick = uint32_t(705600000ull / 60);
        uint32_t       i = time / tick;
        float          wgt = float(double(time - i * tick) / double(tick));
        if (i + 1 >= d->num_interpolation_frames) {  // a shutter sample of exactly 1.0: the last keyframe
            i = d->num_interpolation_frames - 2;
            wgt = 1.f;
        }
        const ZygpuKeyframe& ka = d->keyframes[d->prop_frames[p] + i];
        const ZygpuKeyframe& kb = d->keyframes[d->prop_frames[p] + i + 1];
        return interpolateKeyframes(ka, kb, wgt, ld4(d->origin));
    }
    const ZygpuMaterial& propMaterial(uint32_t prop, uint32_t part) const {
        return d->materials[d->material_ids[d->props[prop].parts_start + part]];
    }
    uint32_t propLightId(uint32_t prop, uint32_t part) const { return d->light_ids[d->props[prop].parts_start + part]; }

    // prop.zig:39-49
    static bool visible(uint32_t flags, uint32_t depth_surface) {
        if (0 == depth_surface) return 0 != (flags & ZYGPU_PROP_VISIBLE_IN_CAMERA);
        return 0 != (flags & ZYGPU_PROP_VISIBLE_IN_REFLECTION);
    }

    // prop.zig:163-197
    // `sampler`: props with evaluate_visibility take shape.intersectOpacity (prop.zig:191-195), whose stochastic mask tests
    // draw from the path's sampler (rectangle.zig:64-100, sphere.zig:94-146, disk.zig:60-96, triangle_tree.zig:111-195)
    bool propIntersect(uint32_t p, const Ray& ray, uint32_t depth_surface, Intersection& isec, Counters& cnt, Sampler* sampler = nullptr) const {
        const ZygpuProp& prop = d->props[p];
        if (!visible(prop.flags, depth_surface)) return false;
        cnt.instances_tested += 1;
        if (!aabbIntersect(d->aabbs[p], ray)) return false;
        const Trafo trafo = propTrafo(p, ray.time);
        const bool  masked = 0 != (prop.flags & ZYGPU_PROP_EVALUATE_VISIBILITY);
        if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {
            float t, u, v;
            if (!rectangleIntersect(ray, trafo, t, u, v)) return false;
            if (masked && !stochasticOpacity(*d, propMaterial(p, 0), 0.5f * (u + 1.f), 0.5f * (v + 1.f), sampler)) return false;
            isec.t = t;
            isec.u = u;
            isec.v = v;
            isec.primitive = 0;
            isec.trafo = trafo;
            return true;
        }
        if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
            const ZygpuMesh& mesh = d->meshes[prop.mesh];
            const bool       found = masked ? meshIntersect(mesh, trafo.worldToObjectRay(ray), isec, cnt,
                                                            [&](uint32_t k, float u, float v) {
                                                                const ZygpuMaterial& m = propMaterial(p, mesh.triangle_parts[k]);
                                                                if (!(m.flags & ZYGPU_MATF_EVALUATE_VISIBILITY)) return true;
                                                                float uv0, uv1;
                                                                meshInterpolateUv(mesh, k, u, v, uv0, uv1);
                                                                return stochasticOpacity(*d, m, uv0, uv1, sampler);
                                                            })
                                            : meshIntersect(mesh, trafo.worldToObjectRay(ray), isec, cnt);
            if (!found) return false;
            isec.trafo = trafo;
            return true;
        }
        if (ZYGPU_SHAPE_SPHERE == prop.shape || ZYGPU_SHAPE_CUBE == prop.shape) {
            float t;
            if (!(ZYGPU_SHAPE_SPHERE == prop.shape
                      ? (masked ? sphereIntersectOpacity(*d, propMaterial(p, 0), ray, trafo, t, sampler) : sphereIntersect(ray, trafo, t))
                      : cubeIntersect(ray, trafo, t)))
                return false;
            isec.t = t;
            isec.u = isec.v = 0.f;
            isec.primitive = 0;
            isec.trafo = trafo;
            return true;
        }
        if (ZYGPU_SHAPE_DISK == prop.shape) {
            float t, u, v;
            if (!diskIntersect(ray, trafo, t, u, v)) return false;
            if (masked && !stochasticOpacity(*d, propMaterial(p, 0), 0.5f * (u + 1.f), 0.5f * (v + 1.f), sampler)) return false;
            isec.t = t;
            isec.u = u;
            isec.v = v;
            isec.primitive = 0;
            isec.trafo = trafo;
            return true;
        }
        return false;
    }

    // prop.zig:199-237 (non-volumetric): true = unoccluded
    // `tr`: the transmittance of the thin-glass and masked surfaces the ray crosses (shape.visibility for props with
    // evaluate_visibility); the masks' stochastic texel picks draw from `sampler`
    bool propVisibility(uint32_t p, const Ray& ray, Counters& cnt, V4& tr, Sampler* sampler = nullptr) const {
        const ZygpuProp& prop = d->props[p];
        if (0 == (prop.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW)) return true;
        cnt.instances_tested += 1;
        if (!aabbIntersect(d->aabbs[p], ray)) return true;
        const Trafo trafo = propTrafo(p, ray.time);
        if (prop.flags & ZYGPU_PROP_EVALUATE_VISIBILITY) {
            if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {  // rectangle.zig:154-186
                float t, u, v;
                if (!rectangleIntersect(ray, trafo, t, u, v)) return true;
                return materialVisibility(*d, propMaterial(p, 0), ray.direction, trafo.r[2], 0.5f * (u + 1.f), 0.5f * (v + 1.f), sampler, tr);
            }
            if (ZYGPU_SHAPE_DISK == prop.shape) {  // disk.zig:136-170
                float t, u, v;
                if (!diskIntersect(ray, trafo, t, u, v)) return true;
                return materialVisibility(*d, propMaterial(p, 0), ray.direction, trafo.r[2], 0.5f * (1.f + u), 0.5f * (1.f + v), sampler, tr);
            }
            if (ZYGPU_SHAPE_SPHERE == prop.shape) {  // sphere.zig:176-228: both roots
                const float idl = 1.f / length3(ray.direction);
                const V4    nd = ray.direction * S(idl);
                const V4    v = trafo.position - ray.origin;
                const float b = dot3(nd, v);
                const V4    remedy_term = v - S(b) * nd;
                const float radius = 0.5f * trafo.scaleX();
                const float discriminant = radius * radius - dot3(remedy_term, remedy_term);
                if (discriminant > 0.f) {
                    const float dist = std::sqrt(discriminant);
                    for (const float t : {(b - dist) * idl, (b + dist) * idl}) {
                        if (t >= ray.min_t && ray.max_t >= t) {
                            const V4 n = normalize3(ray.point(t) - trafo.position);
                            float    uu = 0.f, vv = 0.f;
                            if (kNull != propMaterial(p, 0).mask_image) sphereUv(trafo, n, uu, vv);
                            if (!materialVisibility(*d, propMaterial(p, 0), ray.direction, n, uu, vv, sampler, tr)) return false;
                        }
                    }
                }
                return true;
            }
            if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
                return meshVisibility(*d, d->meshes[prop.mesh], trafo.worldToObjectRay(ray), cnt, tr, sampler,
                                      [&](uint32_t part) -> const ZygpuMaterial& { return propMaterial(p, part); });
            }
            return true;
        }
        if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {
            float t, u, v;
            return !rectangleIntersect(ray, trafo, t, u, v);
        }
        if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
            return !meshIntersectP(d->meshes[prop.mesh], trafo.worldToObjectRay(ray), cnt);
        }
        if (ZYGPU_SHAPE_SPHERE == prop.shape) {  // sphere.zig:148-172 intersectP: the same two roots
            float t;
            return !sphereIntersect(ray, trafo, t);
        }
        if (ZYGPU_SHAPE_DISK == prop.shape) {  // disk.zig:115-133
            float t, u, v;
            return !diskIntersect(ray, trafo, t, u, v);
        }
        if (ZYGPU_SHAPE_CUBE == prop.shape) return !cubeIntersectP(ray, trafo);  // cube.zig:63-67
        return true;
    }

    void shapeFragment(uint32_t prop, const Ray& ray, Fragment& frag) const {
        const ZygpuProp& p = d->props[prop];
        if (ZYGPU_SHAPE_RECTANGLE == p.shape) rectangleFragment(ray, frag);
        else if (ZYGPU_SHAPE_SPHERE == p.shape) sphereFragment(ray, frag);
        else if (ZYGPU_SHAPE_DISK == p.shape) diskFragment(ray, frag);
        else if (ZYGPU_SHAPE_CUBE == p.shape) cubeFragment(ray, frag);
        else meshFragment(d->meshes[p.mesh], frag);
    }

    // prop_tree.zig:56-116 over the solid TLAS; shrinks ray.max_t
    bool intersect(Ray& ray, uint32_t depth_surface, Fragment& frag, Counters& cnt, Sampler* sampler = nullptr) const {
        const ZygpuPropTree& tree = d->solid;
        NodeStack            stack;
        uint32_t             n = 0 == tree.num_nodes ? kNull : 0;
        Intersection         isec{};
        uint32_t             prop = kNull;
        if (kNull != n) cnt.nodes_fetched += 1;
        while (kNull != n) {
            const ZygpuBvhNode& node = tree.nodes[n];
            if (0 != node.count) {
                for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                    const uint32_t p = tree.indices[i];
                    Intersection   candidate = isec;
                    if (propIntersect(p, ray, depth_surface, candidate, cnt, sampler)) {
                        isec = candidate;
                        ray.max_t = isec.t;
                        prop = p;
                    }
                }
                n = stack.pop();
                continue;
            }
            uint32_t a = node.children_or_first, b = a + 1;
            cnt.nodes_fetched += 2;
            float dista = nodeIntersect(tree.nodes[a], ray);
            float distb = nodeIntersect(tree.nodes[b], ray);
            if (dista > distb) {
                std::swap(a, b);
                std::swap(dista, distb);
            }
            if (FLT_MAX == dista) {
                n = stack.pop();
            } else {
                n = a;
                if (FLT_MAX != distb) stack.push(b);
            }
        }
        const bool hit = kNull != prop;
        if (hit) {
            frag.isec = isec;
            shapeFragment(prop, ray, frag);
        }
        frag.prop = prop;
        return hit;
    }

    // prop_tree.zig:185-240 (solid TLAS; the volume TLAS is empty in scope, scene.zig:229-235)
    bool visibility(const Ray& ray, Counters& cnt) const {
        V4 tr = S(1.f);
        return visibility(ray, cnt, tr);
    }
    bool visibility(const Ray& ray, Counters& cnt, V4& tr, Sampler* sampler = nullptr) const {
        const ZygpuPropTree& tree = d->solid;
        NodeStack            stack;
        uint32_t             n = 0 == tree.num_nodes ? kNull : 0;
        if (kNull != n) cnt.nodes_fetched += 1;
        while (kNull != n) {
            const ZygpuBvhNode& node = tree.nodes[n];
            if (0 != node.count) {
                for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                    if (!propVisibility(tree.indices[i], ray, cnt, tr, sampler)) return false;
                }
                n = stack.pop();
                continue;
            }
            uint32_t a = node.children_or_first, b = a + 1;
            cnt.nodes_fetched += 2;
            float dista = nodeIntersect(tree.nodes[a], ray);
            float distb = nodeIntersect(tree.nodes[b], ray);
            if (dista > distb) {
                std::swap(a, b);
                std::swap(dista, distb);
            }
            if (FLT_MAX == dista) {
                n = stack.pop();
            } else {
                n = a;
                if (FLT_MAX != distb) stack.push(b);
            }
        }
        return true;
    }
};

}  // namespace orc
