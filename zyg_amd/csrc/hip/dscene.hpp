// Device view of the compiled scene and the ray casting over it.  Behaviour follows the reference:
//   bvh.Node.intersect           src/core/scene/bvh/node.zig:73-87
//   AABB.intersect               src/base/math/aabb.zig:46-60
//   PropBvh.intersect/visibility src/core/scene/prop/prop_tree.zig:56-116,185-240
//   Prop.intersect/visibility    src/core/scene/prop/prop.zig:163-237
//   TriangleTree.intersect(P)    src/core/scene/shape/triangle/triangle_tree.zig:46-109,197-242
//   triangle.intersect           src/core/scene/shape/triangle/triangle.zig:26-52
//   Rectangle.intersect/fragment src/core/scene/shape/rectangle.zig:30-62,102-124
//   TriangleMesh.fragment        src/core/scene/shape/triangle/triangle_mesh.zig:310-335, triangle_data.zig:106-130
// HBM layout: BVH2 nodes stay in the reference's 32 B form (two float4), so that one inner step fetches the
// two adjacent children as one 64 B run; triangles are pre-gathered to 3 x float4 (a, b, c) in leaf order so
// that a leaf is one contiguous run and a test needs no index indirection.
#pragma once

#include "../../../include/zygpu.h"
#include "dmath.hpp"

namespace zd {

constexpr uint32_t kNull = 0xFFFFFFFFu;
constexpr int      kStackSize = 64;

struct DMesh {
    const float4*   nodes4;    // the wide re-layout (bvh4.hpp): 8 per BVH4 node
    const uint4*    nodes4q;   // the same with quantised child boxes (bvh4.hpp compress4): 4 per BVH4 node
    const float4*   nodes;     // 2 per node
    const float4*   tri_verts; // 3 per triangle (leaf order)
    const uint32_t* triangles; // 3 vertex indices per triangle (shading only)
    const uint16_t* triangle_parts;
    const float*    positions;
    const uint16_t* normals;
    const float*    uvs;
    const float*    part_areas;
    const uint32_t* primitive_mapping;  // triangle -> index within its part; set when a part is a light
    // 1: the relaxed units' per-lane any-hit walks of the reference's trees (the kernels that know every shape: MESHL / EXTRA / POOL)
    // walk this mesh through its quantised four-wide re-layout (kWideBlasMinTriangles triangles or more, leaves and depth inside the
    // walks' limits: zygpu_upload_scene), as the prop-table kernels do
    uint32_t        wide;
    // what meshFragment reads, gathered per triangle (leaf order, 3 per triangle: 48 B): (the corners' packed normals, part),
    // (uv a, uv b), (uv c, -, -); the corners' positions are tri_verts.  Through `triangles` into positions / normals / uvs a
    // fragment was a fourth level of dependent loads and ten cache lines of a large mesh.
    const uint4*    shade;
    // the eight-wide compressed tree of the persistent mesh walkers (bvh8.hpp; relaxed units, meshes of the mesh kernels): five uint4
    // per node; its triangles in its own order, 3 float4 each, the first corner's w = the triangle's index in the reference's order
    const uint4*    nodes8;
    const float4*   tris8;
};

// what a light-tree walk reads: the scene's tree over lights, or the primitive tree of one emissive mesh part
// (ZygpuMeshLight; light_tree.zig:520-720), whose leaves are triangles described by tri_props
struct DLightTree {
    const ZygpuLightNode* nodes;
    const uint32_t*       middles;
    const uint32_t*       orders;
    const uint32_t*       mapping;
    const float4*         tri_props;  // mesh part: 2 per triangle (centre.xyz, radius) (normal.xyz, relative power); scene: null
    const float*          infinite_cdf;  // discrete distribution over the infinite lights (num_infinite + 1 entries)
    float4                bounds_min, bounds_max;
    float                 infinite_weight, infinite_guard;
    uint32_t              max_split_depth, num_nodes, num_infinite, infinite_end;
    uint32_t              two_sided;  // mesh part only
};

struct DImage {  // ZygpuImage on the device
    const void* pixels;
    int32_t     width, height;
    uint32_t    format;
};

struct DMeshLight {
    DLightTree      tree;
    const uint32_t* triangle_mapping;
    uint32_t        mesh, part;
};

struct DImageLight {  // ZygpuImageLight on the device
    const float*    cdf;
    const uint32_t* lut;
    const uint4*    rows;  // cdf offset, size, lut offset, lut_range bits
    uint32_t        num_rows;
    float           total_weight;
    uint32_t        sampler;
};

struct DPortalLight {  // ZygpuPortalLight on the device
    const float* sat;
    const float* func;
    uint32_t     dim, light_link;
};

// A small top-level scene as a table in the __constant__ block itself (zygpu_upload_scene fills it when the solid props are
// at most kUniformProps Rectangles and TriangleMeshes): every lane of a wave tests the same prop at the same time, the
// prop's bounds and transformation arrive through the scalar cache in SGPRs (sceneIntersectUniform, sceneVisibilityUniform).
// 17 props through the per-lane tree walk: 461 Msamples/s against 773 with 16 in the table (round 2); round 3's end
// (tools/prop_curve.sh): from the table 949 / 836 / 613 / 465 / 393 / 321 at 16 / 32 / 64 / 96 / 112 / 128 props, through the tree
// 436 / 456 / 394 / 396 at 33 / 65 / 97 / 129: the two meet at 112
#ifndef ZYG_UNIFORM_PROPS
#define ZYG_UNIFORM_PROPS 112
#endif
constexpr uint32_t kUniformProps = ZYG_UNIFORM_PROPS;
struct TlasEntry {
    float4   bmin, bmax;        // world bounds (s.aabbs)
    float4   r0, r1, r2, pos;   // transformation (s.trafos)
    uint32_t shape, flags, prop, mesh;
    uint32_t small_first, small_count, pad0, pad1;  // a mesh of a few triangles: its items in DScene::small_tris (count 0: none)
    // a Rectangle: rows that take a world point to (u, v, height over the plane), |u|, |v| <= 1 inside (unitRectangleIntersect)
    float4   fu, fv, fn;
};
// The triangles of the scene's smallest meshes (at most kSmallMeshTriangles each, kSmallTriangles in all), in the constant
// block beside the prop tables: the relaxed-arithmetic units test all of them, every lane of the wave the same triangle at the
// same time with its vertices in SGPRs, instead of walking a BVH of three levels lane by lane (sceneIntersectUniform).
// (measured on configs[1] with its two boxes at 18 triangles each: 580 Msamples/s through the per-lane BVH2 walk, 754-759 from
// this table, 810-817 at 12 triangles with either limit; at 48 triangles the table's linear cost (516) is past what the wide BLAS
// gives (605 at 108 triangles): the limit is 32, DESIGN 4 "threshold edges")
#ifndef ZYG_EXPERIMENT_SKIP_SMALL
#define ZYG_EXPERIMENT_SKIP_SMALL 0  // 2: a timing experiment (wrong images): the meshes of the small-triangle table are skipped
#endif
#ifndef ZYG_SMALL_TRIANGLES
#define ZYG_SMALL_TRIANGLES 128
#endif
#ifndef ZYG_SMALL_MESH_TRIANGLES
#define ZYG_SMALL_MESH_TRIANGLES 32
#endif
constexpr uint32_t kSmallTriangles = ZYG_SMALL_TRIANGLES, kSmallMeshTriangles = ZYG_SMALL_MESH_TRIANGLES;
// An item of the table is four float4: 1 (default): the rows of the affine map that takes a triangle to the unit triangle
// (a -> 0, b -> x, c -> y, a + e1 x e2 -> z), worked out in f64 at upload: a test is a transformed origin and direction instead of
// the two cross products and four dot products of Moeller-Trumbore (smallItemIntersect; only the relaxed units read the table,
// whose contract is the RMS bound, not the reference's rounding); 0: a, b, c and the reference's test.  The fourth holds, as
// bits, x: the triangle's index in its mesh | the frame's weight (0: 1-u-v, 1: u, 2: v) that is the weight of its vertex 1 << 8 |
// that of its vertex 2 << 10 (the frame may start at any of its corners); y: the same for the triangle that completes a
// parallelogram with it (the unit square's other half; weights 0: u+v-1, 1: 1-v, 2: 1-u), or kNull (option pair_small).
#ifndef ZYG_SMALL_UNIT
#define ZYG_SMALL_UNIT 1
#endif
constexpr uint32_t kSmallItemWords = 4;

struct DScene {
    TlasEntry            solid_table[kUniformProps];  // cheap shapes first; num_solid_table = 0: not available
    TlasEntry            unocc_table[4];              // the unoccluding tree's single leaf, in leaf order; 0: not available
    float4               small_tris[kSmallItemWords * kSmallTriangles];  // items of one or two triangles, mesh (= leaf) order
    uint32_t             num_solid_table, num_unocc_table;
    uint32_t             first_large;  // solid_table[first_large ..]: the meshes of kMeshKernelMinTriangles triangles or more (they come last)
    // 1: the table holds every solid prop of the scene, Spheres, Disks and Cubes included, none of them animated, masked or of thin
    // Glass (no material evaluates visibility): the relaxed units' complete kernels (MESHL / POOL) walk it wave-wide in place of the
    // top-level tree (sceneIntersect / sceneVisibility / sceneVisibilityTr below)
    uint32_t             table_complete;
    const ZygpuProp*     props;
    // animated props (ZygpuSceneDesc.keyframes): 3 float4 per keyframe (position, scale, rotation quaternion); prop_frames[p] =
    // first keyframe or kNull; null pointers: no prop is animated
    const float4*        keyframes;
    const uint32_t*      prop_frames;
    float4               origin;
    uint32_t             num_interpolation_frames;
    const float4*        trafos; // 4 per prop: r0(+sx), r1(+sy), r2(+sz), position
    const float4*        aabbs;  // 2 per prop
    const uint32_t*      material_ids;
    const uint32_t*      light_ids;
    const DMesh*         meshes;
    const ZygpuMaterial* materials;
    const ZygpuLight*    lights;
    const float4*        light_aabbs; // 2 per light
    const float4*        light_cones;
    const float4*        solid_nodes;
    const uint32_t*      solid_indices;
    const float4*        unocc_nodes;
    const uint32_t*      unocc_indices;
    DLightTree           light_tree;
    const DMeshLight*    mesh_lights;
    const DImageLight*   image_lights;
    const DPortalLight*  portal_lights;
    const float4*        sun_radiance;  // the Sky entity's sun: radiance over its disc (ZygpuSceneDesc.sun_radiance)
    uint32_t             num_sun_radiance;
    const uint32_t*      infinite_props;
    uint32_t             num_infinite_props;
    const DImage*        images;
    const ZygpuProcedural* procedurals;  // what the ZYGPU_TEX_PROCEDURAL textures index (null: the scene has none); dprocedural.hpp
    const float*         srgb_to_float;
    const float*         lut_e_m;
    const float*         lut_e_m_avg;
    const float*         lut_e;
    const float*         lut_e_avg;
    const float*         lut_e_s;
    float                specular_threshold;
    uint32_t             num_solid_nodes, num_unocc_nodes;
    uint32_t             num_props, num_solid_indices, num_unocc_indices;
    uint32_t             has_catcher;  // a prop is a shadow catcher (ZYGPU_PROP_SHADOW_CATCHER): PathtracerMIS' bookkeeping for it
    float4               heavy_aabbs[16];  // world bounds (2 each) of the up to 8 largest mesh props: queue grouping hint
    uint32_t             num_heavy;
};

struct Counters {
    unsigned long long nodes, tris, instances;
};

// ---- 2-D image textures: fetch, address modes, nearest / stochastic-bilinear filter
// (reference src/core/texture/texture_sampler.zig:84-168, sampler_mode.zig:20-88, texture.zig:166-219)
ZD float texAddress(bool repeat, float x) { return repeat ? x - floorf(x) : clampf(x, 0.f, 1.f); }
ZD int   texCoord(bool repeat, int c, int end) {
    if (repeat) {
        const int m = c % end;  // @mod: the result has the sign of the divisor
        return m < 0 ? m + end : m;
    }
    return max(min(c, end - 1), 0);
}
ZD f3 imageFetch3(const DScene& s, const DImage& img, int x, int y) {
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE3_SRGB == img.format) {
        const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + 3 * i;
        const float    r = s.srgb_to_float[p[0]], g = s.srgb_to_float[p[1]], b = s.srgb_to_float[p[2]];
        // aces.zig:9-17 sRGBtoAP1
        return mk3(0.61309732f, 0.07019422f, 0.02061560f) * S3(r) + mk3(0.33952285f, 0.91635557f, 0.10956983f) * S3(g) +
               mk3(0.04737928f, 0.01345259f, 0.86981512f) * S3(b);
    }
    const float* p = static_cast<const float*>(img.pixels) + 3 * i;
    return mk3(p[0], p[1], p[2]);
}
ZD float imageFetch1(const DImage& img, int x, int y) {  // texture.zig:166-175; unorm8ToFloat (base/encoding.zig:10-12)
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE1_UNORM == img.format) return float(static_cast<const uint8_t*>(img.pixels)[i]) * (1.f / 255.f);
    return static_cast<const float*>(img.pixels)[i];
}
// the texel a lookup reads: the nearest one, or the one of the four of the bilinear footprint that r picks
ZD void texelPick(const DImage& img, uint32_t sampler, const float* scale, float u, float v, float r, int& x, int& y) {
    const bool  ru = 0 != (sampler & ZYGPU_TEX_REPEAT_U), rv = 0 != (sampler & ZYGPU_TEX_REPEAT_V);
    const float su = scale[0] * u, sv = scale[1] * v;
    const float dw = float(img.width), dh = float(img.height);
    if (!(sampler & ZYGPU_TEX_LINEAR)) {
        x = min(int(texAddress(ru, su) * dw), img.width - 1);
        y = min(int(texAddress(rv, sv) * dh), img.height - 1);
        return;
    }
    const float mx_ = texAddress(ru, su) * dw - 0.5f, my_ = texAddress(rv, sv) * dh - 0.5f;
    const float fx = floorf(mx_), fy = floorf(my_);
    const float wx = mx_ - fx, wy = my_ - fy;
    const float ox = 1.f - wx, oy = 1.f - wy;
    int         index = 0;
    float       threshold = ox * oy;
    index += r > threshold ? 1 : 0;
    threshold = __fmaf_rn(wx, oy, threshold);
    index += r > threshold ? 1 : 0;
    threshold = __fmaf_rn(ox, wy, threshold);
    index += r > threshold ? 1 : 0;
    x = texCoord(ru, int(fx) + (index & 1), img.width);
    y = texCoord(rv, int(fy) + ((index & 2) >> 1), img.height);
}
// ts.sample2D_3 for an image texture; r = rs.stochastic_r
ZD f3 sampleImageMap(const DScene& s, uint32_t image, uint32_t sampler, const float* scale, float u, float v, float r) {
    const DImage img = s.images[image];
    int          x, y;
    texelPick(img, sampler, scale, u, v, r, x, y);
    return imageFetch3(s, img, x, y);
}
// ts.sample2D_2 for an image texture (texture.zig:177-186): Byte2 as unorm or, for usage Normal, snorm (base/encoding.zig:10-20)
ZD void sampleImageMap2(const DScene& s, uint32_t image, uint32_t sampler, const float* scale, float u, float v, float r, float& o0, float& o1) {
    const DImage img = s.images[image];
    int          x, y;
    texelPick(img, sampler, scale, u, v, r, x, y);
    const size_t i = size_t(y) * size_t(img.width) + size_t(x);
    if (ZYGPU_IMAGE_BYTE2 == img.format) {
        const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + 2 * i;
        const bool     snorm = 0 != (sampler & ZYGPU_TEX_SNORM);
        o0 = snorm ? __fmaf_rn(float(p[0]), 1.f / 128.f, -1.f) : float(p[0]) * (1.f / 255.f);
        o1 = snorm ? __fmaf_rn(float(p[1]), 1.f / 128.f, -1.f) : float(p[1]) * (1.f / 255.f);
    } else {
        const float* p = static_cast<const float*>(img.pixels) + 2 * i;
        o0 = p[0];
        o1 = p[1];
    }
}
// ts.sampleImage2D_1 (texture_sampler.zig:15-20)
ZD float sampleImageMap1(const DScene& s, uint32_t image, uint32_t sampler, const float* scale, float u, float v, float r) {
    const DImage img = s.images[image];
    int          x, y;
    texelPick(img, sampler, scale, u, v, r, x, y);
    return imageFetch1(img, x, y);
}

// Opacity masks draw from the path's sampler while the ray is traversed; `Draw` is that sampler's sample1D. NoDraw stands
// in where no material of the scene has a mask (the calls are compiled out with EXTRA = false).
struct NoDraw {
    ZD float operator()() const { return 0.f; }
};
// Base.stochasticOpacity (material_base.zig:44-55): does the closest-hit traversal accept this hit?
template <typename Draw>
ZD bool stochasticOpacity(const DScene& s, const ZygpuMaterial& m, float u, float v, Draw& draw) {
    if (kNull == m.mask_image) return true;
    const float o = sampleImageMap1(s, m.mask_image, m.mask_sampler, m.mask_scale, u, v, draw());
    if (0.f == o || (o < 1.f && o <= draw())) return false;
    return true;
}
// Base.opacity (material_base.zig:35-41)
template <typename Draw>
ZD float maskOpacity(const DScene& s, const ZygpuMaterial& m, float u, float v, Draw& draw) {
    if (kNull == m.mask_image) return 1.f;
    return sampleImageMap1(s, m.mask_image, m.mask_sampler, m.mask_scale, u, v, draw());
}

struct Hit {
    float    t, u, v;
    uint32_t primitive, prop;
};

ZD Trafo loadTrafo(const DScene& s, uint32_t p) {
    const float4 a = s.trafos[4 * p + 0], b = s.trafos[4 * p + 1], c = s.trafos[4 * p + 2], d = s.trafos[4 * p + 3];
    Trafo        t;
    t.r0 = {a.x, a.y, a.z};
    t.r1 = {b.x, b.y, b.z};
    t.r2 = {c.x, c.y, c.z};
    t.scale = {a.w, b.w, c.w};
    t.position = {d.x, d.y, d.z};
    return t;
}

// Two keyframes interpolated: math.Transformation.lerp (transformation.zig:36-44) with quaternion.slerp (quaternion.zig:217-245),
// the scene origin subtracted after the interpolation (space.zig:121-124), ComposedTransformation.init with
// quaternion.toMat3x3 (composed_transformation.zig:17-29, quaternion.zig:72-116)
ZD Trafo interpolateKeyframes(float4 pa, float4 sa, float4 qa, float4 pb, float4 sb, float4 qb, float wgt, f3 origin) {
    const f3 t3 = S3(wgt);
    Trafo    out;
    out.position = lerp3(mk3(pa.x, pa.y, pa.z), mk3(pb.x, pb.y, pb.z), t3) - origin;
    out.scale = lerp3(mk3(sa.x, sa.y, sa.z), mk3(sb.x, sb.y, sb.z), t3);
    const float a[4] = {qa.x, qa.y, qa.z, qa.w};
    float       end[4] = {qb.x, qb.y, qb.z, qb.w};
    float       cosom = a[0] * end[0] + a[1] * end[1] + a[2] * end[2] + a[3] * end[3];
    if (cosom < 0.f) {
        cosom = -cosom;
        for (int k = 0; k < 4; ++k) end[k] = -end[k];
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
    float q[4];
    for (int k = 0; k < 4; ++k) q[k] = sclp * a[k] + sclq * end[k];
    const float tq[4] = {q[0] + q[0], q[1] + q[1], q[2] + q[2], q[3] + q[3]};
    const float xy = tq[1] * q[0], xz = tq[2] * q[0], yz = tq[1] * q[2];
    const float w0 = tq[0] * q[3], w1 = tq[1] * q[3], w2 = tq[2] * q[3];
    const float t0 = (q[3] + q[1]) * (q[3] - q[1]), t1 = (q[0] + q[2]) * (q[0] - q[2]);
    const float t2 = (q[3] + q[0]) * (q[3] - q[0]), t3q = (q[1] + q[2]) * (q[1] - q[2]);
    out.r0 = mk3(t0 + t1, xy - w2, xz + w1);
    out.r1 = mk3(xy + w2, t2 + t3q, yz - w0);
    out.r2 = mk3(xz - w1, yz + w0, t2 - t3q);
    return out;
}

// Scene.propTransformationAtMaybeStatic (scene.zig:382-385 -> space.zig:99-137); ANIM: the kernel instantiations that know
// animated props (the EXTRA / MESHL ones: a scene with an animated prop runs them)
// The animated branch is a function of its own, not inlined: the slerp (acos, three sines, a 3 x 3 from the quaternion) inside every
// traversal loop cost the kernels that know animated props their registers whether or not a scene has one (measured on the scenes
// without: outdoor 238 -> 247 Msamples/s, cornell_glass 115 -> 124, wf_intersect<POOL> -24 %, with the branch compiled out).
static __device__ __attribute__((noinline)) void propTrafoAnimated(const float4* keyframes, uint32_t first, uint32_t num_interpolation_frames,
                                                                   float4 origin, uint32_t time, Trafo* out) {
    const uint32_t tick = uint32_t(705600000ull / 60);  // space.zig:18-19
    uint32_t       i = time / tick;
    float          wgt = float(double(time - i * tick) / double(tick));
    if (i + 1 >= num_interpolation_frames) {  // a shutter sample of exactly 1.0: the last keyframe
        i = num_interpolation_frames - 2;
        wgt = 1.f;
    }
    const float4* k = keyframes + size_t(first + i) * 3;
    *out = interpolateKeyframes(k[0], k[1], k[2], k[3], k[4], k[5], wgt, mk3(origin.x, origin.y, origin.z));
}

// CALL: reach the animated branch through the function above (the closest-hit walks and wf_shade); the shadow-ray walks and
// wf_scatter keep it inline (a call inside wf_scatter cost its POOL instantiation 7 %)
template <bool ANIM, bool CALL = false>
ZD Trafo propTrafoAt(const DScene& s, uint32_t p, uint32_t time) {
    if (ANIM && s.prop_frames) {
        const uint32_t f = s.prop_frames[p];
        if (kNull != f) {
            Trafo t;
#ifndef ZYG_ANIM_INLINE
#define ZYG_ANIM_INLINE 0
#endif
            if (CALL && !ZYG_ANIM_INLINE) {
                propTrafoAnimated(s.keyframes, f, s.num_interpolation_frames, s.origin, time, &t);
            } else {
                const uint32_t tick = uint32_t(705600000ull / 60);  // space.zig:18-19
                uint32_t       i = time / tick;
                float          wgt = float(double(time - i * tick) / double(tick));
                if (i + 1 >= s.num_interpolation_frames) {  // a shutter sample of exactly 1.0: the last keyframe
                    i = s.num_interpolation_frames - 2;
                    wgt = 1.f;
                }
                const float4* k = s.keyframes + size_t(f + i) * 3;
                t = interpolateKeyframes(k[0], k[1], k[2], k[3], k[4], k[5], wgt, mk3(s.origin.x, s.origin.y, s.origin.z));
            }
            return t;
        }
    }
    return loadTrafo(s, p);
}

// node.zig:73-87; n0 = (min.xyz, children_or_first), n1 = (max.xyz, count)
ZD float nodeIntersect(float4 n0, float4 n1, const Ray& ray) {
    const f3    lower = (mk3(n0.x, n0.y, n0.z) - ray.origin) * ray.inv_direction;
    const f3    upper = (mk3(n1.x, n1.y, n1.z) - ray.origin) * ray.inv_direction;
    const f3    t0 = min3v(lower, upper), t1 = max3v(lower, upper);
    const float tboxmin = mx(t0.x, mx(t0.y, mx(t0.z, ray.min_t)));
    const float tboxmax = mn(t1.x, mn(t1.y, mn(t1.z, ray.max_t)));
    return tboxmin <= tboxmax ? tboxmin : FLT_MAX;
}

ZD bool aabbIntersect(float4 bmin, float4 bmax, const Ray& ray) {
    return FLT_MAX != nodeIntersect(bmin, bmax, ray);  // same arithmetic, boolean result (aabb.zig:46-60)
}

// AABB.intersectP (base/math/aabb.zig:62-83): the entry distance, or the exit distance when the origin is inside
ZD float aabbIntersectP(float4 bmin, float4 bmax, const Ray& ray) {
    const f3    lower = (mk3(bmin.x, bmin.y, bmin.z) - ray.origin) * ray.inv_direction;
    const f3    upper = (mk3(bmax.x, bmax.y, bmax.z) - ray.origin) * ray.inv_direction;
    const f3    t0 = min3v(lower, upper), t1 = max3v(lower, upper);
    const float imin = mx(mx(t0.x, t0.y), t0.z);
    const float imax = mn(mn(t1.x, t1.y), t1.z);
    const float tboxmin = mx(imin, ray.min_t);
    const float tboxmax = mn(imax, ray.max_t);
    if (tboxmin <= tboxmax) return imin < ray.min_t ? imax : imin;
    return FLT_MAX;
}

// triangle.zig:26-52
ZD bool triangleIntersect(const Ray& ray, f3 a, f3 b, f3 c, float& ht, float& hu, float& hv) {
    const f3    e1 = b - a, e2 = c - a;
    const f3    tvec = ray.origin - a;
    const f3    pvec = cross3(ray.direction, e2);
    const f3    qvec = cross3(tvec, e1);
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

ZD f3 xyz(float4 v) { return {v.x, v.y, v.z}; }

// An item of DScene::small_tris against a ray in the mesh's space: rows r0, r1, r2 take a point to the first triangle's own frame
// (u, v, height over its plane).  The acceptance rule of triangleIntersect, for the unit square's other half too when the item
// holds a second triangle; ht / hu / hv / primitive as the reference reports them for the triangle that is hit.
struct SmallItem {
    float4 r0, r1, r2, names;
};
ZD float smallReciprocal(float x) {
#if ZYG_FAST
    return __builtin_amdgcn_rcpf(x);  // one instruction, 1 ulp; the divide's range scaling is of no use to a test that compares
#else
    return 1.f / x;
#endif
}
ZD bool smallItemIntersect(const Ray& ray, const SmallItem& item, float& ht, float& hu, float& hv, uint32_t& primitive) {
    const float4   r0 = item.r0, r1 = item.r1, r2 = item.r2;
    const uint32_t first = __float_as_uint(item.names.x), second = __float_as_uint(item.names.y);
#if ZYG_SMALL_UNIT
    const float oz = fmaf(r2.x, ray.origin.x, fmaf(r2.y, ray.origin.y, fmaf(r2.z, ray.origin.z, r2.w)));
    const float dz = fmaf(r2.x, ray.direction.x, fmaf(r2.y, ray.direction.y, r2.z * ray.direction.z));
    const float hit_t = -oz * smallReciprocal(dz);
    const float ox = fmaf(r0.x, ray.origin.x, fmaf(r0.y, ray.origin.y, fmaf(r0.z, ray.origin.z, r0.w)));
    const float dx = fmaf(r0.x, ray.direction.x, fmaf(r0.y, ray.direction.y, r0.z * ray.direction.z));
    const float oy = fmaf(r1.x, ray.origin.x, fmaf(r1.y, ray.origin.y, fmaf(r1.z, ray.origin.z, r1.w)));
    const float dy = fmaf(r1.x, ray.direction.x, fmaf(r1.y, ray.direction.y, r1.z * ray.direction.z));
    const float u = fmaf(hit_t, dx, ox), v = fmaf(hit_t, dy, oy);
    const float uv = u + v;
    if (!(u >= 0.f && v >= 0.f && hit_t >= ray.min_t && ray.max_t >= hit_t)) return false;
    const bool lower = 1.f >= uv;
    if (!lower && (kNull == second || !(1.f >= u && 1.f >= v))) return false;
    const uint32_t which = lower ? first : second;
    const float    w0 = lower ? 1.f - uv : uv - 1.f, w1 = lower ? u : 1.f - v, w2 = lower ? v : 1.f - u;
    const uint32_t su = (which >> 8) & 3u, sv = (which >> 10) & 3u;
    ht = hit_t;
    hu = 0 == su ? w0 : 1 == su ? w1 : w2;
    hv = 0 == sv ? w0 : 1 == sv ? w1 : w2;
    primitive = which & 0xffu;
    return true;
#else
    primitive = first & 0xffu;
    return triangleIntersect(ray, xyz(r0), xyz(r1), xyz(r2), ht, hu, hv);
#endif
}

// the same for a shadow ray: whether the item blocks it
ZD bool smallItemBlocks(const Ray& ray, const SmallItem& item) {
#if ZYG_SMALL_UNIT
    const float4   r0 = item.r0, r1 = item.r1, r2 = item.r2;
    const uint32_t second = __float_as_uint(item.names.y);
    const float oz = fmaf(r2.x, ray.origin.x, fmaf(r2.y, ray.origin.y, fmaf(r2.z, ray.origin.z, r2.w)));
    const float dz = fmaf(r2.x, ray.direction.x, fmaf(r2.y, ray.direction.y, r2.z * ray.direction.z));
    const float hit_t = -oz * smallReciprocal(dz);
    const float ox = fmaf(r0.x, ray.origin.x, fmaf(r0.y, ray.origin.y, fmaf(r0.z, ray.origin.z, r0.w)));
    const float dx = fmaf(r0.x, ray.direction.x, fmaf(r0.y, ray.direction.y, r0.z * ray.direction.z));
    const float oy = fmaf(r1.x, ray.origin.x, fmaf(r1.y, ray.origin.y, fmaf(r1.z, ray.origin.z, r1.w)));
    const float dy = fmaf(r1.x, ray.direction.x, fmaf(r1.y, ray.direction.y, r1.z * ray.direction.z));
    const float u = fmaf(hit_t, dx, ox), v = fmaf(hit_t, dy, oy);
    return u >= 0.f && v >= 0.f && hit_t >= ray.min_t && ray.max_t >= hit_t && (1.f >= u + v || (kNull != second && 1.f >= u && 1.f >= v));
#else
    float    t, u, v;
    uint32_t primitive;
    return smallItemIntersect(ray, item, t, u, v, primitive);
#endif
}

// The items of one small mesh, in order, handed to test(item).  The table is read through the scalar cache (every lane the same
// item): the next item's words are asked for before the current one is tested, so that a test does not start with a load's latency.
template <typename Test>
ZD void forSmallItems(const DScene& s, uint32_t first, uint32_t count, Test&& test) {
    const float4* words = s.small_tris + kSmallItemWords * first;
    SmallItem     item{words[0], words[1], words[2], words[3]};
    for (uint32_t k = 0; k < count; ++k) {
        const float4*   ahead = words + kSmallItemWords * (k + 1 < count ? k + 1 : k);
        const SmallItem next{ahead[0], ahead[1], ahead[2], ahead[3]};
        test(item);
        item = next;
    }
}

// sphere.zig:28-60 (nearer root first); unit-diameter sphere scaled by trafo.scale.x
ZD bool sphereIntersect(const Ray& ray, const Trafo& trafo, float& out_t) {
    const float idl = 1.f / length3(ray.direction);
    const f3    nd = ray.direction * S3(idl);
    const f3    v = trafo.position - ray.origin;
    const float b = dot3(nd, v);
    const f3    remedy_term = v - S3(b) * nd;
    const float radius = 0.5f * trafo.scale.x;
    const float discriminant = radius * radius - dot3(remedy_term, remedy_term);
    if (discriminant > 0.f) {
        const float dist = sqrtf(discriminant);
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

// Shared BVH2 walk: nearer child first, farther pushed (triangle_tree.zig:76-94 / prop_tree.zig:86-104).
// `leaf(first, count)` returns true to stop the walk (any-hit).
template <bool STATS, typename Leaf>
ZD bool bvhWalk(const float4* nodes, uint32_t num_nodes, Ray& ray, Counters& cnt, Leaf&& leaf) {
    if (0 == num_nodes) return false;
    uint32_t stack[kStackSize];
    int      sp = 0;
    if (STATS) cnt.nodes += 1;
    uint32_t data = __float_as_uint(nodes[0].w), count = __float_as_uint(nodes[1].w);
    for (;;) {
        // "while-while": every lane first descends to its next leaf, then the leaves (the expensive part: prop tests with
        // nested walks, triangle tests) are processed with the wave converged again
        while (0 == count) {
            const uint32_t a = data;
            const float4   a0 = nodes[2 * a], a1 = nodes[2 * a + 1], b0 = nodes[2 * a + 2], b1 = nodes[2 * a + 3];
            if (STATS) cnt.nodes += 2;
            const float dista = nodeIntersect(a0, a1, ray);
            const float distb = nodeIntersect(b0, b1, ray);
            const bool  swap = dista > distb;
            const float dnear = swap ? distb : dista, dfar = swap ? dista : distb;
            if (FLT_MAX == dnear) {
                if (0 == sp) return false;
                const uint32_t n = stack[--sp];
                data = __float_as_uint(nodes[2 * n].w);
                count = __float_as_uint(nodes[2 * n + 1].w);
            } else {
                data = __float_as_uint(swap ? b0.w : a0.w);
                count = __float_as_uint(swap ? b1.w : a1.w);
                if (FLT_MAX != dfar && sp < kStackSize) stack[sp++] = swap ? a : a + 1;
            }
        }
        if (leaf(data, count)) return true;
        if (0 == sp) return false;
        const uint32_t n = stack[--sp];
        data = __float_as_uint(nodes[2 * n].w);
        count = __float_as_uint(nodes[2 * n + 1].w);
    }
}

// The walk of a mesh's wide re-layout (bvh4.hpp): one 128 B node = four child slots per step, the children a ray enters
// visited nearest first, the rest parked with their entry distances (skipped on return if the ray has been shortened past
// them).  Same leaves, same triangle order inside a leaf as the BVH2; `leaf` as in bvhWalk.
constexpr int kStack4 = 48;
template <typename Leaf>
ZD bool bvh4Walk(const float4* nodes, Ray& ray, Leaf&& leaf) {
    uint32_t stack_ref[kStack4], stack_count[kStack4];
    float    stack_dist[kStack4];
    int      sp = 0;
    uint32_t cur = 0, cur_count = 0;  // the root node
    // the next parked child the (possibly shortened) ray still reaches
    auto pop = [&]() {
        while (sp > 0) {
            --sp;
            if (stack_dist[sp] <= ray.max_t) {
                cur = stack_ref[sp];
                cur_count = stack_count[sp];
                return true;
            }
        }
        return false;
    };
    for (;;) {
        while (0 == cur_count) {
            const float4* n = nodes + size_t(cur) * 8;
            const float4  a0 = n[0], a1 = n[1], b0 = n[2], b1 = n[3], c0 = n[4], c1 = n[5], d0 = n[6], d1 = n[7];
            float    t0 = nodeIntersect(a0, a1, ray), t1 = nodeIntersect(b0, b1, ray);
            float    t2 = nodeIntersect(c0, c1, ray), t3 = nodeIntersect(d0, d1, ray);
            uint32_t r0 = __float_as_uint(a0.w), r1 = __float_as_uint(b0.w), r2 = __float_as_uint(c0.w), r3 = __float_as_uint(d0.w);
            uint32_t n0 = __float_as_uint(a1.w), n1 = __float_as_uint(b1.w), n2 = __float_as_uint(c1.w), n3 = __float_as_uint(d1.w);
            // sorting network (0,1)(2,3)(0,2)(1,3)(1,2): ascending entry distance, misses (FLT_MAX) last
#define ZYG_CSWAP(i, j)                        \
    if (t##j < t##i) {                         \
        const float    tt = t##i;              \
        const uint32_t rr = r##i, nn = n##i;   \
        t##i = t##j, r##i = r##j, n##i = n##j; \
        t##j = tt, r##j = rr, n##j = nn;       \
    }
            ZYG_CSWAP(0, 1)
            ZYG_CSWAP(2, 3)
            ZYG_CSWAP(0, 2)
            ZYG_CSWAP(1, 3)
            ZYG_CSWAP(1, 2)
#undef ZYG_CSWAP
            if (FLT_MAX != t3 && sp < kStack4) stack_ref[sp] = r3, stack_count[sp] = n3, stack_dist[sp] = t3, ++sp;
            if (FLT_MAX != t2 && sp < kStack4) stack_ref[sp] = r2, stack_count[sp] = n2, stack_dist[sp] = t2, ++sp;
            if (FLT_MAX != t1 && sp < kStack4) stack_ref[sp] = r1, stack_count[sp] = n1, stack_dist[sp] = t1, ++sp;
            if (FLT_MAX != t0) {
                cur = r0;
                cur_count = n0;
            } else if (!pop()) {  // the node's slots are all missed: back to what was parked before it
                return false;
            }
        }
        if (leaf(cur, cur_count)) return true;
        if (!pop()) return false;
    }
}

// A triangle's corners as the walks' leaves read them (global_load: see Node4q::load)
ZD void triangleVertices(const float4* tri_verts, uint32_t i, f3& a, f3& b, f3& c) {
    typedef float                                            Corner __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) Corner* v = (const __attribute__((address_space(1))) Corner*)(tri_verts + size_t(3) * i);
    const Corner                                             va = v[0], vb = v[1], vc = v[2];
    a = mk3(va.x, va.y, va.z);
    b = mk3(vb.x, vb.y, vb.z);
    c = mk3(vc.x, vc.y, vc.z);
}
ZD void triangleVertices(const DMesh& mesh, uint32_t i, f3& a, f3& b, f3& c) { triangleVertices(mesh.tri_verts, i, a, b, c); }

// The four-wide nodes in their quantised form (bvh4.hpp compress4): one node = 64 B, half the bytes a walk pulls through the caches
// (configs[2]: 484 -> 491.5 Msamples/s).  -DZYG_BVH4Q=0 walks the full-precision nodes.
#ifndef ZYG_BVH4Q
#define ZYG_BVH4Q 1
#endif
// The grid form computes q * (scale * inv_dir) + (corner - origin) * inv_dir, which is 0 * inf or inf - inf = NaN for a ray with a zero
// direction component: harmless only where mn / mx drop a NaN (the hardware v_min / v_max of the relaxed units, dmath.hpp
// ZYG_FAST_MINMAX); with the x86 forms (the precise units' test hook, -DZYG_FAST_MINMAX=0) the boxes are dequantised instead.
#ifndef ZYG_NODE4Q_GRID
#define ZYG_NODE4Q_GRID ZYG_FAST_MINMAX  // 0: the child boxes are dequantised and tested as nodeIntersect tests a full-precision box
#endif
static_assert(!ZYG_NODE4Q_GRID || ZYG_FAST_MINMAX, "the quantised-grid node test needs min / max that drop NaNs");
struct Node4q {
    uint4 a, b, c, d;
    ZD static Node4q load(const uint4* nodes, uint32_t i) {
        // global_load instead of the flat_load a pointer out of the scene block gets (one counter to wait for, no LDS aperture check)
        typedef uint32_t                                          Words __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(1))) Words* n = (const __attribute__((address_space(1))) Words*)(nodes + size_t(i) * 4);
        const Words                                               a = n[0], b = n[1], c = n[2], d = n[3];
        return {make_uint4(a.x, a.y, a.z, a.w), make_uint4(b.x, b.y, b.z, b.w), make_uint4(c.x, c.y, c.z, c.w), make_uint4(d.x, d.y, d.z, d.w)};
    }
    // The ray in the node's quantisation grid: it crosses grid plane q of axis i at distance q * s[i] + c[i], so a child's six
    // distances are six fmas of its bytes (the boxes are never dequantised: 4 x 9 instructions fewer per node).
    struct Slabs {
        f3 s, c;
    };
    ZD Slabs slabs(const Ray& ray) const {
        const f3 o = mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        const f3 sc = mk3(__uint_as_float((a.w & 255u) << 23), __uint_as_float(((a.w >> 8) & 255u) << 23), __uint_as_float(((a.w >> 16) & 255u) << 23));
#if ZYG_NODE4Q_GRID
        return {sc * ray.inv_direction, (o - ray.origin) * ray.inv_direction};
#else
        return {sc, o};
#endif
    }
    // entry distance of child K, FLT_MAX for a miss or an empty slot; `packed` = the child's reference | its triangle count << 24
    // (a WideStack entry)
    template <int K>
    ZD float test(const Slabs& g, const Ray& ray, uint32_t& packed) const {
        const uint32_t refs[4] = {c.z, c.w, d.x, d.y};
        const uint32_t ref = refs[K];
        packed = __builtin_amdgcn_perm(d.z, ref, 0x00020100u | (uint32_t(4 + K) << 24));  // v_perm_b32: bytes 0-2 of ref, byte K of d.z on top
        const f3 qlo = mk3(float((b.x >> (8 * K)) & 255u), float((b.y >> (8 * K)) & 255u), float((b.z >> (8 * K)) & 255u));
        const f3 qhi = mk3(float((b.w >> (8 * K)) & 255u), float((c.x >> (8 * K)) & 255u), float((c.y >> (8 * K)) & 255u));
#if ZYG_NODE4Q_GRID
        const f3    lower = fma3(qlo, g.s, g.c), upper = fma3(qhi, g.s, g.c);
        const f3    t0 = min3v(lower, upper), t1 = max3v(lower, upper);
        const float tboxmin = mx(mx(t0.x, t0.y), mx(t0.z, ray.min_t));
        const float tboxmax = mn(mn(t1.x, t1.y), mn(t1.z, ray.max_t));
        return ((0xFFFFFFFFu != ref) & (tboxmin <= tboxmax)) ? tboxmin : FLT_MAX;
#else
        if (0xFFFFFFFFu == ref) return FLT_MAX;
        const f3 lo = fma3(qlo, g.s, g.c), hi = fma3(qhi, g.s, g.c);
        return nodeIntersect(make_float4(lo.x, lo.y, lo.z, 0.f), make_float4(hi.x, hi.y, hi.z, 0.f), ray);
#endif
    }
};

// The parked children of the quantised walks: the first kLdsStack entries of a lane live in LDS (a push is one ds_write instead
// of three scratch stores through the vector memory path the node loads use), deeper ones in scratch.  An entry packs the child's
// reference with its triangle count (reference < 2^24: zygpu_upload_scene keeps larger meshes on the BVH2 kernels).
// -DZYG_LDS_STACK=0: everything in scratch (the persistent kernels' unit: their LDS holds the path states).
#ifndef ZYG_LDS_STACK
#define ZYG_LDS_STACK 1
#endif
#ifndef ZYG_ANY_NEAREST
#define ZYG_ANY_NEAREST 0  // 1: the any-hit walk of the quantised nodes descends into the nearest child first (measured: see DESIGN)
#endif
#ifndef ZYG_LDS_STACK_ENTRIES
#define ZYG_LDS_STACK_ENTRIES 16  // 32 KB per 256-thread block: four blocks of wf_trace_shade per CU
#endif
constexpr int kLdsStack = ZYG_LDS_STACK ? ZYG_LDS_STACK_ENTRIES : 0;
#ifndef ZYG_LDS_STACK_THREADS
#define ZYG_LDS_STACK_THREADS 256
#endif
constexpr int kLdsStackThreads = ZYG_LDS_STACK_THREADS;  // threads of the largest block that walks

// one array per entry layout: a kernel that runs both walks (wf_tick) keeps them apart, one that runs only the any-hit walk
// (wf_scatter, wf_shadow_u) allocates half
ZD uint2* wideStackLds() {
    __shared__ uint2 entries[(kLdsStack ? kLdsStack : 1) * kLdsStackThreads];
    return entries;
}
ZD uint32_t* wideStackLdsAny() {
    __shared__ uint32_t entries[(kLdsStack ? kLdsStack : 1) * kLdsStackThreads];
    return entries;
}

// The entries past the LDS part.  A local of its own in the caller: an aggregate that is indexed by a variable is never split into
// registers, and as members of WideStack these arrays kept its stack pointer and its LDS address in scratch with them (a scratch
// load and a wait in front of every push and pop, the LDS entries read through flat loads).
template <bool DIST>
struct WideSpill {
    uint32_t ref[kStack4 - kLdsStack];
    float    dist[DIST ? kStack4 - kLdsStack : 1];
};

// The LDS part through pointers that carry their address space: with generic ones the compiler merges a pop's two sources (LDS entry
// or spilled entry) into one flat load of a selected address.
typedef uint32_t                                   WordPair __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) WordPair LdsUint2;
typedef __attribute__((address_space(3))) uint32_t LdsUint;

template <bool DIST>
struct WideStack {
    LdsUint2* lds2;  // DIST: (reference | count, distance) per entry
    LdsUint*  lds1;  // otherwise one word, the lanes' words side by side
    uint32_t* spill_ref;
    float*    spill_dist;
    int       sp = 0;
    ZD explicit WideStack(WideSpill<DIST>& spill)
        : lds2(DIST ? (LdsUint2*)(wideStackLds() + threadIdx.x) : nullptr), lds1(DIST ? nullptr : (LdsUint*)(wideStackLdsAny() + threadIdx.x)),
          spill_ref(spill.ref), spill_dist(spill.dist) {}
    ZD void push(uint32_t packed, float dist) {  // packed = reference | triangle count << 24 (Node4q::test)
        if (sp >= kStack4) return;
        if (sp < kLdsStack) {
            if (DIST) lds2[sp * kLdsStackThreads] = WordPair{packed, __float_as_uint(dist)};
            else lds1[sp * kLdsStackThreads] = packed;
        } else {
            spill_ref[sp - kLdsStack] = packed;
            if (DIST) spill_dist[sp - kLdsStack] = dist;
        }
        ++sp;
    }
    // the top entry (sp > 0)
    ZD void pop(uint32_t& ref, uint32_t& count, float& dist) {
        --sp;
        uint32_t packed;
        if (sp < kLdsStack) {
            if (DIST) {
                const WordPair e = lds2[sp * kLdsStackThreads];
                packed = e.x;
                dist = __uint_as_float(e.y);
            } else {
                packed = lds1[sp * kLdsStackThreads];
            }
        } else {
            packed = spill_ref[sp - kLdsStack];
            if (DIST) dist = spill_dist[sp - kLdsStack];
        }
        ref = packed & 0xFFFFFFu;
        count = packed >> 24;
    }
};

template <typename Leaf>
ZD bool bvh4qWalk(const uint4* nodes, Ray& ray, Leaf&& leaf) {
    WideSpill<true> spill;
    WideStack<true> stack(spill);
    uint32_t        cur = 0, cur_count = 0;  // the root node
    auto pop = [&]() {
        while (stack.sp > 0) {
            uint32_t r, n;
            float    dist;
            stack.pop(r, n, dist);
            if (dist <= ray.max_t) {
                cur = r;
                cur_count = n;
                return true;
            }
        }
        return false;
    };
    for (;;) {
        while (0 == cur_count) {
            const Node4q n = Node4q::load(nodes, cur);
            const Node4q::Slabs g = n.slabs(ray);
            uint32_t     p0, p1, p2, p3;
            float        t0 = n.test<0>(g, ray, p0), t1 = n.test<1>(g, ray, p1), t2 = n.test<2>(g, ray, p2), t3 = n.test<3>(g, ray, p3);
#define ZYG_CSWAP(i, j)           \
    if (t##j < t##i) {            \
        const float    tt = t##i; \
        const uint32_t pp = p##i; \
        t##i = t##j, p##i = p##j; \
        t##j = tt, p##j = pp;     \
    }
            ZYG_CSWAP(0, 1)
            ZYG_CSWAP(2, 3)
            ZYG_CSWAP(0, 2)
            ZYG_CSWAP(1, 3)
            ZYG_CSWAP(1, 2)
#undef ZYG_CSWAP
            if (FLT_MAX != t3) stack.push(p3, t3);
            if (FLT_MAX != t2) stack.push(p2, t2);
            if (FLT_MAX != t1) stack.push(p1, t1);
            if (FLT_MAX != t0) {
                cur = p0 & 0xFFFFFFu;
                cur_count = p0 >> 24;
            } else if (!pop()) {
                return false;
            }
        }
        if (leaf(cur, cur_count)) return true;
        if (!pop()) return false;
    }
}

template <typename Leaf>
ZD bool bvh4qWalkAny(const uint4* nodes, const Ray& ray, Leaf&& leaf) {
    WideSpill<false> spill;
    WideStack<false> stack(spill);
    uint32_t         cur = 0, cur_count = 0;
    float            unused = 0.f;
    for (;;) {
        while (0 == cur_count) {
            const Node4q n = Node4q::load(nodes, cur);
            const Node4q::Slabs g = n.slabs(ray);
            uint32_t     p0, p1, p2, p3;
#if ZYG_ANY_NEAREST
            // the nearest child first, the others parked in storage order
            const float  t0 = n.test<0>(g, ray, p0), t1 = n.test<1>(g, ray, p1), t2 = n.test<2>(g, ray, p2), t3 = n.test<3>(g, ray, p3);
            const float  tmin = mn(mn(t0, t1), mn(t2, t3));
            bool         have = FLT_MAX != tmin;
            uint32_t     nr = 0, nc = 0;
            bool         chosen = false;
            auto         take = [&](float t, uint32_t p) {
                if (FLT_MAX == t) return;
                if (!chosen && t == tmin) {
                    chosen = true;
                    nr = p & 0xFFFFFFu;
                    nc = p >> 24;
                } else {
                    stack.push(p, 0.f);
                }
            };
            take(t3, p3);
            take(t2, p2);
            take(t1, p1);
            take(t0, p0);
#else
            const bool   h0 = FLT_MAX != n.test<0>(g, ray, p0), h1 = FLT_MAX != n.test<1>(g, ray, p1);
            const bool   h2 = FLT_MAX != n.test<2>(g, ray, p2), h3 = FLT_MAX != n.test<3>(g, ray, p3);
            bool         have = false;
            uint32_t     np = 0;
            auto         take = [&](bool h, uint32_t p) {
                if (!h) return;
                if (have) stack.push(np, 0.f);
                np = p;
                have = true;
            };
            take(h3, p3);
            take(h2, p2);
            take(h1, p1);
            take(h0, p0);
#endif
#if ZYG_ANY_NEAREST
            const uint32_t np = nr | (nc << 24);
#endif
            if (have) {
                cur = np & 0xFFFFFFu;
                cur_count = np >> 24;
            } else {
                if (0 == stack.sp) return false;
                stack.pop(cur, cur_count, unused);
            }
        }
        if (leaf(cur, cur_count)) return true;
        if (0 == stack.sp) return false;
        stack.pop(cur, cur_count, unused);
    }
}

// The walks of the quantised nodes one step at a time, for the kernels whose lanes fetch a new ray when theirs is done
// (wavefront.hpp wf_shadow_u, wf_mesh_closest): a step tests one node's four children, or one leaf's triangles.
enum : int { kWalkRunning = 0, kWalkHit = 1, kWalkDone = 2 };

struct AnyWalk4q {  // = bvh4qWalkAny + meshIntersectP's leaf
    WideStack<false> stack;
    uint32_t         cur = 0, cur_count = 0;
    // the walked mesh's arrays, kept by the lane: read through the scene's mesh table at every step they were two dependent loads
    // in front of each node's
    const uint4*     nodes = nullptr;
    const float4*    verts = nullptr;
    ZD explicit AnyWalk4q(WideSpill<false>& spill) : stack(spill) {}
    ZD void start(const DMesh& mesh) {
        nodes = mesh.nodes4q;
        verts = mesh.tri_verts;
        stack.sp = 0;
        cur = 0;
        cur_count = 0;
    }
    ZD int step(const Ray& ray) {
        float unused = 0.f;
        if (0 == cur_count) {
            const Node4q n = Node4q::load(nodes, cur);
            const Node4q::Slabs g = n.slabs(ray);
            uint32_t     p0, p1, p2, p3;
            const bool   h0 = FLT_MAX != n.test<0>(g, ray, p0), h1 = FLT_MAX != n.test<1>(g, ray, p1);
            const bool   h2 = FLT_MAX != n.test<2>(g, ray, p2), h3 = FLT_MAX != n.test<3>(g, ray, p3);
            bool         have = false;
            uint32_t     np = 0;
            auto         take = [&](bool h, uint32_t p) {
                if (!h) return;
                if (have) stack.push(np, 0.f);
                np = p;
                have = true;
            };
            take(h3, p3);
            take(h2, p2);
            take(h1, p1);
            take(h0, p0);
            if (have) {
                cur = np & 0xFFFFFFu;
                cur_count = np >> 24;
            } else {
                if (0 == stack.sp) return kWalkDone;
                stack.pop(cur, cur_count, unused);
            }
            return kWalkRunning;
        }
        for (uint32_t i = cur, e = cur + cur_count; i < e; ++i) {
            f3 a, b, c;
            triangleVertices(verts, i, a, b, c);
            float    t, u, v;
            if (triangleIntersect(ray, a, b, c, t, u, v)) return kWalkHit;
        }
        if (0 == stack.sp) return kWalkDone;
        stack.pop(cur, cur_count, unused);
        return kWalkRunning;
    }
};

struct ClosestWalk4q {  // = bvh4qWalk + meshIntersect's leaf; the hit shrinks ray.max_t
    WideStack<true> stack;
    uint32_t        cur = 0, cur_count = 0;
    const uint4*    nodes = nullptr;  // as AnyWalk4q's
    const float4*   verts = nullptr;
    ZD explicit ClosestWalk4q(WideSpill<true>& spill) : stack(spill) {}
    ZD void start(const DMesh& mesh) {
        nodes = mesh.nodes4q;
        verts = mesh.tri_verts;
        stack.sp = 0;
        cur = 0;
        cur_count = 0;
    }
    ZD bool pop(const Ray& ray) {
        while (stack.sp > 0) {
            uint32_t r, n;
            float    dist;
            stack.pop(r, n, dist);
            if (dist <= ray.max_t) {
                cur = r;
                cur_count = n;
                return true;
            }
        }
        return false;
    }
    // `found(triangle, t, u, v)` for every hit nearer than the ray's end
    template <typename Found>
    ZD int step(Ray& ray, Found&& found) {
        if (0 == cur_count) {
            const Node4q n = Node4q::load(nodes, cur);
            const Node4q::Slabs g = n.slabs(ray);
            uint32_t     p0, p1, p2, p3;
            float        t0 = n.test<0>(g, ray, p0), t1 = n.test<1>(g, ray, p1), t2 = n.test<2>(g, ray, p2), t3 = n.test<3>(g, ray, p3);
#define ZYG_CSWAP(i, j)           \
    if (t##j < t##i) {            \
        const float    tt = t##i; \
        const uint32_t pp = p##i; \
        t##i = t##j, p##i = p##j; \
        t##j = tt, p##j = pp;     \
    }
            ZYG_CSWAP(0, 1)
            ZYG_CSWAP(2, 3)
            ZYG_CSWAP(0, 2)
            ZYG_CSWAP(1, 3)
            ZYG_CSWAP(1, 2)
#undef ZYG_CSWAP
            if (FLT_MAX != t3) stack.push(p3, t3);
            if (FLT_MAX != t2) stack.push(p2, t2);
            if (FLT_MAX != t1) stack.push(p1, t1);
            if (FLT_MAX != t0) {
                cur = p0 & 0xFFFFFFu;
                cur_count = p0 >> 24;
                return kWalkRunning;
            }
            return pop(ray) ? kWalkRunning : kWalkDone;
        }
        for (uint32_t i = cur, e = cur + cur_count; i < e; ++i) {
            f3 a, b, c;
            triangleVertices(verts, i, a, b, c);
            float    t, u, v;
            if (triangleIntersect(ray, a, b, c, t, u, v)) {
                ray.max_t = t;
                found(i, t, u, v);
            }
        }
        return pop(ray) ? kWalkRunning : kWalkDone;
    }
};

// ---- The eight-wide compressed tree (bvh8.hpp) one step at a time: the persistent mesh walkers' form (wavefront.hpp wf_shadow_u8,
// wf_mesh_closest8).  A lane's walk state is two *groups*, each a base index and hit bits: `ng` = the hit inner children of the node it
// came through (bits 24-31, already in this ray's front-to-back order; the low byte keeps the node's inner-child mask, which turns a
// slot into the child's index) and `tg` = the hit triangles among that node's leaves (bits 0-23).  What it cannot look at now goes
// to the stack as a group: one entry per visited node, not one per parked child.  Node steps and triangle steps are separate calls
// so that the kernel can let a wave's lanes do the one or the other together (leaves wait until enough lanes have one).
#ifndef ZYG_NODE8_STRIDE
#define ZYG_NODE8_STRIDE 20  // bvh8.hpp
#endif
#ifndef ZYG_STACK8_LDS
#define ZYG_STACK8_LDS 16  // entries (of two words) per lane in LDS: 32 KB per 256-thread block; deeper ones in scratch
#endif
constexpr int kStack8Lds = ZYG_STACK8_LDS, kStack8 = 40;  // zygpu_upload_scene: 2 entries per level of the tree at most

// the lanes' columns of the four-wide closest-hit walk's array: a lane runs one walk at a time (wf_mesh_closest8 tests the smaller
// meshes four-wide when it fetches a ray, before its eight-wide walk starts)
static_assert(!ZYG_LDS_STACK || kStack8Lds <= kLdsStack, "the group stack shares the wide stack's LDS");
ZD uint2* groupStackLds() {
#if ZYG_LDS_STACK
    return wideStackLds();
#else
    __shared__ uint2 entries[kStack8Lds * kLdsStackThreads];
    return entries;
#endif
}
struct GroupSpill {
    uint32_t x[kStack8 - kStack8Lds], y[kStack8 - kStack8Lds];
};
struct GroupStack {
    LdsUint2* lds;
    uint32_t *spill_x, *spill_y;
    int       sp = 0;
    ZD explicit GroupStack(GroupSpill& spill) : lds((LdsUint2*)(groupStackLds() + threadIdx.x)), spill_x(spill.x), spill_y(spill.y) {}
    ZD void push(uint2 g) {
        if (sp >= kStack8) return;
        if (sp < kStack8Lds) lds[sp * kLdsStackThreads] = WordPair{g.x, g.y};
        else spill_x[sp - kStack8Lds] = g.x, spill_y[sp - kStack8Lds] = g.y;
        ++sp;
    }
    ZD uint2 pop() {  // sp > 0
        --sp;
        if (sp < kStack8Lds) {
            const WordPair e = lds[sp * kLdsStackThreads];
            return make_uint2(e.x, e.y);
        }
        return make_uint2(spill_x[sp - kStack8Lds], spill_y[sp - kStack8Lds]);
    }
};

struct Walk8 {
    GroupStack    stack;
    uint2         ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    const uint4*  nodes = nullptr;
    const float4* tris = nullptr;
    // the ray in the walk's terms: origin, reciprocal direction (finite: a zero component counts as 1e-30), 7 - its octant in every
    // byte of a word
    f3            o, idir;
    uint32_t      oct_inv4 = 0;
    ZD explicit Walk8(GroupSpill& spill) : stack(spill) {}
    ZD void start(const DMesh& mesh, const Ray& ray) {
        nodes = mesh.nodes8;
        tris = mesh.tris8;
        o = ray.origin;
        const f3 d = ray.direction;
        auto     safe = [](float x) { return fabsf(x) > 1e-30f ? x : copysignf(1e-30f, x); };
        idir = mk3(1.f / safe(d.x), 1.f / safe(d.y), 1.f / safe(d.z));
        const uint32_t oct_inv = 7u ^ ((d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u));
        oct_inv4 = oct_inv * 0x01010101u;
        stack.sp = 0;
        ng = make_uint2(0u, (1u << (24u + oct_inv)) | 1u);  // the root: the one inner child, in slot 0, of a group whose first child is node 0
        tg = make_uint2(0u, 0u);
    }
    ZD bool nodePending() const { return 0u != (ng.y & 0xFF000000u); }
    ZD bool trianglesPending() const { return 0u != tg.y; }
    // the nearest hit child of `ng` becomes the node: its children against the ray over [min_t, max_t] -> new ng, tg (nodePending())
    ZD void nodeStep(float min_t, float max_t) {
        const uint32_t bit = 31u - uint32_t(__clz(int(ng.y)));
        ng.y &= ~(1u << bit);
        const uint32_t slot = (bit - 24u) ^ (oct_inv4 & 7u);
        const uint32_t node = ng.x + uint32_t(__popc(ng.y & 0xFFu & ((1u << slot) - 1u)));
        if (ng.y & 0xFF000000u) stack.push(ng);
        typedef uint32_t                                   Words __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(1))) Words* n = (const __attribute__((address_space(1))) Words*)(nodes + size_t(node) * (ZYG_NODE8_STRIDE / 4));
        const Words n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3], n4 = n[4];
        const f3    sc = mk3(__uint_as_float((n0.w & 255u) << 23), __uint_as_float(((n0.w >> 8) & 255u) << 23), __uint_as_float(((n0.w >> 16) & 255u) << 23));
        const f3    s = sc * idir, c = (mk3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z)) - o) * idir;
        // per axis the plane the ray meets first: the low one for a positive direction
        const bool  px = idir.x >= 0.f, py = idir.y >= 0.f, pz = idir.z >= 0.f;
        uint32_t    hits = 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t meta4 = half ? n1.w : n1.z;
            const uint32_t lox = half ? n2.y : n2.x, loy = half ? n2.w : n2.z, loz = half ? n3.y : n3.x;
            const uint32_t hix = half ? n3.w : n3.z, hiy = half ? n4.y : n4.x, hiz = half ? n4.w : n4.z;
            const uint32_t nx = px ? lox : hix, fx = px ? hix : lox, ny = py ? loy : hiy, fy = py ? hiy : loy, nz = pz ? loz : hiz, fz = pz ? hiz : loz;
            const uint32_t inner4 = ((meta4 & (meta4 << 1)) & 0x10101010u) >> 4;  // 1 in the bytes of inner children (position 24 and up)
            const uint32_t index4 = (meta4 ^ (oct_inv4 & (inner4 * 0xFFu))) & 0x1F1F1F1Fu;
            const uint32_t count4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float tnear = mx(mx(__fmaf_rn(float((nx >> (8 * k)) & 255u), s.x, c.x), __fmaf_rn(float((ny >> (8 * k)) & 255u), s.y, c.y)),
                                       mx(__fmaf_rn(float((nz >> (8 * k)) & 255u), s.z, c.z), min_t));
                const float tfar = mn(mn(__fmaf_rn(float((fx >> (8 * k)) & 255u), s.x, c.x), __fmaf_rn(float((fy >> (8 * k)) & 255u), s.y, c.y)),
                                      mn(__fmaf_rn(float((fz >> (8 * k)) & 255u), s.z, c.z), max_t));
                const uint32_t count = (count4 >> (8 * k)) & 7u, index = (index4 >> (8 * k)) & 31u;
                if (tnear <= tfar) hits |= ((1u << count) - 1u) << index;  // (an empty slot counts nothing)
            }
        }
        ng = make_uint2(n1.x, (hits & 0xFF000000u) | (n0.w >> 24));
        tg = make_uint2(n1.y, hits & 0x00FFFFFFu);
    }
    // a popped group is a node group or a triangle group
    ZD void adopt(uint2 g) {
        if (g.y & 0xFF000000u) ng = g;
        else tg = g, ng = make_uint2(0u, 0u);
    }
    // the next triangle of `tg` (trianglesPending()): corners and the reference's index
    ZD void nextTriangle(f3& a, f3& b, f3& c, uint32_t& reference) {
        const uint32_t bit = uint32_t(__ffs(int(tg.y))) - 1u;
        tg.y &= tg.y - 1u;
        typedef float                                            Corner __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(1))) Corner* v = (const __attribute__((address_space(1))) Corner*)(tris + size_t(3) * (tg.x + bit));
        const Corner                                             va = v[0], vb = v[1], vc = v[2];
        a = mk3(va.x, va.y, va.z);
        b = mk3(vb.x, vb.y, vb.z);
        c = mk3(vc.x, vc.y, vc.z);
        reference = __float_as_uint(va.w);
    }
};

// The eight-wide tree walked to the end by one lane (the kernels that are not the persistent walkers: a mesh of the mesh kernels met by
// wf_trace_shade's front entries and first tick, and by the POOL scenes' per-lane walks): found(reference, t, u, v) for every hit
// inside the ray's range; `any`: stop at the first one.
// Measured (round 5, same call): the any-hit walks of the per-lane tree walkers (wf_light / wf_scatter of the POOL scenes) gain 7 % on the
// outdoor scene with configs[2]'s sphere (scatter stage 294 -> 272 ms per 2 steps); the closest-hit walk loses 5 % there (205 -> 215)
// and the prop-table kernels lose with either (cornell_mesh 352 -> 344: the code sits in wf_scatter_h, whose large-mesh rays go to the
// persistent walker anyway).  Bit 0: the any-hit walk of the per-lane tree walkers; bit 1: the closest-hit walks; bit 2: the
// prop-table kernels' any-hit walk.
#ifndef ZYG_WALK8_INLINE
#define ZYG_WALK8_INLINE 1
#endif
template <bool ANY, typename Found>
ZD bool bvh8Walk(const DMesh& mesh, Ray& ray, Found&& found) {
    GroupSpill spill;
    Walk8      walk(spill);
    walk.start(mesh, ray);
    for (;;) {
        if (walk.nodePending() && !walk.trianglesPending()) walk.nodeStep(ray.min_t, ray.max_t);
        while (walk.trianglesPending()) {
            f3       a, b, c;
            uint32_t reference;
            float    t, u, v;
            walk.nextTriangle(a, b, c, reference);
            if (triangleIntersect(ray, a, b, c, t, u, v) && found(reference, t, u, v) && ANY) return true;
        }
        if (!walk.nodePending()) {
            if (0 == walk.stack.sp) return false;
            walk.adopt(walk.stack.pop());
        }
    }
}

// The any-hit form of the wide walk (shadow rays): the order the children are visited in does not change the answer, so nothing
// is sorted and no entry distance is kept: a parked child is a reference and a count.
template <typename Leaf>
ZD bool bvh4WalkAny(const float4* nodes, const Ray& ray, Leaf&& leaf) {
    uint32_t stack_ref[kStack4], stack_count[kStack4];
    int      sp = 0;
    uint32_t cur = 0, cur_count = 0;  // the root node
    for (;;) {
        while (0 == cur_count) {
            const float4* n = nodes + size_t(cur) * 8;
            const float4  a0 = n[0], a1 = n[1], b0 = n[2], b1 = n[3], c0 = n[4], c1 = n[5], d0 = n[6], d1 = n[7];
            const bool    h0 = FLT_MAX != nodeIntersect(a0, a1, ray), h1 = FLT_MAX != nodeIntersect(b0, b1, ray);
            const bool    h2 = FLT_MAX != nodeIntersect(c0, c1, ray), h3 = FLT_MAX != nodeIntersect(d0, d1, ray);
            bool          have = false;
            uint32_t      nr = 0, nc = 0;
            auto          take = [&](bool h, uint32_t r, uint32_t c) {
                if (!h) return;
                if (have && sp < kStack4) stack_ref[sp] = nr, stack_count[sp] = nc, ++sp;
                nr = r;
                nc = c;
                have = true;
            };
            take(h3, __float_as_uint(d0.w), __float_as_uint(d1.w));
            take(h2, __float_as_uint(c0.w), __float_as_uint(c1.w));
            take(h1, __float_as_uint(b0.w), __float_as_uint(b1.w));
            take(h0, __float_as_uint(a0.w), __float_as_uint(a1.w));
            if (have) {
                cur = nr;
                cur_count = nc;
            } else {
                if (0 == sp) return false;
                --sp;
                cur = stack_ref[sp];
                cur_count = stack_count[sp];
            }
        }
        if (leaf(cur, cur_count)) return true;
        if (0 == sp) return false;
        --sp;
        cur = stack_ref[sp];
        cur_count = stack_count[sp];
    }
}

// triangle_tree.zig:46-109; ray in object space; shrinks ray.max_t
// `accept(triangle, u, v)`: intersectOpacity's stochastic mask test (triangle_tree.zig:111-195)
// WIDE: walk the mesh's four-wide re-layout instead of the reference's BVH2: the kernel instantiations for scenes with a large
// mesh, in the relaxed-arithmetic units only (kernel_table.hpp ZYG_WIDE), and not where something depends on the reference's traversal order:
// the counters (STATS) and `accept` when it draws from the path's sampler.  The precise units always walk the BVH2: their
// result is the reference's bit for bit, exact ties between triangles (a ray through a shared edge) included, which the wide
// walk may resolve the other way round.  (For the 12-triangle boxes of the Cornell scene the wide walk's registers cost more
// than its shorter chain saves: measured 643 vs 600 Msamples/s, so scenes without a large mesh run the BVH2 instantiations.)
template <bool STATS, bool WIDE, typename Accept>
ZD bool meshIntersect(const DMesh& mesh, Ray& local_ray, Hit& hit, Counters& cnt, Accept&& accept) {
    uint32_t primitive = kNull;
    float    ht = 0.f, hu = 0.f, hv = 0.f;
    auto     leaf = [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            if (STATS) cnt.tris += 1;
            f3 a, b, c;
            triangleVertices(mesh, i, a, b, c);
            float    t, u, v;
            if (triangleIntersect(local_ray, a, b, c, t, u, v) && accept(i, u, v)) {
                local_ray.max_t = t;
                ht = t;
                hu = u;
                hv = v;
                primitive = i;
            }
        }
        return false;
    };
    // the wide re-layout where the mesh has one (large meshes) and nothing is counted: the counters are the BVH2 walk's
    if ((ZYG_WALK8_INLINE & 2) && !STATS && WIDE && mesh.nodes8) {  // (measured slower than the four-wide walk here: closest hits keep it)
        bvh8Walk<false>(mesh, local_ray, [&](uint32_t reference, float t, float u, float v) {
            if ((t < local_ray.max_t || kNull == primitive || reference < primitive) && accept(reference, u, v)) {
                local_ray.max_t = t;
                ht = t;
                hu = u;
                hv = v;
                primitive = reference;
            }
            return false;
        });
    } else
#if ZYG_BVH4Q
    if (!STATS && WIDE) bvh4qWalk(mesh.nodes4q, local_ray, leaf);
#else
    if (!STATS && WIDE) bvh4Walk(mesh.nodes4, local_ray, leaf);
#endif
    else bvhWalk<STATS>(mesh.nodes, 1u, local_ray, cnt, leaf);
    if (kNull == primitive) return false;
    hit.t = ht;
    hit.u = hu;
    hit.v = hv;
    hit.primitive = primitive;
    return true;
}

template <bool STATS, bool WIDE = false>
ZD bool meshIntersect(const DMesh& mesh, Ray& local_ray, Hit& hit, Counters& cnt) {
    return meshIntersect<STATS, WIDE>(mesh, local_ray, hit, cnt, [](uint32_t, float, float) { return true; });
}
ZD void meshInterpolateUv(const DMesh& mesh, uint32_t prim, float u, float v, float& uv0, float& uv1) {  // triangle_data.zig:137-149
    const uint32_t ia = mesh.triangles[3 * prim], ib = mesh.triangles[3 * prim + 1], ic = mesh.triangles[3 * prim + 2];
    const float    w = 1.f - u - v;
    uv0 = __fmaf_rn(mesh.uvs[2 * ia], w, __fmaf_rn(mesh.uvs[2 * ib], u, mesh.uvs[2 * ic] * v));
    uv1 = __fmaf_rn(mesh.uvs[2 * ia + 1], w, __fmaf_rn(mesh.uvs[2 * ib + 1], u, mesh.uvs[2 * ic + 1] * v));
}

// triangle_tree.zig:197-242
template <bool STATS, bool WIDE = false, bool TREE = false>  // TREE: called from a per-lane walk of the scene's trees (the POOL / MESHL / EXTRA kernels)
ZD bool meshIntersectP(const DMesh& mesh, Ray& local_ray, Counters& cnt) {
    auto leaf = [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            if (STATS) cnt.tris += 1;
            f3 a, b, c;
            triangleVertices(mesh, i, a, b, c);
            float    t, u, v;
            if (triangleIntersect(local_ray, a, b, c, t, u, v)) return true;
        }
        return false;
    };
    if ((ZYG_WALK8_INLINE & (TREE ? 1 : 4)) && !STATS && WIDE && mesh.nodes8) return bvh8Walk<true>(mesh, local_ray, [](uint32_t, float, float, float) { return true; });
#if ZYG_BVH4Q
    if (!STATS && WIDE) return bvh4qWalkAny(mesh.nodes4q, local_ray, leaf);
#else
    if (!STATS && WIDE) return bvh4WalkAny(mesh.nodes4, local_ray, leaf);
#endif
    return bvhWalk<STATS>(mesh.nodes, 1u, local_ray, cnt, leaf);
}

// rectangle.zig:30-62
ZD bool rectangleIntersect(const Ray& ray, const Trafo& trafo, float& out_t, float& out_u, float& out_v) {
    const f3    n = trafo.r2;
    const float d = dot3(n, trafo.position);
    const float hit_t = -(dot3(n, ray.origin) - d) / dot3(n, ray.direction);
    if (hit_t >= ray.min_t && ray.max_t >= hit_t) {
        const f3    p = ray.point(hit_t);
        const f3    k = p - trafo.position;
        const f3    t = -trafo.r0;
        const float u = dot3(t, k) / (0.5f * trafo.scale.x);
        if (u > 1.f || u < -1.f) return false;
        const f3    b = -trafo.r1;
        const float v = dot3(b, k) / (0.5f * trafo.scale.y);
        if (v > 1.f || v < -1.f) return false;
        out_t = hit_t;
        out_u = u;
        out_v = v;
        return true;
    }
    return false;
}

ZD Ray worldToObjectRay(const Trafo& t, const Ray& ray) {  // composed_transformation.zig:123-130
    return Ray::init(t.worldToObjectPoint(ray.origin), t.worldToObjectVector(ray.direction), ray.min_t, ray.max_t);
}

ZD bool propVisibleAtDepth(uint32_t flags, uint32_t depth_surface) {  // prop.zig:39-49
    return 0 == depth_surface ? 0 != (flags & ZYGPU_PROP_VISIBLE_IN_CAMERA) : 0 != (flags & ZYGPU_PROP_VISIBLE_IN_REFLECTION);
}

// ---- Disk (disk.zig:28-60, 100-116) and Cube (cube.zig:23-67): geometry only
ZD bool diskIntersect(const Ray& ray, const Trafo& trafo, float& out_t, float& out_u, float& out_v) {
    const f3    normal = trafo.r2;
    const float d = dot3(normal, trafo.position);
    const float denom = -dot3(normal, ray.direction);
    const float numer = dot3(normal, ray.origin) - d;
    const float hit_t = numer / denom;
    if (hit_t >= ray.min_t && ray.max_t >= hit_t) {
        const f3    k = ray.point(hit_t) - trafo.position;
        const float l = dot3(k, k);
        const float radius = 0.5f * trafo.scale.x;
        if (l <= radius * radius) {
            const f3 sk = k / S3(radius);
            out_u = -dot3(trafo.r0, sk);
            out_v = -dot3(trafo.r1, sk);
            out_t = hit_t;
            return true;
        }
    }
    return false;
}
ZD bool cubeIntersect(const Ray& ray, const Trafo& trafo, float& out_t) {
    const Ray   local = worldToObjectRay(trafo, ray);
    const float hit_t = aabbIntersectP(make_float4(-0.5f, -0.5f, -0.5f, -0.5f), make_float4(0.5f, 0.5f, 0.5f, 0.5f), local);
    if (hit_t < ray.max_t) {
        out_t = hit_t;
        return true;
    }
    return false;
}
ZD bool cubeIntersectP(const Ray& ray, const Trafo& trafo) {
    const Ray local = worldToObjectRay(trafo, ray);
    return aabbIntersect(make_float4(-0.5f, -0.5f, -0.5f, -0.5f), make_float4(0.5f, 0.5f, 0.5f, 0.5f), local);
}

// the uv of a point on the sphere with world normal n (sphere.zig:110-115)
ZD void sphereUv(const Trafo& trafo, f3 n, float& u, float& v) {
    const f3    xyz_ = normalize3(trafo.worldToObjectNormal(n));
    const float phi = -atan2F(xyz_.x, xyz_.z) + kPi;
    const float theta = acosF(xyz_.y);
    u = phi * (0.5f * kPiInv);
    v = theta * kPiInv;
}
// sphere.zig:94-146: the nearer root the mask lets through
template <typename Draw>
ZD bool sphereIntersectOpacity(const DScene& s, const ZygpuMaterial& m, const Ray& ray, const Trafo& trafo, float& out_t, Draw& draw) {
    const float idl = 1.f / length3(ray.direction);
    const f3    nd = ray.direction * S3(idl);
    const f3    v = trafo.position - ray.origin;
    const float b = dot3(nd, v);
    const f3    remedy_term = v - S3(b) * nd;
    const float radius = 0.5f * trafo.scale.x;
    const float discriminant = radius * radius - dot3(remedy_term, remedy_term);
    if (discriminant > 0.f) {
        const float dist = sqrtf(discriminant);
        const float ts[2] = {(b - dist) * idl, (b + dist) * idl};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float t = ts[i];
            if (t >= ray.min_t && ray.max_t >= t) {
                float uu, vv;
                sphereUv(trafo, normalize3(ray.point(t) - trafo.position), uu, vv);
                if (stochasticOpacity(s, m, uu, vv, draw)) {
                    out_t = t;
                    return true;
                }
            }
        }
    }
    return false;
}

// prop_tree.zig:56-116 over the solid TLAS (closest hit); shrinks ray.max_t
// EXTRA: the scene has analytic shapes other than Rectangle (Sphere, Disk, Cube); kept out of the common instantiations,
// where their code cost 12 % (registers of wf_shade's first pass and wf_intersect; measured 426 vs 374 Msamples/s)
// `draw`: the path sampler's sample1D for the opacity masks (prop.zig:191-195 -> shape.intersectOpacity), EXTRA only
// WIDEC (relaxed units, scenes with a mesh that has the quantised four-wide re-layout: DMesh::wide): such a mesh's closest hit
// through that walk.  An instantiation of its own: the walk's LDS stack and registers cost a scene without such a mesh 3 %
// (cornell_glass 176 -> 171 Msamples/s), the outdoor scene's 1 M-triangle mesh returns 2-6 % (3840x2160: 495 -> 505, 1080p: 428 -> 450).
template <bool WIDEC>
ZD bool tableIntersectComplete(const DScene& s, Ray& ray, uint32_t depth_surface, Hit& hit);
ZD bool tableVisibilityComplete(const DScene& s, Ray& ray);

template <bool STATS, bool EXTRA, bool WIDEC = false, typename Draw>
ZD bool sceneIntersect(const DScene& s, Ray& ray, uint32_t depth_surface, Hit& hit, Counters& cnt, Draw& draw) {
    if (ZYG_FAST && !STATS && EXTRA && s.table_complete) return tableIntersectComplete<WIDEC>(s, ray, depth_surface, hit);  // (uniform; no masks: `draw` idle)
    hit.prop = kNull;
    bvhWalk<STATS>(s.solid_nodes, s.num_solid_nodes, ray, cnt, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            const uint32_t  p = s.solid_indices[i];
            const ZygpuProp prop = s.props[p];
            if (!propVisibleAtDepth(prop.flags, depth_surface)) continue;
            if (STATS) cnt.instances += 1;
            if (!aabbIntersect(s.aabbs[2 * p], s.aabbs[2 * p + 1], ray)) continue;
            const Trafo trafo = propTrafoAt<EXTRA, true>(s, p, ray.time);
            const bool  masked = EXTRA && 0 != (prop.flags & ZYGPU_PROP_EVALUATE_VISIBILITY);
            if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {
                float t, u, v;
                if (rectangleIntersect(ray, trafo, t, u, v) &&  // rectangle.zig:64-100
                    !(masked && !stochasticOpacity(s, s.materials[s.material_ids[prop.parts_start]], 0.5f * (u + 1.f), 0.5f * (v + 1.f), draw))) {
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit.primitive = 0;
                    hit.prop = p;
                    ray.max_t = t;
                }
            } else if (EXTRA && (ZYGPU_SHAPE_SPHERE == prop.shape || ZYGPU_SHAPE_CUBE == prop.shape)) {
                float t;
                if (ZYGPU_SHAPE_SPHERE == prop.shape
                        ? (masked ? sphereIntersectOpacity(s, s.materials[s.material_ids[prop.parts_start]], ray, trafo, t, draw) : sphereIntersect(ray, trafo, t))
                        : cubeIntersect(ray, trafo, t)) {
                    hit.t = t;
                    hit.u = hit.v = 0.f;
                    hit.primitive = 0;
                    hit.prop = p;
                    ray.max_t = t;
                }
            } else if (EXTRA && ZYGPU_SHAPE_DISK == prop.shape) {
                float t, u, v;
                if (diskIntersect(ray, trafo, t, u, v) &&  // disk.zig:60-96
                    !(masked && !stochasticOpacity(s, s.materials[s.material_ids[prop.parts_start]], 0.5f * (u + 1.f), 0.5f * (v + 1.f), draw))) {
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit.primitive = 0;
                    hit.prop = p;
                    ray.max_t = t;
                }
            } else if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
                Ray          local = worldToObjectRay(trafo, ray);
                const DMesh& mesh = s.meshes[prop.mesh];
                Hit          h;
                bool         found;
                if (masked) {
                    found = meshIntersect<STATS, false>(mesh, local, h, cnt, [&](uint32_t k, float u, float v) {
                        const ZygpuMaterial& m = s.materials[s.material_ids[prop.parts_start + mesh.triangle_parts[k]]];
                        if (!(m.flags & ZYGPU_MATF_EVALUATE_VISIBILITY)) return true;
                        float uv0, uv1;
                        meshInterpolateUv(mesh, k, u, v, uv0, uv1);
                        return stochasticOpacity(s, m, uv0, uv1, draw);
                    });
                } else {
                    found = (WIDEC && ZYG_FAST && !STATS && mesh.wide) ? meshIntersect<false, true>(mesh, local, h, cnt) : meshIntersect<STATS>(mesh, local, h, cnt);
                }
                if (found) {
                    hit.t = h.t;
                    hit.u = h.u;
                    hit.v = h.v;
                    hit.primitive = h.primitive;
                    hit.prop = p;
                    ray.max_t = h.t;
                }
            }
        }
        return false;
    });
    return kNull != hit.prop;
}

template <bool STATS, bool EXTRA>
ZD bool sceneIntersect(const DScene& s, Ray& ray, uint32_t depth_surface, Hit& hit, Counters& cnt) {
    NoDraw none;
    return sceneIntersect<STATS, EXTRA>(s, ray, depth_surface, hit, cnt, none);
}

// ---------------------------------------------------------------- small scenes: the wave walks the prop table together
// In a scene of a few props nearly every ray overlaps nearly every prop's bounds (Cornell box: 8.2 instance tests per ray
// for 7 + 2 props), so the top-level tree orders nothing and its per-lane walk only scatters the wave: lanes sit in
// different props and shapes, every step is a dependent vector load.  Here all lanes test prop k at the same time, with its
// record in SGPRs, cheap shapes first (their hits shorten the ray before the meshes' bounds are tested), and a mesh's BLAS
// is entered by all its lanes at once.
//
// Same result as the tree walk: the closest hit is the minimum over the props whatever the order; inside a mesh the walk is
// the reference's.  Only an exact tie between two props (bit-equal t) depends on the reference's order, where the prop
// tested later wins (`max_t >= hit_t`): then the two props' places in this ray's walk of the tree decide (tlasOrderLater).
ZD Trafo entryTrafo(const TlasEntry& e) {
    Trafo t;
    t.r0 = {e.r0.x, e.r0.y, e.r0.z};
    t.r1 = {e.r1.x, e.r1.y, e.r1.z};
    t.r2 = {e.r2.x, e.r2.y, e.r2.z};
    t.scale = {e.r0.w, e.r1.w, e.r2.w};
    t.position = {e.pos.x, e.pos.y, e.pos.z};
    return t;
}

// whether prop `b` comes after prop `a` when `ray` walks the solid tree nearer child first (prop_tree.zig:86-104); the
// order of two props does not depend on how far the ray reaches, so the walk runs without culling by max_t
ZD bool tlasOrderLater(const DScene& s, const Ray& ray_in, uint32_t a, uint32_t b) {
    Ray ray = ray_in;
    ray.max_t = kRayMaxT;
    Counters none{0, 0, 0};
    uint32_t seen = kNull;
    bvhWalk<false>(s.solid_nodes, s.num_solid_nodes, ray, none, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            const uint32_t p = s.solid_indices[i];
            if (p == a || p == b) {
                seen = p;
                return true;
            }
        }
        return false;
    });
    return seen == a;  // a was met first: b is the later one
}

// Whether a ray has to be tested against a prop-table entry's shape.  The reference tests the prop's world bounds first
// (prop.zig:199-237); for a Rectangle the shape's own test costs as much as the bounds' and decides the same thing (a hit lies
// inside the bounds), and in a wave of incoherent rays some lane passes the bounds anyway: the relaxed units go straight to
// the Rectangle (RELAXED_RULES; an ulp at the bounds' edge is within their contract, not within the precise units').  Closest hits
// only: measured in one call on configs[1], wf_trace_shade 111.6 -> 108.4 ms per 8 steps; the shadow rays of wf_scatter, most of
// which miss most bounds, 153.8 -> 154.7 with the same change, so they keep the bounds test.
#ifndef ZYG_RECT_DIRECT
#define ZYG_RECT_DIRECT 1
#endif
template <bool RELAXED_RULES>
ZD bool entryBoundsHit(const TlasEntry& e, const Ray& ray) {
    if (RELAXED_RULES && ZYG_RECT_DIRECT && ZYGPU_SHAPE_RECTANGLE == e.shape) return true;  // uniform
    return aabbIntersect(e.bmin, e.bmax, ray);
}

// A Rectangle of the prop tables as the relaxed units test it: the entry's rows, worked out in f64 at upload from the
// transformation rectangleIntersect uses, give the ray's origin and direction in the Rectangle's own (u, v, height) frame: the
// same t, u, v as rectangle.zig:30-62 to rounding, without the bounds test, the plane constant, the scaling divisions.
#ifndef ZYG_RECT_UNIT
#define ZYG_RECT_UNIT 1
#endif
#ifndef ZYG_RECT_UNIT_SHADOW
#define ZYG_RECT_UNIT_SHADOW ZYG_RECT_UNIT  // the shadow rays too (they kept the bounds test in front of the reference's form)
#endif
ZD bool unitRectangleIntersect(const Ray& ray, const TlasEntry& e, float& out_t, float& out_u, float& out_v) {
    const float4 r0 = e.fu, r1 = e.fv, r2 = e.fn;
    const float  oz = fmaf(r2.x, ray.origin.x, fmaf(r2.y, ray.origin.y, fmaf(r2.z, ray.origin.z, r2.w)));
    const float  dz = fmaf(r2.x, ray.direction.x, fmaf(r2.y, ray.direction.y, r2.z * ray.direction.z));
    const float  hit_t = -oz * smallReciprocal(dz);
    const float  ox = fmaf(r0.x, ray.origin.x, fmaf(r0.y, ray.origin.y, fmaf(r0.z, ray.origin.z, r0.w)));
    const float  dx = fmaf(r0.x, ray.direction.x, fmaf(r0.y, ray.direction.y, r0.z * ray.direction.z));
    const float  oy = fmaf(r1.x, ray.origin.x, fmaf(r1.y, ray.origin.y, fmaf(r1.z, ray.origin.z, r1.w)));
    const float  dy = fmaf(r1.x, ray.direction.x, fmaf(r1.y, ray.direction.y, r1.z * ray.direction.z));
    const float  u = fmaf(hit_t, dx, ox), v = fmaf(hit_t, dy, oy);
    if (hit_t >= ray.min_t && ray.max_t >= hit_t && fabsf(u) <= 1.f && fabsf(v) <= 1.f) {
        out_t = hit_t;
        out_u = u;
        out_v = v;
        return true;
    }
    return false;
}

// closest hit over the solid props (the semantics of sceneIntersect<false, false>); WIDE: meshes through their wide re-layout
template <bool EXACT_TIES, bool WIDE>
ZD bool sceneIntersectUniform(const DScene& s, Ray& ray, uint32_t depth_surface, Hit& hit) {
    hit.prop = kNull;
    Counters none{0, 0, 0};
    for (uint32_t k = 0; k < s.num_solid_table; ++k) {
        const TlasEntry& e = s.solid_table[k];
        if (!propVisibleAtDepth(e.flags, depth_surface)) continue;  // uniform
        if (!entryBoundsHit<!EXACT_TIES>(e, ray)) continue;
        const Trafo trafo = entryTrafo(e);
        Hit         h;
        bool        found;
        if (ZYGPU_SHAPE_RECTANGLE == e.shape) {
            h.primitive = 0;
            found = (!EXACT_TIES && ZYG_RECT_UNIT) ? unitRectangleIntersect(ray, e, h.t, h.u, h.v) : rectangleIntersect(ray, trafo, h.t, h.u, h.v);
#if ZYG_EXPERIMENT_SKIP_SMALL  // timing experiment only: the small meshes are not there
        } else if (!EXACT_TIES && e.small_count > 0 && ZYG_EXPERIMENT_SKIP_SMALL > 1) {
            found = false;
#endif
        } else if (!EXACT_TIES && e.small_count > 0) {
            // every triangle of a small mesh, in the mesh's order (the reference's order only matters for exact ties)
            Ray local = worldToObjectRay(trafo, ray);
            found = false;
            forSmallItems(s, e.small_first, e.small_count, [&](const SmallItem& item) {
                float    t, u, v;
                uint32_t primitive;
                if (smallItemIntersect(local, item, t, u, v, primitive)) {
                    local.max_t = t;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                    h.primitive = primitive;
                    found = true;
                }
            });
        } else {
            Ray local = worldToObjectRay(trafo, ray);
            found = meshIntersect<false, WIDE>(s.meshes[e.mesh], local, h, none);
        }
        if (found && (h.t < ray.max_t || kNull == hit.prop || !EXACT_TIES || tlasOrderLater(s, ray, hit.prop, e.prop))) {
            hit.t = h.t;
            hit.u = h.u;
            hit.v = h.v;
            hit.primitive = h.primitive;
            hit.prop = e.prop;
            ray.max_t = h.t;
        }
    }
    return kNull != hit.prop;
}

// closest hit over the prop table's entries in front of its large meshes (the relaxed units' rules: no tie rule, small meshes
// triangle by triangle); shrinks ray.max_t
ZD void tableClosestBeforeLargeMeshes(const DScene& s, Ray& ray, uint32_t depth_surface, Hit& hit) {
    Counters none{0, 0, 0};
    for (uint32_t k = 0; k < s.first_large; ++k) {
        const TlasEntry& e = s.solid_table[k];
        if (!propVisibleAtDepth(e.flags, depth_surface)) continue;
        if (!entryBoundsHit<true>(e, ray)) continue;
        const Trafo trafo = entryTrafo(e);
        Hit         h;
        bool        found;
        if (ZYGPU_SHAPE_RECTANGLE == e.shape) {
            h.primitive = 0;
            found = ZYG_RECT_UNIT ? unitRectangleIntersect(ray, e, h.t, h.u, h.v) : rectangleIntersect(ray, trafo, h.t, h.u, h.v);
        } else if (e.small_count > 0) {
            Ray local = worldToObjectRay(trafo, ray);
            found = false;
            forSmallItems(s, e.small_first, e.small_count, [&](const SmallItem& item) {
                float    t, u, v;
                uint32_t primitive;
                if (smallItemIntersect(local, item, t, u, v, primitive)) {
                    local.max_t = t;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                    h.primitive = primitive;
                    found = true;
                }
            });
        } else {
            Ray local = worldToObjectRay(trafo, ray);
            found = meshIntersect<false, true>(s.meshes[e.mesh], local, h, none);
        }
        if (found) {
            hit.t = h.t;
            hit.u = h.u;
            hit.v = h.v;
            hit.primitive = h.primitive;
            hit.prop = e.prop;
            ray.max_t = h.t;
        }
    }
}

// The prop table's entries in front of its large meshes (DScene::first_large: Rectangles and the meshes below the wide-BLAS
// threshold), as the relaxed units test them (small meshes triangle by triangle): true = one of them blocks the ray.
ZD bool tableBlocksBeforeLargeMeshes(const DScene& s, const Ray& ray) {
    Counters none{0, 0, 0};
    bool     blocked = false;
    for (uint32_t k = 0; k < s.first_large && !blocked; ++k) {
        const TlasEntry& e = s.solid_table[k];
        if (0 == (e.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW)) continue;
        if (ZYG_RECT_UNIT_SHADOW && ZYGPU_SHAPE_RECTANGLE == e.shape) {  // uniform
            float t, u, v;
            blocked = unitRectangleIntersect(ray, e, t, u, v);
            continue;
        }
        if (!aabbIntersect(e.bmin, e.bmax, ray)) continue;
        const Trafo trafo = entryTrafo(e);
        if (ZYGPU_SHAPE_RECTANGLE == e.shape) {
            float t, u, v;
            blocked = rectangleIntersect(ray, trafo, t, u, v);
        } else if (e.small_count > 0) {
            const Ray local = worldToObjectRay(trafo, ray);
            forSmallItems(s, e.small_first, e.small_count, [&](const SmallItem& item) { blocked = blocked || smallItemBlocks(local, item); });
        } else {
            Ray local = worldToObjectRay(trafo, ray);
            blocked = meshIntersectP<false, true>(s.meshes[e.mesh], local, none);  // (only the wide-BLAS instantiations call this)
        }
    }
    return blocked;
}

// any-hit over the solid props (the semantics of sceneVisibility<false, false>): true = unoccluded
template <bool WIDE, bool SMALL = false>
ZD bool sceneVisibilityUniform(const DScene& s, Ray& ray) {
    Counters none{0, 0, 0};
    bool     blocked = false;
    for (uint32_t k = 0; k < s.num_solid_table; ++k) {
        const TlasEntry& e = s.solid_table[k];
        if (0 == (e.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW)) continue;  // uniform
        if (SMALL && ZYG_RECT_UNIT_SHADOW && ZYGPU_SHAPE_RECTANGLE == e.shape) {  // uniform; SMALL: the relaxed units' rules
            float t, u, v;
            if (!blocked) blocked = unitRectangleIntersect(ray, e, t, u, v);
        } else if (!blocked && aabbIntersect(e.bmin, e.bmax, ray)) {
            const Trafo trafo = entryTrafo(e);
            if (ZYGPU_SHAPE_RECTANGLE == e.shape) {
                float t, u, v;
                blocked = rectangleIntersect(ray, trafo, t, u, v);
#if ZYG_EXPERIMENT_SKIP_SMALL
            } else if (SMALL && e.small_count > 0 && ZYG_EXPERIMENT_SKIP_SMALL > 1) {
#endif
            } else if (SMALL && e.small_count > 0) {
                const Ray local = worldToObjectRay(trafo, ray);
                forSmallItems(s, e.small_first, e.small_count, [&](const SmallItem& item) { blocked = blocked || smallItemBlocks(local, item); });
            } else {
                Ray local = worldToObjectRay(trafo, ray);
                blocked = meshIntersectP<false, WIDE>(s.meshes[e.mesh], local, none);
            }
        }
        if (0 == __ballot(!blocked)) break;  // every lane of the wave has its occluder
    }
    return !blocked;
}

// The prop table in the complete kernels (DScene::table_complete; relaxed units): the walks above with the shapes the complete
// kernels know (Sphere, Disk, Cube: sceneIntersect's / sceneVisibility's own tests) and a mesh walked as those kernels' per-lane walk
// walks it (its wide re-layout where it has one: the walk's stack is the one these kernels carry anyway).  What the wave gains is what
// the plain kernels gained in round 2: no per-lane walk of the top-level tree, the props' records through the scalar cache, all lanes in
// the same shape test, cheap shapes first.
template <bool WIDEC>
ZD bool tableIntersectComplete(const DScene& s, Ray& ray, uint32_t depth_surface, Hit& hit) {
    hit.prop = kNull;
    Counters none{0, 0, 0};
    for (uint32_t k = 0; k < s.num_solid_table; ++k) {
        const TlasEntry& e = s.solid_table[k];
        if (!propVisibleAtDepth(e.flags, depth_surface)) continue;  // uniform
        if (!entryBoundsHit<true>(e, ray)) continue;
        Hit  h;
        bool found;
        h.u = h.v = 0.f;
        h.primitive = 0;
        if (ZYGPU_SHAPE_RECTANGLE == e.shape) {
            found = unitRectangleIntersect(ray, e, h.t, h.u, h.v);
        } else if (ZYGPU_SHAPE_TRIANGLE_MESH == e.shape) {
            const Trafo trafo = entryTrafo(e);
            Ray         local = worldToObjectRay(trafo, ray);
            if (e.small_count > 0) {
                found = false;
                forSmallItems(s, e.small_first, e.small_count, [&](const SmallItem& item) {
                    float    t, u, v;
                    uint32_t primitive;
                    if (smallItemIntersect(local, item, t, u, v, primitive)) {
                        local.max_t = t;
                        h.t = t;
                        h.u = u;
                        h.v = v;
                        h.primitive = primitive;
                        found = true;
                    }
                });
            } else {
                const DMesh& mesh = s.meshes[e.mesh];
                found = (WIDEC && mesh.wide) ? meshIntersect<false, true>(mesh, local, h, none) : meshIntersect<false>(mesh, local, h, none);
            }
        } else if (ZYGPU_SHAPE_SPHERE == e.shape) {
            found = sphereIntersect(ray, entryTrafo(e), h.t);
        } else if (ZYGPU_SHAPE_CUBE == e.shape) {
            found = cubeIntersect(ray, entryTrafo(e), h.t);
        } else {
            found = diskIntersect(ray, entryTrafo(e), h.t, h.u, h.v);
        }
        if (found) {
            hit.t = h.t;
            hit.u = h.u;
            hit.v = h.v;
            hit.primitive = h.primitive;
            hit.prop = e.prop;
            ray.max_t = h.t;
        }
    }
    return kNull != hit.prop;
}

ZD bool tableVisibilityComplete(const DScene& s, Ray& ray) {  // true = unoccluded
    Counters none{0, 0, 0};
    bool     blocked = false;
    for (uint32_t k = 0; k < s.num_solid_table; ++k) {
        const TlasEntry& e = s.solid_table[k];
        if (0 == (e.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW)) continue;  // uniform
        if (ZYGPU_SHAPE_RECTANGLE == e.shape) {                      // uniform
            float t, u, v;
            if (!blocked) blocked = unitRectangleIntersect(ray, e, t, u, v);
        } else if (!blocked && aabbIntersect(e.bmin, e.bmax, ray)) {
            const Trafo trafo = entryTrafo(e);
            if (ZYGPU_SHAPE_TRIANGLE_MESH == e.shape) {
                Ray local = worldToObjectRay(trafo, ray);
                if (e.small_count > 0) {
                    forSmallItems(s, e.small_first, e.small_count, [&](const SmallItem& item) { blocked = blocked || smallItemBlocks(local, item); });
                } else {
                    const DMesh& mesh = s.meshes[e.mesh];
                    blocked = mesh.wide ? meshIntersectP<false, true, true>(mesh, local, none) : meshIntersectP<false>(mesh, local, none);
                }
            } else if (ZYGPU_SHAPE_SPHERE == e.shape) {
                float t;
                blocked = sphereIntersect(ray, trafo, t);
            } else if (ZYGPU_SHAPE_CUBE == e.shape) {
                blocked = cubeIntersectP(ray, trafo);
            } else {
                float t, u, v;
                blocked = diskIntersect(ray, trafo, t, u, v);
            }
        }
        if (0 == __ballot(!blocked)) break;  // every lane of the wave has its occluder
    }
    return !blocked;
}

// prop_tree.zig:185-240 + prop.zig:199-237: true = unoccluded
template <bool STATS, bool EXTRA>
ZD bool sceneVisibility(const DScene& s, Ray& ray, Counters& cnt) {
    if (ZYG_FAST && !STATS && EXTRA && s.table_complete) return tableVisibilityComplete(s, ray);  // uniform
    const bool blocked = bvhWalk<STATS>(s.solid_nodes, s.num_solid_nodes, ray, cnt, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            const uint32_t  p = s.solid_indices[i];
            const ZygpuProp prop = s.props[p];
            if (0 == (prop.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW)) continue;
            if (STATS) cnt.instances += 1;
            if (!aabbIntersect(s.aabbs[2 * p], s.aabbs[2 * p + 1], ray)) continue;
            const Trafo trafo = propTrafoAt<EXTRA>(s, p, ray.time);
            if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {
                float t, u, v;
                if (rectangleIntersect(ray, trafo, t, u, v)) return true;
            } else if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
                Ray          local = worldToObjectRay(trafo, ray);
                const DMesh& mesh = s.meshes[prop.mesh];
                if ((ZYG_FAST && !STATS && mesh.wide) ? meshIntersectP<false, true, true>(mesh, local, cnt) : meshIntersectP<STATS>(mesh, local, cnt)) return true;
            } else if (EXTRA && ZYGPU_SHAPE_SPHERE == prop.shape) {  // sphere.zig:148-172 intersectP: the same two roots
                float t;
                if (sphereIntersect(ray, trafo, t)) return true;
            } else if (EXTRA && ZYGPU_SHAPE_DISK == prop.shape) {
                float t, u, v;
                if (diskIntersect(ray, trafo, t, u, v)) return true;
            } else if (EXTRA && ZYGPU_SHAPE_CUBE == prop.shape) {
                if (cubeIntersectP(ray, trafo)) return true;
            }
        }
        return false;
    });
    return !blocked;
}

ZD f3 attenuation3(f3 c, float distance) {  // collision_coefficients.zig:67-69
    const f3 x = S3(-distance) * c;
    return mk3(expF(x.x), expF(x.y), expF(x.z));
}

// Glass.visibility (glass_material.zig:69-115): thin glass = Fresnel transmission x absorption over the slab, let through
// further by the mask's openings; o = the mask's opacity at the hit; false = blocked
ZD bool glassVisibility(const ZygpuMaterial& m, f3 wi, f3 geo_n, float o, f3& tr) {
    if (m.thickness > 0.f) {
        const float eta_i = 1.f, eta_t = m.ior;
        const float n_dot_wo = mn(fabsf(dot3(geo_n, wi)), 1.f);
        const float eta = eta_i / eta_t;
        const float sint2 = (eta * eta) * (1.f - n_dot_wo * n_dot_wo);
        if (sint2 >= 1.f) {
            if (o < 1.f) {
                tr = tr * S3(1.f - o);
                return true;
            }
            return false;
        }
        const float n_dot_t = sqrtf(1.f - sint2);
        const float t0 = eta_t * n_dot_wo, t1 = eta_i * n_dot_t;  // fresnel.dielectric (fresnel.zig:31-43)
        const float r_p = (t0 - t1) / (t0 + t1);
        const float t2 = eta_i * n_dot_wo, t3 = eta_t * n_dot_t;
        const float r_o = (t2 - t3) / (t2 + t3);
        const float f = 0.5f * (r_p * r_p + r_o * r_o);
        const float n_dot_wi = clampf(n_dot_wo, 0.00001f, 1.f);
        const float approx_dist = m.thickness / n_dot_wi;
        const f3    attenuation = attenuation3(mk3(m.absorption[0], m.absorption[1], m.absorption[2]), approx_dist);
        const f3    ta = min3v(S3(1.f - o) + attenuation, S3(1.f));
        tr = tr * (S3(1.f - f) * ta);
        return true;
    }
    if (o < 1.f) {
        tr = tr * S3(1.f - o);
        return true;
    }
    return false;
}
// Material.visibility (material.zig:227-248): the mask's opacity, drawn with the path's sampler, scales the transmittance
// MASKS = false: no material of the scene has a mask (opacity 1); the texture code stays out of the kernel
template <bool MASKS, typename Draw>
ZD bool materialVisibility(const DScene& s, const ZygpuMaterial& m, f3 wi, f3 geo_n, float u, float v, Draw& draw, f3& tr) {
    const float o = MASKS ? maskOpacity(s, m, u, v, draw) : 1.f;
    if (ZYGPU_MAT_GLASS == m.kind) return glassVisibility(m, wi, geo_n, o, tr);
    if (o < 1.f) {
        tr = tr * S3(1.f - o);
        return true;
    }
    return false;
}

// Scene.visibility with transmittance (prop_tree.zig:185-240, prop.zig:199-237, shape.zig:235-245): props whose materials
// evaluate visibility (thin Glass, opacity masks) attenuate `tr` instead of blocking (rectangle.zig:154-186,
// sphere.zig:176-228, disk.zig:136-170, triangle_tree.zig:244-306); true = not blocked
template <bool STATS, bool MASKS, typename Draw>
ZD bool sceneVisibilityTr(const DScene& s, Ray& ray, Counters& cnt, f3& tr, Draw& draw) {
    if (ZYG_FAST && !STATS && s.table_complete) return tableVisibilityComplete(s, ray);  // (uniform; nothing in such a scene attenuates: tr stays)
    const bool blocked = bvhWalk<STATS>(s.solid_nodes, s.num_solid_nodes, ray, cnt, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            const uint32_t  p = s.solid_indices[i];
            const ZygpuProp prop = s.props[p];
            if (0 == (prop.flags & ZYGPU_PROP_VISIBLE_IN_SHADOW)) continue;
            if (STATS) cnt.instances += 1;
            if (!aabbIntersect(s.aabbs[2 * p], s.aabbs[2 * p + 1], ray)) continue;
            const Trafo trafo = propTrafoAt<true>(s, p, ray.time);  // the POOL / mask kernels: all know animated props
            const bool  evaluate = 0 != (prop.flags & ZYGPU_PROP_EVALUATE_VISIBILITY);
            if (ZYGPU_SHAPE_RECTANGLE == prop.shape) {
                float t, u, v;
                if (!rectangleIntersect(ray, trafo, t, u, v)) continue;
                if (!evaluate || !materialVisibility<MASKS>(s, s.materials[s.material_ids[prop.parts_start]], ray.direction, trafo.r2, 0.5f * (u + 1.f),
                                                     0.5f * (v + 1.f), draw, tr))
                    return true;
            } else if (ZYGPU_SHAPE_SPHERE == prop.shape) {
                if (!evaluate) {
                    float t;
                    if (sphereIntersect(ray, trafo, t)) return true;
                    continue;
                }
                const float idl = 1.f / length3(ray.direction);
                const f3    nd = ray.direction * S3(idl);
                const f3    v = trafo.position - ray.origin;
                const float b = dot3(nd, v);
                const f3    remedy_term = v - S3(b) * nd;
                const float radius = 0.5f * trafo.scale.x;
                const float discriminant = radius * radius - dot3(remedy_term, remedy_term);
                if (discriminant > 0.f) {
                    const float          dist = sqrtf(discriminant);
                    const ZygpuMaterial& m = s.materials[s.material_ids[prop.parts_start]];
                    const float          ts[2] = {(b - dist) * idl, (b + dist) * idl};
                    for (int k = 0; k < 2; ++k) {
                        if (!(ts[k] >= ray.min_t && ray.max_t >= ts[k])) continue;
                        const f3 n = normalize3(ray.point(ts[k]) - trafo.position);
                        float    uu = 0.f, vv = 0.f;
                        if (MASKS && kNull != m.mask_image) sphereUv(trafo, n, uu, vv);
                        if (!materialVisibility<MASKS>(s, m, ray.direction, n, uu, vv, draw, tr)) return true;
                    }
                }
            } else if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {
                Ray          local = worldToObjectRay(trafo, ray);
                const DMesh& mesh = s.meshes[prop.mesh];
                if (!evaluate) {
                    if ((ZYG_FAST && !STATS && mesh.wide) ? meshIntersectP<false, true, true>(mesh, local, cnt) : meshIntersectP<STATS>(mesh, local, cnt)) return true;
                    continue;
                }
                const bool mesh_blocked = bvhWalk<STATS>(mesh.nodes, 1u, local, cnt, [&](uint32_t tfirst, uint32_t tcount) {
                    for (uint32_t k = tfirst, ke = tfirst + tcount; k < ke; ++k) {
                        if (STATS) cnt.tris += 1;
                        const f3 a = xyz(mesh.tri_verts[3 * k]), b = xyz(mesh.tri_verts[3 * k + 1]), c = xyz(mesh.tri_verts[3 * k + 2]);
                        float    t, u, v;
                        if (!triangleIntersect(local, a, b, c, t, u, v)) continue;
                        const ZygpuMaterial& m = s.materials[s.material_ids[prop.parts_start + mesh.triangle_parts[k]]];
                        if (!(m.flags & ZYGPU_MATF_EVALUATE_VISIBILITY)) return true;
                        float uv0 = 0.f, uv1 = 0.f;
                        if (MASKS && kNull != m.mask_image) meshInterpolateUv(mesh, k, u, v, uv0, uv1);
                        if (!materialVisibility<MASKS>(s, m, local.direction, normalize3(cross3(b - a, c - a)), uv0, uv1, draw, tr)) return true;
                    }
                    return false;
                });
                if (mesh_blocked) return true;
            } else if (ZYGPU_SHAPE_DISK == prop.shape) {  // the material decides (Material.visibility)
                float t, u, v;
                if (!diskIntersect(ray, trafo, t, u, v)) continue;
                if (!evaluate || !materialVisibility<MASKS>(s, s.materials[s.material_ids[prop.parts_start]], ray.direction, trafo.r2, 0.5f * (1.f + u),
                                                     0.5f * (1.f + v), draw, tr))
                    return true;
            } else if (ZYGPU_SHAPE_CUBE == prop.shape) {  // cube.zig:69-80: always blocks
                if (cubeIntersectP(ray, trafo)) return true;
            }
        }
        return false;
    });
    return !blocked;
}
template <bool STATS>
ZD bool sceneVisibilityTr(const DScene& s, Ray& ray, Counters& cnt, f3& tr) {
    NoDraw none;
    return sceneVisibilityTr<STATS, false>(s, ray, cnt, tr, none);
}

struct Fragment {  // shape/intersection.zig:64-124
    Trafo    trafo;
    float    t_hit, u, v;
    uint32_t primitive, prop, part;
    f3       p, geo_n, t, b, n;
    float    uv0, uv1;

    ZD bool sameHemisphere(f3 v_) const { return dot3(geo_n, v_) > 0.f; }
    ZD f3 offsetP(f3 v_) const {  // intersection.zig:115-119; uvw[3] (the offset) is 0 for rectangles and meshes
        const f3 nn = sameHemisphere(v_) ? geo_n : -geo_n;
        return offsetRay(fma3(S3(0.f), nn, p), nn);
    }
};

ZD void rectangleFragment(const Ray& ray, Fragment& f) {  // rectangle.zig:102-124
    const f3 p = ray.point(f.t_hit);
    const f3 n = f.trafo.r2, t = -f.trafo.r0, b = -f.trafo.r1;
    f.p = p;
    f.t = t;
    f.b = b;
    f.n = n;
    f.geo_n = n;
    if (f.trafo.scale.z < 0.f) {
        const f3    k = p - f.trafo.position;
        const float u = dot3(t, k) * 2.f, v = dot3(b, k) * 2.f;
        f.uv0 = 0.5f * (u + 1.f);
        f.uv1 = 0.5f * (v + 1.f);
    } else {
        f.uv0 = 0.5f * (f.u + 1.f);
        f.uv1 = 0.5f * (f.v + 1.f);
    }
    f.part = 0;
}

ZD void sphereFragment(const Ray& ray, Fragment& f) {  // sphere.zig:62-90
    const f3 p = ray.point(f.t_hit);
    const f3 n = normalize3(p - f.trafo.position);
    f.p = p;
    f.geo_n = n;
    f.n = n;
    f.part = 0;
    const f3    xyz_ = normalize3(f.trafo.worldToObjectNormal(n));
    const float phi = -atan2F(xyz_.x, xyz_.z) + kPi;
    const float theta = acosF(xyz_.y);
    const float sin_phi = sinF(phi), cos_phi = cosF(phi);
    const float sin_theta = mx(sinF(theta), 0.00001f);  // avoid the singularity at the poles
    const f3    t = normalize3(f.trafo.objectToWorldNormal(mk3(sin_theta * cos_phi, 0.f, sin_theta * sin_phi)));
    f.t = t;
    f.b = -cross3(t, n);
    f.uv0 = phi * (0.5f * kPiInv);
    f.uv1 = theta * kPiInv;
}

ZD f3 interpolate3(f3 a, f3 b, f3 c, float u, float v) {  // triangle.zig:141-149
    const float w = 1.f - u - v;
    return fma3(a, S3(w), fma3(b, S3(u), c * S3(v)));
}

ZD f3 meshPosition(const DMesh& m, uint32_t i) { return {m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]}; }
ZD f3 meshShadingNormal(const DMesh& m, uint32_t i) {
    return octDecode(snorm16ToFloat(m.normals[2 * i]), snorm16ToFloat(m.normals[2 * i + 1]));
}

ZD void meshFragment(const DMesh& mesh, Fragment& f) {  // triangle_mesh.zig:310-335 + triangle_data.zig:106-149
    const uint32_t prim = f.primitive;
    typedef uint32_t                                         Words __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) Words* record = (const __attribute__((address_space(1))) Words*)(mesh.shade + size_t(3) * prim);
    const Words                                              r0 = record[0], r1 = record[1], r2 = record[2];
    f.part = r0.w;
    f3 pa, pb, pc;
    triangleVertices(mesh.tri_verts, prim, pa, pb, pc);

    const f3 geo_n = normalize3(cross3(pb - pa, pc - pa));
    f.geo_n = f.trafo.objectToWorldNormal(geo_n);

    const f3    p = interpolate3(pa, pb, pc, f.u, f.v);
    const float uva0 = __uint_as_float(r1.x), uva1 = __uint_as_float(r1.y);
    const float uvb0 = __uint_as_float(r1.z), uvb1 = __uint_as_float(r1.w);
    const float uvc0 = __uint_as_float(r2.x), uvc1 = __uint_as_float(r2.y);
    const float w = 1.f - f.u - f.v;
    f.uv0 = __fmaf_rn(uva0, w, __fmaf_rn(uvb0, f.u, uvc0 * f.v));
    f.uv1 = __fmaf_rn(uva1, w, __fmaf_rn(uvb1, f.u, uvc1 * f.v));

    const f3 na = octDecode(snorm16ToFloat(r0.x & 0xFFFFu), snorm16ToFloat(r0.x >> 16));
    const f3 nb = octDecode(snorm16ToFloat(r0.y & 0xFFFFu), snorm16ToFloat(r0.y >> 16));
    const f3 nc = octDecode(snorm16ToFloat(r0.z & 0xFFFFu), snorm16ToFloat(r0.z >> 16));
    const f3 ni = normalize3(interpolate3(na, nb, nc, f.u, f.v));

    // triangle.zig:102-131 positionDifferentials
    const float duv02_0 = uva0 - uvc0, duv02_1 = uva1 - uvc1;
    const float duv12_0 = uvb0 - uvc0, duv12_1 = uvb1 - uvc1;
    const float determinant = duv02_0 * duv12_1 - duv02_1 * duv12_0;
    const f3    dp02 = pa - pc, dp12 = pb - pc;
    f3          dpdu, dpdv;
    if (0.f == fabsf(determinant)) {
        const f3 ng = normalize3(cross3(pc - pa, pb - pa));
        if (fabsf(ng.x) > fabsf(ng.y)) dpdu = mk3(-ng.z, 0.f, ng.x) / S3(sqrtf(ng.x * ng.x + ng.z * ng.z));
        else dpdu = mk3(0.f, ng.z, -ng.y) / S3(sqrtf(ng.y * ng.y + ng.z * ng.z));
        dpdv = cross3(ng, dpdu);
    } else {
        const float invdet = 1.f / determinant;
        dpdu = S3(invdet) * fma3(S3(duv12_1), dp02, S3(-duv02_1) * dp12);
        dpdv = S3(invdet) * fma3(S3(-duv12_0), dp02, S3(duv02_0) * dp12);
    }
    const f3 t = normalize3(gramSchmidt(dpdu, ni));
    const f3 b = normalize3(gramSchmidt(dpdv, ni));

    f.p = f.trafo.objectToWorldPoint(p);
    f.t = f.trafo.objectToWorldNormal(t);
    f.b = f.trafo.objectToWorldNormal(b);
    f.n = f.trafo.objectToWorldNormal(ni);
}

// Shape.fragment for a recorded hit (deterministic recomputation from the hit record + the traced ray)
ZD void diskFragment(const Ray& ray, Fragment& f) {  // disk.zig:100-116
    f.p = ray.point(f.t_hit);
    f.t = -f.trafo.r0;
    f.b = -f.trafo.r1;
    f.n = f.geo_n = f.trafo.r2;
    f.uv0 = 0.5f * (f.u + 1.f);
    f.uv1 = 0.5f * (f.v + 1.f);
    f.part = 0;
}
ZD void cubeFragment(const Ray& ray, Fragment& f) {  // cube.zig:41-61
    f.p = ray.point(f.t_hit);
    const f3    local_p = worldToObjectRay(f.trafo, ray).point(f.t_hit);
    const float d0 = fabsf(0.5f - fabsf(local_p.x)), d1 = fabsf(0.5f - fabsf(local_p.y)), d2 = fabsf(0.5f - fabsf(local_p.z));
    const int   i = d0 < d1 ? (d0 < d2 ? 0 : 2) : (d1 < d2 ? 1 : 2);  // indexMinComponent3
    const float c = pick(0 == i, local_p.x, pick(1 == i, local_p.y, local_p.z));
    // (component by component: a conditional between the three members themselves is an lvalue, i.e. a selected address, and a
    // Fragment that is addressed by a variable stays in scratch as a whole, in every kernel that knows this shape)
    const f3    r0 = f.trafo.r0, r1 = f.trafo.r1, r2 = f.trafo.r2;
    const f3    axis = mk3(pick(0 == i, r0.x, pick(1 == i, r1.x, r2.x)), pick(0 == i, r0.y, pick(1 == i, r1.y, r2.y)), pick(0 == i, r0.z, pick(1 == i, r1.z, r2.z)));
    const f3    n = S3(copysignf(1.f, c)) * axis;
    f.part = 0;
    f.geo_n = f.n = n;
    f.uv0 = f.uv1 = 0.f;
    const Frame fr = Frame::init(n);  // orthonormalBasis3
    f.t = fr.x;
    f.b = fr.y;
}

template <bool EXTRA, bool CALL = false>
ZD void makeFragment(const DScene& s, const Ray& ray, const Hit& hit, Fragment& f) {
    f.prop = hit.prop;
    f.t_hit = hit.t;
    f.u = hit.u;
    f.v = hit.v;
    f.primitive = hit.primitive;
    f.trafo = propTrafoAt<EXTRA, CALL>(s, hit.prop, ray.time);
    const ZygpuProp prop = s.props[hit.prop];
    if (ZYGPU_SHAPE_RECTANGLE == prop.shape) rectangleFragment(ray, f);
    else if (EXTRA && ZYGPU_SHAPE_SPHERE == prop.shape) sphereFragment(ray, f);
    else if (EXTRA && ZYGPU_SHAPE_DISK == prop.shape) diskFragment(ray, f);
    else if (EXTRA && ZYGPU_SHAPE_CUBE == prop.shape) cubeFragment(ray, f);
    else meshFragment(s.meshes[prop.mesh], f);
}

}  // namespace zd
