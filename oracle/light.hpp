// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp).
//
// Light sampling restated from the reference:
//   light BVH traversal/pdf   src/core/scene/light/light_tree.zig:25-227 (Node), 346-517 (Tree)
//   Rectangle sampling        src/core/scene/shape/rectangle.zig:199-357 (SphQuad, sampleTo), 554-575 (pdf)
//   shadow ray                src/core/scene/shape/shape.zig:401-416
//   emitted radiance          src/core/scene/light/emittance.zig:29-59
//   shape sampler numSamples  src/core/scene/shape/shape_sampler.zig:35-41
#pragma once

#include "sampler.hpp"
#include "scene.hpp"
#include "texture.hpp"

namespace orc {

constexpr float kLowThreshold = 0.00000001f;  // integrator/helper.zig:29

struct LightPick {
    uint32_t offset;
    float    pdf;
};

struct SampleTo {  // shape/sample.zig:7-29
    V4    p, n, wi, uvw;
    float pdf() const { return p[3]; }
};

inline uint32_t numSamples(uint32_t num, float split_threshold) { return split_threshold <= kLowThreshold ? 1u : num; }

// ---------- light tree ----------

struct LightProps {
    V4    sphere, cone;
    float power;
    bool  two_sided;
};

inline LightProps lightProperties(const ZygpuSceneDesc& d, uint32_t l) {  // scene.zig:664-676
    const ZygpuAabb& box = d.light_aabbs[l];
    const V4         pos = S(0.5f) * (ld4(box.min) + ld4(box.max));
    LightProps       lp;
    lp.sphere = mk(pos[0], pos[1], pos[2], box.max[3]);
    lp.cone = ld4(d.light_cones + 4 * size_t(l));
    lp.power = box.min[3];
    lp.two_sided = 0 != d.lights[l].two_sided;
    return lp;
}

inline float clampedCosSub(float cos_a, float cos_b, float sin_a, float sin_b) {
    const float angle = std::fmaf(cos_a, cos_b, sin_a * sin_b);
    return cos_a > cos_b ? 1.f : angle;
}
inline float clampedSinSub(float cos_a, float cos_b, float sin_a, float sin_b) {
    const float angle = std::fmaf(sin_a, cos_b, -sin_b * cos_a);
    return cos_a > cos_b ? 0.f : angle;
}

// light_tree.zig:186-227
inline float importance(V4 p, V4 n, V4 center, V4 cone, float radius, float power, bool two_sided, bool total_sphere) {
    const V4    axis = p - center;
    const float l = length3(axis);
    const V4    na = axis / S(l);
    const V4    da = cone;

    const float sin_cu = mn(radius / l, 1.f);
    const float cos_cone = cone[3];
    const float dd = dot3(da, na);
    const float cos_a = two_sided ? std::fabs(dd) : dd;
    const float cos_n = mx(-dot3(n, na), 0.f);

    const V4 sa = mk(sin_cu, cos_cone, cos_a, cos_n);
    const V4 sb = max4(fma4(sa, -sa, S(1.f)), S(0.f));
    const V4 sr = sqrt4(sb);

    const float cos_cu = sr[0], sin_cone = sr[1], sin_a = sr[2], sin_n = sr[3];

    const float ta = clampedCosSub(cos_a, cos_cone, sin_a, sin_cone);
    const float tb = clampedSinSub(cos_a, cos_cone, sin_a, sin_cone);
    const float tc = clampedCosSub(ta, cos_cu, tb, sin_cu);
    const float tn = clampedCosSub(cos_n, cos_cu, sin_n, sin_cu);

    const float ra = total_sphere ? 1.f : tn;
    const float rb = mx(tc, 0.f);

    const float clamped_dist = mx(l, 0.5f * radius);
    const float rc = power / (clamped_dist * clamped_dist);

    return mx(ra * rb * rc, 0.f);
}

// The scene's tree over lights, or (ml set) the primitive tree over the emissive triangles of one mesh part
// (light_tree.zig:520-720: same traversal, split depth 6, no infinite lights, properties from shape_sampler.zig:199-223).
struct LightTreeView {
    const ZygpuSceneDesc* d;
    const ZygpuMeshLight* ml = nullptr;
    const ZygpuLightTree& t() const { return ml ? ml->tree : d->light_tree; }

    static bool     hasChildren(const ZygpuLightNode& n) { return 0 != (n.meta & 1u); }
    static bool     nodeTwoSided(const ZygpuLightNode& n) { return 0 != (n.meta & 2u); }
    static uint32_t childrenOrLight(const ZygpuLightNode& n) { return n.meta >> 2; }

    V4 decompressCenter(const ZygpuLightNode& n) const {  // light_tree.zig:37-40
        const V4 tt = mk(unorm16ToFloat(n.center[0]), unorm16ToFloat(n.center[1]), unorm16ToFloat(n.center[2]),
                         unorm16ToFloat(n.center[3]));
        return lerp4(ld4(t().bounds.min), ld4(t().bounds.max), tt);
    }

    float nodeWeight(const ZygpuLightNode& n, V4 p, V4 nrm, bool total_sphere) const {  // :54-62
        const V4 center = decompressCenter(n);
        const V4 cone = mk(snorm16ToFloat(n.cone[0]), snorm16ToFloat(n.cone[1]), snorm16ToFloat(n.cone[2]),
                           snorm16ToFloat(n.cone[3]));
        return importance(p, nrm, center, cone, center[3], n.power, nodeTwoSided(n), total_sphere);
    }

    bool nodeSplit(const ZygpuLightNode& n, V4 p, float threshold) const {  // :64-88
        const V4    center = decompressCenter(n);
        const float r = center[3];
        const float dd = mn(distance3(p, center), 1.0e6f);
        const float a = mx(dd - r, 0.001f);
        const float b = dd + r;
        const float eg = 1.f / (a * b);
        const float eg2 = eg * eg;
        const float a3 = a * a * a, b3 = b * b * b;
        const float e2g = (b3 - a3) / (3.f * (b - a) * a3 * b3);
        const float vg = e2g - eg2;
        const float ve = n.variance, ee = n.power;
        const float s2 = mx(ve * vg + ve * eg2 + ee * ee * vg, 0.f);
        const float ns = 1.f / (1.f + std::sqrt(s2));
        return ns < threshold;
    }

    float lightWeight(V4 p, V4 n, bool total_sphere, uint32_t light) const {  // :229-235
        if (ml) {
            const float* q = ml->tri_props + size_t(light) * 8;
            return importance(p, n, mk(q[0], q[1], q[2], q[3]), mk(q[4], q[5], q[6], 1.f), q[3], q[7], 0 != ml->two_sided, total_sphere);
        }
        const LightProps lp = lightProperties(*d, light);
        return importance(p, n, lp.sphere, lp.cone, lp.sphere[3], lp.power, lp.two_sided, total_sphere);
    }

    // Node.randomLight, light_tree.zig:90-147 (bidirectional CDF)
    LightPick nodeRandomLight(const ZygpuLightNode& node, V4 p, V4 n, bool total_sphere, float random) const {
        const uint32_t* light_mapping = t().light_mapping;
        const uint32_t  num_lights = node.num_lights;
        const uint32_t  light = childrenOrLight(node);
        if (1 == num_lights) return {light_mapping[light], 1.f};

        uint32_t front = light, back = light + num_lights - 1;
        float    w_front = lightWeight(p, n, total_sphere, light_mapping[front]);
        float    w_back = lightWeight(p, n, total_sphere, light_mapping[back]);
        float    w_sum_front = w_front, w_sum_back = w_back, w_sum = 0.f;

        while (front != back) {
            w_sum = w_sum_front + w_sum_back;
            if (w_sum_front <= random * w_sum) {
                front += 1;
                if (front != back) {
                    w_front = lightWeight(p, n, total_sphere, light_mapping[front]);
                    w_sum_front += w_front;
                } else {
                    w_front = w_back;
                }
            } else {
                back -= 1;
                if (front != back) {
                    w_back = lightWeight(p, n, total_sphere, light_mapping[back]);
                    w_sum_back += w_back;
                }
            }
        }
        if (0.f == w_sum) return {0, 0.f};
        return {light_mapping[front], w_front / w_sum};
    }

    // Node.pdf, light_tree.zig:149-183
    float nodePdf(const ZygpuLightNode& node, V4 p, V4 n, bool total_sphere, uint32_t id) const {
        const uint32_t num_lights = node.num_lights;
        if (1 == num_lights) return 1.f;
        const uint32_t light = childrenOrLight(node), end = light + num_lights;
        float          w_id = 0.f, sum = 0.f;
        for (uint32_t i = light; i < end; ++i) {
            const float lw = lightWeight(p, n, total_sphere, t().light_mapping[i]);
            sum += lw;
            if (id == i) w_id = lw;
        }
        if (0.f == sum) return 0.f;
        return w_id / sum;
    }

    // Tree.randomLight, light_tree.zig:346-447 (no infinite lights in scope: ip = 0, infinite_guard = 0)
    uint32_t randomLight(V4 p, V4 n, bool total_sphere, float random, float split_threshold, LightPick* buffer) const {
        const ZygpuLightTree& tr = t();
        uint32_t              current_light = 0;
        float                 ip = 0.f;
        const bool            split = split_threshold > 0.f;

        if (split && tr.num_infinite_lights < 64 - 1) {
            for (uint32_t i = 0; i < tr.num_infinite_lights; ++i) buffer[current_light++] = {tr.light_mapping[i], 1.f};
        } else {
            ip = tr.infinite_weight;
            if (random < tr.infinite_guard) {
                // Distribution1D.sampleDiscrete (distribution_1d.zig:49-59, 209-217; the lut only shortens the search)
                const uint32_t size = tr.num_infinite_lights + 1;
                uint32_t       i = 1;
                while (i < size - 1 && !(tr.infinite_cdf[i] >= random)) ++i;
                const uint32_t offset = i - 1;
                buffer[0] = {tr.light_mapping[offset], (tr.infinite_cdf[offset + 1] - tr.infinite_cdf[offset]) * ip};
                return 1;
            }
        }
        if (0 == tr.num_nodes) return current_light;

        const float    pd = 1.f - ip;
        const uint32_t max_split_depth = tr.max_split_depth;

        struct Value {
            float    pdf, random;
            uint32_t node, depth;
        };
        Value    stack[11 + 1];
        uint32_t end = 0;

        Value tv{pd, (random - ip) / pd, 0, split ? 0u : max_split_depth};
        stack[end++] = tv;

        while (0 != end) {
            const ZygpuLightNode& node = tr.nodes[tv.node];
            if (hasChildren(node)) {
                const bool     do_split = tv.depth < max_split_depth && nodeSplit(node, p, split_threshold);
                const uint32_t c0 = childrenOrLight(node), c1 = c0 + 1;
                if (do_split) {
                    tv.depth += 1;
                    tv.node = c0;
                    stack[end++] = Value{tv.pdf, tv.random, c1, tv.depth};
                } else {
                    tv.depth = max_split_depth;
                    float       p0 = nodeWeight(tr.nodes[c0], p, n, total_sphere);
                    float       p1 = nodeWeight(tr.nodes[c1], p, n, total_sphere);
                    const float pt = p0 + p1;
                    if (0.f == pt) {
                        tv = stack[--end];
                        continue;
                    }
                    p0 /= pt;
                    p1 /= pt;
                    if (tv.random < p0) {
                        tv.node = c0;
                        tv.pdf *= p0;
                        tv.random /= p0;
                    } else {
                        tv.node = c1;
                        tv.pdf *= p1;
                        tv.random = mn((tv.random - p0) / p1, 1.f);
                    }
                }
            } else {
                const LightPick pick = nodeRandomLight(node, p, n, total_sphere, tv.random);
                if (pick.pdf > 0.f) {
                    buffer[current_light] = {pick.offset, pick.pdf * tv.pdf};
                    current_light += 1;
                }
                tv = stack[--end];
            }
        }
        return current_light;
    }

    // Tree.pdf, light_tree.zig:449-517
    float pdf(V4 p, V4 n, bool total_sphere, float split_threshold, uint32_t id) const {
        const ZygpuLightTree& tr = t();
        const uint32_t        lo = tr.light_orders[id];
        const bool            split = split_threshold > 0.f;
        const bool            split_infinite = split && tr.num_infinite_lights < 64 - 1;
        if (lo < tr.infinite_end) return split_infinite ? 1.f : tr.infinite_weight * (tr.infinite_cdf[lo + 1] - tr.infinite_cdf[lo]);
        if (0 == tr.num_nodes) return 0.f;

        const float    ip = split_infinite ? 0.f : tr.infinite_weight;
        const uint32_t max_split_depth = tr.max_split_depth;
        float          pd = 1.f - ip;
        uint32_t       nid = 0;
        uint32_t       depth = split ? 0u : max_split_depth;
        for (;;) {
            const ZygpuLightNode& node = tr.nodes[nid];
            if (hasChildren(node)) {
                const bool     do_split = depth < max_split_depth && nodeSplit(node, p, split_threshold);
                const uint32_t c0 = childrenOrLight(node), c1 = c0 + 1;
                const uint32_t middle = tr.node_middles[nid];
                if (do_split) {
                    depth += 1;
                    nid = lo < middle ? c0 : c1;
                } else {
                    depth = max_split_depth;
                    const float p0 = nodeWeight(tr.nodes[c0], p, n, total_sphere);
                    const float p1 = nodeWeight(tr.nodes[c1], p, n, total_sphere);
                    const float pt = p0 + p1;
                    if (0.f == pt) return 0.f;
                    if (lo < middle) {
                        nid = c0;
                        pd *= p0 / pt;
                    } else {
                        nid = c1;
                        pd *= p1 / pt;
                    }
                }
            } else {
                return pd * nodePdf(node, p, n, total_sphere, lo);
            }
        }
    }
};

// ---------- Rectangle as a light ----------

struct SphQuad {  // rectangle.zig:199-303 (Ureña et al.)
    V4    o, x, y, z;
    float z0, x0, y0, x1, y1, b0, b1, k, Sa;

    static SphQuad init(V4 scale, V4 o) {
        const V4 s = mk(-0.5f * scale[0], -0.5f * scale[1], 0.f, 0.f);
        const V4 ex = mk(scale[0], 0.f, 0.f, 0.f);
        const V4 ey = mk(0.f, scale[1], 0.f, 0.f);
        SphQuad  q;
        q.o = o;
        const float exl = length3(ex), eyl = length3(ey);
        q.x = ex / S(exl);
        q.y = ey / S(eyl);
        q.z = cross3(q.x, q.y);
        const V4 d = s - o;
        q.z0 = dot3(d, q.z);
        if (q.z0 > 0.f) {
            q.z = -q.z;
            q.z0 = -q.z0;
        }
        q.x0 = dot3(d, q.x);
        q.y0 = dot3(d, q.y);
        q.x1 = q.x0 + exl;
        q.y1 = q.y0 + eyl;
        const V4 v00 = mk(q.x0, q.y0, q.z0), v01 = mk(q.x0, q.y1, q.z0), v10 = mk(q.x1, q.y0, q.z0), v11 = mk(q.x1, q.y1, q.z0);
        const V4 n0 = normalize3(cross3(v00, v10));
        const V4 n1 = normalize3(cross3(v10, v11));
        const V4 n2 = normalize3(cross3(v11, v01));
        const V4 n3 = normalize3(cross3(v01, v00));
        const float g0 = acosF(-dot3(n0, n1));
        const float g1 = acosF(-dot3(n1, n2));
        const float g2 = acosF(-dot3(n2, n3));
        const float g3 = acosF(-dot3(n3, n0));
        q.b0 = n0[2];
        q.b1 = n2[2];
        q.k = 2.f * kPi - g2 - g3;
        q.Sa = g0 + g1 - q.k;
        return q;
    }

    V4 sample(float u0, float u1) const {
        const float au = u0 * Sa + k;
        const float fu = (cosF(au) * b0 - b1) / sinF(au);
        float       cu = 1.f / std::sqrt(fu * fu + b0 * b0) * (fu > 0.f ? 1.f : -1.f);
        cu = stdClamp(cu, -1.f, 1.f);
        float xu = -(cu * z0) / std::sqrt(1.f - cu * cu);
        xu = stdClamp(xu, x0, x1);
        const float dd = std::sqrt(xu * xu + z0 * z0);
        const float h0 = y0 / std::sqrt(dd * dd + y0 * y0);
        const float h1 = y1 / std::sqrt(dd * dd + y1 * y1);
        const float hv = h0 + u1 * (h1 - h0);
        const float hv2 = hv * hv;
        uint32_t    eps_bits = 0x35800000u;
        float       eps;
        std::memcpy(&eps, &eps_bits, 4);
        const float yv = hv2 < 1.f - eps ? ((hv * dd) / std::sqrt(1.f - hv2)) : y1;
        return o + S(xu) * x + S(yv) * y + S(z0) * z;
    }

    float pdf(V4 scale) const {
        const V4    lp = o;
        const float sqr_dist = squaredLength3(lp);
        const float area = scale[0] * scale[1];
        const float numer = area * std::fabs(lp[2]);
        const float denom = sqr_dist * std::sqrt(sqr_dist);
        return numer > denom * kDotMin ? (1.f / Sa) : (denom / numer);
    }
};

// rectangle.zig:305-357 (UseSphericalSampling = true)
inline uint32_t rectangleSampleTo(V4 p, V4 n, const Trafo& trafo, bool two_sided, bool total_sphere, float split_threshold,
                                  uint32_t light_num_samples, Sampler& sampler, SampleTo* buffer) {
    const uint32_t num_samples = numSamples(light_num_samples, split_threshold);
    const float    nsf = float(num_samples);
    const V4       scale = trafo.scale();
    const V4       lp = trafo.worldToFramePoint(p);
    const SphQuad  squad = SphQuad::init(scale, lp);
    const float    sample_pdf = nsf * squad.pdf(scale);

    uint32_t current_sample = 0;
    for (uint32_t s = 0; s < num_samples; ++s) {
        float uv[2];
        sampler.sample2D(uv);
        const V4 ls = squad.sample(uv[0], uv[1]);
        const V4 ws = trafo.frameToWorldPoint(ls);
        const V4 dir = normalize3(ws - p);
        V4       wn = trafo.r[2];
        if (two_sided && dot3(wn, dir) > 0.f) wn = -wn;
        if (-dot3(wn, dir) < kDotMin || 0.f == squad.Sa || (dot3(dir, n) <= 0.f && !total_sphere)) continue;
        SampleTo& out = buffer[current_sample++];
        out.p = mk(ws[0], ws[1], ws[2], sample_pdf);
        out.n = wn;
        out.wi = dir;
        out.uvw = mk(uv[0], uv[1], 0.f, 0.f);
    }
    return current_sample;
}

// rectangle.zig:554-575 via shape.zig:469-492
inline float rectanglePdf(V4 p, const Trafo& frag_trafo, float split_threshold, uint32_t light_num_samples) {
    const float   nsf = float(numSamples(light_num_samples, split_threshold));
    const V4      scale = frag_trafo.scale();
    const V4      lp = frag_trafo.worldToFramePoint(p);
    const SphQuad squad = SphQuad::init(scale, lp);
    return nsf * squad.pdf(scale);
}

// ---------- Sphere lights: sphere.zig:318-400 (sampleTo, uniform cone of directions), 469-485 (pdf) ----------

inline float conePdfUniform(float one_minus_cos_theta_max) {  // sampling.zig:103-106
    return 1.f / ((2.f * kPi) * mx(one_minus_cos_theta_max, 1.0e-20f));
}

inline uint32_t sphereSampleTo(V4 p, V4 n, const Trafo& trafo, bool total_sphere, float split_threshold, uint32_t light_num_samples,
                               Sampler& sampler, SampleTo* buffer) {
    const V4    v = trafo.position - p;
    const float l = length3(v);
    const float r = 0.5f * trafo.scaleX();
    if (l <= (r + 0.0000001f)) return 0;
    const V4 z = S(1.f / l) * v;
    V4       fx, fy;
    orthonormalBasis3(z, fx, fy);  // Frame.init (frame.zig:10-13)
    const uint32_t num_samples = numSamples(light_num_samples, split_threshold);
    const float    nsf = float(num_samples);
    uint32_t       current_sample = 0;
    for (uint32_t i = 0; i < num_samples; ++i) {
        const float sin_theta_max = r / l;
        const float sin2_theta_max = sin_theta_max * sin_theta_max;
        const float cos_theta_max = std::sqrt(1.f - sin2_theta_max);
        float       one_minus_cos_theta_max = 1.f - cos_theta_max;
        float       s2[2];
        sampler.sample2D(s2);
        float cos_theta = (cos_theta_max - 1.f) * s2[0] + 1.f;
        float sin2_theta = 1.f - (cos_theta * cos_theta);
        if (sin2_theta_max < 0.00068523f) {
            sin2_theta = sin2_theta_max * s2[0];
            cos_theta = std::sqrt(1.f - sin2_theta);
            one_minus_cos_theta_max = 0.5f * sin2_theta_max;
        }
        const float cos_alpha = mn(sin2_theta / sin_theta_max + cos_theta * std::sqrt(1.f - mn(sin2_theta / sin2_theta_max, 1.f)), 1.f);
        const float sin_alpha = std::sqrt(1.f - cos_alpha * cos_alpha);
        const float phi = s2[1] * (2.f * kPi);
        const V4    w = mk(cosF(phi) * sin_alpha, sinF(phi) * sin_alpha, cos_alpha);  // sampling.zig:78-83 sphereDirection
        const V4    mw = -w;
        const V4    wn = fma4(S(mw[2]), z, fma4(S(mw[1]), fy, S(mw[0]) * fx));  // frame.frameToWorld (frame.zig:23-37)
        const V4    lp = trafo.position + S(r) * wn;
        const V4    dir = normalize3(lp - p);
        if (dot3(dir, n) <= 0.f && !total_sphere) continue;
        SampleTo& out = buffer[current_sample++];
        out.p = mk(lp[0], lp[1], lp[2], nsf * conePdfUniform(one_minus_cos_theta_max));
        out.n = wn;
        out.wi = dir;
        out.uvw = S(0.f);
    }
    return current_sample;
}

inline float spherePdf(V4 p, const Trafo& trafo, float split_threshold, uint32_t light_num_samples) {
    const V4    v = trafo.position - p;
    const float l2 = squaredLength3(v);
    const float r = 0.5f * trafo.scaleX();
    const float sin2_theta_max = (r * r) / l2;
    const float one_minus_cos_theta_max = sin2_theta_max < 0.00068523f ? 0.5f * sin2_theta_max : 1.f - std::sqrt(mx(1.f - sin2_theta_max, 0.f));
    return float(numSamples(light_num_samples, split_threshold)) * conePdfUniform(one_minus_cos_theta_max);
}

// ---------- Disk lights: equi-angular sampling of the chord through the projected shading point, then of the chord
// across it (disk.zig:180-251, 252-359, 492-533; UseEquiAngularSampling = true) ----------

struct DiskSamplerData {  // disk.zig:181-194
    V4 xd, yd;
    static DiskSamplerData init(V4 p) {
        const V4 td = mk(p[1], -p[0], 0.f, 0.f);
        const V4 xd = (0.f == td[0] && 0.f == td[1] && 0.f == td[2]) ? mk(1.f, 0.f, 0.f, 0.f) : normalize3(td);
        return {xd, mk(-xd[1], xd[0], 0.f, 0.f)};
    }
};
struct EquiAngularSampling {  // disk.zig:196-250
    float offset, min_t, max_t, scale, scale_sqr, angle_min, angle_extent;
    static EquiAngularSampling init(V4 source, V4 origin, V4 direction, float min_t, float max_t) {
        const float offset = dot3(direction, source - origin) / squaredLength3(direction);
        const float scale_sqr = squaredLength3((origin + S(offset) * direction) - source);
        const float scale = std::sqrt(scale_sqr);
        const float inv_scale = 0.f == scale ? 0.f : 1.f / scale;
        const float angle_min = atanF((min_t - offset) * inv_scale);
        const float angle_max = atanF((max_t - offset) * inv_scale);
        return {offset, min_t, max_t, scale, scale_sqr, angle_min, angle_max - angle_min};
    }
    float sample(float u, float& t) const {
        const float lt = scale * tanF(angle_min + u * angle_extent);
        const float p = scale / (angle_extent * (scale_sqr + lt * lt));
        t = clampf(lt + offset, min_t, max_t);
        return p;
    }
    float pdf(float t) const {
        if (min_t <= t && t < max_t) {
            const float lt = t - offset;
            return scale / (angle_extent * (scale_sqr + lt * lt));
        }
        return 0.f;
    }
    float pdfAndSample(float t, float& u) const {
        const float lt = t - offset;
        u = clampf((atanF(lt / scale) - angle_min) / angle_extent, 0.f, 1.f);  // math.saturate
        return scale / (angle_extent * (scale_sqr + lt * lt));
    }
};

inline uint32_t diskSampleTo(V4 p, V4 n, const Trafo& trafo, bool two_sided, bool total_sphere, float split_threshold,
                             uint32_t light_num_samples, Sampler& sampler, SampleTo* buffer) {
    const uint32_t num_samples = numSamples(light_num_samples, split_threshold);
    const float    nsf = float(num_samples);
    const float    radius = 0.5f * trafo.scaleX();
    const V4       lp = trafo.worldToFramePoint(p);
    const auto     dsd = DiskSamplerData::init(lp);
    const auto     eas0 = EquiAngularSampling::init(lp, S(0.f), dsd.yd, -radius, radius);
    if (0.f == eas0.angle_extent) return 0;
    uint32_t current_sample = 0;
    for (uint32_t i = 0; i < num_samples; ++i) {
        float r2[2];
        sampler.sample2D(r2);
        float x, y;
        diskConcentric(r2[0], r2[1], x, y);
        float u = x;
        float pdf_ = std::sqrt(1.f - u * u) / (0.25f * kPi);
        u = (u + 1.f) * 0.5f;
        float y_coord;
        pdf_ *= eas0.sample(u, y_coord);
        const float x_chord = std::sqrt(radius * radius - y_coord * y_coord);
        if (0.f == x_chord) continue;
        const auto eas1 = EquiAngularSampling::init(lp, S(y_coord) * dsd.yd, dsd.xd, -x_chord, x_chord);
        if (0.f == eas1.angle_extent) continue;
        float x_coord;
        pdf_ *= eas1.sample(sampler.sample1D(), x_coord);
        const V4 l_direction = S(x_coord) * dsd.xd + S(y_coord) * dsd.yd - lp;
        const V4 axis = trafo.objectToWorldNormal(l_direction);
        const V4 ws = p + axis;
        V4       wn = trafo.r[2];
        if (two_sided && dot3(wn, ws - p) > 0.f) wn = -wn;
        const float sl = squaredLength3(axis);
        const V4    dir = axis / S(std::sqrt(sl));
        const float c = -dot3(wn, dir);
        if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) continue;
        const float v = (y + 1.f) * 0.5f;
        SampleTo&   out = buffer[current_sample++];
        out.p = mk(ws[0], ws[1], ws[2], (nsf * pdf_ * sl) / c);
        out.n = wn;
        out.wi = dir;
        out.uvw = mk(u, v, 0.f, 0.f);
    }
    return current_sample;
}

inline float diskPdf(V4 dir, V4 p, const Trafo& trafo, V4 frag_p, float split_threshold, uint32_t light_num_samples) {
    const float c = std::fabs(dot3(trafo.r[2], dir));
    const float nsf = float(numSamples(light_num_samples, split_threshold));
    const float radius = 0.5f * trafo.scaleX();
    const float sl = squaredDistance3(p, frag_p);
    const V4    lp = trafo.worldToFramePoint(p);
    const auto  dsd = DiskSamplerData::init(lp);
    const auto  eas0 = EquiAngularSampling::init(lp, S(0.f), dsd.yd, -radius, radius);
    const V4    l_point = trafo.worldToFramePoint(frag_p);
    const float y_coord = dot3(l_point, dsd.yd);
    float       u;
    const float eas_pdf = eas0.pdfAndSample(y_coord, u);
    u = u * 2.f - 1.f;
    float pdf_ = std::sqrt(1.f - u * u) / (0.25f * kPi);
    pdf_ *= eas_pdf;
    const float x_chord = std::sqrt(radius * radius - y_coord * y_coord);
    const auto  eas1 = EquiAngularSampling::init(lp, S(y_coord) * dsd.yd, dsd.xd, -x_chord, x_chord);
    const float x_coord = dot3(l_point, dsd.xd);
    pdf_ *= eas1.pdf(x_coord);
    return (nsf * pdf_ * sl) / c;
}

// ---------- Dome / Canopy with uniform emission: dome.zig:26-103, 214-220; canopy.zig:24-92 ----------

inline V4 hemisphereUniform(float u0, float u1) {  // sampling.zig:56-65
    const float z = 1.f - u0;
    const float r = std::sqrt(mx(0.f, 1.f - z * z));
    const float phi = u1 * (2.f * kPi);
    return mk(cosF(phi) * r, sinF(phi) * r, z);
}
inline V4 sphereUniform(float u0, float u1) {  // sampling.zig:67-76
    const float z = 1.f - 2.f * u0;
    const float r = std::sqrt(mx(0.f, 1.f - z * z));
    const float phi = u1 * (2.f * kPi);
    return mk(cosF(phi) * r, sinF(phi) * r, z);
}
constexpr float kCanopyEps = -0.0005f;  // canopy.zig:25

inline uint32_t domeSampleTo(V4 n, bool total_sphere, Sampler& sampler, SampleTo* buffer) {
    float uv[2];
    sampler.sample2D(uv);
    V4    dir;
    float pdf;
    if (total_sphere) {
        dir = sphereUniform(uv[0], uv[1]);
        pdf = 1.f / (4.f * kPi);
    } else {
        const V4 dl = hemisphereUniform(uv[0], uv[1]);
        V4       fx, fy;
        orthonormalBasis3(n, fx, fy);
        dir = fma4(S(dl[2]), n, fma4(S(dl[1]), fy, S(dl[0]) * fx));
        if (dot3(dir, n) <= 0.f) return 0;
        pdf = 1.f / (2.f * kPi);
    }
    const V4 p = S(kRayMaxT) * dir;
    buffer[0].p = mk(p[0], p[1], p[2], pdf);
    buffer[0].n = -dir;
    buffer[0].wi = dir;
    buffer[0].uvw = S(0.f);  // the uv only addresses emission maps, which are not in scope
    return 1;
}

inline uint32_t canopySampleTo(V4 n, const Trafo& trafo, bool total_sphere, Sampler& sampler, SampleTo* buffer) {
    float uv[2];
    sampler.sample2D(uv);
    const V4 dl = hemisphereUniform(uv[0], uv[1]);
    const V4 dir = fma4(S(dl[2]), trafo.r[2], fma4(S(dl[1]), trafo.r[1], S(dl[0]) * trafo.r[0]));
    if (dot3(dir, n) <= 0.f && !total_sphere) return 0;
    const V4 p = S(kRayMaxT) * dir;
    buffer[0].p = mk(p[0], p[1], p[2], 1.f / (2.f * kPi));
    buffer[0].n = -dir;
    buffer[0].wi = dir;
    buffer[0].uvw = S(0.f);
    return 1;
}

// ---------- Dome / Canopy with an emission map: dome.zig:40-56, 105-158, 222-234; canopy.zig:44-58, 94-133, 175-211 ----------

inline void domeUv(V4 dir, const Trafo& trafo, float& u, float& v) {  // Dome.fragment's uvw
    const V4 xyz = normalize3(trafo.transformVectorTransposed(dir));
    u = atan2F(xyz[0], xyz[2]) * (kPiInv * 0.5f) + 0.5f;
    v = acosF(xyz[1]) * kPiInv;
}
inline void hemisphereToDiskEquidistant(V4 dir, float& dx, float& dy) {
    const float colatitude = acosF(dir[2]);
    const float longitude = atan2F(-dir[1], dir[0]);
    const float r = colatitude * (kPiInv * 2.f);
    dx = r * cosF(longitude);
    dy = r * sinF(longitude);
}
inline V4 diskToHemisphereEquidistant(float x, float y) {
    const float longitude = atan2F(-y, x);
    const float r = std::sqrt(x * x + y * y);
    const float colatitude = r * (kPi / 2.f);
    const float sin_col = sinF(colatitude), cos_col = cosF(colatitude);
    const float sin_lon = sinF(longitude), cos_lon = cosF(longitude);
    return mk(sin_col * cos_lon, sin_col * sin_lon, cos_col);
}
inline void canopyUv(V4 dir, const Trafo& trafo, float& u, float& v) {  // Canopy.fragment's uvw
    const V4 xyz = normalize3(trafo.transformVectorTransposed(dir));
    float    dx, dy;
    hemisphereToDiskEquidistant(xyz, dx, dy);
    u = 0.5f * dx + 0.5f;
    v = 0.5f * dy + 0.5f;
}

inline uint32_t domeSampleMaterialTo(const ZygpuImageLight& il, V4 n, const Trafo& trafo, bool total_sphere, float split_threshold,
                                     uint32_t light_samples, Sampler& sampler, SampleTo* buffer) {
    const uint32_t num_samples = numSamples(light_samples, split_threshold);
    const float    nsf = float(num_samples);
    uint32_t       current = 0;
    for (uint32_t i = 0; i < num_samples; ++i) {
        float r2[2];
        sampler.sample2D(r2);
        float       u, v;
        const float pdf = imageLightSample(il, r2[0], r2[1], u, v);
        if (0.f == pdf) continue;
        const float phi = (u - 0.5f) * (2.f * kPi), theta = v * kPi;
        const float sin_phi = sinF(phi), cos_phi = cosF(phi), sin_theta = sinF(theta), cos_theta = cosF(theta);
        const V4    dir = trafo.transformVector(mk(sin_phi * sin_theta, cos_theta, cos_phi * sin_theta));
        if (dot3(dir, n) <= 0.f && !total_sphere) continue;
        const V4 p = S(kRayMaxT) * dir;
        buffer[current].p = mk(p[0], p[1], p[2], (nsf * pdf) / ((4.f * kPi) * sin_theta));
        buffer[current].n = -dir;
        buffer[current].wi = dir;
        buffer[current].uvw = mk(u, v, 0.f, 0.f);
        current += 1;
    }
    return current;
}
inline float domeMaterialPdf(const ZygpuImageLight& il, float u, float v, float split_threshold, uint32_t light_samples) {
    const float sin_theta = sinF(v * kPi);
    if (0.f == sin_theta) return 0.f;
    const uint32_t num_samples = numSamples(light_samples, split_threshold);
    const float    material_pdf = imageLightPdf(il, u, v) * float(num_samples);
    return material_pdf / ((4.f * kPi) * sin_theta);
}

inline uint32_t canopySampleMaterialTo(const ZygpuImageLight& il, V4 n, const Trafo& trafo, bool total_sphere, Sampler& sampler,
                                       SampleTo* buffer) {
    float r2[2];
    sampler.sample2D(r2);
    float       u, v;
    const float pdf = imageLightSample(il, r2[0], r2[1], u, v);
    if (0.f == pdf) return 0;
    const float dx = 2.f * u - 1.f, dy = 2.f * v - 1.f;
    if (dx * dx + dy * dy > 1.f) return 0;
    const V4 dir = trafo.transformVector(diskToHemisphereEquidistant(dx, dy));
    if (dot3(dir, n) <= 0.f && !total_sphere) return 0;
    const V4 p = S(kRayMaxT) * dir;
    buffer[0].p = mk(p[0], p[1], p[2], pdf / (2.f * kPi));
    buffer[0].n = -dir;
    buffer[0].wi = dir;
    buffer[0].uvw = mk(u, v, 0.f, 0.f);
    return 1;
}

// ---------- Rectangle with an emission map: rectangle.zig:399-456, 577-595 ----------
inline uint32_t rectangleSampleMaterialTo(const ZygpuImageLight& il, V4 p, V4 n, const Trafo& trafo, bool two_sided, bool total_sphere,
                                          float split_threshold, uint32_t light_samples, Sampler& sampler, SampleTo* buffer) {
    const uint32_t num_samples = numSamples(light_samples, split_threshold);
    const float    nsf = float(num_samples);
    const V4       scale = trafo.scale();
    const float    area = scale[0] * scale[1];
    uint32_t       current = 0;
    for (uint32_t i = 0; i < num_samples; ++i) {
        float r2[2];
        sampler.sample2D(r2);
        float       u, v;
        const float pdf = imageLightSample(il, r2[0], r2[1], u, v);
        if (0.f == pdf) continue;
        const V4 ls = mk(-1.f * u + 0.5f, -1.f * v + 0.5f, 0.f, 0.f);
        const V4 ws = trafo.objectToWorldPoint(ls);
        const V4 axis = ws - p;
        V4       wn = trafo.r[2];
        if (two_sided && dot3(wn, axis) > 0.f) wn = -wn;
        const float sl = squaredLength3(axis);
        const float t = std::sqrt(sl);
        const V4    dir = axis / S(t);
        const float c = -dot3(wn, dir);
        if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) continue;
        buffer[current].p = mk(ws[0], ws[1], ws[2], (nsf * pdf * sl) / (c * area));
        buffer[current].n = wn;
        buffer[current].wi = dir;
        buffer[current].uvw = mk(u, v, 0.f, 0.f);
        current += 1;
    }
    return current;
}
inline float rectangleMaterialPdf(const ZygpuImageLight& il, V4 dir, V4 p, const Trafo& trafo, V4 frag_p, float u, float v,
                                  float split_threshold, uint32_t light_samples) {
    const float c = std::fabs(dot3(trafo.r[2], dir));
    const V4    scale = trafo.scale();
    const float area = scale[0] * scale[1];
    const V4    dp = p - frag_p;
    const float sl = dot3(dp, dp);
    const uint32_t num_samples = numSamples(light_samples, split_threshold);
    const float    material_pdf = imageLightPdf(il, u, v) * float(num_samples);
    return (material_pdf * sl) / (c * area);
}

// Disk.sampleMaterialTo (disk.zig:361-426) / materialPdf (535-553): area sampling by the emission map over the unit square
// around the disc (uvWeight leaves the corners empty, disk.zig:428-436)
inline uint32_t diskSampleMaterialTo(const ZygpuImageLight& il, V4 p, V4 n, const Trafo& trafo, bool two_sided, bool total_sphere,
                                     float split_threshold, uint32_t light_samples, Sampler& sampler, SampleTo* buffer) {
    const uint32_t num_samples = numSamples(light_samples, split_threshold);
    const float    nsf = float(num_samples);
    const float    radius = 0.5f * trafo.scaleX();
    const float    area = kPi * (radius * radius);
    uint32_t       current = 0;
    for (uint32_t i = 0; i < num_samples; ++i) {
        float r2[2];
        sampler.sample2D(r2);
        float       u, v;
        const float pdf = imageLightSample(il, r2[0], r2[1], u, v);
        if (0.f == pdf) continue;
        const V4    ls = mk(-2.f * u + 1.f, -2.f * v + 1.f, 0.f, 0.f);
        const V4    k = S(radius) * trafo.objectToWorldNormal(ls);
        const float l = dot3(k, k);
        if (l > radius * radius) continue;
        const V4 ws = trafo.position + k;
        V4       wn = trafo.r[2];
        if (two_sided && dot3(wn, ws - p) > 0.f) wn = -wn;
        const V4    axis = ws - p;
        const float sl = squaredLength3(axis);
        const float t = std::sqrt(sl);
        const V4    dir = axis / S(t);
        const float c = -dot3(wn, dir);
        if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) continue;
        buffer[current].p = mk(ws[0], ws[1], ws[2], (nsf * pdf * sl) / (c * area));
        buffer[current].n = wn;
        buffer[current].wi = dir;
        buffer[current].uvw = mk(u, v, 0.f, 0.f);
        current += 1;
    }
    return current;
}
inline float diskMaterialPdf(const ZygpuImageLight& il, V4 dir, V4 p, const Trafo& trafo, V4 frag_p, float u, float v, float split_threshold,
                             uint32_t light_samples) {
    const float    c = std::fabs(dot3(trafo.r[2], dir));
    const float    radius = 0.5f * trafo.scaleX();
    const float    area = kPi * (radius * radius);
    const V4       dp = p - frag_p;
    const float    sl = dot3(dp, dp);
    const uint32_t num_samples = numSamples(light_samples, split_threshold);
    const float    material_pdf = imageLightPdf(il, u, v) * float(num_samples);
    return (material_pdf * sl) / (c * area);
}

// Sphere.sampleMaterialTo (sphere.zig:395-446) / materialPdf (489-502): one sample of the lat-long emission map
inline uint32_t sphereSampleMaterialTo(const ZygpuImageLight& il, V4 p, V4 n, const Trafo& trafo, bool total_sphere, Sampler& sampler,
                                       SampleTo* buffer) {
    float r2[2];
    sampler.sample2D(r2);
    float       u, v;
    const float pdf = imageLightSample(il, r2[0], r2[1], u, v);
    if (0.f == pdf) return 0;
    const float phi = (u + 0.75f) * (2.f * kPi);
    const float theta = v * kPi;
    const float sin_theta = mx(sinF(theta), 0.00001f);  // avoid singularity at poles
    const float cos_theta = cosF(theta);
    const float sin_phi = sinF(phi), cos_phi = cosF(phi);
    const float r = 0.5f * trafo.scaleX();
    const V4    ls = S(r) * mk(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi, 0.f);
    const V4    ws = trafo.objectToWorldPoint(ls);
    const V4    axis = ws - p;
    const float sl = squaredLength3(axis);
    const float t = std::sqrt(sl);
    const V4    dir = axis / S(t);
    const V4    wn = normalize3(ws - trafo.position);
    const float c = -dot3(wn, dir);
    if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) return 0;
    const float area = (4.f * kPi) * (r * r);
    buffer[0].p = mk(ws[0], ws[1], ws[2], (pdf * sl) / (c * area * sin_theta));
    buffer[0].n = wn;
    buffer[0].wi = dir;
    buffer[0].uvw = mk(u, v, 0.f, 0.f);
    return 1;
}
inline float sphereMaterialPdf(const ZygpuImageLight& il, V4 dir, V4 p, const Trafo& trafo, V4 frag_p, V4 frag_geo_n, float u, float v) {
    const float sin_theta = mx(sinF(v * kPi), 0.00001f);
    const V4    dp = p - frag_p;
    const float sl = dot3(dp, dp);
    const float c = -dot3(frag_geo_n, dir);
    const float r = 0.5f * trafo.scaleX();
    const float area = (4.f * kPi) * (r * r);
    return (imageLightPdf(il, u, v) * sl) / (c * area * sin_theta);
}

// ---------- Distant lights: distant.zig:21-146 ----------

inline float distantSolidAngle(float radius) { return (2.f * kPi) * (1.f - std::sqrt(1.f / (radius * radius + 1.f))); }

inline uint32_t distantSampleTo(V4 n, const Trafo& trafo, bool total_sphere, Sampler& sampler, SampleTo* buffer) {
    const float radius = trafo.scaleX();
    if (radius <= 0.f) return 0;
    float r2[2];
    sampler.sample2D(r2);
    float x, y;
    diskConcentric(r2[0], r2[1], x, y);
    const V4 ws = S(radius) * trafo.transformVector(mk(x, y, 0.f));
    const V4 dir = normalize3(ws - trafo.r[2]);
    if (dot3(dir, n) <= 0.f && !total_sphere) return 0;
    const V4 p = S(kRayMaxT) * dir;
    buffer[0].p = mk(p[0], p[1], p[2], 1.f / distantSolidAngle(radius));
    buffer[0].n = trafo.r[2];
    buffer[0].wi = dir;
    buffer[0].uvw = S(0.f);
    return 1;
}

// Distant.intersect (distant.zig:22-52): only rays that left the scene (max_t == RayMaxT)
inline bool distantIntersect(const Ray& ray, const Trafo& trafo, float& u, float& v) {
    const float radius = trafo.scaleX();
    const V4    n = trafo.r[2];
    const float b = dot3(n, ray.direction);
    if (b > 0.f || ray.max_t < kRayMaxT || radius <= 0.f) return false;
    const float det = (b * b) - dot3(n, n) + (radius * radius);
    if (det >= 0.f) {
        const V4 k = ray.direction - n;
        const V4 sk = k / S(radius);
        u = dot3(trafo.r[0], sk);
        v = dot3(trafo.r[1], sk);
        return true;
    }
    return false;
}

// ---------- TriangleMesh lights: triangle_mesh.zig:391-703 ----------

inline V4 orthogonalize(V4 a, V4 b) { return normalize3(fma4(S(-dot3(a, b)), a, b)); }  // a is normalised (391-394)

struct SphericalSample {
    V4    dir;
    float u, v, pdf;
};

// triangle.zig:82-100
inline void barycentricCoords(V4 dir, V4 a, V4 b, V4 c, float& u, float& v) {
    const V4    e1 = b - a, e2 = c - a, tvec = -a;
    const V4    pvec = cross3(dir, e2), qvec = cross3(tvec, e1);
    const float e1_d_pv = dot3(e1, pvec), tv_d_pv = dot3(tvec, pvec), di_d_qv = dot3(dir, qvec);
    const float inv_det = 1.f / e1_d_pv;
    u = tv_d_pv * inv_det;
    v = di_d_qv * inv_det;
}

// Stratified Sampling of Spherical Triangles, James Arvo (triangle_mesh.zig:402-459)
inline bool sampleSpherical(V4 pos, V4 pa, V4 pb, V4 pc, float r0, float r1, SphericalSample& out) {
    const V4 pap = pa - pos, pbp = pb - pos, pcp = pc - pos;
    const V4 A = normalize3(pap), B = normalize3(pbp), C = normalize3(pcp);
    const V4 BA = orthogonalize(A, B - A), CA = orthogonalize(A, C - A);
    const V4 AB = orthogonalize(B, A - B), CB = orthogonalize(B, C - B);
    const V4 BC = orthogonalize(C, B - C), AC = orthogonalize(C, A - C);
    const float cos_alpha = clampf(dot3(BA, CA), -1.f, 1.f);
    const float alpha = acosF(cos_alpha);
    const float beta = acosF(clampf(dot3(AB, CB), -1.f, 1.f));
    const float gamma = acosF(clampf(dot3(BC, AC), -1.f, 1.f));
    const float sarea = alpha + beta + gamma - kPi;
    if (0.f == sarea) return false;
    const float cos_c = clampf(dot3(A, B), -1.f, 1.f);
    const float area_S = r0 * sarea;
    const float angle_delta = area_S - alpha;
    const float p = sinF(angle_delta), q = cosF(angle_delta);
    const float sin_alpha = std::sqrt(1.f - cos_alpha * cos_alpha);
    const float u = q - cos_alpha, v = p + sin_alpha * cos_c;
    const float s = clampf(((v * q - u * p) * cos_alpha - v) / ((v * p + u * q) * sin_alpha), -1.f, 1.f);
    const V4    C_s = S(s) * A + S(std::sqrt(1.f - s * s)) * orthogonalize(A, C);
    const float cs_b = dot3(C_s, B);
    const float z = 1.f - r1 * (1.f - cs_b);
    const V4    P = S(z) * B + S(std::sqrt(1.f - z * z)) * orthogonalize(B, C_s);
    out.dir = P;
    barycentricCoords(P, pap, pbp, pcp, out.u, out.v);
    out.pdf = 1.f / sarea;
    return true;
}

inline float pdfSpherical(V4 pos, V4 pa, V4 pb, V4 pc) {  // triangle_mesh.zig:461-485
    const V4 A = normalize3(pa - pos), B = normalize3(pb - pos), C = normalize3(pc - pos);
    const V4 BA = orthogonalize(A, B - A), CA = orthogonalize(A, C - A);
    const V4 AB = orthogonalize(B, A - B), CB = orthogonalize(B, C - B);
    const V4 BC = orthogonalize(C, B - C), AC = orthogonalize(C, A - C);
    const float alpha = acosF(clampf(dot3(BA, CA), -1.f, 1.f));
    const float beta = acosF(clampf(dot3(AB, CB), -1.f, 1.f));
    const float gamma = acosF(clampf(dot3(BC, AC), -1.f, 1.f));
    const float sarea = alpha + beta + gamma - kPi;
    return 1.f / sarea;
}

inline void triangleUniform(float u0, float u1, float& x, float& y) {  // sampling.zig:39-47 (Heitz)
    if (u1 > u0) {
        x = 0.5f * u0;
        y = u1 - x;
    } else {
        y = 0.5f * u1;
        x = u0 - y;
    }
}

constexpr float kAreaDistanceRatio = 0.001f;  // triangle_mesh.zig:487

// Mesh.sampleTo (triangle_mesh.zig:489-597): the part's primitive tree picks triangles (one 1-D draw), every picked
// triangle takes one 2-D draw: solid-angle sampling when the triangle is large for its distance, area sampling else
inline uint32_t meshSampleTo(const ZygpuSceneDesc& d, const ZygpuMeshLight& ml, V4 p, V4 n, const Trafo& trafo, bool two_sided,
                             bool total_sphere, float split_threshold, Sampler& sampler, SampleTo* buffer) {
    const ZygpuMesh& mesh = d.meshes[ml.mesh];
    const MeshView   mv{&mesh};
    const V4         op = trafo.worldToObjectPoint(p), on = trafo.worldToObjectNormal(n);
    const V4         scale = trafo.scale();
    const V4         scale_squared = scale * scale;

    LightPick           picks[64];
    const LightTreeView tree{&d, &ml};
    const uint32_t      num_picks = tree.randomLight(op, on, total_sphere, sampler.sample1D(), split_threshold, picks);

    uint32_t current_sample = 0;
    for (uint32_t i = 0; i < num_picks; ++i) {
        const LightPick& s = picks[i];
        const uint32_t   global = ml.triangle_mapping[s.offset];
        const uint32_t*  tri = mesh.triangles + size_t(global) * 3;
        const V4         a = mv.position(tri[0]), b = mv.position(tri[1]), c = mv.position(tri[2]);
        const V4         e1 = b - a, e2 = c - a;
        const V4         cross_axis = cross3(e1, e2);
        const V4         ca = scale_squared * cross_axis;
        const float      lca = length3(ca);
        const V4         sn = ca / S(lca);
        V4               wn = trafo.objectToWorldNormal(sn);
        const float      tri_area = 0.5f * lca;
        const V4         center = (a + b + c) / S(3.f);

        V4    dir, v;
        float bu, bv, sample_pdf, n_dot_dir;
        float r2[2];
        if (tri_area / distance3(center, op) > kAreaDistanceRatio) {
            sampler.sample2D(r2);
            SphericalSample ss;
            if (!sampleSpherical(op, a, b, c, r2[0], r2[1], ss)) continue;
            if (dot3(ss.dir, on) <= 0.f && !total_sphere) continue;
            bu = ss.u;
            bv = ss.v;
            dir = trafo.objectToWorldNormal(ss.dir);
            v = trafo.objectToWorldPoint(interpolate3(a, b, c, bu, bv));
            sample_pdf = s.pdf * ss.pdf;
            if (two_sided && dot3(wn, dir) > 0.f) wn = -wn;
            n_dot_dir = -dot3(wn, dir);
        } else {
            sampler.sample2D(r2);
            triangleUniform(r2[0], r2[1], bu, bv);
            v = trafo.objectToWorldPoint(interpolate3(a, b, c, bu, bv));
            const V4    axis = v - p;
            const float sl = squaredLength3(axis);
            const float dd = std::sqrt(sl);
            dir = axis / S(dd);
            if (dot3(dir, n) <= 0.f && !total_sphere) continue;
            if (two_sided && dot3(wn, dir) > 0.f) wn = -wn;
            n_dot_dir = -dot3(wn, dir);
            sample_pdf = (s.pdf * sl) / (n_dot_dir * tri_area);
        }
        if (n_dot_dir < kDotMin) continue;

        float tc[2];
        interpolate2(mesh.uvs + size_t(tri[0]) * 2, mesh.uvs + size_t(tri[1]) * 2, mesh.uvs + size_t(tri[2]) * 2, bu, bv, tc);
        SampleTo& out = buffer[current_sample++];
        out.p = mk(v[0], v[1], v[2], sample_pdf);
        out.n = wn;
        out.wi = dir;
        out.uvw = mk(tc[0], tc[1], 0.f, 0.f);
    }
    return current_sample;
}

// Mesh.pdf (triangle_mesh.zig:647-691): the density sampleTo would have had for the triangle a ray hit
inline float meshPdf(const ZygpuSceneDesc& d, const ZygpuMeshLight& ml, V4 dir, V4 p, V4 n, const Fragment& frag, bool total_sphere,
                     float split_threshold) {
    const ZygpuMesh& mesh = d.meshes[ml.mesh];
    const MeshView   mv{&mesh};
    const Trafo&     trafo = frag.isec.trafo;
    const float      n_dot_dir = std::fabs(dot3(frag.geo_n, dir));
    const V4         op = trafo.worldToObjectPoint(p), on = trafo.worldToObjectNormal(n);
    const uint32_t   pm = mesh.primitive_mapping[frag.isec.primitive];
    const float      tri_pdf = LightTreeView{&d, &ml}.pdf(op, on, total_sphere, split_threshold, pm);

    const uint32_t* tri = mesh.triangles + size_t(frag.isec.primitive) * 3;
    const V4        a = mv.position(tri[0]), b = mv.position(tri[1]), c = mv.position(tri[2]);
    const V4        cross_axis = cross3(b - a, c - a);
    const V4        scale = trafo.scale();
    const V4        ca = (scale * scale) * cross_axis;
    const float     tri_area = 0.5f * length3(ca);
    const V4        center = (a + b + c) / S(3.f);
    if (tri_area / distance3(center, op) > kAreaDistanceRatio) return tri_pdf * pdfSpherical(op, a, b, c);
    const float sl = squaredDistance3(p, frag.p);
    return (tri_pdf * sl) / (n_dot_dir * tri_area);
}

// shape.zig:401-416
inline Ray shadowRay(V4 origin, const SampleTo& sample, bool infinite_shape = false) {
    if (infinite_shape) return Ray::init(origin, sample.wi, 0.f, kRayMaxT);
    const V4    light_pos = offsetRay(sample.p, sample.n);
    const V4    shadow_axis = light_pos - origin;
    const float shadow_len = length3(shadow_axis);
    return Ray::init(origin, shadow_axis / S(shadow_len), 0.f, shadow_len);
}

// emittance.zig:29-59; (u, v) = rs.uv(), r = rs.stochastic_r for an emission map and a profile; origin = rs.origin
inline V4 emittanceRadiance(const ZygpuSceneDesc& scene, const ZygpuMaterial& m, V4 wi, const Trafo& trafo, float area, bool in_camera, float u,
                            float v, float r, V4 origin) {
    if (-dot3(wi, trafo.r[2]) < m.cos_a) return S(0.f);
    float factor = in_camera ? m.camera_weight : 1.f;
    if (kNull != m.profile_image) {  // emittance.zig:45-52: the profile at octEncode(-normalize(worldToObjectPoint(rs.origin)))
        const V4    lwi = -normalize3(trafo.worldToObjectPoint(origin));
        const float inorm = 1.f / (std::fabs(lwi[0]) + std::fabs(lwi[1]) + std::fabs(lwi[2]));  // base/encoding.zig:82-87
        const float t = mx(lwi[2], 0.f);
        const float ox = (lwi[0] + (lwi[0] > 0.f ? t : -t)) * inorm, oy = (lwi[1] + (lwi[1] > 0.f ? t : -t)) * inorm;
        factor *= sampleImageMap1(scene, m.profile_image, m.profile_sampler, m.profile_scale, (ox + 1.f) * 0.5f, (oy + 1.f) * 0.5f, r);
    }
    const V4    texel = kNull != m.emission_image ? sampleImageMap(scene, m.emission_image, m.emission_sampler, m.emission_scale, u, v, r)
                                                  : S(1.f);
    const V4    intensity = mk(m.emission[0], m.emission[1], m.emission[2], 0.f) * texel;
    if (m.emission_normalize) return S(factor / area) * intensity;
    return S(factor) * intensity;
}

}  // namespace orc
