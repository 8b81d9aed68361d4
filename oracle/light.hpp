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

struct LightTreeView {
    const ZygpuSceneDesc* d;
    const ZygpuLightTree& t() const { return d->light_tree; }

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

        if (!(split && tr.num_infinite_lights < 64 - 1)) {
            ip = tr.infinite_weight;
            // random < infinite_guard never holds without infinite lights (guard == 0 or infinite_weight == 0)
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
        if (lo < tr.infinite_end) return split_infinite ? 1.f : 0.f;
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

// shape.zig:401-416 (finite shapes)
inline Ray shadowRay(V4 origin, const SampleTo& sample) {
    const V4    light_pos = offsetRay(sample.p, sample.n);
    const V4    shadow_axis = light_pos - origin;
    const float shadow_len = length3(shadow_axis);
    return Ray::init(origin, shadow_axis / S(shadow_len), 0.f, shadow_len);
}

// emittance.zig:29-59 with uniform emission map / no profile; trafo_r2 = rs.trafo.rotation.r[2]
inline V4 emittanceRadiance(const ZygpuMaterial& m, V4 wi, V4 trafo_r2, float area, bool in_camera) {
    if (-dot3(wi, trafo_r2) < m.cos_a) return S(0.f);
    const float factor = in_camera ? m.camera_weight : 1.f;
    const V4    intensity = mk(m.emission[0], m.emission[1], m.emission[2], 0.f) * S(1.f);
    if (m.emission_normalize) return S(factor / area) * intensity;
    return S(factor) * intensity;
}

}  // namespace orc
