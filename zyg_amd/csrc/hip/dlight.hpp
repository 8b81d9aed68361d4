// Device light sampling.  Behaviour follows the reference:
//   light BVH (adaptive splitting)  src/core/scene/light/light_tree.zig:25-227 (Node, importance), 346-517 (Tree)
//   Rectangle solid-angle sampling  src/core/scene/shape/rectangle.zig:199-357, 554-575 (Urena et al.)
//   shadow ray                      src/core/scene/shape/shape.zig:401-416
//   Emittance.radiance              src/core/scene/light/emittance.zig:29-59
#pragma once

#include "dsampler.hpp"
#include "dscene.hpp"
#include "dscene.hpp"

namespace zd {

constexpr float kLowThreshold = 0.00000001f;  // integrator/helper.zig:29
constexpr int   kMaxLightPicks = 64;          // light_tree.zig:249

ZD uint32_t numSamples(uint32_t num, float split_threshold) { return split_threshold <= kLowThreshold ? 1u : num; }

struct LightPick {
    uint32_t offset;
    float    pdf;
};

ZD float clampedCosSub(float cos_a, float cos_b, float sin_a, float sin_b) {
    const float angle = __fmaf_rn(cos_a, cos_b, sin_a * sin_b);
    return cos_a > cos_b ? 1.f : angle;
}
ZD float clampedSinSub(float cos_a, float cos_b, float sin_a, float sin_b) {
    const float angle = __fmaf_rn(sin_a, cos_b, -sin_b * cos_a);
    return cos_a > cos_b ? 0.f : angle;
}

// light_tree.zig:186-227
ZD float importance(f3 p, f3 n, f3 center, f3 cone_axis, float cos_cone, float radius, float power, bool two_sided,
                    bool total_sphere) {
    const f3    axis = p - center;
    const float l = length3(axis);
    const f3    na = axis / S3(l);

    const float sin_cu = mn(radius / l, 1.f);
    const float dd = dot3(cone_axis, na);
    const float cos_a = two_sided ? fabsf(dd) : dd;
    const float cos_n = mx(-dot3(n, na), 0.f);

    const float cos_cu = sqrtf(mx(__fmaf_rn(sin_cu, -sin_cu, 1.f), 0.f));
    const float sin_cone = sqrtf(mx(__fmaf_rn(cos_cone, -cos_cone, 1.f), 0.f));
    const float sin_a = sqrtf(mx(__fmaf_rn(cos_a, -cos_a, 1.f), 0.f));
    const float sin_n = sqrtf(mx(__fmaf_rn(cos_n, -cos_n, 1.f), 0.f));

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

struct LightTree {
    const DScene&     s;
    const DLightTree& t;  // s.light_tree, or a mesh part's primitive tree

    ZD static bool     hasChildren(const ZygpuLightNode& n) { return 0 != (n.meta & 1u); }
    ZD static bool     nodeTwoSided(const ZygpuLightNode& n) { return 0 != (n.meta & 2u); }
    ZD static uint32_t childrenOrLight(const ZygpuLightNode& n) { return n.meta >> 2; }

    ZD void decompressCenter(const ZygpuLightNode& n, f3& c, float& r) const {  // light_tree.zig:37-40
        const f3 bmin = xyz(t.bounds_min), bmax = xyz(t.bounds_max);
        const f3 tt = {unorm16ToFloat(n.center[0]), unorm16ToFloat(n.center[1]), unorm16ToFloat(n.center[2])};
        c = lerp3(bmin, bmax, tt);
        r = lerpf(t.bounds_min.w, t.bounds_max.w, unorm16ToFloat(n.center[3]));
    }

    ZD float nodeWeight(const ZygpuLightNode& n, f3 p, f3 nrm, bool total_sphere) const {
        f3    c;
        float r;
        decompressCenter(n, c, r);
        const f3 axis = {snorm16ToFloat(n.cone[0]), snorm16ToFloat(n.cone[1]), snorm16ToFloat(n.cone[2])};
        return importance(p, nrm, c, axis, snorm16ToFloat(n.cone[3]), r, n.power, nodeTwoSided(n), total_sphere);
    }

    ZD bool nodeSplit(const ZygpuLightNode& n, f3 p, float threshold) const {  // light_tree.zig:64-88
        f3    c;
        float r;
        decompressCenter(n, c, r);
        const float dd = mn(distance3(p, c), 1.0e6f);
        const float a = mx(dd - r, 0.001f);
        const float b = dd + r;
        const float eg = 1.f / (a * b);
        const float eg2 = eg * eg;
        const float a3 = a * a * a, b3 = b * b * b;
        const float e2g = (b3 - a3) / (3.f * (b - a) * a3 * b3);
        const float vg = e2g - eg2;
        const float ve = n.variance, ee = n.power;
        const float s2 = mx(ve * vg + ve * eg2 + ee * ee * vg, 0.f);
        const float ns = 1.f / (1.f + sqrtf(s2));
        return ns < threshold;
    }

    ZD float lightWeight(f3 p, f3 n, bool total_sphere, uint32_t light) const {  // light_tree.zig:229-235; scene.zig:664-676
        if (t.tri_props) {  // shape_sampler.zig:199-223
            const float4 c = t.tri_props[2 * light], q = t.tri_props[2 * light + 1];
            return importance(p, n, xyz(c), xyz(q), 1.f, c.w, q.w, 0 != t.two_sided, total_sphere);
        }
        const float4 bmin = s.light_aabbs[2 * light], bmax = s.light_aabbs[2 * light + 1];
        const f3     pos = S3(0.5f) * (xyz(bmin) + xyz(bmax));
        const float4 cone = s.light_cones[light];
        return importance(p, n, pos, xyz(cone), cone.w, bmax.w, bmin.w, 0 != s.lights[light].two_sided, total_sphere);
    }

    ZD LightPick nodeRandomLight(const ZygpuLightNode& node, f3 p, f3 n, bool total_sphere, float random) const {  // :90-147
        const uint32_t num_lights = node.num_lights;
        const uint32_t light = childrenOrLight(node);
        if (1 == num_lights) return {t.mapping[light], 1.f};

        uint32_t front = light, back = light + num_lights - 1;
        float    w_front = lightWeight(p, n, total_sphere, t.mapping[front]);
        float    w_back = lightWeight(p, n, total_sphere, t.mapping[back]);
        float    w_sum_front = w_front, w_sum_back = w_back, w_sum = 0.f;
        while (front != back) {
            w_sum = w_sum_front + w_sum_back;
            if (w_sum_front <= random * w_sum) {
                front += 1;
                if (front != back) {
                    w_front = lightWeight(p, n, total_sphere, t.mapping[front]);
                    w_sum_front += w_front;
                } else {
                    w_front = w_back;
                }
            } else {
                back -= 1;
                if (front != back) {
                    w_back = lightWeight(p, n, total_sphere, t.mapping[back]);
                    w_sum_back += w_back;
                }
            }
        }
        if (0.f == w_sum) return {0, 0.f};
        return {t.mapping[front], w_front / w_sum};
    }

    ZD float nodePdf(const ZygpuLightNode& node, f3 p, f3 n, bool total_sphere, uint32_t id) const {  // :149-183
        const uint32_t num_lights = node.num_lights;
        if (1 == num_lights) return 1.f;
        const uint32_t light = childrenOrLight(node), end = light + num_lights;
        float          w_id = 0.f, sum = 0.f;
        for (uint32_t i = light; i < end; ++i) {
            const float lw = lightWeight(p, n, total_sphere, t.mapping[i]);
            sum += lw;
            if (id == i) w_id = lw;
        }
        if (0.f == sum) return 0.f;
        return w_id / sum;
    }

    // Tree.randomLight (light_tree.zig:346-447), finite lights only.  Calls visit(index, pick) for every pick in
    // the reference's order; returns the number of picks.
    template <typename Timer, typename Visit>
    ZD uint32_t randomLight(Timer& tm, f3 p, f3 n, bool total_sphere, float random, float split_threshold, Visit&& visit) const {
        uint32_t   current_light = 0;
        float      ip = 0.f;
        const bool split = split_threshold > 0.f;
        // infinite lights are never part of the BVH (light_tree.zig:357-376): all of them when splitting, else one from the
        // discrete distribution over their powers when the random number falls below the guard
        if (split && t.num_infinite < kMaxLightPicks - 1) {
            for (uint32_t i = 0; i < t.num_infinite; ++i) {
                visit(current_light, LightPick{t.mapping[i], 1.f});
                current_light += 1;
            }
        } else {
            ip = t.infinite_weight;
            if (random < t.infinite_guard) {
                // Distribution1D.sampleDiscrete (distribution_1d.zig:49-59, 209-217; the lut only shortens the search)
                const uint32_t size = t.num_infinite + 1;
                uint32_t       i = 1;
                while (i < size - 1 && !(t.infinite_cdf[i] >= random)) ++i;
                const uint32_t offset = i - 1;
                visit(0u, LightPick{t.mapping[offset], (t.infinite_cdf[offset + 1] - t.infinite_cdf[offset]) * ip});
                return 1;
            }
        }
        if (0 == t.num_nodes) return current_light;

        const float    pd = 1.f - ip;
        const uint32_t max_split_depth = t.max_split_depth;

        // traversal stack of at most MaxSplitDepth + 1 entries (light_tree.zig:721-751)
        float    st_pdf[12], st_random[12];
        uint32_t st_node[12], st_depth[12];
        int      end = 0;

        // (entry 0 is the reference's copy of the start state: popped last and never used, so it is neither written nor read here:
        // a walk that does not split stays out of scratch)
        float    t_pdf = pd, t_random = (random - ip) / pd;
        uint32_t t_node = 0, t_depth = split ? 0u : max_split_depth;
        end = 1;
        auto pop = [&]() {
            end -= 1;
            if (end > 0) {
                t_pdf = st_pdf[end];
                t_random = st_random[end];
                t_node = st_node[end];
                t_depth = st_depth[end];
            }
        };

        while (0 != end) {
            const ZygpuLightNode node = t.nodes[t_node];
            if (hasChildren(node)) {
                tm.mark(8);
                const bool     do_split = t_depth < max_split_depth && nodeSplit(node, p, split_threshold);
                tm.mark(9);
                const uint32_t c0 = childrenOrLight(node), c1 = c0 + 1;
                if (do_split) {
                    t_depth += 1;
                    t_node = c0;
                    st_pdf[end] = t_pdf;
                    st_random[end] = t_random;
                    st_node[end] = c1;
                    st_depth[end] = t_depth;
                    end += 1;
                } else {
                    t_depth = max_split_depth;
                    float       p0 = nodeWeight(t.nodes[c0], p, n, total_sphere);
                    float       p1 = nodeWeight(t.nodes[c1], p, n, total_sphere);
                    const float pt = p0 + p1;
                    if (0.f == pt) {
                        pop();
                        continue;
                    }
                    p0 /= pt;
                    p1 /= pt;
                    if (t_random < p0) {
                        t_node = c0;
                        t_pdf *= p0;
                        t_random /= p0;
                    } else {
                        t_node = c1;
                        t_pdf *= p1;
                        t_random = mn((t_random - p0) / p1, 1.f);
                    }
                }
            } else {
                tm.mark(10);
                const LightPick pick = nodeRandomLight(node, p, n, total_sphere, t_random);
                tm.mark(11);
                if (pick.pdf > 0.f) {
                    visit(current_light, LightPick{pick.offset, pick.pdf * t_pdf});
                    current_light += 1;
                }
                pop();
            }
        }
        return current_light;
    }

    // Tree.pdf (light_tree.zig:449-517)
    ZD float pdf(f3 p, f3 n, bool total_sphere, float split_threshold, uint32_t id) const {
        const uint32_t lo = t.orders[id];
        const bool     split = split_threshold > 0.f;
        const bool     split_infinite = split && t.num_infinite < kMaxLightPicks - 1;
        if (lo < t.infinite_end) return split_infinite ? 1.f : t.infinite_weight * (t.infinite_cdf[lo + 1] - t.infinite_cdf[lo]);
        if (0 == t.num_nodes) return 0.f;
        const float    ip = split_infinite ? 0.f : t.infinite_weight;
        const uint32_t max_split_depth = t.max_split_depth;
        float          pd = 1.f - ip;
        uint32_t       nid = 0;
        uint32_t       depth = split ? 0u : max_split_depth;
        for (;;) {
            const ZygpuLightNode node = t.nodes[nid];
            if (hasChildren(node)) {
                const bool     do_split = depth < max_split_depth && nodeSplit(node, p, split_threshold);
                const uint32_t c0 = childrenOrLight(node), c1 = c0 + 1;
                const uint32_t middle = t.middles[nid];
                if (do_split) {
                    depth += 1;
                    nid = lo < middle ? c0 : c1;
                } else {
                    depth = max_split_depth;
                    const float p0 = nodeWeight(t.nodes[c0], p, n, total_sphere);
                    const float p1 = nodeWeight(t.nodes[c1], p, n, total_sphere);
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

struct SphQuad {  // rectangle.zig:199-303
    f3    o, x, y, z;
    float z0, x0, y0, x1, y1, b0, b1, k, Sa;

    ZD static SphQuad init(f3 scale, f3 o) {
        const f3 sv = {-0.5f * scale.x, -0.5f * scale.y, 0.f};
        const f3 ex = {scale.x, 0.f, 0.f};
        const f3 ey = {0.f, scale.y, 0.f};
        SphQuad  q;
        q.o = o;
        const float exl = length3(ex), eyl = length3(ey);
        q.x = ex / S3(exl);
        q.y = ey / S3(eyl);
        q.z = cross3(q.x, q.y);
        const f3 d = sv - o;
        q.z0 = dot3(d, q.z);
        if (q.z0 > 0.f) {
            q.z = -q.z;
            q.z0 = -q.z0;
        }
        q.x0 = dot3(d, q.x);
        q.y0 = dot3(d, q.y);
        q.x1 = q.x0 + exl;
        q.y1 = q.y0 + eyl;
        const f3 v00 = {q.x0, q.y0, q.z0}, v01 = {q.x0, q.y1, q.z0}, v10 = {q.x1, q.y0, q.z0}, v11 = {q.x1, q.y1, q.z0};
        const f3 n0 = normalize3(cross3(v00, v10));
        const f3 n1 = normalize3(cross3(v10, v11));
        const f3 n2 = normalize3(cross3(v11, v01));
        const f3 n3 = normalize3(cross3(v01, v00));
        const float g0 = acosF(-dot3(n0, n1));
        const float g1 = acosF(-dot3(n1, n2));
        const float g2 = acosF(-dot3(n2, n3));
        const float g3 = acosF(-dot3(n3, n0));
        q.b0 = n0.z;
        q.b1 = n2.z;
        q.k = 2.f * kPi - g2 - g3;
        q.Sa = g0 + g1 - q.k;
        return q;
    }

    ZD f3 sample(float u0, float u1) const {
        const float au = u0 * Sa + k;
        const float fu = (cosF(au) * b0 - b1) / sinF(au);
        float       cu = 1.f / sqrtf(fu * fu + b0 * b0) * (fu > 0.f ? 1.f : -1.f);
        cu = stdClamp(cu, -1.f, 1.f);
        float xu = -(cu * z0) / sqrtD(1.f - cu * cu);
        xu = stdClamp(xu, x0, x1);
        const float dd = sqrtf(xu * xu + z0 * z0);
        const float h0 = y0 / sqrtf(dd * dd + y0 * y0);
        const float h1 = y1 / sqrtf(dd * dd + y1 * y1);
        const float hv = h0 + u1 * (h1 - h0);
        const float hv2 = hv * hv;
        const float eps = __uint_as_float(0x35800000u);
        const float yv = hv2 < 1.f - eps ? ((hv * dd) / sqrtD(1.f - hv2)) : y1;
        return o + S3(xu) * x + S3(yv) * y + S3(z0) * z;
    }

    ZD float pdf(f3 scale) const {
        const float sqr_dist = squaredLength3(o);
        const float area = scale.x * scale.y;
        const float numer = area * fabsf(o.z);
        const float denom = sqr_dist * sqrtf(sqr_dist);
        return numer > denom * kDotMin ? (1.f / Sa) : (denom / numer);
    }
};

struct SampleTo {  // shape/sample.zig:7-29
    f3    p, n, wi;
    float pdf;
    float u = 0.f, v = 0.f;  // uvw: set by the image lights only (it addresses the emission map)
};

// rectangle.zig:305-357 (UseSphericalSampling): the set-up shared by a light's samples, then one sample per 2-D draw
struct RectangleSampler {
    SphQuad squad;
    float   sample_pdf;

    ZD static RectangleSampler init(f3 p, const Trafo& trafo, uint32_t num_samples) {
        RectangleSampler r;
        r.squad = SphQuad::init(trafo.scale, trafo.worldToFramePoint(p));
        r.sample_pdf = float(num_samples) * r.squad.pdf(trafo.scale);
        return r;
    }
    ZD bool sample(float u0, float u1, f3 p, f3 n, const Trafo& trafo, bool two_sided, bool total_sphere, SampleTo& out) const {
        const f3 ls = squad.sample(u0, u1);
        const f3 ws = trafo.frameToWorldPoint(ls);
        const f3 dir = normalize3(ws - p);
        f3       wn = trafo.r2;
        if (two_sided && dot3(wn, dir) > 0.f) wn = -wn;
        if (-dot3(wn, dir) < kDotMin || 0.f == squad.Sa || (dot3(dir, n) <= 0.f && !total_sphere)) return false;
        out.p = ws;
        out.n = wn;
        out.wi = dir;
        out.pdf = sample_pdf;
        return true;
    }
};

// one-sample form (what the staged NEE rounds use)
ZD bool rectangleSampleTo(f3 p, f3 n, const Trafo& trafo, bool two_sided, bool total_sphere, Sampler& sampler, SampleTo& out) {
    const RectangleSampler rs = RectangleSampler::init(p, trafo, 1);
    float                  u0, u1;
    sampler.sample2D(u0, u1);
    return rs.sample(u0, u1, p, n, trafo, two_sided, total_sphere, out);
}

ZD float rectanglePdf(f3 p, const Trafo& frag_trafo, uint32_t num_samples) {  // rectangle.zig:554-575 via shape.zig:469-492
    const SphQuad squad = SphQuad::init(frag_trafo.scale, frag_trafo.worldToFramePoint(p));
    return float(num_samples) * squad.pdf(frag_trafo.scale);
}

// ---------------------------------------------------------------- Sphere lights (sphere.zig:318-400, 469-485)

ZD float conePdfUniform(float one_minus_cos_theta_max) {  // sampling.zig:103-106
    return 1.f / ((2.f * kPi) * mx(one_minus_cos_theta_max, 1.0e-20f));
}

// Sphere.sampleTo: the set-up shared by a light's samples (frame around the direction to the centre), then one sample
// of the uniform cone of directions per 2-D draw
struct SphereSampler {
    f3    fx, fy, z;
    float l, r, nsf;
    bool  valid;

    ZD static SphereSampler init(f3 p, const Trafo& trafo, uint32_t num_samples) {
        SphereSampler q;
        const f3      v = trafo.position - p;
        q.l = length3(v);
        q.r = 0.5f * trafo.scale.x;
        q.valid = !(q.l <= (q.r + 0.0000001f));
        q.z = S3(1.f / q.l) * v;
        // Frame.init -> orthonormalBasis3 (vector4.zig:98-110)
        const float sign = copysignf(1.f, q.z.z);
        const float c = -1.f / (sign + q.z.z);
        const float d = q.z.x * q.z.y * c;
        q.fx = mk3(1.f + sign * q.z.x * q.z.x * c, sign * d, -sign * q.z.x);
        q.fy = mk3(d, sign + q.z.y * q.z.y * c, -q.z.y);
        q.nsf = float(num_samples);
        return q;
    }
    ZD bool sample(float s0, float s1, f3 p, f3 n, const Trafo& trafo, bool total_sphere, SampleTo& out) const {
        if (!valid) return false;
        const float sin_theta_max = r / l;
        const float sin2_theta_max = sin_theta_max * sin_theta_max;
        const float cos_theta_max = sqrtD(1.f - sin2_theta_max);
        float       one_minus_cos_theta_max = 1.f - cos_theta_max;
        float       cos_theta = (cos_theta_max - 1.f) * s0 + 1.f;
        float       sin2_theta = 1.f - (cos_theta * cos_theta);
        if (sin2_theta_max < 0.00068523f) {
            sin2_theta = sin2_theta_max * s0;
            cos_theta = sqrtD(1.f - sin2_theta);
            one_minus_cos_theta_max = 0.5f * sin2_theta_max;
        }
        const float cos_alpha = mn(sin2_theta / sin_theta_max + cos_theta * sqrtD(1.f - mn(sin2_theta / sin2_theta_max, 1.f)), 1.f);
        const float sin_alpha = sqrtD(1.f - cos_alpha * cos_alpha);
        const float phi = s1 * (2.f * kPi);
        const f3    mw = -mk3(cosF(phi) * sin_alpha, sinF(phi) * sin_alpha, cos_alpha);  // sampling.zig:78-83 sphereDirection
        const f3    wn = fma3(S3(mw.z), z, fma3(S3(mw.y), fy, S3(mw.x) * fx));             // frame.frameToWorld (frame.zig:23-37)
        const f3    lp = trafo.position + S3(r) * wn;
        const f3    dir = normalize3(lp - p);
        if (dot3(dir, n) <= 0.f && !total_sphere) return false;
        out.p = lp;
        out.n = wn;
        out.wi = dir;
        out.pdf = nsf * conePdfUniform(one_minus_cos_theta_max);
        return true;
    }
};

ZD float spherePdf(f3 p, const Trafo& trafo, uint32_t num_samples) {  // sphere.zig:469-485
    const f3    v = trafo.position - p;
    const float l2 = squaredLength3(v);
    const float r = 0.5f * trafo.scale.x;
    const float sin2_theta_max = (r * r) / l2;
    const float one_minus_cos_theta_max = sin2_theta_max < 0.00068523f ? 0.5f * sin2_theta_max : 1.f - sqrtf(mx(1.f - sin2_theta_max, 0.f));
    return float(num_samples) * conePdfUniform(one_minus_cos_theta_max);
}

// ---------------------------------------------------------------- Disk lights (disk.zig:180-359, 492-533)
// equi-angular sampling (UseEquiAngularSampling) of the chord through the projected shading point, then of the chord across it

struct DiskSamplerData {  // disk.zig:181-194
    f3 xd, yd;
    ZD static DiskSamplerData init(f3 p) {
        const f3 td = mk3(p.y, -p.x, 0.f);
        DiskSamplerData d;
        d.xd = (0.f == td.x && 0.f == td.y) ? mk3(1.f, 0.f, 0.f) : normalize3(td);
        d.yd = mk3(-d.xd.y, d.xd.x, 0.f);
        return d;
    }
};
struct EquiAngularSampling {  // disk.zig:196-250
    float offset, min_t, max_t, scale, scale_sqr, angle_min, angle_extent;
    ZD static EquiAngularSampling init(f3 source, f3 origin, f3 direction, float min_t, float max_t) {
        EquiAngularSampling e;
        e.offset = dot3(direction, source - origin) / squaredLength3(direction);
        e.scale_sqr = squaredLength3((origin + S3(e.offset) * direction) - source);
        e.scale = sqrtf(e.scale_sqr);
        const float inv_scale = 0.f == e.scale ? 0.f : 1.f / e.scale;
        e.angle_min = atanF((min_t - e.offset) * inv_scale);
        const float angle_max = atanF((max_t - e.offset) * inv_scale);
        e.angle_extent = angle_max - e.angle_min;
        e.min_t = min_t;
        e.max_t = max_t;
        return e;
    }
    ZD float sample(float u, float& t) const {
        const float lt = scale * tanF(angle_min + u * angle_extent);
        const float p = scale / (angle_extent * (scale_sqr + lt * lt));
        t = clampf(lt + offset, min_t, max_t);
        return p;
    }
    ZD float pdf(float t) const {
        if (min_t <= t && t < max_t) {
            const float lt = t - offset;
            return scale / (angle_extent * (scale_sqr + lt * lt));
        }
        return 0.f;
    }
    ZD float pdfAndSample(float t, float& u) const {
        const float lt = t - offset;
        u = clampf((atanF(lt / scale) - angle_min) / angle_extent, 0.f, 1.f);
        return scale / (angle_extent * (scale_sqr + lt * lt));
    }
};
// Disk.sampleTo (disk.zig:252-324): the set-up shared by a light's samples; sample() makes one sample's draws (a 2-D one,
// then a 1-D one unless the chord degenerates) from `smp`
struct DiskSampler {
    f3                  lp;
    DiskSamplerData     dsd;
    EquiAngularSampling eas0;
    float               radius, nsf;
    bool                valid;
    ZD static DiskSampler init(f3 p, const Trafo& trafo, uint32_t num_samples) {
        DiskSampler d;
        d.radius = 0.5f * trafo.scale.x;
        d.nsf = float(num_samples);
        d.lp = trafo.worldToObjectNormal(p - trafo.position);  // worldToFramePoint
        d.dsd = DiskSamplerData::init(d.lp);
        d.eas0 = EquiAngularSampling::init(d.lp, S3(0.f), d.dsd.yd, -d.radius, d.radius);
        d.valid = 0.f != d.eas0.angle_extent;
        return d;
    }
    ZD bool sample(Sampler& smp, f3 p, f3 n, const Trafo& trafo, bool two_sided, bool total_sphere, SampleTo& out) const {
        float r0, r1;
        smp.sample2D(r0, r1);
        float x, y;
        diskConcentric(r0, r1, x, y);
        float u = x;
        float pdf_ = sqrtD(1.f - u * u) / (0.25f * kPi);
        u = (u + 1.f) * 0.5f;
        float y_coord;
        pdf_ *= eas0.sample(u, y_coord);
        const float x_chord = sqrtD(radius * radius - y_coord * y_coord);
        if (0.f == x_chord) return false;
        const EquiAngularSampling eas1 = EquiAngularSampling::init(lp, S3(y_coord) * dsd.yd, dsd.xd, -x_chord, x_chord);
        if (0.f == eas1.angle_extent) return false;
        float x_coord;
        pdf_ *= eas1.sample(smp.sample1D(), x_coord);
        const f3 l_direction = S3(x_coord) * dsd.xd + S3(y_coord) * dsd.yd - lp;
        const f3 axis = trafo.objectToWorldNormal(l_direction);
        const f3 ws = p + axis;
        f3       wn = trafo.r2;
        if (two_sided && dot3(wn, ws - p) > 0.f) wn = -wn;
        const float sl = squaredLength3(axis);
        const f3    dir = axis / S3(sqrtf(sl));
        const float c = -dot3(wn, dir);
        if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) return false;
        out.p = ws;
        out.n = wn;
        out.wi = dir;
        out.pdf = (nsf * pdf_ * sl) / c;
        out.u = u;
        out.v = (y + 1.f) * 0.5f;
        return true;
    }
};
ZD float diskPdf(f3 dir, f3 p, const Trafo& trafo, f3 frag_p, uint32_t num_samples) {  // disk.zig:492-533
    const float               c = fabsf(dot3(trafo.r2, dir));
    const float               nsf = float(num_samples);
    const float               radius = 0.5f * trafo.scale.x;
    const f3                  d = p - frag_p;
    const float               sl = dot3(d, d);  // squaredDistance3
    const f3                  lp = trafo.worldToObjectNormal(p - trafo.position);
    const DiskSamplerData     dsd = DiskSamplerData::init(lp);
    const EquiAngularSampling eas0 = EquiAngularSampling::init(lp, S3(0.f), dsd.yd, -radius, radius);
    const f3                  l_point = trafo.worldToObjectNormal(frag_p - trafo.position);
    const float               y_coord = dot3(l_point, dsd.yd);
    float                     u;
    const float               eas_pdf = eas0.pdfAndSample(y_coord, u);
    u = u * 2.f - 1.f;
    float pdf_ = sqrtD(1.f - u * u) / (0.25f * kPi);
    pdf_ *= eas_pdf;
    const float               x_chord = sqrtD(radius * radius - y_coord * y_coord);
    const EquiAngularSampling eas1 = EquiAngularSampling::init(lp, S3(y_coord) * dsd.yd, dsd.xd, -x_chord, x_chord);
    const float               x_coord = dot3(l_point, dsd.xd);
    pdf_ *= eas1.pdf(x_coord);
    return (nsf * pdf_ * sl) / c;
}

// ---------------------------------------------------------------- Distant lights (distant.zig:21-146)

ZD float distantSolidAngle(float radius) { return (2.f * kPi) * (1.f - sqrtf(1.f / (radius * radius + 1.f))); }

ZD bool distantSampleTo(f3 n, const Trafo& trafo, bool total_sphere, Sampler& sampler, SampleTo& out) {
    const float radius = trafo.scale.x;
    if (radius <= 0.f) return false;
    float r0, r1;
    sampler.sample2D(r0, r1);
    float x, y;
    diskConcentric(r0, r1, x, y);
    const f3 ws = S3(radius) * trafo.transformVector(mk3(x, y, 0.f));
    const f3 dir = normalize3(ws - trafo.r2);
    if (dot3(dir, n) <= 0.f && !total_sphere) return false;
    out.p = S3(kRayMaxT) * dir;
    out.n = trafo.r2;
    out.wi = dir;
    out.pdf = 1.f / distantSolidAngle(radius);
    return true;
}

// Distant.intersect (distant.zig:22-52): only rays that left the scene (max_t == RayMaxT)
ZD bool distantIntersect(const Ray& ray, const Trafo& trafo, float& u, float& v) {
    const float radius = trafo.scale.x;
    const f3    n = trafo.r2;
    const float b = dot3(n, ray.direction);
    if (b > 0.f || ray.max_t < kRayMaxT || radius <= 0.f) return false;
    const float det = (b * b) - dot3(n, n) + (radius * radius);
    if (det >= 0.f) {
        const f3 k = ray.direction - n;
        const f3 sk = k / S3(radius);
        u = dot3(trafo.r0, sk);
        v = dot3(trafo.r1, sk);
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- Dome / Canopy, uniform emission (dome.zig, canopy.zig)

ZD f3 hemisphereUniform(float u0, float u1) {  // sampling.zig:56-65
    const float z = 1.f - u0;
    const float r = sqrtf(mx(0.f, 1.f - z * z));
    const float phi = u1 * (2.f * kPi);
    return mk3(cosF(phi) * r, sinF(phi) * r, z);
}
ZD f3 sphereUniform(float u0, float u1) {  // sampling.zig:67-76
    const float z = 1.f - 2.f * u0;
    const float r = sqrtf(mx(0.f, 1.f - z * z));
    const float phi = u1 * (2.f * kPi);
    return mk3(cosF(phi) * r, sinF(phi) * r, z);
}
constexpr float kCanopyEps = -0.0005f;  // canopy.zig:25

// Dome.sampleTo (dome.zig:58-103)
ZD bool domeSampleTo(f3 n, bool total_sphere, Sampler& sampler, SampleTo& out) {
    float u0, u1;
    sampler.sample2D(u0, u1);
    f3 dir;
    if (total_sphere) {
        dir = sphereUniform(u0, u1);
        out.pdf = 1.f / (4.f * kPi);
    } else {
        const f3 dl = hemisphereUniform(u0, u1);
        dir = Frame::init(n).frameToWorld(dl);
        if (dot3(dir, n) <= 0.f) return false;
        out.pdf = 1.f / (2.f * kPi);
    }
    out.p = S3(kRayMaxT) * dir;
    out.n = -dir;
    out.wi = dir;
    return true;
}

// Canopy.sampleTo (canopy.zig:60-92)
ZD bool canopySampleTo(f3 n, const Trafo& trafo, bool total_sphere, Sampler& sampler, SampleTo& out) {
    float u0, u1;
    sampler.sample2D(u0, u1);
    const f3 dl = hemisphereUniform(u0, u1);
    const f3 dir = Frame{trafo.r0, trafo.r1, trafo.r2}.frameToWorld(dl);
    if (dot3(dir, n) <= 0.f && !total_sphere) return false;
    out.p = S3(kRayMaxT) * dir;
    out.n = -dir;
    out.wi = dir;
    out.pdf = 1.f / (2.f * kPi);
    return true;
}

// ---------------------------------------------------------------- image lights: emission maps on Dome / Canopy
// Distribution1D / 2D over the flattened rows (base/math/distribution_1d.zig:49-102, 209-217, distribution_2d.zig:67-90),
// ImageImpl (shape_sampler.zig:128-152)
struct Dist1D {
    const float*    cdf;
    const uint32_t* lut;
    uint32_t        size;
    float           lut_range;
    ZD static Dist1D row(const DImageLight& il, uint32_t r) {
        const uint4 w = il.rows[r];
        return {il.cdf + w.x, il.lut + w.z, w.y, __uint_as_float(w.w)};
    }
    // The reference scans the cdf from the bucket's first entry to the first one >= r (distribution_1d.zig:209-217).  The bucket's
    // entries end where the next bucket's begin (lut[bucket + 1]: the first entry whose cdf maps past this bucket, so its cdf is > r),
    // and the cdf does not decrease: the same entry by bisection, in 4-5 dependent loads where the scan of a 1024-entry row of a smooth
    // image takes 8 on average and as many as the longest lane of the wave (the sky image of the outdoor scene: a quarter of the
    // light-sampling stage's wave cycles went here)
    ZD uint32_t sample(float r) const {
        const uint32_t bucket = uint32_t(r * lut_range);
        uint32_t       lo = lut[bucket], hi = min(lut[bucket + 1u], size - 1u);
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cdf[mid] >= r) hi = mid;
            else lo = mid + 1u;
        }
        return lo - 1u;
    }
    ZD void sampleContinuous(float r, float& offset_out, float& pdf) const {
        const uint32_t offset = sample(r);
        const float    c = cdf[offset + 1];
        const float    v = c - cdf[offset];
        if (0.f == v) {
            offset_out = 0.f;
            pdf = 0.f;
            return;
        }
        const float t = (c - r) / v;
        offset_out = (float(offset) + t) / float(size - 1);
        pdf = v;
    }
    ZD float pdfF(float u) const {
        const uint32_t o = min(uint32_t(u * float(size - 1)), size - 2);
        return cdf[o + 1] - cdf[o];
    }
};

ZD float imageLightSample(const DImageLight& il, float r0, float r1, float& u, float& v) {  // pdf * total_weight
    float v_pdf, u_pdf;
    Dist1D::row(il, il.num_rows).sampleContinuous(r1, v, v_pdf);
    const uint32_t c = min(uint32_t(v * float(il.num_rows)), il.num_rows - 1);
    Dist1D::row(il, c).sampleContinuous(r0, u, u_pdf);
    return (u_pdf * v_pdf) * il.total_weight;
}
ZD float imageLightPdf(const DImageLight& il, float u, float v) {
    const float    au = texAddress(0 != (il.sampler & ZYGPU_TEX_REPEAT_U), u), av = texAddress(0 != (il.sampler & ZYGPU_TEX_REPEAT_V), v);
    const float    v_pdf = Dist1D::row(il, il.num_rows).pdfF(av);
    const uint32_t c = min(uint32_t(av * float(il.num_rows)), il.num_rows - 1);
    const float    u_pdf = Dist1D::row(il, c).pdfF(au);
    return (u_pdf * v_pdf) * il.total_weight;
}

ZD void domeUv(f3 dir, const Trafo& trafo, float& u, float& v) {  // Dome.fragment's uvw (dome.zig:40-49)
    const f3 xyz = normalize3(trafo.transformVectorTransposed(dir));
    u = atan2F(xyz.x, xyz.z) * (kPiInv * 0.5f) + 0.5f;
    v = acosF(xyz.y) * kPiInv;
}
ZD void canopyUv(f3 dir, const Trafo& trafo, float& u, float& v) {  // Canopy.fragment's uvw (canopy.zig:44-52, 175-187)
    const f3    xyz = normalize3(trafo.transformVectorTransposed(dir));
    const float colatitude = acosF(xyz.z);
    const float longitude = atan2F(-xyz.y, xyz.x);
    const float r = colatitude * (kPiInv * 2.f);
    u = 0.5f * (r * cosF(longitude)) + 0.5f;
    v = 0.5f * (r * sinF(longitude)) + 0.5f;
}
ZD f3 diskToHemisphereEquidistant(float x, float y) {  // canopy.zig:189-211
    const float longitude = atan2F(-y, x);
    const float r = sqrtf(x * x + y * y);
    const float colatitude = r * (kPi / 2.f);
    const float sin_col = sinF(colatitude), cos_col = cosF(colatitude);
    const float sin_lon = sinF(longitude), cos_lon = cosF(longitude);
    return mk3(sin_col * cos_lon, sin_col * sin_lon, cos_col);
}

// one sample of Dome.sampleMaterialTo (dome.zig:105-158) from its 2-D draw; nsf = the light's sample count
ZD bool domeSampleMaterial(const DImageLight& il, float r0, float r1, float nsf, f3 n, const Trafo& trafo, bool total_sphere, SampleTo& out) {
    float       u, v;
    const float pdf = imageLightSample(il, r0, r1, u, v);
    if (0.f == pdf) return false;
    const float phi = (u - 0.5f) * (2.f * kPi), theta = v * kPi;
    const float sin_phi = sinF(phi), cos_phi = cosF(phi), sin_theta = sinF(theta), cos_theta = cosF(theta);
    const f3    dir = trafo.transformVector(mk3(sin_phi * sin_theta, cos_theta, cos_phi * sin_theta));
    if (dot3(dir, n) <= 0.f && !total_sphere) return false;
    out.p = S3(kRayMaxT) * dir;
    out.n = -dir;
    out.wi = dir;
    out.pdf = (nsf * pdf) / ((4.f * kPi) * sin_theta);
    out.u = u;
    out.v = v;
    return true;
}
ZD float domeMaterialPdf(const DImageLight& il, float u, float v, uint32_t num_samples) {  // dome.zig:222-234
    const float sin_theta = sinF(v * kPi);
    if (0.f == sin_theta) return 0.f;
    const float material_pdf = imageLightPdf(il, u, v) * float(num_samples);
    return material_pdf / ((4.f * kPi) * sin_theta);
}
// Canopy.sampleMaterialTo (canopy.zig:94-133)
ZD bool canopySampleMaterialTo(const DImageLight& il, f3 n, const Trafo& trafo, bool total_sphere, Sampler& sampler, SampleTo& out) {
    float r0, r1;
    sampler.sample2D(r0, r1);
    float       u, v;
    const float pdf = imageLightSample(il, r0, r1, u, v);
    if (0.f == pdf) return false;
    const float dx = 2.f * u - 1.f, dy = 2.f * v - 1.f;
    if (dx * dx + dy * dy > 1.f) return false;
    const f3 dir = trafo.transformVector(diskToHemisphereEquidistant(dx, dy));
    if (dot3(dir, n) <= 0.f && !total_sphere) return false;
    out.p = S3(kRayMaxT) * dir;
    out.n = -dir;
    out.wi = dir;
    out.pdf = pdf / (2.f * kPi);
    out.u = u;
    out.v = v;
    return true;
}

// ---- Portal lights: a Rectangle that is a window onto an environment map
//   Portal.imageToWorld / worldToImageWeighted / imageBounds      scene/shape/portal.zig:16-80
//   SummedAreaTable.integral / lookup, EvaluatorX / EvaluatorY      base/math/summed_area_table.zig:59-289
//   WindowedDistribution2D.sampleContinuous / pdf / eval            base/math/windowed_distribution_2d.zig:31-105
//   PortalImpl.portalUvw                                            scene/shape/shape_sampler.zig:274-286
//   Rectangle.samplePortalTo / portalPdf                            scene/shape/rectangle.zig:458-513, 597-611
struct Bounds2 {
    float min0, min1, max0, max1;
};

ZD bool portalWorldToImage(f3 dir, const Trafo& trafo, float& u, float& v) {
    const f3 w = trafo.worldToObjectNormal(dir);
    if (w.z <= 0.f) return false;
    const float a = atan2F(w.x, w.z), b = atan2F(w.y, w.z);
    u = clampf((a + kPi / 2.f) / kPi, 0.f, 1.f);
    v = clampf((b + kPi / 2.f) / kPi, 0.f, 1.f);
    return true;
}
ZD bool portalWorldToImageWeighted(f3 dir, const Trafo& trafo, float& u, float& v, float& weight) {
    const f3 w = trafo.worldToObjectNormal(dir);
    if (w.z <= 0.f) return false;
    const float alpha = atan2F(w.x, w.z), beta = atan2F(w.y, w.z);
    weight = (kPi * kPi) * (1.f - w.x * w.x) * (1.f - w.y * w.y) / w.z;
    u = clampf((alpha + kPi / 2.f) / kPi, 0.f, 1.f);
    v = clampf((beta + kPi / 2.f) / kPi, 0.f, 1.f);
    return true;
}
ZD bool portalImageBounds(f3 p, const Trafo& trafo, Bounds2& out) {
    const f3 a = trafo.r0 * S3(trafo.scale.x), b = trafo.r1 * S3(trafo.scale.y);
    const f3 ab = a + b;
    const f3 o = p - trafo.position;
    const f3 c0 = S3(0.5f) * ab + o, c1 = S3(-0.5f) * ab + o;
    float    u0, v0, u1, v1;
    if (!portalWorldToImage(normalize3(c0), trafo, u0, v0)) return false;
    if (!portalWorldToImage(normalize3(c1), trafo, u1, v1)) return false;
    out.min0 = mn(u0, u1);
    out.min1 = mn(v0, v1);
    out.max0 = mx(u0, u1);
    out.max1 = mx(v0, v1);
    return true;
}

struct SatView {
    const float* buffer;
    const float* func;
    int          dim;
    ZD float lookup(int x0, int x1, int y0, int y1, float dx, float dy) const {
        const int   w = dim + 1;
        const float v00 = buffer[x0 + y0 * w], v10 = buffer[x1 + y0 * w], v01 = buffer[x0 + y1 * w], v11 = buffer[x1 + y1 * w];
        const float odx = 1.f - dx, ody = 1.f - dy;
        return ody * (odx * v00 + dx * v10) + dy * (odx * v01 + dx * v11);
    }
    ZD float integral(const Bounds2& e) const {
        const float fd = float(dim);
        const float ax = e.min0 * fd, ay = e.min1 * fd, bx = e.max0 * fd, by = e.max1 * fd;
        const int   ax0 = int(ax), ay0 = int(ay), bx0 = int(bx), by0 = int(by);
        const int   axx0 = min(ax0, dim), ayy0 = min(ay0, dim), axx1 = min(ax0 + 1, dim), ayy1 = min(ay0 + 1, dim);
        const int   bxx0 = min(bx0, dim), byy0 = min(by0, dim), bxx1 = min(bx0 + 1, dim), byy1 = min(by0 + 1, dim);
        const float adx = ax - float(ax0), ady = ay - float(ay0), bdx = bx - float(bx0), bdy = by - float(by0);
        const float s = (lookup(bxx0, bxx1, byy0, byy1, bdx, bdy) - lookup(axx0, axx1, byy0, byy1, adx, bdy)) +
                        (lookup(axx0, axx1, ayy0, ayy1, adx, ady) - lookup(bxx0, bxx1, ayy0, ayy1, bdx, ady));
        return mx(s / (fd * fd), 0.f);
    }
    ZD float eval(float u, float v) const {
        const uint32_t x = uint32_t(u * float(dim)), y = uint32_t(v * float(dim));
        return func[min(y, uint32_t(dim - 1)) * uint32_t(dim) + min(x, uint32_t(dim - 1))];
    }
};

struct SatEvaluatorX {
    const SatView& sat;
    int            ay0, by0;
    float          lookup_b, lookup_c, ady, bdy, integral_area;
    ZD SatEvaluatorX(const SatView& s, const Bounds2& e, float integral) : sat(s) {
        const int   d = s.dim;
        const float fd = float(d);
        const float ax = e.min0 * fd, ay = e.min1 * fd, by = e.max1 * fd;
        const int   ax0 = int(ax);
        ay0 = int(ay);
        by0 = int(by);
        const int axx0 = min(ax0, d), ayy0 = min(ay0, d), axx1 = min(ax0 + 1, d), ayy1 = min(ay0 + 1, d);
        const int byy0 = min(by0, d), byy1 = min(by0 + 1, d);
        const float adx = ax - float(ax0);
        ady = ay - float(ay0);
        bdy = by - float(by0);
        lookup_b = s.lookup(axx0, axx1, byy0, byy1, adx, bdy);
        lookup_c = s.lookup(axx0, axx1, ayy0, ayy1, adx, ady);
        integral_area = integral * (fd * fd);
    }
    ZD float eval(float x) const {
        const int   d = sat.dim;
        const float bx = x * float(d);
        const int   bx0 = int(bx);
        const int   ayy0 = min(ay0, d), ayy1 = min(ay0 + 1, d);
        const int   bxx0 = min(bx0, d), byy0 = min(by0, d), bxx1 = min(bx0 + 1, d), byy1 = min(by0 + 1, d);
        const float bdx = bx - float(bx0);
        const float lookup_a = sat.lookup(bxx0, bxx1, byy0, byy1, bdx, bdy), lookup_d = sat.lookup(bxx0, bxx1, ayy0, ayy1, bdx, ady);
        const float s = (lookup_a - lookup_b) + (lookup_c - lookup_d);
        return mx(s / integral_area, 0.f);
    }
};

struct SatEvaluatorY {
    const SatView& sat;
    int            ax0, bx0;
    float          lookup_c, lookup_d, adx, bdx, integral_area;
    ZD SatEvaluatorY(const SatView& s, const Bounds2& e, float integral) : sat(s) {
        const int   d = s.dim;
        const float fd = float(d);
        const float ax = e.min0 * fd, ay = e.min1 * fd, bx = e.max0 * fd;
        ax0 = int(ax);
        const int ay0 = int(ay);
        bx0 = int(bx);
        const int axx0 = min(ax0, d), ayy0 = min(ay0, d), axx1 = min(ax0 + 1, d), ayy1 = min(ay0 + 1, d);
        const int bxx0 = min(bx0, d), bxx1 = min(bx0 + 1, d);
        adx = ax - float(ax0);
        const float ady = ay - float(ay0);
        bdx = bx - float(bx0);
        lookup_c = s.lookup(axx0, axx1, ayy0, ayy1, adx, ady);
        lookup_d = s.lookup(bxx0, bxx1, ayy0, ayy1, bdx, ady);
        integral_area = integral * (fd * fd);
    }
    ZD float eval(float y) const {
        const int   d = sat.dim;
        const float by = y * float(d);
        const int   by0 = int(by);
        const int   axx0 = min(ax0, d), axx1 = min(ax0 + 1, d);
        const int   bxx0 = min(bx0, d), byy0 = min(by0, d), bxx1 = min(bx0 + 1, d), byy1 = min(by0 + 1, d);
        const float bdy = by - float(by0);
        const float lookup_a = sat.lookup(bxx0, bxx1, byy0, byy1, bdx, bdy), lookup_b = sat.lookup(axx0, axx1, byy0, byy1, adx, bdy);
        const float s = (lookup_a - lookup_b) + (lookup_c - lookup_d);
        return mx(s / integral_area, 0.f);
    }
};

template <typename Cdf>
ZD float sampleBisection(const Cdf& cdf, float u, float imin, float imax, float nf) {
    float lo = imin, hi = imax;
    while (ceilf(nf * hi) - floorf(nf * lo) > 1.f) {
        const float mid = (lo + hi) / 2.f;
        if (cdf.eval(mid) > u) hi = mid;
        else lo = mid;
    }
    const float pmin = cdf.eval(lo), pmax = cdf.eval(hi);
    const float t = (u - pmin) / (pmax - pmin);
    return clampf(lerpf(lo, hi, t), lo, hi);
}

ZD bool windowedSample(const SatView& sat, float r0, float r1, const Bounds2& b, float& u, float& v, float& pdf) {
    const float integral = sat.integral(b);
    if (0.f == integral) return false;
    const SatEvaluatorX cdfx(sat, b, integral);
    const float         nx = float(sat.dim);
    u = sampleBisection(cdfx, r0, b.min0, b.max0, nx);
    Bounds2 bcond{floorf(u * nx) / nx, b.min1, ceilf(u * nx) / nx, b.max1};
    if (bcond.min0 == bcond.max0) bcond.max0 += 1.f / nx;
    const float cond_int = sat.integral(bcond);
    if (0.f == cond_int) return false;
    const SatEvaluatorY cdfy(sat, bcond, cond_int);
    v = sampleBisection(cdfy, r1, b.min1, b.max1, float(sat.dim));
    pdf = sat.eval(u, v) / integral;
    return true;
}

// the direction in the linked dome's frame, as its lat-long uv
ZD void portalUvw(const Trafo& dome_trafo, f3 dir, float& u, float& v) {
    const f3 xyz = normalize3(dome_trafo.transformVectorTransposed(dir));
    u = atan2F(xyz.x, xyz.z) * ((1.f / kPi) * 0.5f) + 0.5f;
    v = acosF(xyz.y) * (1.f / kPi);
}

// one sample of Rectangle.samplePortalTo from its 2-D draw; nsf = the light's sample count
ZD bool rectangleSamplePortal(const DPortalLight& pl, const Trafo& dome_trafo, const Bounds2& bounds, float r0, float r1, float nsf, f3 p, f3 n,
                              const Trafo& trafo, bool total_sphere, SampleTo& out) {
    const SatView sat{pl.sat, pl.func, int(pl.dim)};
    float         u, v, pdf;
    if (!windowedSample(sat, r0, r1, bounds, u, v, pdf) || 0.f == pdf) return false;
    // Portal.imageToWorld
    const float a = -kPi * 0.5f + u * kPi, b = -kPi * 0.5f + v * kPi;
    const f3    w = normalize3(mk3(tanF(a), tanF(b), 1.f));
    const float weight = (kPi * kPi) * (1.f - w.x * w.x) * (1.f - w.y * w.y) / w.z;
    const f3    dir = -trafo.objectToWorldNormal(w);
    if ((dot3(dir, n) <= 0.f && !total_sphere) || 0.f == weight) return false;
    const f3    wn = trafo.r2;
    const float d = dot3(wn, trafo.position);
    const float hit_t = -(dot3(wn, p) - d) / dot3(wn, dir);
    out.p = p + S3(hit_t) * dir;
    out.n = wn;
    out.wi = dir;
    out.pdf = (nsf * pdf) / weight;
    portalUvw(dome_trafo, dir, out.u, out.v);
    return true;
}

ZD float rectanglePortalPdf(const DPortalLight& pl, f3 dir, f3 p, const Trafo& frag_trafo, float nsf) {
    float u, v, weight;
    if (!portalWorldToImageWeighted(-dir, frag_trafo, u, v, weight)) return 0.f;
    Bounds2 b;
    if (!portalImageBounds(p, frag_trafo, b)) return 0.f;
    const SatView sat{pl.sat, pl.func, int(pl.dim)};
    const float   integral = sat.integral(b);
    const float   ppdf = 0.f == integral ? 0.f : sat.eval(u, v) / integral;
    return (ppdf * nsf) / weight;
}

// one sample of Rectangle.sampleMaterialTo (rectangle.zig:399-456) from its 2-D draw; nsf = the light's sample count
ZD bool rectangleSampleMaterial(const DImageLight& il, float r0, float r1, float nsf, f3 p, f3 n, const Trafo& trafo, bool two_sided,
                                bool total_sphere, SampleTo& out) {
    float       u, v;
    const float pdf = imageLightSample(il, r0, r1, u, v);
    if (0.f == pdf) return false;
    const float area = trafo.scale.x * trafo.scale.y;
    const f3    ws = trafo.objectToWorldPoint(mk3(-1.f * u + 0.5f, -1.f * v + 0.5f, 0.f));
    const f3    axis = ws - p;
    f3          wn = trafo.r2;
    if (two_sided && dot3(wn, axis) > 0.f) wn = -wn;
    const float sl = dot3(axis, axis);
    const float t = sqrtf(sl);
    const f3    dir = axis / S3(t);
    const float c = -dot3(wn, dir);
    if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) return false;
    out.p = ws;
    out.n = wn;
    out.wi = dir;
    out.pdf = (nsf * pdf * sl) / (c * area);
    out.u = u;
    out.v = v;
    return true;
}
ZD float rectangleMaterialPdf(const DImageLight& il, f3 dir, f3 p, const Trafo& trafo, f3 frag_p, float u, float v, uint32_t num_samples) {
    const float c = fabsf(dot3(trafo.r2, dir));
    const float area = trafo.scale.x * trafo.scale.y;
    const f3    dp = p - frag_p;
    const float sl = dot3(dp, dp);
    const float material_pdf = imageLightPdf(il, u, v) * float(num_samples);
    return (material_pdf * sl) / (c * area);
}

// one sample of Disk.sampleMaterialTo (disk.zig:361-426) from its 2-D draw; materialPdf (535-553)
ZD bool diskSampleMaterial(const DImageLight& il, float r0, float r1, float nsf, f3 p, f3 n, const Trafo& trafo, bool two_sided,
                           bool total_sphere, SampleTo& out) {
    float       u, v;
    const float pdf = imageLightSample(il, r0, r1, u, v);
    if (0.f == pdf) return false;
    const float radius = 0.5f * trafo.scale.x;
    const float area = kPi * (radius * radius);
    const f3    k = S3(radius) * trafo.objectToWorldNormal(mk3(-2.f * u + 1.f, -2.f * v + 1.f, 0.f));
    const float l = dot3(k, k);
    if (l > radius * radius) return false;
    const f3 ws = trafo.position + k;
    f3       wn = trafo.r2;
    if (two_sided && dot3(wn, ws - p) > 0.f) wn = -wn;
    const f3    axis = ws - p;
    const float sl = squaredLength3(axis);
    const float t = sqrtf(sl);
    const f3    dir = axis / S3(t);
    const float c = -dot3(wn, dir);
    if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) return false;
    out.p = ws;
    out.n = wn;
    out.wi = dir;
    out.pdf = (nsf * pdf * sl) / (c * area);
    out.u = u;
    out.v = v;
    return true;
}
ZD float diskMaterialPdf(const DImageLight& il, f3 dir, f3 p, const Trafo& trafo, f3 frag_p, float u, float v, uint32_t num_samples) {
    const float c = fabsf(dot3(trafo.r2, dir));
    const float radius = 0.5f * trafo.scale.x;
    const float area = kPi * (radius * radius);
    const f3    dp = p - frag_p;
    const float sl = dot3(dp, dp);
    const float material_pdf = imageLightPdf(il, u, v) * float(num_samples);
    return (material_pdf * sl) / (c * area);
}
// Sphere.sampleMaterialTo (sphere.zig:395-446): one sample of the lat-long emission map; materialPdf (489-502)
ZD bool sphereSampleMaterialTo(const DImageLight& il, f3 p, f3 n, const Trafo& trafo, bool total_sphere, Sampler& sampler, SampleTo& out) {
    float r0, r1;
    sampler.sample2D(r0, r1);
    float       u, v;
    const float pdf = imageLightSample(il, r0, r1, u, v);
    if (0.f == pdf) return false;
    const float phi = (u + 0.75f) * (2.f * kPi);
    const float theta = v * kPi;
    const float sin_theta = mx(sinF(theta), 0.00001f);  // avoid singularity at poles
    const float cos_theta = cosF(theta);
    const float sin_phi = sinF(phi), cos_phi = cosF(phi);
    const float r = 0.5f * trafo.scale.x;
    const f3    ws = trafo.objectToWorldPoint(S3(r) * mk3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi));
    const f3    axis = ws - p;
    const float sl = squaredLength3(axis);
    const float t = sqrtf(sl);
    const f3    dir = axis / S3(t);
    const f3    wn = normalize3(ws - trafo.position);
    const float c = -dot3(wn, dir);
    if (c < kDotMin || (dot3(dir, n) <= 0.f && !total_sphere)) return false;
    const float area = (4.f * kPi) * (r * r);
    out.p = ws;
    out.n = wn;
    out.wi = dir;
    out.pdf = (pdf * sl) / (c * area * sin_theta);
    out.u = u;
    out.v = v;
    return true;
}
ZD float sphereMaterialPdf(const DImageLight& il, f3 dir, f3 p, const Trafo& trafo, f3 frag_p, f3 frag_geo_n, float u, float v) {
    const float sin_theta = mx(sinF(v * kPi), 0.00001f);
    const f3    dp = p - frag_p;
    const float sl = dot3(dp, dp);
    const float c = -dot3(frag_geo_n, dir);
    const float r = 0.5f * trafo.scale.x;
    const float area = (4.f * kPi) * (r * r);
    return (imageLightPdf(il, u, v) * sl) / (c * area * sin_theta);
}

// ---------------------------------------------------------------- TriangleMesh lights (triangle_mesh.zig:391-703)

ZD f3 orthogonalize(f3 a, f3 b) { return normalize3(fma3(S3(-dot3(a, b)), a, b)); }  // a is normalised (391-394)

ZD void barycentricCoords(f3 dir, f3 a, f3 b, f3 c, float& u, float& v) {  // triangle.zig:82-100
    const f3    e1 = b - a, e2 = c - a, tvec = -a;
    const f3    pvec = cross3(dir, e2), qvec = cross3(tvec, e1);
    const float e1_d_pv = dot3(e1, pvec), tv_d_pv = dot3(tvec, pvec), di_d_qv = dot3(dir, qvec);
    const float inv_det = 1.f / e1_d_pv;
    u = tv_d_pv * inv_det;
    v = di_d_qv * inv_det;
}

// Stratified Sampling of Spherical Triangles, James Arvo (triangle_mesh.zig:402-459)
ZD bool sampleSpherical(f3 pos, f3 pa, f3 pb, f3 pc, float r0, float r1, f3& dir, float& bu, float& bv, float& pdf) {
    const f3 pap = pa - pos, pbp = pb - pos, pcp = pc - pos;
    const f3 A = normalize3(pap), B = normalize3(pbp), C = normalize3(pcp);
    const f3 BA = orthogonalize(A, B - A), CA = orthogonalize(A, C - A);
    const f3 AB = orthogonalize(B, A - B), CB = orthogonalize(B, C - B);
    const f3 BC = orthogonalize(C, B - C), AC = orthogonalize(C, A - C);
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
    const float sin_alpha = sqrtD(1.f - cos_alpha * cos_alpha);
    const float u = q - cos_alpha, v = p + sin_alpha * cos_c;
    const float sc = clampf(((v * q - u * p) * cos_alpha - v) / ((v * p + u * q) * sin_alpha), -1.f, 1.f);
    const f3    C_s = S3(sc) * A + S3(sqrtD(1.f - sc * sc)) * orthogonalize(A, C);
    const float cs_b = dot3(C_s, B);
    const float z = 1.f - r1 * (1.f - cs_b);
    const f3    P = S3(z) * B + S3(sqrtD(1.f - z * z)) * orthogonalize(B, C_s);
    dir = P;
    barycentricCoords(P, pap, pbp, pcp, bu, bv);
    pdf = 1.f / sarea;
    return true;
}

ZD float pdfSpherical(f3 pos, f3 pa, f3 pb, f3 pc) {  // triangle_mesh.zig:461-485
    const f3 A = normalize3(pa - pos), B = normalize3(pb - pos), C = normalize3(pc - pos);
    const f3 BA = orthogonalize(A, B - A), CA = orthogonalize(A, C - A);
    const f3 AB = orthogonalize(B, A - B), CB = orthogonalize(B, C - B);
    const f3 BC = orthogonalize(C, B - C), AC = orthogonalize(C, A - C);
    const float alpha = acosF(clampf(dot3(BA, CA), -1.f, 1.f));
    const float beta = acosF(clampf(dot3(AB, CB), -1.f, 1.f));
    const float gamma = acosF(clampf(dot3(BC, AC), -1.f, 1.f));
    const float sarea = alpha + beta + gamma - kPi;
    return 1.f / sarea;
}

ZD void triangleUniform(float u0, float u1, float& x, float& y) {  // sampling.zig:39-47 (Heitz)
    if (u1 > u0) {
        x = 0.5f * u0;
        y = u1 - x;
    } else {
        y = 0.5f * u1;
        x = u0 - y;
    }
}

constexpr float kAreaDistanceRatio = 0.001f;  // triangle_mesh.zig:487

// The object-space set-up of Mesh.sampleTo (triangle_mesh.zig:505-511) and its per-triangle body (518-592): one 2-D draw
// (r0, r1) per picked triangle; solid-angle sampling when the triangle is large for its distance, area sampling else.
struct MeshLightSampler {
    f3 op, on, scale_squared;

    ZD static MeshLightSampler init(f3 p, f3 n, const Trafo& trafo) {
        return {trafo.worldToObjectPoint(p), trafo.worldToObjectNormal(n), trafo.scale * trafo.scale};
    }

    ZD bool sample(const DMesh& mesh, const DMeshLight& ml, LightPick tri, float r0, float r1, f3 p, f3 n, const Trafo& trafo,
                   bool two_sided, bool total_sphere, SampleTo& out) const {
        const uint32_t global = ml.triangle_mapping[tri.offset];
        const f3       a = xyz(mesh.tri_verts[3 * global]), b = xyz(mesh.tri_verts[3 * global + 1]), c = xyz(mesh.tri_verts[3 * global + 2]);
        const f3       cross_axis = cross3(b - a, c - a);
        const f3       ca = scale_squared * cross_axis;
        const float    lca = length3(ca);
        const f3       sn = ca / S3(lca);
        f3             wn = trafo.objectToWorldNormal(sn);
        const float    tri_area = 0.5f * lca;
        const f3       center = (a + b + c) / S3(3.f);

        f3    dir, v;
        float bu, bv, sample_pdf, n_dot_dir;
        if (tri_area / distance3(center, op) > kAreaDistanceRatio) {
            f3    sdir;
            float spdf;
            if (!sampleSpherical(op, a, b, c, r0, r1, sdir, bu, bv, spdf)) return false;
            if (dot3(sdir, on) <= 0.f && !total_sphere) return false;
            dir = trafo.objectToWorldNormal(sdir);
            v = trafo.objectToWorldPoint(interpolate3(a, b, c, bu, bv));
            sample_pdf = tri.pdf * spdf;
            if (two_sided && dot3(wn, dir) > 0.f) wn = -wn;
            n_dot_dir = -dot3(wn, dir);
        } else {
            triangleUniform(r0, r1, bu, bv);
            v = trafo.objectToWorldPoint(interpolate3(a, b, c, bu, bv));
            const f3    axis = v - p;
            const float sl = squaredLength3(axis);
            const float dd = sqrtf(sl);
            dir = axis / S3(dd);
            if (dot3(dir, n) <= 0.f && !total_sphere) return false;
            if (two_sided && dot3(wn, dir) > 0.f) wn = -wn;
            n_dot_dir = -dot3(wn, dir);
            sample_pdf = (tri.pdf * sl) / (n_dot_dir * tri_area);
        }
        if (n_dot_dir < kDotMin) return false;
        out.p = v;
        out.n = wn;
        out.wi = dir;
        out.pdf = sample_pdf;
        return true;
    }
};

// Mesh.pdf (triangle_mesh.zig:647-691): the density sampleTo would have had for the triangle a ray hit
ZD float meshLightPdf(const DScene& s, const DMeshLight& ml, f3 dir, f3 p, f3 n, const Trafo& trafo, f3 frag_p, f3 frag_geo_n,
                      uint32_t primitive, bool total_sphere, float split_threshold) {
    const DMesh&    mesh = s.meshes[ml.mesh];
    const float     n_dot_dir = fabsf(dot3(frag_geo_n, dir));
    const f3        op = trafo.worldToObjectPoint(p), on = trafo.worldToObjectNormal(n);
    const uint32_t  pm = mesh.primitive_mapping[primitive];
    const LightTree tree{s, ml.tree};
    const float     tri_pdf = tree.pdf(op, on, total_sphere, split_threshold, pm);
    const f3        a = xyz(mesh.tri_verts[3 * primitive]), b = xyz(mesh.tri_verts[3 * primitive + 1]), c = xyz(mesh.tri_verts[3 * primitive + 2]);
    const f3        ca = (trafo.scale * trafo.scale) * cross3(b - a, c - a);
    const float     tri_area = 0.5f * length3(ca);
    const f3        center = (a + b + c) / S3(3.f);
    if (tri_area / distance3(center, op) > kAreaDistanceRatio) return tri_pdf * pdfSpherical(op, a, b, c);
    const float sl = squaredLength3(p - frag_p);
    return (tri_pdf * sl) / (n_dot_dir * tri_area);
}

// shape.zig:401-416
ZD void shadowRay(f3 origin, const SampleTo& sample, f3& dir, float& len, bool infinite_shape = false) {
    if (infinite_shape) {
        dir = sample.wi;
        len = kRayMaxT;
        return;
    }
    const f3 light_pos = offsetRay(sample.p, sample.n);
    const f3 shadow_axis = light_pos - origin;
    len = length3(shadow_axis);
    dir = shadow_axis / S3(len);
}

// emittance.zig:29-59; TEX: the scene can have emission maps and profiles, (u, v) = rs.uv(), r = rs.stochastic_r, origin = rs.origin
// (the point the light is seen from: the previous vertex for a ray that hits the light, the shaded point for a light sample)
template <bool TEX>
ZD f3 emittanceRadiance(const DScene& s, const ZygpuMaterial& m, f3 wi, const Trafo& trafo, float area, bool in_camera, float u, float v, float r, f3 origin) {
    if (-dot3(wi, trafo.r2) < m.cos_a) return S3(0.f);
    float factor = in_camera ? m.camera_weight : 1.f;
    if (TEX && kNull != m.profile_image) {  // the profile over the octahedral map of the direction light -> origin in the light's space (:45-52)
        const f3    lwi = -normalize3(trafo.worldToObjectPoint(origin));
        const float inorm = 1.f / (fabsf(lwi.x) + fabsf(lwi.y) + fabsf(lwi.z));  // octEncode (base/encoding.zig:82-87)
        const float t = mx(lwi.z, 0.f);
        const float ox = (lwi.x + (lwi.x > 0.f ? t : -t)) * inorm, oy = (lwi.y + (lwi.y > 0.f ? t : -t)) * inorm;
        factor *= sampleImageMap1(s, m.profile_image, m.profile_sampler, m.profile_scale, (ox + 1.f) * 0.5f, (oy + 1.f) * 0.5f, r);
    }
    const f3    texel = (TEX && kNull != m.emission_image) ? sampleImageMap(s, m.emission_image, m.emission_sampler, m.emission_scale, u, v, r)
                                                           : S3(1.f);
    const f3    intensity = mk3(m.emission[0], m.emission[1], m.emission[2]) * texel;
    if (m.emission_normalize) return S3(factor / area) * intensity;
    return S3(factor) * intensity;
}

}  // namespace zd
