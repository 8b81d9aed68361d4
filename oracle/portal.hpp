// Portal lights (test infrastructure, like everything under oracle/): a Rectangle that is a window onto an environment map.
//   Portal.imageToWorld / worldToImageWeighted / imageBounds      scene/shape/portal.zig:16-80
//   SummedAreaTable.integral / lookup, EvaluatorX / EvaluatorY      base/math/summed_area_table.zig:59-289
//   WindowedDistribution2D.sampleContinuous / pdf / eval            base/math/windowed_distribution_2d.zig:31-105
//   PortalImpl.portalUvw / sample / pdf                             scene/shape/shape_sampler.zig:263-299
//   Rectangle.samplePortalTo / portalPdf                            scene/shape/rectangle.zig:458-513, 597-611
#pragma once

#include "light.hpp"

namespace orc {

struct Bounds2 {
    float min[2], max[2];
};

struct PortalWorldSample {
    V4    dir;
    float weight;
};

inline PortalWorldSample portalImageToWorld(float u, float v, const Trafo& trafo) {  // portal.zig:16-25
    const float a = -kPi * 0.5f + u * kPi, b = -kPi * 0.5f + v * kPi;
    const float x = tanF(a), y = tanF(b);
    const V4    w = normalize3(mk(x, y, 1.f, 0.f));
    const float weight = (kPi * kPi) * (1.f - w[0] * w[0]) * (1.f - w[1] * w[1]) / w[2];
    return {trafo.objectToWorldNormal(w), weight};
}

inline bool portalWorldToImage(V4 dir, const Trafo& trafo, float& u, float& v) {  // portal.zig:27-39
    const V4 w = trafo.worldToObjectNormal(dir);
    if (w[2] <= 0.f) return false;
    const float a = atan2F(w[0], w[2]), b = atan2F(w[1], w[2]);
    u = clampf((a + kPi / 2.f) / kPi, 0.f, 1.f);
    v = clampf((b + kPi / 2.f) / kPi, 0.f, 1.f);
    return true;
}

inline bool portalWorldToImageWeighted(V4 dir, const Trafo& trafo, float& u, float& v, float& weight) {  // portal.zig:46-64
    const V4 w = trafo.worldToObjectNormal(dir);
    if (w[2] <= 0.f) return false;
    const float alpha = atan2F(w[0], w[2]), beta = atan2F(w[1], w[2]);
    weight = (kPi * kPi) * (1.f - w[0] * w[0]) * (1.f - w[1] * w[1]) / w[2];
    u = clampf((alpha + kPi / 2.f) / kPi, 0.f, 1.f);
    v = clampf((beta + kPi / 2.f) / kPi, 0.f, 1.f);
    return true;
}

inline bool portalImageBounds(V4 p, const Trafo& trafo, Bounds2& out) {  // portal.zig:66-80
    const V4 a = trafo.r[0] * S(trafo.r[0][3]), b = trafo.r[1] * S(trafo.r[1][3]);
    const V4 ab = a + b;
    const V4 o = p - trafo.position;
    const V4 c0 = S(0.5f) * ab + o, c1 = S(-0.5f) * ab + o;
    float    u0, v0, u1, v1;
    if (!portalWorldToImage(normalize3(c0), trafo, u0, v0)) return false;
    if (!portalWorldToImage(normalize3(c1), trafo, u1, v1)) return false;
    out.min[0] = mn(u0, u1);
    out.min[1] = mn(v0, v1);
    out.max[0] = mx(u0, u1);
    out.max[1] = mx(v0, v1);
    return true;
}

// ---- summed_area_table.zig
struct SatView {
    const float* buffer;  // (dim + 1)^2 entries, first row and column zero
    const float* func;    // dim^2: the tabulated function
    int32_t      dim;
    float lookup(int32_t x0, int32_t x1, int32_t y0, int32_t y1, float dx, float dy) const {  // :114-126
        const int32_t w = dim + 1;
        const float   v00 = buffer[x0 + y0 * w], v10 = buffer[x1 + y0 * w], v01 = buffer[x0 + y1 * w], v11 = buffer[x1 + y1 * w];
        const float   odx = 1.f - dx, ody = 1.f - dy;
        return ody * (odx * v00 + dx * v10) + dy * (odx * v01 + dx * v11);
    }
    float integral(const Bounds2& e) const {  // :59-112
        const float   fd = float(dim);
        const float   ax = e.min[0] * fd, ay = e.min[1] * fd, bx = e.max[0] * fd, by = e.max[1] * fd;
        const int32_t ax0 = int32_t(ax), ay0 = int32_t(ay), bx0 = int32_t(bx), by0 = int32_t(by);
        const int32_t axx0 = std::min(ax0, dim), ayy0 = std::min(ay0, dim), axx1 = std::min(ax0 + 1, dim), ayy1 = std::min(ay0 + 1, dim);
        const int32_t bxx0 = std::min(bx0, dim), byy0 = std::min(by0, dim), bxx1 = std::min(bx0 + 1, dim), byy1 = std::min(by0 + 1, dim);
        const float   adx = ax - float(ax0), ady = ay - float(ay0), bdx = bx - float(bx0), bdy = by - float(by0);
        const float   s = (lookup(bxx0, bxx1, byy0, byy1, bdx, bdy) - lookup(axx0, axx1, byy0, byy1, adx, bdy)) +
                        (lookup(axx0, axx1, ayy0, ayy1, adx, ady) - lookup(bxx0, bxx1, ayy0, ayy1, bdx, ady));
        return mx(s / (fd * fd), 0.f);
    }
    float eval(float u, float v) const {  // windowed_distribution_2d.zig:94-105
        const uint32_t x = uint32_t(u * float(dim)), y = uint32_t(v * float(dim));
        return func[std::min(y, uint32_t(dim - 1)) * uint32_t(dim) + std::min(x, uint32_t(dim - 1))];
    }
};

struct SatEvaluatorX {  // :128-208
    const SatView* sat;
    int32_t        ay0, by0;
    float          lookup_b, lookup_c, ady, bdy, integral_area;
    SatEvaluatorX(const SatView& s, const Bounds2& e, float integral) : sat(&s) {
        const int32_t d = s.dim;
        const float   fd = float(d);
        const float   ax = e.min[0] * fd, ay = e.min[1] * fd, by = e.max[1] * fd;
        const int32_t ax0 = int32_t(ax);
        ay0 = int32_t(ay);
        by0 = int32_t(by);
        const int32_t axx0 = std::min(ax0, d), ayy0 = std::min(ay0, d), axx1 = std::min(ax0 + 1, d), ayy1 = std::min(ay0 + 1, d);
        const int32_t byy0 = std::min(by0, d), byy1 = std::min(by0 + 1, d);
        const float   adx = ax - float(ax0);
        ady = ay - float(ay0);
        bdy = by - float(by0);
        lookup_b = s.lookup(axx0, axx1, byy0, byy1, adx, bdy);
        lookup_c = s.lookup(axx0, axx1, ayy0, ayy1, adx, ady);
        integral_area = integral * (fd * fd);
    }
    float eval(float x) const {
        const int32_t d = sat->dim;
        const float   bx = x * float(d);
        const int32_t bx0 = int32_t(bx);
        const int32_t ayy0 = std::min(ay0, d), ayy1 = std::min(ay0 + 1, d);
        const int32_t bxx0 = std::min(bx0, d), byy0 = std::min(by0, d), bxx1 = std::min(bx0 + 1, d), byy1 = std::min(by0 + 1, d);
        const float   bdx = bx - float(bx0);
        const float   lookup_a = sat->lookup(bxx0, bxx1, byy0, byy1, bdx, bdy), lookup_d = sat->lookup(bxx0, bxx1, ayy0, ayy1, bdx, ady);
        const float   s = (lookup_a - lookup_b) + (lookup_c - lookup_d);
        return mx(s / integral_area, 0.f);
    }
};

struct SatEvaluatorY {  // :210-289
    const SatView* sat;
    int32_t        ax0, bx0;
    float          lookup_c, lookup_d, adx, bdx, integral_area;
    SatEvaluatorY(const SatView& s, const Bounds2& e, float integral) : sat(&s) {
        const int32_t d = s.dim;
        const float   fd = float(d);
        const float   ax = e.min[0] * fd, ay = e.min[1] * fd, bx = e.max[0] * fd;
        ax0 = int32_t(ax);
        const int32_t ay0 = int32_t(ay);
        bx0 = int32_t(bx);
        const int32_t axx0 = std::min(ax0, d), ayy0 = std::min(ay0, d), axx1 = std::min(ax0 + 1, d), ayy1 = std::min(ay0 + 1, d);
        const int32_t bxx0 = std::min(bx0, d), bxx1 = std::min(bx0 + 1, d);
        adx = ax - float(ax0);
        const float ady = ay - float(ay0);
        bdx = bx - float(bx0);
        lookup_c = s.lookup(axx0, axx1, ayy0, ayy1, adx, ady);
        lookup_d = s.lookup(bxx0, bxx1, ayy0, ayy1, bdx, ady);
        integral_area = integral * (fd * fd);
    }
    float eval(float y) const {
        const int32_t d = sat->dim;
        const float   by = y * float(d);
        const int32_t by0 = int32_t(by);
        const int32_t axx0 = std::min(ax0, d), axx1 = std::min(ax0 + 1, d);
        const int32_t bxx0 = std::min(bx0, d), byy0 = std::min(by0, d), bxx1 = std::min(bx0 + 1, d), byy1 = std::min(by0 + 1, d);
        const float   bdy = by - float(by0);
        const float   lookup_a = sat->lookup(bxx0, bxx1, byy0, byy1, bdx, bdy), lookup_b = sat->lookup(axx0, axx1, byy0, byy1, adx, bdy);
        const float   s = (lookup_a - lookup_b) + (lookup_c - lookup_d);
        return mx(s / integral_area, 0.f);
    }
};

template <typename Cdf>
inline float sampleBisection(const Cdf& cdf, float u, float imin, float imax, float nf) {  // windowed_distribution_2d.zig:75-92
    float min = imin, max = imax;
    while (std::ceil(nf * max) - std::floor(nf * min) > 1.f) {
        const float mid = (min + max) / 2.f;
        if (cdf.eval(mid) > u) max = mid;
        else min = mid;
    }
    const float pmin = cdf.eval(min), pmax = cdf.eval(max);
    const float t = (u - pmin) / (pmax - pmin);
    return clampf(lerpf(min, max, t), min, max);
}

// WindowedDistribution2D.sampleContinuous: false = pdf 0
inline bool windowedSample(const SatView& sat, float r0, float r1, const Bounds2& b, float& u, float& v, float& pdf) {
    const float integral = sat.integral(b);
    if (0.f == integral) return false;
    const SatEvaluatorX cdfx(sat, b, integral);
    const float         nx = float(sat.dim);
    u = sampleBisection(cdfx, r0, b.min[0], b.max[0], nx);
    Bounds2 bcond{{std::floor(u * nx) / nx, b.min[1]}, {std::ceil(u * nx) / nx, b.max[1]}};
    if (bcond.min[0] == bcond.max[0]) bcond.max[0] += 1.f / nx;
    const float cond_int = sat.integral(bcond);
    if (0.f == cond_int) return false;
    const SatEvaluatorY cdfy(sat, bcond, cond_int);
    v = sampleBisection(cdfy, r1, b.min[1], b.max[1], float(sat.dim));
    pdf = sat.eval(u, v) / integral;
    return true;
}

inline float windowedPdf(const SatView& sat, float u, float v, const Bounds2& b) {
    const float integral = sat.integral(b);
    if (0.f == integral) return 0.f;
    return sat.eval(u, v) / integral;
}

inline SatView satView(const ZygpuPortalLight& pl) { return SatView{pl.sat, pl.func, int32_t(pl.dim)}; }

// PortalImpl.portalUvw (shape_sampler.zig:274-286): the direction in the linked dome's frame, as its lat-long uv
inline void portalUvw(const Trafo& dome_trafo, V4 dir, float& u, float& v) {
    const V4 xyz = normalize3(dome_trafo.transformVectorTransposed(dir));
    u = atan2F(xyz[0], xyz[2]) * ((1.f / kPi) * 0.5f) + 0.5f;
    v = acosF(xyz[1]) * (1.f / kPi);
}

// Rectangle.samplePortalTo (rectangle.zig:458-513)
inline uint32_t rectangleSamplePortalTo(const ZygpuPortalLight& pl, const Trafo& dome_trafo, V4 p, V4 n, const Trafo& trafo, bool total_sphere,
                                        float split_threshold, uint32_t light_num_samples, Sampler& sampler, SampleTo* buffer) {
    Bounds2 bounds;
    if (!portalImageBounds(p, trafo, bounds)) return 0;
    const SatView  sat = satView(pl);
    const uint32_t num_samples = numSamples(light_num_samples, split_threshold);
    const float    nsf = float(num_samples);
    uint32_t       current = 0;
    for (uint32_t i = 0; i < num_samples; ++i) {
        float r2[2];
        sampler.sample2D(r2);
        float u, v, pdf;
        if (!windowedSample(sat, r2[0], r2[1], bounds, u, v, pdf) || 0.f == pdf) continue;
        const PortalWorldSample ps = portalImageToWorld(u, v, trafo);
        const V4                dir = -ps.dir;
        if ((dot3(dir, n) <= 0.f && !total_sphere) || 0.f == ps.weight) continue;
        const V4    wn = trafo.r[2];
        const float d = dot3(wn, trafo.position);
        const float hit_t = -(dot3(wn, p) - d) / dot3(wn, dir);
        const V4    ws = p + S(hit_t) * dir;
        float       du, dv;
        portalUvw(dome_trafo, dir, du, dv);
        buffer[current].p = mk(ws[0], ws[1], ws[2], (nsf * pdf) / ps.weight);
        buffer[current].n = wn;
        buffer[current].wi = dir;
        buffer[current].uvw = mk(du, dv, 0.f, 0.f);
        current += 1;
    }
    return current;
}

// Rectangle.portalPdf (rectangle.zig:597-611)
inline float rectanglePortalPdf(const ZygpuPortalLight& pl, V4 dir, V4 p, const Trafo& frag_trafo, float split_threshold, uint32_t light_num_samples) {
    float u, v, weight;
    if (!portalWorldToImageWeighted(-dir, frag_trafo, u, v, weight)) return 0.f;
    Bounds2 b;
    if (!portalImageBounds(p, frag_trafo, b)) return 0.f;
    const float ppdf = windowedPdf(satView(pl), u, v, b);
    return (ppdf * float(numSamples(light_num_samples, split_threshold))) / weight;
}

}  // namespace orc
