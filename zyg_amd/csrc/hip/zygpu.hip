// MI355X (gfx950) wavefront path-tracing core: kernels + the zygpu_* C ABI (include/zygpu.h).
//
// One batch = `B` iterations x every pixel of the row range, one path per (pixel, iteration), path slot ==
// sample index.  Per tick two launches run over device-side queues of slot indices:
//   wf_trace : closest-hit for the extension queue (Scene.intersect, reference prop_tree.zig:56-116) and any-hit for
//              the shadow queue (Scene.visibility, prop_tree.zig:185-240)
//   wf_shade : the integrator state machine (pathtracer*.zig li loops) up to the next ray it needs
// and the film is accumulated from a per-sample staging buffer in iteration order (no float atomics, so the film
// is deterministic and, for the radius-0 sensor, bit-identical to sequential accumulation).
//
// The reference draws a sampler dimension *after* each successful shadow test (light/light.zig:127), so the sampler
// state after a vertex depends on visibility results; shading is therefore a resumable state machine
// (PH_HIT -> [PH_SHADOW]* -> next extension ray) rather than "emit all shadow rays, then extend".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dmaterial.hpp"

using namespace zd;

namespace {

// ---------------------------------------------------------------- device-side structures

enum : uint32_t { PH_HIT = 0, PH_SHADOW = 1 };

// meta word: [0:2] sobol dimension, [3:7] depth_surface, [8:12] vertex.depth, [13:18] state flags, [19] phase, [20:26] pick
enum : uint32_t {
    ST_PRIMARY = 1u << 0,
    ST_TRANSPARENT = 1u << 1,
    ST_SINGULAR = 1u << 2,
    ST_SPECULAR = 1u << 3,
    ST_TRANSLUCENT = 1u << 4,
    ST_STARTED_SPECULAR = 1u << 5
};

struct Pool {
    uint4*    q0;  // run_seed, meta, buffer[4] bits, -
    float4*   q1;  // sobol buffer[0..3]
    uint2*    pcg; // PCG state
    float4*   ray_o;   // origin.xyz, max_t
    float4*   ray_d;   // direction.xyz, bxdf_pdf
    float4*   thr;     // throughput.xyz, light select
    float4*   vorigin; // vertex.origin
    float4*   vgeo_n;  // vertex.geo_n
    float4*   emission;
    float4*   direct;
    float4*   indirect;
    uint4*    hit;     // t, u, v (bits), primitive
    uint32_t* hit_prop;
    float4*   sh_o;    // shadow origin.xyz, max_t
    float4*   sh_d;    // shadow direction.xyz
    uint32_t* visible;
    float4*   contrib; // pre-weighted NEE contribution of the pending shadow ray
    float4*   nee;     // running sum of this vertex's NEE
};

struct Queues {
    uint32_t* ext[2];
    uint32_t* sh[2];
};

struct Control {  // device-resident counters
    uint32_t           ext_count[2];
    uint32_t           sh_count[2];
    uint32_t           error;  // bit0: traversal stack overflow
    uint32_t           pad[3];
    unsigned long long stat_nodes, stat_tris, stat_instances, stat_ext, stat_shadow, stat_samples;
};

struct View {
    ZygpuViewDesc v;
    int32_t       y_begin, y_end; // sample rows incl. filter padding
    int32_t       x_begin, x_end; // sample columns incl. filter padding
    uint32_t      frame, iteration0, batch_iterations;
    uint32_t      npix;           // samples per iteration in this batch
};

struct Params {
    DScene   scene;
    Pool     pool;
    Queues   q;
    Control* ctl;
    float4*  stage_color; // per sample: composed rgb (clamped sum), alpha
    float2*  stage_uv;    // per sample: pixel_uv
    float4*  film;
    uint32_t parity;      // current queue side
};

__constant__ View c_view;

// ---------------------------------------------------------------- helpers

ZD uint32_t laneId() { return __lane_id(); }

// ordered append of the active lanes' values to a queue: one atomic per wave
ZD void queuePush(uint32_t* queue, uint32_t* count, bool active, uint32_t value) {
    const unsigned long long mask = __ballot(active);
    if (0 == mask) return;
    const uint32_t lane = laneId();
    const uint32_t leader = uint32_t(__ffsll((long long)mask) - 1);
    uint32_t       base = 0;
    if (lane == leader) base = atomicAdd(count, uint32_t(__popcll(mask)));
    base = __shfl(base, int(leader));
    if (active) {
        const uint32_t offset = uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
        queue[base + offset] = value;
    }
}

ZD void slotPixel(uint32_t slot, int& x, int& y, uint32_t& iteration) {
    const uint32_t it = slot / c_view.npix, pix = slot - it * c_view.npix;
    const uint32_t w = uint32_t(c_view.x_end - c_view.x_begin);
    y = c_view.y_begin + int(pix / w);
    x = c_view.x_begin + int(pix % w);
    iteration = c_view.iteration0 + it;
}

ZD uint32_t pixelId(int x, int y) {  // worker.zig:127-128,136,141
    const int fr = c_view.v.filter_radius_int;
    const int r0 = c_view.v.resolution[0] + 2 * fr;
    return uint32_t((y + fr) * r0) + uint32_t(x + fr);
}

ZD uint64_t pcgSequence(uint32_t pixel_id, uint32_t iteration) {  // worker.zig:129-130,143
    const int      fr = c_view.v.filter_radius_int;
    const uint32_t a = uint32_t(c_view.v.resolution[0] + 2 * fr) * uint32_t(c_view.v.resolution[1] + 2 * fr);
    return uint64_t(pixel_id) + uint64_t(iteration) * a;
}

ZD float splitThreshold(uint32_t total_depth) {  // integrator/helper.zig:33-39
    const float t = c_view.v.split_threshold;
    return mn(total_depth < 4 ? t : kLowThreshold, t);
}

ZD float powerHeuristic(float f_pdf, float g_pdf) {  // helper.zig:64-67
    const float f2 = f_pdf * f_pdf;
    return f2 / __fmaf_rn(g_pdf, g_pdf, f2);
}
ZD float predividedPowerHeuristic(float f_pdf, float g_pdf) {  // helper.zig:70-73
    const float f2 = f_pdf * f_pdf;
    return f_pdf / __fmaf_rn(g_pdf, g_pdf, f2);
}

struct PathState {  // register image of one path between yields
    Sampler  smp;
    uint32_t depth_surface, vdepth, flags, pick;
    Ray      ray;
    float    bxdf_pdf, select;
    f3       throughput, vorigin, vgeo_n, emission, direct, indirect, nee;
};

ZD void loadSampler(const Params& P, uint32_t slot, uint32_t pixel_id, uint32_t iteration, PathState& st, uint32_t& meta) {
    const uint4  a = P.pool.q0[slot];
    const float4 b = P.pool.q1[slot];
    const uint2  g = P.pool.pcg[slot];
    meta = a.y;
    st.smp.run_seed = a.x;
    st.smp.buffer[0] = b.x;
    st.smp.buffer[1] = b.y;
    st.smp.buffer[2] = b.z;
    st.smp.buffer[3] = b.w;
    st.smp.buffer[4] = __uint_as_float(a.z);
    st.smp.dimension = meta & 7u;
    const uint64_t sample_index = uint64_t(pixel_id) * uint64_t(c_view.v.num_expected_samples) + uint64_t(iteration);
    st.smp.sample = uint32_t(sample_index);
    st.smp.rng.state = uint64_t(g.x) | (uint64_t(g.y) << 32);
    st.smp.rng.inc = (pcgSequence(pixel_id, iteration) << 1) | 1ull;
    st.smp.sobol_kind = ZYGPU_SAMPLER_SOBOL == c_view.v.sampler;
}

ZD uint32_t packMeta(const PathState& st, uint32_t phase) {
    return (st.smp.dimension & 7u) | ((st.depth_surface & 31u) << 3) | ((st.vdepth & 31u) << 8) | ((st.flags & 63u) << 13) |
           ((phase & 1u) << 19) | ((st.pick & 127u) << 20);
}

ZD void storeSampler(const Params& P, uint32_t slot, const PathState& st, uint32_t phase) {
    P.pool.q0[slot] = make_uint4(st.smp.run_seed, packMeta(st, phase), __float_as_uint(st.smp.buffer[4]), 0u);
    P.pool.q1[slot] = make_float4(st.smp.buffer[0], st.smp.buffer[1], st.smp.buffer[2], st.smp.buffer[3]);
    P.pool.pcg[slot] = make_uint2(uint32_t(st.smp.rng.state), uint32_t(st.smp.rng.state >> 32));
}

// ---------------------------------------------------------------- K0: generate
// Worker.render seeding (worker.zig:141-149) + Sensor.cameraSample (sensor.zig:152-166) +
// Perspective.generateVertex (camera_perspective.zig:124-150)
__global__ void __launch_bounds__(256) wf_generate(Params P, uint32_t num_paths) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= num_paths) return;

    int      x, y;
    uint32_t iteration;
    slotPixel(slot, x, y, iteration);
    const uint32_t pixel_id = pixelId(x, y);

    PathState st;
    st.smp.sobol_kind = ZYGPU_SAMPLER_SOBOL == c_view.v.sampler;
    st.smp.use_primary = true;
    st.smp.rng.start(0, pcgSequence(pixel_id, iteration));

    const uint32_t nexp = c_view.v.num_expected_samples;
    const uint64_t sample_index = uint64_t(pixel_id) * uint64_t(nexp) + uint64_t(iteration);
    const uint32_t seed = uint32_t(sample_index >> 32) + iteration / nexp;
    st.smp.buffer[0] = st.smp.buffer[1] = st.smp.buffer[2] = st.smp.buffer[3] = st.smp.buffer[4] = 0.f;
    st.smp.startPixel(uint32_t(sample_index), seed);

    float s0, s1, s2, s3;
    st.smp.sample4D(s0, s1, s2, s3);
    const float s_time = st.smp.sample1D();
    (void)s_time;  // static scenes: the shutter time only selects keyframes (space.zig:99-108)
    st.smp.incrementPadding();

    const float cx = float(x) + s0, cy = float(y) + s1;
    const View& vw = c_view;
    f3 direction = mk3(vw.v.left_top[0], vw.v.left_top[1], vw.v.left_top[2]) + mk3(vw.v.d_x[0], vw.v.d_x[1], vw.v.d_x[2]) * S3(cx) +
                   mk3(vw.v.d_y[0], vw.v.d_y[1], vw.v.d_y[2]) * S3(cy);
    f3 origin;
    if (vw.v.lens_radius > 0.f) {
        float lx, ly;
        diskConcentric(s2, s3, lx, ly);
        origin = mk3(lx * vw.v.lens_radius, ly * vw.v.lens_radius, 0.f);
        const f3 focus = S3(vw.v.focus_distance / direction.z) * direction;
        direction = focus - origin;
    } else {
        origin = mk3(vw.v.eye_offset[0], vw.v.eye_offset[1], vw.v.eye_offset[2]);
    }
    Trafo ct;
    ct.r0 = mk3(vw.v.camera_trafo.r[0][0], vw.v.camera_trafo.r[0][1], vw.v.camera_trafo.r[0][2]);
    ct.r1 = mk3(vw.v.camera_trafo.r[1][0], vw.v.camera_trafo.r[1][1], vw.v.camera_trafo.r[1][2]);
    ct.r2 = mk3(vw.v.camera_trafo.r[2][0], vw.v.camera_trafo.r[2][1], vw.v.camera_trafo.r[2][2]);
    ct.scale = mk3(vw.v.camera_trafo.r[0][3], vw.v.camera_trafo.r[1][3], vw.v.camera_trafo.r[2][3]);
    ct.position = mk3(vw.v.camera_trafo.position[0], vw.v.camera_trafo.position[1], vw.v.camera_trafo.position[2]);
    const f3 origin_w = ct.objectToWorldPoint(origin);
    const f3 direction_w = ct.objectToWorldVector(normalize3(direction));

    // Vertex.init (vertex.zig:67-85)
    st.depth_surface = 0;
    st.vdepth = 0;
    st.flags = ST_PRIMARY | ST_TRANSPARENT | ST_SINGULAR;
    st.pick = 0;
    storeSampler(P, slot, st, PH_HIT);
    P.pool.ray_o[slot] = make_float4(origin_w.x, origin_w.y, origin_w.z, kRayMaxT);
    P.pool.ray_d[slot] = make_float4(direction_w.x, direction_w.y, direction_w.z, 0.f);
    P.pool.thr[slot] = make_float4(1.f, 1.f, 1.f, 0.f);
    P.pool.vorigin[slot] = make_float4(origin_w.x, origin_w.y, origin_w.z, 0.f);
    P.pool.vgeo_n[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    P.pool.emission[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    P.pool.direct[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    P.pool.indirect[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    P.stage_uv[slot] = make_float2(s0, s1);
    P.q.ext[P.parity][slot] = slot;
    if (0 == slot) {
        P.ctl->ext_count[P.parity] = num_paths;
        P.ctl->sh_count[P.parity] = 0;
    }
}

// ---------------------------------------------------------------- K1/K3: trace
template <bool STATS>
__global__ void __launch_bounds__(256) wf_trace(Params P) {
    const uint32_t cur = P.parity;
    const uint32_t n_ext = P.ctl->ext_count[cur], n_sh = P.ctl->sh_count[cur];
    const uint32_t total = n_ext + n_sh;
    if (0 == blockIdx.x && 0 == threadIdx.x) {  // the shade launch of this tick fills the other side
        P.ctl->ext_count[cur ^ 1u] = 0;
        P.ctl->sh_count[cur ^ 1u] = 0;
    }
    Counters cnt{0, 0, 0};
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        if (i < n_ext) {
            const uint32_t slot = P.q.ext[cur][i];
            const float4   o = P.pool.ray_o[slot], d = P.pool.ray_d[slot];
            const uint32_t depth = (P.pool.q0[slot].y >> 3) & 31u;
            Ray            ray = Ray::init(mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), 0.f, o.w);
            Hit            hit;
            sceneIntersect<STATS>(P.scene, ray, depth, hit, cnt);
            if (kNull == hit.prop) hit.t = hit.u = hit.v = 0.f, hit.primitive = 0;
            P.pool.hit[slot] = make_uint4(__float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v), hit.primitive);
            P.pool.hit_prop[slot] = hit.prop;
        } else {
            const uint32_t slot = P.q.sh[cur][i - n_ext];
            const float4   o = P.pool.sh_o[slot], d = P.pool.sh_d[slot];
            Ray            ray = Ray::init(mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), 0.f, o.w);
            P.pool.visible[slot] = sceneVisibility<STATS>(P.scene, ray, cnt) ? 1u : 0u;
        }
    }
    if (STATS) {
        atomicAdd(&P.ctl->stat_nodes, cnt.nodes);
        atomicAdd(&P.ctl->stat_tris, cnt.tris);
        atomicAdd(&P.ctl->stat_instances, cnt.instances);
        if (0 == blockIdx.x && 0 == threadIdx.x) {
            atomicAdd(&P.ctl->stat_ext, (unsigned long long)n_ext);
            atomicAdd(&P.ctl->stat_shadow, (unsigned long long)n_sh);
        }
    }
}

// ---------------------------------------------------------------- K2: shade

struct ShadeCtx {
    const Params&  P;
    const DScene&  s;
    Counters&      cnt;
};

ZD float shapeArea(const DScene& s, uint32_t prop, uint32_t part, const Trafo& trafo) {  // shape.zig:132-152
    const ZygpuProp p = s.props[prop];
    if (ZYGPU_SHAPE_RECTANGLE == p.shape) return trafo.scale.x * trafo.scale.y;
    return s.meshes[p.mesh].part_areas[part] * (trafo.scale.x * trafo.scale.y);
}

ZD f3 materialRadiance(const DScene& s, const ZygpuMaterial& m, f3 wi, const Trafo& trafo, uint32_t prop, uint32_t part,
                       bool in_camera) {
    if (ZYGPU_MAT_LIGHT != m.kind && ZYGPU_MAT_SUBSTITUTE != m.kind) return S3(0.f);
    const float area = m.emission_normalize ? shapeArea(s, prop, part, trafo) : 1.f;
    return emittanceRadiance(m, wi, trafo.r2, area, in_camera);
}

// Scene.lightPdf (scene.zig:624-634)
ZD float lightPdf(const DScene& s, const PathState& st, const Fragment& frag, float light_split_threshold) {
    const uint32_t light_id = s.light_ids[s.props[frag.prop].parts_start + frag.part];
    if ((st.flags & ST_SINGULAR) || kNull == light_id) return 1.f;
    const LightTree lt{s};
    const float     select_pdf = lt.pdf(st.vorigin, st.vgeo_n, 0 != (st.flags & ST_TRANSLUCENT), light_split_threshold, light_id);
    const float     sample_pdf = rectanglePdf(st.vorigin, frag.trafo);
    return powerHeuristic(st.bxdf_pdf, sample_pdf * select_pdf);
}

// Vertex.evaluateRadiance (vertex.zig:183-212)
ZD f3 evaluateRadiance(const DScene& s, PathState& st, const Fragment& frag, float light_split_threshold) {
    const f3             wo = -st.ray.direction;
    const ZygpuMaterial& m = s.materials[s.material_ids[s.props[frag.prop].parts_start + frag.part]];
    if (!(m.flags & ZYGPU_MATF_EMISSIVE) || (!(m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo))) return S3(0.f);
    (void)st.smp.sample1D();  // rs.stochastic_r
    const bool  in_camera = 0 == st.depth_surface;
    const f3    energy = materialRadiance(s, m, wo, frag.trafo, frag.prop, frag.part, in_camera);
    const float weight = lightPdf(s, st, frag, light_split_threshold);
    return S3(weight) * energy;
}

// PropBvh.emission over the unoccluding TLAS (prop_tree.zig:302-356, prop.zig:239-264, rectangle.zig:188-196)
template <bool STATS>
ZD f3 unoccludingEmission(const DScene& s, PathState& st, float light_split_threshold, Counters& cnt) {
    f3  energy = S3(0.f);
    Ray ray = st.ray;
    bvhWalk<STATS>(s.unocc_nodes, s.num_unocc_nodes, ray, cnt, [&](uint32_t first, uint32_t count) {
        for (uint32_t i = first, e = first + count; i < e; ++i) {
            const uint32_t  p = s.unocc_indices[i];
            const ZygpuProp prop = s.props[p];
            if (!propVisibleAtDepth(prop.flags, st.depth_surface)) continue;
            if (STATS) cnt.instances += 1;
            if (!aabbIntersect(s.aabbs[2 * p], s.aabbs[2 * p + 1], ray)) continue;
            if (ZYGPU_SHAPE_RECTANGLE != prop.shape) continue;
            Fragment frag;
            frag.trafo = loadTrafo(s, p);
            frag.prop = p;
            float t, u, v;
            if (!rectangleIntersect(ray, frag.trafo, t, u, v)) continue;
            frag.t_hit = t;
            frag.u = u;
            frag.v = v;
            frag.primitive = 0;
            rectangleFragment(ray, frag);
            energy = energy + evaluateRadiance(s, st, frag, light_split_threshold);
        }
        return false;
    });
    return energy;
}

ZD void ivalueAdd(PathState& st, f3 value, uint32_t depth, uint32_t direct_cutoff, bool is_emission, bool singular) {  // helper.zig:11-19
    if (is_emission) st.emission = st.emission + value;
    else if (singular || depth < direct_cutoff) st.direct = st.direct + value;
    else st.indirect = st.indirect + value;
}

ZD f3 clampColor(f3 color, float max) {  // sensor.zig:615-624
    const float mc = hmax3(color);
    if (mc > max) return S3(max / mc) * color;
    return color;
}

ZD void finalize(const Params& P, uint32_t slot, const PathState& st) {  // Sensor.addSample's colour part (sensor.zig:177-181)
    const f3    e = clampColor(st.emission, c_view.v.clamp_emission);
    const f3    d = clampColor(st.direct, c_view.v.clamp_direct);
    const f3    i = clampColor(st.indirect, c_view.v.clamp_indirect);
    const f3    summed = e + d + i;
    const float alpha = (st.flags & ST_TRANSPARENT) ? mx(1.f - average3(st.throughput), 0.f) : 1.f;
    P.stage_color[slot] = make_float4(summed.x, summed.y, summed.z, alpha);
}

template <bool STATS, bool INLINE_SH>
ZD void shadeSlot(const Params& P, uint32_t slot, bool from_shadow, Counters& cnt, unsigned long long& shadow_rays, bool& push_ext,
                  bool& push_sh) {
    const DScene&  s = P.scene;
    const uint32_t kind = c_view.v.integrator;
    const bool     mis = ZYGPU_INTEGRATOR_PTMIS == kind;

    int      px, py;
    uint32_t iteration;
    slotPixel(slot, px, py, iteration);
    const uint32_t pixel_id = pixelId(px, py);

    PathState st;
    uint32_t  meta;
    loadSampler(P, slot, pixel_id, iteration, st, meta);
    st.depth_surface = (meta >> 3) & 31u;
    st.vdepth = (meta >> 8) & 31u;
    st.flags = (meta >> 13) & 63u;
    st.pick = (meta >> 20) & 127u;
    st.smp.use_primary = st.depth_surface < 3;  // worker.zig:201-207

    const float4 ro = P.pool.ray_o[slot], rd = P.pool.ray_d[slot], th = P.pool.thr[slot];
    const float4 vo = P.pool.vorigin[slot], vg = P.pool.vgeo_n[slot];
    const float4 em = P.pool.emission[slot], di = P.pool.direct[slot], in = P.pool.indirect[slot];
    st.bxdf_pdf = rd.w;
    st.select = th.w;
    st.throughput = mk3(th.x, th.y, th.z);
    st.vorigin = mk3(vo.x, vo.y, vo.z);
    st.vgeo_n = mk3(vg.x, vg.y, vg.z);
    st.emission = mk3(em.x, em.y, em.z);
    st.direct = mk3(di.x, di.y, di.z);
    st.indirect = mk3(in.x, in.y, in.z);

    const uint4    hr = P.pool.hit[slot];
    const uint32_t hit_prop = P.pool.hit_prop[slot];
    Hit            hit{__uint_as_float(hr.x), __uint_as_float(hr.y), __uint_as_float(hr.z), hr.w, hit_prop};
    const bool     has_hit = kNull != hit_prop;
    // after Scene.intersect the probe's max_t is the hit distance (prop_tree.zig:76-79)
    st.ray = Ray::init(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), 0.f, has_hit ? hit.t : ro.w);

    const uint32_t total_depth = st.depth_surface;
    Fragment       frag;
    if (has_hit) makeFragment(s, st.ray, hit, frag);

    float light_split_threshold = 0.f;

    if (!from_shadow) {
        // ---- connectLight (pathtracer.zig:110-143, pathtracer_dl.zig:173-207, pathtracer_mis.zig:280-341)
        f3 energy = S3(0.f);
        if (!(!c_view.v.caustics_path && (st.flags & ST_SPECULAR) && !(st.flags & ST_PRIMARY))) {
            if (ZYGPU_INTEGRATOR_PT == kind) {
                const uint32_t saved = st.flags;
                st.flags |= ST_SINGULAR;
                if (has_hit) energy = evaluateRadiance(s, st, frag, 0.f);
                energy = energy + unoccludingEmission<STATS>(s, st, 0.f, cnt);
                st.flags = saved;
            } else if (ZYGPU_INTEGRATOR_PTDL == kind) {
                if (has_hit) {
                    const uint32_t light_id = s.light_ids[s.props[frag.prop].parts_start + frag.part];
                    if ((st.flags & ST_SINGULAR) || kNull == light_id) energy = evaluateRadiance(s, st, frag, 0.f);
                }
                if (st.flags & ST_SINGULAR) energy = energy + unoccludingEmission<STATS>(s, st, 0.f, cnt);
            } else {
                light_split_threshold = splitThreshold(st.vdepth);
                if (has_hit) energy = evaluateRadiance(s, st, frag, light_split_threshold);
                energy = energy + unoccludingEmission<STATS>(s, st, light_split_threshold, cnt);
            }
        }
        ivalueAdd(st, st.throughput * energy, total_depth, mis ? 2u : 1u, 0 == total_depth, 0 != (st.flags & ST_SINGULAR));

        if (!has_hit || st.depth_surface >= c_view.v.max_depth_surface || 0 >= c_view.v.max_depth_volume) {
            if (mis && (st.flags & ST_TRANSPARENT)) st.throughput = st.throughput * (S3(1.f) - min3v(energy, S3(1.f)));
            finalize(P, slot, st);
            return;
        }

        // russianRoulette (helper.zig:75-89)
        {
            const float r = st.smp.sample1D();
            const float continuation_probability = hmax3(st.throughput) / 0.1f;
            if (continuation_probability < 1.f) {
                if (r >= continuation_probability) {
                    finalize(P, slot, st);
                    return;
                }
                st.throughput = st.throughput / S3(continuation_probability);
            }
        }
    }

    // ---- Vertex.sample (vertex.zig:137-181): closure at the hit (pure apart from the stochastic_r draw)
    const f3             wo = -st.ray.direction;
    const ZygpuMaterial& m = s.materials[s.material_ids[s.props[frag.prop].parts_start + frag.part]];
    f3                   geo_n = frag.geo_n, n = frag.n;
    if ((m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo)) {
        geo_n = -geo_n;
        n = -n;
    }
    if (!from_shadow) (void)st.smp.sample1D();  // rs.stochastic_r
    const bool           caustics = (st.flags & ST_PRIMARY) ? true : 0 != c_view.v.caustics_path;
    const MaterialSample ms = makeMaterialSample(s, m, wo, geo_n, frag.t, frag.b, n, caustics);

    // ---- next-event estimation (pathtracer_mis.zig:174-278, pathtracer_dl.zig:120-171)
    const bool do_nee = ZYGPU_INTEGRATOR_PT != kind && ms.can_evaluate;
    if (do_nee) {
        const float nee_threshold = splitThreshold(st.depth_surface);  // MIS: vertex.light_split_threshold; DL: the same value
        const LightTree lt{s};
        // one light pick: Light.sampleTo -> shadow ray -> pre-weighted contribution (pathtracer_mis.zig:214-278)
        auto prepare = [&](const LightPick& pick, f3& sorigin, f3& sdir, float& slen, f3& contribution) -> bool {
            const ZygpuLight     light = s.lights[pick.offset];
            const Trafo          trafo = loadTrafo(s, light.prop);
            const ZygpuMaterial& lm = s.materials[s.material_ids[s.props[light.prop].parts_start + light.part]];
            SampleTo             ls;
            if (!rectangleSampleTo(frag.p, ms.geo_n, trafo, 0 != light.two_sided, ms.translucent, st.smp, ls)) return false;
            sorigin = frag.offsetP(ls.wi);
            shadowRay(sorigin, ls, sdir, slen);
            const f3         radiance = materialRadiance(s, lm, ls.wi, trafo, light.prop, light.part, false);
            const BxdfResult bx = ms.evaluate(s, ls.wi);
            if (mis) {
                const float light_pdf = ls.pdf * pick.pdf;
                const float weight = predividedPowerHeuristic(light_pdf, bx.pdf);
                contribution = S3(weight) * radiance * bx.reflection;
            } else {
                const float weight = 1.f / (pick.pdf * ls.pdf);
                contribution = S3(weight) * (S3(1.f) * radiance * bx.reflection);
            }
            return true;
        };

        if (INLINE_SH) {
            // shadow rays traced in place: the sampler draw that follows a successful test (light.zig:127) makes the
            // light picks of one vertex strictly sequential anyway, so nothing is lost by not queueing them
            st.select = st.smp.sample1D();
            st.nee = S3(0.f);
            lt.randomLight(frag.p, ms.geo_n, ms.translucent, st.select, nee_threshold, [&](uint32_t, LightPick pick) {
                f3    sorigin, sdir, contribution;
                float slen;
                if (!prepare(pick, sorigin, sdir, slen, contribution)) return;
                Ray shadow = Ray::init(sorigin, sdir, 0.f, slen);
                if (STATS) shadow_rays += 1;
                if (sceneVisibility<STATS>(s, shadow, cnt)) {
                    (void)st.smp.sample1D();
                    st.nee = st.nee + contribution;
                }
            });
        } else {
            if (!from_shadow) {
                st.select = st.smp.sample1D();
                st.pick = 0;
                st.nee = S3(0.f);
            } else {
                const float4 ns = P.pool.nee[slot];
                st.nee = mk3(ns.x, ns.y, ns.z);
                if (P.pool.visible[slot]) {
                    (void)st.smp.sample1D();  // light.zig:127 rs.stochastic_r, drawn only for unoccluded samples
                    const float4 c = P.pool.contrib[slot];
                    st.nee = st.nee + mk3(c.x, c.y, c.z);
                }
                st.pick += 1;
            }

            LightPick      picks[kMaxLightPicks];
            const uint32_t num_picks = lt.randomLight(frag.p, ms.geo_n, ms.translucent, st.select, nee_threshold,
                                                      [&](uint32_t i, LightPick pk) { picks[i] = pk; });
            while (st.pick < num_picks) {
                f3    sorigin, sdir, contribution;
                float slen;
                if (prepare(picks[st.pick], sorigin, sdir, slen, contribution)) {
                    P.pool.sh_o[slot] = make_float4(sorigin.x, sorigin.y, sorigin.z, slen);
                    P.pool.sh_d[slot] = make_float4(sdir.x, sdir.y, sdir.z, 0.f);
                    P.pool.contrib[slot] = make_float4(contribution.x, contribution.y, contribution.z, 0.f);
                    P.pool.nee[slot] = make_float4(st.nee.x, st.nee.y, st.nee.z, 0.f);
                    P.pool.thr[slot] = make_float4(st.throughput.x, st.throughput.y, st.throughput.z, st.select);
                    P.pool.emission[slot] = make_float4(st.emission.x, st.emission.y, st.emission.z, 0.f);
                    P.pool.direct[slot] = make_float4(st.direct.x, st.direct.y, st.direct.z, 0.f);
                    P.pool.indirect[slot] = make_float4(st.indirect.x, st.indirect.y, st.indirect.z, 0.f);
                    storeSampler(P, slot, st, PH_SHADOW);
                    push_sh = true;
                    return;
                }
                st.pick += 1;
            }
        }
        ivalueAdd(st, st.throughput * st.nee, total_depth, 1u, false, false);
    }

    // ---- bxdf sample + path extension (pathtracer_mis.zig:123-166)
    BxdfSample bs;
    if (!ms.sample(s, st.smp, bs)) {
        st.throughput = S3(0.f);
        finalize(P, slot, st);
        return;
    }

    // Vertex.State.update (vertex.zig:30-43); every sample in scope is a reflection (never .Straight)
    if (kScatterSpecular == bs.scattering) {
        st.flags |= ST_SPECULAR;
        st.flags = (0.f == bs.reg_alpha) ? (st.flags | ST_SINGULAR) : (st.flags & ~uint32_t(ST_SINGULAR));
        if (st.flags & ST_PRIMARY) st.flags |= ST_STARTED_SPECULAR;
    } else {
        st.flags &= ~uint32_t(ST_SPECULAR | ST_SINGULAR | ST_PRIMARY);
    }
    if (mis) {
        st.flags = ms.translucent ? (st.flags | ST_TRANSLUCENT) : (st.flags & ~uint32_t(ST_TRANSLUCENT));
        st.vdepth = st.depth_surface;
        st.bxdf_pdf = bs.pdf;
        st.vgeo_n = ms.geo_n;
    }
    st.vorigin = frag.p;
    st.throughput = st.throughput * (bs.reflection / S3(bs.pdf));
    const f3 new_origin = frag.offsetP(bs.wi);
    st.depth_surface += 1;
    st.flags &= ~uint32_t(ST_TRANSPARENT);  // transparent && (Transmission || Straight) is false for a reflection
    st.smp.incrementPadding();
    st.pick = 0;

    P.pool.ray_o[slot] = make_float4(new_origin.x, new_origin.y, new_origin.z, kRayMaxT);
    P.pool.ray_d[slot] = make_float4(bs.wi.x, bs.wi.y, bs.wi.z, st.bxdf_pdf);
    P.pool.thr[slot] = make_float4(st.throughput.x, st.throughput.y, st.throughput.z, st.select);
    P.pool.vorigin[slot] = make_float4(st.vorigin.x, st.vorigin.y, st.vorigin.z, 0.f);
    P.pool.vgeo_n[slot] = make_float4(st.vgeo_n.x, st.vgeo_n.y, st.vgeo_n.z, 0.f);
    P.pool.emission[slot] = make_float4(st.emission.x, st.emission.y, st.emission.z, 0.f);
    P.pool.direct[slot] = make_float4(st.direct.x, st.direct.y, st.direct.z, 0.f);
    P.pool.indirect[slot] = make_float4(st.indirect.x, st.indirect.y, st.indirect.z, 0.f);
    storeSampler(P, slot, st, PH_HIT);
    push_ext = true;
}

template <bool STATS, bool INLINE_SH>
__global__ void __launch_bounds__(256) wf_shade(Params P) {
    const uint32_t cur = P.parity, nxt = cur ^ 1u;
    const uint32_t n_ext = P.ctl->ext_count[cur], n_sh = P.ctl->sh_count[cur];
    const uint32_t total = n_ext + n_sh;
    Counters           cnt{0, 0, 0};
    unsigned long long shadow_rays = 0;
    const uint32_t     stride = gridDim.x * blockDim.x;
    const uint32_t     rounded = (total + 63u) & ~63u;  // whole waves stay converged for the ballot in queuePush
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride) {
        bool     push_ext = false, push_sh = false;
        uint32_t slot = 0;
        if (i < total) {
            const bool from_shadow = i >= n_ext;
            slot = from_shadow ? P.q.sh[cur][i - n_ext] : P.q.ext[cur][i];
            shadeSlot<STATS, INLINE_SH>(P, slot, from_shadow, cnt, shadow_rays, push_ext, push_sh);
        }
        queuePush(P.q.ext[nxt], &P.ctl->ext_count[nxt], push_ext, slot);
        if (!INLINE_SH) queuePush(P.q.sh[nxt], &P.ctl->sh_count[nxt], push_sh, slot);
    }
    if (STATS) {
        atomicAdd(&P.ctl->stat_nodes, cnt.nodes);
        atomicAdd(&P.ctl->stat_tris, cnt.tris);
        atomicAdd(&P.ctl->stat_instances, cnt.instances);
        atomicAdd(&P.ctl->stat_shadow, shadow_rays);
    }
}

// ---------------------------------------------------------------- K4: film
// Sensor.addSample's splat (sensor.zig:183-326) as a gather per film pixel, samples taken in iteration order and,
// inside one iteration, in raster order; Opaque.addPixel arithmetic (buffer_opaque.zig:39-45).
ZD float filterEval(float sv) {  // sensor.zig:626-628 + interpolated_function.zig:154-166
    const float    range_end = c_view.v.filter_radius;
    const float    inverse_interval = 1.f / (range_end / float(ZYGPU_FILTER_TAPS - 1));
    const float    cx = mn(fabsf(sv), range_end);
    const float    o = cx * inverse_interval;
    const uint32_t offset = uint32_t(o);
    const float    t = o - float(offset);
    return lerpf(c_view.v.filter_samples[offset], c_view.v.filter_samples[min(offset + 1u, uint32_t(ZYGPU_FILTER_TAPS - 1))], t);
}

__global__ void __launch_bounds__(256) film_accumulate(Params P) {
    const int fr = c_view.v.filter_radius_int;
    const int w = c_view.x_end - c_view.x_begin;  // sample columns
    // film rows that can receive samples from this batch
    const int fy0 = max(c_view.y_begin - fr, c_view.v.crop[1]), fy1 = min(c_view.y_end + fr, c_view.v.crop[3]);
    const int fx0 = c_view.v.crop[0], fx1 = c_view.v.crop[2];
    const int fw = fx1 - fx0, fh = fy1 - fy0;
    const int idx = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= fw * fh) return;
    const int x = fx0 + idx % fw, y = fy0 + idx / fw;

    float4* pixel = P.film + (size_t(c_view.v.resolution[0]) * size_t(y) + size_t(x));
    float4  value = *pixel;
    for (uint32_t it = 0; it < c_view.batch_iterations; ++it) {
        const size_t base = size_t(it) * c_view.npix;
        if (0 == fr) {
            if (y >= c_view.y_begin && y < c_view.y_end) {
                const float4 c = P.stage_color[base + size_t(y - c_view.y_begin) * w + size_t(x - c_view.x_begin)];
                const f3     wc = S3(1.f) * mk3(c.x, c.y, c.z);
                value.x += wc.x;
                value.y += wc.y;
                value.z += wc.z;
                value.w += 1.f;
            }
        } else {
            for (int sy = y - fr; sy <= y + fr; ++sy) {
                if (sy < c_view.y_begin || sy >= c_view.y_end) continue;
                for (int sx = x - fr; sx <= x + fr; ++sx) {
                    if (sx < c_view.x_begin || sx >= c_view.x_end) continue;
                    const size_t si = base + size_t(sy - c_view.y_begin) * w + size_t(sx - c_view.x_begin);
                    const float4 c = P.stage_color[si];
                    const float2 uv = P.stage_uv[si];
                    // the sample at (sx, sy) reaches pixel (sx + i - fr, sy + j - fr) with weight eval(ox + fr - i) * eval(oy + fr - j)
                    const float ox = uv.x - 0.5f, oy = uv.y - 0.5f;
                    const float wx = filterEval(ox + float(sx - x)), wy = filterEval(oy + float(sy - y));
                    const float weight = wx * wy;
                    const f3    wc = S3(weight) * mk3(c.x, c.y, c.z);
                    value.x += wc.x;
                    value.y += wc.y;
                    value.z += wc.z;
                    value.w += weight;
                }
            }
        }
    }
    *pixel = value;
}

__global__ void film_clear(float4* film, uint32_t n, float weight) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) film[i] = make_float4(0.f, 0.f, 0.f, weight);
}

// resolveTonemap (buffer_opaque.zig:73-79) with the Linear tonemapper (tonemapper.zig:36-39, aces.zig:19-27)
__global__ void film_resolve(const float4* film, float4* out, uint32_t n, float exposure_factor) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = film[i];
    const f3     q = mk3(p.x, p.y, p.z) / S3(p.w);
    const f3     color = mk3(fabsf(q.x), fabsf(q.y), fabsf(q.z));
    const f3     scaled = S3(exposure_factor) * color;
    const f3     srgb = mk3(1.70505155f, -0.13025714f, -0.02400328f) * S3(scaled.x) + mk3(-0.62179068f, 1.14080289f, -0.12896877f) * S3(scaled.y) +
                    mk3(-0.08325840f, -0.01054853f, 1.15297171f) * S3(scaled.z);
    out[i] = make_float4(srgb.x, srgb.y, srgb.z, 1.f);
}


// ---------------------------------------------------------------- device unit-test hook
// Evaluates small device functions on caller-provided inputs so that tests can compare them with the oracle:
//  op 0: x -> sinF, cosF, acosF                       (1 in, 3 out)
//  op 1: scale.xy, o.xyz, u0, u1 -> SphQuad S, pdf, sample.xyz (7 in, 5 out)
//  op 2: sample, seed -> the first 10 Sobol 1D draws with a padding increment after 5 (2 in as bits, 10 out)
//  op 3: state.lo, state.hi(seq) -> 4 PCG floats      (2 in as bits, 4 out)
//  op 4: p.xyz, n.xyz -> offsetRay                    (6 in, 3 out)
__global__ void test_eval(uint32_t op, uint32_t n, const float* in, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (0 == op) {
        const float x = in[i];
        out[3 * i] = sinF(x);
        out[3 * i + 1] = cosF(x);
        out[3 * i + 2] = acosF(x);
    } else if (1 == op) {
        const float*  a = in + 7 * i;
        const f3      scale = mk3(a[0], a[1], 1.f);
        const SphQuad q = SphQuad::init(scale, mk3(a[2], a[3], a[4]));
        const f3      sp = q.sample(a[5], a[6]);
        float*        o = out + 5 * i;
        o[0] = q.Sa;
        o[1] = q.pdf(scale);
        o[2] = sp.x;
        o[3] = sp.y;
        o[4] = sp.z;
    } else if (2 == op) {
        Sampler smp;
        smp.sobol_kind = true;
        smp.use_primary = true;
        smp.startPixel(__float_as_uint(in[2 * i]), __float_as_uint(in[2 * i + 1]));
        for (int k = 0; k < 10; ++k) {
            out[10 * i + k] = smp.sample1D();
            if (4 == k) smp.incrementPadding();
        }
    } else if (3 == op) {
        Pcg g;
        g.start(uint64_t(__float_as_uint(in[2 * i])), uint64_t(__float_as_uint(in[2 * i + 1])));
        for (int k = 0; k < 4; ++k) out[4 * i + k] = g.randomFloat();
    } else if (5 == op) {
        const float a = in[2 * i], b = in[2 * i + 1];
        out[6 * i] = a / b;
        out[6 * i + 1] = sqrtf(fabsf(a));
        out[6 * i + 2] = a * b + a;
        out[6 * i + 3] = 1.f / a;
        out[6 * i + 4] = acosR(fabsf(a) * 0.25f);
        const float z = fabsf(a) * 0.25f;
        out[6 * i + 5] = z * (1.6666586697e-01f + z * (-4.2743422091e-02f + z * -8.6563630030e-03f));
    } else if (4 == op) {
        const float* a = in + 6 * i;
        const f3     r = offsetRay(mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]));
        out[3 * i] = r.x;
        out[3 * i + 1] = r.y;
        out[3 * i + 2] = r.z;
    }
}

// ---------------------------------------------------------------- host side

thread_local std::string g_error;

bool check(hipError_t e, const char* what) {
    if (hipSuccess == e) return true;
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

#define HIP_OK(call)                   \
    do {                               \
        if (!check((call), #call)) return -1; \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> allocs;
    ~DeviceBuffers() { release(); }
    void release() {
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
    }
    template <typename T>
    bool upload(const T* src, size_t count, const T** out) {
        *out = nullptr;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        void*        d = nullptr;
        if (!check(hipMalloc(&d, bytes), "hipMalloc")) return false;
        allocs.push_back(d);
        if (count && src && !check(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy")) return false;
        *out = static_cast<const T*>(d);
        return true;
    }
    template <typename T>
    bool alloc(size_t count, T** out) {
        void* d = nullptr;
        if (!check(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)), "hipMalloc")) return false;
        allocs.push_back(d);
        *out = static_cast<T*>(d);
        return true;
    }
};

}  // namespace

struct ZygpuContext {
    int           device = 0;
    hipStream_t   stream = nullptr;
    DeviceBuffers scene_mem, pool_mem, film_mem;
    DScene        scene{};
    bool          has_scene = false, has_view = false;
    ZygpuViewDesc view{};
    Pool          pool{};
    Queues        queues{};
    Control*      ctl = nullptr;
    float4*       stage_color = nullptr;
    float2*       stage_uv = nullptr;
    float4*       film = nullptr;         // internal film
    float4*       film_active = nullptr;  // internal or external
    float4*       resolve_tmp = nullptr;
    size_t        pool_capacity = 0;      // paths
    size_t        film_pixels = 0;
    bool          stats = false;
    bool          inline_shadows = true;  // shadow rays traced inside wf_shade (default) or through the shadow queue
    uint32_t      max_light_picks = 1;
    int           num_cus = 256;
    // kernel timing
    bool                                          timing = true;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events, shade_events;
    size_t                                        trace_used = 0, shade_used = 0;
    double                                        trace_ms = 0.0, shade_ms = 0.0;
    uint64_t                                      trace_launches = 0, shade_launches = 0;
};

namespace {

int ensurePool(ZygpuContext* c, size_t paths) {
    if (paths <= c->pool_capacity) return 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    c->pool_mem.release();
    DeviceBuffers& m = c->pool_mem;
    Pool&          p = c->pool;
    bool           ok = m.alloc(paths, &p.q0) && m.alloc(paths, &p.q1) && m.alloc(paths, &p.pcg) && m.alloc(paths, &p.ray_o) &&
              m.alloc(paths, &p.ray_d) && m.alloc(paths, &p.thr) && m.alloc(paths, &p.vorigin) && m.alloc(paths, &p.vgeo_n) &&
              m.alloc(paths, &p.emission) && m.alloc(paths, &p.direct) && m.alloc(paths, &p.indirect) && m.alloc(paths, &p.hit) &&
              m.alloc(paths, &p.hit_prop) && m.alloc(paths, &p.sh_o) && m.alloc(paths, &p.sh_d) && m.alloc(paths, &p.visible) &&
              m.alloc(paths, &p.contrib) && m.alloc(paths, &p.nee) && m.alloc(paths, &c->queues.ext[0]) &&
              m.alloc(paths, &c->queues.ext[1]) && m.alloc(paths, &c->queues.sh[0]) && m.alloc(paths, &c->queues.sh[1]) &&
              m.alloc(paths, &c->stage_color) && m.alloc(paths, &c->stage_uv);
    if (!ok) return -1;
    c->pool_capacity = paths;
    return 0;
}

void collectTimings(ZygpuContext* c) {
    auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& ev, size_t& used, double& ms, uint64_t& launches) {
        for (size_t i = 0; i < used; ++i) {
            float t = 0.f;
            if (hipSuccess == hipEventElapsedTime(&t, ev[i].first, ev[i].second)) {
                ms += double(t);
                launches += 1;
            }
        }
        used = 0;
    };
    drain(c->trace_events, c->trace_used, c->trace_ms, c->trace_launches);
    drain(c->shade_events, c->shade_used, c->shade_ms, c->shade_launches);
}

std::pair<hipEvent_t, hipEvent_t>* nextEvents(std::vector<std::pair<hipEvent_t, hipEvent_t>>& ev, size_t& used) {
    if (used == ev.size()) {
        hipEvent_t a, b;
        if (hipSuccess != hipEventCreate(&a) || hipSuccess != hipEventCreate(&b)) return nullptr;
        ev.emplace_back(a, b);
    }
    return &ev[used++];
}

}  // namespace

extern "C" {

const char* zygpu_last_error(void) { return g_error.c_str(); }

int zygpu_create(int device_id, ZygpuContext** out_ctx) {
    if (!out_ctx) return -1;
    *out_ctx = nullptr;
    int count = 0;
    if (!check(hipGetDeviceCount(&count), "hipGetDeviceCount") || device_id < 0 || device_id >= count) {
        if (g_error.empty()) g_error = "no such HIP device";
        return -1;
    }
    HIP_OK(hipSetDevice(device_id));
    auto* c = new ZygpuContext();
    c->device = device_id;
    if (!check(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) {
        delete c;
        return -1;
    }
    hipDeviceProp_t prop;
    if (hipSuccess == hipGetDeviceProperties(&prop, device_id)) c->num_cus = prop.multiProcessorCount;
    void* ctl = nullptr;
    if (!check(hipMalloc(&ctl, sizeof(Control)), "hipMalloc") || !check(hipMemset(ctl, 0, sizeof(Control)), "hipMemset")) {
        delete c;
        return -1;
    }
    c->ctl = static_cast<Control*>(ctl);
    *out_ctx = c;
    return 0;
}

int zygpu_destroy(ZygpuContext* c) {
    if (!c) return -1;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& e : c->trace_events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (auto& e : c->shade_events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (c->ctl) (void)hipFree(c->ctl);
    if (c->resolve_tmp) (void)hipFree(c->resolve_tmp);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int zygpu_upload_scene(ZygpuContext* c, const ZygpuSceneDesc* d) {
    if (!c || !d) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    c->scene_mem.release();
    c->has_scene = false;
    DeviceBuffers& m = c->scene_mem;
    DScene&        s = c->scene;
    std::memset(&s, 0, sizeof(s));

    for (uint32_t l = 0; l < d->num_lights; ++l) {
        if (d->lights[l].num_samples > 1) {
            g_error = "lights with emittance num_samples > 1 are not supported on the device path";
            return -2;
        }
        if (ZYGPU_SHAPE_RECTANGLE != d->props[d->lights[l].prop].shape) {
            g_error = "only Rectangle lights are supported on the device path";
            return -2;
        }
    }
    for (uint32_t p = 0; p < d->num_props; ++p) {
        const uint32_t sh = d->props[p].shape;
        const bool     in_tree = 0 != (d->props[p].flags & ZYGPU_PROP_SOLID);
        (void)in_tree;
        if (ZYGPU_SHAPE_RECTANGLE != sh && ZYGPU_SHAPE_TRIANGLE_MESH != sh && ZYGPU_SHAPE_DISTANT != sh) {
            g_error = "unsupported shape class on the device path";
            return -2;
        }
    }
    c->max_light_picks = std::max<uint32_t>(1, std::min<uint32_t>(64, d->num_lights));

    bool ok = m.upload(d->props, d->num_props, &s.props);
    ok = ok && m.upload(reinterpret_cast<const float4*>(d->trafos), size_t(d->num_props) * 4, &s.trafos);
    ok = ok && m.upload(reinterpret_cast<const float4*>(d->aabbs), size_t(d->num_props) * 2, &s.aabbs);
    ok = ok && m.upload(d->material_ids, d->num_parts, &s.material_ids);
    ok = ok && m.upload(d->light_ids, d->num_parts, &s.light_ids);
    ok = ok && m.upload(d->materials, d->num_materials, &s.materials);
    ok = ok && m.upload(d->lights, d->num_lights, &s.lights);
    ok = ok && m.upload(reinterpret_cast<const float4*>(d->light_aabbs), size_t(d->num_lights) * 2, &s.light_aabbs);
    ok = ok && m.upload(reinterpret_cast<const float4*>(d->light_cones), d->num_lights, &s.light_cones);
    ok = ok && m.upload(reinterpret_cast<const float4*>(d->solid.nodes), size_t(d->solid.num_nodes) * 2, &s.solid_nodes);
    ok = ok && m.upload(d->solid.indices, d->solid.num_indices, &s.solid_indices);
    ok = ok && m.upload(reinterpret_cast<const float4*>(d->unoccluding.nodes), size_t(d->unoccluding.num_nodes) * 2, &s.unocc_nodes);
    ok = ok && m.upload(d->unoccluding.indices, d->unoccluding.num_indices, &s.unocc_indices);
    ok = ok && m.upload(d->light_tree.nodes, d->light_tree.num_nodes, &s.lt_nodes);
    ok = ok && m.upload(d->light_tree.node_middles, d->light_tree.num_nodes, &s.lt_middles);
    ok = ok && m.upload(d->light_tree.light_orders, d->light_tree.num_lights, &s.lt_orders);
    ok = ok && m.upload(d->light_tree.light_mapping, d->light_tree.num_lights, &s.lt_mapping);
    ok = ok && m.upload(d->luts.e_m, d->luts.e_m ? 32 * 32 : 0, &s.lut_e_m);
    ok = ok && m.upload(d->luts.e_m_avg, d->luts.e_m_avg ? 32 : 0, &s.lut_e_m_avg);
    ok = ok && m.upload(d->luts.e, d->luts.e ? 16 * 16 * 16 : 0, &s.lut_e);
    ok = ok && m.upload(d->luts.e_avg, d->luts.e_avg ? 16 * 16 : 0, &s.lut_e_avg);
    if (!ok) return -1;

    for (uint32_t mi = 0; mi < d->num_materials; ++mi) {
        if (ZYGPU_MAT_SUBSTITUTE == d->materials[mi].kind && !d->luts.e_m) {
            g_error = "Substitute materials need the albedo LUTs in ZygpuSceneDesc.luts";
            return -2;
        }
    }

    std::vector<DMesh> meshes(d->num_meshes);
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        const ZygpuMesh& hm = d->meshes[i];
        DMesh&           dm = meshes[i];
        // pre-gathered triangle vertices, leaf order: (a, b, c) as float4 each
        std::vector<float4> tv(size_t(hm.num_triangles) * 3);
        for (uint32_t t = 0; t < hm.num_triangles; ++t) {
            for (int k = 0; k < 3; ++k) {
                const uint32_t vi = hm.triangles[size_t(t) * 3 + k];
                tv[size_t(t) * 3 + k] = make_float4(hm.positions[size_t(vi) * 3], hm.positions[size_t(vi) * 3 + 1], hm.positions[size_t(vi) * 3 + 2], 0.f);
            }
        }
        ok = m.upload(reinterpret_cast<const float4*>(hm.nodes), size_t(hm.num_nodes) * 2, &dm.nodes);
        ok = ok && m.upload(tv.data(), tv.size(), &dm.tri_verts);
        ok = ok && m.upload(hm.triangles, size_t(hm.num_triangles) * 3, &dm.triangles);
        ok = ok && m.upload(hm.triangle_parts, hm.num_triangles, &dm.triangle_parts);
        ok = ok && m.upload(hm.positions, size_t(hm.num_vertices) * 3 + 1, &dm.positions);
        ok = ok && m.upload(hm.normals, size_t(hm.num_vertices) * 2, &dm.normals);
        ok = ok && m.upload(hm.uvs, size_t(hm.num_vertices) * 2, &dm.uvs);
        ok = ok && m.upload(hm.part_areas, hm.num_parts, &dm.part_areas);
        if (!ok) return -1;
    }
    if (!m.upload(meshes.data(), meshes.size(), &s.meshes)) return -1;

    const ZygpuLightTree& lt = d->light_tree;
    s.lt_bounds_min = make_float4(lt.bounds.min[0], lt.bounds.min[1], lt.bounds.min[2], lt.bounds.min[3]);
    s.lt_bounds_max = make_float4(lt.bounds.max[0], lt.bounds.max[1], lt.bounds.max[2], lt.bounds.max[3]);
    s.lt_infinite_weight = lt.infinite_weight;
    s.lt_max_split_depth = lt.max_split_depth;
    s.lt_num_nodes = lt.num_nodes;
    s.lt_num_infinite = lt.num_infinite_lights;
    s.specular_threshold = d->specular_threshold;
    s.num_solid_nodes = d->solid.num_nodes;
    s.num_unocc_nodes = d->unoccluding.num_nodes;
    c->has_scene = true;
    return 0;
}

int zygpu_set_view(ZygpuContext* c, const ZygpuViewDesc* v) {
    if (!c || !v) return -1;
    HIP_OK(hipSetDevice(c->device));
    if (v->resolution[0] <= 0 || v->resolution[1] <= 0 || v->filter_radius_int < 0 || v->filter_radius_int > 2 ||
        v->max_depth_surface > 30 || 0 == v->num_expected_samples) {
        g_error = "view out of range (resolution, filter radius 0..2, surface depth <= 30, samples per pixel >= 1)";
        return -2;
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    c->view = *v;
    const size_t pixels = size_t(v->resolution[0]) * size_t(v->resolution[1]);
    if (pixels != c->film_pixels) {
        c->film_mem.release();
        if (!c->film_mem.alloc(pixels, &c->film)) return -1;
        c->film_pixels = pixels;
        HIP_OK(hipMemset(c->film, 0, pixels * sizeof(float4)));
        if (c->resolve_tmp) (void)hipFree(c->resolve_tmp);
        c->resolve_tmp = nullptr;
    }
    c->film_active = c->film;
    c->has_view = true;
    return 0;
}

int zygpu_set_external_film(ZygpuContext* c, void* device_rgbw) {
    if (!c || !c->has_view) return -1;
    HIP_OK(hipStreamSynchronize(c->stream));
    c->film_active = device_rgbw ? static_cast<float4*>(device_rgbw) : c->film;
    return 0;
}

int zygpu_clear_film(ZygpuContext* c, float weight) {
    if (!c || !c->has_view) return -1;
    HIP_OK(hipSetDevice(c->device));
    const uint32_t n = uint32_t(c->film_pixels);
    hipLaunchKernelGGL(film_clear, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->film_active, n, weight);
    HIP_OK(hipGetLastError());
    return 0;
}

int zygpu_render(ZygpuContext* c, uint32_t frame, uint32_t iteration, uint32_t num_iterations, int32_t y_begin, int32_t y_end) {
    if (!c || !c->has_scene || !c->has_view) {
        g_error = "zygpu_render before zygpu_upload_scene / zygpu_set_view";
        return -1;
    }
    HIP_OK(hipSetDevice(c->device));
    const ZygpuViewDesc& v = c->view;
    const int            fr = v.filter_radius_int;
    y_begin = std::max(y_begin, v.crop[1]);
    y_end = std::min(y_end, v.crop[3]);
    if (y_end <= y_begin || 0 == num_iterations) return 0;
    // rows at the crop edge own the filter padding, as tiles there do (tile_queue.zig:66-82)
    int yb = y_begin, ye = y_end;
    if (fr > 0) {
        if (yb == v.crop[1]) yb -= fr;
        if (ye == v.crop[3]) ye += fr;
    }
    const int xb = v.crop[0] - fr, xe = v.crop[2] + fr;
    const size_t npix = size_t(xe - xb) * size_t(ye - yb);

    // batch size: as many iterations as fit ~8M paths (bounded by what is left)
    const size_t   target_paths = size_t(8) << 20;
    const uint32_t per_batch = uint32_t(std::max<size_t>(1, std::min<size_t>(num_iterations, target_paths / std::max<size_t>(npix, 1))));
    if (0 != ensurePool(c, npix * per_batch)) return -1;

    Params P;
    P.scene = c->scene;
    P.pool = c->pool;
    P.q = c->queues;
    P.ctl = c->ctl;
    P.stage_color = c->stage_color;
    P.stage_uv = c->stage_uv;
    P.film = c->film_active;
    P.parity = 0;

    const uint32_t picks = (ZYGPU_INTEGRATOR_PT == v.integrator || c->inline_shadows) ? 0u : c->max_light_picks;
    const uint32_t ticks = (v.max_depth_surface + 1) * (1 + picks);
    const dim3     block(256);
    const dim3     grid_persistent(uint32_t(c->num_cus) * 8u);

    for (uint32_t done = 0; done < num_iterations; done += per_batch) {
        const uint32_t b = std::min(per_batch, num_iterations - done);
        View           view;
        view.v = v;
        view.y_begin = yb;
        view.y_end = ye;
        view.x_begin = xb;
        view.x_end = xe;
        view.frame = frame;
        view.iteration0 = iteration + done;
        view.batch_iterations = b;
        view.npix = uint32_t(npix);
        HIP_OK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_view), &view, sizeof(View), 0, hipMemcpyHostToDevice, c->stream));

        const uint32_t num_paths = uint32_t(npix * b);
        P.parity = 0;
        hipLaunchKernelGGL(wf_generate, dim3((num_paths + 255) / 256), block, 0, c->stream, P, num_paths);
        if (c->stats) {
            // camera samples are counted on the host side of the launch
        }
        for (uint32_t t = 0; t < ticks; ++t) {
            auto* te = c->timing ? nextEvents(c->trace_events, c->trace_used) : nullptr;
            if (te) (void)hipEventRecord(te->first, c->stream);
            if (c->stats) hipLaunchKernelGGL(wf_trace<true>, grid_persistent, block, 0, c->stream, P);
            else hipLaunchKernelGGL(wf_trace<false>, grid_persistent, block, 0, c->stream, P);
            if (te) (void)hipEventRecord(te->second, c->stream);
            auto* se = c->timing ? nextEvents(c->shade_events, c->shade_used) : nullptr;
            if (se) (void)hipEventRecord(se->first, c->stream);
            if (c->inline_shadows) {
                if (c->stats) hipLaunchKernelGGL((wf_shade<true, true>), grid_persistent, block, 0, c->stream, P);
                else hipLaunchKernelGGL((wf_shade<false, true>), grid_persistent, block, 0, c->stream, P);
            } else {
                if (c->stats) hipLaunchKernelGGL((wf_shade<true, false>), grid_persistent, block, 0, c->stream, P);
                else hipLaunchKernelGGL((wf_shade<false, false>), grid_persistent, block, 0, c->stream, P);
            }
            if (se) (void)hipEventRecord(se->second, c->stream);
            P.parity ^= 1u;
        }
        const int fh = std::min(ye + fr, v.crop[3]) - std::max(yb - fr, v.crop[1]);
        const int fw = v.crop[2] - v.crop[0];
        const uint32_t fn = uint32_t(std::max(0, fh) * std::max(0, fw));
        if (fn) hipLaunchKernelGGL(film_accumulate, dim3((fn + 255) / 256), block, 0, c->stream, P);
        HIP_OK(hipGetLastError());
        if (c->stats) {
            // stat_samples is host-known; add it on the device copy for a single read-back path
            unsigned long long add = num_paths;
            Control            h;
            HIP_OK(hipStreamSynchronize(c->stream));
            HIP_OK(hipMemcpy(&h, c->ctl, sizeof(Control), hipMemcpyDeviceToHost));
            h.stat_samples += add;
            HIP_OK(hipMemcpy(&c->ctl->stat_samples, &h.stat_samples, sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
        if (c->timing && (c->trace_used > 4096)) {
            HIP_OK(hipStreamSynchronize(c->stream));
            collectTimings(c);
        }
    }
    return 0;
}

int zygpu_synchronize(ZygpuContext* c) {
    if (!c) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    Control h;
    HIP_OK(hipMemcpy(&h, c->ctl, sizeof(Control), hipMemcpyDeviceToHost));
    if (h.error) {
        g_error = "device error flags: " + std::to_string(h.error);
        return -3;
    }
    return 0;
}

int zygpu_download_film(ZygpuContext* c, float* host_rgbw) {
    if (!c || !c->has_view || !host_rgbw) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(host_rgbw, c->film_active, c->film_pixels * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}

int zygpu_upload_film(ZygpuContext* c, const float* host_rgbw) {
    if (!c || !c->has_view || !host_rgbw) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(c->film_active, host_rgbw, c->film_pixels * sizeof(float4), hipMemcpyHostToDevice));
    return 0;
}

int zygpu_film_device_ptr(ZygpuContext* c, void** out_ptr, uint64_t* out_num_floats) {
    if (!c || !c->has_view || !out_ptr) return -1;
    *out_ptr = c->film_active;
    if (out_num_floats) *out_num_floats = uint64_t(c->film_pixels) * 4;
    return 0;
}

int zygpu_resolve(ZygpuContext* c, float* host_rgba, uint32_t num_pixels) {
    if (!c || !c->has_view || !host_rgba) return -1;
    HIP_OK(hipSetDevice(c->device));
    const uint32_t n = std::min<uint32_t>(num_pixels, uint32_t(c->film_pixels));
    if (!c->resolve_tmp) {
        void* d = nullptr;
        HIP_OK(hipMalloc(&d, c->film_pixels * sizeof(float4)));
        c->resolve_tmp = static_cast<float4*>(d);
    }
    hipLaunchKernelGGL(film_resolve, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->film_active, c->resolve_tmp, n, c->view.exposure_factor);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(host_rgba, c->resolve_tmp, size_t(n) * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}

int zygpu_test_eval(ZygpuContext* c, uint32_t op, uint32_t n, const float* in, uint32_t in_stride, float* out, uint32_t out_stride) {
    if (!c || !in || !out) return -1;
    HIP_OK(hipSetDevice(c->device));
    float *din = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&din), size_t(n) * in_stride * sizeof(float)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dout), size_t(n) * out_stride * sizeof(float)));
    HIP_OK(hipMemcpy(din, in, size_t(n) * in_stride * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(test_eval, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, n, din, dout);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out, dout, size_t(n) * out_stride * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(din);
    (void)hipFree(dout);
    return 0;
}

int zygpu_set_option(ZygpuContext* c, const char* name, int value) {
    if (!c || !name) return -1;
    if (0 == std::strcmp(name, "inline_shadows")) {
        c->inline_shadows = 0 != value;
        return 0;
    }
    g_error = std::string("unknown option ") + name;
    return -1;
}

int zygpu_enable_stats(ZygpuContext* c, int enabled) {
    if (!c) return -1;
    c->stats = 0 != enabled;
    return 0;
}

int zygpu_get_stats(ZygpuContext* c, ZygpuStats* out) {
    if (!c || !out) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    Control h;
    HIP_OK(hipMemcpy(&h, c->ctl, sizeof(Control), hipMemcpyDeviceToHost));
    out->camera_samples = h.stat_samples;
    out->extension_rays = h.stat_ext;
    out->shadow_rays = h.stat_shadow;
    out->nodes_fetched = h.stat_nodes;
    out->triangles_tested = h.stat_tris;
    out->instances_tested = h.stat_instances;
    return 0;
}

int zygpu_reset_stats(ZygpuContext* c) {
    if (!c) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemset(c->ctl, 0, sizeof(Control)));
    return 0;
}

int zygpu_kernel_time(ZygpuContext* c, double* out_ms, uint64_t* out_launches, int reset) {
    if (!c) return -1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    collectTimings(c);
    // [0] = wf_trace, [1] = wf_shade
    if (out_ms) {
        out_ms[0] = c->trace_ms;
        out_ms[1] = c->shade_ms;
    }
    if (out_launches) {
        out_launches[0] = c->trace_launches;
        out_launches[1] = c->shade_launches;
    }
    if (reset) {
        c->trace_ms = c->shade_ms = 0.0;
        c->trace_launches = c->shade_launches = 0;
    }
    return 0;
}

}  // extern "C"
