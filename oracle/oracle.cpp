// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in zyg_amd/ (the product) may include, link or call this.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
//
// PARITY PINNING: the reference (Opioid/zyg, Zig) ships no tests, golden vectors or images for this path and
// cannot be compiled here (no Zig toolchain), so this restatement is pinned only by (a) the public PCG32 demo
// vector, (b) sobol5 / hash known answers derived from the reference's direction table and hash constants
// (tests/golden), (c) analytic radiometry tests.  Everything else is "parity unpinned" by the reference itself.
//
// The per-pixel integration loop, restated from:
//   Worker.render                 src/core/rendering/worker.zig:104-168 (pickSampler 201-207)
//   TileQueue / Gilbert curve     src/core/rendering/tile_queue.zig:47-87, gilbert.zig:4-82
//   Sensor.cameraSample/addSample src/core/rendering/sensor/sensor.zig:152-195, 559-628; buffer_opaque.zig:39-79
//   Perspective.generateVertex    src/core/camera/camera_perspective.zig:124-150, camera_base.zig:121-136
//   Vertex / State                src/core/scene/vertex.zig:18-213
//   Context.nextEvent / emission  src/core/scene/context.zig:46-73, prop_tree.zig:302-356, prop.zig:239-264
//   Pathtracer / DL / MIS         src/core/rendering/integrator/surface/pathtracer.zig:27-152,
//                                 pathtracer_dl.zig:31-207, pathtracer_mis.zig:37-341, integrator/helper.zig:6-89
//   Scene.lightPdf                src/core/scene/scene.zig:612-634
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "light.hpp"
#include "material.hpp"

namespace orc {

// ---------------- Sobol directions by the standard Joe-Kuo recurrence ----------------
// dims: van der Corput; then (s, a, m) = (1,0,{1}), (2,1,{1,3}), (3,1,{1,3,1}), (3,2,{1,1,1})
const uint32_t* sobol::directions() {
    static uint32_t D[5 * 32];
    static bool     init = false;
    if (!init) {
        for (uint32_t b = 0; b < 32; ++b) D[b] = 1u << (31 - b);
        const uint32_t S_[4] = {1, 2, 3, 3}, A_[4] = {0, 1, 1, 2};
        const uint32_t M_[4][3] = {{1, 0, 0}, {1, 3, 0}, {1, 3, 1}, {1, 1, 1}};
        for (int d = 0; d < 4; ++d) {
            uint32_t*      v = D + (d + 1) * 32;
            const uint32_t s = S_[d], a = A_[d];
            for (uint32_t i = 0; i < s; ++i) v[i] = M_[d][i] << (31 - i);
            for (uint32_t i = s; i < 32; ++i) {
                v[i] = v[i - s] ^ (v[i - s] >> s);
                for (uint32_t k = 1; k < s; ++k) v[i] ^= (((a >> (s - 1 - k)) & 1u) * v[i - k]);
            }
        }
        init = true;
    }
    return D;
}

// ---------------- helper.zig ----------------
struct IValue {
    V4   emission = S(0.f), direct = S(0.f), indirect = S(0.f);
    void add(V4 value, uint32_t depth, uint32_t direct_cutoff, bool is_emission, bool singular) {
        if (is_emission) emission += value;
        else if (singular || depth < direct_cutoff) direct += value;
        else indirect += value;
    }
};

inline float powerHeuristic(float f_pdf, float g_pdf) {
    const float f2 = f_pdf * f_pdf;
    return f2 / std::fmaf(g_pdf, g_pdf, f2);
}
inline float predividedPowerHeuristic(float f_pdf, float g_pdf) {
    const float f2 = f_pdf * f_pdf;
    return f_pdf / std::fmaf(g_pdf, g_pdf, f2);
}
inline bool russianRoulette(V4& throughput, float r) {
    const float max = hmax3(throughput);
    const float continuation_probability = max / 0.1f;
    if (continuation_probability < 1.f) {
        if (r >= continuation_probability) return true;
        throughput /= S(continuation_probability);
    }
    return false;
}
inline float splitThreshold(float split_threshold, uint32_t total_depth) {  // helper.zig:33-39
    return mn(total_depth < 4 ? split_threshold : kLowThreshold, split_threshold);
}

// ---------------- vertex.zig ----------------
struct VState {
    bool primary_ray = true, transparent = true, singular = true, specular = false, translucent = false,
         started_specular = false;
    void update(const PathInfo& path) {
        if (Scattering::Specular == path.scattering) {
            specular = true;
            singular = path.singular();
            if (primary_ray) started_specular = true;
        } else if (PathEvent::Straight != path.event) {
            specular = false;
            singular = false;
            primary_ray = false;
        }
    }
};

// prop/medium.zig:30-153: the dielectrics the path is inside of.  ior, priority and the absorption coefficient of an
// entry are those of its material (vertex.zig:96-108 pushes material.ior(), priority and the sample's cc, which for
// Glass is the material's coefficient), so an entry is the (prop, part) pair
struct MediumStack {
    static constexpr uint32_t kNumEntries = 4;
    uint32_t index = 0;
    uint32_t prop[kNumEntries], part[kNumEntries];

    bool empty() const { return 0 == index; }
    bool matches(uint32_t i, const Fragment& frag) const { return prop[i] == frag.prop && part[i] == frag.part; }
    void push(const Fragment& frag) {
        if (index < kNumEntries - 1) {
            prop[index] = frag.prop;
            part[index] = frag.part;
            index += 1;
        }
    }
    void remove(const Fragment& frag) {
        const int32_t back = int32_t(index) - 1;
        for (int32_t i = back; i >= 0; --i) {
            if (matches(uint32_t(i), frag)) {
                for (int32_t j = i; j < back; ++j) {
                    prop[j] = prop[j + 1];
                    part[j] = part[j + 1];
                }
                index -= 1;
                return;
            }
        }
    }
};

struct Vertex {
    Ray      ray;
    uint32_t depth_surface = 0;  // probe.depth (volume depth is always 0 in scope)
    uint64_t time = 0;
    VState   state;
    uint32_t depth = 0;  // Vertex.depth: probe.depth at the last scattering vertex
    float    bxdf_pdf = 0.f, light_split_threshold = 0.f;
    float    split_weight = 1.f;
    uint32_t path_count = 1;
    V4       throughput = S(1.f), origin, geo_n = S(0.f);
    MediumStack mediums;
};

struct RenderCtx {
    const ZygpuSceneDesc* scene;
    const ZygpuViewDesc*  view;
    SceneView             sv;
    LightTreeView         lt;
};

struct Worker {
    const RenderCtx* rc;
    Pcg              rng;
    Sampler          samplers[2];
    Counters         cnt;

    Sampler& pickSampler(uint32_t bounce) { return bounce < 3 ? samplers[0] : samplers[1]; }

    float shapeArea(uint32_t prop, uint32_t part, const Trafo& trafo) const {  // shape.zig:132-152
        const ZygpuProp& p = rc->scene->props[prop];
        const V4         s = trafo.scale();
        if (ZYGPU_SHAPE_RECTANGLE == p.shape) return s[0] * s[1];
        if (ZYGPU_SHAPE_DISTANT == p.shape) return distantSolidAngle(s[0]);  // "the solid angle, not the area" (shape.zig:140-142)
        if (ZYGPU_SHAPE_SPHERE == p.shape) return (4.f * kPi) * ((0.5f * s[0]) * (0.5f * s[0]));  // shape.zig:148
        if (ZYGPU_SHAPE_DISK == p.shape) return kPi * ((0.5f * s[0]) * (0.5f * s[0]));
        if (ZYGPU_SHAPE_CUBE == p.shape) return 2.f * (s[0] * s[1] + s[0] * s[2] + s[1] * s[2]);
        if (ZYGPU_SHAPE_DOME == p.shape) return 4.f * kPi;
        if (ZYGPU_SHAPE_CANOPY == p.shape) return 2.f * kPi;
        return rc->scene->meshes[p.mesh].part_areas[part] * (s[0] * s[1]);
    }

    // Material.evaluateRadiance for Light / Substitute (material.zig:196-212; light_material.zig:131-140;
    // substitute_material.zig:275-296 without coating)
    // (u, v) = rs.uv(), r = rs.stochastic_r: the emission map's stochastic-bilinear fetch (emittance.zig:49)
    V4 materialRadiance(const ZygpuMaterial& m, V4 wi, const Trafo& trafo, uint32_t prop, uint32_t part, bool in_camera, float u = 0.f,
                        float v = 0.f, float r = 0.f) const {
        if (ZYGPU_MAT_LIGHT != m.kind && ZYGPU_MAT_SUBSTITUTE != m.kind) return S(0.f);
        const float area = m.emission_normalize ? shapeArea(prop, part, trafo) : 1.f;
        return emittanceRadiance(*rc->scene, m, wi, trafo.r[2], area, in_camera, u, v, r);
    }

    // scene.zig:624-634
    float lightPdf(const Vertex& v, const Fragment& frag) const {
        const uint32_t light_id = rc->sv.propLightId(frag.prop, frag.part);
        if (v.state.singular || kNull == light_id) return 1.f;
        const float       select_pdf = rc->lt.pdf(v.origin, v.geo_n, v.state.translucent, v.light_split_threshold, light_id);
        const ZygpuLight& l = rc->scene->lights[light_id];
        // light.zig:149-157 -> shape.zig:469-492 -> rectangle.zig:554-575 / triangle_mesh.zig:647-691
        const float sample_pdf =
            ZYGPU_LIGHT_PROP_IMAGE == l.light_class  // light.zig:371-374 -> shape.zig:509-526
                ? (ZYGPU_SHAPE_DOME == rc->scene->props[l.prop].shape
                       ? domeMaterialPdf(rc->scene->image_lights[l.sampler], frag.uvw[0], frag.uvw[1], v.light_split_threshold, l.num_samples)
                       : imageLightPdf(rc->scene->image_lights[l.sampler], frag.uvw[0], frag.uvw[1]) / (2.f * kPi))
            : kNull != l.sampler
                ? meshPdf(*rc->scene, rc->scene->mesh_lights[l.sampler], v.ray.direction, v.origin, v.geo_n, frag, v.state.translucent,
                          v.light_split_threshold)
            : ZYGPU_SHAPE_DISTANT == rc->scene->props[l.prop].shape
                ? 1.f / distantSolidAngle(frag.isec.trafo.scaleX())  // distant.zig:139-141
            : ZYGPU_SHAPE_SPHERE == rc->scene->props[l.prop].shape
                ? spherePdf(v.origin, frag.isec.trafo, v.light_split_threshold, l.num_samples)  // sphere.zig:469-485
            : ZYGPU_SHAPE_DOME == rc->scene->props[l.prop].shape ? (v.state.translucent ? 1.f / (4.f * kPi) : 1.f / (2.f * kPi))  // dome.zig:214-220
            : ZYGPU_SHAPE_CANOPY == rc->scene->props[l.prop].shape ? 1.f / (2.f * kPi)  // shape.zig:483
                : rectanglePdf(v.origin, frag.isec.trafo, v.light_split_threshold, l.num_samples);
        return powerHeuristic(v.bxdf_pdf, sample_pdf * select_pdf);
    }

    // vertex.zig:183-212
    V4 evaluateRadiance(const Vertex& v, const Fragment& frag, Sampler& sampler) const {
        const V4             wo = -v.ray.direction;
        const ZygpuMaterial& m = rc->sv.propMaterial(frag.prop, frag.part);
        if (!(m.flags & ZYGPU_MATF_EMISSIVE) || (!(m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo))) return S(0.f);
        const float stochastic_r = sampler.sample1D();  // rs.stochastic_r
        const bool  in_camera = 0 == v.depth_surface;
        const V4    energy = materialRadiance(m, wo, frag.isec.trafo, frag.prop, frag.part, in_camera, frag.uvw[0], frag.uvw[1], stochastic_r);
        const float weight = lightPdf(v, frag);
        return S(weight) * energy;
    }

    // prop_tree.zig:302-356 over the unoccluding TLAS + prop.zig:239-264 + rectangle.zig:188-196
    V4 emission(const Vertex& v, Sampler& sampler) {
        const ZygpuPropTree& tree = rc->scene->unoccluding;
        NodeStack            stack;
        uint32_t             n = 0 == tree.num_nodes ? kNull : 0;
        V4                   energy = S(0.f);
        if (kNull != n) cnt.nodes_fetched += 1;
        while (kNull != n) {
            const ZygpuBvhNode& node = tree.nodes[n];
            if (0 != node.count) {
                for (uint32_t i = node.children_or_first, e = i + node.count; i < e; ++i) {
                    const uint32_t   p = tree.indices[i];
                    const ZygpuProp& prop = rc->scene->props[p];
                    if (!SceneView::visible(prop.flags, v.depth_surface)) continue;
                    cnt.instances_tested += 1;
                    if (!aabbIntersect(rc->scene->aabbs[p], v.ray)) continue;
                    Fragment frag;
                    frag.isec.trafo = rc->sv.propTrafo(p);
                    frag.prop = p;
                    if (ZYGPU_SHAPE_TRIANGLE_MESH == prop.shape) {  // triangle_mesh.zig:379-388 -> triangle_tree.zig:405-483
                        const ZygpuMesh& mesh = rc->scene->meshes[prop.mesh];
                        const MeshView   mv{&mesh};
                        const Ray        local_ray = frag.isec.trafo.worldToObjectRay(v.ray);
                        meshEmissionWalk(mesh, local_ray, cnt, [&](uint32_t i, float t, float u, float vv) {
                            frag.isec.t = t;
                            frag.isec.u = u;
                            frag.isec.v = vv;
                            frag.isec.primitive = i;
                            frag.part = mesh.triangle_parts[i];
                            const uint32_t* tri = mesh.triangles + size_t(i) * 3;
                            const V4        a = mv.position(tri[0]), b = mv.position(tri[1]), c = mv.position(tri[2]);
                            frag.p = frag.isec.trafo.objectToWorldPoint(interpolate3(a, b, c, u, vv));
                            frag.geo_n = frag.isec.trafo.objectToWorldNormal(normalize3(cross3(b - a, c - a)));
                            energy += evaluateRadiance(v, frag, sampler);
                        });
                        continue;
                    }
                    if (ZYGPU_SHAPE_SPHERE == prop.shape) {  // sphere.zig:269-277
                        float t;
                        if (!sphereIntersect(v.ray, frag.isec.trafo, t)) continue;
                        frag.isec.t = t;
                        frag.isec.u = frag.isec.v = 0.f;
                        frag.isec.primitive = 0;
                        sphereFragment(v.ray, frag);
                        energy += evaluateRadiance(v, frag, sampler);
                        continue;
                    }
                    if (ZYGPU_SHAPE_RECTANGLE != prop.shape) continue;
                    float t, u, vv;
                    if (!rectangleIntersect(v.ray, frag.isec.trafo, t, u, vv)) continue;
                    frag.isec.t = t;
                    frag.isec.u = u;
                    frag.isec.v = vv;
                    frag.isec.primitive = 0;
                    rectangleFragment(v.ray, frag);
                    energy += evaluateRadiance(v, frag, sampler);
                }
                n = stack.pop();
                continue;
            }
            uint32_t a = node.children_or_first, b = a + 1;
            cnt.nodes_fetched += 2;
            float dista = nodeIntersect(tree.nodes[a], v.ray);
            float distb = nodeIntersect(tree.nodes[b], v.ray);
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
        return energy;
    }

    // ---- the medium stack's views of the scene (prop/medium.zig:55-116)
    const ZygpuMaterial& mediumMaterial(const MediumStack& m, uint32_t i) const { return rc->sv.propMaterial(m.prop[i], m.part[i]); }
    float mediumIor(const MediumStack& m, uint32_t i) const {  // material.zig:140-146
        const ZygpuMaterial& mat = mediumMaterial(m, i);
        return (ZYGPU_MAT_GLASS == mat.kind || ZYGPU_MAT_SUBSTITUTE == mat.kind) ? mat.ior : 1.f;
    }
    float topIor(const MediumStack& m) const { return m.index > 0 ? mediumIor(m, m.index - 1) : 1.f; }
    float peekIor(const MediumStack& m, const Fragment& frag) const {
        if (m.index <= 1) return 1.f;
        const uint32_t back = m.index - 1;
        return m.matches(back, frag) ? mediumIor(m, back - 1) : mediumIor(m, back);
    }
    int32_t highestPriority(const MediumStack& m) const {
        int32_t priority = -128;
        for (uint32_t i = 0; i < m.index; ++i) priority = std::max(priority, mediumMaterial(m, i).priority);
        return priority;
    }
    V4 topAbsorption(const MediumStack& m) const {  // topCC: the last entry of the highest priority
        int32_t  priority = -128;
        uint32_t highest = 0;
        for (uint32_t i = 0; i < m.index; ++i) {
            const int32_t lp = mediumMaterial(m, i).priority;
            if (lp >= priority) {
                priority = lp;
                highest = i;
            }
        }
        return ld3(mediumMaterial(m, highest).absorption);
    }

    // Context.nextEvent (context.zig:54-69); inside a dielectric: VolumeIntegrator.integrate's branch for materials that do
    // not scatter (volume_integrator.zig:84-129, 62-65): the ray ends at the far side of the medium's bounding box, and
    // the segment to the hit attenuates the throughput by Beer-Lambert
    void nextEvent(Vertex& vertex, Fragment& frag) {
        cnt.extension_rays += 1;
        if (vertex.mediums.empty()) {
            rc->sv.intersect(vertex.ray, vertex.depth_surface, frag, cnt);
            return;
        }
        const uint32_t medium_prop = vertex.mediums.prop[vertex.mediums.index - 1];
        const float    ray_max_t = vertex.ray.max_t;
        const float    limit = aabbIntersectP(rc->scene->aabbs[medium_prop], vertex.ray);
        vertex.ray.max_t = mn(offsetF(limit), ray_max_t);
        if (!rc->sv.intersect(vertex.ray, vertex.depth_surface, frag, cnt)) return;
        const float d = vertex.ray.max_t;
        vertex.throughput *= attenuation3(topAbsorption(vertex.mediums), d - vertex.ray.min_t);
    }

    // Vertex.interfaceChange (vertex.zig:94-108)
    void interfaceChange(Vertex& vertex, V4 dir, const Fragment& frag) const {
        if (frag.sameHemisphere(dir)) vertex.mediums.remove(frag);
        else vertex.mediums.push(frag);
    }

    // vertex.zig:137-181
    MaterialSample vertexSample(const Vertex& v, const Fragment& frag, Sampler& sampler, bool caustics) const {
        const V4             wo = -v.ray.direction;
        const ZygpuMaterial& m = rc->sv.propMaterial(frag.prop, frag.part);
        Renderstate          rs;
        rs.trafo = frag.isec.trafo;
        rs.p = frag.p;
        rs.t = frag.t;
        rs.b = frag.b;
        if ((m.flags & ZYGPU_MATF_TWO_SIDED) && !frag.sameHemisphere(wo)) {
            rs.geo_n = -frag.geo_n;
            rs.n = -frag.n;
        } else {
            rs.geo_n = frag.geo_n;
            rs.n = frag.n;
        }
        rs.origin = v.origin;
        rs.uvw = frag.uvw;
        rs.stochastic_r = sampler.sample1D();
        rs.ior = frag.sameHemisphere(wo) ? topIor(v.mediums) : peekIor(v.mediums, frag);  // iorOutside, vertex.zig:87-92
        rs.prop = frag.prop;
        rs.part = frag.part;
        rs.caustics = caustics;
        rs.highest_priority = highestPriority(v.mediums);
        return materialSample(m, *rc->scene, wo, rs);
    }

    bool causticsResolve(const VState& s) const { return s.primary_ray ? true : 0 != rc->view->caustics_path; }

    // the infinite props a ray reaches when it leaves the scene (Distant only): Context.propIntersect + Distant.intersect /
    // fragment (prop.zig:163-197, distant.zig:22-74), then Vertex.evaluateRadiance; `only_non_lights`: PathtracerDL's filter
    V4 infiniteEmission(const Vertex& v, Sampler& sampler, bool only_non_lights) {
        V4 energy = S(0.f);
        for (uint32_t i = 0; i < rc->scene->num_infinite_props; ++i) {
            const uint32_t   p = rc->scene->infinite_props[i];
            const ZygpuProp& prop = rc->scene->props[p];
            if (!SceneView::visible(prop.flags, v.depth_surface)) continue;
            cnt.instances_tested += 1;
            if (!aabbIntersect(rc->scene->aabbs[p], v.ray)) continue;
            Fragment frag;
            frag.isec.trafo = rc->sv.propTrafo(p);
            float u = 0.f, vv = 0.f;
            if (ZYGPU_SHAPE_DISTANT == prop.shape) {
                if (!distantIntersect(v.ray, frag.isec.trafo, u, vv)) continue;
                frag.geo_n = frag.n = frag.isec.trafo.r[2];
                frag.uvw = mk((u + 1.f) * 0.5f, (vv + 1.f) * 0.5f, 0.f, 0.f);
            } else {  // dome.zig:26-56, canopy.zig:27-58: the whole sphere / the hemisphere around the prop's z axis
                if (v.ray.max_t < kRayMaxT) continue;
                if (ZYGPU_SHAPE_CANOPY == prop.shape && dot3(v.ray.direction, frag.isec.trafo.r[2]) < kCanopyEps) continue;
                frag.geo_n = frag.n = -v.ray.direction;
                float eu, ev;
                if (ZYGPU_SHAPE_DOME == prop.shape) domeUv(v.ray.direction, frag.isec.trafo, eu, ev);
                else canopyUv(v.ray.direction, frag.isec.trafo, eu, ev);
                frag.uvw = mk(eu, ev, 0.f, 0.f);
            }
            frag.prop = p;
            frag.isec.t = kRayMaxT;
            frag.isec.u = u;
            frag.isec.v = vv;
            frag.isec.primitive = 0;
            frag.p = S(kRayMaxT) * v.ray.direction;
            frag.t = frag.isec.trafo.r[0];
            frag.b = frag.isec.trafo.r[1];
            frag.part = 0;
            if (only_non_lights && !(v.state.singular || kNull == rc->sv.propLightId(p, 0))) continue;
            energy += evaluateRadiance(v, frag, sampler);
        }
        return energy;
    }

    // connectLight of the three integrators: pathtracer.zig:110-143, pathtracer_dl.zig:173-207, pathtracer_mis.zig:280-341
    V4 connectLight(Vertex& v, const Fragment& frag, Sampler& sampler) {
        const uint32_t kind = rc->view->integrator;
        if (!rc->view->caustics_path && v.state.specular && !v.state.primary_ray) return S(0.f);

        V4 energy = S(0.f);
        if (ZYGPU_INTEGRATOR_PT == kind) {
            const bool singular = v.state.singular;
            v.state.singular = true;
            if (frag.hit()) energy = evaluateRadiance(v, frag, sampler);
            energy += emission(v, sampler);
            energy += infiniteEmission(v, sampler, false);
            v.state.singular = singular;
        } else if (ZYGPU_INTEGRATOR_PTDL == kind) {
            if (frag.hit()) {
                if (v.state.singular || kNull == rc->sv.propLightId(frag.prop, frag.part)) energy = evaluateRadiance(v, frag, sampler);
            }
            if (v.state.singular) energy += emission(v, sampler);
            energy += infiniteEmission(v, sampler, true);
        } else {
            v.light_split_threshold = splitThreshold(rc->view->split_threshold, v.depth);
            if (frag.hit()) energy = evaluateRadiance(v, frag, sampler);
            energy += emission(v, sampler);
            if (kRayMaxT == v.ray.max_t) energy += infiniteEmission(v, sampler, false);
        }
        return energy;
    }

    // pathtracer_mis.zig:174-278 (MIS) / pathtracer_dl.zig:120-171 (DL)
    V4 sampleLights(const Vertex& v, const Fragment& frag, const MaterialSample& ms, Sampler& sampler, bool mis, uint32_t max_splits = 1) {
        V4 result = S(0.f);
        if (!ms.can_evaluate) return result;

        const V4    p = frag.p;
        const V4    n = ms.geo_n;
        const bool  translucent = ms.translucent;
        const float select = sampler.sample1D();
        const float split_threshold =
            mis ? v.light_split_threshold : splitThreshold(rc->view->split_threshold, v.depth_surface);

        LightPick      lights[64];
        const uint32_t num_lights = rc->lt.randomLight(p, n, translucent, select, split_threshold, lights);

        for (uint32_t li = 0; li < num_lights; ++li) {
            const LightPick&     pick = lights[li];
            const ZygpuLight&    light = rc->scene->lights[pick.offset];
            const Trafo          trafo = rc->sv.propTrafo(light.prop);
            const ZygpuMaterial& lm = rc->sv.propMaterial(light.prop, light.part);

            SampleTo       samples[64];
            const float    sample_threshold = mis ? v.light_split_threshold : 0.f;
            const uint32_t light_shape = rc->scene->props[light.prop].shape;
            const bool     distant = ZYGPU_SHAPE_DISTANT == light_shape || ZYGPU_SHAPE_DOME == light_shape || ZYGPU_SHAPE_CANOPY == light_shape;
            const uint32_t num_samples =
                ZYGPU_LIGHT_PROP_IMAGE == light.light_class  // light.zig:191-215 -> shape.zig:348-371
                    ? (ZYGPU_SHAPE_DOME == light_shape
                           ? domeSampleMaterialTo(rc->scene->image_lights[light.sampler], n, trafo, translucent, sample_threshold,
                                                  light.num_samples, sampler, samples)
                           : canopySampleMaterialTo(rc->scene->image_lights[light.sampler], n, trafo, translucent, sampler, samples))
                : kNull != light.sampler
                    ? meshSampleTo(*rc->scene, rc->scene->mesh_lights[light.sampler], p, n, trafo, 0 != light.two_sided, translucent,
                                   sample_threshold, sampler, samples)
                : ZYGPU_SHAPE_DOME == light_shape   ? domeSampleTo(n, translucent, sampler, samples)
                : ZYGPU_SHAPE_CANOPY == light_shape ? canopySampleTo(n, trafo, translucent, sampler, samples)
                : distant ? distantSampleTo(n, trafo, translucent, sampler, samples)
                : ZYGPU_SHAPE_SPHERE == rc->scene->props[light.prop].shape
                    ? sphereSampleTo(p, n, trafo, translucent, sample_threshold, light.num_samples, sampler, samples)
                    : rectangleSampleTo(p, n, trafo, 0 != light.two_sided, translucent, sample_threshold, light.num_samples, sampler,
                                        samples);

            V4 occluded = S(0.f);
            for (uint32_t si = 0; si < num_samples; ++si) {
                const SampleTo& ls = samples[si];
                const Ray       shadow = shadowRay(fragOffsetP(frag, ls.wi), ls, distant);
                cnt.shadow_rays += 1;
                V4 tr = S(1.f);
                if (!rc->sv.visibility(shadow, cnt, tr)) continue;

                if (mis) {
                    const float      lr = sampler.sample1D();  // light.zig:127 rs.stochastic_r
                    const V4         radiance = materialRadiance(lm, ls.wi, trafo, light.prop, light.part, false, ls.uvw[0], ls.uvw[1], lr);
                    const BxdfResult bx = ms.evaluate(ls.wi, false, max_splits);
                    const float      light_pdf = ls.pdf() * pick.pdf;
                    const float      weight = predividedPowerHeuristic(light_pdf, bx.pdf);
                    occluded += tr * (S(weight) * radiance * bx.reflection);
                } else {
                    const BxdfResult bx = ms.evaluate(ls.wi, false);
                    const float lr = sampler.sample1D();
                    const V4    radiance = materialRadiance(lm, ls.wi, trafo, light.prop, light.part, false, ls.uvw[0], ls.uvw[1], lr);
                    const float weight = 1.f / (pick.pdf * ls.pdf());
                    occluded += S(weight) * (tr * radiance * bx.reflection);
                }
            }
            result += occluded;
        }
        return result;
    }

    // Vertex.Pool (vertex.zig:214-310): the vertices of one camera sample, level by level; a level's vertices are consumed
    // in order and share the worker's samplers
    struct VertexPool {
        static constexpr uint32_t kNumVertices = 4;
        V4       transparency = S(0.f);
        Vertex   buffer[2 * kNumVertices];
        uint32_t terminated = 0, current_id = kNumVertices, current_start = kNumVertices, current_end = kNumVertices, next_start = 0,
                 next_end = 1;

        explicit VertexPool(const Vertex& v) { buffer[0] = v; }
        Vertex* iterate(uint32_t id) {
            terminated |= 1u << (id & 31u);
            current_id = id + 1;
            return &buffer[id];
        }
        Vertex* consume() {
            const uint32_t old_end = current_end;
            if (current_id < old_end) return iterate(current_id);
            for (uint32_t i = current_start; i < old_end; ++i) {
                if (0 == (terminated & (1u << (i & 31u)))) continue;
                const Vertex& v = buffer[i];
                if (v.state.transparent) {
                    transparency[3] += mx((1.f - average3(v.throughput)) * v.split_weight, 0.f);
                } else {  // no shadow catchers: shadow_catcher_emission = 0
                    const float alpha = transparency[3];
                    transparency += S(0.f) * S(v.split_weight);
                    transparency[3] = alpha + v.split_weight;
                }
            }
            current_id = current_start = next_start;
            current_end = next_end;
            next_start = next_end = (kNumVertices == current_start) ? 0 : kNumVertices;
            if (current_start < current_end) return iterate(current_start);
            return nullptr;
        }
        Vertex* newVertex() {
            terminated &= ~(1u << ((current_id - 1) & 31u));
            return &buffer[next_end++];
        }
        static uint32_t maxSplits(const Vertex& v, uint32_t depth) {
            const uint32_t m = kNumVertices / v.path_count;
            return m - (v.state.primary_ray ? 0u : std::min(depth, m - 1));
        }
    };

    // PathtracerMIS.li (pathtracer_mis.zig:37-172)
    IValue liMis(const Vertex& input) {
        const ZygpuViewDesc& view = *rc->view;
        IValue               result;
        VertexPool           vertices(input);
        while (Vertex* vp = vertices.consume()) {
            Vertex&        vertex = *vp;
            const uint32_t total_depth = vertex.depth_surface;
            Sampler&       sampler = pickSampler(total_depth);

            Fragment frag;
            nextEvent(vertex, frag);

            const V4 energy = connectLight(vertex, frag, sampler);
            const V4 split_weight = S(vertex.split_weight);
            V4       split_throughput = vertex.throughput * split_weight;
            result.add(split_throughput * energy, total_depth, 2, 0 == total_depth, vertex.state.singular);

            if (!frag.hit() || vertex.depth_surface >= view.max_depth_surface || 0 >= view.max_depth_volume) {
                if (vertex.state.transparent) vertex.throughput *= S(1.f) - min4(energy, S(1.f));
                continue;
            }
            if (russianRoulette(vertex.throughput, sampler.sample1D())) continue;

            const bool           caustics = causticsResolve(vertex.state);
            const MaterialSample ms = vertexSample(vertex, frag, sampler, caustics);
            split_throughput = vertex.throughput * split_weight;

            vertex.light_split_threshold = splitThreshold(view.split_threshold, vertex.depth_surface);
            const uint32_t max_splits = VertexPool::maxSplits(vertex, total_depth);
            const V4       lighting = sampleLights(vertex, frag, ms, sampler, true, max_splits);
            result.add(split_throughput * lighting, total_depth, 1, false, false);

            BxdfSample     samples[2];
            const uint32_t path_count = ms.sampleSplit(sampler, max_splits, samples);
            if (0 == path_count) vertex.throughput = S(0.f);
            for (uint32_t i = 0; i < path_count; ++i) {
                const BxdfSample& bs = samples[i];
                Vertex&           next = *vertices.newVertex();
                next = vertex;
                next.path_count = vertex.path_count * path_count;
                next.split_weight = vertex.split_weight * bs.split_weight;
                next.state.update(bs.path);
                if (PathEvent::Straight != bs.path.event) {
                    next.state.translucent = ms.translucent;
                    next.depth = next.depth_surface;
                    next.bxdf_pdf = bs.pdf;
                    next.origin = frag.p;
                    next.geo_n = ms.geo_n;
                }
                next.throughput *= bs.reflection / S(bs.pdf);
                next.ray = fragOffsetRay(frag, bs.wi);
                next.depth_surface += 1;
                if (PathEvent::Transmission == bs.path.event) interfaceChange(next, bs.wi, frag);
                next.state.transparent =
                    next.state.transparent && (PathEvent::Transmission == bs.path.event || PathEvent::Straight == bs.path.event);
            }
            sampler.incrementPadding();
        }
        result.direct[3] = 0.f;
        result.direct += vertices.transparency;
        return result;
    }

    IValue li(Vertex vertex) {
        const ZygpuViewDesc& view = *rc->view;
        const uint32_t       kind = view.integrator;
        const bool           mis = ZYGPU_INTEGRATOR_PTMIS == kind;
        if (mis) return liMis(vertex);
        IValue result;

        for (;;) {
            const uint32_t total_depth = vertex.depth_surface;
            Sampler&       sampler = pickSampler(total_depth);

            Fragment frag;
            nextEvent(vertex, frag);

            const V4 energy = connectLight(vertex, frag, sampler);
            result.add(vertex.throughput * energy, total_depth, mis ? 2 : 1, 0 == total_depth, vertex.state.singular);

            if (!frag.hit() || vertex.depth_surface >= view.max_depth_surface || 0 >= view.max_depth_volume) {
                if (mis && vertex.state.transparent) vertex.throughput *= S(1.f) - min4(energy, S(1.f));
                break;
            }

            if (russianRoulette(vertex.throughput, sampler.sample1D())) break;

            const bool           caustics = causticsResolve(vertex.state);
            const MaterialSample ms = vertexSample(vertex, frag, sampler, caustics);

            if (ZYGPU_INTEGRATOR_PT != kind) {
                if (mis) vertex.light_split_threshold = splitThreshold(view.split_threshold, vertex.depth_surface);
                const V4 lighting = sampleLights(vertex, frag, ms, sampler, mis);
                result.add(vertex.throughput * lighting, total_depth, 1, false, false);
            }

            BxdfSample bs;
            if (0 == ms.sample(sampler, bs)) {
                vertex.throughput = S(0.f);
                break;
            }

            vertex.state.update(bs.path);
            if (PathEvent::Straight != bs.path.event) {
                if (mis) {
                    vertex.state.translucent = ms.translucent;
                    vertex.depth = vertex.depth_surface;
                    vertex.bxdf_pdf = bs.pdf;
                    vertex.geo_n = ms.geo_n;
                }
                vertex.origin = frag.p;
            }

            vertex.throughput *= bs.reflection / S(bs.pdf);
            vertex.ray = fragOffsetRay(frag, bs.wi);
            vertex.depth_surface += 1;
            if (PathEvent::Transmission == bs.path.event) interfaceChange(vertex, bs.wi, frag);
            vertex.state.transparent =
                vertex.state.transparent && (PathEvent::Transmission == bs.path.event || PathEvent::Straight == bs.path.event);

            sampler.incrementPadding();
        }

        // alpha: helper.zig:60-62 (PT/DL); vertex.zig:240-287 Pool.consume for one un-split path (MIS)
        const float alpha = vertex.state.transparent ? mx(1.f - average3(vertex.throughput), 0.f) : 1.f;
        result.direct[3] = alpha;
        return result;
    }
};

// ---------------- gilbert.zig ----------------
static int sgn(int x) { return int(0 < x) - int(x < 0); }
static void gilbertR(int dst_idx, int cur_idx, int x, int y, int ax, int ay, int bx, int by, int out[2]) {
    const int w = std::abs(ax + ay), h = std::abs(bx + by);
    const int dax = sgn(ax), day = sgn(ay), dbx = sgn(bx), dby = sgn(by);
    const int di = dst_idx - cur_idx;
    if (1 == h) {
        out[0] = x + dax * di;
        out[1] = y + day * di;
        return;
    }
    if (1 == w) {
        out[0] = x + dbx * di;
        out[1] = y + dby * di;
        return;
    }
    int       ax2 = ax >> 1, ay2 = ay >> 1, bx2 = bx >> 1, by2 = by >> 1;
    const int w2 = std::abs(ax2 + ay2), h2 = std::abs(bx2 + by2);
    if ((2 * w) > (3 * h)) {
        if ((w2 & 1) != 0 && (w > 2)) {
            ax2 += dax;
            ay2 += day;
        }
        const int nxt_idx = cur_idx + std::abs((ax2 + ay2) * (bx + by));
        if ((cur_idx <= dst_idx) && (dst_idx < nxt_idx)) return gilbertR(dst_idx, cur_idx, x, y, ax2, ay2, bx, by, out);
        return gilbertR(dst_idx, nxt_idx, x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by, out);
    }
    if ((h2 & 1) != 0 && (h > 2)) {
        bx2 += dbx;
        by2 += dby;
    }
    const int nxt_idx = cur_idx + std::abs((bx2 + by2) * (ax2 + ay2));
    if ((cur_idx <= dst_idx) && (dst_idx < nxt_idx)) return gilbertR(dst_idx, cur_idx, x, y, bx2, by2, ax2, ay2, out);
    const int nxtnxt_idx = nxt_idx + std::abs((ax + ay) * ((bx - bx2) + (by - by2)));
    if ((nxt_idx <= dst_idx) && (dst_idx < nxtnxt_idx))
        return gilbertR(dst_idx, nxt_idx, x + bx2, y + by2, ax, ay, bx - bx2, by - by2, out);
    const int xres = x + (ax - dax) + (bx2 - dbx), yres = y + (ay - day) + (by2 - dby);
    return gilbertR(dst_idx, nxtnxt_idx, xres, yres, -bx2, -by2, -(ax - ax2), -(ay - ay2), out);
}
static void gilbertD2xy(int idx, int w, int h, int out[2]) {
    if (w >= h) gilbertR(idx, 0, 0, 0, w, 0, 0, h, out);
    else gilbertR(idx, 0, 0, 0, 0, h, w, 0, out);
}

// ---------------- sensor ----------------
struct Film {
    const ZygpuViewDesc* view;
    float*               pixels;  // {sum w r, sum w g, sum w b, sum w}
    float*               alpha = nullptr;  // Transparent sensor: sum of w * alpha per pixel (buffer_transparent.zig:46-55)

    static V4 clampColor(V4 color, float max) {  // sensor.zig:615-624
        const float mc = hmax3(color);
        if (mc > max) return S(max / mc) * color;
        return color;
    }
    float eval(float s) const {  // sensor.zig:626-628 + interpolated_function.zig:154-166
        const float    range_end = view->filter_radius;
        const float    inverse_interval = 1.f / (range_end / float(ZYGPU_FILTER_TAPS - 1));
        const float    cx = mn(std::fabs(s), range_end);
        const float    o = cx * inverse_interval;
        const uint32_t offset = uint32_t(o);
        const float    t = o - float(offset);
        return lerpf(view->filter_samples[offset], view->filter_samples[std::min<uint32_t>(offset + 1, ZYGPU_FILTER_TAPS - 1)], t);
    }
    static void atomicAdd(float* p, float v) {
        auto*    a = reinterpret_cast<std::atomic<uint32_t>*>(p);
        uint32_t old = a->load(std::memory_order_relaxed);
        for (;;) {
            float f;
            std::memcpy(&f, &old, 4);
            f += v;
            uint32_t nw;
            std::memcpy(&nw, &f, 4);
            if (a->compare_exchange_weak(old, nw, std::memory_order_relaxed)) break;
        }
    }
    void add(int px, int py, float weight, V4 color, const int bounds[4], const int isolated[4]) {  // sensor.zig:559-574
        if (uint32_t(px - bounds[0]) <= uint32_t(bounds[2]) && uint32_t(py - bounds[1]) <= uint32_t(bounds[3])) {
            const size_t id = size_t(view->resolution[0]) * size_t(py) + size_t(px);
            float*       p = pixels + id * 4;
            const V4     wc = S(weight) * color;
            if (uint32_t(px - isolated[0]) <= uint32_t(isolated[2]) && uint32_t(py - isolated[1]) <= uint32_t(isolated[3])) {
                p[0] += wc[0];
                p[1] += wc[1];
                p[2] += wc[2];
                p[3] += weight;
                if (alpha) alpha[id] += wc[3];
            } else {
                atomicAdd(p + 0, wc[0]);
                atomicAdd(p + 1, wc[1]);
                atomicAdd(p + 2, wc[2]);
                atomicAdd(p + 3, weight);
                if (alpha) atomicAdd(alpha + id, wc[3]);
            }
        }
    }
    // sensor.zig:168-385 (no AOVs)
    void addSample(int x, int y, float pu, float pv, const IValue& value, const int bounds[4], const int isolated[4]) {
        const V4 emission = clampColor(value.emission, view->clamp_emission);
        const V4 direct = clampColor(value.direct, view->clamp_direct);
        const V4 indirect = clampColor(value.indirect, view->clamp_indirect);
        const V4 summed = emission + direct + indirect;
        const V4 composed = mk(summed[0], summed[1], summed[2], value.direct[3]);

        const int   fr = view->filter_radius_int;
        const float ox = pu - 0.5f, oy = pv - 0.5f;
        if (0 == fr) {
            const size_t id = size_t(view->resolution[0]) * size_t(y) + size_t(x);
            float*       p = pixels + id * 4;
            const V4     wc = S(1.f) * composed;
            if (alpha) alpha[id] += wc[3];
            p[0] += wc[0];
            p[1] += wc[1];
            p[2] += wc[2];
            p[3] += 1.f;
            return;
        }
        float wx[5], wy[5];
        const int n = 2 * fr + 1;
        for (int i = 0; i < n; ++i) {
            wx[i] = eval(ox + float(fr - i));
            wy[i] = eval(oy + float(fr - i));
        }
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) add(x - fr + i, y - fr + j, wx[i] * wy[j], composed, bounds, isolated);
    }
};

// ---------------- Worker.render ----------------
static void renderTile(Worker& w, Film& film, uint32_t frame, const int tile[4], uint32_t iteration, uint32_t num_samples,
                       int y_begin, int y_end) {
    const ZygpuViewDesc& view = *w.rc->view;
    int                  crop[4] = {view.crop[0], view.crop[1], view.crop[2], view.crop[3]};
    crop[2] -= crop[0] + 1;
    crop[3] -= crop[1] + 1;
    const int fr = view.filter_radius_int;
    int       isolated[4] = {tile[0] + fr, tile[1] + fr, tile[2] - fr, tile[3] - fr};
    isolated[2] -= isolated[0];
    isolated[3] -= isolated[1];

    const int      r0 = view.resolution[0] + 2 * fr, r1 = view.resolution[1] + 2 * fr;
    const uint32_t a = uint32_t(r0) * uint32_t(r1);
    const uint64_t o = uint64_t(iteration) * a;
    const uint32_t num_expected_samples = view.num_expected_samples;
    const uint32_t so = iteration / num_expected_samples;

    const Trafo cam_trafo = Trafo::from(view.camera_trafo);

    for (int y = tile[1]; y <= tile[3]; ++y) {
        if (y < y_begin || y >= y_end) continue;
        const uint32_t pixel_n = uint32_t((y + fr) * r0);
        for (int x = tile[0]; x <= tile[2]; ++x) {
            const uint32_t pixel_id = pixel_n + uint32_t(x + fr);
            w.rng.start(0, uint64_t(pixel_id) + o);

            const uint64_t sample_index = uint64_t(pixel_id) * uint64_t(num_expected_samples) + uint64_t(iteration);
            const uint32_t tsi = uint32_t(sample_index);
            const uint32_t seed = uint32_t(sample_index >> 32) + so;
            w.samplers[0].startPixel(tsi, seed);

            for (uint32_t s = 0; s < num_samples; ++s) {
                // Sensor.cameraSample
                float s4[4];
                w.samplers[0].sample4D(s4);
                const float s1 = w.samplers[0].sample1D();
                w.samplers[0].incrementPadding();

                // Perspective.generateVertex
                const float cx = float(x) + s4[0], cy = float(y) + s4[1];
                V4 direction = ld4(view.left_top) + ld4(view.d_x) * S(cx) + ld4(view.d_y) * S(cy);
                V4 origin;
                if (view.lens_radius > 0.f) {
                    float lx, ly;
                    diskConcentric(s4[2], s4[3], lx, ly);
                    origin = mk(lx * view.lens_radius, ly * view.lens_radius, 0.f, 0.f);
                    const V4 focus = S(view.focus_distance / direction[2]) * direction;
                    direction = focus - origin;
                } else {
                    origin = ld4(view.eye_offset);
                }
                // camera_base.zig:121-136 (no shutter distribution configured through the C ABI)
                const double   delta = double(s1), duration = double(view.frame_duration);
                const uint64_t fdi = uint64_t(std::llround(delta * duration));
                const uint64_t time = uint64_t(frame) * view.frame_step + fdi;

                const V4 origin_w = cam_trafo.objectToWorldPoint(origin);
                const V4 direction_w = cam_trafo.objectToWorldVector(normalize3(direction));

                Vertex v;
                v.ray = Ray::init(origin_w, direction_w, 0.f, kRayMaxT);
                v.time = time;
                v.origin = origin_w;

                w.cnt.camera_samples += 1;
                const IValue ivalue = w.li(v);
                film.addSample(x, y, s4[0], s4[1], ivalue, crop, isolated);

                w.samplers[0].incrementSample();
            }
        }
    }
}

struct TileQueue {  // tile_queue.zig:21-87
    int              crop[4], num_tiles[2], tile_dimensions, filter_padding;
    std::atomic<int> current{0};
    void configure(const int c[4], int td, int fp) {
        for (int i = 0; i < 4; ++i) crop[i] = c[i];
        tile_dimensions = td;
        filter_padding = fp;
        num_tiles[0] = int(std::ceil(float(c[2] - c[0]) / float(td)));
        num_tiles[1] = int(std::ceil(float(c[3] - c[1]) / float(td)));
        current = 0;
    }
    bool pop(int tile[4]) {
        const int cur = current.fetch_add(1, std::memory_order_relaxed);
        if (cur >= num_tiles[0] * num_tiles[1]) return false;
        int start[2];
        gilbertD2xy(cur, num_tiles[0], num_tiles[1], start);
        start[0] = start[0] * tile_dimensions + crop[0];
        start[1] = start[1] * tile_dimensions + crop[1];
        int end[2] = {std::min(start[0] + tile_dimensions, crop[2]), std::min(start[1] + tile_dimensions, crop[3])};
        if (filter_padding > 0) {
            if (crop[1] == start[1]) start[1] -= filter_padding;
            if (crop[3] == end[1]) end[1] += filter_padding;
            if (crop[0] == start[0]) start[0] -= filter_padding;
            if (crop[2] == end[0]) end[0] += filter_padding;
        }
        tile[0] = start[0];
        tile[1] = start[1];
        tile[2] = end[0] - 1;
        tile[3] = end[1] - 1;
        return true;
    }
};

}  // namespace orc

using namespace orc;

extern "C" {

// Renders iterations [iteration, iteration + num_iterations) of pixel rows [y_begin, y_end) into `film_rgbw`
// (accumulating).  samples_per_call = 1 is the progressive form every parity run uses (one sample per
// (pixel, iteration), reseeded as worker.zig:141-149); samples_per_call = n > 1 reproduces the reference's one-shot
// loop (worker.zig:153-165) for a single call at `iteration` with num_iterations ignored.
// film_alpha: the Transparent sensor's second plane (one float per pixel), or null for the Opaque sensor
int zyg_oracle_render_alpha(const ZygpuSceneDesc* scene, const ZygpuViewDesc* view, uint32_t frame, uint32_t iteration,
                            uint32_t num_iterations, uint32_t samples_per_call, int32_t y_begin, int32_t y_end, float* film_rgbw,
                            float* film_alpha, int32_t num_threads, ZygpuStats* stats);

int zyg_oracle_render(const ZygpuSceneDesc* scene, const ZygpuViewDesc* view, uint32_t frame, uint32_t iteration,
                      uint32_t num_iterations, uint32_t samples_per_call, int32_t y_begin, int32_t y_end, float* film_rgbw,
                      int32_t num_threads, ZygpuStats* stats) {
    return zyg_oracle_render_alpha(scene, view, frame, iteration, num_iterations, samples_per_call, y_begin, y_end, film_rgbw, nullptr,
                                   num_threads, stats);
}

int zyg_oracle_render_alpha(const ZygpuSceneDesc* scene, const ZygpuViewDesc* view, uint32_t frame, uint32_t iteration,
                            uint32_t num_iterations, uint32_t samples_per_call, int32_t y_begin, int32_t y_end, float* film_rgbw,
                            float* film_alpha, int32_t num_threads, ZygpuStats* stats) {
    if (!scene || !view || !film_rgbw) return -1;
    RenderCtx rc{scene, view, SceneView{scene}, LightTreeView{scene}};

    const int fr = view->filter_radius_int;
    int       yb = y_begin, ye = y_end;
    if (fr > 0) {  // edge rows own the filter padding, as tiles at the crop edge do (tile_queue.zig:66-82)
        if (yb <= view->crop[1]) yb = view->crop[1] - fr;
        if (ye >= view->crop[3]) ye = view->crop[3] + fr;
    }

    const uint32_t nthreads = uint32_t(std::max(1, num_threads));
    std::vector<Worker> workers(nthreads);
    for (auto& w : workers) {
        w.rc = &rc;
        w.samplers[0].is_sobol = ZYGPU_SAMPLER_SOBOL == view->sampler;
        w.samplers[0].rng = &w.rng;
        w.samplers[1].is_sobol = false;
        w.samplers[1].rng = &w.rng;
    }

    Film film{view, film_rgbw, film_alpha};

    const uint32_t passes = samples_per_call > 1 ? 1 : num_iterations;
    const uint32_t ns = samples_per_call > 1 ? samples_per_call : 1;
    for (uint32_t it = 0; it < passes; ++it) {
        TileQueue tiles;
        tiles.configure(view->crop, 32, fr);  // worker.zig:46
        auto run = [&](uint32_t id) {
            int tile[4];
            while (tiles.pop(tile)) renderTile(workers[id], film, frame, tile, iteration + it, ns, yb, ye);
        };
        if (1 == nthreads) {
            run(0);
        } else {
            std::vector<std::thread> th;
            for (uint32_t i = 0; i < nthreads; ++i) th.emplace_back(run, i);
            for (auto& t : th) t.join();
        }
    }

    if (stats) {
        Counters c;
        for (auto& w : workers) c.add(w.cnt);
        stats->camera_samples += c.camera_samples;
        stats->extension_rays += c.extension_rays;
        stats->shadow_rays += c.shadow_rays;
        stats->nodes_fetched += c.nodes_fetched;
        stats->triangles_tested += c.triangles_tested;
        stats->instances_tested += c.instances_tested;
    }
    return 0;
}

}  // extern "C"

// Tonemapper.tonemap (tonemapper.zig:18-71) with aces.zig:19-64 and agx.zig:14-101 (look Substitute, take_loader.zig:268-270)
static V4 tonemap(uint32_t cls, V4 scaled) {
    auto ap1ToSrgb = [](V4 c) {
        return mk(1.70505155f, -0.13025714f, -0.02400328f) * S(c[0]) + mk(-0.62179068f, 1.14080289f, -0.12896877f) * S(c[1]) +
               mk(-0.08325840f, -0.01054853f, 1.15297171f) * S(c[2]);
    };
    if (ZYGPU_TONEMAP_ACES == cls) {
        const V4 rrt = mk(0.970889f * scaled[0] + 0.026963f * scaled[1] + 0.002148f * scaled[2],
                          0.010889f * scaled[0] + 0.986963f * scaled[1] + 0.002148f * scaled[2],
                          0.010889f * scaled[0] + 0.026963f * scaled[1] + 0.962148f * scaled[2]);
        const V4 a = rrt * (rrt + S(0.0245786f)) - S(0.000090537f);
        const V4 b = rrt * (S(0.983729f) * rrt + S(0.4329510f)) + S(0.238081f);
        const V4 odt = mk(a[0] / b[0], a[1] / b[1], a[2] / b[2]);
        return mk(1.60475f * odt[0] - 0.53108f * odt[1] - 0.07367f * odt[2], -0.10208f * odt[0] + 1.10813f * odt[1] - 0.00605f * odt[2],
                  -0.00327f * odt[0] - 0.07276f * odt[1] + 1.07602f * odt[2]);
    }
    const V4 srgb = ap1ToSrgb(scaled);
    if (ZYGPU_TONEMAP_AGX == cls) {
        V4 val = mk(0.8424790622530940f * srgb[0] + 0.0784335999999992f * srgb[1] + 0.0792237451477643f * srgb[2],
                    0.0423282422610123f * srgb[0] + 0.8784686364697720f * srgb[1] + 0.0791661274605434f * srgb[2],
                    0.0423756549057051f * srgb[0] + 0.0784336000000000f * srgb[1] + 0.8791429737931040f * srgb[2]);
        const float min_ev = -12.47393f, max_ev = 4.026069f;
        for (int k = 0; k < 3; ++k) val[k] = (clampf(std::log2(val[k]), min_ev, max_ev) - min_ev) / (max_ev - min_ev);
        const V4 x = val, x2 = x * x, x4 = x2 * x2;
        V4 c = S(15.5f) * x4 * x2 - S(40.14f) * x4 * x + S(31.96f) * x4 - S(6.868f) * x2 * x + S(0.4298f) * x2 + S(0.1191f) * x - S(0.00232f);
        for (int k = 0; k < 3; ++k) c[k] = mx(c[k], 0.f);
        const float luma = dot3(c, mk(0.2126f, 0.7152f, 0.072f));
        V4          l;
        for (int k = 0; k < 3; ++k) l[k] = luma + 1.05f * (std::pow(c[k] * 1.f, 1.1f) - luma);
        const V4 lin = mk(1.19687900512017f * l[0] - 0.0980208811401368f * l[1] - 0.0990297440797205f * l[2],
                          -0.0528968517574562f * l[0] + 1.15190312990417f * l[1] - 0.0989611768448433f * l[2],
                          -0.0529716355144438f * l[0] - 0.0980434501171241f * l[1] + 1.15107367264116f * l[2]);
        return mk(mx(std::pow(lin[0], 2.2f), 0.f), mx(std::pow(lin[1], 2.2f), 0.f), mx(std::pow(lin[2], 2.2f), 0.f));
    }
    if (ZYGPU_TONEMAP_PBR_NEUTRAL == cls) {
        const float start_compression = 0.8f - 0.04f, desaturation = 0.15f;
        const float x = mn(srgb[0], mn(srgb[1], srgb[2]));
        const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
        V4          color = srgb - S(offset);
        const float peak = hmax3(color);
        if (peak < start_compression) return color;
        const float d = 1.f - start_compression;
        const float new_peak = 1.f - d * d / (peak + d - start_compression);
        color = color * S(new_peak / peak);
        const float g = 1.f - 1.f / (desaturation * (peak - new_peak) + 1.f);
        return lerp4(color, S(new_peak), S(g));
    }
    return srgb;
}

extern "C" {
// buffer_opaque.zig:73-79 + tonemapper.zig:18-46 + aces.zig:19-27
int zyg_oracle_resolve_alpha(const ZygpuViewDesc* view, const float* film_rgbw, const float* film_alpha, float* out_rgba,
                             uint32_t num_pixels);

int zyg_oracle_resolve(const ZygpuViewDesc* view, const float* film_rgbw, float* out_rgba, uint32_t num_pixels) {
    return zyg_oracle_resolve_alpha(view, film_rgbw, nullptr, out_rgba, num_pixels);
}

// buffer_opaque.zig:73-79 / buffer_transparent.zig:84-94 (resolveTonemap), Linear tonemapper
int zyg_oracle_resolve_alpha(const ZygpuViewDesc* view, const float* film_rgbw, const float* film_alpha, float* out_rgba,
                             uint32_t num_pixels) {
    const uint32_t n = std::min<uint32_t>(num_pixels, uint32_t(view->resolution[0]) * uint32_t(view->resolution[1]));
    for (uint32_t i = 0; i < n; ++i) {
        const float* p = film_rgbw + size_t(i) * 4;
        const V4     q = mk(p[0], p[1], p[2], 0.f) / S(p[3]);
        const V4     color = mk(std::fabs(q[0]), std::fabs(q[1]), std::fabs(q[2]), std::fabs(q[3]));
        const V4     scaled = S(view->exposure_factor) * color;
        const V4     srgb = tonemap(view->tonemapper, scaled);
        float* o = out_rgba + size_t(i) * 4;
        o[0] = srgb[0];
        o[1] = srgb[1];
        o[2] = srgb[2];
        o[3] = film_alpha ? std::fabs(film_alpha[i] / p[3]) : 1.f;
    }
    return 0;
}

// ---- unit-test hooks ----

// rays: 8 floats each (o.xyz, min_t, d.xyz, max_t); hits: 8 words each (t, u, v as floats; primitive, prop as u32; 3 pad)
int zyg_oracle_intersect(const ZygpuSceneDesc* scene, uint32_t n, const float* rays, uint32_t depth_surface, uint32_t* hits) {
    SceneView sv{scene};
    Counters  cnt;
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = rays + size_t(i) * 8;
        Ray          ray = Ray::init(mk(r[0], r[1], r[2]), mk(r[4], r[5], r[6]), r[3], r[7]);
        Fragment     frag;
        const bool   hit = sv.intersect(ray, depth_surface, frag, cnt);
        uint32_t*    h = hits + size_t(i) * 8;
        std::memset(h, 0, 32);
        if (hit) {
            std::memcpy(h + 0, &frag.isec.t, 4);
            std::memcpy(h + 1, &frag.isec.u, 4);
            std::memcpy(h + 2, &frag.isec.v, 4);
            h[3] = frag.isec.primitive;
        }
        h[4] = frag.prop;
    }
    return 0;
}

int zyg_oracle_visibility(const ZygpuSceneDesc* scene, uint32_t n, const float* rays, uint32_t* visible) {
    SceneView sv{scene};
    Counters  cnt;
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = rays + size_t(i) * 8;
        const Ray    ray = Ray::init(mk(r[0], r[1], r[2]), mk(r[4], r[5], r[6]), r[3], r[7]);
        visible[i] = sv.visibility(ray, cnt) ? 1u : 0u;
    }
    return 0;
}

void zyg_oracle_pcg(uint64_t state, uint64_t sequence, uint32_t n, uint32_t* out_uints, float* out_floats) {
    Pcg g;
    g.start(state, sequence);
    for (uint32_t i = 0; i < n; ++i) out_uints[i] = g.randomUint();
    g.start(state, sequence);
    for (uint32_t i = 0; i < n; ++i) out_floats[i] = g.randomFloat();
}

uint32_t zyg_oracle_hash(uint32_t x) { return sobol::hash(x); }
void     zyg_oracle_sobol5(uint32_t index, uint32_t* out5) { sobol::sobol5(index, out5); }
void     zyg_oracle_sobol_directions(uint32_t* out160) { std::memcpy(out160, sobol::directions(), 160 * 4); }

// draws `n` 1D samples after startPixel(sample, seed), with a padding increment every `pad_every` draws (0 = never)
void zyg_oracle_sobol_stream(uint32_t sample, uint32_t seed, uint32_t n, uint32_t pad_every, float* out) {
    Sobol s;
    s.startPixel(sample, seed);
    for (uint32_t i = 0; i < n; ++i) {
        out[i] = s.sample1D();
        if (pad_every && 0 == ((i + 1) % pad_every)) s.incrementPadding();
    }
}

void zyg_oracle_gilbert(int32_t idx, int32_t w, int32_t h, int32_t* out_xy) { gilbertD2xy(idx, w, h, out_xy); }

// the restated sin / cos / acos (omath.hpp), for accuracy tests against a float64 reference
void zyg_oracle_trig(uint32_t n, const float* x, float* out_sin, float* out_cos, float* out_acos) {
    for (uint32_t i = 0; i < n; ++i) {
        out_sin[i] = sinF(x[i]);
        out_cos[i] = cosF(x[i]);
        out_acos[i] = acosF(x[i]);
    }
}

void zyg_oracle_atan2(uint32_t n, const float* y, const float* x, float* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = atan2F(y[i], x[i]);
}

void zyg_oracle_exp(uint32_t n, const float* x, float* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = expF(x[i]);
}

void zyg_oracle_offset_ray(const float* p, const float* n, float* out) {
    const V4 r = offsetRay(mk(p[0], p[1], p[2]), mk(n[0], n[1], n[2]));
    out[0] = r[0];
    out[1] = r[1];
    out[2] = r[2];
}

// picks: pairs (offset, pdf bits); returns the number of picks
uint32_t zyg_oracle_random_light(const ZygpuSceneDesc* scene, const float* p, const float* n, uint32_t total_sphere, float random,
                                 float split_threshold, uint32_t* out_offsets, float* out_pdfs) {
    LightTreeView  lt{scene};
    LightPick      picks[64];
    const uint32_t c = lt.randomLight(mk(p[0], p[1], p[2]), mk(n[0], n[1], n[2]), 0 != total_sphere, random, split_threshold, picks);
    for (uint32_t i = 0; i < c; ++i) {
        out_offsets[i] = picks[i].offset;
        out_pdfs[i] = picks[i].pdf;
    }
    return c;
}

float zyg_oracle_light_pdf(const ZygpuSceneDesc* scene, const float* p, const float* n, uint32_t total_sphere, float split_threshold,
                           uint32_t id) {
    LightTreeView lt{scene};
    return lt.pdf(mk(p[0], p[1], p[2]), mk(n[0], n[1], n[2]), 0 != total_sphere, split_threshold, id);
}

// solid angle and pdf of a rectangle light as seen from p (SphQuad)
void zyg_oracle_sphquad(const ZygpuTrafo* trafo, const float* p, float* out_solid_angle, float* out_pdf) {
    const Trafo   t = Trafo::from(*trafo);
    const SphQuad q = SphQuad::init(t.scale(), t.worldToFramePoint(mk(p[0], p[1], p[2])));
    *out_solid_angle = q.Sa;
    *out_pdf = q.pdf(t.scale());
}

}  // extern "C"

extern "C" {
// SphQuad on explicit inputs: in = scale.xy, o.xyz, u0, u1 -> out = S, pdf, sample.xyz
void zyg_oracle_sphquad_eval(uint32_t n, const float* in, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float*  a = in + 7 * i;
        const V4      scale = mk(a[0], a[1], 1.f, 1.f);
        const SphQuad q = SphQuad::init(scale, mk(a[2], a[3], a[4]));
        const V4      sp = q.sample(a[5], a[6]);
        float*        o = out + 5 * i;
        o[0] = q.Sa;
        o[1] = q.pdf(scale);
        o[2] = sp[0];
        o[3] = sp[1];
        o[4] = sp[2];
    }
}
}

extern "C" void zyg_oracle_basic_ops(uint32_t n, const float* in, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float a = in[2 * i], b = in[2 * i + 1];
        out[6 * i] = a / b;
        out[6 * i + 1] = std::sqrt(std::fabs(a));
        out[6 * i + 2] = a * b + a;
        out[6 * i + 3] = 1.f / a;
        out[6 * i + 4] = acosR(std::fabs(a) * 0.25f);
        const float z = std::fabs(a) * 0.25f;
        out[6 * i + 5] = z * (1.6666586697e-01f + z * (-4.2743422091e-02f + z * -8.6563630030e-03f));
    }
}
