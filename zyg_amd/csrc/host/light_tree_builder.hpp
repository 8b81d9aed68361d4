// Scene-level light BVH builder (Conty-Estevez/Kulla, adaptive splitting): host-side scene
// build product.  Follows reference
// src/core/scene/light/light_tree_builder.zig:281-387 (build), 445-492 (split), 570-600 (assign),
// 622-660 (serialize, variance), 662-790 (evaluateSplits), 813-821 (coneCost) and
// light_tree.zig:37-52 (compressCenter).  Only finite, non-prototype lights are in scope
// (infinite lights: sky/sun are SURVEY §8(f) "next").
#pragma once

#include <vector>

#include "../../../include/zygpu.h"
#include "hmath.hpp"

namespace zyg {

struct LightTreeInput {
    const ZygpuAabb* aabbs;   // scene lights: min[3] = power, max[3] = radius; mesh triangles: plain boxes
    const float*     cones;   // 4 per light
    const uint8_t*   two_sided;         // per light, or nullptr with all_two_sided (a mesh part)
    uint32_t         num_lights;
    const float*     powers = nullptr;  // per light; nullptr: aabbs[l].min[3]
    bool             all_two_sided = false;
    const uint8_t*   infinite = nullptr;  // per light: its shape is infinite (Distant); nullptr: all finite
    const uint8_t*   prototype = nullptr; // per light: never sampled itself (a Portal's linked light, scene.zig:558-576); nullptr: none
};

struct LightTreeOutput {
    std::vector<ZygpuLightNode> nodes;
    std::vector<uint32_t>       node_middles;
    std::vector<uint32_t>       light_orders;
    std::vector<uint32_t>       light_mapping;
    std::vector<float>          infinite_cdf;  // Distribution1D over the infinite lights' powers (num_infinite + 1 entries)
    ZygpuLightTree              tree{};
};

class LightTreeBuilder {
  public:
    static constexpr uint32_t MaxSplitDepth = 10;  // light_tree.zig:248
    static constexpr uint32_t MaxLights = 64;      // light_tree.zig:249
    static constexpr uint32_t SceneSweepThreshold = 128;
    static constexpr uint32_t NumSlices = 16;

    void build(const LightTreeInput& in, LightTreeOutput& out) {
        in_ = &in;
        out_ = &out;
        sweep_threshold_ = SceneSweepThreshold;
        primitive_ = false;

        // scene.numLights() / numSampleableLights() (light_tree_builder.zig:281-282): prototypes are in neither list
        const uint32_t num_all_lights = in.num_lights;
        auto           sampleable = [&](uint32_t l) { return !(in.prototype && in.prototype[l]); };
        uint32_t       num_lights = 0;
        for (uint32_t l = 0; l < num_all_lights; ++l) num_lights += sampleable(l) ? 1u : 0u;
        out.light_mapping.resize(num_lights);
        out.light_orders.assign(num_all_lights, 0);
        light_order_ = 0;

        // light_tree_builder.zig:289-330: infinite lights first (sampled from a discrete distribution over their powers, never
        // part of the BVH), then the finite ones
        uint32_t lm = 0;
        for (uint32_t l = 0; l < num_all_lights; ++l)
            if (sampleable(l) && in.infinite && in.infinite[l]) out.light_mapping[lm++] = l;
        const uint32_t num_infinite_lights = lm;
        for (uint32_t l = 0; l < num_all_lights; ++l)
            if (sampleable(l) && !(in.infinite && in.infinite[l])) out.light_mapping[lm++] = l;

        float              infinite_total_power = 0.f;
        std::vector<float> infinite_powers(num_infinite_lights);
        for (uint32_t i = 0; i < num_infinite_lights; ++i) {
            const uint32_t l = out.light_mapping[i];
            infinite_powers[i] = lightPower(l);
            out.light_orders[l] = light_order_;
            light_order_ += 1;
            infinite_total_power += infinite_powers[i];
        }
        const uint32_t infinite_end = light_order_;
        // Distribution1D.precomputePdfCdf (base/math/distribution_1d.zig:87-124)
        out.infinite_cdf.assign(size_t(num_infinite_lights) + 1, 0.f);
        if (num_infinite_lights > 0) {
            float integral = 0.f;
            for (float d : infinite_powers) integral += d;
            if (0.f == integral) {
                out.infinite_cdf.assign(2, 1.f);
            } else {
                const float ii = 1.f / integral;
                float       p = 0.f;
                for (uint32_t i = 0; i + 1 < num_infinite_lights; ++i) {
                    const float c = std::fmaf(infinite_powers[i], ii, p);
                    out.infinite_cdf[i + 1] = c;
                    p = c;
                }
                out.infinite_cdf[num_infinite_lights] = 1.f;
            }
        }
        const uint32_t num_finite_lights = num_lights - num_infinite_lights;

        uint32_t max_split_depth = MaxSplitDepth;

        ZygpuLightTree& tree = out.tree;
        std::memset(&tree, 0, sizeof(tree));

        if (num_finite_lights > 0) {
            build_nodes_.assign(2 * num_finite_lights - 1, BuildNode{});
            current_node_ = 1;

            AABB  bounds = emptyAabb();
            Vec4f cone = splat(1.f);
            bool  two_sided = false;
            float total_power = 0.f;

            for (uint32_t i = num_infinite_lights; i < num_lights; ++i) {
                const uint32_t l = out.light_mapping[i];
                aabbMerge(bounds, lightAabb(l));
                cone = coneMerge(cone, lightCone(l));
                two_sided = two_sided || lightTwoSided(l);
                total_power += lightPower(l);
            }

            split(0, num_infinite_lights, num_lights, bounds, cone, two_sided, total_power, 0);

            out.nodes.resize(current_node_);
            out.node_middles.resize(current_node_);
            aabbCacheRadius(build_nodes_[0].bounds);
            serialize(build_nodes_[0].bounds);

            const AABB& b = build_nodes_[0].bounds;
            for (int i = 0; i < 4; ++i) {
                tree.bounds.min[i] = b.b[0][i];
                tree.bounds.max[i] = b.b[1][i];
            }

            // light_tree_builder.zig:352-366
            uint32_t split_lights[MaxSplitDepth][2];
            std::memset(split_lights, 0, sizeof(split_lights));
            countPotentialLights(0, 0, split_lights);

            uint32_t num_split_lights = 0;
            for (uint32_t i = 0; i < MaxSplitDepth; ++i) {
                num_split_lights += split_lights[i][0];
                if ((num_split_lights + split_lights[i][1]) > (MaxLights - num_infinite_lights) || 0 == split_lights[i][1]) {
                    max_split_depth = i;
                    break;
                }
            }
        }

        tree.max_split_depth = max_split_depth;

        const float p0 = infinite_total_power;
        const float p1 = 0 == num_finite_lights ? 0.f : build_nodes_[0].power;
        const float pt = p0 + p1;
        const float infinite_weight = (0 == num_lights || 0.f == pt) ? 0.f : p0 / pt;

        tree.infinite_weight = infinite_weight;
        // light_tree_builder.zig:381-385 ("afraid of the 1.0 == random case")
        tree.infinite_guard = 0 == num_finite_lights ? (0 == num_infinite_lights ? 0.f : 1.1f) : infinite_weight;
        tree.infinite_end = infinite_end;
        tree.infinite_cdf = out.infinite_cdf.data();
        tree.num_lights = num_lights;
        tree.num_infinite_lights = num_infinite_lights;
        tree.num_nodes = uint32_t(out.nodes.size());
        tree.nodes = out.nodes.data();
        tree.node_middles = out.node_middles.data();
        tree.light_orders = out.light_orders.data();
        tree.light_mapping = out.light_mapping.data();
    }

    // light_tree_builder.zig:389-428 (buildPrimitive): the tree over the emissive triangles of one mesh part.
    // `in` holds per-triangle boxes, (normal, 1) cones and relative powers; the part's own box, cone and power seed the root.
    void buildPrimitive(const LightTreeInput& in, const AABB& part_bounds, const Vec4f& part_cone, float total_power,
                        LightTreeOutput& out) {
        in_ = &in;
        out_ = &out;
        sweep_threshold_ = PartSweepThreshold;
        primitive_ = true;

        const uint32_t num_lights = in.num_lights;
        out.light_mapping.resize(num_lights);
        out.light_orders.assign(num_lights, 0);
        light_order_ = 0;
        for (uint32_t l = 0; l < num_lights; ++l) out.light_mapping[l] = l;

        ZygpuLightTree& tree = out.tree;
        std::memset(&tree, 0, sizeof(tree));
        if (num_lights > 0) {
            build_nodes_.assign(2 * num_lights - 1, BuildNode{});
            current_node_ = 1;
            splitPrimitive(0, 0, num_lights, part_bounds, part_cone, total_power);
            out.nodes.resize(current_node_);
            out.node_middles.resize(current_node_);
            aabbCacheRadius(build_nodes_[0].bounds);
            serialize(build_nodes_[0].bounds);
            const AABB& b = build_nodes_[0].bounds;
            for (int i = 0; i < 4; ++i) {
                tree.bounds.min[i] = b.b[0][i];
                tree.bounds.max[i] = b.b[1][i];
            }
        }
        tree.max_split_depth = PrimitiveMaxSplitDepth;
        tree.num_lights = num_lights;
        tree.num_nodes = uint32_t(out.nodes.size());
        tree.nodes = out.nodes.data();
        tree.node_middles = out.node_middles.data();
        tree.light_orders = out.light_orders.data();
        tree.light_mapping = out.light_mapping.data();
    }

    static constexpr uint32_t PartSweepThreshold = 32;       // light_tree_builder.zig:24
    static constexpr uint32_t PrimitiveMaxSplitDepth = 6;    // light_tree.zig:523

  private:
    struct BuildNode {  // light_tree_builder.zig:27-58
        AABB     bounds;
        Vec4f    cone;
        float    power;
        float    variance;
        uint32_t middle;
        uint32_t children_or_light;
        uint32_t num_lights;
        bool     two_sided;
    };

    struct Candidate {  // light_tree_builder.zig:60-120
        enum Kind { Axis, Angle, Partition } kind;
        float    d;
        uint32_t axis;
        Vec4f    n;
        uint32_t part_num;
        uint32_t part_left[2];

        AABB  aabbs[2];
        Vec4f cones[2];
        float powers[2];
        float cost;
        bool  two_sided[2];
        bool  exhausted;
    };

    AABB lightAabb(uint32_t l) const {
        const ZygpuAabb& a = in_->aabbs[l];
        return {{vec(a.min[0], a.min[1], a.min[2], a.min[3]), vec(a.max[0], a.max[1], a.max[2], a.max[3])}};
    }
    Vec4f lightCone(uint32_t l) const {
        const float* c = in_->cones + 4 * l;
        return vec(c[0], c[1], c[2], c[3]);
    }
    bool  lightTwoSided(uint32_t l) const { return in_->two_sided ? 0 != in_->two_sided[l] : in_->all_two_sided; }
    float lightPower(uint32_t l) const { return in_->powers ? in_->powers[l] : in_->aabbs[l].min[3]; }

    bool leftSide(const Candidate& c, uint32_t l) const {
        switch (c.kind) {
            case Candidate::Axis: return lightAabb(l).b[1][c.axis] < c.d;
            case Candidate::Angle: return dot3(c.n, lightCone(l)) < 0.f;
            default:
                for (uint32_t i = 0; i < c.part_num; ++i)
                    if (l == c.part_left[i]) return true;
                return false;
        }
    }

    static float coneCost(float cos, bool two_sided) {  // light_tree_builder.zig:813-821
        const float pi = float(M_PI);
        const float o = two_sided ? pi : std::acos(cos);
        const float w = fmin_(o + (pi / 2.f), pi);
        const float sin = std::sin(o);
        const float b = (pi / 2.f) * (2.f * w * sin - std::cos(o - 2.f * w) - 2.f * o * sin + cos);
        return (2.f * pi) * (1.f - cos) + b;
    }

    // light_tree_builder.zig:135-195 (evaluateScene)
    void evaluate(Candidate& c, const uint32_t* lights, uint32_t len, const AABB& bounds, float cone_weight) const {
        uint32_t num_sides[2] = {0, 0};
        AABB     boxs[2] = {emptyAabb(), emptyAabb()};
        Vec4f    cones[2] = {splat(1.f), splat(1.f)};
        bool     two_sided[2] = {false, false};
        float    powers[2] = {0.f, 0.f};

        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t l = lights[i];
            const float    power = lightPower(l);
            if (0.f == power) continue;

            const uint32_t side = leftSide(c, l) ? 0 : 1;
            num_sides[side] += 1;
            aabbMerge(boxs[side], lightAabb(l));
            cones[side] = coneMerge(cones[side], lightCone(l));
            two_sided[side] = two_sided[side] || lightTwoSided(l);
            powers[side] += power;
        }

        const Vec4f extent = aabbExtent(bounds);
        const float surface_area = aabbSurfaceArea(bounds);

        for (int i = 0; i < 2; ++i) {
            c.aabbs[i] = boxs[i];
            c.cones[i] = cones[i];
            c.powers[i] = powers[i];
            c.two_sided[i] = two_sided[i];
        }

        const bool empty_side = 0 == num_sides[0] || 0 == num_sides[1];
        if (empty_side) {
            const float reg = hmax3(extent) / hmin3(extent);
            c.cost = float(len) * reg * (powers[0] + powers[1]);
            c.exhausted = true;
        } else {
            const float maxe = hmax3(extent);
            const float reg = Candidate::Axis == c.kind ? maxe / extent[c.axis] : maxe / hmin3(extent);

            const float cone_weight_a = coneCost(cones[0][3], two_sided[0]);
            const float cone_weight_b = coneCost(cones[1][3], two_sided[1]);
            const float surface_area_a = aabbSurfaceArea(boxs[0]);
            const float surface_area_b = aabbSurfaceArea(boxs[1]);

            c.cost = reg * (((powers[0] * cone_weight_a * surface_area_a) + (powers[1] * cone_weight_b * surface_area_b)) /
                            (surface_area * cone_weight));
            c.exhausted = false;
        }
    }

    // light_tree_builder.zig:197-266 (evaluateSampler): the side cones come from the power-weighted dominant axis
    void evaluateSampler(Candidate& c, const uint32_t* lights, uint32_t len, const AABB& bounds, float cone_weight) const {
        uint32_t num_sides[2] = {0, 0};
        AABB     boxs[2] = {emptyAabb(), emptyAabb()};
        Vec4f    dominant_axis[2] = {splat(0.f), splat(0.f)};
        float    powers[2] = {0.f, 0.f};

        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t l = lights[i];
            const float    power = lightPower(l);
            if (0.f == power) continue;
            const uint32_t side = leftSide(c, l) ? 0 : 1;
            num_sides[side] += 1;
            aabbMerge(boxs[side], lightAabb(l));
            dominant_axis[side] = dominant_axis[side] + splat(power) * lightCone(l);
            powers[side] += power;
        }
        dominant_axis[0] = normalize3(dominant_axis[0] / splat(powers[0]));
        dominant_axis[1] = normalize3(dominant_axis[1] / splat(powers[1]));

        float angles[2] = {0.f, 0.f};
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t l = lights[i];
            if (0.f == lightPower(l)) continue;
            const uint32_t side = leftSide(c, l) ? 0 : 1;
            const float    cs = clamp_(dot3(dominant_axis[side], lightCone(l)), -1.f, 1.f);
            angles[side] = fmax_(angles[side], std::acos(cs));
        }

        const bool two_sided = in_->all_two_sided;
        for (int i = 0; i < 2; ++i) {
            c.aabbs[i] = boxs[i];
            c.cones[i] = vec(dominant_axis[i][0], dominant_axis[i][1], dominant_axis[i][2], std::cos(angles[i]));
            c.powers[i] = powers[i];
            c.two_sided[i] = two_sided;
        }

        const Vec4f extent = aabbExtent(bounds);
        const bool  empty_side = 0 == num_sides[0] || 0 == num_sides[1];
        if (empty_side) {
            const float reg = hmax3(extent) / hmin3(extent);
            c.cost = float(len) * reg * (powers[0] + powers[1]);
            c.exhausted = true;
        } else {
            const float surface_area = aabbSurfaceArea(bounds);
            const float maxe = hmax3(extent);
            const float reg = Candidate::Axis == c.kind ? maxe / extent[c.axis] : maxe / hmin3(extent);
            const float cone_weight_a = coneCost(c.cones[0][3], two_sided);
            const float cone_weight_b = coneCost(c.cones[1][3], two_sided);
            const float surface_area_a = aabbSurfaceArea(boxs[0]);
            const float surface_area_b = aabbSurfaceArea(boxs[1]);
            c.cost = reg * (((powers[0] * cone_weight_a * surface_area_a) + (powers[1] * cone_weight_b * surface_area_b)) /
                            (surface_area * cone_weight));
            c.exhausted = false;
        }
    }

    static Candidate axisCandidate(const Vec4f& p, uint32_t axis) {
        Candidate c{};
        c.kind = Candidate::Axis;
        c.d = p[axis];
        c.axis = axis;
        return c;
    }
    static Candidate partitionCandidate(uint32_t a) {
        Candidate c{};
        c.kind = Candidate::Partition;
        c.part_num = 1;
        c.part_left[0] = a;
        return c;
    }
    static Candidate partitionCandidate(uint32_t a, uint32_t b) {
        Candidate c = partitionCandidate(a);
        c.part_num = 2;
        c.part_left[1] = b;
        return c;
    }

    // light_tree_builder.zig:662-790
    Candidate evaluateSplits(const uint32_t* lights, uint32_t len, const AABB& bounds, const Vec4f& cone, bool two_sided) {
        candidates_.clear();

        if (2 == len) {
            candidates_.push_back(partitionCandidate(lights[0]));
        } else if (3 == len) {
            for (int i = 0; i < 3; ++i) candidates_.push_back(partitionCandidate(lights[i]));
        } else if (4 == len) {
            for (int i = 0; i < 4; ++i) candidates_.push_back(partitionCandidate(lights[i]));
            for (int i = 1; i < 4; ++i) candidates_.push_back(partitionCandidate(lights[0], lights[i]));
        } else {
            if (len <= sweep_threshold_) {
                for (uint32_t i = 0; i < len; ++i) {
                    const Vec4f mx = lightAabb(lights[i]).b[1];
                    for (uint32_t a = 0; a < 3; ++a) candidates_.push_back(axisCandidate(mx, a));
                }
            } else {
                const Vec4f position = aabbPosition(bounds);
                const Vec4f extent = aabbExtent(bounds);
                const Vec4f mn = bounds.b[0];

                const uint32_t la = indexMaxComponent3(extent);
                const float    step = extent[la] / float(NumSlices);

                for (uint32_t a = 0; a < 3; ++a) {
                    const float    extent_a = extent[a];
                    const uint32_t num_steps = uint32_t(std::ceil(extent_a / step));
                    const float    step_a = extent_a / float(num_steps);
                    for (uint32_t i = 1; i < num_steps; ++i) {
                        Vec4f slice = position;
                        slice[a] = mn[a] + float(i) * step_a;
                        candidates_.push_back(axisCandidate(slice, a));
                    }
                }
            }

            // light_tree_builder.zig:749-754: the three angle candidates share one slot; the last (cone axis) survives
            Candidate c{};
            c.kind = Candidate::Angle;
            c.n = cone;
            candidates_.push_back(c);
        }

        const float cone_weight = coneCost(cone[3], two_sided);

        for (auto& c : candidates_) {
            if (primitive_) evaluateSampler(c, lights, len, bounds, cone_weight);
            else evaluate(c, lights, len, bounds, cone_weight);
        }

        float  min_cost = candidates_[0].cost;
        size_t sc = 0;
        for (size_t i = 1; i < candidates_.size(); ++i) {
            if (candidates_[i].cost < min_cost) {
                sc = i;
                min_cost = candidates_[i].cost;
            }
        }
        return candidates_[sc];
    }

    float variance(const uint32_t* lights, uint32_t len) const {  // light_tree_builder.zig:642-659
        float    ap = 0.f, aps = 0.f;
        uint32_t n = 0;
        for (uint32_t i = 0; i < len; ++i) {
            const float p = lightPower(lights[i]);
            if (p > 0.f) {
                n += 1;
                const float in = 1.f / float(n);
                ap += (p - ap) * in;
                aps += (p * p - aps) * in;
            }
        }
        return std::fabs(aps - ap * ap);
    }

    uint32_t assign(uint32_t node_id, uint32_t begin, uint32_t end, const AABB& bounds, const Vec4f& cone, float total_power) {
        uint32_t*      lights = out_->light_mapping.data() + begin;
        const uint32_t len = end - begin;

        bool node_two_sided = false;
        for (uint32_t i = 0; i < len; ++i) {
            out_->light_orders[lights[i]] = light_order_;
            light_order_ += 1;
            node_two_sided = node_two_sided || lightTwoSided(lights[i]);
        }

        BuildNode& node = build_nodes_[node_id];
        node.bounds = bounds;
        node.cone = cone;
        node.power = total_power;
        node.variance = variance(lights, len);
        node.middle = 0;
        node.children_or_light = begin;
        node.num_lights = len;
        node.two_sided = node_two_sided;
        return begin + len;
    }

    // base/memory/partition.zig:3-30
    uint32_t partition(uint32_t* data, uint32_t len, const Candidate& sc) const {
        uint32_t first = len;
        for (uint32_t i = 0; i < len; ++i) {
            if (!leftSide(sc, data[i])) {
                first = i;
                break;
            }
        }
        if (first == len) return first;
        for (uint32_t i = first + 1; i < len; ++i) {
            if (leftSide(sc, data[i])) {
                std::swap(data[i], data[first]);
                first += 1;
            }
        }
        return first;
    }

    // light_tree_builder.zig:445-492
    uint32_t split(uint32_t node_id, uint32_t begin, uint32_t end, const AABB& bounds, const Vec4f& cone, bool two_sided,
                   float total_power, uint32_t depth) {
        uint32_t*      lights = out_->light_mapping.data() + begin;
        const uint32_t len = end - begin;

        if (1 == len || (2 == len && depth > MaxSplitDepth)) {
            return assign(node_id, begin, end, bounds, cone, total_power);
        }

        const uint32_t child0 = current_node_;

        const Candidate sc = evaluateSplits(lights, len, bounds, cone, two_sided);
        if (sc.exhausted) {
            return assign(node_id, begin, end, bounds, cone, total_power);
        }

        const uint32_t split_node = begin + partition(lights, len, sc);

        current_node_ += 2;
        const uint32_t c0_end =
            split(child0, begin, split_node, sc.aabbs[0], sc.cones[0], sc.two_sided[0], sc.powers[0], depth + 1);
        const uint32_t c1_end =
            split(child0 + 1, split_node, end, sc.aabbs[1], sc.cones[1], sc.two_sided[1], sc.powers[1], depth + 1);

        BuildNode& node = build_nodes_[node_id];
        node.bounds = bounds;
        node.cone = cone;
        node.power = total_power;
        node.variance = variance(lights, len);
        node.middle = c0_end;
        node.children_or_light = child0;
        node.num_lights = len;
        node.two_sided = two_sided;
        return c1_end;
    }

    // light_tree_builder.zig:494-541 (splitPrimitive) + 596-620 (assignPrimitive)
    uint32_t splitPrimitive(uint32_t node_id, uint32_t begin, uint32_t end, const AABB& bounds, const Vec4f& cone, float total_power) {
        uint32_t*      lights = out_->light_mapping.data() + begin;
        const uint32_t len = end - begin;
        const bool     two_sided = in_->all_two_sided;

        auto assign_leaf = [&]() {
            for (uint32_t i = 0; i < len; ++i) {
                out_->light_orders[lights[i]] = light_order_;
                light_order_ += 1;
            }
            BuildNode& node = build_nodes_[node_id];
            node.bounds = bounds;
            node.cone = cone;
            node.power = total_power;
            node.variance = variance(lights, len);
            node.middle = 0;
            node.children_or_light = begin;
            node.num_lights = len;
            node.two_sided = two_sided;
            return begin + len;
        };

        if (len <= 4) return assign_leaf();

        const uint32_t  child0 = current_node_;
        const Candidate sc = evaluateSplits(lights, len, bounds, cone, two_sided);
        if (sc.exhausted) return assign_leaf();

        const uint32_t split_node = begin + partition(lights, len, sc);
        current_node_ += 2;
        const uint32_t c0_end = splitPrimitive(child0, begin, split_node, sc.aabbs[0], sc.cones[0], sc.powers[0]);
        const uint32_t c1_end = splitPrimitive(child0 + 1, split_node, end, sc.aabbs[1], sc.cones[1], sc.powers[1]);

        BuildNode& node = build_nodes_[node_id];
        node.bounds = bounds;
        node.cone = cone;
        node.power = total_power;
        node.variance = variance(lights, len);
        node.middle = c0_end;
        node.children_or_light = child0;
        node.num_lights = len;
        node.two_sided = two_sided;
        return c1_end;
    }

    void countPotentialLights(uint32_t node_id, uint32_t depth, uint32_t num_lights[MaxSplitDepth][2]) const {
        const BuildNode& n = build_nodes_[node_id];
        if (!(n.middle > 0)) {
            num_lights[depth][0] += 1;
        } else {
            num_lights[depth][1] += 2;
            const uint32_t next_depth = depth + 1;
            if (next_depth < MaxSplitDepth) {
                countPotentialLights(n.children_or_light, next_depth, num_lights);
                countPotentialLights(n.children_or_light + 1, next_depth, num_lights);
            }
        }
    }

    // light_tree_builder.zig:622-640 + light_tree.zig:37-52
    void serialize(const AABB& total_bounds) {
        for (uint32_t i = 0; i < current_node_; ++i) {
            const BuildNode& source = build_nodes_[i];
            ZygpuLightNode&  dest = out_->nodes[i];

            const AABB&  bounds = source.bounds;
            const Vec4f  p = aabbPosition(bounds);
            const Vec4f  center = vec(p[0], p[1], p[2], 0.5f * length3(aabbExtent(bounds)));
            const Vec4f  d = center - total_bounds.b[0];
            const Vec4f  e = aabbExtent(total_bounds);
            const Vec4f  div = vec(0.f == e[0] ? 1.f : e[0], 0.f == e[1] ? 1.f : e[1], 0.f == e[2] ? 1.f : e[2],
                                   total_bounds.b[1][3]);
            const Vec4f  q = d / div;
            for (int k = 0; k < 4; ++k) {
                dest.center[k] = floatToUnorm16(q[k]);
                dest.cone[k] = floatToSnorm16(source.cone[k]);
            }
            dest.power = source.power;
            dest.variance = source.variance;
            const bool has_children = source.middle > 0;
            dest.meta = (has_children ? 1u : 0u) | (source.two_sided ? 2u : 0u) | (source.children_or_light << 2);
            dest.num_lights = source.num_lights;
            out_->node_middles[i] = source.middle;
        }
    }

    uint32_t               sweep_threshold_ = SceneSweepThreshold;
    bool                   primitive_ = false;  // evaluateSampler instead of evaluateScene
    const LightTreeInput*  in_ = nullptr;
    LightTreeOutput*       out_ = nullptr;
    std::vector<BuildNode> build_nodes_;
    std::vector<Candidate> candidates_;
    uint32_t               current_node_ = 0;
    uint32_t               light_order_ = 0;
};

}  // namespace zyg
