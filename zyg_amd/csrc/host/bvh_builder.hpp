// SAH BVH2 builder with spatial splits: host-side scene build product.
// Follows the behaviour of reference src/core/scene/bvh/builder_base.zig:65-245 and
// split_candidate.zig:75-198 (binned 16 slices / sweep <= 64 refs / leaf <= 4, spatial
// split candidates near the root, child boxes clipped to the parent).
//
// The reference defers sub-trees to parallel tasks and merges them in task order
// (builder_base.zig:83-95,330-390); both of its serializers then renumber nodes depth-first
// (triangle_tree_builder.zig:166-206, prop_tree_builder.zig:64-96), so the emitted node and
// primitive order is independent of the task split.  This builder emits that serialized order directly: small inputs by one
// depth-first recursion; large ones (round 5: a million triangles took 3.6 s on one core) in three steps that give the same arrays
// bit for bit: (1) the top of the tree by the same recursion, with the candidates of a large node evaluated on all threads and
// every sub-tree of at most `task_references` references set aside as a task; (2) the tasks on all threads, each into arrays of
// its own; (3) one depth-first pass that numbers the nodes as the plain recursion would and splices the tasks' arrays in.
#pragma once

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../../include/zygpu.h"
#include "hmath.hpp"

namespace zyg {

struct BvhReference {  // split_candidate.zig:10-70
    float    mn[3];
    uint32_t prim;
    float    mx[3];
};

struct BvhSettings {
    uint32_t num_slices = 16;
    uint32_t sweep_threshold = 64;
    uint32_t max_primitives = 4;
    uint32_t parallel_from = 32768;  // references from which a build runs on several threads (0: never)
    uint32_t threads = 0;            // 0: the machine's, at most 16
};

class BvhBuilder {
  public:
    explicit BvhBuilder(BvhSettings s = BvhSettings()) : settings_(s) {}

    // Output: nodes in depth-first serialized order (children adjacent: a, a+1), leaf nodes hold
    // (first, count) into `primitives`, which lists primitive ids in leaf order (with duplicates
    // where spatial splits put one primitive into several leaves).
    void build(std::vector<BvhReference> references, const AABB& bounds, std::vector<ZygpuBvhNode>& nodes,
               std::vector<uint32_t>& primitives) {
        nodes.clear();
        primitives.clear();
        if (references.empty()) return;

        // builder_base.zig:310-313
        const float log2n = std::log2(float(references.size()));
        spatial_split_threshold_ = uint32_t(std::round(log2n / 2.f));

        const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
        num_threads_ = settings_.threads ? settings_.threads : std::min(16u, hw);
        if (0 == settings_.parallel_from || references.size() < settings_.parallel_from || num_threads_ < 2) {
            nodes_ = &nodes;
            prims_ = &primitives;
            nodes.push_back(ZygpuBvhNode{});
            split(0, std::move(references), bounds, 0);
            return;
        }
        // (1) the top: a skeleton whose deferred sub-trees are single nodes (count = kDeferred, children_or_first = the task)
        std::vector<ZygpuBvhNode> skeleton;
        std::vector<uint32_t>     skeleton_prims;
        task_references_ = std::max<size_t>(2048, references.size() / (size_t(num_threads_) * 8));
        tasks_.clear();
        nodes_ = &skeleton;
        prims_ = &skeleton_prims;
        skeleton.push_back(ZygpuBvhNode{});
        split(0, std::move(references), bounds, 0);
        task_references_ = 0;
        // (2) the tasks, largest first
        std::vector<uint32_t> order(tasks_.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tasks_[a].refs.size() > tasks_[b].refs.size(); });
        std::atomic<uint32_t> next{0};
        auto                  work = [&]() {
            BvhBuilder local(settings_);
            local.spatial_split_threshold_ = spatial_split_threshold_;
            for (;;) {
                const uint32_t i = next.fetch_add(1);
                if (i >= order.size()) return;
                Task& t = tasks_[order[i]];
                local.nodes_ = &t.nodes;
                local.prims_ = &t.prims;
                t.nodes.push_back(ZygpuBvhNode{});
                local.split(0, std::move(t.refs), t.aabb, t.depth);
            }
        };
        {
            std::vector<std::thread> pool;
            for (uint32_t t = 1; t < num_threads_; ++t) pool.emplace_back(work);
            work();
            for (std::thread& t : pool) t.join();
        }
        // (3) the plain recursion's numbering: a node's two children take the next two indices when the node is visited, the first
        // child's sub-tree is numbered before the second's
        nodes.push_back(ZygpuBvhNode{});
        struct Visit {
            uint32_t skeleton, out;
        };
        std::vector<Visit> stack{{0u, 0u}};
        while (!stack.empty()) {
            const Visit v = stack.back();
            stack.pop_back();
            const ZygpuBvhNode& sn = skeleton[v.skeleton];
            if (kDeferred == sn.count) {
                const Task&    t = tasks_[sn.children_or_first];
                const uint32_t node_base = uint32_t(nodes.size()), prim_base = uint32_t(primitives.size());
                auto           remap = [&](ZygpuBvhNode n) {
                    if (0 == n.count) n.children_or_first = node_base + n.children_or_first - 1u;
                    else n.children_or_first += prim_base;
                    return n;
                };
                nodes[v.out] = remap(t.nodes[0]);
                for (size_t k = 1; k < t.nodes.size(); ++k) nodes.push_back(remap(t.nodes[k]));
                primitives.insert(primitives.end(), t.prims.begin(), t.prims.end());
            } else if (0 != sn.count) {
                ZygpuBvhNode n = sn;
                n.children_or_first = uint32_t(primitives.size());
                for (uint32_t k = 0; k < sn.count; ++k) primitives.push_back(skeleton_prims[sn.children_or_first + k]);
                nodes[v.out] = n;
            } else {
                ZygpuBvhNode n = sn;
                n.children_or_first = uint32_t(nodes.size());
                nodes[v.out] = n;
                nodes.push_back(ZygpuBvhNode{});
                nodes.push_back(ZygpuBvhNode{});
                stack.push_back({sn.children_or_first + 1u, n.children_or_first + 1u});  // (popped after the first child's whole sub-tree)
                stack.push_back({sn.children_or_first, n.children_or_first});
            }
        }
        tasks_.clear();
    }

  private:
    struct Candidate {  // split_candidate.zig:75-92
        AABB     aabbs[2];
        uint32_t num_sides[2];
        float    d;
        float    cost;
        uint8_t  axis;
        bool     spatial;
    };

    static bool behind(const Candidate& c, const float* p) { return p[c.axis] < c.d; }

    static void mergeRef(AABB& a, const BvhReference& r) {
        a.b[0] = min4(a.b[0], vec(r.mn[0], r.mn[1], r.mn[2], 0.f));
        a.b[1] = max4(a.b[1], vec(r.mx[0], r.mx[1], r.mx[2], 0.f));
    }

    // split_candidate.zig:94-163
    static void evaluate(Candidate& c, const std::vector<BvhReference>& refs, float aabb_surface_area) {
        uint32_t num_sides[2] = {0, 0};
        AABB     aabbs[2] = {emptyAabb(), emptyAabb()};

        if (c.spatial) {
            bool used_spatial = false;
            for (const auto& r : refs) {
                if (behind(c, r.mx)) {
                    num_sides[0] += 1;
                    mergeRef(aabbs[0], r);
                } else if (!behind(c, r.mn)) {
                    num_sides[1] += 1;
                    mergeRef(aabbs[1], r);
                } else {
                    num_sides[0] += 1;
                    num_sides[1] += 1;
                    mergeRef(aabbs[0], r);
                    mergeRef(aabbs[1], r);
                    used_spatial = true;
                }
            }
            if (used_spatial) {
                aabbs[0].b[1][c.axis] = fmin_(c.d, aabbs[0].b[1][c.axis]);  // clipMax
                aabbs[1].b[0][c.axis] = fmax_(c.d, aabbs[1].b[0][c.axis]);  // clipMin
            } else {
                c.spatial = false;
            }
        } else {
            for (const auto& r : refs) {
                if (behind(c, r.mx)) {
                    num_sides[0] += 1;
                    mergeRef(aabbs[0], r);
                } else {
                    num_sides[1] += 1;
                    mergeRef(aabbs[1], r);
                }
            }
        }

        const bool empty_side = 0 == num_sides[0] || 0 == num_sides[1];
        if (empty_side) {
            c.cost = 2.f + float(refs.size());
        } else {
            const float weight0 = float(num_sides[0]) * aabbSurfaceArea(aabbs[0]);
            const float weight1 = float(num_sides[1]) * aabbSurfaceArea(aabbs[1]);
            const float duplication_penalty = 0.125f * float(num_sides[0] + num_sides[1] - uint32_t(refs.size()));
            c.cost = 2.f + (weight0 + weight1) / aabb_surface_area + duplication_penalty;
        }

        c.num_sides[0] = num_sides[0];
        c.num_sides[1] = num_sides[1];
        c.aabbs[0] = aabbs[0];
        c.aabbs[1] = aabbs[1];
    }

    static Candidate makeCandidate(uint8_t axis, const Vec4f& p, bool spatial) {
        Candidate c;
        c.d = p[axis];
        c.axis = axis;
        c.spatial = spatial;
        c.cost = 0.f;
        return c;
    }

    // builder_base.zig:151-245; returns false for "no usable plane"
    bool splittingPlane(const std::vector<BvhReference>& refs, const AABB& aabb, uint32_t depth, Candidate& out) {
        candidates_.clear();

        const float aabb_surface_area = aabbSurfaceArea(aabb);
        if (0.f == aabb_surface_area) return false;

        const uint32_t num_references = uint32_t(refs.size());
        const Vec4f    position = aabbPosition(aabb);

        for (uint8_t a = 0; a < 3; ++a) candidates_.push_back(makeCandidate(a, position, true));

        if (num_references <= settings_.sweep_threshold) {
            for (const auto& r : refs) {
                const Vec4f mx = vec(r.mx[0], r.mx[1], r.mx[2], 0.f);
                for (uint8_t a = 0; a < 3; ++a) candidates_.push_back(makeCandidate(a, mx, false));
            }
        } else {
            const Vec4f extent = aabbExtent(aabb);
            const Vec4f mn = aabb.b[0];

            const uint32_t la = indexMaxComponent3(extent);
            const float    step = extent[la] / float(settings_.num_slices);

            for (uint8_t a = 0; a < 3; ++a) {
                const float extent_a = extent[a];
                uint32_t    num_steps = uint32_t(std::ceil(extent_a / step));
                if (num_steps < 1) num_steps = 1;
                const float step_a = extent_a / float(num_steps);

                for (uint32_t i = 1; i < num_steps; ++i) {
                    const float fi = float(i);
                    Vec4f       slice = position;
                    slice[a] = mn[a] + fi * step_a;
                    candidates_.push_back(makeCandidate(a, slice, false));
                    if (depth < spatial_split_threshold_) candidates_.push_back(makeCandidate(a, slice, true));
                }
            }
        }

        if (task_references_ && refs.size() >= 8192 && num_threads_ > 1) {  // the top of a large tree: the candidates on all threads
            std::atomic<size_t>      next{0};
            auto                     work = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < candidates_.size();) evaluate(candidates_[i], refs, aabb_surface_area);
            };
            std::vector<std::thread> pool;
            for (uint32_t t = 1; t < num_threads_; ++t) pool.emplace_back(work);
            work();
            for (std::thread& t : pool) t.join();
        } else {
            for (auto& c : candidates_) evaluate(c, refs, aabb_surface_area);
        }

        size_t sc = 0;
        float  min_cost = candidates_[0].cost;
        for (size_t i = 1; i < candidates_.size(); ++i) {
            if (candidates_[i].cost < min_cost) {
                sc = i;
                min_cost = candidates_[i].cost;
            }
        }

        const Candidate& sp = candidates_[sc];
        if ((aabbCovers(sp.aabbs[0], aabb) && num_references == sp.num_sides[0]) ||
            (aabbCovers(sp.aabbs[1], aabb) && num_references == sp.num_sides[1])) {
            return false;
        }

        out = sp;
        return true;
    }

    // split_candidate.zig:165-193
    static void distribute(const Candidate& sp, const std::vector<BvhReference>& refs, std::vector<BvhReference>& r0,
                           std::vector<BvhReference>& r1) {
        r0.reserve(sp.num_sides[0]);
        r1.reserve(sp.num_sides[1]);
        if (sp.spatial) {
            for (const auto& r : refs) {
                if (behind(sp, r.mx)) {
                    r0.push_back(r);
                } else if (!behind(sp, r.mn)) {
                    r1.push_back(r);
                } else {
                    BvhReference a = r, b = r;
                    a.mx[sp.axis] = fmin_(sp.d, a.mx[sp.axis]);  // clippedMax
                    b.mn[sp.axis] = fmax_(sp.d, b.mn[sp.axis]);  // clippedMin
                    r0.push_back(a);
                    r1.push_back(b);
                }
            }
        } else {
            for (const auto& r : refs) {
                if (behind(sp, r.mx))
                    r0.push_back(r);
                else
                    r1.push_back(r);
            }
        }
    }

    void setAabb(uint32_t node_id, const AABB& box) {
        ZygpuBvhNode& n = (*nodes_)[node_id];
        for (int i = 0; i < 3; ++i) {
            n.min[i] = box.b[0][i];
            n.max[i] = box.b[1][i];
        }
    }

    void assign(uint32_t node_id, const std::vector<BvhReference>& refs) {
        ZygpuBvhNode& n = (*nodes_)[node_id];
        n.children_or_first = uint32_t(prims_->size());
        n.count = uint32_t(refs.size());
        for (const auto& r : refs) prims_->push_back(r.prim);
    }

    // builder_base.zig:65-149
    void split(uint32_t node_id, std::vector<BvhReference> refs, const AABB& aabb, uint32_t depth) {
        setAabb(node_id, aabb);

        const uint32_t num_primitives = uint32_t(refs.size());
        if (num_primitives <= settings_.max_primitives) {
            assign(node_id, refs);
            return;
        }
        if (task_references_ && num_primitives <= task_references_) {  // (the skeleton pass of a parallel build: this sub-tree is a task)
            ZygpuBvhNode& n = (*nodes_)[node_id];
            n.children_or_first = uint32_t(tasks_.size());
            n.count = kDeferred;
            tasks_.push_back(Task{std::move(refs), aabb, depth, {}, {}});
            return;
        }

        Candidate sp;
        if (splittingPlane(refs, aabb, depth, sp)) {
            if (num_primitives <= 0xFF && float(num_primitives) <= sp.cost) {
                assign(node_id, refs);
            } else {
                std::vector<BvhReference> r0, r1;
                distribute(sp, refs, r0, r1);

                if (num_primitives <= 0x2FF && (r0.empty() || r1.empty())) {
                    assign(node_id, refs);
                } else {
                    std::vector<BvhReference>().swap(refs);

                    const uint32_t child0 = uint32_t(nodes_->size());
                    (*nodes_)[node_id].children_or_first = child0;
                    (*nodes_)[node_id].count = 0;
                    nodes_->push_back(ZygpuBvhNode{});
                    nodes_->push_back(ZygpuBvhNode{});

                    const AABB b0 = aabbIntersection(sp.aabbs[0], aabb);
                    const AABB b1 = aabbIntersection(sp.aabbs[1], aabb);
                    // depth-first: child0's whole sub-tree is numbered before child1's, matching the
                    // reference serializers' renumbering
                    split(child0, std::move(r0), b0, depth + 1);
                    split(child0 + 1, std::move(r1), b1, depth + 1);
                }
            }
        } else {
            // builder_base.zig:144-148 logs an error above 0x2FF primitives; a leaf is the only option
            assign(node_id, refs);
        }
    }

    static constexpr uint32_t kDeferred = 0xFFFFFFFFu;
    struct Task {
        std::vector<BvhReference> refs;
        AABB                      aabb;
        uint32_t                  depth;
        std::vector<ZygpuBvhNode> nodes;
        std::vector<uint32_t>     prims;
    };
    std::vector<Task>          tasks_;
    size_t                     task_references_ = 0;  // > 0: the skeleton pass of a parallel build
    uint32_t                   num_threads_ = 1;
    BvhSettings                settings_;
    uint32_t                   spatial_split_threshold_ = 0;
    std::vector<Candidate>     candidates_;
    std::vector<ZygpuBvhNode>* nodes_ = nullptr;
    std::vector<uint32_t>*     prims_ = nullptr;
};

}  // namespace zyg
