// Wide re-layout of a mesh's BLAS for the device (north_star "BVH nodes laid out for wavefront traversal").
//
// The reference's BVH2 (bvh/node.zig:9-20: 32 B nodes, children adjacent) makes a ray's walk a chain of dependent 64 B
// loads, one per level: ~19 levels for a million triangles, most of them beyond the 4 MB L2.  collapse4() turns the same tree
// (the same SAH / spatial splits, the same leaves with the same triangle ranges in the same order) into nodes of four
// children: a child slot is a BVH2-format node (min.xyz, ref | max.xyz, count), a BVH4 node is four slots = one 128 B line.
// Half the levels, half the nodes (n / 3 instead of n), one line per step.
//
// Collapse rule: a node's children start as its two BVH2 children; while there is room the inner child with the largest surface
// area is replaced (in place, keeping left / right order) by its own two children.
#pragma once

#include <cfloat>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <utility>
#include <vector>

#include "../../../include/zygpu.h"

namespace zyg {

struct Bvh4Slot {  // the reference node layout: leaf iff count != 0; inner: ref = BVH4 node index; empty: a far point
    float    min[3];
    uint32_t ref;
    float    max[3];
    uint32_t count;
};
static_assert(sizeof(Bvh4Slot) == 32, "slot = BVH2 node");

// nodes: the BVH2 of one mesh (root at 0, root inner).  Returns 4 slots per BVH4 node, root node first.
inline std::vector<Bvh4Slot> collapse4(const ZygpuBvhNode* nodes, uint32_t num_nodes) {
    std::vector<Bvh4Slot> out;
    if (0 == num_nodes) return out;
    if (0 != nodes[0].count) {  // the whole mesh is one leaf: a node with that one child
        out.resize(4);
        for (uint32_t k = 0; k < 4; ++k) {
            Bvh4Slot& s = out[k];
            for (int a = 0; a < 3; ++a) {
                s.min[a] = 0 == k ? nodes[0].min[a] : FLT_MAX;
                s.max[a] = 0 == k ? nodes[0].max[a] : FLT_MAX;
            }
            s.ref = 0 == k ? nodes[0].children_or_first : 0xFFFFFFFFu;
            s.count = 0 == k ? nodes[0].count : 0;
        }
        return out;
    }
    auto area = [&](uint32_t n) {
        const float dx = nodes[n].max[0] - nodes[n].min[0], dy = nodes[n].max[1] - nodes[n].min[1], dz = nodes[n].max[2] - nodes[n].min[2];
        return 2.f * (dx * dy + dx * dz + dy * dz);
    };
    struct Todo {
        uint32_t bvh2, node4;
    };
    std::vector<Todo> todo{{0u, 0u}};
    out.resize(4);
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        uint32_t kids[4] = {nodes[t.bvh2].children_or_first, nodes[t.bvh2].children_or_first + 1, 0, 0};
        uint32_t n = 2;
        while (n < 4) {
            int   best = -1;
            float best_area = -1.f;
            for (uint32_t k = 0; k < n; ++k)
                if (0 == nodes[kids[k]].count && area(kids[k]) > best_area) {
                    best = int(k);
                    best_area = area(kids[k]);
                }
            if (best < 0) break;
            const uint32_t c = nodes[kids[best]].children_or_first;
            for (uint32_t k = n; k > uint32_t(best) + 1; --k) kids[k] = kids[k - 1];
            kids[best] = c;
            kids[best + 1] = c + 1;
            n += 1;
        }
        for (uint32_t k = 0; k < 4; ++k) {
            Bvh4Slot s;
            if (k < n) {
                const ZygpuBvhNode& c = nodes[kids[k]];
                for (int a = 0; a < 3; ++a) {
                    s.min[a] = c.min[a];
                    s.max[a] = c.max[a];
                }
                if (0 != c.count) {
                    s.ref = c.children_or_first;
                    s.count = c.count;
                } else {
                    s.ref = uint32_t(out.size() / 4);
                    s.count = 0;
                    todo.push_back({kids[k], s.ref});
                    out.resize(out.size() + 4);
                }
            } else {
                // an empty slot is the point (FLT_MAX, FLT_MAX, FLT_MAX): the slab test (node.zig:73-87) sorts each axis'
                // two distances, so an inverted box would read as a huge one; both distances of a far point have the same
                // sign and the test fails for every ray
                for (int a = 0; a < 3; ++a) s.min[a] = s.max[a] = FLT_MAX;
                s.ref = 0xFFFFFFFFu;
                s.count = 0;
            }
            out[size_t(t.node4) * 4 + k] = s;
        }
    }
    return out;
}

// Levels of inner nodes below (and including) the root.  A walk parks at most three children per level it descends, so a tree
// of depth d needs a stack of 3 d entries at most (dscene.hpp kStack4); zygpu_upload_scene keeps deeper trees on the BVH2 kernels.
inline uint32_t depth4(const std::vector<Bvh4Slot>& slots) {
    if (slots.empty()) return 0;
    uint32_t deepest = 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};  // (node, its level)
    while (!todo.empty()) {
        const auto [node, level] = todo.back();
        todo.pop_back();
        deepest = std::max(deepest, level);
        for (uint32_t k = 0; k < 4; ++k) {
            const Bvh4Slot& c = slots[size_t(node) * 4 + k];
            if (0xFFFFFFFFu != c.ref && 0 == c.count) todo.push_back({c.ref, level + 1});
        }
    }
    return deepest;
}

// The four-wide nodes with their child boxes quantised to 8 bits per coordinate relative to the node's own box (64 B per node
// instead of 128): word 0-2 the node's lower corner, word 3 the three scales as biased binary exponents (x | y << 8 | z << 16),
// words 4-9 the children's qmin.x / .y / .z and qmax.x / .y / .z (child k in byte k), words 10-13 their references, word 14 their
// leaf counts (byte k; 0 = inner node or empty), word 15 unused.  A child's box is corner + q * 2^(e - 127) per axis, rounded
// outwards, so it contains the exact box: the walk visits a superset of the nodes, the triangles are tested as ever.  An empty
// slot has reference 0xFFFFFFFF.  Returns nothing when a leaf holds more than 255 triangles.
inline std::vector<uint32_t> compress4(const std::vector<Bvh4Slot>& slots) {
    std::vector<uint32_t> out(slots.size() / 4 * 16, 0u);
    for (size_t n = 0; n < slots.size() / 4; ++n) {
        const Bvh4Slot* c = &slots[4 * n];
        float           lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int k = 0; k < 4; ++k) {
            if (0xFFFFFFFFu == c[k].ref) continue;
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], c[k].min[a]);
                hi[a] = std::max(hi[a], c[k].max[a]);
            }
        }
        uint32_t* w = &out[16 * n];
        uint32_t  exps = 0;
        float     scale[3];
        for (int a = 0; a < 3; ++a) {
            std::memcpy(&w[a], &lo[a], 4);
            const float extent = hi[a] - lo[a];
            int         e = 1;  // biased exponent: the smallest power of two with 250 steps covering the extent
            while (e < 254 && std::ldexp(250.f, e - 127) < extent) ++e;
            scale[a] = std::ldexp(1.f, e - 127);
            exps |= uint32_t(e) << (8 * a);
        }
        w[3] = exps;
        for (int k = 0; k < 4; ++k) {
            w[10 + k] = c[k].ref;
            if (0xFFFFFFFFu == c[k].ref) continue;
            if (c[k].count > 255u) return {};
            w[14] |= c[k].count << (8 * k);
            for (int a = 0; a < 3; ++a) {
                int qmin = int(std::floor((c[k].min[a] - lo[a]) / scale[a]));
                int qmax = int(std::ceil((c[k].max[a] - lo[a]) / scale[a]));
                qmin = std::max(0, std::min(qmin, 255));
                qmax = std::max(0, std::min(qmax, 255));
                while (qmin > 0 && std::fmaf(float(qmin), scale[a], lo[a]) > c[k].min[a]) --qmin;
                while (qmax < 255 && std::fmaf(float(qmax), scale[a], lo[a]) < c[k].max[a]) ++qmax;
                if (std::fmaf(float(qmin), scale[a], lo[a]) > c[k].min[a] || std::fmaf(float(qmax), scale[a], lo[a]) < c[k].max[a]) return {};
                w[4 + a] |= uint32_t(qmin) << (8 * k);
                w[7 + a] |= uint32_t(qmax) << (8 * k);
            }
        }
    }
    return out;
}

}  // namespace zyg
