// Eight-wide compressed BLAS for the relaxed units' persistent mesh walkers (wavefront.hpp wf_shadow_u / wf_mesh_closest; dscene.hpp
// Walk8).  The relaxed units owe the reference the same closest hit, not the same tree: this one is laid out for a wave64 whose lanes
// each walk their own incoherent ray (after Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide
// BVHs", HPG 2017, restated for this device):
//
//   * a node is 80 B = five 16 B loads: the node's corner (3 x f32), per-axis power-of-two scales (3 x u8) and the inner-child mask
//     (u8); the index of its first inner child and of its first triangle (2 x u32); eight meta bytes; the eight children's boxes
//     as 8-bit grid coordinates, axis by axis (6 x 8 B), rounded outwards;
//   * inner children are stored consecutively (child = first + popcount(imask below its slot)), the triangles of a node's leaf
//     children consecutively too (at most 24 per node, at most 7 per leaf): a child needs no reference of its own, and the walk's
//     stack holds *groups*: (first child, hit bits) or (first triangle, hit bits), one entry per visited node instead of up to
//     seven parked children;
//   * children sit in the slot whose octant they lie in relative to the node's centre (slot bit a = the high side of axis a), so a
//     ray visits the hit children in the order slot ^ (7 - octant of its direction), front to back without sorting: the hit bits
//     are assembled already permuted (meta byte: 3 bits count | 5 bits position, position 24 + slot for an inner child, the
//     offset of its first triangle for a leaf) and the walk takes the highest set bit;
//   * triangles are copied in this tree's order (3 x float4 per triangle, the first corner's w = the triangle's index in the
//     reference's order, which is what a hit reports: Intersection.primitive, triangle_tree.zig:46-109).
//
// Built from the reference's BVH2 (the same SAH / spatial splits and leaves: bvh/builder_base.zig), collapsed greedily: while a
// node has fewer than eight children the inner child with the largest surface area is replaced by its two children, as long as
// the node's leaves keep to the 24 triangles its hit mask has bits for.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../../include/zygpu.h"

namespace zyg {

#ifndef ZYG_NODE8_STRIDE
#define ZYG_NODE8_STRIDE 20  // words from one node to the next (20: packed; 32: one 128 B line per node)
#endif
constexpr uint32_t kNode8Words = ZYG_NODE8_STRIDE;  // 80 B of content
constexpr uint32_t kNode8MaxTriangles = 24;  // per node: bits 0..23 of a hit mask
constexpr uint32_t kLeaf8MaxTriangles = 7;   // per leaf: 3 bits of the meta byte

struct Bvh8 {
    std::vector<uint32_t> nodes;      // kNode8Words per node, root first
    std::vector<float>    triangles;  // 12 floats per triangle: a.xyz, bits(reference index), b.xyz, 0, c.xyz, 0
    uint32_t              depth = 0;  // levels of nodes
    bool                  ok = false;
};

// nodes: the BVH2 of one mesh (root at 0); positions of triangle t's corners: tri_verts[3 t .. 3 t + 2] (float4 each, leaf order)
inline Bvh8 build8(const ZygpuBvhNode* nodes, uint32_t num_nodes, const float* tri_verts, uint32_t num_triangles) {
    Bvh8 out;
    if (0 == num_nodes || 0 == num_triangles) return out;
    for (uint32_t n = 0; n < num_nodes; ++n)
        if (0 != nodes[n].count && nodes[n].count > kLeaf8MaxTriangles) return out;  // (a leaf the meta byte cannot count)
    auto area = [&](uint32_t n) {
        const float dx = nodes[n].max[0] - nodes[n].min[0], dy = nodes[n].max[1] - nodes[n].min[1], dz = nodes[n].max[2] - nodes[n].min[2];
        return dx * dy + dx * dz + dy * dz;
    };
    // Which BVH2 nodes become the children of a wide node: the forest that minimises the summed surface area of the wide tree's
    // inner nodes (the expected number of node visits), by dynamic programming over the BVH2 (Ylitie et al. section 4.1 with the
    // leaves kept as they are): cost[n][i - 1] = the cheapest way to stand for subtree n with at most i roots; a single root is
    // a wide node whose children are the best forest of at most eight roots of n's two subtrees.
    struct Plan {
        float   cost[7];
        uint8_t left[8];  // [i - 1], i = 2..7: roots given to the left subtree (0: same as i - 1 roots); [7]: for the eight of a node
    };
    std::vector<Plan> plan(num_nodes);
    for (uint32_t n = num_nodes; n-- > 0;) {  // children lie behind their parent
        Plan& p = plan[n];
        if (0 != nodes[n].count) {
            for (int i = 0; i < 7; ++i) p.cost[i] = 0.f;
            std::memset(p.left, 0, sizeof(p.left));
            continue;
        }
        const uint32_t l = nodes[n].children_or_first, r = l + 1;
        if (r >= num_nodes || l <= n) return Bvh8();
        auto distribute = [&](int j, uint8_t& k_best) {  // the best split of j roots over the two subtrees
            float best = FLT_MAX;
            for (int k = 1; k < j; ++k) {
                const float c = plan[l].cost[std::min(k, 7) - 1] + plan[r].cost[std::min(j - k, 7) - 1];
                if (c < best) best = c, k_best = uint8_t(k);
            }
            return best;
        };
        p.cost[0] = area(n) + distribute(8, p.left[7]);
        p.left[0] = 0;
        for (int i = 2; i <= 7; ++i) {
            uint8_t     k = 0;
            const float c = distribute(i, k);
            if (c < p.cost[i - 2]) p.cost[i - 1] = c, p.left[i - 1] = k;
            else p.cost[i - 1] = p.cost[i - 2], p.left[i - 1] = 0;
        }
    }
    // the roots that stand for subtree n when it may use at most i of them
    std::vector<uint32_t> gathered;
    auto gather = [&](auto&& self, uint32_t n, int i) -> void {
        if (0 != nodes[n].count) {
            gathered.push_back(n);
            return;
        }
        while (i > 1 && 0 == plan[n].left[i - 1]) --i;  // (the same forest as with one root fewer)
        if (1 == i) {
            gathered.push_back(n);
            return;
        }
        const int k = plan[n].left[i - 1];
        self(self, nodes[n].children_or_first, std::min(k, 7));
        self(self, nodes[n].children_or_first + 1, std::min(i - k, 7));
    };
    struct Todo {
        uint32_t bvh2, node8, level;
    };
    std::vector<Todo> todo;
    out.nodes.assign(kNode8Words, 0u);
    out.triangles.reserve(size_t(num_triangles) * 12 + 12 * 64);
    if (0 != nodes[0].count) todo.push_back({0xFFFFFFFFu, 0u, 1u});  // the whole mesh is one leaf: a root with that one child
    else todo.push_back({0u, 0u, 1u});
    size_t at = 0;
    while (at < todo.size()) {  // breadth first: a node's inner children are consecutive, siblings' subtrees follow level by level
        const Todo t = todo[at++];
        out.depth = std::max(out.depth, t.level);
        uint32_t kids[8];
        uint32_t n = 0;
        auto leafTriangles = [&]() {
            uint32_t sum = 0;
            for (uint32_t k = 0; k < n; ++k) sum += nodes[kids[k]].count;
            return sum;
        };
        if (0xFFFFFFFFu == t.bvh2) {
            kids[n++] = 0;
        } else {
            // the planned eight (or fewer); with fewer roots while the leaves among them hold more triangles than a hit mask has bits
            const uint32_t l = nodes[t.bvh2].children_or_first;
            for (int roots = 8; roots >= 2; --roots) {
                gathered.clear();
                if (8 == roots) {
                    const int k = plan[t.bvh2].left[7];
                    gather(gather, l, std::min(k, 7));
                    gather(gather, l + 1, std::min(8 - k, 7));
                } else {
                    gather(gather, l, std::min(roots - 1, 7) > 1 ? (roots + 1) / 2 : 1);
                    gather(gather, l + 1, std::max(roots / 2, 1));
                }
                if (gathered.size() > 8) return Bvh8();
                n = uint32_t(gathered.size());
                std::copy(gathered.begin(), gathered.end(), kids);
                if (leafTriangles() <= kNode8MaxTriangles) break;
            }
        }
        if (leafTriangles() > kNode8MaxTriangles) return Bvh8();  // (two leaves of the BVH2's root already too many: cannot happen with 7 + 7)
        // the node's box and grid
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t k = 0; k < n; ++k)
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], nodes[kids[k]].min[a]);
                hi[a] = std::max(hi[a], nodes[kids[k]].max[a]);
            }
        float    scale[3];
        uint32_t exps = 0;
        for (int a = 0; a < 3; ++a) {
            const float extent = hi[a] - lo[a];
            int         e = 1;  // biased exponent: the smallest power of two with 254 steps covering the extent
            while (e < 254 && std::ldexp(254.f, e - 127) < extent) ++e;
            scale[a] = std::ldexp(1.f, e - 127);
            exps |= uint32_t(e) << (8 * a);
        }
        // slots by octant: the pair (child, slot) with the largest sum of signed centre offsets first
        int   slot_of[8], child_in[8];
        for (int k = 0; k < 8; ++k) slot_of[k] = child_in[k] = -1;
        float offs[8][3];
        for (uint32_t k = 0; k < n; ++k)
            for (int a = 0; a < 3; ++a) offs[k][a] = 0.5f * (nodes[kids[k]].min[a] + nodes[kids[k]].max[a]) - 0.5f * (lo[a] + hi[a]);
        for (uint32_t placed = 0; placed < n; ++placed) {
            int   bk = -1, bs = -1;
            float best = -FLT_MAX;
            for (uint32_t k = 0; k < n; ++k) {
                if (slot_of[k] >= 0) continue;
                for (int s = 0; s < 8; ++s) {
                    if (child_in[s] >= 0) continue;
                    float gain = 0.f;
                    for (int a = 0; a < 3; ++a) gain += ((s >> a) & 1) ? offs[k][a] : -offs[k][a];
                    if (gain > best) best = gain, bk = int(k), bs = s;
                }
            }
            slot_of[bk] = bs;
            child_in[bs] = bk;
        }
        uint32_t* w = &out.nodes[size_t(t.node8) * kNode8Words];
        for (int a = 0; a < 3; ++a) std::memcpy(&w[a], &lo[a], 4);
        uint32_t imask = 0;
        const uint32_t child_base = uint32_t(out.nodes.size() / kNode8Words), tri_base = uint32_t(out.triangles.size() / 12);
        uint32_t tri_offset = 0;
        uint8_t  meta[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[6][8];
        std::memset(q, 0, sizeof(q));
        for (int s = 0; s < 8; ++s) {
            if (child_in[s] < 0) continue;
            const ZygpuBvhNode& c = nodes[kids[child_in[s]]];
            for (int a = 0; a < 3; ++a) {
                int qmin = int(std::floor((c.min[a] - lo[a]) / scale[a])), qmax = int(std::ceil((c.max[a] - lo[a]) / scale[a]));
                qmin = std::max(0, std::min(qmin, 255));
                qmax = std::max(0, std::min(qmax, 255));
                while (qmin > 0 && std::fmaf(float(qmin), scale[a], lo[a]) > c.min[a]) --qmin;
                while (qmax < 255 && std::fmaf(float(qmax), scale[a], lo[a]) < c.max[a]) ++qmax;
                if (std::fmaf(float(qmin), scale[a], lo[a]) > c.min[a] || std::fmaf(float(qmax), scale[a], lo[a]) < c.max[a]) return Bvh8();
                q[a][s] = uint8_t(qmin);
                q[3 + a][s] = uint8_t(qmax);
            }
            if (0 == c.count) {
                imask |= 1u << s;
                meta[s] = uint8_t((1u << 5) | (24u + uint32_t(s)));
            } else {
                meta[s] = uint8_t((c.count << 5) | tri_offset);
                for (uint32_t i = 0; i < c.count; ++i) {
                    const uint32_t ref = c.children_or_first + i;
                    const float*   v = tri_verts + size_t(ref) * 12;
                    float          rec[12] = {v[0], v[1], v[2], 0.f, v[4], v[5], v[6], 0.f, v[8], v[9], v[10], 0.f};
                    std::memcpy(&rec[3], &ref, 4);
                    out.triangles.insert(out.triangles.end(), rec, rec + 12);
                }
                tri_offset += c.count;
            }
        }
        // inner children in slot order: consecutive nodes from child_base on
        for (int s = 0; s < 8; ++s) {
            if (0 == (imask & (1u << s))) continue;
            todo.push_back({kids[child_in[s]], uint32_t(out.nodes.size() / kNode8Words), t.level + 1});
            out.nodes.resize(out.nodes.size() + kNode8Words, 0u);
        }
        w = &out.nodes[size_t(t.node8) * kNode8Words];  // (the vector may have moved)
        w[3] = exps | (imask << 24);
        w[4] = child_base;
        w[5] = tri_base;
        std::memcpy(&w[6], meta, 8);
        for (int p = 0; p < 6; ++p) std::memcpy(&w[8 + 2 * p], q[p], 8);
    }
    if (out.nodes.size() / kNode8Words >= (1u << 24) || out.triangles.size() / 12 >= (1u << 28)) return Bvh8();
    out.ok = true;
    return out;
}

// Host-side reference walk of the packed nodes (tests/cpp/bvh8_check.cpp): every triangle a ray reaches through the groups
template <typename Visit>
inline void walk8(const Bvh8& b, const float o[3], const float d[3], float min_t, float max_t, Visit&& visit) {
    struct Group {
        uint32_t base, bits;
    };
    std::vector<Group> stack;
    float              idir[3];
    uint32_t           octant = 0;
    for (int a = 0; a < 3; ++a) {
        const float da = std::fabs(d[a]) > 1e-30f ? d[a] : std::copysign(1e-30f, d[a]);
        idir[a] = 1.f / da;
        if (d[a] < 0.f) octant |= 1u << a;
    }
    const uint32_t oct_inv = 7u ^ octant;
    Group          ng{0u, (1u << (24u + oct_inv)) | 1u};  // the root: the one inner child, in slot 0, of a group whose first child is node 0
    for (;;) {
        Group tg{0, 0};
        if (ng.bits & 0xFF000000u) {
            const uint32_t bit = 31u - uint32_t(__builtin_clz(ng.bits));
            ng.bits &= ~(1u << bit);
            const uint32_t slot = (bit - 24u) ^ oct_inv;
            const uint32_t node = ng.base + uint32_t(__builtin_popcount(ng.bits & 0xFFu & ((1u << slot) - 1u)));
            if (ng.bits & 0xFF000000u) stack.push_back(ng);
            const uint32_t* w = &b.nodes[size_t(node) * kNode8Words];
            float           lo[3];
            std::memcpy(lo, w, 12);
            const uint32_t imask = w[3] >> 24;
            uint32_t       hits = 0;
            const uint8_t* meta = reinterpret_cast<const uint8_t*>(&w[6]);
            const uint8_t* q = reinterpret_cast<const uint8_t*>(&w[8]);
            for (int s = 0; s < 8; ++s) {
                if (0 == meta[s]) continue;
                float tmin = min_t, tmax = max_t;
                for (int a = 0; a < 3; ++a) {
                    const float sc = std::ldexp(1.f, int((w[3] >> (8 * a)) & 255u) - 127);
                    const float blo = lo[a] + float(q[a * 8 + s]) * sc, bhi = lo[a] + float(q[(3 + a) * 8 + s]) * sc;
                    const float t0 = (blo - o[a]) * idir[a], t1 = (bhi - o[a]) * idir[a];
                    tmin = std::max(tmin, std::min(t0, t1));
                    tmax = std::min(tmax, std::max(t0, t1));
                }
                if (tmin > tmax) continue;
                const uint32_t count = meta[s] >> 5, pos = meta[s] & 31u;
                const bool     inner = pos >= 24u;
                hits |= ((1u << count) - 1u) << (inner ? (24u + ((pos - 24u) ^ oct_inv)) : pos);
            }
            ng = Group{w[4], (hits & 0xFF000000u) | imask};
            tg = Group{w[5], hits & 0x00FFFFFFu};
        } else {
            tg = ng;
            ng = Group{0, 0};
        }
        while (tg.bits) {
            const uint32_t bit = uint32_t(__builtin_ctz(tg.bits));
            tg.bits &= tg.bits - 1u;
            if (visit(tg.base + bit)) return;
        }
        if (0 == (ng.bits & 0xFF000000u)) {
            if (stack.empty()) return;
            ng = stack.back();
            stack.pop_back();
        }
    }
}

}  // namespace zyg
