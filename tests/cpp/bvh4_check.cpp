// Test harness (CPU): the wide re-layout of a BLAS (zyg_amd/csrc/hip/bvh4.hpp) on a BVH2 read from a file of ZygpuBvhNode records.
// Checks: collapse4 keeps every leaf (same triangle ranges, each exactly once) under boxes that contain it; compress4's decoded
// child boxes contain the exact ones, with the arithmetic the device uses (corner + q * 2^(e - 127), one fused multiply-add).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../zyg_amd/csrc/hip/bvh4.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<ZygpuBvhNode> nodes;
    ZygpuBvhNode              n;
    while (1 == std::fread(&n, sizeof(n), 1, f)) nodes.push_back(n);
    std::fclose(f);
    std::map<uint32_t, uint32_t> leaves2;  // first triangle -> count
    for (const ZygpuBvhNode& b : nodes)
        if (b.count) leaves2[b.children_or_first] = b.count;
    const std::vector<zyg::Bvh4Slot> wide = zyg::collapse4(nodes.data(), uint32_t(nodes.size()));
    if (wide.empty() || wide.size() % 4) return 3;
    std::map<uint32_t, uint32_t> leaves4;
    std::vector<uint32_t>        visited(wide.size() / 4, 0);
    visited[0] = 1;
    for (size_t i = 0; i < wide.size(); ++i) {
        const zyg::Bvh4Slot& s = wide[i];
        if (0xFFFFFFFFu == s.ref) continue;
        if (s.count) {
            if (leaves4.count(s.ref)) return 4;  // a leaf twice
            leaves4[s.ref] = s.count;
        } else {
            if (s.ref >= visited.size() || visited[s.ref]) return 5;  // a node with two parents, or out of range
            visited[s.ref] = 1;
            // the child node's slots lie inside this slot's box
            for (int k = 0; k < 4; ++k) {
                const zyg::Bvh4Slot& c = wide[size_t(s.ref) * 4 + k];
                if (0xFFFFFFFFu == c.ref) continue;
                for (int a = 0; a < 3; ++a)
                    if (c.min[a] < s.min[a] || c.max[a] > s.max[a]) return 6;
            }
        }
    }
    if (leaves4 != leaves2) return 7;
    for (uint32_t v : visited)
        if (!v) return 8;
    const std::vector<uint32_t> packed = zyg::compress4(wide);
    if (packed.size() != wide.size() / 4 * 16) return 9;
    double slack = 0.0;
    for (size_t node = 0; node < wide.size() / 4; ++node) {
        const uint32_t* w = &packed[16 * node];
        float           o[3], sc[3];
        for (int a = 0; a < 3; ++a) {
            std::memcpy(&o[a], &w[a], 4);
            const uint32_t bits = ((w[3] >> (8 * a)) & 255u) << 23;
            std::memcpy(&sc[a], &bits, 4);
        }
        for (int k = 0; k < 4; ++k) {
            const zyg::Bvh4Slot& s = wide[node * 4 + k];
            if (w[10 + k] != s.ref) return 10;
            if (0xFFFFFFFFu == s.ref) continue;
            if (((w[14] >> (8 * k)) & 255u) != s.count) return 11;
            for (int a = 0; a < 3; ++a) {
                const float lo = std::fmaf(float((w[4 + a] >> (8 * k)) & 255u), sc[a], o[a]);
                const float hi = std::fmaf(float((w[7 + a] >> (8 * k)) & 255u), sc[a], o[a]);
                if (lo > s.min[a] || hi < s.max[a]) return 12;
                slack += double(s.min[a] - lo) + double(hi - s.max[a]);
            }
        }
    }
    // depth4 against the BVH2's own depth: a BVH4 level swallows one or two BVH2 levels
    std::vector<uint32_t> level2(nodes.size(), 0);
    uint32_t              depth2 = 0;
    level2[0] = 1;
    for (size_t i = 0; i < nodes.size(); ++i) {  // children lie behind their parent in the array
        if (0 == level2[i]) return 13;
        if (nodes[i].count) continue;
        depth2 = std::max(depth2, level2[i]);
        level2[nodes[i].children_or_first] = level2[nodes[i].children_or_first + 1] = level2[i] + 1;
    }
    const uint32_t depth = zyg::depth4(wide);
    if (depth > depth2 || 2 * depth < depth2) return 14;
    std::printf("OK nodes2 %zu nodes4 %zu leaves %zu mean slack %.3g depth2 %u depth4 %u\n", nodes.size(), wide.size() / 4, leaves4.size(),
                slack / double(6 * wide.size()), depth2, depth);
    return 0;
}
