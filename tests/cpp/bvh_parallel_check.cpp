// Test harness (CPU): the host BVH builder's parallel form (zyg_amd/csrc/host/bvh_builder.hpp: skeleton + tasks + splice) gives the
// arrays of its plain depth-first recursion, byte for byte: nodes, primitive order, spatial-split duplicates; prints both build times.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../zyg_amd/csrc/host/bvh_builder.hpp"

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? uint32_t(std::atoi(argv[1])) : 300000u;
    const uint32_t threads = argc > 2 ? uint32_t(std::atoi(argv[2])) : 8u;
    std::mt19937                          rng(3);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    std::vector<zyg::BvhReference>        refs(n);
    zyg::AABB                             bounds = zyg::emptyAabb();
    for (uint32_t i = 0; i < n; ++i) {
        // a bumpy sheet of small triangles plus a few long thin ones (what spatial splits are for)
        const float x = uni(rng) * 10.f, z = uni(rng) * 10.f, y = std::sin(x) * std::cos(z) + 0.05f * uni(rng);
        const float s = (0 == i % 97) ? 2.f : 0.02f;
        zyg::BvhReference& r = refs[i];
        r.prim = i;
        r.mn[0] = x, r.mn[1] = y, r.mn[2] = z;
        r.mx[0] = x + s * uni(rng), r.mx[1] = y + 0.02f * uni(rng), r.mx[2] = z + s * uni(rng);
        bounds.b[0] = zyg::min4(bounds.b[0], zyg::vec(r.mn[0], r.mn[1], r.mn[2], 0.f));
        bounds.b[1] = zyg::max4(bounds.b[1], zyg::vec(r.mx[0], r.mx[1], r.mx[2], 0.f));
    }
    std::vector<ZygpuBvhNode> nodes[2];
    std::vector<uint32_t>     prims[2];
    double                    seconds[2];
    for (int k = 0; k < 2; ++k) {
        zyg::BvhSettings settings;
        settings.parallel_from = k ? 1024u : 0u;
        settings.threads = threads;
        zyg::BvhBuilder builder(settings);
        const auto      t0 = std::chrono::steady_clock::now();
        builder.build(refs, bounds, nodes[k], prims[k]);
        seconds[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (nodes[0].size() != nodes[1].size() || prims[0].size() != prims[1].size()) return 3;
    if (0 != std::memcmp(nodes[0].data(), nodes[1].data(), nodes[0].size() * sizeof(ZygpuBvhNode))) return 4;
    if (0 != std::memcmp(prims[0].data(), prims[1].data(), prims[0].size() * sizeof(uint32_t))) return 5;
    std::printf("OK references %u nodes %zu primitives %zu (duplicates %zu) serial %.3f s parallel(%u) %.3f s\n", n, nodes[0].size(), prims[0].size(), prims[0].size() - n,
                seconds[0], threads, seconds[1]);
    return 0;
}
