// Test harness (CPU): the eight-wide compressed BLAS (zyg_amd/csrc/hip/bvh8.hpp) built from a BVH2 read from a file of ZygpuBvhNode
// records and the mesh's triangle corners (12 floats per triangle, leaf order).  Checks: every reference triangle is stored exactly
// once with its corners; inner children are consecutive and every node has one parent; a leaf child's decoded (quantised) box
// contains its triangles; and for random rays the closest hit over the triangles walk8() reaches equals the brute-force closest hit
// over all triangles (same Moeller-Trumbore arithmetic, so bit for bit, ties by distance only).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../zyg_amd/csrc/hip/bvh8.hpp"

static bool hitTriangle(const float* o, const float* d, float min_t, float max_t, const float* v, float& t_out) {
    const float e1[3] = {v[4] - v[0], v[5] - v[1], v[6] - v[2]}, e2[3] = {v[8] - v[0], v[9] - v[1], v[10] - v[2]};
    const float tv[3] = {o[0] - v[0], o[1] - v[1], o[2] - v[2]};
    const float pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    const float inv = 1.f / det;
    const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv, w = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
    const float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    if (u >= 0.f && 1.f >= u && w >= 0.f && 1.f >= u + w && t >= min_t && max_t >= t) {
        t_out = t;
        return true;
    }
    return false;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::vector<ZygpuBvhNode> nodes;
    {
        FILE* f = std::fopen(argv[1], "rb");
        if (!f) return 2;
        ZygpuBvhNode n;
        while (1 == std::fread(&n, sizeof(n), 1, f)) nodes.push_back(n);
        std::fclose(f);
    }
    std::vector<float> verts;
    {
        FILE* f = std::fopen(argv[2], "rb");
        if (!f) return 2;
        float rec[12];
        while (1 == std::fread(rec, sizeof(rec), 1, f)) verts.insert(verts.end(), rec, rec + 12);
        std::fclose(f);
    }
    const uint32_t  num_triangles = uint32_t(verts.size() / 12);
    const zyg::Bvh8 b = zyg::build8(nodes.data(), uint32_t(nodes.size()), verts.data(), num_triangles);
    if (!b.ok) return 3;
    const size_t num_nodes = b.nodes.size() / zyg::kNode8Words;
    if (b.triangles.size() != size_t(num_triangles) * 12) return 4;
    std::vector<uint8_t> seen(num_triangles, 0), parent(num_nodes, 0);
    parent[0] = 1;
    for (size_t t = 0; t < num_triangles; ++t) {
        uint32_t ref;
        std::memcpy(&ref, &b.triangles[t * 12 + 3], 4);
        if (ref >= num_triangles || seen[ref]) return 5;
        seen[ref] = 1;
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a)
                if (b.triangles[t * 12 + 4 * k + a] != verts[size_t(ref) * 12 + 4 * k + a]) return 6;
    }
    double children = 0.0;
    for (size_t n = 0; n < num_nodes; ++n) {
        const uint32_t* w = &b.nodes[n * zyg::kNode8Words];
        float           lo[3];
        std::memcpy(lo, w, 12);
        const uint32_t imask = w[3] >> 24;
        const uint8_t* meta = reinterpret_cast<const uint8_t*>(&w[6]);
        const uint8_t* q = reinterpret_cast<const uint8_t*>(&w[8]);
        uint32_t       inner = 0, tris = 0;
        for (int s = 0; s < 8; ++s) {
            if (0 == meta[s]) {
                if (imask & (1u << s)) return 7;
                continue;
            }
            children += 1.0;
            const uint32_t count = meta[s] >> 5, pos = meta[s] & 31u;
            if (pos >= 24u) {
                if (pos != 24u + uint32_t(s) || 1 != count || 0 == (imask & (1u << s))) return 8;
                const uint32_t child = w[4] + inner++;
                if (child >= num_nodes || parent[child]) return 9;
                parent[child] = 1;
                continue;
            }
            if (imask & (1u << s)) return 10;
            if (pos != tris || count > zyg::kLeaf8MaxTriangles) return 11;  // (leaves in slot order, offsets running)
            for (uint32_t i = 0; i < count; ++i) {
                const float* v = &b.triangles[size_t(w[5] + pos + i) * 12];
                for (int a = 0; a < 3; ++a) {
                    const float sc = std::ldexp(1.f, int((w[3] >> (8 * a)) & 255u) - 127);
                    const float blo = std::fmaf(float(q[a * 8 + s]), sc, lo[a]), bhi = std::fmaf(float(q[(3 + a) * 8 + s]), sc, lo[a]);
                    for (int k = 0; k < 3; ++k)
                        if (v[4 * k + a] < blo || v[4 * k + a] > bhi) return 12;
                }
            }
            tris += count;
        }
        if (tris > zyg::kNode8MaxTriangles) return 13;
    }
    for (uint8_t p : parent)
        if (!p) return 14;
    // rays: from outside through the mesh's box, and from points near the surface in random directions
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < verts.size(); i += 4)
        for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], verts[i + a]), hi[a] = std::max(hi[a], verts[i + a]);
    std::mt19937                          rng(5);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    const int                             num_rays = argc > 3 ? std::atoi(argv[3]) : 300;
    int                                   hits = 0;
    double                                visited = 0.0;
    for (int r = 0; r < num_rays; ++r) {
        float o[3], d[3];
        if (r % 3) {
            for (int a = 0; a < 3; ++a) o[a] = lo[a] + (hi[a] - lo[a]) * (uni(rng) * 1.6f - 0.3f);
        } else {
            const float* v = &verts[size_t(rng() % num_triangles) * 12];
            for (int a = 0; a < 3; ++a) o[a] = v[a] + (uni(rng) - 0.5f) * 0.02f * (hi[a] - lo[a]);
        }
        float len = 0.f;
        for (int a = 0; a < 3; ++a) d[a] = uni(rng) * 2.f - 1.f, len += d[a] * d[a];
        if (0 == r % 7) d[r % 3] = 0.f;  // axis-parallel components
        if (0 == r % 11) d[(r + 1) % 3] = -0.f;
        len = std::sqrt(std::max(len, 1e-12f));
        for (int a = 0; a < 3; ++a) d[a] /= len;
        float    best = FLT_MAX, t;
        uint32_t best_ref = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < num_triangles; ++i)
            if (hitTriangle(o, d, 0.f, best, &verts[size_t(i) * 12], t) && t < best) best = t, best_ref = i;
        float    found = FLT_MAX;
        uint32_t found_ref = 0xFFFFFFFFu;
        float    max_t = FLT_MAX;
        zyg::walk8(b, o, d, 0.f, max_t, [&](uint32_t tri) {
            visited += 1.0;
            const float* v = &b.triangles[size_t(tri) * 12];
            if (hitTriangle(o, d, 0.f, found, v, t) && t < found) {
                found = t;
                std::memcpy(&found_ref, &v[3], 4);
            }
            return false;
        });
        if (found != best) {
            std::printf("ray %d: walk8 %g (%u) brute force %g (%u)\n", r, found, found_ref, best, best_ref);
            return 15;
        }
        hits += FLT_MAX != best;
    }
    if (hits < num_rays / 10) return 16;
    std::printf("OK nodes2 %zu nodes8 %zu triangles %u children/node %.2f rays %d hits %d triangle tests/ray %.1f depth8 %u\n", nodes.size(), num_nodes, num_triangles,
                children / double(num_nodes), num_rays, hits, visited / num_rays, b.depth);
    return 0;
}
