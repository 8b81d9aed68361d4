// Device samplers: PCG32 (reference src/base/random/generator.zig:13-47) and the Owen-scrambled 5-D Sobol
// sampler with per-refill seed padding (src/core/sampler/sobol.zig:18-192).  Integer work on the VALU;
// the Sobol direction words are compile-time constants of the instruction stream (no table loads).
#pragma once

#include "dmath.hpp"

namespace zd {

struct Pcg {
    uint64_t state, inc;

    ZD uint32_t randomUint() {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        const uint32_t xrs = uint32_t(((old >> 18) ^ old) >> 27);
        const uint32_t rot = uint32_t(old >> 59);
        return (xrs >> rot) | (xrs << ((0u - rot) & 31u));
    }
    ZD void start(uint64_t st, uint64_t sequence) {
        state = 0;
        inc = (sequence << 1) | 1;
        randomUint();
        state += st;
        randomUint();
    }
    ZD float randomFloat() {
        uint32_t bits = randomUint();
        bits &= 0x007FFFFFu;
        bits |= 0x3F800000u;
        return __uint_as_float(bits) - 1.f;
    }
};

namespace sobol {

ZD uint32_t hash(uint32_t i) {  // sobol.zig:107-124
    uint32_t x = i ^ (i >> 16);
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
ZD uint32_t hashCombine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
ZD uint32_t laineKarras(uint32_t i, uint32_t seed) {  // sobol.zig:142-174
    uint32_t x = i ^ (i * 0x3d20adeau);
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return x;
}
ZD uint32_t nus(uint32_t x, uint32_t seed) { return __brev(laineKarras(__brev(x), seed)); }  // sobol.zig:136-140

// Direction numbers of Sobol dimensions 2..5 (sobol.zig:194-245) by the Joe-Kuo recurrence
// (s, a, m): (1,0,{1}), (2,1,{1,3}), (3,1,{1,3,1}), (3,2,{1,1,1}); dimension 1 is bit reversal.
struct Directions {
    uint32_t v[4][32];
};
constexpr Directions makeDirections() {
    Directions     D{};
    const uint32_t S_[4] = {1, 2, 3, 3}, A_[4] = {0, 1, 1, 2};
    const uint32_t M_[4][3] = {{1, 0, 0}, {1, 3, 0}, {1, 3, 1}, {1, 1, 1}};
    for (int d = 0; d < 4; ++d) {
        const uint32_t s = S_[d], a = A_[d];
        for (uint32_t i = 0; i < s; ++i) D.v[d][i] = M_[d][i] << (31 - i);
        for (uint32_t i = s; i < 32; ++i) {
            uint32_t x = D.v[d][i - s] ^ (D.v[d][i - s] >> s);
            for (uint32_t k = 1; k < s; ++k) x ^= (((a >> (s - 1 - k)) & 1u) * D.v[d][i - k]);
            D.v[d][i] = x;
        }
    }
    return D;
}
constexpr Directions kDirections = makeDirections();
static_assert(kDirections.v[0][1] == 0xc0000000u && kDirections.v[1][4] == 0xe8000000u && kDirections.v[2][3] == 0x50000000u &&
                  kDirections.v[3][3] == 0xb0000000u && kDirections.v[3][31] == 0xb00ddb9du,
              "Sobol direction numbers");

// The 32-step XOR of sobol.zig:176-192 folded per index byte: c_sobol_lut[k * 256 + b] holds, for dimensions 2..5, the XOR of
// the direction words 8k..8k+7 selected by the bits of b (16 KB, L1-resident; filled once by sobol_lut_init).  Four 16 B
// loads and twelve XORs replace ~290 VALU instructions per refill; XOR is exact, so the result is the same word for word.
// The __constant__ blocks (this one, c_view / c_scene in wavefront.hpp): one copy per translation unit (kernel_table.hpp), so each
// carries its unit in its name; external linkage with protected visibility, so that the runtime finds the symbol
// (hipMemcpyToSymbol) and the kernels address it pc-relative: as `static` variables every read went through a
// global-offset-table entry first, a dependent scalar load (in the small-mesh loop: one per item).
#define ZYG_BLOCK_CAT2(a, b) a##b
#define ZYG_BLOCK_CAT(a, b) ZYG_BLOCK_CAT2(a, b)
#ifdef ZYG_TU
#define ZYG_BLOCK_NAME(name) ZYG_BLOCK_CAT(name##_u, ZYG_TU)
#else
#define ZYG_BLOCK_NAME(name) name##_main
#endif
#define ZYG_CONSTANT_BLOCK __constant__ __attribute__((visibility("protected")))
#define c_sobol_lut ZYG_BLOCK_NAME(c_sobol_lut)
ZYG_CONSTANT_BLOCK const uint4* c_sobol_lut;  // per translation unit, like c_view / c_scene (wavefront.hpp)

static __global__ void sobol_lut_init(uint4* lut) {  // <<<4, 256>>>
    const uint32_t k = blockIdx.x, b = threadIdx.x;
    uint32_t       x[4] = {0, 0, 0, 0};
    for (uint32_t j = 0; j < 8; ++j) {
        if ((b >> j) & 1u) {
            for (uint32_t d = 0; d < 4; ++d) x[d] ^= kDirections.v[d][8 * k + j];
        }
    }
    lut[k * 256 + b] = make_uint4(x[0], x[1], x[2], x[3]);
}

ZD void sobol5(uint32_t index, uint32_t out[5]) {  // sobol.zig:176-192
    out[0] = __brev(index);
#ifdef ZYG_SOBOL_UNROLLED
    uint32_t x1 = 0, x2 = 0, x3 = 0, x4 = 0;
#pragma unroll
    for (int bit = 0; bit < 32; ++bit) {
        const uint32_t mask = 0u - ((index >> bit) & 1u);
        x1 ^= mask & kDirections.v[0][bit];
        x2 ^= mask & kDirections.v[1][bit];
        x3 ^= mask & kDirections.v[2][bit];
        x4 ^= mask & kDirections.v[3][bit];
    }
    out[1] = x1;
    out[2] = x2;
    out[3] = x3;
    out[4] = x4;
#else
    const uint4* lut = c_sobol_lut;
    const uint4  a = lut[index & 255u], b = lut[256u + ((index >> 8) & 255u)], c = lut[512u + ((index >> 16) & 255u)],
                d = lut[768u + (index >> 24)];
    out[1] = a.x ^ b.x ^ c.x ^ d.x;
    out[2] = a.y ^ b.y ^ c.y ^ d.y;
    out[3] = a.z ^ b.z ^ c.z ^ d.z;
    out[4] = a.w ^ b.w ^ c.w ^ d.w;
#endif
}

}  // namespace sobol

// One path's sampler pair (worker.zig:201-207: bounces 0-2 use samplers[0], later ones the PCG stream).
struct Sampler {
    // Sobol (sobol.zig:8-16); start_seed is not kept: a path is one sample, incrementSample never runs
    float    buffer[5];
    uint32_t sample, dimension, run_seed;
    Pcg      rng;
    bool     sobol_kind;  // view sampler is Sobol (take_loader.zig:232-244)
    bool     use_primary; // current bounce < 3

    ZD bool isSobol() const { return use_primary && sobol_kind; }

    ZD void startPixel(uint32_t smp, uint32_t seed) {
        sample = smp;
        dimension = 5;
        run_seed = sobol::hash(seed);
    }
    ZD void incrementPadding() {
        if (isSobol()) dimension = 5;
    }
    ZD void incrementSeed() {  // sobol.zig:36-60
        const float    Sc = 1.f / 4294967296.f;
        const uint32_t s = run_seed;
        const uint32_t i = sobol::nus(sample, s);
        uint32_t       sob[5];
        sobol::sobol5(i, sob);
#pragma unroll
        for (uint32_t d = 0; d < 5; ++d) buffer[d] = float(sobol::nus(sob[d], sobol::hashCombine(s, d))) * Sc;
        run_seed = sobol::hash(s + 1);
        dimension = 0;
    }
    ZD float buf(uint32_t d) const {  // register-resident select instead of a dynamically indexed array (dmath.hpp pick)
        const float b0 = buffer[0], b1 = buffer[1], b2 = buffer[2], b3 = buffer[3], b4 = buffer[4];
        return pick(d == 0, b0, pick(d == 1, b1, pick(d == 2, b2, pick(d == 3, b3, b4))));
    }
    ZD float sample1D() {
        if (!isSobol()) return rng.randomFloat();
        if (dimension >= 5) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 1;
        return buf(d);
    }
    ZD void sample2D(float& a, float& b) {
        if (!isSobol()) {
            a = rng.randomFloat();
            b = rng.randomFloat();
            return;
        }
        if (dimension >= 4) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 2;
        a = buf(d);
        b = buf(d + 1);
    }
    ZD void sample3D(float& a, float& b, float& c) {
        if (!isSobol()) {
            a = rng.randomFloat();
            b = rng.randomFloat();
            c = rng.randomFloat();
            return;
        }
        if (dimension >= 3) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 3;
        a = buf(d);
        b = buf(d + 1);
        c = buf(d + 2);
    }
    ZD void sample4D(float& a, float& b, float& c, float& e) {
        if (!isSobol()) {
            a = rng.randomFloat();
            b = rng.randomFloat();
            c = rng.randomFloat();
            e = rng.randomFloat();
            return;
        }
        if (dimension >= 2) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 4;
        a = buf(d);
        b = buf(d + 1);
        c = buf(d + 2);
        e = buf(d + 3);
    }
};

}  // namespace zd
