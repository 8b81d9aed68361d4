// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp).
//
// PCG32 (reference src/base/random/generator.zig:1-47), Owen-scrambled Sobol with per-refill
// seed padding (src/core/sampler/sobol.zig:8-245) and the Sampler union (src/core/sampler/sampler.zig:16-72).
#pragma once

#include <cstdint>
#include <cstring>

namespace orc {

struct Pcg {  // generator.zig:1-47
    uint64_t state, inc;

    void start(uint64_t st, uint64_t sequence) {
        state = 0;
        inc = (sequence << 1) | 1;
        randomUint();
        state += st;
        randomUint();
    }
    uint32_t randomUint() {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        const uint32_t xrs = uint32_t(((old >> 18) ^ old) >> 27);
        const uint32_t rot = uint32_t(old >> 59);
        return (xrs >> rot) | (xrs << ((0u - rot) & 31u));
    }
    float randomFloat() {
        uint32_t bits = randomUint();
        bits &= 0x007FFFFFu;
        bits |= 0x3F800000u;
        float f;
        std::memcpy(&f, &bits, 4);
        return f - 1.f;
    }
};

namespace sobol {

inline uint32_t hash(uint32_t i) {  // sobol.zig:107-124
    uint32_t x = i ^ (i >> 16);
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t hashCombine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }  // :126-134
inline uint32_t bitReverse(uint32_t x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    return x;
}
inline uint32_t laineKarras(uint32_t i, uint32_t seed) {  // :142-174
    uint32_t x = i ^ (i * 0x3d20adeau);
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return x;
}
inline uint32_t nus(uint32_t x, uint32_t seed) { return bitReverse(laineKarras(bitReverse(x), seed)); }  // :136-140

// Direction numbers (sobol.zig:194-245) are not literals here: dimension d uses the Joe-Kuo primitive polynomial /
// initial m_i of the first five Sobol dimensions, generated once by the standard recurrence (see directions()).
const uint32_t* directions();  // [5][32]

inline void sobol5(uint32_t index, uint32_t out[5]) {  // :176-192
    const uint32_t* D = directions();
    uint32_t        x[5] = {0, 0, 0, 0, 0};
    for (uint32_t bit = 0; bit < 32; ++bit) {
        const uint32_t mask = (index >> bit) & 1u;
        for (int d = 0; d < 5; ++d) x[d] ^= mask * D[d * 32 + bit];
    }
    for (int d = 0; d < 5; ++d) out[d] = x[d];
}

}  // namespace sobol

struct Sobol {  // sobol.zig:8-105
    float    buffer[5];
    uint32_t sample, dimension, start_seed, run_seed;

    void startPixel(uint32_t smp, uint32_t seed) {
        sample = smp;
        dimension = 5;
        const uint32_t hashed = sobol::hash(seed);
        start_seed = hashed;
        run_seed = hashed;
    }
    void incrementSample() {
        sample += 1;
        dimension = 5;
        run_seed = start_seed;
    }
    void incrementPadding() { dimension = 5; }
    void incrementSeed() {
        const float    Sc = 1.f / 4294967296.f;
        const uint32_t s = run_seed;
        const uint32_t i = sobol::nus(sample, s);
        uint32_t       sob[5];
        sobol::sobol5(i, sob);
        for (uint32_t d = 0; d < 4; ++d) buffer[d] = float(sobol::nus(sob[d], sobol::hashCombine(s, d))) * Sc;
        buffer[4] = float(sobol::nus(sob[4], sobol::hashCombine(s, 4))) * Sc;
        run_seed = sobol::hash(s + 1);
        dimension = 0;
    }
    float sample1D() {
        if (dimension >= 5) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 1;
        return buffer[d];
    }
    void sample2D(float out[2]) {
        if (dimension >= 4) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 2;
        out[0] = buffer[d];
        out[1] = buffer[d + 1];
    }
    void sample3D(float out[3]) {
        if (dimension >= 3) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 3;
        out[0] = buffer[d];
        out[1] = buffer[d + 1];
        out[2] = buffer[d + 2];
    }
    void sample4D(float out[4]) {
        if (dimension >= 2) incrementSeed();
        const uint32_t d = dimension;
        dimension = d + 4;
        for (int k = 0; k < 4; ++k) out[k] = buffer[d + k];
    }
};

struct Sampler {  // sampler.zig:16-72: Random draws from the worker's PCG, ignores padding / sample increments
    bool  is_sobol;
    Sobol sobol;
    Pcg*  rng;

    void startPixel(uint32_t smp, uint32_t seed) {
        if (is_sobol) sobol.startPixel(smp, seed);
    }
    void incrementSample() {
        if (is_sobol) sobol.incrementSample();
    }
    void incrementPadding() {
        if (is_sobol) sobol.incrementPadding();
    }
    float sample1D() { return is_sobol ? sobol.sample1D() : rng->randomFloat(); }
    void  sample2D(float o[2]) {
        if (is_sobol) {
            sobol.sample2D(o);
        } else {
            o[0] = rng->randomFloat();
            o[1] = rng->randomFloat();
        }
    }
    void sample3D(float o[3]) {
        if (is_sobol) {
            sobol.sample3D(o);
        } else {
            for (int k = 0; k < 3; ++k) o[k] = rng->randomFloat();
        }
    }
    void sample4D(float o[4]) {
        if (is_sobol) {
            sobol.sample4D(o);
        } else {
            for (int k = 0; k < 4; ++k) o[k] = rng->randomFloat();
        }
    }
};

}  // namespace orc
