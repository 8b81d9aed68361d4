// The Sky entity's bake (core/sky/sky.zig:172-269, sky_model.zig:23-137): the sun's radiance over its disc and the canopy image,
// computed from the Prague sky model (Wilkie et al.; the reference links the authors' C, thirdparty/arpraguesky).  The model's
// evaluator here is this repository's own (prague_sky.hpp: dataset reader, sky radiance, solar radiance, transmittance), checked
// against the reference's library compiled as test infrastructure (oracle/_ref/libarpraguesky.so, tests/test_prague_sky.py); the
// model's dataset (sky/SkyModelDataset.dat.gz, not shipped with the reference) comes through the engine's mounts.  Without it the
// engine's procedural stand-in remains (Engine::compileSky), flagged in the log as before.
//
//   Model.init / evaluateSky / evaluateSun     sky_model.zig:32-137 (12 spectral bins over 380..740 nm, 24 golden-ratio wavelength
//                                              samples per bin, XYZ -> ACES AP1)
//   DiscreteSpectralPowerDistribution(12, ..)  base/spectrum/discrete.zig:6-92, interpolated.zig:50-94, xyz.zig (cie_xyz_1931.hpp)
//   Sky.bakeSky, SkyContext.bakeSky, sunWiDotZ, clippedCanopyMapping   sky.zig:172-343
#pragma once


#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cie_xyz_1931.hpp"
#include "hmath.hpp"
#include "prague_sky.hpp"

namespace zyg {
namespace skybake {

// base/random/generator.zig:1-47
struct Pcg {
    uint64_t state = 0, inc = 1;
    void     start(uint64_t st, uint64_t sequence) {
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

// Interpolated.integrate (interpolated.zig:50-94) over the 1 nm CIE tables: exact for the piecewise-linear function, zero outside
inline float integrateCie(const float* intensities, float a, float b) {
    const int   len = cie::kNum;
    const float first = cie::kStart, last = cie::kStart + float(len - 1);
    const float start = std::max(a, first), end = std::min(b, last);
    if (end <= start) return 0.f;
    // std.sort.lowerBound over the wavelengths 360, 361, ...: the first entry >= start
    int it = int(std::ceil(start - first));
    it = std::min(std::max(it, 0), len);
    int   index = std::max(it, 1) - 1;
    float integral = 0.f;
    for (; index + 1 < len && end >= cie::kStart + float(index); ++index) {
        const float wl0 = cie::kStart + float(index), wl1 = cie::kStart + float(index + 1);
        const float c0 = std::max(wl0, start), c1 = std::min(wl1, end);
        if (c1 <= c0) continue;
        const float int0 = intensities[index], int1 = intensities[index + 1];
        const float inv = 1.f / (wl1 - wl0);
        auto        lerp = [](float x, float y, float t) { return std::fmaf(1.f - t, x, t * y); };  // math.lerp (util.zig:3-16)
        const float interp0 = lerp(int0, int1, (c0 - wl0) * inv), interp1 = lerp(int0, int1, (c1 - wl0) * inv);
        integral += 0.5f * (interp0 + interp1) * (c1 - c0);
    }
    return integral;
}

// DiscreteSpectralPowerDistribution(12, 380, 740) (sky_model.zig:9)
struct Spectrum {
    static constexpr int   kBins = 12;
    static constexpr float kStart = 380.f, kEnd = 740.f, kStep = (kEnd - kStart) / float(kBins);
    float values[kBins];

    static const Vec4f* cie() {  // staticInit (discrete.zig:21-38): the bin averages of the three matching functions
        static Vec4f table[kBins];
        static bool  ready = false;
        if (!ready) {
            for (int i = 0; i < kBins; ++i) {
                const float a = kStart + float(i) * kStep, b = a + kStep;
                table[i] = vec(integrateCie(cie::kX, a, b) / kStep, integrateCie(cie::kY, a, b) / kStep, integrateCie(cie::kZ, a, b) / kStep, 0.f);
            }
            ready = true;
        }
        return table;
    }
    static float randomWavelength(int bin, float r) { return kStart + (float(bin) + r) * kStep; }
    Vec4f        XYZ() const {  // discrete.zig:77-84
        const Vec4f* c = cie();
        Vec4f        tri = splat(0.f);
        for (int i = 0; i < kBins; ++i) tri = tri + splat(kStep * values[i]) * c[i];
        return tri;
    }
};

inline Vec4f XYZtoAP1(const Vec4f& xyz) {  // base/spectrum/aces.zig:66-74
    return vec(1.66058533f, -0.65992606f, 0.00900257f, 0.f) * splat(xyz[0]) + vec(-0.31529556f, 1.60839147f, -0.00356688f, 0.f) * splat(xyz[1]) +
           vec(-0.24150933f, 0.01729859f, 0.91364331f, 0.f) * splat(xyz[2]);
}

inline void goldenRatio1D(float* samples, int n, float r) {  // base/math/sample_distribution.zig:21-34
    float x = r;
    for (int i = 0; i < n; ++i) {
        samples[i] = x;
        x += 0.618033988749894f;
        if (x >= 1.f) x -= 1.f;
    }
}

// sky_model.zig:23-137 over prague::Model.  The reference asks the model for one wavelength at a time (12 bins x 24 golden-ratio
// samples per direction); the model's answer only depends on the wavelength's channel, so a direction's channels are evaluated once
// and the samples pick theirs (prague_sky.hpp); the f32 sums run in the reference's order.
struct Model {
    std::unique_ptr<prague::Model> model;
    Vec4f                          sun_direction, shadow_direction;

    void init(std::shared_ptr<const prague::Dataset> dataset, const Vec4f& sun_dir, float visibility, float albedo) {
        const float sun_elevation = -std::asin(sun_dir[1]);
        model.reset(new prague::Model(std::move(dataset), double(sun_elevation), double(visibility), double(albedo)));
        sun_direction = sun_dir;
        const float d = std::sqrt(sun_dir[0] * sun_dir[0] + sun_dir[2] * sun_dir[2]);
        shadow_direction = vec((-sun_dir[0] / d) * sun_dir[1], std::sqrt(1.f - sun_dir[1] * sun_dir[1]), (-sun_dir[2] / d) * sun_dir[1], 0.f);
    }
    // per bin the mean over 24 wavelengths of value(wavelength), as sky_model.zig:88-107 / 118-134 accumulate it
    template <class F>
    Vec4f spectral(Pcg& rng, F value) const {
        float    samples[24];
        Spectrum radiance;
        for (int i = 0; i < Spectrum::kBins; ++i) {
            goldenRatio1D(samples, 24, rng.randomFloat());
            float rwl = 0.f;
            for (float s : samples) rwl += float(value(double(Spectrum::randomWavelength(i, s)))) / 24.f;
            radiance.values[i] = rwl;
        }
        return XYZtoAP1(radiance.XYZ());
    }
    Vec4f evaluateSky(const Vec4f& wi, Pcg& rng) const {  // sky_model.zig:80-110
        const float theta = std::acos(clamp_(wi[1], -1.f, 1.f));
        const float gamma = std::acos(clamp_(-dot3(wi, sun_direction), -1.f, 1.f));
        const float shadow = std::acos(clamp_(dot3(wi, shadow_direction), -1.f, 1.f));
        double      channel[prague::Model::kMaxChannels];
        model->skyChannels(double(theta), double(gamma), double(shadow), channel);
        return spectral(rng, [&](double wavelength) {
            const int c = model->channelOf(wavelength);
            return c < 0 || c >= prague::Model::kMaxChannels ? 0. : channel[c];
        });
    }
    Vec4f evaluateSun(float wi_dot_z, Pcg& rng) const {  // sky_model.zig:112-137
        const float theta = std::acos(clamp_(wi_dot_z, 0.f, 1.f));
        double      tau[prague::Model::kMaxChannels];
        model->transmittanceChannels(double(theta), prague::kOutOfAtmosphere, tau);
        return spectral(rng, [&](double wavelength) {
            const int c = model->channelOf(wavelength);
            return c < 0 || c >= prague::Model::kMaxChannels ? 0. : prague::Model::extraterrestrial(wavelength) * tau[c];
        });
    }
};

// SkyContext.clippedCanopyMapping (sky.zig:324-341): false = outside the (slightly enlarged) disc
inline bool clippedCanopyMapping(float u, float v, float e, Vec4f& wi) {
    float       dx = 2.f * u - 1.f, dy = 2.f * v - 1.f;
    const float l = std::sqrt(dx * dx + dy * dy);
    if (l > 1.f + e) return false;
    if (l >= 1.f - e) {
        dx /= l + e;
        dy /= l + e;
    }
    // Canopy.diskToHemisphereEquidistant (canopy.zig:189-211)
    const float longitude = std::atan2(-dy, dx);
    const float r = std::sqrt(dx * dx + dy * dy);
    const float colatitude = r * float(M_PI / 2.0);
    const float sin_col = std::sin(colatitude), cos_col = std::cos(colatitude);
    const float sin_lon = std::sin(longitude), cos_lon = std::cos(longitude);
    const Vec4f dir = vec(sin_col * cos_lon, sin_col * sin_lon, cos_col);
    wi = vec(dir[0], dir[2], -dir[1], 0.f);
    return true;
}

// Sky.bakeSky (sky.zig:172-221 + SkyContext.bakeSky 284-322): sun_rgb = sun_samples x 3 (the radiance across the disc, v = i / (n - 1)
// through sunWiDotZ), sky_rgb = dimension^2 x 3 (the canopy image, its left half computed, the right half its mirror image).
// sun_sin_elevation = sun_direction[1] of the Sky's rotation (sky.zig:176-180).
inline void bake(std::shared_ptr<const prague::Dataset> dataset, float sun_sin_elevation, float visibility, float albedo, float sun_radius, uint32_t sun_samples,
                 int32_t dimension, std::vector<float>& sun_rgb, std::vector<float>& sky_rgb) {
    const float sin_el = sun_sin_elevation;
    Model       model;
    model.init(std::move(dataset), vec(0.f, sin_el, std::sqrt(1.f - sin_el * sin_el), 0.f), visibility, albedo);
    sun_rgb.assign(size_t(sun_samples) * 3, 0.f);
    const float n = float(sun_samples - 1);
    Pcg         rng;
    rng.start(0, 0);
    const float sun_elevation = -std::asin(sin_el);
    for (uint32_t i = 0; i < sun_samples; ++i) {
        const float v = float(i) / n;
        const float wi_dot_z = std::sin(sun_elevation + ((2.f * v) - 1.f) * sun_radius);  // sunWiDotZ (sky.zig:271-275)
        const Vec4f s = model.evaluateSun(wi_dot_z, rng);
        for (int c = 0; c < 3; ++c) sun_rgb[3 * i + size_t(c)] = s[c];
    }
    sky_rgb.assign(size_t(dimension) * size_t(dimension) * 3, 0.f);
    const float           idf = 1.f / float(dimension);
    const int32_t         half_width = dimension / 2, back = dimension - 1;
    std::atomic<uint32_t> current{0};
    auto                  rows = [&]() {
        Pcg row_rng;
        for (;;) {
            const uint32_t y = current.fetch_add(1);
            if (y >= uint32_t(dimension)) return;
            const float v = idf * (float(y) + 0.5f);
            for (int32_t x = 0; x < half_width; ++x) {
                row_rng.start(0, uint64_t(y) * uint64_t(dimension) + uint64_t(x));
                const float u = idf * (float(x) + 0.5f);
                Vec4f       wi, li = splat(0.f);
                if (clippedCanopyMapping(u, v, 3.5f * idf, wi)) li = model.evaluateSky(normalize3(wi), row_rng);
                for (int32_t xx : {x, back - x}) {
                    float* o = sky_rgb.data() + 3 * (size_t(y) * size_t(dimension) + size_t(xx));
                    o[0] = li[0], o[1] = li[1], o[2] = li[2];
                }
            }
        }
    };
    const unsigned           workers = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < workers; ++t) pool.emplace_back(rows);
    rows();
    for (std::thread& t : pool) t.join();
}

}  // namespace skybake
}  // namespace zyg
