// The Prague sky model, ground-level version (Wilkie et al. 2021, "A Fitted Radiance and Attenuation Model for Realistic
// Atmospheres"): an own reader of the model's dataset and an own evaluator of its sky radiance, solar radiance and transmittance,
// for the Sky entity's bake (sky_bake.hpp).  The reference links the authors' C implementation
// (thirdparty/arpraguesky/ArPragueSkyModelGround.c; interface ArPragueSkyModelGround.h:170-216, caller sky_model.zig:32-137); this
// file computes the same numbers from the same file, written for the bake's access pattern:
//
//   * the dataset is parsed once into `Dataset` (shared by every Model made from it);
//   * a `Model` fixes (solar elevation, visibility, albedo), as the reference's state does, and resolves the four parameter axes to
//     (lower grid index, fraction, whether the upper neighbour takes part) when it is made, not at every query;
//   * sky radiance does not interpolate across wavelength: a query's answer is one of `channels` numbers per direction
//     (ArPragueSkyModelGround.c:706-736), and the transmittance's wavelength weight is fixed at zero (:1101-1103).  The bake asks for
//     288 wavelengths per direction, so `skyChannels` / `transmittanceChannels` evaluate a direction's channels once and the
//     wavelength samples pick from them: 26 times fewer reconstructions than calling the per-wavelength entry 288 times.
//
// Arithmetic follows the C implementation step for step where rounding is involved (f32 slope / value pairs, the f32 roots of the
// circle intersection, the f32 partial sums of the transmittance's basis) so that the own-vs-reference test
// (tests/test_prague_sky.py, against oracle/_ref/libarpraguesky.so compiled from the reference's file) can hold 1e-9 relative.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "solar_spectrum.hpp"

namespace zyg {
namespace prague {

// A gzip (or plain: zlib reads both) stream of little-endian records
class Stream {
   public:
    explicit Stream(const std::string& path) : file_(gzopen(path.c_str(), "rb")) {}
    ~Stream() {
        if (file_) gzclose(file_);
    }
    bool ok() const { return nullptr != file_; }
    template <class T>
    bool read(T* dst, size_t count) {
        size_t bytes = sizeof(T) * count;
        char*  out = reinterpret_cast<char*>(dst);
        while (bytes > 0) {  // (gzread takes an unsigned length)
            const unsigned chunk = unsigned(std::min<size_t>(bytes, 1u << 30));
            const int      got = gzread(file_, out, chunk);
            if (got <= 0) return false;
            out += got;
            bytes -= size_t(got);
        }
        return true;
    }
    template <class T>
    bool array(std::vector<T>& dst, int64_t count, int64_t limit = int64_t(1) << 31) {
        if (count < 0 || count > limit) return false;
        dst.resize(size_t(count));
        return 0 == count || read(dst.data(), size_t(count));
    }

   private:
    gzFile file_;
};

inline double halfToDouble(uint16_t h) {
    const int    exponent = (h >> 10) & 31, mantissa = h & 1023;
    const double sign = (h & 0x8000) ? -1.0 : 1.0;
    if (0 == exponent) return sign * std::ldexp(double(mantissa), -24);
    if (31 == exponent) return mantissa ? std::nan("") : sign * HUGE_VAL;
    return sign * std::ldexp(double(1024 + mantissa), exponent - 25);
}

// A piecewise-linear function over shared break points, one (slope, value at the segment's start) f32 pair per segment
struct Piecewise {
    static int segments(int nbreaks) { return nbreaks - 1; }
    static int floats(int nbreaks) { return 2 * (nbreaks - 1); }
    // (ArPragueSkyModelGround.c:97-109: values f16 / scale rounded to f32, the slope an f32 quotient)
    static void fit(const std::vector<double>& breaks, const uint16_t* values, double scale, float* out) {
        for (size_t i = 0; i + 1 < breaks.size(); ++i) {
            const float here = float(halfToDouble(values[i]) / scale), next = float(halfToDouble(values[i + 1]) / scale);
            out[2 * i] = (next - here) / float(breaks[i + 1] - breaks[i]);
            out[2 * i + 1] = here;
        }
    }
    // the first segment whose end is not below x; an x past the last break extends the last segment
    static int segmentOf(const std::vector<double>& breaks, double x) {
        const int last = int(breaks.size()) - 2;
        int       s = 0;
        while (s < last && breaks[size_t(s) + 1] < x) ++s;
        return s;
    }
    static double at(const std::vector<double>& breaks, const float* pairs, int segment, double x) {
        return double(pairs[2 * segment]) * (x - breaks[size_t(segment)]) + double(pairs[2 * segment + 1]);
    }
};

struct Dataset {
    // radiance: a grid over (visibility, albedo, altitude, elevation, channel), channel innermost; per grid point `components`
    // products sun(gamma) * zenith(alpha) summed, times emph(theta)
    std::vector<double> visibilities, albedos, altitudes, elevations;  // km, [0, 1], m, degrees
    int32_t             channels = 0;
    double              channel_start = 0., channel_width = 0.;  // nm
    int32_t             components = 0;
    std::vector<double> sun_breaks, zenith_breaks, emph_breaks;
    size_t              component_floats = 0, point_floats = 0;
    std::vector<float>  radiance;
    // transmittance: sum over `rank` of U[altitude][d][a][k] * V[visibility][altitude][wavelength][k], squared
    int32_t            n_a = 0, n_d = 0, rank = 0;
    std::vector<float> t_altitudes, t_visibilities, u, v;
    static constexpr int kTransmittanceWavelengths = 11;

    const float* point(int visibility, int albedo, int altitude, int elevation, int channel) const {
        size_t i = size_t(visibility);
        i = i * albedos.size() + size_t(albedo);
        i = i * altitudes.size() + size_t(altitude);
        i = i * elevations.size() + size_t(elevation);
        i = i * size_t(channels) + size_t(channel);
        return radiance.data() + i * point_floats;
    }

    bool load(const std::string& path, std::string& why) {
        Stream in(path);
        if (!in.ok()) return why = "cannot read " + path, false;
        auto axis = [&](std::vector<double>& values, int least) {
            int32_t n = 0;
            return in.read(&n, 1) && n >= least && in.array(values, n, 1 << 20);
        };
        if (!axis(visibilities, 1) || !axis(albedos, 1) || !axis(altitudes, 1) || !axis(elevations, 1)) return why = path + ": parameter axes", false;
        if (!in.read(&channels, 1) || channels < 1 || !in.read(&channel_start, 1) || channel_start < 0. || !in.read(&channel_width, 1) || !(channel_width > 0.))
            return why = path + ": channels", false;
        if (!in.read(&components, 1) || components < 1 || components > 64) return why = path + ": tensor components", false;
        if (!axis(sun_breaks, 2) || !axis(zenith_breaks, 2) || !axis(emph_breaks, 2)) return why = path + ": break points", false;
        const int ns = int(sun_breaks.size()), nz = int(zenith_breaks.size()), ne = int(emph_breaks.size());
        component_floats = size_t(Piecewise::floats(ns) + Piecewise::floats(nz));
        point_floats = size_t(components) * component_floats + size_t(Piecewise::floats(ne));
        const size_t points = visibilities.size() * albedos.size() * altitudes.size() * elevations.size() * size_t(channels);
        if (points * point_floats > (size_t(1) << 32)) return why = path + ": implausible size", false;
        radiance.resize(points * point_floats);
        std::vector<uint16_t> halves(size_t(std::max(ns, std::max(nz, ne))));
        float*                out = radiance.data();
        for (size_t p = 0; p < points; ++p) {
            for (int c = 0; c < components; ++c) {
                double zenith_scale = 1.;
                if (!in.read(halves.data(), size_t(ns))) return why = path + ": radiance data ends early", false;
                Piecewise::fit(sun_breaks, halves.data(), 1., out);
                out += Piecewise::floats(ns);
                if (!in.read(&zenith_scale, 1) || !in.read(halves.data(), size_t(nz))) return why = path + ": radiance data ends early", false;
                Piecewise::fit(zenith_breaks, halves.data(), zenith_scale, out);
                out += Piecewise::floats(nz);
            }
            if (!in.read(halves.data(), size_t(ne))) return why = path + ": radiance data ends early", false;
            Piecewise::fit(emph_breaks, halves.data(), 1., out);
            out += Piecewise::floats(ne);
        }
        int32_t head[5];  // n_d, n_a, visibilities, altitudes, rank
        if (!in.read(head, 5) || head[0] < 1 || head[1] < 1 || head[2] < 1 || head[3] < 1 || head[4] < 1 || head[4] > 64)
            return why = path + ": transmittance header", false;
        n_d = head[0], n_a = head[1], rank = head[4];
        if (!in.array(t_altitudes, head[3], 1 << 20) || !in.array(t_visibilities, head[2], 1 << 20)) return why = path + ": transmittance axes", false;
        if (!in.array(u, int64_t(n_d) * n_a * rank * head[3]) || !in.array(v, int64_t(head[2]) * rank * kTransmittanceWavelengths * head[3]))
            return why = path + ": transmittance data ends early", false;
        return true;
    }
};

// Where a parameter falls on its axis: the grid index below it, the fraction towards the next one, and whether that next one counts
struct AxisPosition {
    int    low = 0;
    double fraction = 0.;
    bool   blend = false;

    // a value below / above the axis sticks to its end; within 1e-6 of a grid value it is that grid value (:443-471); a fraction
    // below 1e-6 and the axis' last index do not blend (:499, 565, 619, 673)
    static AxisPosition on(const std::vector<double>& values, double param) {
        const int n = int(values.size());
        double    position = 0.;
        if (param < values[0]) {
            position = 0.;
        } else if (param > values[size_t(n) - 1]) {
            position = double(n) - 1.;
        } else {
            for (int i = 0; i < n; ++i) {
                const double value = values[size_t(i)];
                if (std::fabs(value - param) < 1e-6) {
                    position = double(i);
                    break;
                }
                if (param < value) {
                    position = double(i) - (value - param) / (value - values[size_t(i) - 1]);
                    break;
                }
            }
        }
        AxisPosition p;
        p.low = int(position);
        p.fraction = position - double(p.low);
        p.blend = !(p.fraction < 1e-6 || p.low >= n - 1);
        return p;
    }
};

inline double mix(double from, double to, double t) { return (1. - t) * from + t * to; }

constexpr double kPlanetRadius = 6378000., kAtmosphereWidth = 100000.;  // m (ArPragueSkyModelGround.h:88-91)
constexpr double kLongestPath = 1571524.413613;                         // m along the ground to the atmosphere's edge (:868)
constexpr double kOutOfAtmosphere = 5.78960446186580977117855E+76;      // the distance solar radiance is attenuated over (:76, 812-817)

class Model {
   public:
    static constexpr int kMaxChannels = 16;

    // elevation: the sun's, radians; visibility km; albedo [0, 1]
    Model(std::shared_ptr<const Dataset> dataset, double elevation, double visibility, double albedo) : data_(std::move(dataset)), below_horizon_(elevation < 0.) {
        const Dataset& d = *data_;
        axes_[0] = AxisPosition::on(d.elevations, elevation * (180. / 3.141592653589793));
        axes_[1] = AxisPosition::on(d.altitudes, 0.);
        axes_[2] = AxisPosition::on(d.visibilities, visibility);
        axes_[3] = AxisPosition::on(d.albedos, albedo);
        // the transmittance's own visibility axis (f32 values), clamped to its ends (:1067-1092)
        const std::vector<float>& tv = d.t_visibilities;
        const int                 n = int(tv.size());
        if (visibility <= double(tv[0])) {
            t_vis_low_ = 0, t_vis_step_ = 0, t_vis_weight_ = 1.;
        } else if (visibility >= double(tv[size_t(n) - 1])) {
            t_vis_low_ = n - 1, t_vis_step_ = 0, t_vis_weight_ = 0.;
        } else {
            for (int i = 1; i < n; ++i) {
                if (visibility < double(tv[size_t(i)])) {
                    t_vis_low_ = i - 1, t_vis_step_ = 1;
                    t_vis_weight_ = (visibility - double(tv[size_t(i) - 1])) / double(tv[size_t(i)] - tv[size_t(i) - 1]);
                    break;
                }
            }
        }
    }

    int channels() const { return std::min(data_->channels, int32_t(kMaxChannels)); }
    // the channel a wavelength (nm) falls into, -1 outside the model's range (:750-752)
    int channelOf(double wavelength) const {
        const double c = (wavelength - data_->channel_start) / data_->channel_width;
        return (c >= double(data_->channels) || c < 0.) ? -1 : int(c);
    }

    // Sky radiance of every channel for a view direction: theta from the zenith, gamma from the sun, shadow from the shadow plane's
    // normal (used instead of theta while the sun is below the horizon), radians (:738-787)
    void skyChannels(double theta, double gamma, double shadow, double* out) const {
        const Dataset& d = *data_;
        const double   alpha = below_horizon_ ? shadow : theta;
        Angles         q;
        q.gamma = gamma, q.alpha = alpha, q.theta = theta;
        q.gamma_segment = Piecewise::segmentOf(d.sun_breaks, gamma);
        q.alpha_segment = Piecewise::segmentOf(d.zenith_breaks, alpha);
        q.theta_segment = Piecewise::segmentOf(d.emph_breaks, theta);
        for (int c = 0, n = channels(); c < n; ++c) {
            int index[4] = {0, 0, 0, 0};
            out[c] = blended(3, index, c, q);
        }
    }
    double skyRadiance(double theta, double gamma, double shadow, double wavelength) const {
        const int c = channelOf(wavelength);
        if (c < 0 || c >= kMaxChannels) return 0.;
        double all[kMaxChannels];
        skyChannels(theta, gamma, shadow, all);  // (the test hook's form; the bake calls skyChannels once per direction)
        return all[c];
    }

    // Transmittance of every channel from the ground along theta over `distance` metres (:1094-1134 with altitude 0)
    void transmittanceChannels(double theta, double distance, double* out) const {
        double a = 0., d = 0.;
        atmospherePosition(theta, distance, a, d);
        const Basis b = basisAt(a, d);
        for (int c = 0, n = std::min(channels(), int(Dataset::kTransmittanceWavelengths)); c < n; ++c) {
            const double low = basisValue(b, t_vis_low_, c), high = basisValue(b, t_vis_low_ + t_vis_step_, c);
            double       t = mix(low, high, t_vis_weight_);
            t = t < 0. ? 0. : (t > 1. ? 1. : t);
            out[c] = t * t;
        }
        for (int c = Dataset::kTransmittanceWavelengths; c < channels(); ++c) out[c] = 0.;
    }
    double transmittance(double theta, double wavelength, double distance) const {
        const int c = channelOf(wavelength);
        if (c < 0 || c >= kMaxChannels || c >= Dataset::kTransmittanceWavelengths) return 0.;
        double all[kMaxChannels];
        transmittanceChannels(theta, distance, all);
        return all[c];
    }
    // the sun's spectral radiance before the atmosphere, interpolated in the 1 nm table (:801-810)
    static double extraterrestrial(double wavelength) {
        const double at = (wavelength - solar::kStart) / solar::kStep;
        if (!(at >= 0.)) return 0.;
        const double below = std::floor(at), t = at - below;
        const int    i = std::min(int(below), solar::kNum - 2);
        return solar::kRadiance[i] * (1. - t) + solar::kRadiance[i + 1] * t;
    }
    double solarRadiance(double theta, double wavelength) const { return extraterrestrial(wavelength) * transmittance(theta, wavelength, kOutOfAtmosphere); }

   private:
    struct Angles {
        double gamma, alpha, theta;
        int    gamma_segment, alpha_segment, theta_segment;
    };
    struct Basis {  // the up to four (a, d) grid cells a position blends, and its weights
        int    a_low, d_low, a_step, d_step;
        double a_weight, d_weight;
    };

    // one grid point's radiance: sum of sun * zenith products, times the emphasis term, not negative (:420-441)
    double reconstruct(const float* point, const Angles& q) const {
        const Dataset& d = *data_;
        const int      sun_floats = Piecewise::floats(int(d.sun_breaks.size()));
        double         sum = 0.;
        for (int c = 0; c < d.components; ++c) {
            const float* sun = point + size_t(c) * d.component_floats;
            sum += Piecewise::at(d.sun_breaks, sun, q.gamma_segment, q.gamma) * Piecewise::at(d.zenith_breaks, sun + sun_floats, q.alpha_segment, q.alpha);
        }
        sum *= Piecewise::at(d.emph_breaks, point + size_t(d.components) * d.component_floats, q.theta_segment, q.theta);
        return sum > 0. ? sum : 0.;
    }
    // multilinear blend over the axes [0, axis], innermost axis 0 = elevation, then altitude, visibility, albedo (:478-704)
    double blended(int axis, int* index, int channel, const Angles& q) const {
        const AxisPosition& p = axes_[axis];
        auto                lower = [&](int i) {
            index[axis] = i;
            return 0 == axis ? reconstruct(data_->point(index[2], index[3], index[1], index[0], channel), q) : blended(axis - 1, index, channel, q);
        };
        const double low = lower(p.low);
        if (!p.blend) return low;
        return mix(low, lower(p.low + 1), p.fraction);
    }

    // the nearer positive root of |origin + t * direction| = radius for origin (0, height), as the f32 it is rounded to (:827-854)
    static bool circleDistance(double x_v, double y_v, double height, double radius, double& t) {
        const double qa = x_v * x_v + y_v * y_v, qb = 2. * height * y_v, qc = height * height - radius * radius;
        double       disc = qb * qb - 4. * qa * qc;
        if (disc <= 0.) return false;
        disc = std::sqrt(disc);
        const float r1 = float((-qb + disc) / (2. * qa)), r2 = float((-qb - disc) / (2. * qa));
        t = (r1 > 0.f && r2 > 0.f) ? double(std::min(r1, r2)) : double(std::max(r1, r2));
        return t > 0.;
    }
    // (a, d): cube root of the relative height and fourth root of the relative ground distance of the point `distance` along theta,
    // or of the atmosphere's edge if that is nearer (:856-936, view point on the ground)
    static void atmospherePosition(double theta, double distance, double& a, double& d) {
        const double x_v = std::sin(theta), y_v = std::cos(theta);
        double       reach = 0.;
        a = d = 0.;
        if (theta <= 0.5 * 3.141592653589793) {
            if (!circleDistance(x_v, y_v, kPlanetRadius, kPlanetRadius + kAtmosphereWidth, reach)) return;
        }
        const double along = std::min(distance, reach);
        const double x = x_v * along, y = y_v * along + kPlanetRadius + 0.;
        const double r = std::sqrt(x * x + y * y);
        a = std::cbrt(std::max(r - kPlanetRadius, 0.) / kAtmosphereWidth);
        const double ground = std::acos(y / r) * kPlanetRadius / kLongestPath;
        d = std::min(std::sqrt(std::sqrt(ground)), 1.);
    }
    // where w lies between the grid lines lo and hi in the power-p space the grid is uniform in (:1009-1014)
    static double powerWeight(double lo, double hi, double w, double p) {
        const double c1 = std::pow(lo, p), c2 = std::pow(hi, p);
        return (std::pow(w, p) - c1) / (c2 - c1);
    }
    Basis basisAt(double a, double d) const {  // (:1027-1058)
        const Dataset& ds = *data_;
        Basis          b;
        b.a_low = int(std::floor(a * double(ds.n_a))), b.d_low = int(std::floor(d * double(ds.n_d)));
        b.a_step = b.d_step = 0;
        b.a_weight = b.d_weight = 0.;
        if (b.a_low < ds.n_a - 1) {
            b.a_step = 1;
            b.a_weight = powerWeight(double(b.a_low) / double(ds.n_a), double(b.a_low + 1) / double(ds.n_a), a, 3.);
        } else {
            b.a_low = ds.n_a - 1;
        }
        if (b.d_low < ds.n_d - 1) {
            b.d_step = 1;
            b.d_weight = powerWeight(double(b.d_low) / double(ds.n_d), double(b.d_low + 1) / double(ds.n_d), d, 4.);
        } else {
            b.d_low = ds.n_d - 1;
        }
        b.a_weight = std::min(std::max(b.a_weight, 0.), 1.);
        b.d_weight = std::min(std::max(b.d_weight, 0.), 1.);
        return b;
    }
    // sum_k U[d][a][k] * V[visibility][wavelength][k] at the basis' cells (f32 partial sums), blended along d, then along a (:967-1007)
    double basisValue(const Basis& b, int visibility, int wavelength) const {
        const Dataset& ds = *data_;
        const size_t   altitudes = ds.t_altitudes.size();
        const float*   vk = ds.v.data() + (size_t(visibility) * altitudes * Dataset::kTransmittanceWavelengths + size_t(wavelength)) * size_t(ds.rank);
        float          t[4] = {0.f, 0.f, 0.f, 0.f};
        int            cell = 0;
        for (int al = b.a_low; al <= b.a_low + b.a_step; ++al) {
            for (int dl = b.d_low; dl <= b.d_low + b.d_step; ++dl, ++cell) {
                const float* uk = ds.u.data() + (size_t(dl) * size_t(ds.n_a) + size_t(al)) * size_t(ds.rank);
                for (int k = 0; k < ds.rank; ++k) t[cell] = float(double(t[cell]) + double(uk[k]) * double(vk[k]));
            }
        }
        if (1 == b.d_step) {
            t[0] = float(mix(double(t[0]), double(t[1]), b.d_weight));
            t[1] = float(mix(double(t[2]), double(t[3]), b.d_weight));
        }
        if (1 == b.a_step) t[0] = float(mix(double(t[0]), double(t[1]), b.a_weight));
        return double(t[0]);
    }

    std::shared_ptr<const Dataset> data_;
    bool                           below_horizon_;
    AxisPosition                   axes_[4];  // elevation, altitude, visibility, albedo
    int                            t_vis_low_ = 0, t_vis_step_ = 0;
    double                         t_vis_weight_ = 0.;
};

}  // namespace prague
}  // namespace zyg
