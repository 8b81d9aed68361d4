// Image files read on the host for file-backed textures (core/image/image_provider.zig:27-69):
//   PNG   core/image/encoding/png/png_reader.zig: 8 bits per channel, Grayscale / Truecolor / Truecolor + alpha, not interlaced
//         (anything else is the reference's PNGBitDepthNotSupported / IndexedPNGNotSupperted / InterlacedPNGNotSupported); the
//         zlib stream through libz (the reference links miniz), the five scanline filters (png_reader.zig:336-408)
//   RGBE  core/image/encoding/rgbe/rgbe_reader.zig: "#?" header, FORMAT=32-bit_rle_rgbe, "-Y h +X w", new-style RLE scanlines or
//         flat pixels; a pixel is (mantissa + 0.5) * 2^(e - 136) per channel (rgbeToVec4f), sRGB -> AP1, stored as f16
//   EXR   core/image/encoding/exr/exr_reader.zig: scan-line files with ZIP compression (16-line blocks, a block whose size is the
//         uncompressed one is stored as it is), all channels half or all float, the channels in the file's (alphabetical) order so
//         that R is the last; a colour usage converts sRGB -> AP1
// IES files are not read (refused by name where a material asks for them).
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "hmath.hpp"

namespace zyg {
namespace imgr {

inline uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

// pixels: height rows of width * channels bytes, top row first
inline bool readPng(const std::string& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels, std::string& error) {
    static const unsigned char kSignature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || 0 != std::memcmp(file.data(), kSignature, 8)) return error = "not a PNG file", false;
    const unsigned char* d = reinterpret_cast<const unsigned char*>(file.data());
    size_t               at = 8;
    std::string          idat;
    bool                 header = false;
    while (at + 12 <= file.size()) {
        const uint32_t length = be32(d + at);
        const char*    type = file.data() + at + 4;
        if (at + 12 + size_t(length) > file.size()) return error = "a PNG chunk reaches past the file", false;
        if (0 == std::memcmp(type, "IHDR", 4)) {
            if (length < 13) return error = "a short PNG header", false;
            width = int(be32(d + at + 8));
            height = int(be32(d + at + 12));
            if (8 != d[at + 16]) return error = "a PNG bit depth other than 8", false;
            const unsigned color_type = d[at + 17];
            channels = 0 == color_type ? 1 : 2 == color_type ? 3 : 6 == color_type ? 4 : 0;
            if (0 == channels) return error = "an indexed or grey + alpha PNG", false;
            if (d[at + 20] > 0) return error = "an interlaced PNG", false;
            header = true;
        } else if (0 == std::memcmp(type, "IDAT", 4)) {
            idat.append(file.data() + at + 8, length);
        } else if (0 == std::memcmp(type, "IEND", 4)) {
            break;
        }
        at += 12 + size_t(length);
    }
    if (!header || width <= 0 || height <= 0) return error = "a PNG without a header", false;
    // sizes come from the file: bound them before anything is allocated from them (a deflate stream expands at most ~1032 times)
    if (width > 65536 || height > 65536) return error = "a PNG larger than 65536 pixels on a side", false;
    const size_t         row = size_t(width) * size_t(channels);
    if ((row + 1) * size_t(height) > 1032 * idat.size() + 65536) return error = "the PNG's image data does not inflate to its size", false;
    std::vector<uint8_t> raw((row + 1) * size_t(height));
    uLongf               raw_size = uLongf(raw.size());
    if (Z_OK != uncompress(raw.data(), &raw_size, reinterpret_cast<const Bytef*>(idat.data()), uLong(idat.size())) || raw_size != raw.size())
        return error = "the PNG's image data does not inflate to its size", false;
    pixels.assign(row * size_t(height), 0);
    const int bpp = channels;
    for (int y = 0; y < height; ++y) {
        const uint8_t  filter = raw[(row + 1) * size_t(y)];
        const uint8_t* in = raw.data() + (row + 1) * size_t(y) + 1;
        uint8_t*       out = pixels.data() + row * size_t(y);
        const uint8_t* up = y > 0 ? out - row : nullptr;
        for (size_t x = 0; x < row; ++x) {
            const int a = x >= size_t(bpp) ? out[x - size_t(bpp)] : 0, b = up ? up[x] : 0, c = (up && x >= size_t(bpp)) ? up[x - size_t(bpp)] : 0;
            int       predicted = 0;
            switch (filter) {
                case 0: predicted = 0; break;
                case 1: predicted = a; break;
                case 2: predicted = b; break;
                case 3: predicted = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    predicted = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return error = "an unknown PNG scanline filter", false;
            }
            out[x] = uint8_t(in[x] + predicted);
        }
    }
    return true;
}

inline float halfRounded(float f) {  // the value @floatCast(f16) keeps (round to nearest even; the image is Half3 in the reference)
    if (!(std::fabs(f) < 65520.f)) return f > 0.f ? INFINITY : (f < 0.f ? -INFINITY : f);
    if (std::fabs(f) < 6.103515625e-05f) {  // subnormal halves: multiples of 2^-24
        return std::nearbyint(f * 16777216.f) / 16777216.f;
    }
    int         e;
    const float m = std::frexp(f, &e);  // f = m * 2^e, 0.5 <= |m| < 1: 11 significant bits
    return std::ldexp(std::nearbyint(m * 2048.f) / 2048.f, e);
}

// rgb: height rows of width * 3 floats (AP1), top row first
inline bool readRgbe(const std::string& file, int& width, int& height, std::vector<float>& rgb, std::string& error) {
    size_t at = 0;
    auto   line = [&](std::string& out) {
        const size_t end = file.find('\n', at);
        if (std::string::npos == end) return false;
        out.assign(file, at, end - at);
        at = end + 1;
        return true;
    };
    std::string l;
    if (!line(l) || 0 != l.compare(0, 2, "#?")) return error = "not a Radiance picture", false;
    bool format = false;
    for (;;) {
        if (!line(l)) return error = "a Radiance header without an end", false;
        if (l.empty() || 0 == l[0]) break;
        if (l == "FORMAT=32-bit_rle_rgbe") format = true;
    }
    if (!format) return error = "a Radiance picture that is not 32-bit_rle_rgbe", false;
    if (!line(l) || 2 != std::sscanf(l.c_str(), "-Y %d +X %d", &height, &width) || width <= 0 || height <= 0)
        return error = "a Radiance picture without \"-Y h +X w\"", false;
    const unsigned char* d = reinterpret_cast<const unsigned char*>(file.data());
    rgb.assign(size_t(width) * size_t(height) * 3, 0.f);
    auto store = [&](size_t pixel, const unsigned char* rgbe) {
        if (0 == rgbe[3]) return;
        const float f = std::ldexp(1.f, int(rgbe[3]) - (128 + 8));
        const Vec4f c = sRGBtoAP1(vec((float(rgbe[0]) + 0.5f) * f, (float(rgbe[1]) + 0.5f) * f, (float(rgbe[2]) + 0.5f) * f, 0.f));
        for (int k = 0; k < 3; ++k) rgb[3 * pixel + size_t(k)] = halfRounded(c[k]);
    };
    const size_t pixels = size_t(width) * size_t(height);
    auto         flat = [&](size_t first) {
        if (at + (pixels - first) * 4 > file.size()) return error = "a Radiance picture shorter than its size", false;
        for (size_t p = first; p < pixels; ++p, at += 4) store(p, d + at);
        return true;
    };
    if (width < 8 || width > 0x7fff) return flat(0);
    std::vector<unsigned char> scanline(size_t(width) * 4);
    for (int y = 0; y < height; ++y) {
        if (at + 4 > file.size()) return error = "a Radiance picture shorter than its size", false;
        const unsigned char* h = d + at;
        if (2 != h[0] || 2 != h[1] || (h[2] & 0x80)) return flat(size_t(y) * size_t(width));  // not run-length encoded from here on
        if (((int(h[2]) << 8) | int(h[3])) != width) return error = "a Radiance scanline of the wrong width", false;
        at += 4;
        for (int channel = 0; channel < 4; ++channel) {
            size_t x = 0;
            while (x < size_t(width)) {
                if (at + 2 > file.size()) return error = "a Radiance picture shorter than its size", false;
                const unsigned count = d[at], value = d[at + 1];
                at += 2;
                if (count > 128) {  // a run
                    const unsigned n = count - 128;
                    if (0 == n || x + n > size_t(width)) return error = "bad Radiance scanline data", false;
                    std::memset(scanline.data() + size_t(channel) * size_t(width) + x, int(value), n);
                    x += n;
                } else {  // a literal span: `count` bytes, the first of which has been read
                    if (0 == count || x + count > size_t(width) || at + count - 1 > file.size()) return error = "bad Radiance scanline data", false;
                    scanline[size_t(channel) * size_t(width) + x] = static_cast<unsigned char>(value);
                    std::memcpy(scanline.data() + size_t(channel) * size_t(width) + x + 1, d + at, count - 1);
                    at += count - 1;
                    x += count;
                }
            }
        }
        for (int x = 0; x < width; ++x) {
            const unsigned char rgbe[4] = {scanline[size_t(x)], scanline[size_t(width) + size_t(x)], scanline[2 * size_t(width) + size_t(x)],
                                           scanline[3 * size_t(width) + size_t(x)]};
            store(size_t(y) * size_t(width) + size_t(x), rgbe);
        }
    }
    return true;
}

// ---- IES photometric files (image/encoding/ies/ies_reader.zig:16-394): the luminaire's candela table, resampled into a square
// one-channel image over the octahedral map of directions (the light material's "profile", emittance.zig:20, 45-52).  Read: the
// "IES" header line, everything up to the "TILT=NONE" line is skipped, then whitespace-separated numbers: lamps, lumens, multiplier,
// vertical count, horizontal count, photometric type, seven more that are skipped, the vertical angles, the horizontal angles, the
// candela values (one run of vertical values per horizontal angle).  Resolution = round(360 / the smallest angle step); texel (x, y) =
// the table at the direction octDecode(2 (uv - 0.5)) turned to the file's axes, relative to the table's maximum, saturated, as f16.
struct IesTable {
    uint32_t           type = 1;  // 1: type C (vertical angle from the nadir, horizontal around it); otherwise type A / B angles
    std::vector<float> vertical, horizontal, candela;

    static float catmullRom(const float c[4], float t) {  // ies_reader.zig:224-232
        const float t2 = t * t;
        const float a0 = -0.5f * c[0] + 1.5f * c[1] - 1.5f * c[2] + 0.5f * c[3];
        const float a1 = c[0] - 2.5f * c[1] + 2.0f * c[2] - 0.5f * c[3];
        const float a2 = -0.5f * c[0] + 0.5f * c[2];
        const float a3 = c[1];
        return a0 * t * t2 + a1 * t2 + a2 * t + a3;
    }
    // a neighbour index: mirrored below the first angle, held back from the end (ies_reader.zig:208-216)
    static uint32_t neighbour(int32_t x, int32_t i, int32_t b) {
        const int32_t y = x + i;
        if (y < 0) return uint32_t(1 - y);
        return uint32_t(y >= b ? b - i : y);
    }
    // the candela value towards (phi around the axis, theta from it), radians (Data.sample, ies_reader.zig:31-206)
    float at(float rad_phi, float rad_theta) const {
        float phi = rad_phi * (180.f / 3.14159265358979323846f), theta = rad_theta * (180.f / 3.14159265358979323846f);
        bool  vertical_symmetric = false, non_symmetric = false;
        const float vf = vertical.front(), vb = vertical.back(), hf = horizontal.front(), hb = horizontal.back();
        if (1 == type) {
            if (0.f == hb) {
                phi = 0.f;
                vertical_symmetric = true;
            } else if (90.f == hb) {
                if (phi > 90.f && phi < 180.f) phi = 180.f - phi;
                if (phi > 180.f && phi < 270.f) phi -= 180.f;
                if (phi > 270.f && phi < 360.f) phi = 360.f - phi;
                if (phi < 0.f || phi > 90.f || theta < 0.f || theta > 180.f) return 0.f;
            } else if (180.f == hb) {
                if (phi > 180.f) phi = 360.f - phi;
                if (phi < 0.f || phi > 180.f || theta < 0.f || theta > 180.f) return 0.f;
            } else if (90.f == hf && 270.f == hb) {
                if (phi < 90.f || phi > 270.f) phi = 180.f - phi;
                if (phi < 90.f || phi > 270.f) return 0.f;
            } else {
                non_symmetric = true;
            }
        } else {
            theta -= 90.f;
            phi -= 180.f;
            if (vf >= 0.f) theta = std::fabs(theta);
            if (hf >= 0.f) phi = std::fabs(phi);
            if (phi < hf || phi > hb) return 0.f;
            if (theta < vf || theta > vb) return 0.f;
        }
        if (vertical_symmetric && (phi < hf || phi > hb)) return 0.f;
        if (theta < vf || theta > vb) return 0.f;
        // the last angle below the value (std.sort.lowerBound - 1; an exact first angle stays on the first interval)
        auto below = [](const std::vector<float>& a, float x) {
            const size_t lb = size_t(std::lower_bound(a.begin(), a.end(), x) - a.begin());
            return lb > 0 ? lb - 1 : size_t(0);
        };
        const size_t hit = below(horizontal, phi), vit = below(vertical, theta);
        const size_t nv = vertical.size(), nh = horizontal.size();
        float        next_phi;
        if (non_symmetric) {
            if (vit >= nv - 1) return 0.f;
            next_phi = hit < nh - 1 ? horizontal[hit + 1] : horizontal[(hit + 1) % nh];
        } else {
            if (!vertical_symmetric && (hit >= nh - 1 || vit >= nv - 1)) return 0.f;
            next_phi = hit < nh - 1 ? horizontal[hit + 1] : horizontal[hit];
        }
        const float next_theta = vit < nv - 1 ? vertical[vit + 1] : vertical[vit];
        const float d_theta = (theta - vertical[vit]) / (next_theta - vertical[vit]);
        if (vertical_symmetric) {
            const float in0 = candela[vit], in1 = candela[(vit + 1) % nv];
            return std::fmaf(1.f - d_theta, in0, d_theta * in1);  // math.lerp (util.zig:3-16)
        }
        const float   d_phi = (phi - horizontal[hit]) / (next_phi - horizontal[hit]);
        const int32_t iv = int32_t(vit), ih = int32_t(hit), bv = int32_t(nv), bh = int32_t(nh);
        float         rows[4];
        for (int32_t j = -1; j <= 2; ++j) {
            const size_t h = size_t(neighbour(ih, j, bh)) * nv;
            const float  c[4] = {candela[neighbour(iv, -1, bv) + h], candela[neighbour(iv, 0, bv) + h], candela[neighbour(iv, 1, bv) + h], candela[neighbour(iv, 2, bv) + h]};
            rows[j + 1] = catmullRom(c, d_theta);
        }
        return catmullRom(rows, d_phi);
    }
};

inline bool readIes(const std::string& file, int& width, int& height, std::vector<float>& values, std::string& error) {
    if (file.size() < 3 || 0 != file.compare(0, 3, "IES")) return error = "not an IES file", false;
    size_t at = file.find('\n');
    bool   tilt = false;
    while (std::string::npos != at && !tilt) {
        const size_t begin = at + 1, end = file.find('\n', begin);
        tilt = 0 == file.compare(begin, 9, "TILT=NONE");
        at = end;
    }
    if (!tilt || std::string::npos == at) return error = "an IES file without a TILT=NONE line (tilted luminaires are not read)", false;
    std::vector<double> numbers;
    const char*         p = file.c_str() + at;
    for (;;) {
        while (*p && (' ' == *p || '\r' == *p || '\n' == *p || '\t' == *p || ',' == *p)) ++p;
        if (!*p) break;
        char*        stop = nullptr;
        const double x = std::strtod(p, &stop);
        if (stop == p) return error = "an IES file with something that is not a number in its table", false;
        numbers.push_back(x);
        p = stop;
    }
    if (numbers.size() < 13) return error = "an IES file that ends before its angles", false;
    const double nv = numbers[3], nh = numbers[4];
    if (!(nv >= 1) || !(nh >= 1) || nv > 4096 || nh > 4096) return error = "an IES file with implausible angle counts", false;
    IesTable t;
    t.type = uint32_t(numbers[5]);
    const size_t v = size_t(nv), h = size_t(nh);
    if (numbers.size() < 13 + v + h + v * h) return error = "an IES file that ends before its candela values", false;
    float min_angle = 360.f;
    auto  angles = [&](size_t first, size_t count, std::vector<float>& out) {
        out.resize(count);
        float previous = float(numbers[first]);
        out[0] = previous;
        for (size_t i = 1; i < count; ++i) {
            const float c = float(numbers[first + i]);
            min_angle = std::min(min_angle, std::fabs(c - previous));
            out[i] = c;
            previous = c;
        }
    };
    angles(13, v, t.vertical);
    angles(13 + v, h, t.horizontal);
    t.candela.resize(v * h);
    float most = 0.f;
    for (size_t i = 0; i < v * h; ++i) {
        t.candela[i] = float(numbers[13 + v + h + i]);
        most = std::max(most, t.candela[i]);
    }
    if (!(min_angle > 0.f) || !(most > 0.f)) return error = "an IES file without an angle step or without light", false;
    const int res = int(360.f / min_angle + 0.5f);
    if (res < 1 || res > 4096) return error = "an IES file whose angle step gives an implausible resolution", false;
    width = height = res;
    values.resize(size_t(res) * size_t(res));
    const float idf = 1.f / float(res), imi = 1.f / most;
    for (int y = 0; y < res; ++y) {
        const float vv = idf * (float(y) + 0.5f);
        for (int x = 0; x < res; ++x) {
            const float u = idf * (float(x) + 0.5f);
            // octDecode (base/encoding.zig:89-98) of 2 (uv - 0.5)
            float       d[3] = {2.f * (u - 0.5f), 2.f * (vv - 0.5f), 0.f};
            d[2] = -1.f + std::fabs(d[0]) + std::fabs(d[1]);
            const float tt = std::max(d[2], 0.f);
            d[0] += d[0] > 0.f ? -tt : tt;
            d[1] += d[1] > 0.f ? -tt : tt;
            const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            for (float& c : d) c /= len;
            // dirToLatlong of (x, -z, -y) (ies_reader.zig:371-391)
            const float lx = d[0], ly = -d[2], lz = -d[1];
            const float phi = std::atan2(lx, lz), theta = std::acos(ly);
            const float value = t.at(phi < 0.f ? 2.f * 3.14159265358979323846f + phi : phi, theta);
            values[size_t(y) * size_t(res) + size_t(x)] = halfRounded(std::min(std::max(value * imi, 0.f), 1.f));
        }
    }
    return true;
}

inline float halfToFloat(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16, exponent = (h >> 10) & 31u, mantissa = h & 1023u;
    uint32_t       bits;
    if (0 == exponent) {
        if (0 == mantissa) {
            bits = sign;
        } else {  // subnormal: mantissa * 2^-24
            const float    v = std::ldexp(float(mantissa), -24);
            uint32_t       b;
            std::memcpy(&b, &v, 4);
            bits = sign | b;
        }
    } else if (31 == exponent) {
        bits = sign | 0x7F800000u | (mantissa << 13);
    } else {
        bits = sign | ((exponent + 112u) << 23) | (mantissa << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// channels_wanted: 1 or 3 (the swizzle's channel count, at most the file's); out: the display window, rows of width * channels floats
// (a half file's values as floats; with `color` the f16 rounding of the converted colour, as the reference's Half3 image keeps it)
inline bool readExr(const std::string& file, uint32_t channels_wanted, bool color, int& width, int& height, uint32_t& channels, std::vector<float>& out,
                    std::string& error) {
    static const unsigned char kSignature[4] = {0x76, 0x2f, 0x31, 0x01};
    if (file.size() < 8 || 0 != std::memcmp(file.data(), kSignature, 4)) return error = "not an OpenEXR file", false;
    size_t at = 8;  // (the version word is skipped)
    auto   text = [&](std::string& o) {
        const size_t end = file.find('\0', at);
        if (std::string::npos == end) return false;
        o.assign(file, at, end - at);
        at = end + 1;
        return true;
    };
    auto read32 = [&](void* o) {
        if (at + 4 > file.size()) return false;
        std::memcpy(o, file.data() + at, 4);
        at += 4;
        return true;
    };
    int32_t               data_window[4] = {0, 0, 0, 0}, display_window[4] = {0, 0, 0, 0};
    uint8_t               compression = 8;
    std::vector<uint32_t> formats;  // per channel: 0 uint, 1 half, 2 float
    for (;;) {
        std::string name, type;
        if (!text(name)) return error = "an OpenEXR header without an end", false;
        if (name.empty()) break;
        uint32_t size = 0;
        if (!text(type) || !read32(&size) || at + size > file.size()) return error = "a broken OpenEXR attribute", false;
        const size_t next = at + size;
        if ("box2i" == type && 16 == size) {
            int32_t* w = "dataWindow" == name ? data_window : "displayWindow" == name ? display_window : nullptr;
            if (w) std::memcpy(w, file.data() + at, 16);
        } else if ("chlist" == type) {
            for (;;) {
                std::string channel;
                if (!text(channel)) return error = "a broken OpenEXR channel list", false;
                if (channel.empty()) break;
                uint32_t format = 0;
                if (!read32(&format)) return error = "a broken OpenEXR channel list", false;
                at += 12;  // pLinear, reserved, xSampling, ySampling
                formats.push_back(format);
            }
        } else if ("compression" == type && size >= 1) {
            compression = uint8_t(file[at]);
        }
        at = next;
    }
    if (data_window[0] < display_window[0] || data_window[1] < display_window[1] || data_window[2] > display_window[2] || data_window[3] > display_window[3])
        return error = "an OpenEXR data window that is not inside the display window", false;
    // the windows come from the file: the reference traps on what follows, here it is refused (origins at or above 0, a window that is
    // not empty, sizes that fit the arithmetic below)
    if (display_window[0] != 0 || display_window[1] != 0 || data_window[2] < data_window[0] || data_window[3] < data_window[1] ||
        display_window[2] >= 65536 || display_window[3] >= 65536)
        return error = "an OpenEXR display window that does not start at the origin, an empty data window, or an image larger than 65536 pixels on a side", false;
    if (formats.empty()) return error = "an OpenEXR file without channels", false;
    for (uint32_t f : formats)
        if (f != formats[0]) return error = "an OpenEXR file with mixed channel formats", false;
    if (1 != formats[0] && 2 != formats[0]) return error = "an OpenEXR file with integer channels", false;
    if (3 != compression) return error = "an OpenEXR file that is not ZIP compressed", false;
    const bool     half = 1 == formats[0];
    const uint32_t file_channels = uint32_t(formats.size());
    channels = std::min(channels_wanted, file_channels);
    if (2 == channels) channels = 1;
    if (channels > 3) return error = "four-channel OpenEXR images are not provided", false;
    const uint32_t w = uint32_t(data_window[2] - data_window[0] + 1), h = uint32_t(data_window[3] - data_window[1] + 1);
    width = display_window[2] - display_window[0] + 1;
    height = display_window[3] - display_window[1] + 1;
    if (file_channels > 64) return error = "an OpenEXR file with more than 64 channels", false;
    const uint32_t bytes_per_pixel = file_channels * (half ? 2u : 4u), bytes_per_row = w * bytes_per_pixel, rows_per_block = 16u;  // < 2^16 * 2^8
    const uint32_t blocks = (h + rows_per_block - 1u) / rows_per_block;
    at += size_t(blocks) * 8;  // the line offset table
    if (at > file.size()) return error = "an OpenEXR file shorter than its line offset table", false;
    out.assign(size_t(width) * size_t(height) * channels, 0.f);
    std::vector<unsigned char> packed(size_t(bytes_per_row) * rows_per_block), plain(packed.size());
    const uint32_t r_o = file_channels - 1u, g_o = file_channels >= 2 ? file_channels - 2u : 0u, b_o = file_channels >= 3 ? file_channels - 3u : 0u;
    for (uint32_t b = 0; b < blocks; ++b) {
        uint32_t row = 0, size = 0;
        if (!read32(&row) || !read32(&size) || at + size > file.size() || size > packed.size()) return error = "a broken OpenEXR scan-line block", false;
        // a block's first row is a file value too: it must lie in the data window (the rows below it that the window still holds)
        if (int64_t(int32_t(row)) < data_window[1] || int64_t(int32_t(row)) > data_window[3]) return error = "an OpenEXR scan-line block outside the data window", false;
        const uint32_t rows_here = std::min<uint32_t>(uint32_t(data_window[3]) + 1u - row, rows_per_block);
        const size_t   bytes_here = size_t(rows_here) * bytes_per_row;
        if (size < packed.size()) {
            uLongf n = uLongf(plain.size());
            if (Z_OK != uncompress(plain.data(), &n, reinterpret_cast<const Bytef*>(file.data() + at), size)) return error = "an OpenEXR block that does not inflate", false;
            for (size_t t = 1; t < bytes_here; ++t) plain[t] = static_cast<unsigned char>(plain[t - 1] + plain[t] - 128);  // reconstructScalar
            size_t t1 = 0, t2 = (bytes_here + 1) / 2;                                                                       // interleaveScalar
            for (size_t s2 = 0; s2 < bytes_here;) {
                packed[s2++] = plain[t1++];
                if (s2 < bytes_here) packed[s2++] = plain[t2++];
            }
        } else {
            std::memcpy(packed.data(), file.data() + at, size);
        }
        at += size;
        auto value = [&](uint32_t y, uint32_t channel, uint32_t x) {
            const size_t index = size_t(file_channels) * y * w + size_t(channel) * w + x;
            if (half) {
                uint16_t v;
                std::memcpy(&v, packed.data() + 2 * index, 2);
                return halfToFloat(v);
            }
            float v;
            std::memcpy(&v, packed.data() + 4 * index, 4);
            return v;
        };
        for (uint32_t y = 0; y < rows_here; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const size_t p = (size_t(row + y) * size_t(width) + size_t(data_window[0]) + x) * channels;
                if (1 == channels) {
                    out[p] = value(y, r_o, x);
                } else {
                    Vec4f c = vec(value(y, r_o, x), value(y, g_o, x), value(y, b_o, x), 0.f);
                    if (color) c = sRGBtoAP1(c);
                    for (int k = 0; k < 3; ++k) out[p + size_t(k)] = (half && color) ? halfRounded(c[k]) : c[k];
                }
            }
    }
    return true;
}

}  // namespace imgr
}  // namespace zyg
