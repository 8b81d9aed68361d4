// Image files read on the host for file-backed textures (core/image/image_provider.zig:27-69):
//   PNG   core/image/encoding/png/png_reader.zig: 8 bits per channel, Grayscale / Truecolor / Truecolor + alpha, not interlaced
//         (anything else is the reference's PNGBitDepthNotSupported / IndexedPNGNotSupperted / InterlacedPNGNotSupported); the
//         zlib stream through libz (the reference links miniz), the five scanline filters (png_reader.zig:336-408)
//   RGBE  core/image/encoding/rgbe/rgbe_reader.zig: "#?" header, FORMAT=32-bit_rle_rgbe, "-Y h +X w", new-style RLE scanlines or
//         flat pixels; a pixel is (mantissa + 0.5) * 2^(e - 136) per channel (rgbeToVec4f), sRGB -> AP1, stored as f16
// EXR and IES files are not read (refused by name where a material asks for them).
#pragma once

#include <zlib.h>

#include <cmath>
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
    const size_t         row = size_t(width) * size_t(channels);
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

}  // namespace imgr
}  // namespace zyg
