// Image exporters behind su_export_frame (reference src/core/exporting/image_sequence.zig:24-53, take/take.zig:303-331):
//   PNG  8-bit sRGB, optional error diffusion   image/encoding/srgb.zig:34-283, png/png_writer.zig:33-59
//   EXR  half / float scan lines                 image/encoding/exr/exr_writer.zig:24-230
//   RGBE Radiance .hdr, run-length encoded       image/encoding/rgbe/rgbe_writer.zig:17-207
// The reference compresses through miniz (not vendored here); these writers emit the same pixels in the formats'
// uncompressed modes (PNG: stored deflate blocks, EXR: NO_COMPRESSION), which every reader accepts.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace zyg {
namespace imgw {

inline float linearToGamma(float x) {  // base/spectrum/srgb.zig:6-16
    if (x <= 0.f) return 0.f;
    if (x < 0.0031308f) return 12.92f * x;
    if (x < 1.f) return 1.055f * std::pow(x, 1.f / 2.4f) - 0.055f;
    return 1.f;
}

struct ImageView {  // a resolved Float4 image (driver.zig target) and the camera's crop window (x0, y0, x1, y1)
    const float* rgba;
    int          width, height;
    int          crop[4];
};

// Srgb.toSrgb, colour encodings (srgb.zig:34-260): bytes of the full image, zero outside the crop
inline std::vector<uint8_t> toSrgb(const ImageView& im, bool alpha, bool error_diffusion) {
    const int            nc = alpha ? 4 : 3;
    std::vector<uint8_t> out(size_t(im.width) * im.height * nc, 0);
    for (int y = im.crop[1]; y < im.crop[3]; ++y) {
        const float g = float(uint32_t(y)) * 0.618033988749894f;  // goldenRatio(y) - 0.5 (srgb.zig:280-282)
        float       err[4];
        for (float& e : err) e = (g - std::floor(g)) - 0.5f;
        for (int x = im.crop[0]; x < im.crop[2]; ++x) {
            const size_t i = size_t(y) * im.width + x;
            const float* p = im.rgba + i * 4;
            float        color[4] = {linearToGamma(p[0]), linearToGamma(p[1]), linearToGamma(p[2]), alpha ? std::fmin(p[3], 1.f) : 0.f};
            for (int k = 0; k < nc; ++k) {
                if (error_diffusion) {
                    const float   cf = 255.f * color[k];
                    const uint8_t ci = uint8_t(cf + err[k] + 0.5f);
                    err[k] += cf - float(ci);
                    out[i * nc + k] = ci;
                } else {
                    out[i * nc + k] = uint8_t(color[k] * 255.f + 0.5f);  // enc.floatToUnorm8
                }
            }
        }
    }
    return out;
}

// ---- PNG: signature, IHDR, one IDAT with a zlib stream of stored blocks, IEND
inline uint32_t crc32(const uint8_t* data, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool     init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
inline void be32(std::vector<uint8_t>& o, uint32_t v) {
    o.push_back(uint8_t(v >> 24));
    o.push_back(uint8_t(v >> 16));
    o.push_back(uint8_t(v >> 8));
    o.push_back(uint8_t(v));
}
inline void pngChunk(std::vector<uint8_t>& o, const char type[4], const std::vector<uint8_t>& data) {
    be32(o, uint32_t(data.size()));
    const size_t start = o.size();
    o.insert(o.end(), type, type + 4);
    o.insert(o.end(), data.begin(), data.end());
    be32(o, crc32(o.data() + start, o.size() - start));
}
inline std::vector<uint8_t> encodePng(const uint8_t* pixels, int width, int height, int num_channels) {
    std::vector<uint8_t> raw;  // filter byte 0 + the row
    raw.reserve(size_t(height) * (size_t(width) * num_channels + 1));
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), pixels + size_t(y) * width * num_channels, pixels + size_t(y + 1) * width * num_channels);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t             a = 1, b = 0;  // adler32
    for (uint8_t v : raw) {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    size_t at = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n));
        z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        at += n;
    } while (at < raw.size());
    be32(z, (b << 16) | a);

    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, uint32_t(width));
    be32(ihdr, uint32_t(height));
    ihdr.push_back(8);
    ihdr.push_back(1 == num_channels ? 0 : 3 == num_channels ? 2 : 6);  // grey / RGB / RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    pngChunk(o, "IHDR", ihdr);
    pngChunk(o, "IDAT", z);
    pngChunk(o, "IEND", {});
    return o;
}

// ---- EXR (exr_writer.zig:24-230): single-part scan-line file, channels in alphabetical order, no compression
inline uint16_t floatToHalf(float f) {  // @floatCast(f16): round to nearest even
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return uint16_t(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0u));
    if (x >= 0x477FF000u) return uint16_t(sign | 0x7C00u);  // rounds to infinity
    if (x < 0x38800000u) {                                  // subnormal half or zero
        if (x < 0x33000000u) return uint16_t(sign);
        const int      e = int(x >> 23);
        const uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int      shift = 126 - e;  // 14..24
        uint32_t       h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h += 1;
        return uint16_t(sign | h);
    }
    uint32_t       h = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h += 1;
    return uint16_t(sign | h);
}
inline void exrString(std::vector<uint8_t>& o, const char* s) { o.insert(o.end(), s, s + std::strlen(s) + 1); }
template <typename T>
inline void exrScalar(std::vector<uint8_t>& o, T v) {
    uint8_t b[sizeof(T)];
    std::memcpy(b, &v, sizeof(T));
    o.insert(o.end(), b, b + sizeof(T));
}
inline std::vector<uint8_t> encodeExr(const ImageView& im, bool alpha, bool half) {
    std::vector<uint8_t> o = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    const char*          names[4] = {"A", "B", "G", "R"};
    const int            first = alpha ? 0 : 1, nc = alpha ? 4 : 3;
    exrString(o, "channels");
    exrString(o, "chlist");
    exrScalar<uint32_t>(o, uint32_t(nc * (2 + 4 + 4 + 4 + 4) + 1));
    for (int c = first; c < 4; ++c) {
        exrString(o, names[c]);
        exrScalar<uint32_t>(o, half ? 1u : 2u);  // pixel type HALF / FLOAT
        exrScalar<uint32_t>(o, 0u);              // pLinear + reserved
        exrScalar<uint32_t>(o, 1u);
        exrScalar<uint32_t>(o, 1u);
    }
    o.push_back(0);
    exrString(o, "compression");
    exrString(o, "compression");
    exrScalar<uint32_t>(o, 1);
    o.push_back(0);  // NO_COMPRESSION
    exrString(o, "dataWindow");
    exrString(o, "box2i");
    exrScalar<uint32_t>(o, 16);
    exrScalar<int32_t>(o, im.crop[0]);
    exrScalar<int32_t>(o, im.crop[1]);
    exrScalar<int32_t>(o, im.crop[2] - 1);
    exrScalar<int32_t>(o, im.crop[3] - 1);
    exrString(o, "displayWindow");
    exrString(o, "box2i");
    exrScalar<uint32_t>(o, 16);
    exrScalar<int32_t>(o, 0);
    exrScalar<int32_t>(o, 0);
    exrScalar<int32_t>(o, im.width - 1);
    exrScalar<int32_t>(o, im.height - 1);
    exrString(o, "lineOrder");
    exrString(o, "lineOrder");
    exrScalar<uint32_t>(o, 1);
    o.push_back(0);
    exrString(o, "pixelAspectRatio");
    exrString(o, "float");
    exrScalar<uint32_t>(o, 4);
    exrScalar<float>(o, 1.f);
    exrString(o, "screenWindowCenter");
    exrString(o, "v2f");
    exrScalar<uint32_t>(o, 8);
    exrScalar<float>(o, 0.f);
    exrScalar<float>(o, 0.f);
    exrString(o, "screenWindowWidth");
    exrString(o, "float");
    exrScalar<uint32_t>(o, 4);
    exrScalar<float>(o, 1.f);
    o.push_back(0);

    const int      cw = im.crop[2] - im.crop[0], ch = im.crop[3] - im.crop[1];
    const size_t   bytes_per_value = half ? 2 : 4;
    const size_t   row_bytes = size_t(cw) * nc * bytes_per_value;
    const uint64_t table_start = o.size();
    uint64_t       offset = table_start + uint64_t(ch) * 8;
    for (int y = 0; y < ch; ++y) {
        exrScalar<uint64_t>(o, offset);
        offset += 8 + row_bytes;
    }
    const int channel_of[4] = {3, 2, 1, 0};  // A, B, G, R -> index into rgba
    for (int y = im.crop[1]; y < im.crop[3]; ++y) {
        exrScalar<int32_t>(o, y);
        exrScalar<uint32_t>(o, uint32_t(row_bytes));
        for (int c = first; c < 4; ++c) {
            for (int x = im.crop[0]; x < im.crop[2]; ++x) {
                const float v = im.rgba[(size_t(y) * im.width + x) * 4 + channel_of[c]];
                if (half) exrScalar<uint16_t>(o, floatToHalf(v));
                else exrScalar<float>(o, v);
            }
        }
    }
    return o;
}

// ---- RGBE (rgbe_writer.zig:17-207): header, then per scan line the four byte planes run-length encoded
inline void floatToRgbe(const float* c, uint8_t out[4]) {  // rgbe_writer.zig:189-206
    const float v = std::fmax(c[0], std::fmax(c[1], c[2]));
    if (v < 1e-32f) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    int         e;
    const float f = std::frexp(v, &e) * 256.f / v;
    out[0] = uint8_t(c[0] * f);
    out[1] = uint8_t(c[1] * f);
    out[2] = uint8_t(c[2] * f);
    out[3] = uint8_t(e + 128);
}
inline void rgbeRle(std::vector<uint8_t>& o, const uint8_t* data, uint32_t len) {
    const uint32_t min_run = 4;
    uint32_t       cur = 0;
    while (cur < len) {
        uint32_t begin_run = cur, run_count = 0, old_run_count = 0;
        while (run_count < min_run && begin_run < len) {
            begin_run += run_count;
            old_run_count = run_count;
            run_count = 1;
            while (begin_run + run_count < len && run_count < 127 && data[begin_run] == data[begin_run + run_count]) run_count += 1;
        }
        // a short run just before the long one is written as a run as well
        if (old_run_count > 1 && old_run_count == begin_run - cur) {
            o.push_back(uint8_t(128 + old_run_count));
            o.push_back(data[cur]);
            cur = begin_run;
        }
        while (cur < begin_run) {  // literal bytes up to the run
            const uint32_t n = std::min<uint32_t>(128, begin_run - cur);
            o.push_back(uint8_t(n));
            o.insert(o.end(), data + cur, data + cur + n);
            cur += n;
        }
        if (run_count >= min_run) {
            o.push_back(uint8_t(128 + run_count));
            o.push_back(data[begin_run]);
            cur += run_count;
        }
    }
}
inline std::vector<uint8_t> encodeRgbe(const ImageView& im) {
    std::vector<uint8_t> o;
    const std::string    header = "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(im.height) + " +X " + std::to_string(im.width) + "\n";
    o.insert(o.end(), header.begin(), header.end());
    const uint32_t w = uint32_t(im.width);
    const bool     rle = w >= 8 && w <= 0x7fff;
    std::vector<uint8_t> planes(size_t(w) * 4);
    for (int y = 0; y < im.height; ++y) {
        if (rle) {
            o.push_back(2);
            o.push_back(2);
            o.push_back(uint8_t(w >> 8));
            o.push_back(uint8_t(w & 0xFF));
        }
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t px[4] = {0, 0, 0, 0};
            if (y >= im.crop[1] && y < im.crop[3] && int(x) >= im.crop[0] && int(x) < im.crop[2]) {
                const float* p = im.rgba + (size_t(y) * w + x) * 4;
                const float  c[3] = {std::fmax(p[0], 0.f), std::fmax(p[1], 0.f), std::fmax(p[2], 0.f)};
                floatToRgbe(c, px);
            }
            if (rle) {
                for (int k = 0; k < 4; ++k) planes[x + size_t(w) * k] = px[k];
            } else {
                o.insert(o.end(), px, px + 4);
            }
        }
        if (rle)
            for (int k = 0; k < 4; ++k) rgbeRle(o, planes.data() + size_t(w) * k, w);
    }
    return o;
}

inline bool writeFile(const std::string& path, const std::vector<uint8_t>& bytes) {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes.size() == std::fwrite(bytes.data(), 1, bytes.size(), f);
    return 0 == std::fclose(f) && ok;
}

struct Exporter {  // take.zig:303-331 {"Image": {"format": "PNG" | "EXR" | "RGBE", "bitdepth": 16 | 32, "error_diffusion": b}}
    enum Format { PNG, EXR, RGBE } format = PNG;
    bool half = true, error_diffusion = false;
    const char* extension() const { return PNG == format ? "png" : EXR == format ? "exr" : "hdr"; }
};

// ImageSequence.write (image_sequence.zig:24-53): image_CC_FFFFFF.ext in `directory` (the reference writes to the cwd)
inline bool exportImage(const Exporter& e, const ImageView& im, bool alpha, uint32_t camera_id, uint32_t frame, const std::string& directory,
                        std::string* written = nullptr) {
    char name[64];
    std::snprintf(name, sizeof(name), "image_%02u_%06u.%s", camera_id, frame, e.extension());
    const std::string path = directory.empty() ? std::string(name) : directory + "/" + name;
    std::vector<uint8_t> bytes;
    if (Exporter::PNG == e.format) {
        const std::vector<uint8_t> px = toSrgb(im, alpha, e.error_diffusion);
        bytes = encodePng(px.data(), im.width, im.height, alpha ? 4 : 3);
    } else if (Exporter::EXR == e.format) {
        bytes = encodeExr(im, alpha, e.half);
    } else {
        bytes = encodeRgbe(im);  // take.zig:319-323: never with alpha
    }
    if (written) *written = path;
    return writeFile(path, bytes);
}

}  // namespace imgw
}  // namespace zyg
