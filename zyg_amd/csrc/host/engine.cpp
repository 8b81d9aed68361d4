#include "engine.hpp"

#include "ggx_lut.hpp"
#include "image_readers.hpp"
#include "sky_bake.hpp"

#include <algorithm>
#include <fstream>
#include <sstream>

namespace zyg {

static const float kMinRoughness = 0.01314f;  // material/ggx.zig:14-15

static Vec4f readVec4f3(const JsonValue& v) {  // base/json.zig:130-140
    if (JsonValue::Array == v.type && v.arr.size() >= 3) return vec(v.arr[0].f32(), v.arr[1].f32(), v.arr[2].f32(), 0.f);
    return splat(v.f32());
}

// base/json.zig:251-277 (array / number, or {"sRGB": [...]}); temperature + gamma forms are out of scope
static Vec4f readColor(const JsonValue& v) {
    if (JsonValue::Object == v.type) {
        Vec4f rgb = splat(0.f);
        if (const JsonValue* s = v.get("sRGB")) rgb = readVec4f3(*s);
        return sRGBtoAP1(rgb);
    }
    return sRGBtoAP1(readVec4f3(v));
}

// material_provider.zig:438-632 (TextureDescriptor) + 634-678 (readTextureSampler), image ids only: {"usage": "Color", "id": n,
// "sampler": {"filter": "Nearest" | "Linear", "address": "Clamp" | "Repeat" | [u, v]}, "scale": s | [sx, sy]}
enum TexUsage : uint32_t { kUsageColor = 0, kUsageEmission = 1, kUsageNormal = 2, kUsageOpacity = 3, kUsageWeight = 4 };
static bool readImageTexture(const JsonValue& v, uint32_t& image, uint32_t& sampler, float scale[2], uint32_t usage);
static bool readColorTexture(const JsonValue& v, ZygpuMaterial& m) {
    return readImageTexture(v, m.color_image, m.color_sampler, m.color_scale, kUsageColor);
}
// the engine whose material is being parsed (its mounts and image cache serve the "file" textures)
static thread_local Engine* t_engine = nullptr;

// Set while a material is parsed when its JSON asks for a texture feature that is not built (procedural textures, image
// files, swizzles, inversion, Triplanar / ObjectPos coordinates): the material gets ZYGPU_MATF_UNSUPPORTED and a scene that
// uses it does not compile, instead of rendering with a default in that place.
static thread_local bool t_unsupported_texture = false;
static thread_local int  t_procedural_depth = 0;

// readValue / readTypedValue (material_provider.zig:741-815) for a procedural's child: a number or an array is the uniform value of
// the usage's type, an object a texture (an image or another procedural)
static void readProceduralChild(const JsonValue& x, uint32_t usage, ZygpuTexture& t, float uniform[4]) {
    t = ZygpuTexture{ZYGPU_NULL, 0u, {1.f, 1.f}};
    uniform[0] = uniform[1] = uniform[2] = uniform[3] = 0.f;
    if (JsonValue::Object == x.type) {
        if (readImageTexture(x, t.image, t.sampler, t.scale, usage)) {
            if (kUsageNormal == usage && !(t.sampler & ZYGPU_TEX_PROCEDURAL)) t.sampler |= ZYGPU_TEX_SNORM;  // texture_provider.zig:70
        }
        return;  // (an object that names no texture: readImageTexture has flagged the material)
    }
    if (kUsageColor == usage) {
        const Vec4f c = readColor(x);
        uniform[0] = c[0], uniform[1] = c[1], uniform[2] = c[2];
    } else if (kUsageNormal == usage) {
        if (JsonValue::Array == x.type && x.arr.size() >= 2) uniform[0] = x.arr[0].f32(0.f), uniform[1] = x.arr[1].f32(0.f);
    } else {
        uniform[0] = x.f32(0.f);
    }
}

// TextureDescriptor.init's procedural entries (material_provider.zig:465-578) -> Engine::procedurals_; the entry's index or ZYGPU_NULL
static uint32_t readProcedural(const std::string& name, const JsonValue& body, uint32_t usage) {
    ZygpuProcedural pr;
    std::memset(&pr, 0, sizeof(pr));
    for (ZygpuTexture& t : pr.child) t = ZygpuTexture{ZYGPU_NULL, 0u, {1.f, 1.f}};
    if (JsonValue::Object != body.type || !t_engine) return ZYGPU_NULL;
    if (++t_procedural_depth > ZYGPU_PROCEDURAL_MAX_DEPTH) {  // (the device evaluates a bounded nesting without recursion)
        --t_procedural_depth;
        t_unsupported_texture = true;
        return ZYGPU_NULL;
    }
    auto child = [&](const char* key, uint32_t k, uint32_t child_usage) {
        if (const JsonValue* c = body.get(key)) readProceduralChild(*c, child_usage, pr.child[k], pr.child_uniform[k]);
    };
    if (name == "ChannelMix") {
        pr.kind = ZYGPU_PROC_CHANNEL_MIX;
        child("source", 0, usage);
        if (const JsonValue* ch = body.get("channels"))
            if (JsonValue::Array == ch->type && ch->arr.size() >= 3)
                for (int k = 0; k < 3; ++k) {
                    const Vec4f row = readVec4f3(ch->arr[size_t(k)]);
                    for (int i = 0; i < 3; ++i) pr.value[k][i] = row[i];
                }
    } else if (name == "Checker") {  // procedural_checker.zig:20-41
        pr.kind = ZYGPU_PROC_CHECKER;
        pr.checker_scale = body.memberF32("scale", 1.f);
        if (const JsonValue* cs = body.get("colors"))
            if (JsonValue::Array == cs->type && cs->arr.size() >= 2)
                for (int k = 0; k < 2; ++k) {
                    const Vec4f c = readColor(cs->arr[size_t(k)]);
                    for (int i = 0; i < 3; ++i) pr.value[k][i] = c[i];
                }
    } else if (name == "DetailNormal") {
        pr.kind = ZYGPU_PROC_DETAIL_NORMAL;
        child("base", 0, usage);
        child("detail", 1, usage);
    } else if (name == "Max" || name == "Mul") {
        pr.kind = name == "Max" ? ZYGPU_PROC_MAX : ZYGPU_PROC_MUL;
        child("a", 0, usage);
        child("b", 1, usage);
    } else if (name == "Mix") {
        pr.kind = ZYGPU_PROC_MIX;
        child("a", 0, usage);
        child("b", 1, usage);
        child("weight", 2, kUsageOpacity);  // material_provider.zig:536
    } else {  // Noise (material_provider.zig:558-578)
        pr.kind = ZYGPU_PROC_NOISE;
        const JsonValue* cl = body.get("class");
        if (cl && JsonValue::String == cl->type && cl->str == "Cellular") pr.flags |= ZYGPU_NOISE_CELLULAR;
        if (body.memberBool("absolute", false)) pr.flags |= ZYGPU_NOISE_ABSOLUTE;
        if (body.memberBool("invert", false)) pr.flags |= ZYGPU_NOISE_INVERT;
        pr.levels = body.memberU32("levels", 1);
        pr.attenuation = body.memberF32("attenuation", 0.f);
        pr.ratio = body.memberF32("ratio", 0.5f);
        pr.transition = body.memberF32("transition", 0.5f);
        const JsonValue* sc = body.get("scale");
        const JsonValue* pe = body.get("period");
        const Vec4f      scale = sc ? readVec4f3(*sc) : splat(1.f), period = pe ? readVec4f3(*pe) : splat(0.f);
        for (int i = 0; i < 3; ++i) pr.scale[i] = scale[i], pr.period[i] = period[i];
    }
    --t_procedural_depth;
    return t_engine->appendProcedural(pr);
}

static bool proceduralName(const std::string& k) {
    return k == "ChannelMix" || k == "Checker" || k == "DetailNormal" || k == "Max" || k == "Mix" || k == "Mul" || k == "Noise";
}

// material_provider.zig:438-632: {"usage": .., "id": n, "sampler": {"filter", "address"}, "scale"}
static bool readImageTexture(const JsonValue& v, uint32_t& image, uint32_t& sampler_out, float scale_out[2], uint32_t usage) {
    if (JsonValue::Object != v.type) return false;
    const JsonValue* id = v.get("id");
    const JsonValue* file = v.get("file");
    uint32_t         procedural = ZYGPU_NULL;
    bool             is_procedural = false;
    for (const auto& kv : v.obj) {
        const std::string& k = kv.first;
        // "swizzle" and "invert" are options of the image-file loader: with an image id the reference ignores them, too
        // (createTexture, material_provider.zig:700-736)
        if (k == "id" || k == "file" || k == "usage" || k == "scale" || k == "swizzle" || k == "invert") continue;
        if (k == "sampler") continue;
        if (proceduralName(k) && (kUsageColor == usage || kUsageWeight == usage || kUsageNormal == usage || (kUsageOpacity == usage && t_procedural_depth > 0))) {
            // (the last one named wins: every entry overwrites desc.procedural; masks and emission maps take images only)
            procedural = readProcedural(k, kv.second, usage);
            is_procedural = true;
            continue;
        }
        t_unsupported_texture = true;  // (a key the reference does not know, or a procedural where only images are evaluated)
    }
    struct { uint32_t color_image, color_sampler; float color_scale[2]; } m;
    if (is_procedural) {  // a procedural wins over "file" and "id" (createTexture, material_provider.zig:712-713)
        if (ZYGPU_NULL == procedural) {
            t_unsupported_texture = true;
            return false;
        }
        m.color_image = procedural;
    } else
    if (file && JsonValue::String == file->type && t_engine) {  // "file" wins over "id" (createTexture, material_provider.zig:715-733)
        const JsonValue* sw = v.get("swizzle");
        m.color_image = t_engine->loadImageFile(file->str, usage, sw && JsonValue::String == sw->type ? sw->str : std::string(), v.memberBool("invert", false));
        if (ZYGPU_NULL == m.color_image) {  // the reference logs and takes a uniform default; here the scene does not compile
            t_unsupported_texture = true;
            return false;
        }
    } else if (id) {
        m.color_image = uint32_t(id->f32(0.f));
    } else {
        t_unsupported_texture = true;
        return false;
    }
    m.color_sampler = ZYGPU_TEX_LINEAR | ZYGPU_TEX_REPEAT_U | ZYGPU_TEX_REPEAT_V;  // texture.zig:22-27 DefaultMode
    m.color_scale[0] = m.color_scale[1] = 1.f;
    auto address = [](const JsonValue& a) { return JsonValue::String == a.type && a.str == "Clamp" ? 0u : 1u; };
    if (const JsonValue* smp = v.get("sampler")) {
        if (JsonValue::Object == smp->type) {
            for (const auto& kv : smp->obj) {
                if (kv.first == "filter" && JsonValue::String == kv.second.type) {
                    if (kv.second.str == "Nearest") m.color_sampler &= ~uint32_t(ZYGPU_TEX_LINEAR);
                    else if (kv.second.str == "Linear") m.color_sampler |= ZYGPU_TEX_LINEAR;
                } else if (kv.first == "address") {
                    uint32_t au, av;
                    if (JsonValue::Array == kv.second.type && kv.second.arr.size() >= 2) {
                        au = address(kv.second.arr[0]);
                        av = address(kv.second.arr[1]);
                    } else {
                        au = av = address(kv.second);
                    }
                    m.color_sampler = (m.color_sampler & ~uint32_t(ZYGPU_TEX_REPEAT_U | ZYGPU_TEX_REPEAT_V)) | (au ? ZYGPU_TEX_REPEAT_U : 0u) |
                                      (av ? ZYGPU_TEX_REPEAT_V : 0u);
                } else if (kv.first == "texcoord" && JsonValue::String == kv.second.type) {  // material_provider.zig:664-674
                    m.color_sampler &= ~uint32_t(ZYGPU_TEX_TRIPLANAR | ZYGPU_TEX_OBJECTPOS);
                    if (kv.second.str == "Triplanar") m.color_sampler |= ZYGPU_TEX_TRIPLANAR;
                    else if (kv.second.str == "ObjectPos") m.color_sampler |= ZYGPU_TEX_OBJECTPOS;
                }
            }
        }
    }
    if (const JsonValue* sc = v.get("scale")) {
        if (JsonValue::Array == sc->type && sc->arr.size() >= 2) {
            m.color_scale[0] = sc->arr[0].f32(1.f);
            m.color_scale[1] = sc->arr[1].f32(1.f);
        } else {
            m.color_scale[0] = m.color_scale[1] = sc->f32(1.f);
        }
    }
    if (is_procedural) {  // Texture.initProcedural (texture.zig:91-93): id, data and the mode; no image scale
        m.color_sampler |= ZYGPU_TEX_PROCEDURAL;
        m.color_scale[0] = m.color_scale[1] = 1.f;
    }
    image = m.color_image;
    sampler_out = m.color_sampler;
    scale_out[0] = m.color_scale[0];
    scale_out[1] = m.color_scale[1];
    return true;
}

// math.Transformation.lerp (transformation.zig:36-44) with quaternion.slerp (quaternion.zig:217-245); host side: bounds only
static Transformation lerpTransformation(const Transformation& a, const Transformation& b, float t) {
    Transformation r;
    r.position = splat(1.f - t) * a.position + splat(t) * b.position;
    r.scale = splat(1.f - t) * a.scale + splat(t) * b.scale;
    Vec4f end = b.rotation;
    float cosom = a.rotation[0] * end[0] + a.rotation[1] * end[1] + a.rotation[2] * end[2] + a.rotation[3] * end[3];
    if (cosom < 0.f) {
        cosom = -cosom;
        end = splat(-1.f) * end;
    }
    float sclp = 1.f - t, sclq = t;
    if (1.f - cosom > 0.00001f) {
        const float omega = std::acos(cosom), sinom = std::sin(omega);
        sclp = std::sin((1.f - t) * omega) / sinom;
        sclq = std::sin(t * omega) / sinom;
    }
    r.rotation = splat(sclp) * a.rotation + splat(sclq) * end;
    return r;
}

static ZygpuTrafo composedTrafo(const Transformation& t, const Vec4f& origin) {  // composed_transformation.zig:17-29
    const Mat3x3 r = quaternionToMat3x3(t.rotation);
    ZygpuTrafo   out;
    for (int i = 0; i < 3; ++i) {
        out.r[i][0] = r.r[i][0];
        out.r[i][1] = r.r[i][1];
        out.r[i][2] = r.r[i][2];
        out.r[i][3] = t.scale[i];
    }
    const Vec4f p = t.position - origin;  // space.zig:101-105 translate(-origin)
    for (int i = 0; i < 4; ++i) out.position[i] = p[i];
    return out;
}

static Mat4x4 objectToWorld(const Transformation& t) {  // composed_transformation.zig:47-49
    const Mat3x3 r = quaternionToMat3x3(t.rotation);
    return compose(r, vec(t.scale[0], t.scale[1], t.scale[2], 1.f), t.position);
}

struct Mitchell {  // sensor.zig:43-58
    float b, c;
    float eval(float x) const {
        const float xx = x * x;
        if (x > 1.f) {
            return ((-b - 6.f * c) * xx * x + (6.f * b + 30.f * c) * xx + (-12.f * b - 48.f * c) * x + (8.f * b + 24.f * c)) / 6.f;
        }
        return ((12.f - 9.f * b - 6.f * c) * xx * x + (-18.f + 12.f * b + 6.f * c) * xx + (6.f - 2.f * b)) / 6.f;
    }
};

struct Blackman {  // sensor.zig:27-41
    float r;
    float eval(float x) const {
        const float a0 = 0.35875f, a1 = 0.48829f, a2 = 0.14128f, a3 = 0.01168f;
        const float b = (float(M_PI) * (x + r)) / r;
        return a0 - a1 * std::cos(b) + a2 * std::cos(2.f * b) - a3 * std::cos(3.f * b);
    }
};

static float filterEval(const SensorSettings& s, float x) {  // interpolated_function.zig:154-166
    const float    inverse_interval = 1.f / (s.filter_radius / float(ZYGPU_FILTER_TAPS - 1));
    const float    cx = fmin_(std::fabs(x), s.filter_radius);
    const float    o = cx * inverse_interval;
    const uint32_t offset = uint32_t(o);
    const float    t = o - float(offset);
    const float    a = s.filter_samples[offset];
    const float    b = s.filter_samples[std::min<uint32_t>(offset + 1, ZYGPU_FILTER_TAPS - 1)];
    return std::fmaf(1.f - t, a, t * b);
}

void Engine::setFilter(float radius, int kind) {  // sensor.zig:105-120, 630-645
    sensor.filter_radius = radius;
    sensor.filter_radius_int = int32_t(std::ceil(radius));
    std::fill(sensor.filter_samples, sensor.filter_samples + ZYGPU_FILTER_TAPS + 1, 0.f);
    if (!(radius > 0.f)) return;

    const float interval = radius / float(ZYGPU_FILTER_TAPS - 1);
    float       s = 0.f;
    for (int i = 0; i < ZYGPU_FILTER_TAPS; ++i) {
        sensor.filter_samples[i] = 0 == kind ? Mitchell{1.f / 3.f, 1.f / 3.f}.eval(s) : Blackman{radius}.eval(s);
        s += interval;
    }

    const uint32_t num_samples = 64;
    const float    iv = radius / float(num_samples);
    float          x = 0.5f * iv, sum = 0.f;
    for (uint32_t i = 0; i < num_samples; ++i) {
        sum += filterEval(sensor, x) * iv;
        x += iv;
    }
    const float scale = 1.f / (sum + sum);
    for (int i = 0; i < ZYGPU_FILTER_TAPS; ++i) sensor.filter_samples[i] *= scale;
}

Engine::Engine() {
    shapes_.resize(7);  // resource/manager.zig:36-44
    for (uint32_t i = 0; i < 7; ++i) shapes_[i].kind = i;

    specular_threshold = kMinRoughness * kMinRoughness;  // resource/manager.zig:57

    // Resources.MaterialProvider.createFallbackMaterial (material_provider.zig:127-129)
    ZygpuMaterial fb{};
    parseMaterial(fb, JsonValue(), ZYGPU_MAT_DEBUG);
    commitMaterial(fb);
    materials_.push_back(fb);
    fallback_material_ = 0;

    setFilter(2.f, 0);  // take.zig:59-64: Opaque + Mitchell(1/3, 1/3), radius 2
}

// measured optical constants (eta, k) at RGB of the named metals, and fresnel.conductor at wo.h = 1
static Vec4f conductorColor(const std::string& name) {
    struct Preset { const char* name; float eta[3], k[3]; };
    static const Preset presets[] = {
        {"Aluminium", {1.50694f, 0.926041f, 0.68251f}, {7.6307f, 6.3849f, 5.6230f}},
        {"Gold", {0.18267f, 0.49447f, 1.3761f}, {3.1178f, 2.3515f, 1.8324f}},
        {"Silver", {0.13708f, 0.12945f, 0.14075f}, {4.0625f, 3.1692f, 2.6034f}},
        {"Iron", {2.9038f, 2.8857f, 2.6420f}, {3.0798f, 2.9157f, 2.8025f}},
    };
    Vec4f eta = splat(1.5f), k = splat(1.f);
    for (const Preset& p : presets)
        if (name == p.name) {
            eta = vec(p.eta[0], p.eta[1], p.eta[2], 0.f);
            k = vec(p.k[0], p.k[1], p.k[2], 0.f);
        }
    const float wo_dot_h = 1.f;
    const Vec4f tmp_f = eta * eta + k * k;
    const Vec4f wo_dot_h2 = splat(wo_dot_h * wo_dot_h);
    const Vec4f tmp = wo_dot_h2 * tmp_f;
    const Vec4f a = splat(2.f * wo_dot_h) * eta;
    const Vec4f r_p = (tmp - a + splat(1.f)) / (tmp + a + splat(1.f));
    const Vec4f r_o = (tmp_f - a + wo_dot_h2) / (tmp_f + a + wo_dot_h2);
    return splat(0.5f) * (r_p + r_o);
}

// material_provider.zig:131-162, 262-342, 411-436 (uniform values only; image textures are SURVEY §8(f))
// init = false: su_material_update (material_provider.zig:108-125: the same update* function a new material goes through, on top
// of the material as it is)
std::string Engine::resolveFile(const std::string& name) const {  // file/system.zig:73-120: the newest mount first, then the name as it is
    if (!name.empty() && '/' == name[0]) return name;
    for (auto it = mounts.rbegin(); it != mounts.rend(); ++it) {
        const std::string p = it->empty() ? name : *it + "/" + name;
        if (std::ifstream(p)) return p;
    }
    return name;
}

uint32_t Engine::loadImageFile(const std::string& name, uint32_t usage, const std::string& swizzle_in, bool invert) {
    // a file's sizes are bounded by the readers, but an allocation can still fail: no exception leaves through the extern "C" entries
    try {
        return loadImageFileUnguarded(name, usage, swizzle_in, invert);
    } catch (const std::exception& e) {
        return file_error = "\"" + name + "\": " + e.what(), ZYGPU_NULL;
    }
}

uint32_t Engine::loadImageFileUnguarded(const std::string& name, uint32_t usage, const std::string& swizzle_in, bool invert) {
    // texture_provider.zig:42-51: the usage's default swizzle
    static const char* const kDefault[5] = {"XYZ", "XYZ", "XY", "W", "X"};
    const std::string swizzle = swizzle_in.empty() ? std::string(kDefault[std::min<uint32_t>(usage, 4u)]) : swizzle_in;
    const std::string path = resolveFile(name);
    const std::string key = path + "|" + swizzle + (invert ? "|i" : "|");
    const auto        cached = image_files_.find(key);
    if (image_files_.end() != cached) return cached->second;
    std::ifstream f(path, std::ios::binary);
    if (!f) return file_error = "cannot read the image \"" + name + "\"", ZYGPU_NULL;
    std::ostringstream ss;
    ss << f.rdbuf();
    const std::string data = ss.str();
    uint32_t          image = ZYGPU_NULL;
    int               w = 0, h = 0, channels = 0;
    std::string       error;
    if (data.size() >= 8 && 0 == std::memcmp(data.data(), "\x89PNG", 4)) {
        std::vector<uint8_t> px;
        if (!imgr::readPng(data, w, h, channels, px, error)) return file_error = "\"" + name + "\": " + error, ZYGPU_NULL;
        // Info.allocateImage / fillImage (png_reader.zig:145-331): as many channels as the swizzle names, at most the file's
        const uint32_t wanted = uint32_t(swizzle.size());
        if (wanted > 3 && channels > 3) return file_error = "\"" + name + "\": four-channel images (usage ColorAndOpacity) are not provided", ZYGPU_NULL;
        const uint32_t       out_channels = std::min<uint32_t>(std::min<uint32_t>(wanted, 3u), uint32_t(channels));
        std::vector<uint8_t> out(size_t(w) * size_t(h) * out_channels);
        for (size_t i = 0, n = size_t(w) * size_t(h); i < n; ++i) {
            const uint8_t* in = px.data() + i * size_t(channels);
            if (1 == out_channels) {
                uint32_t c = swizzle == "Y" ? 1u : swizzle == "Z" ? 2u : swizzle == "W" ? 3u : 0u;
                if (c >= uint32_t(channels)) c = 0;
                out[i] = invert ? uint8_t(255 - in[c]) : in[c];
            } else if (2 == out_channels) {
                const bool yx = swizzle == "YX", yz = swizzle == "YZ" && channels >= 3;
                out[2 * i] = in[(yx || yz) ? 1 : 0];
                out[2 * i + 1] = in[yx ? 0 : yz ? 2 : 1];
            } else {
                out[3 * i] = in[0], out[3 * i + 1] = in[1], out[3 * i + 2] = in[2];
            }
        }
        image = createImage(ZYGPU_NULL, 0, out_channels, uint32_t(w), uint32_t(h), 1, out_channels, out.data());
    } else if (data.size() >= 2 && 0 == data.compare(0, 2, "#?")) {
        std::vector<float> rgb;
        if (!imgr::readRgbe(data, w, h, rgb, error)) return file_error = "\"" + name + "\": " + error, ZYGPU_NULL;
        image = createImage(ZYGPU_NULL, 4, 3, uint32_t(w), uint32_t(h), 1, 12, reinterpret_cast<const uint8_t*>(rgb.data()));
    } else if (data.size() >= 4 && 0 == std::memcmp(data.data(), "\x76\x2f\x31\x01", 4)) {
        std::vector<float> values;
        uint32_t           out_channels = 0;
        const bool         color = kUsageColor == usage || kUsageEmission == usage;  // texture_provider.zig:37-40
        if (!imgr::readExr(data, uint32_t(swizzle.size()), color, w, h, out_channels, values, error)) return file_error = "\"" + name + "\": " + error, ZYGPU_NULL;
        image = createImage(ZYGPU_NULL, 4, out_channels, uint32_t(w), uint32_t(h), 1, 4 * out_channels, reinterpret_cast<const uint8_t*>(values.data()));
    } else if (data.size() >= 3 && 0 == data.compare(0, 3, "IES")) {  // file.zig:34-35 -> ies_reader.zig: a Half1 image, whatever the usage
        std::vector<float> values;
        if (!imgr::readIes(data, w, h, values, error)) return file_error = "\"" + name + "\": " + error, ZYGPU_NULL;
        image = createImage(ZYGPU_NULL, 4, 1, uint32_t(w), uint32_t(h), 1, 4, reinterpret_cast<const uint8_t*>(values.data()));
    } else {
        return file_error = "\"" + name + "\": an image type this build does not read (PNG, Radiance RGBE, ZIP-compressed OpenEXR and IES are)", ZYGPU_NULL;
    }
    if (ZYGPU_NULL == image) return file_error = "\"" + name + "\": the image could not be created", ZYGPU_NULL;
    image_files_[key] = image;
    return image;
}

void Engine::parseMaterial(ZygpuMaterial& m, const JsonValue& v, uint32_t kind, bool init) {
    t_unsupported_texture = false;
    t_engine = this;
    parseMaterialBody(m, v, kind, init);
    t_engine = nullptr;
    if (t_unsupported_texture) m.flags |= ZYGPU_MATF_UNSUPPORTED;
}

void Engine::parseMaterialBody(ZygpuMaterial& m, const JsonValue& v, uint32_t kind, bool init) {
    if (init) {
        std::memset(&m, 0, sizeof(m));
        m.kind = kind;
        m.cos_a = -1.f;
        m.camera_weight = 1.f;
        m.emission_num_samples = 1;
        m.color_image = ZYGPU_NULL;
        m.emission_image = ZYGPU_NULL;
        m.mask_image = ZYGPU_NULL;
        m.profile_image = ZYGPU_NULL;
        m.roughness = 0.8f;
        m.metallic = 0.f;
        m.specular = 1.f;
        m.ior = 1.f;
        for (ZygpuTexture& t : m.tex) t = ZygpuTexture{ZYGPU_NULL, 0u, {1.f, 1.f}};
        m.coating_scale = 1.f;  // substitute_material.zig:37-60
        m.coating_roughness = 0.2f;
        m.coating_ior = 1.5f;
        m.flakes_color[0] = m.flakes_color[1] = m.flakes_color[2] = 0.8f;
        m.flakes_alpha = 0.01f;
    }
    auto setTwoSided = [&](const JsonValue& x) {  // Base.setTwoSided (material_base.zig): set or cleared
        if (x.b) m.flags |= ZYGPU_MATF_TWO_SIDED;
        else m.flags &= ~uint32_t(ZYGPU_MATF_TWO_SIDED);
    };

    // readValue (material_provider.zig:634-660): a number is the uniform value, an object names an image
    auto readWeight = [&](const JsonValue& x, float& uniform, uint32_t slot) {
        ZygpuTexture& t = m.tex[slot];
        t.image = ZYGPU_NULL;
        if (!readImageTexture(x, t.image, t.sampler, t.scale, kUsageWeight)) uniform = x.f32(uniform);
    };
    auto readNormal = [&](const JsonValue& x, uint32_t slot) {  // usage Normal: Byte2 is snorm (texture_provider.zig:70)
        ZygpuTexture& t = m.tex[slot];
        t.image = ZYGPU_NULL;
        if (readImageTexture(x, t.image, t.sampler, t.scale, kUsageNormal)) t.sampler |= ZYGPU_TEX_SNORM;
    };

    auto loadEmittance = [&](const JsonValue& e) {
        Vec4f color = splat(1.f);
        if (const JsonValue* s = e.get("spectrum")) color = readColor(*s);
        const float value = e.memberF32("value", 1.f);
        const Vec4f em = splat(value) * color;
        m.emission[0] = em[0];
        m.emission[1] = em[1];
        m.emission[2] = em[2];
        m.emission_normalize = e.memberBool("normalize", false) ? 1u : 0u;
        // "profile" (material_provider.zig:412-414): a texture of usage Emission, in practice an IES file (image_provider.zig:48)
        m.profile_image = ZYGPU_NULL;
        if (const JsonValue* pv = e.get("profile")) readImageTexture(*pv, m.profile_image, m.profile_sampler, m.profile_scale, kUsageEmission);
        // angleFromProfile (emittance.zig:73-103): without a profile pi; with one the widest angle from the light's axis at which a texel
        // of the octahedral map is lit, plus a texel
        float profile_angle = float(M_PI);
        if (ZYGPU_NULL != m.profile_image && m.profile_image < images_.size() && ZYGPU_IMAGE_FLOAT1 == images_[m.profile_image].format) {
            const HostImage& img = images_[m.profile_image];
            const float*     px = reinterpret_cast<const float*>(img.pixels.data());
            const float      idf0 = 1.f / float(img.width), idf1 = 1.f / float(img.height);
            float            cos_a = 1.f;
            for (int32_t y = 0; y < img.height; ++y) {
                const float vv = idf1 * (float(y) + 0.5f);
                for (int32_t x = 0; x < img.width; ++x) {
                    if (!(px[size_t(y) * size_t(img.width) + size_t(x)] > 0.f)) continue;
                    const float u = idf0 * (float(x) + 0.5f);
                    Vec4f       d = vec(2.f * (u - 0.5f), 2.f * (vv - 0.5f), 0.f, 0.f);  // octDecode (base/encoding.zig:89-98)
                    d[2] = -1.f + std::fabs(d[0]) + std::fabs(d[1]);
                    const float tt = fmax_(d[2], 0.f);
                    d[0] += d[0] > 0.f ? -tt : tt;
                    d[1] += d[1] > 0.f ? -tt : tt;
                    d = normalize3(d);
                    cos_a = fmin_(cos_a, -d[2]);
                }
            }
            profile_angle = std::acos(fmax_(cos_a - fmax_(idf0, idf1), -1.f));
        } else if (ZYGPU_NULL != m.profile_image) {
            t_unsupported_texture = true;  // (a profile that is not a one-channel float image: only IES files are read as profiles)
        }
        m.cos_a = std::cos(e.memberF32("angle", profile_angle * (180.f / float(M_PI))) * (float(M_PI) / 180.f));
        m.camera_weight = e.memberF32("camera_weight", 1.f);
        m.emission_num_samples = std::min<uint32_t>(e.memberU32("num_samples", 1), 64u);
        if (const JsonValue* em = e.get("emission_map")) {
            readImageTexture(*em, m.emission_image, m.emission_sampler, m.emission_scale, kUsageEmission);
            // light samples carry no position for a Triplanar lookup (light.zig:119-132 leaves rs.p unset): uv only
            if (m.emission_sampler & ZYGPU_TEX_TRIPLANAR) t_unsupported_texture = true;
            m.emission_sampler &= ~uint32_t(ZYGPU_TEX_OBJECTPOS);
        }
        // black-body temperature maps: not built
        if (e.get("temperature_map")) t_unsupported_texture = true;
    };

    // material_provider.zig:177-178, 244-245, 265-266: Glass, Light and Substitute read "mask" (usage Opacity)
    if (ZYGPU_MAT_DEBUG != kind)
        if (const JsonValue* mask = v.get("mask")) {
            readImageTexture(*mask, m.mask_image, m.mask_sampler, m.mask_scale, kUsageOpacity);
            m.mask_sampler &= ~uint32_t(ZYGPU_TEX_TRIPLANAR | ZYGPU_TEX_OBJECTPOS);  // Base.opacity takes the uv (material_base.zig:35-41)
        }

    if (ZYGPU_MAT_DEBUG == kind) {
        // debug_material.zig:11-18: two-sided, fixed front/back albedo.
        // Extension (SURVEY §8(d) cfg2 "Lambert with per-wall albedo"): {"Debug": {"color": [r,g,b]}}.
        m.flags |= ZYGPU_MATF_TWO_SIDED;
        const float front[3] = {0.4f, 0.9f, 0.1f}, back[3] = {0.9f, 0.1f, 0.4f};
        for (int i = 0; init && i < 3; ++i) {
            m.color_a[i] = front[i];
            m.color_b[i] = back[i];
        }
        if (const JsonValue* c = v.get("color")) {
            const Vec4f col = readColor(*c);
            for (int i = 0; i < 3; ++i) m.color_a[i] = m.color_b[i] = col[i];
        }
    } else if (ZYGPU_MAT_LIGHT == kind) {
        if (init) m.emission[0] = m.emission[1] = m.emission[2] = 1.f;  // light_material.zig:43
        if (JsonValue::Object == v.type) {
            for (const auto& kv : v.obj) {
                if (kv.first == "emittance")
                    loadEmittance(kv.second);
                else if (kv.first == "two_sided")
                    setTwoSided(kv.second);
            }
        }
    } else if (ZYGPU_MAT_SUBSTITUTE == kind) {
        // substitute_material.zig:32-60 defaults
        const Vec4f half = splat(0.5f);
        for (int i = 0; init && i < 3; ++i) m.color_a[i] = half[i];
        if (init) m.ior = 1.46f;
        if (JsonValue::Object == v.type) {
            for (const auto& kv : v.obj) {
                const std::string& k = kv.first;
                const JsonValue&   x = kv.second;
                if (k == "color" || k == "subsurface_color") {  // material_provider.zig:267
                    if (!readColorTexture(x, m)) {
                        const Vec4f col = readColor(x);
                        for (int i = 0; i < 3; ++i) m.color_a[i] = col[i];
                    }
                } else if (k == "roughness") {
                    readWeight(x, m.roughness, ZYGPU_STEX_ROUGHNESS);
                } else if (k == "metallic") {
                    readWeight(x, m.metallic, ZYGPU_STEX_METALLIC);
                } else if (k == "normal") {
                    readNormal(x, ZYGPU_STEX_NORMAL);
                } else if (k == "anisotropy_rotation") {
                    readWeight(x, m.rotation, ZYGPU_STEX_ROTATION);
                } else if (k == "translucency") {
                    readWeight(x, m.translucency, ZYGPU_STEX_TRANSLUCENCY);
                } else if (k == "attenuation_distance") {
                    m.attenuation_distance = x.f32(0.f);
                } else if (k == "coating" && JsonValue::Object == x.type) {  // material_provider.zig:303-326
                    Vec4f coating_color = splat(1.f);
                    float coating_attenuation_distance = 0.1f;
                    for (const auto& c : x.obj) {
                        if (c.first == "color") coating_color = readColor(c.second);
                        else if (c.first == "attenuation_distance") coating_attenuation_distance = c.second.f32(coating_attenuation_distance);
                        else if (c.first == "ior") m.coating_ior = c.second.f32(m.coating_ior);
                        else if (c.first == "normal") readNormal(c.second, ZYGPU_STEX_COATING_NORMAL);
                        else if (c.first == "roughness") readWeight(c.second, m.coating_roughness, ZYGPU_STEX_COATING_ROUGHNESS);
                        else if (c.first == "scale") readWeight(c.second, m.coating_scale, ZYGPU_STEX_COATING_SCALE);
                        else if (c.first == "thickness") m.coating_thickness = c.second.f32(0.f);
                    }
                    // setCoatingAttenuation -> ccoef.attenuationCoefficient (collision_coefficients.zig:35-44)
                    for (int i = 0; i < 3; ++i) {
                        const float ca = std::max(std::min(coating_color[i], 0.991102f), 0.01f);
                        m.coating_absorption[i] = 0.f == coating_attenuation_distance ? 0.f : -std::log(ca) / coating_attenuation_distance;
                    }
                } else if (k == "flakes" && JsonValue::Object == x.type) {  // material_provider.zig:327-340
                    for (const auto& c : x.obj) {
                        if (c.first == "color") {
                            const Vec4f col = readColor(c.second);
                            for (int i = 0; i < 3; ++i) m.flakes_color[i] = col[i];
                        } else if (c.first == "coverage") {
                            readWeight(c.second, m.flakes_coverage, ZYGPU_STEX_FLAKES_COVERAGE);
                        } else if (c.first == "roughness") {  // setFlakesRoughness (substitute_material.zig:85-88)
                            const float r = std::max(c.second.f32(0.f), 0.01314f);  // ggx.clampRoughness (ggx.zig:14, 48-50)
                            m.flakes_alpha = r * r;
                        } else if (c.first == "size") {  // setFlakesSize (substitute_material.zig:90-95)
                            const float size = c.second.f32(0.f);
                            const float N = 1.5396f / (size * size);
                            m.flakes_res = std::max(4.f, std::ceil(std::sqrt(N / 4.f)));
                        }
                    }
                } else if (k == "metal_preset") {
                    // material_provider.zig:277-280: a conductor is Substitute with metallic = 1 and the normal-incidence
                    // Fresnel reflectance of the metal's (eta, k) as colour (fresnel.zig:45-57, metal_presets.zig:6-39)
                    const Vec4f f = conductorColor(JsonValue::String == x.type ? x.str : std::string());
                    for (int i = 0; i < 3; ++i) m.color_a[i] = f[i];
                    m.metallic = 1.f;
                } else if (k == "specular") {
                    readWeight(x, m.specular, ZYGPU_STEX_SPECULAR);
                } else if (k == "ior") {
                    m.ior = x.f32(m.ior);
                } else if (k == "anisotropy") {
                    m.anisotropy = x.f32(0.f);
                } else if (k == "two_sided") {
                    setTwoSided(x);
                } else if (k == "emittance") {
                    loadEmittance(x);
                } else if (k == "priority") {
                    m.priority = int32_t(x.f32(0.f));
                }
            }
        }
    } else if (ZYGPU_MAT_GLASS == kind) {
        // glass_material.zig:19-33 defaults; material_provider.zig:172-201.  An update that names no colour recomputes the absorption
        // from white, like updateGlass' local `attenuation_color` does
        if (init) {
            m.roughness = 0.f;
            m.specular = 1.f;
            m.ior = 1.46f;
            m.attenuation_distance = 1.f;
        }
        Vec4f attenuation_color = splat(1.f);
        float attenuation_distance = m.attenuation_distance;
        if (JsonValue::Object == v.type) {
            for (const auto& kv : v.obj) {
                const std::string& k = kv.first;
                const JsonValue&   x = kv.second;
                if (k == "color" || k == "attenuation_color") attenuation_color = readColor(x);
                else if (k == "attenuation_distance") attenuation_distance = x.f32(attenuation_distance);
                else if (k == "roughness") readWeight(x, m.roughness, ZYGPU_STEX_ROUGHNESS);  // material_provider.zig:179-188
                else if (k == "specular") readWeight(x, m.specular, ZYGPU_STEX_SPECULAR);
                else if (k == "normal") readNormal(x, ZYGPU_STEX_NORMAL);
                else if (k == "priority") m.priority = int32_t(x.f32(0.f));
                else if (k == "ior") m.ior = x.f32(m.ior);
                else if (k == "abbe") m.abbe = x.f32(0.f);
                else if (k == "thickness") m.thickness = x.f32(0.f);
            }
        }
        // setVolumetric -> ccoef.attenuationCoefficient (collision_coefficients.zig:35-44)
        for (int i = 0; i < 3; ++i) {
            m.color_a[i] = attenuation_color[i];
            const float ca = std::max(std::min(attenuation_color[i], 0.991102f), 0.01f);
            m.absorption[i] = 0.f == attenuation_distance ? 0.f : -std::log(ca) / attenuation_distance;
        }
        m.attenuation_distance = attenuation_distance;
    }
}

uint32_t Engine::appendProcedural(const ZygpuProcedural& p) {
    procedurals_.push_back(p);
    return uint32_t(procedurals_.size() - 1);
}

void Engine::commitMaterial(ZygpuMaterial& m) {
    {  // a procedural colour or weight: evaluated per vertex ahead of the closure (wf_textures), which then needs the complete form
        bool procedural = ZYGPU_NULL != m.color_image && (m.color_sampler & ZYGPU_TEX_PROCEDURAL);
        for (const ZygpuTexture& t : m.tex) procedural = procedural || (ZYGPU_NULL != t.image && (t.sampler & ZYGPU_TEX_PROCEDURAL));
        if (procedural && (ZYGPU_MAT_SUBSTITUTE == m.kind || ZYGPU_MAT_GLASS == m.kind)) m.flags |= ZYGPU_MATF_PROCEDURAL;
        else m.flags &= ~uint32_t(ZYGPU_MATF_PROCEDURAL);
    }
    // light_material.zig:45-51 / substitute_material.zig:62-75
    const bool masked = ZYGPU_NULL != m.mask_image;  // mask.isImage()
    if (ZYGPU_MAT_LIGHT == m.kind || ZYGPU_MAT_SUBSTITUTE == m.kind) {
        if (m.emission[0] > 0.f || m.emission[1] > 0.f || m.emission[2] > 0.f) m.flags |= ZYGPU_MATF_EMISSIVE;
        else m.flags &= ~uint32_t(ZYGPU_MATF_EMISSIVE);
        if (masked) m.flags |= ZYGPU_MATF_EVALUATE_VISIBILITY;  // light_material.zig:47, substitute_material.zig:65
        else m.flags &= ~uint32_t(ZYGPU_MATF_EVALUATE_VISIBILITY);
    }
    if (ZYGPU_MAT_SUBSTITUTE == m.kind) {
        // properties.caustic = roughness.isUniform() and roughness <= specular_threshold (substitute_material.zig:69)
        if (ZYGPU_NULL == m.tex[ZYGPU_STEX_ROUGHNESS].image && m.roughness <= specular_threshold) m.flags |= ZYGPU_MATF_CAUSTIC;
        else m.flags &= ~uint32_t(ZYGPU_MATF_CAUSTIC);
        bool extended = m.rotation > 0.f || m.translucency > 0.f || m.coating_thickness > 0.f || m.flakes_coverage > 0.f;
        for (const ZygpuTexture& t : m.tex) extended = extended || ZYGPU_NULL != t.image;
        // a Triplanar colour map reads the fragment's position, like the complete closure
        extended = extended || (ZYGPU_NULL != m.color_image && (m.color_sampler & (ZYGPU_TEX_TRIPLANAR | ZYGPU_TEX_PROCEDURAL)));
        if (extended) m.flags |= ZYGPU_MATF_EXTENDED;
        else m.flags &= ~uint32_t(ZYGPU_MATF_EXTENDED);
    }
    if (ZYGPU_MAT_GLASS == m.kind) {  // glass_material.zig:35-42
        const bool thin = m.thickness > 0.f;
        m.flags &= ~uint32_t(ZYGPU_MATF_TWO_SIDED | ZYGPU_MATF_EVALUATE_VISIBILITY | ZYGPU_MATF_CAUSTIC);
        if (thin) m.flags |= ZYGPU_MATF_TWO_SIDED;
        if (thin || masked) m.flags |= ZYGPU_MATF_EVALUATE_VISIBILITY;
        // properties.caustic = roughness.isUniform() and roughness^2 <= specular_threshold (glass_material.zig:41)
        if (ZYGPU_NULL == m.tex[ZYGPU_STEX_ROUGHNESS].image && m.roughness * m.roughness <= specular_threshold) m.flags |= ZYGPU_MATF_CAUSTIC;
        // a roughness / specular / normal map: the closure needs the fragment's uv
        if (ZYGPU_NULL != m.tex[ZYGPU_STEX_ROUGHNESS].image || ZYGPU_NULL != m.tex[ZYGPU_STEX_SPECULAR].image ||
            ZYGPU_NULL != m.tex[ZYGPU_STEX_NORMAL].image)
            m.flags |= ZYGPU_MATF_EXTENDED;
        else m.flags &= ~uint32_t(ZYGPU_MATF_EXTENDED);
    }
}

static bool materialKind(const JsonValue& v, uint32_t& kind, const JsonValue*& body) {
    const JsonValue* rendering = v.get("rendering");
    if (!rendering || JsonValue::Object != rendering->type) return false;
    for (const auto& kv : rendering->obj) {  // first recognised entry wins (material_provider.zig:136-160)
        body = &kv.second;
        if (kv.first == "Debug") { kind = ZYGPU_MAT_DEBUG; return true; }
        if (kv.first == "Light") { kind = ZYGPU_MAT_LIGHT; return true; }
        if (kv.first == "Substitute") { kind = ZYGPU_MAT_SUBSTITUTE; return true; }
        if (kv.first == "Glass") { kind = ZYGPU_MAT_GLASS; return true; }
    }
    return false;  // Hair / Volumetric / Sky: out of scope
}

uint32_t Engine::createMaterial(uint32_t id, const JsonValue& v) {
    uint32_t         kind;
    const JsonValue* body = nullptr;
    if (!materialKind(v, kind, body)) return ZYGPU_NULL;
    ZygpuMaterial m;
    parseMaterial(m, *body, kind);
    commitMaterial(m);
    if (id < materials_.size()) {  // resource/cache.zig store(): overwrite an existing id
        materials_[id] = m;
        return id;
    }
    materials_.push_back(m);
    return uint32_t(materials_.size() - 1);
}

int Engine::updateMaterial(uint32_t id, const JsonValue& v) {
    if (id >= materials_.size()) return -3;  // capi.zig:360-362
    ZygpuMaterial& m = materials_[id];
    // capi.zig:366-371 passes the *inner* object to updateMaterial (e.g. {"roughness": 0.2}): the kind's own parser runs on top of
    // the material as it is (material_provider.zig:108-125), so everything a description can say can be updated
    if (JsonValue::Object != v.type) return -4;
    parseMaterial(m, v, m.kind, false);
    commitMaterial(m);
    return 0;
}

// shape_provider.zig:274-306 (loadData) + 847-913 (buildDescAsync) + triangle_tree_builder.zig:33-65,166-206
uint32_t Engine::createMesh(uint32_t id, uint32_t num_parts_in, const uint32_t* parts_in, uint32_t num_triangles,
                            const uint32_t* indices, uint32_t num_vertices, const float* positions,
                            uint32_t positions_stride, const float* normals, uint32_t normals_stride, const float* uvs,
                            uint32_t uvs_stride) {
    (void)id;
    HostMesh mesh;

    // Client and file data (su_triangle_mesh_create, .sub / JSON meshes) is not trusted: every index must name a vertex, part
    // ids are stored as u16 (triangle_data.zig:22-29), a part's index range must lie inside the index buffer, and a part's
    // material slot must be small enough to size the prop's material list.  The reference reads such a mesh out of bounds.
    const uint32_t num_parts = num_parts_in > 0 ? num_parts_in : 1;
    if (0 == num_triangles || 0 == num_vertices || num_parts > 65535u || !positions || !normals) return ZYGPU_NULL;
    if (num_triangles > 0x55555555u) return ZYGPU_NULL;  // 3 * num_triangles must fit 32 bits
    if (indices) {
        for (size_t i = 0, n = size_t(num_triangles) * 3; i < n; ++i)
            if (indices[i] >= num_vertices) return ZYGPU_NULL;
    } else if (size_t(num_triangles) * 3 > num_vertices) {
        return ZYGPU_NULL;
    }
    if (num_parts_in > 0 && parts_in) {
        for (uint32_t i = 0; i < num_parts; ++i) {
            const uint64_t start = parts_in[i * 3 + 0], count = parts_in[i * 3 + 1];
            if (start + count > uint64_t(num_triangles) * 3 || parts_in[i * 3 + 2] > 65535u) return ZYGPU_NULL;
        }
    }
    mesh.part_materials.assign(num_parts, 0);
    if (num_parts_in > 0 && parts_in) {
        for (uint32_t i = 0; i < num_parts; ++i) mesh.part_materials[i] = parts_in[i * 3 + 2];
    }

    struct IndexTriangle {
        uint32_t i[3];
        uint32_t part;
    };
    std::vector<IndexTriangle> triangles(num_triangles, IndexTriangle{{0, 0, 0}, 0});

    const uint32_t  empty_part[3] = {0, num_triangles * 3, 0};
    const uint32_t* parts = (num_parts_in > 0 && parts_in) ? parts_in : empty_part;
    for (uint32_t p = 0; p < num_parts; ++p) {
        const uint32_t start_index = parts[p * 3 + 0];
        const uint32_t num_indices = parts[p * 3 + 1];
        const uint32_t tb = start_index / 3, te = std::min((start_index + num_indices) / 3, num_triangles);
        for (uint32_t i = tb; i < te; ++i) {
            const uint32_t t = i * 3;
            if (indices) {
                triangles[i].i[0] = indices[t + 0];
                triangles[i].i[1] = indices[t + 1];
                triangles[i].i[2] = indices[t + 2];
            } else {
                triangles[i].i[0] = t + 0;
                triangles[i].i[1] = t + 1;
                triangles[i].i[2] = t + 2;
            }
            triangles[i].part = p;
        }
    }

    // vertex_buffer.zig:214-245 (CAPI.copy)
    mesh.num_vertices = num_vertices;
    mesh.positions.assign(size_t(num_vertices) * 3 + 1, 0.f);
    mesh.normals.assign(size_t(num_vertices) * 2, 0);
    mesh.uvs.assign(size_t(num_vertices) * 2, 0.f);
    for (uint32_t i = 0; i < num_vertices; ++i) {
        const size_t s = size_t(i) * positions_stride;
        mesh.positions[3 * i + 0] = positions[s + 0];
        mesh.positions[3 * i + 1] = positions[s + 1];
        mesh.positions[3 * i + 2] = positions[s + 2];
    }
    for (uint32_t i = 0; i < num_vertices; ++i) {
        const size_t s = size_t(i) * normals_stride;
        float        o[2];
        octEncode(vec(normals[s + 0], normals[s + 1], normals[s + 2], 0.f), o);
        mesh.normals[2 * i + 0] = floatToSnorm16(o[0]);
        mesh.normals[2 * i + 1] = floatToSnorm16(o[1]);
    }
    if (uvs) {
        for (uint32_t i = 0; i < num_vertices; ++i) {
            const size_t s = size_t(i) * uvs_stride;
            mesh.uvs[2 * i + 0] = uvs[s + 0];
            mesh.uvs[2 * i + 1] = uvs[s + 1];
        }
    }

    // references (triangle_tree_builder.zig:113-135)
    std::vector<BvhReference> refs(num_triangles);
    AABB                      bounds = emptyAabb();
    for (uint32_t r = 0; r < num_triangles; ++r) {
        const Vec4f a = mesh.position(triangles[r].i[0]);
        const Vec4f b = mesh.position(triangles[r].i[1]);
        const Vec4f c = mesh.position(triangles[r].i[2]);
        const Vec4f mn = min4(a, min4(b, c));
        const Vec4f mx = max4(a, max4(b, c));
        for (int k = 0; k < 3; ++k) {
            refs[r].mn[k] = mn[k];
            refs[r].mx[k] = mx[k];
        }
        refs[r].prim = r;
        bounds.b[0] = min4(bounds.b[0], mn);
        bounds.b[1] = max4(bounds.b[1], mx);
    }

    BvhBuilder            builder(BvhSettings{16, 64, 4});  // shape_provider.zig:922
    std::vector<uint32_t> prim_ids;
    builder.build(std::move(refs), bounds, mesh.nodes, prim_ids);

    mesh.triangles.resize(prim_ids.size() * 3);
    mesh.triangle_parts.resize(prim_ids.size());
    for (size_t i = 0; i < prim_ids.size(); ++i) {
        const IndexTriangle& t = triangles[prim_ids[i]];
        mesh.triangles[3 * i + 0] = t.i[0];
        mesh.triangles[3 * i + 1] = t.i[1];
        mesh.triangles[3 * i + 2] = t.i[2];
        mesh.triangle_parts[i] = uint16_t(t.part);
    }

    // triangle_mesh.zig:705-720 (calculateAreas; duplicated triangles count twice, as in the reference)
    mesh.part_areas.assign(num_parts, 0.f);
    for (size_t t = 0; t < prim_ids.size(); ++t) {
        const Vec4f a = mesh.position(mesh.triangles[3 * t + 0]);
        const Vec4f b = mesh.position(mesh.triangles[3 * t + 1]);
        const Vec4f c = mesh.position(mesh.triangles[3 * t + 2]);
        mesh.part_areas[mesh.triangle_parts[t]] += 0.5f * length3(cross3(b - a, c - a));
    }

    meshes_.push_back(std::move(mesh));
    HostShape s;
    s.kind = ZYGPU_SHAPE_TRIANGLE_MESH;
    s.mesh = uint32_t(meshes_.size() - 1);
    shapes_.push_back(s);
    return uint32_t(shapes_.size() - 1);
}

uint32_t Engine::createImage(uint32_t id, uint32_t format, uint32_t num_channels, uint32_t width, uint32_t height, uint32_t depth,
                             uint32_t pixel_stride, const uint8_t* data) {
    (void)id;  // ids are assigned in creation order, like the other resources created with id = NULL
    const bool u8 = 0 == format, f32 = 4 == format;  // capi.zig:31-37 Format: UInt8, UInt16, UInt32, Float16, Float32
    if (!(u8 || f32) || num_channels < 1 || num_channels > 3 || 1 != depth || 0 == width || 0 == height || !data) return ZYGPU_NULL;
    const uint32_t bpp = (u8 ? 1u : 4u) * num_channels;
    if (bpp != pixel_stride) return ZYGPU_NULL;  // the reference only copies tightly packed pixels (capi.zig:264-266)
    HostImage img;
    img.format = 3 == num_channels   ? (u8 ? ZYGPU_IMAGE_BYTE3_SRGB : ZYGPU_IMAGE_FLOAT3)
                 : 2 == num_channels ? (u8 ? ZYGPU_IMAGE_BYTE2 : ZYGPU_IMAGE_FLOAT2)
                                     : (u8 ? ZYGPU_IMAGE_BYTE1_UNORM : ZYGPU_IMAGE_FLOAT1);
    img.width = int32_t(width);
    img.height = int32_t(height);
    img.pixels.assign(data, data + size_t(width) * height * bpp);
    images_.push_back(std::move(img));
    return uint32_t(images_.size() - 1);
}

bool Engine::shapeFinite(uint32_t shape) const {  // shape.zig:92-97
    const uint32_t k = shapes_[shape].kind;
    return !(ZYGPU_SHAPE_CANOPY == k || ZYGPU_SHAPE_DISTANT == k || ZYGPU_SHAPE_DOME == k);
}

AABB Engine::shapeAabb(uint32_t shape) const {  // shape.zig:106-113
    const HostShape& s = shapes_[shape];
    switch (s.kind) {
        case ZYGPU_SHAPE_CANOPY:
        case ZYGPU_SHAPE_DISTANT:
        case ZYGPU_SHAPE_DOME: return emptyAabb();
        case ZYGPU_SHAPE_DISK:
        case ZYGPU_SHAPE_RECTANGLE: return {{vec(-0.5f, -0.5f, 0.f, 0.f), vec(0.5f, 0.5f, 0.f, 0.f)}};
        case ZYGPU_SHAPE_CUBE:
        case ZYGPU_SHAPE_SPHERE: return {{splat(-0.5f), splat(0.5f)}};
        default: return meshes_[s.mesh].aabb();
    }
}

uint32_t Engine::shapeNumParts(uint32_t shape) const {
    const HostShape& s = shapes_[shape];
    return ZYGPU_SHAPE_TRIANGLE_MESH == s.kind ? meshes_[s.mesh].numParts() : 1;
}

uint32_t Engine::createEntity() {  // scene.zig:259-265: a Distant-shaped prop without materials, not classified
    HostProp p;
    p.shape = ZYGPU_SHAPE_DISTANT;
    // configureShape with no materials: not mono -> solid, not unoccluding
    props_.push_back(p);
    Transformation t;
    t.position = splat(0.f);
    t.scale = vec(1.f, 1.f, 1.f, 0.f);
    t.rotation = vec(0.f, 0.f, 0.f, 1.f);
    world_transformations_.push_back(t);
    return uint32_t(props_.size() - 1);
}

void Engine::classifyProp(uint32_t p) {  // scene.zig:324-342
    const HostProp& po = props_[p];
    if (po.flags & ZYGPU_PROP_VOLUME) volume_props_.push_back(p);
    if (po.flags & ZYGPU_PROP_SOLID) {
        if (shapeFinite(po.shape)) {
            if (po.flags & ZYGPU_PROP_UNOCCLUDING) unoccluding_props_.push_back(p);
            else finite_props_.push_back(p);
        } else {
            infinite_props_.push_back(p);
        }
    }
}

// scene.zig:267-293 + prop.zig:96-135
int32_t Engine::createPropShape(uint32_t shape, const uint32_t* materials, uint32_t num_materials, bool unocc) {
    if (shape >= shapes_.size()) return -1;

    HostProp p;
    p.shape = shape;

    bool mono = false, pure_emissive = true;
    if (num_materials > 0) {
        mono = true;
        const uint32_t mid0 = materials[0];
        for (uint32_t i = 0; i < num_materials; ++i) {
            const ZygpuMaterial& m = materials_[materials[i]];
            if (m.flags & ZYGPU_MATF_EVALUATE_VISIBILITY) p.flags |= ZYGPU_PROP_EVALUATE_VISIBILITY;
            if (m.flags & ZYGPU_MATF_CAUSTIC) p.flags |= ZYGPU_PROP_CAUSTIC;
            if (ZYGPU_MAT_LIGHT != m.kind && ZYGPU_MAT_SKY != m.kind) pure_emissive = false;
            if (materials[i] != mid0) mono = false;
        }
    }

    auto materialIor = [&](uint32_t mid) {  // material.zig:140-146
        const ZygpuMaterial& m = materials_[mid];
        return (ZYGPU_MAT_SUBSTITUTE == m.kind || ZYGPU_MAT_GLASS == m.kind) ? m.ior : 1.f;
    };
    const bool finite = shapeFinite(shape);
    const bool volumetric = finite && mono && materialIor(materials[0]) < 1.f;
    if (volumetric) {
        p.flags &= ~uint32_t(ZYGPU_PROP_SOLID);
        p.flags |= ZYGPU_PROP_VOLUME;
    }
    if (unocc && finite && pure_emissive) p.flags |= ZYGPU_PROP_UNOCCLUDING;

    p.parts_start = uint32_t(material_ids_.size());
    const uint32_t   num_parts = shapeNumParts(shape);
    const HostShape& s = shapes_[shape];
    for (uint32_t i = 0; i < num_parts; ++i) {
        const uint32_t part_mat = ZYGPU_SHAPE_TRIANGLE_MESH == s.kind ? meshes_[s.mesh].part_materials[i] : i;
        uint32_t       material_id = 0;
        if (num_materials > 0) material_id = part_mat < num_materials ? materials[part_mat] : fallback_material_;
        material_ids_.push_back(material_id);
        light_ids_.push_back(ZYGPU_NULL);
    }

    props_.push_back(p);
    Transformation t;
    t.position = splat(0.f);
    t.scale = vec(1.f, 1.f, 1.f, 0.f);
    t.rotation = vec(0.f, 0.f, 0.f, 1.f);
    world_transformations_.push_back(t);

    const uint32_t id = uint32_t(props_.size() - 1);
    classifyProp(id);
    return int32_t(id);
}

int32_t Engine::createPropInstance(uint32_t entity) {  // scene.zig:307-316
    if (entity >= props_.size()) return -1;
    props_.push_back(props_[entity]);
    world_transformations_.push_back(world_transformations_[entity]);
    const uint32_t id = uint32_t(props_.size() - 1);
    classifyProp(id);
    return int32_t(id);
}

int32_t Engine::createPortal(uint32_t shape, uint32_t light_id) {  // scene_loader.zig:371-397
    if (light_id >= lights_.size() || shape >= shapes_.size()) return -1;  // Error.UndefinedLight
    const ZygpuLight& l = lights_[light_id];
    setVisibility(l.prop, false, false, false, false);
    lights_[light_id].prototype = 1;  // scene.lightSetPrototype
    const uint32_t        material = material_ids_[props_[l.prop].parts_start + l.part];
    std::vector<uint32_t> materials(shapeNumMaterials(shape), material);
    return createPropShape(shape, materials.data(), uint32_t(materials.size()), true);
}

int Engine::createLight(uint32_t entity, uint32_t light_link) {  // scene.zig:344-372
    if (entity >= props_.size()) return -1;
    const HostProp& p = props_[entity];
    const uint32_t  num_parts = shapeNumParts(p.shape);
    for (uint32_t i = 0; i < num_parts; ++i) {
        const ZygpuMaterial& m = materials_[material_ids_[p.parts_start + i]];
        if (!(m.flags & ZYGPU_MATF_EMISSIVE)) continue;
        ZygpuLight l{};
        l.prop = entity;
        l.part = i;
        l.num_samples = 1;
        l.light_class = ZYGPU_LIGHT_PROP;
        l.two_sided = (m.flags & ZYGPU_MATF_TWO_SIDED) ? 1 : 0;
        l.shadow_catcher_light = (p.flags & ZYGPU_PROP_SHADOW_CATCHER_LIGHT) ? 1 : 0;
        l.prototype = 0;
        lights_.push_back(l);
        light_links_.resize(lights_.size(), ZYGPU_NULL);
        light_links_.back() = light_link;
    }
    return 0;
}

int Engine::setTransformation(uint32_t prop, const float* m16) {  // capi.zig:482-503
    if (prop >= props_.size()) return -1;
    Mat4x4 m;
    for (int r = 0; r < 4; ++r) m.r[r] = vec(m16[4 * r + 0], m16[4 * r + 1], m16[4 * r + 2], m16[4 * r + 3]);
    Mat3x3         basis;
    Transformation t;
    decompose(m, basis, t.scale, t.position);
    t.rotation = quaternionFromMat3x3(basis);
    world_transformations_[prop] = t;
    return 0;
}

// Base.setShutter (camera_base.zig:43-118): the shutter function at the centres of 63 bins (a ramp or a cubic Bezier segment while it
// opens and closes, 1 between), and Distribution1DN.configure (distribution_1d.zig:167-186) over it
void Engine::setShutter(float open, float close, const float* slope, uint32_t num_slope) {
    constexpr int N = 63;
    auto lerp = [](float a, float b, float t) { return std::fmaf(1.f - t, a, t * b); };  // util.zig:3-8
    struct P2 {
        float x, y;
    };
    auto lerp2 = [&](P2 a, P2 b, float t) { return P2{lerp(a.x, b.x, t), lerp(a.y, b.y, t)}; };
    auto bezier = [&](const P2 (&cp)[4], float u) {  // cubicBezierEvaluate
        const P2 a = lerp2(cp[0], cp[1], u), b = lerp2(cp[1], cp[2], u), c = lerp2(cp[2], cp[3], u);
        return lerp2(lerp2(a, b, u), lerp2(b, c, u), u);
    };
    auto search = [&](const P2 (&cp)[4], float x) {  // searchBezier: 16 bisection steps on the curve's parameter
        float u = 0.5f, step = u * 0.5f;
        P2    c{0.f, 0.f};
        for (int i = 0; i < 16; ++i) {
            c = bezier(cp, u);
            if (x < c.x) u -= step;
            else if (x > c.x) u += step;
            else break;
            step *= 0.5f;
        }
        return c.y;
    };
    float function[N], integral = 0.f;
    for (int i = 0; i < N; ++i) {
        const float t = (float(i) + 0.5f) / float(N);
        float       f = 1.f;
        if (t < open) {
            if (num_slope >= 4) {
                const P2 cp[4] = {{0.f, 0.f}, {slope[0], slope[1]}, {slope[2], slope[3]}, {open, 1.f}};
                f = search(cp, t);
            } else {
                f = lerp(0.f, 1.f, t / open);
            }
        } else if (t > close) {
            if (num_slope >= 8) {
                const P2 cp[4] = {{close, 1.f}, {slope[4], slope[5]}, {slope[6], slope[7]}, {1.f, 0.f}};
                f = search(cp, t);
            } else {
                f = lerp(1.f, 0.f, (t - close) / (1.f - close));
            }
        }
        function[i] = f;
        integral += f;
    }
    const float ii = 1.f / integral;
    float       p = 0.f;
    shutter_cdf[0] = 0.f;
    for (int i = 0; i < N - 1; ++i) {
        p = std::fmaf(function[i], ii, p);
        shutter_cdf[i + 1] = p;
    }
    shutter_cdf[N] = 1.f;
    shutter_configured = true;  // (Distribution1DN.integral >= 0)
}

uint32_t Engine::numInterpolationFrames() const {  // scene.zig:254-257 + countFrames (724-734)
    const uint64_t tick = 705600000ull / 60;
    auto matching = [](uint64_t a, uint64_t b) { return 0 == (a > b ? a % b : (0 == a ? 0 : b % a)); };
    const uint32_t a = std::max<uint32_t>(uint32_t(frame_duration / tick), 1u);
    const uint32_t b = matching(frame_step, tick) ? 0u : 1u, c = matching(frame_duration, tick) ? 0u : 1u;
    return a + b + c + 1;
}

int Engine::setTransformationFrame(uint32_t prop, uint32_t frame, const float* m16) {  // capi.zig:506-533
    if (prop >= props_.size() || frame >= numInterpolationFrames() || !m16) return -1;
    Mat4x4 m;
    for (int r = 0; r < 4; ++r) m.r[r] = vec(m16[4 * r + 0], m16[4 * r + 1], m16[4 * r + 2], m16[4 * r + 3]);
    Mat3x3         basis;
    Transformation t;
    decompose(m, basis, t.scale, t.position);
    t.rotation = quaternionFromMat3x3(basis);
    if (keyframes_.size() < props_.size()) keyframes_.resize(props_.size());
    // propAllocateFrames (scene.zig:317-322) leaves new keyframes uninitialised; here they start as the static transformation
    if (keyframes_[prop].empty()) keyframes_[prop].assign(numInterpolationFrames(), world_transformations_[prop]);
    keyframes_[prop][frame] = t;
    return 0;
}

int Engine::createAnimation(const std::vector<uint64_t>& times, const std::vector<Transformation>& frames) {
    if (times.empty() || times.size() != frames.size()) return -1;
    animations_.push_back({times, frames});
    return int(animations_.size() - 1);
}

int Engine::createGraphNode(uint32_t prop, int parent, int animation, const Transformation& local) {
    if ((ZYGPU_NULL != prop && prop >= props_.size()) || parent >= int(graph_nodes_.size()) || animation >= int(animations_.size())) return -1;
    if (parent < 0 && animation < 0) return -1;  // nothing moves it
    graph_nodes_.push_back({prop, parent, animation, local, {}});
    return int(graph_nodes_.size() - 1);
}

namespace {
// util/animation.zig:96-141
float catmullT(Vec4f p0, Vec4f p1, float t, float alpha) {  // getT
    const Vec4f d = p1 - p0;
    const float a = dot3(d, d);
    const float b = std::pow(a, alpha * 0.5f);
    return b + t;
}
Vec4f catmullRom(Vec4f p0, Vec4f p1, Vec4f p2, Vec4f p3, float t, float alpha) {  // centripetal Catmull-Rom spline
    const float t0 = 0.f;
    const float t1 = catmullT(p0, p1, t0, alpha);
    const float t2 = catmullT(p1, p2, t1, alpha);
    const float t3 = catmullT(p2, p3, t2, alpha);
    const float tt = std::fmaf(1.f - t, t1, t * t2);  // math.lerp (base/math/util.zig:3-8)
    if (0.f == t1) return p1;
    const Vec4f A1 = splat((t1 - tt) / (t1 - t0)) * p0 + splat((tt - t0) / (t1 - t0)) * p1;
    const Vec4f A2 = splat((t2 - tt) / (t2 - t1)) * p1 + splat((tt - t1) / (t2 - t1)) * p2;
    const Vec4f A3 = splat((t3 - tt) / (t3 - t2)) * p2 + splat((tt - t2) / (t3 - t2)) * p3;
    const Vec4f B1 = splat((t2 - tt) / (t2 - t0)) * A1 + splat((tt - t0) / (t2 - t0)) * A2;
    const Vec4f B2 = splat((t3 - tt) / (t3 - t1)) * A2 + splat((tt - t1) / (t3 - t1)) * A3;
    return splat((t2 - tt) / (t2 - t1)) * B1 + splat((tt - t1) / (t2 - t1)) * B2;
}
Vec4f slerp(Vec4f a, Vec4f b, float t) {  // quaternion.slerp (quaternion.zig:217-245)
    const Vec4f ab = a * b;
    float       cosom = ab[0] + ab[1] + ab[2] + ab[3];
    Vec4f       end = b;
    if (cosom < 0.f) {
        cosom = -cosom;
        end = splat(0.f) - end;
    }
    float sclp, sclq;
    if (1.f - cosom > 0.00001f) {
        const float omega = std::acos(cosom), sinom = std::sin(omega);
        sclp = std::sin((1.f - t) * omega) / sinom;
        sclq = std::sin(t * omega) / sinom;
    } else {
        sclp = 1.f - t;
        sclq = t;
    }
    return splat(sclp) * a + splat(sclq) * end;
}
Vec4f lerp4(Vec4f a, Vec4f b, float t) {  // math.lerp on vectors: mulAdd(1 - t, a, t * b)
    Vec4f r;
    for (int i = 0; i < 4; ++i) r[i] = std::fmaf(1.f - t, a[i], t * b[i]);
    return r;
}
Transformation extrapolate(const Transformation& f1, const Transformation& f2) {
    return {f2.position + (f2.position - f1.position), f2.scale + (f2.scale - f1.scale), f1.rotation};
}
Transformation interpolate(const Transformation& f0, const Transformation& f1, const Transformation& f2, const Transformation& f3, float t) {
    return {catmullRom(f0.position, f1.position, f2.position, f3.position, t, 0.5f), lerp4(f1.scale, f2.scale, t), slerp(f1.rotation, f2.rotation, t)};
}
}  // namespace

// Graph.simulate (scene_graph.zig:106-118) -> Animation.resample (animation.zig:48-93) -> Animation.update: the frame's ticks, from
// the tick at or before its start to the tick at or after its end; a time outside the keyframes' span takes the last keyframe, as
// the reference's search does
void Engine::simulate(uint32_t frame) {
    if (graph_nodes_.empty()) return;
    const uint64_t tick = 705600000ull / 60;
    const uint64_t start = uint64_t(frame) * frame_step, end = start + frame_duration;
    const uint64_t frames_start = start - (start % tick);
    const uint64_t end_rem = end % tick;
    const uint64_t frames_end = end + (end_rem > 0 ? tick - end_rem : 0);
    const uint32_t nif = numInterpolationFrames();
    std::vector<std::vector<Transformation>> resampled(animations_.size());
    for (size_t k = 0; k < animations_.size(); ++k) {
        const HostAnimation&         a = animations_[k];
        std::vector<Transformation>& out = resampled[k];
        out.assign(nif, a.frames.back());
        const uint32_t frames_back = uint32_t(a.frames.size()) - 1u;
        uint32_t       last_frame = 0, i = 0;
        for (uint64_t time = frames_start; time <= frames_end && i < nif; time += tick, ++i) {
            for (uint32_t j = last_frame; j < frames_back; ++j) {
                const uint64_t        a_time = a.times[j], b_time = a.times[j + 1];
                const Transformation &f1 = a.frames[j], &f2 = a.frames[j + 1];
                if (time >= a_time && time < b_time) {
                    const Transformation f0 = j > 0 ? a.frames[j - 1] : extrapolate(f2, f1);
                    const Transformation f3 = j + 1 < frames_back ? a.frames[j + 2] : extrapolate(f1, f2);
                    const float          t = float(double(time - a_time) / double(b_time - a_time));
                    out[i] = interpolate(f0, f1, f2, f3, t);
                    break;
                }
                if (j + 1 == frames_back) {
                    out[i] = f2;
                    break;
                }
                last_frame += 1;
            }
        }
    }
    // calculateWorldTransformations / propInheritTransformations (scene_graph.zig:250-310): parents before children
    if (keyframes_.size() < props_.size()) keyframes_.resize(props_.size());
    for (GraphNode& n : graph_nodes_) {
        n.frames.resize(nif);
        for (uint32_t i = 0; i < nif; ++i) {
            const Transformation& own = n.animation >= 0 ? resampled[size_t(n.animation)][i] : n.local;
            n.frames[i] = n.parent >= 0 ? transformBy(graph_nodes_[size_t(n.parent)].frames[i], own) : own;
        }
        if (ZYGPU_NULL != n.prop) keyframes_[n.prop] = n.frames;
    }
}

int Engine::updateImage(uint32_t id, uint32_t pixel_stride, const uint8_t* data) {  // capi.zig:300-338
    if (id >= images_.size() || !data) return -1;
    HostImage&     img = images_[id];
    static const uint32_t kBpp[6] = {3u, 12u, 1u, 4u, 2u, 8u};  // by ZygpuImageFormat
    const uint32_t        bpp = kBpp[img.format];
    // a stride that does not match the image's pixel size copies nothing and still succeeds (capi.zig:316-334)
    if (bpp == pixel_stride) std::memcpy(img.pixels.data(), data, img.pixels.size());
    return 0;
}

int Engine::setWorldTransformation(uint32_t prop, const Transformation& t) {
    if (prop >= props_.size()) return -1;
    world_transformations_[prop] = t;
    return 0;
}

int Engine::setVisibility(uint32_t prop, bool in_camera, bool in_reflection, bool in_sss, bool shadow_catcher_light) {  // prop.zig:75-88
    if (prop >= props_.size()) return -1;
    uint32_t& f = props_[prop].flags;
    auto      set = [&](uint32_t bit, bool on) { f = on ? (f | bit) : (f & ~bit); };
    set(ZYGPU_PROP_VISIBLE_IN_CAMERA, in_camera);
    set(ZYGPU_PROP_VISIBLE_IN_REFLECTION, in_reflection);
    set(ZYGPU_PROP_VISIBLE_IN_SSS, in_sss);
    set(ZYGPU_PROP_SHADOW_CATCHER_LIGHT, shadow_catcher_light);
    set(ZYGPU_PROP_VISIBLE_IN_SHADOW, (f & ZYGPU_PROP_SHADOW_CATCHER) ? false : in_reflection);
    return 0;
}

int Engine::setShadowCatcher(uint32_t prop) {  // Prop.setShadowCatcher (prop.zig:89-92)
    if (prop >= props_.size()) return -1;
    props_[prop].flags |= ZYGPU_PROP_SHADOW_CATCHER;
    props_[prop].flags &= ~uint32_t(ZYGPU_PROP_VISIBLE_IN_SHADOW);
    return 0;
}

void Engine::cameraCreate(uint32_t width, uint32_t height) {  // capi.zig:143-167
    resolution[0] = int32_t(width);
    resolution[1] = int32_t(height);
    crop[0] = 0;
    crop[1] = 0;
    crop[2] = resolution[0];
    crop[3] = resolution[1];
    fov = float(80.0 * (M_PI / 180.0));
    if (ZYGPU_NULL == camera_entity) camera_entity = createEntity();
}

// Shaper.drawAperture + resolve(Byte1) (rendering/shaper.zig:156-301, 36-48) as camera_perspective.zig:284-300 calls them: a regular
// polygon of `blades` edges (up to 8: ApertureN's edge planes, 96-131; more: the same planes on the fly, 262-286), radius 0.5 around the
// centre of a 128 x 128 image, rotated by pi, its edges blended towards the circle by `roundness`; 4 x 4 samples per pixel, a pixel's
// coverage as a unorm byte
static std::vector<uint8_t> drawApertureBlades(uint32_t blades, float roundness, int dim) {
    std::vector<uint8_t> out(size_t(dim) * size_t(dim), 0);
    const float    pi = 3.14159265358979323846f, rot = pi, radius = 0.5f, px = 0.5f, py = 0.5f;
    const float    delta = (2.f * pi) / float(blades);
    const int      sub_samples = 4;
    const float    end = float(dim), ss = 1.f / float(sub_samples), ssx = ss / end, so = 0.5f * ss, ss2 = 1.f / float(sub_samples * sub_samples);
    auto inside = [&](float u, float v) {
        const float cx = u - px, cy = v - py;
        const float lc = std::sqrt(cx * cx + cy * cy);
        if (lc > radius) return false;
        float bx = std::sin(rot), by = std::cos(rot), mt = 0.f;
        for (uint32_t i = 0; i < blades; ++i) {
            const float angle = float(i + 1) * delta + rot;
            const float ccx = std::sin(angle), ccy = std::cos(angle);
            const float ex = ccx - bx, ey = ccy - by;
            const float len = std::sqrt(ey * ey + ex * ex);  // normalize2(.{cb[1], -cb[0]})
            const float nx = ey / len, ny = -ex / len;
            const float d = nx * bx + ny * by;
            mt = fmax_(mt, (nx * cx + ny * cy) / d);
            bx = ccx, by = ccy;
        }
        return std::fmaf(1.f - roundness, mt, roundness * lc) <= radius;  // math.lerp(mt, lc, roundness)
    };
    const int begin = int((px - radius) * end), finish = int((px + radius) * end);
    for (int y = begin; y < finish; ++y)
        for (int x = begin; x < finish; ++x) {
            float w = 0.f, v = (float(y) + so) / end;
            for (int sy = 0; sy < sub_samples; ++sy) {
                float u = (float(x) + so) / end;
                for (int sx = 0; sx < sub_samples; ++sx) {
                    if (inside(u, v)) w += ss2;
                    u += ssx;
                }
                v += ssx;
            }
            // pixel = lerp(0, 1, w) = w; resolve: floatToUnorm8(clamp(w, 0, 1)) (base/encoding.zig:6-8)
            if (w > 0.f) out[size_t(y) * size_t(dim) + size_t(x)] = uint8_t(fmin_(w, 1.f) * 255.f + 0.5f);
        }
    return out;
}

// Perspective.setParameters (camera_perspective.zig:260-304) + loadFocus (348-361)
bool Engine::loadCameraParameters(const JsonValue& v) {
    camera_error.clear();  // (a refusal belongs to the call that caused it: a later valid call replaces the parameters)
    if (JsonValue::Object != v.type) return true;
    for (const auto& kv : v.obj) {
        if (kv.first == "fov") {
            fov = kv.second.f32() * float(M_PI / 180.0);
        } else if (kv.first == "lens") {
            lens_radius = kv.second.memberF32("radius", lens_radius);
        } else if (kv.first == "aperture") {
            lens_radius = kv.second.memberF32("radius", lens_radius);
            const JsonValue* shape = kv.second.get("shape");
            aperture_shape_image = ZYGPU_NULL;  // (Aperture is replaced with the node: a shape lasts as long as the node names it)
            if (shape && JsonValue::String == shape->type && !shape->str.empty()) {
                // tx.Provider.loadFile with usage Opacity (camera_perspective.zig:274-283): one channel, the file's W (or only) one
                aperture_shape_image = loadImageFile(shape->str, kUsageOpacity, std::string(), false);
                if (ZYGPU_NULL == aperture_shape_image) return camera_error = "the camera's aperture shape: " + file_error, false;
                const HostImage& img = images_[aperture_shape_image];
                // Aperture.setShape (aperture.zig:22-44) sizes its row of weights by the image's height and fills it by its width: only
                // a square image is read as it was meant
                if (img.width != img.height) return camera_error = "the camera's aperture shape must be a square image (camera/aperture.zig:22-44 reads no other correctly)", false;
                if (ZYGPU_IMAGE_BYTE1_UNORM != img.format && ZYGPU_IMAGE_FLOAT1 != img.format)
                    return camera_error = "the camera's aperture shape must be a one-channel image", false;
            } else if (kv.second.memberU32("blades", 0) > 3) {
                // camera_perspective.zig:284-300: the blades drawn into a 128 x 128 Byte1 image, which then is the aperture's shape
                const uint32_t blades = kv.second.memberU32("blades", 0);
                if (blades > 64) return camera_error = "a camera aperture with more than 64 \"blades\"", false;
                const std::vector<uint8_t> shape_pixels = drawApertureBlades(blades, kv.second.memberF32("roundness", 0.f), 128);
                aperture_shape_image = createImage(ZYGPU_NULL, 0, 1, 128, 128, 1, 1, shape_pixels.data());
                if (ZYGPU_NULL == aperture_shape_image) return camera_error = "the camera's aperture blades: the image could not be created", false;
            }
        } else if (kv.first == "focus") {
            if (JsonValue::Object != kv.second.type) continue;
            focus_use_point = false;  // loadFocus makes a new Focus (camera_perspective.zig:348-361): a point lasts as long as the node names it
            for (const auto& f : kv.second.obj) {
                if (f.first == "point" && JsonValue::Array == f.second.type && f.second.arr.size() >= 3) {
                    for (int i = 0; i < 3; ++i) focus_point[i] = f.second.arr[size_t(i)].f32();
                    focus_use_point = true;
                } else if (f.first == "distance") {
                    focus_distance = f.second.f32();
                }
            }
        } else if (kv.first == "stereo") {
            // camera_perspective.zig:99-113, 301-303: two eye layers; the film, the sensor and the C ABI's resolve here have one
            return camera_error = "a \"stereo\" camera (two layers: camera_perspective.zig:99-113) is not provided", false;
        }
    }
    return true;
}

bool Engine::focusProbe(float ray[8]) const {
    if (!focus_use_point || !(lens_radius > 0.f)) return false;  // (or stereo.ipd > 0: refused above)
    const ZygpuViewDesc& v = view_desc_;
    // setFocus scales the point by the resolution (camera_perspective.zig:306-313); updateFocus' direction (322-327)
    const float px = focus_point[0] * float(resolution[0]), py = focus_point[1] * float(resolution[1]);
    Vec4f       dir = vec(v.left_top[0], v.left_top[1], v.left_top[2], 0.f) + vec(v.d_x[0], v.d_x[1], v.d_x[2], 0.f) * splat(px) +
                vec(v.d_y[0], v.d_y[1], v.d_y[2], 0.f) * splat(py);
    dir = normalize3(dir);
    // scene.propTransformationAt(camera, time) at the frame's start: the static transformation, or the first keyframe (whose position
    // is the scene's origin, space.zig:121-124)
    const ZygpuTrafo& t = v.camera_trafo;
    const Vec4f r0 = vec(t.r[0][0], t.r[0][1], t.r[0][2], 0.f), r1 = vec(t.r[1][0], t.r[1][1], t.r[1][2], 0.f), r2 = vec(t.r[2][0], t.r[2][1], t.r[2][2], 0.f);
    // ComposedTransformation.objectToWorldVector: rotation columns scaled (composed_transformation.zig:60-70)
    const Vec4f a = r0 * splat(t.r[0][3]), b = r1 * splat(t.r[1][3]), c = r2 * splat(t.r[2][3]);
    Vec4f       w = splat(dir[0]) * a;
    w = vec(std::fmaf(dir[1], b[0], w[0]), std::fmaf(dir[1], b[1], w[1]), std::fmaf(dir[1], b[2], w[2]), 0.f);
    w = vec(std::fmaf(dir[2], c[0], w[0]), std::fmaf(dir[2], c[1], w[1]), std::fmaf(dir[2], c[2], w[2]), 0.f);
    ray[0] = t.position[0], ray[1] = t.position[1], ray[2] = t.position[2], ray[3] = 0.f;
    ray[4] = w[0], ray[5] = w[1], ray[6] = w[2], ray[7] = 2.14748313e+09f;  // ro.RayMaxT (ray_offset.zig:5)
    return true;
}

void Engine::applyFocusProbe(bool hit, float t) {
    if (hit) view_desc_.focus_distance = t + focus_point[2];  // camera_perspective.zig:341-343; no hit: the distance stays
}

// take/take.zig:131-271
void Engine::loadIntegrators(const JsonValue& v) {
    if (JsonValue::Object != v.type) return;
    if (const JsonValue* st = v.get("specular_threshold")) {
        const float x = st->f32();
        specular_threshold = x * x;
    }
    // the light tracer ("particle", take.zig:136-138, 227-237), photon mapping ("photon" with photons, 146-148) and the AOV
    // surface integrator (154-187) are other algorithms than the path tracers in scope: a frame that asks for one does not
    // start (compile returns -10) instead of rendering without it
    unsupported_integrator_ = false;
    if (const JsonValue* particle = v.get("particle"))
        if (JsonValue::Object == particle->type && particle->memberU32("particles_per_pixel", 1) > 0) unsupported_integrator_ = true;
    if (const JsonValue* photon = v.get("photon"))
        if (JsonValue::Object == photon->type && photon->memberU32("num_photons", 0) > 0) unsupported_integrator_ = true;
    const JsonValue* surface = v.get("surface");
    if (!surface || JsonValue::Object != surface->type) return;
    if (surface->get("AOV")) unsupported_integrator_ = true;

    auto loadDepth = [&](const JsonValue& n) {
        max_depth_surface = 16;
        max_depth_volume = 256;
        if (const JsonValue* d = n.get("depth")) {
            max_depth_surface = uint16_t(d->memberU32("surface", 16));
            max_depth_volume = uint16_t(d->memberU32("surface", 256));  // reads "surface" for both (take.zig:257-258)
        }
    };
    auto loadLightSampling = [&](const JsonValue& n) {
        float st = 0.5f;
        if (const JsonValue* ls = n.get("light_sampling")) st = clamp_(ls->memberF32("split_threshold", 0.5f), 0.f, 1.f);
        const float st2 = st * st;
        split_threshold = st2 * st2;
    };

    for (const auto& kv : surface->obj) {
        const JsonValue& n = kv.second;
        if (kv.first == "PT") {
            integrator = ZYGPU_INTEGRATOR_PT;
            caustics_path = n.memberBool("caustics", true);
            loadDepth(n);
        } else if (kv.first == "PTDL") {
            integrator = ZYGPU_INTEGRATOR_PTDL;
            caustics_path = n.memberBool("caustics", true);
            loadDepth(n);
            loadLightSampling(n);
        } else if (kv.first == "PTMIS") {
            integrator = ZYGPU_INTEGRATOR_PTMIS;
            regularize_roughness = n.memberF32("regularize_roughness", 0.f);
            caustics_path = n.memberBool("caustics", true);
            loadDepth(n);
            loadLightSampling(n);
        }
    }
}

// cli/take_loader.zig:193-230 (sensor node of a take); the reference C ABI has no entry for it
void Engine::loadSensor(const JsonValue& v) {
    sensor.alpha_transparency = v.memberBool("alpha_transparency", false);  // take_loader.zig:193-196
    sensor.clamp_emission = sensor.clamp_direct = sensor.clamp_indirect = FLT_MAX;
    if (const JsonValue* c = v.get("clamp")) {
        if (JsonValue::Object == c->type) {
            sensor.clamp_emission = c->memberF32("emission", sensor.clamp_emission);
            sensor.clamp_direct = c->memberF32("direct", sensor.clamp_direct);
            sensor.clamp_indirect = c->memberF32("indirect", sensor.clamp_indirect);
        }
    }
    // the take's post-processor (take_loader.zig:245-283), accepted here next to the sensor's own settings:
    // {"tonemapper": {"ACES" | "AgX" | "Linear" | "PbrNeutral": {"exposure": e}}}
    sensor.tonemapper = ZYGPU_TONEMAP_LINEAR;
    sensor.exposure = 0.f;
    if (const JsonValue* t = v.get("tonemapper")) {
        if (JsonValue::Object == t->type) {
            for (const auto& kv : t->obj) {
                const float exposure = JsonValue::Object == kv.second.type ? kv.second.memberF32("exposure", 0.f) : 0.f;
                uint32_t    cls = ZYGPU_NULL;
                if (kv.first == "ACES") cls = ZYGPU_TONEMAP_ACES;
                else if (kv.first == "AgX") cls = ZYGPU_TONEMAP_AGX;
                else if (kv.first == "Linear") cls = ZYGPU_TONEMAP_LINEAR;
                else if (kv.first == "PbrNeutral") cls = ZYGPU_TONEMAP_PBR_NEUTRAL;
                if (ZYGPU_NULL != cls) {
                    sensor.tonemapper = cls;
                    sensor.exposure = exposure;
                    break;
                }
            }
        }
    }
    if (const JsonValue* f = v.get("filter")) {
        if (JsonValue::Object == f->type) {
            for (const auto& kv : f->obj) {
                if (kv.first == "Blackman") { setFilter(2.f, 1); return; }
                if (kv.first == "Mitchell") { setFilter(2.f, 0); return; }
            }
        }
    }
    setFilter(0.f, 1);
}

void Engine::buildPropTree(const std::vector<uint32_t>& indices, std::vector<ZygpuBvhNode>& nodes,
                           std::vector<uint32_t>& out_indices) {  // prop_tree_builder.zig:24-62
    nodes.clear();
    out_indices.clear();
    if (indices.empty()) return;
    std::vector<BvhReference> refs(indices.size());
    AABB                      bounds = emptyAabb();
    for (size_t i = 0; i < indices.size(); ++i) {
        const ZygpuAabb& b = c_aabbs_[indices[i]];
        for (int k = 0; k < 3; ++k) {
            refs[i].mn[k] = b.min[k];
            refs[i].mx[k] = b.max[k];
        }
        refs[i].prim = indices[i];
        aabbMerge(bounds, {{vec(b.min[0], b.min[1], b.min[2], b.min[3]), vec(b.max[0], b.max[1], b.max[2], b.max[3])}});
    }
    BvhBuilder builder(BvhSettings{16, 64, 4});  // prop_tree_builder.zig:17
    builder.build(std::move(refs), bounds, nodes, out_indices);
}

// triangle_mesh.zig:721-749 (Mesh.prepareSampling) + 57-149 (Part.configure), uniform emission, one task
void Engine::ensureSrgbTable() {  // texture/encoding.zig:41-51 over spectrum/srgb.zig:28-38
    if (!srgb_to_float_.empty()) return;
    srgb_to_float_.resize(256);
    for (int i = 0; i < 256; ++i) {
        const float c = float(i) / 255.f;
        srgb_to_float_[i] = c <= 0.f ? 0.f : c < 0.04045f ? c / 12.92f : c < 1.f ? std::pow((c + 0.055f) / 1.055f, 2.4f) : 1.f;
    }
}

namespace {
// Distribution1D.configure(data, 0) (base/math/distribution_1d.zig:29-36, 104-171), appended to the flat arrays;
// returns the distribution's integral
float appendDistribution1D(const std::vector<float>& data, HostImageLight& out) {
    const uint32_t cdf_offset = uint32_t(out.cdf.size()), lut_offset = uint32_t(out.lut.size());
    float integral = 0.f;
    for (float d : data) integral += d;
    uint32_t size;
    if (0.f == integral) {
        size = 2;
        out.cdf.push_back(1.f);
        out.cdf.push_back(1.f);
    } else {
        size = uint32_t(data.size()) + 1;
        const float ii = 1.f / integral;
        float       p = 0.f;
        out.cdf.push_back(0.f);
        for (size_t i = 0; i + 1 < data.size(); ++i) {
            p = std::fmaf(data[i], ii, p);
            out.cdf.push_back(p);
        }
        out.cdf.push_back(1.f);
    }
    uint32_t lut_size = uint32_t(data.size() / 16);
    lut_size = std::min(std::max(lut_size, 1u), size);
    const float lut_range = float(lut_size);
    out.lut.resize(lut_offset + lut_size + 1, 0u);
    uint32_t*    lut = out.lut.data() + lut_offset;
    const float* cdf = out.cdf.data() + cdf_offset;
    lut[0] = 1;
    uint32_t border = 0;
    for (uint32_t i = 1; i < size; ++i) {
        const uint32_t mapped = uint32_t(cdf[i] * lut_range);
        if (mapped > border) {
            for (uint32_t k = border + 1; k <= mapped; ++k) lut[k] = i;
            border = mapped;
        }
    }
    uint32_t range_bits;
    std::memcpy(&range_bits, &lut_range, 4);
    out.rows.insert(out.rows.end(), {cdf_offset, size, lut_offset, range_bits});
    return integral;
}
}  // namespace

// Light.Material.prepareSampling for an emission map (light_material.zig:52-125, 201-271): uv-weighted luminance per texel,
// MIS compensation (luminance minus 0.6 x the mean), one Distribution1D per row and the marginal over the rows' integrals
uint32_t Engine::prepareImageLight(uint32_t shape_kind, const ZygpuMaterial& m, bool compensate) {
    ensureSrgbTable();
    const HostImage& img = images_[m.emission_image];
    const int        w = img.width, h = img.height;
    image_lights_.emplace_back();
    HostImageLight& il = image_lights_.back();
    il.num_rows = uint32_t(h);
    il.sampler = m.emission_sampler;

    auto texel = [&](int x, int y) {  // Texture.image2D_3 (texture.zig:188-219)
        const size_t i = size_t(y) * size_t(w) + size_t(x);
        if (ZYGPU_IMAGE_BYTE3_SRGB == img.format) {
            const uint8_t* p = img.pixels.data() + 3 * i;
            const float    r = srgb_to_float_[p[0]], g = srgb_to_float_[p[1]], b = srgb_to_float_[p[2]];
            return vec(0.61309732f, 0.07019422f, 0.02061560f, 0.f) * splat(r) + vec(0.33952285f, 0.91635557f, 0.10956983f, 0.f) * splat(g) +
                   vec(0.04737928f, 0.01345259f, 0.86981512f, 0.f) * splat(b);
        }
        const float* p = reinterpret_cast<const float*>(img.pixels.data()) + 3 * i;
        return vec(p[0], p[1], p[2], 0.f);
    };
    auto uvWeight = [&](float u, float v) {  // shape.zig:535-542
        if (ZYGPU_SHAPE_DOME == shape_kind) return std::sin(v * float(M_PI));
        if (ZYGPU_SHAPE_CANOPY == shape_kind || ZYGPU_SHAPE_DISK == shape_kind) {  // canopy.zig uvWeight; disk.zig:428-436
            const float dx = 2.f * u - 1.f, dy = 2.f * v - 1.f;
            return dx * dx + dy * dy > 1.f ? 0.f : 1.f;
        }
        return 1.f;
    };

    std::vector<float> luminance(size_t(w) * size_t(h));
    const float        idf0 = 1.f / float(w), idf1 = 1.f / float(h);
    Vec4f              avg = splat(0.f);
    for (int y = 0; y < h; ++y) {
        const float v = idf1 * (float(y) + 0.5f);
        for (int x = 0; x < w; ++x) {
            const float u = idf0 * (float(x) + 0.5f);
            const float uv_weight = uvWeight(u, v);
            const Vec4f wr = splat(uv_weight) * texel(x, y);
            avg = avg + vec(wr[0], wr[1], wr[2], uv_weight);
            luminance[size_t(y) * w + x] = std::max(std::max(wr[0], wr[1]), wr[2]);
        }
    }
    const Vec4f average = avg / splat(avg[3]);
    il.total_weight = avg[3];
    il.average_emission = vec(m.emission[0], m.emission[1], m.emission[2], 0.f) * average;

    // the Sky material's map is sampled by its luminance as it is (sky_material.zig:89-133, 170-215: no compensation)
    const float        al = 0.6f * std::max(std::max(average[0], average[1]), average[2]);
    std::vector<float> integrals(h), row(w);
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const float l = luminance[size_t(y) * w + x];
            row[x] = compensate ? std::max(l - al, std::min(l, 0.0025f)) : l;
        }
        integrals[y] = appendDistribution1D(row, il);
    }
    appendDistribution1D(integrals, il);
    return uint32_t(image_lights_.size() - 1);
}

// Light.Material.preparePortalSampling (light_material.zig:142-199, 275-375): the linked Dome's emission map seen through the
// portal.  average_emission: the uv-weighted mean of the map over the hemisphere behind the portal (PortalHemisphereContext); the
// function the WindowedDistribution2D tabulates: for every texel of the portal's own dim x dim image (Portal.imageToWorld, 4 x 4
// subsamples) the map's brightest channel in that direction (nearest texel) times the mapping's weight (PortalLuminanceContext);
// then its summed-area table (summed_area_table.zig:18-48).
uint32_t Engine::preparePortalLight(const ZygpuMaterial& m, const Transformation& portal, const Transformation& dome, uint32_t light_link) {
    ensureSrgbTable();
    const HostImage& img = images_[m.emission_image];
    const int        w = img.width, h = img.height;
    portal_lights_.emplace_back();
    HostPortalLight& pl = portal_lights_.back();
    pl.light_link = light_link;
    pl.dim = uint32_t(h);
    auto texel = [&](int x, int y) {  // Texture.image2D_3 (texture.zig:188-219)
        const size_t i = size_t(y) * size_t(w) + size_t(x);
        if (ZYGPU_IMAGE_BYTE3_SRGB == img.format) {
            const uint8_t* p = img.pixels.data() + 3 * i;
            const float    r = srgb_to_float_[p[0]], g = srgb_to_float_[p[1]], b = srgb_to_float_[p[2]];
            return vec(0.61309732f, 0.07019422f, 0.02061560f, 0.f) * splat(r) + vec(0.33952285f, 0.91635557f, 0.10956983f, 0.f) * splat(g) +
                   vec(0.04737928f, 0.01345259f, 0.86981512f, 0.f) * splat(b);
        }
        const float* p = reinterpret_cast<const float*>(img.pixels.data()) + 3 * i;
        return vec(p[0], p[1], p[2], 0.f);
    };
    const Mat3x3 portal_rot = quaternionToMat3x3(portal.rotation), dome_rot = quaternionToMat3x3(dome.rotation);
    const float  pi = float(M_PI);
    const Vec4f  portal_dir = portal_rot.r[2];
    {  // PortalHemisphereContext
        const float idf0 = 1.f / float(w), idf1 = 1.f / float(h);
        Vec4f       avg = splat(0.f);
        for (int y = 0; y < h; ++y) {
            const float v = idf1 * (float(y) + 0.5f);
            for (int x = 0; x < w; ++x) {
                const float u = idf0 * (float(x) + 0.5f);
                // Dome.imageToWorld (dome.zig:68-80)
                const float phi = (u - 0.5f) * (2.f * pi), theta = v * pi;
                const Vec4f ldir = vec(std::sin(phi) * std::sin(theta), std::cos(theta), std::cos(phi) * std::sin(theta));
                const Vec4f dir = transformVector(dome_rot, ldir);
                if (dot3(portal_dir, dir) <= 0.f) {  // "We see the hemisphere behind the portal"
                    const float uv_weight = std::sin(v * pi);  // Shape.uvWeight of a Dome (shape.zig:539)
                    const Vec4f wr = splat(uv_weight) * texel(x, y);
                    avg = avg + vec(wr[0], wr[1], wr[2], uv_weight);
                }
            }
        }
        const Vec4f average = avg / splat(avg[3]);
        pl.average_emission = vec(m.emission[0], m.emission[1], m.emission[2], 0.f) * average;
    }
    const uint32_t dim = pl.dim;
    pl.func.assign(size_t(dim) * dim, 0.f);
    {  // PortalLuminanceContext
        const float idf = 1.f / float(dim), nsf = 4.f, so = 0.5f / nsf;
        const bool  ru = 0 != (m.emission_sampler & ZYGPU_TEX_REPEAT_U), rv = 0 != (m.emission_sampler & ZYGPU_TEX_REPEAT_V);
        auto address = [](bool repeat, float x) { return repeat ? x - std::floor(x) : std::min(std::max(x, 0.f), 1.f); };
        for (uint32_t y = 0; y < dim; ++y) {
            for (uint32_t x = 0; x < dim; ++x) {
                float luminance = 0.f;
                for (int yy = 0; yy < 4; ++yy) {
                    const float v = idf * (float(y) + so + float(yy) / nsf);
                    for (int xx = 0; xx < 4; ++xx) {
                        const float u = idf * (float(x) + so + float(xx) / nsf);
                        // Portal.imageToWorld (portal.zig:16-25)
                        const float a = -pi * 0.5f + u * pi, b = -pi * 0.5f + v * pi;
                        const Vec4f wv = normalize3(vec(std::tan(a), std::tan(b), 1.f));
                        const float weight = (pi * pi) * (1.f - wv[0] * wv[0]) * (1.f - wv[1] * wv[1]) / wv[2];
                        const Vec4f dir = -transformVector(portal_rot, wv);
                        // Dome.worldToImage (dome.zig:59-66)
                        const Vec4f xyz = normalize3(transformVectorTransposed(dome_rot, dir));
                        const float du = std::atan2(xyz[0], xyz[2]) * ((1.f / pi) * 0.5f) + 0.5f, dv = std::acos(xyz[1]) * (1.f / pi);
                        // ts.sampleImageNearest2D_3 (texture_sampler.zig:105-123)
                        const float s_ = address(ru, m.emission_scale[0] * du), t_ = address(rv, m.emission_scale[1] * dv);
                        const int   tx = std::min(int(s_ * float(w)), w - 1), ty = std::min(int(t_ * float(h)), h - 1);
                        luminance += hmax3(texel(tx, ty)) * weight;
                    }
                }
                pl.func[size_t(y) * dim + x] = luminance / (nsf * nsf);
            }
        }
    }
    {  // SummedAreaTable.init
        const int32_t sw = int32_t(dim) + 1;
        pl.sat.assign(size_t(sw) * sw, 0.f);
        auto at = [&](int32_t x, int32_t y) -> float& { return pl.sat[size_t(y) * sw + x]; };
        auto f = [&](int32_t x, int32_t y) { return pl.func[size_t(y) * dim + x]; };
        at(1, 1) = f(0, 0);
        for (int32_t x = 2; x < sw; ++x) at(x, 1) = f(x - 1, 0) + at(x - 1, 1);
        for (int32_t y = 2; y < sw; ++y) at(1, y) = f(0, y - 1) + at(1, y - 1);
        for (int32_t y = 2; y < sw; ++y)
            for (int32_t x = 2; x < sw; ++x) at(x, y) = f(x - 1, y - 1) + at(x - 1, y) + at(x, y - 1) - at(x - 1, y - 1);
    }
    return uint32_t(portal_lights_.size() - 1);
}

uint32_t Engine::prepareMeshLight(uint32_t mesh_id, uint32_t part, uint32_t material, bool two_sided) {
    for (size_t i = 0; i < mesh_lights_.size(); ++i) {
        const HostMeshLight& ml = mesh_lights_[i];
        if (ml.mesh == mesh_id && ml.part == part && ml.material == material) return uint32_t(i);
    }
    HostMesh&      mesh = meshes_[mesh_id];
    const uint32_t nt = mesh.numTriangles();
    if (mesh.primitive_mapping.empty()) {
        mesh.primitive_mapping.resize(nt);
        std::vector<uint32_t> per_part(mesh.numParts(), 0);
        for (uint32_t t = 0; t < nt; ++t) mesh.primitive_mapping[t] = per_part[mesh.triangle_parts[t]]++;
    }

    mesh_lights_.emplace_back();
    HostMeshLight& ml = mesh_lights_.back();
    ml.mesh = mesh_id;
    ml.part = part;
    ml.material = material;
    ml.two_sided = two_sided;
    for (uint32_t t = 0; t < nt; ++t)
        if (mesh.triangle_parts[t] == part) ml.triangle_mapping.push_back(t);
    const uint32_t num = uint32_t(ml.triangle_mapping.size());

    auto tri = [&](uint32_t local, Vec4f& a, Vec4f& b, Vec4f& c) {
        const uint32_t t = ml.triangle_mapping[local];
        a = mesh.position(mesh.triangles[3 * t + 0]);
        b = mesh.position(mesh.triangles[3 * t + 1]);
        c = mesh.position(mesh.triangles[3 * t + 2]);
    };

    std::vector<float>     powers(num);
    std::vector<ZygpuAabb> boxes(num);
    std::vector<float>     cones(size_t(num) * 4);
    AABB                   bb = emptyAabb();
    Vec4f                  dominant_axis = splat(0.f);
    float                  total_power = 0.f;
    for (uint32_t i = 0; i < num; ++i) {
        Vec4f a, b, c;
        tri(i, a, b, c);
        const float pow = 0.5f * length3(cross3(b - a, c - a));  // triangle.area; uniform emission: pow = area
        powers[i] = pow;
        const Vec4f n = normalize3(cross3(b - a, c - a));
        const AABB  box = {{min4(a, min4(b, c)), max4(a, max4(b, c))}};
        for (int k = 0; k < 4; ++k) {
            boxes[i].min[k] = box.b[0][k];
            boxes[i].max[k] = box.b[1][k];
            cones[4 * i + k] = k < 3 ? n[k] : 1.f;
        }
        if (pow > 0.f) {
            dominant_axis = dominant_axis + splat(pow) * n;
            aabbMerge(bb, box);
            total_power += pow;
        }
    }

    if (dominant_axis[0] == dominant_axis[1] && dominant_axis[1] == dominant_axis[2]) {
        ml.cone = vec(0.f, 0.f, 1.f, -1.f);
    } else {
        const Vec4f da = normalize3(dominant_axis / splat(total_power));
        float       angle = 0.f;
        for (uint32_t i = 0; i < num; ++i) angle = fmax_(angle, std::acos(dot3(da, vec(cones[4 * i], cones[4 * i + 1], cones[4 * i + 2]))));
        ml.cone = vec(da[0], da[1], da[2], std::cos(angle));
    }
    ml.aabb = bb;

    // Distribution1D.precomputePdfCdf (base/math/distribution_1d.zig:87-124); pdfI = cdf[i+1] - cdf[i]
    std::vector<float> cdf(size_t(num) + 1, 0.f);
    float              integral = 0.f;
    for (float d : powers) integral += d;
    std::vector<float> rel(num, 0.f);
    if (integral > 0.f) {
        const float ii = 1.f / integral;
        float       p = 0.f;
        for (uint32_t i = 0; i + 1 < num; ++i) {
            const float cc = std::fmaf(powers[i], ii, p);
            cdf[i + 1] = cc;
            p = cc;
        }
        cdf[num] = 1.f;
        for (uint32_t i = 0; i < num; ++i) rel[i] = cdf[i + 1] - cdf[i];
    }

    // MeshImpl.lightProperties (shape_sampler.zig:199-223)
    ml.tri_props.resize(size_t(num) * 8);
    for (uint32_t i = 0; i < num; ++i) {
        Vec4f a, b, c;
        tri(i, a, b, c);
        const Vec4f center = (a + b + c) / splat(3.f);
        const float radius = std::sqrt(fmax_(squaredLength3(a - center), fmax_(squaredLength3(b - center), squaredLength3(c - center))));
        float*      o = ml.tri_props.data() + size_t(i) * 8;
        o[0] = center[0];
        o[1] = center[1];
        o[2] = center[2];
        o[3] = radius;
        o[4] = cones[4 * i + 0];
        o[5] = cones[4 * i + 1];
        o[6] = cones[4 * i + 2];
        o[7] = rel[i];
    }

    LightTreeInput lin;
    lin.aabbs = boxes.data();
    lin.cones = cones.data();
    lin.two_sided = nullptr;
    lin.all_two_sided = two_sided;
    lin.powers = rel.data();  // lightPower = distribution.pdfI (shape_sampler.zig:193-197)
    lin.num_lights = num;
    LightTreeBuilder builder;
    builder.buildPrimitive(lin, bb, ml.cone, integral, ml.tree);  // shape_sampler.power() = distribution.integral
    return uint32_t(mesh_lights_.size() - 1);
}


// ---------------------------------------------------------------- Sky (sky/sky.zig, sky/sky_material.zig)
namespace {
// Model.AngularRadius = arp.PSMG_SUN_RADIUS = 0.2667 degrees (thirdparty/arpraguesky/ArPragueSkyModelGround.h:88);
// Sky.Radius = tan of it as f32 (sky.zig:46)
const float  kSunAngularRadius = float(0.2667 * (3.141592653589793 / 180.0));
const float  kSkyRadius = 0.0046548266f;  // tan(kSunAngularRadius) as f32, the literal the device and the oracle use
inline float sunWiDotZ(float elevation, float v) {  // sky.zig:271-275
    const float y = (2.f * v) - 1.f;
    return std::sin(elevation + y * kSkyRadius);
}
inline Mat3x3 mulMat3(const Mat3x3& a, const Mat3x3& b) {  // matrix3x3.zig:99-111
    Mat3x3 o;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o.r[i][j] = a.r[i][0] * b.r[0][j] + a.r[i][1] * b.r[1][j] + a.r[i][2] * b.r[2][j];
        o.r[i][3] = 0.f;
    }
    return o;
}
}  // namespace

int32_t Engine::createSky() {  // Scene.createSky -> Sky.configure (sky.zig:53-76)
    if (ZYGPU_NULL != sky_.sun) return int32_t(sky_.sun);
    // the sun: a Sky material with no map (its radiance is the table), on a Distant shape
    JsonValue none;
    none.type = JsonValue::Object;
    ZygpuMaterial sun{};
    parseMaterial(sun, none, ZYGPU_MAT_SKY);  // the defaults
    sun.flags |= ZYGPU_MATF_EMISSIVE;  // SkyMaterial.commit (sky_material.zig:60-63)
    materials_.push_back(sun);
    sky_.sun_material = uint32_t(materials_.size() - 1);
    const int32_t sun_prop = createPropShape(ZYGPU_SHAPE_DISTANT, &sky_.sun_material, 1, true);
    // the sky: the baked image as the emission map of an emitter on a Canopy; evaluateRadiance is the map's texel
    // (sky_material.zig:150-159), which is what a Light material with emittance 1 and that map gives
    HostImage img;
    img.format = ZYGPU_IMAGE_FLOAT3;
    img.width = img.height = 1;
    img.pixels.assign(12, 0);
    images_.push_back(std::move(img));
    sky_.image = uint32_t(images_.size() - 1);
    ZygpuMaterial sky{};
    parseMaterial(sky, none, ZYGPU_MAT_LIGHT);
    sky.emission[0] = sky.emission[1] = sky.emission[2] = 1.f;
    sky.emission_image = sky_.image;
    sky.emission_sampler = ZYGPU_TEX_LINEAR;  // Texture.DefaultClampMode: clamp, the default (bilinear) filter
    sky.emission_scale[0] = sky.emission_scale[1] = 1.f;
    sky.flags |= ZYGPU_MATF_EMISSIVE;
    materials_.push_back(sky);
    sky_.sky_material = uint32_t(materials_.size() - 1);
    const int32_t sky_prop = createPropShape(ZYGPU_SHAPE_CANOPY, &sky_.sky_material, 1, true);
    if (sun_prop < 0 || sky_prop < 0) return -1;
    sky_.sun = uint32_t(sun_prop);
    sky_.sky = uint32_t(sky_prop);
    createLight(sky_.sun);
    createLight(sky_.sky);
    return sun_prop;
}

void Engine::setSkyParameters(const JsonValue& v) {  // Sky.setParameters (sky.zig:78-93)
    if (JsonValue::Object != v.type) return;
    for (const auto& kv : v.obj) {
        if (kv.first == "sun") {
            Vec4f angles = splat(0.f);
            if (const JsonValue* r = kv.second.get("rotation"))
                if (JsonValue::Array == r->type && r->arr.size() >= 3) angles = vec(r->arr[0].f32(), r->arr[1].f32(), r->arr[2].f32());
            // json.createRotationMatrix (base/json.zig:169-175): Rz * Rx * Ry of the angles in degrees
            const float k = float(M_PI / 180.0);
            auto rx = [](float a) { const float c = std::cos(a), s = std::sin(a); return Mat3x3{{vec(1.f, 0.f, 0.f), vec(0.f, c, -s), vec(0.f, s, c)}}; };
            auto ry = [](float a) { const float c = std::cos(a), s = std::sin(a); return Mat3x3{{vec(c, 0.f, s), vec(0.f, 1.f, 0.f), vec(-s, 0.f, c)}}; };
            auto rz = [](float a) { const float c = std::cos(a), s = std::sin(a); return Mat3x3{{vec(c, -s, 0.f), vec(s, c, 0.f), vec(0.f, 0.f, 1.f)}}; };
            sky_.sun_rotation = mulMat3(mulMat3(rz(angles[2] * k), rx(angles[0] * k)), ry(angles[1] * k));
        } else if (kv.first == "turbidity") {  // Model.turbidityToVisibility (sky_model.zig:140-142)
            const float t = kv.second.f32();
            sky_.visibility = 7487.f * std::exp(-3.41f * t) + 117.1f * std::exp(-0.4768f * t);
        } else if (kv.first == "visibility") {
            sky_.visibility = kv.second.f32();
        } else if (kv.first == "albedo") {
            sky_.albedo = kv.second.f32();
        }
    }
}

int Engine::setSkyTables(const float* sun_rgb, const float* sky_rgb, uint32_t sky_width, uint32_t sky_height) {
    if (ZYGPU_NULL == sky_.sun || !sun_rgb || !sky_rgb || 0 == sky_width || 0 == sky_height) return -1;
    sky_.sun_rgb.assign(sun_rgb, sun_rgb + size_t(kSkySunSamples) * 3);
    HostImage& img = images_[sky_.image];
    img.width = int32_t(sky_width);
    img.height = int32_t(sky_height);
    img.pixels.resize(size_t(sky_width) * sky_height * 12);
    std::memcpy(img.pixels.data(), sky_rgb, img.pixels.size());
    sky_.tables_given = true;
    return 0;
}

// Sky.compile (sky.zig:95-170): the two props' transformations from the sun's rotation, then the bake's products
void Engine::compileSky() {
    if (ZYGPU_NULL == sky_.sun) return;
    {  // the sky is seen where the sun is (sky.zig:100)
        const uint32_t f = props_[sky_.sun].flags;
        setVisibility(sky_.sky, 0 != (f & ZYGPU_PROP_VISIBLE_IN_CAMERA), 0 != (f & ZYGPU_PROP_VISIBLE_IN_REFLECTION), false);
    }
    // "HACK: artificially set sun radius to zero if under horizon to get an early out during light sampling" (sky.zig:102-105)
    const bool  under_horizon = sky_.sun_rotation.r[2][1] > kSunAngularRadius;
    const Vec4f scale = under_horizon ? splat(0.f) : vec(kSkyRadius, kSkyRadius, kSkyRadius, 1.f);
    setWorldTransformation(sky_.sun, Transformation{splat(0.f), scale, quaternionFromMat3x3(sky_.sun_rotation)});
    const Vec4f sun_direction = sky_.sun_rotation.r[2];
    const float sun_azimuth = -std::atan2(sun_direction[0], sun_direction[2]);
    const float half_x = float(90.0 * (M_PI / 180.0)) * 0.5f, half_y = sun_azimuth * 0.5f;
    const Vec4f qx = vec(std::sin(half_x), 0.f, 0.f, std::cos(half_x)), qy = vec(0.f, std::sin(half_y), 0.f, std::cos(half_y));
    setWorldTransformation(sky_.sky, Transformation{splat(0.f), splat(1.f), quaternionMul(qx, qy)});
    const float sun_elevation = -std::asin(sun_direction[1]);

    // Sky.bakeSky (sky.zig:172-221): the Prague model (host/prague_sky.hpp) evaluated over the sun's disc and the canopy
    // (host/sky_bake.hpp), when its dataset (sky/SkyModelDataset.dat.gz, looked up through the mounts like every file: su_mount) is
    // there.  The reference caches the two images as ../cache/{sky,sun}_<hash>.exr; here the bake is kept while (visibility, albedo,
    // elevation) stay what they were, and the parsed dataset while its file stays the one it was read from.
    bool baked = false;
    if (!sky_.tables_given) {
        const float key[3] = {sky_.visibility, sky_.albedo, sun_direction[1]};
        if (sky_.baked && 0 == std::memcmp(key, sky_.baked_key, sizeof(key))) {
            baked = true;
        } else {
            std::string       why;
            const std::string dataset = resolveFile("sky/SkyModelDataset.dat.gz");
            if (!std::ifstream(dataset)) why = "sky/SkyModelDataset.dat.gz is in none of the mounted directories";
            if (why.empty() && (!sky_.dataset || dataset != sky_.dataset_path)) {
                auto parsed = std::make_shared<prague::Dataset>();
                if (parsed->load(dataset, why)) {
                    sky_.dataset = std::move(parsed);
                    sky_.dataset_path = dataset;
                } else {
                    sky_.dataset.reset();
                }
            }
            std::vector<float> sun_rgb, sky_rgb;
            if (why.empty()) {
                skybake::bake(sky_.dataset, sun_direction[1], sky_.visibility, sky_.albedo, kSkyRadius, kSkySunSamples, kSkyBakeDimension, sun_rgb, sky_rgb);
                sky_.sun_rgb = std::move(sun_rgb);
                HostImage& img = images_[sky_.image];
                img.width = img.height = kSkyBakeDimension;
                img.pixels.resize(sky_rgb.size() * 4);
                std::memcpy(img.pixels.data(), sky_rgb.data(), img.pixels.size());
                std::memcpy(sky_.baked_key, key, sizeof(key));
                sky_.baked = baked = true;
            } else {
                sky_.bake_refusal = why;
                sky_.baked = false;
            }
        }
    }
    if (!sky_.tables_given && !baked) {
        // The reference bakes both tables from the Prague sky model (sky.zig:172-269), whose dataset
        // (sky/SkyModelDataset.dat.gz) does not ship with it.  What stands in, until a client hands over the products of a
        // bake (setSkyTables): a solar disc with limb darkening whose colour reddens with the air mass, and a hemisphere that
        // is bluer towards the zenith and brighter around the sun, written through the bake's own mappings (sunWiDotZ; the
        // clipped canopy mapping with the model's axes: x right, y up, the sun in the direction -z at its elevation).
        warnings_.push_back("Sky: the Prague sky model's dataset is not available; the sun's radiance table and the sky image are a procedural stand-in (" +
                            sky_.bake_refusal + ")");
        sky_.sun_rgb.resize(size_t(kSkySunSamples) * 3);
        const float n = float(kSkySunSamples - 1);
        for (uint32_t i = 0; i < kSkySunSamples; ++i) {
            const float v = float(i) / n;
            const float wi_dot_z = sunWiDotZ(sun_elevation, v);
            const float y = 2.f * v - 1.f;
            const float limb = 1.f - 0.6f * (1.f - std::sqrt(std::max(0.f, 1.f - y * y)));
            const float air = 1.f / std::max(wi_dot_z, 0.05f);
            const float tau[3] = {0.10f, 0.16f, 0.28f}, tint[3] = {1.f, 0.92f, 0.78f};
            for (int c = 0; c < 3; ++c) sky_.sun_rgb[3 * i + c] = wi_dot_z < 0.f ? 0.f : 1.6e4f * tint[c] * limb * std::exp(-tau[c] * air);
        }
        HostImage& img = images_[sky_.image];
        img.width = img.height = kSkyBakeDimension;
        img.pixels.resize(size_t(kSkyBakeDimension) * kSkyBakeDimension * 12);
        float*      px = reinterpret_cast<float*>(img.pixels.data());
        const float idf = 1.f / float(kSkyBakeDimension), e = 3.5f * idf;
        const float sin_el = sun_direction[1];
        const Vec4f to_sun = vec(0.f, -sin_el, -std::sqrt(std::max(0.f, 1.f - sin_el * sin_el)));
        for (int32_t yy = 0; yy < kSkyBakeDimension; ++yy) {
            for (int32_t xx = 0; xx < kSkyBakeDimension; ++xx) {
                float       dx = 2.f * (idf * (float(xx) + 0.5f)) - 1.f, dy = 2.f * (idf * (float(yy) + 0.5f)) - 1.f;
                const float l = std::sqrt(dx * dx + dy * dy);
                float*      o = px + 3 * (size_t(yy) * kSkyBakeDimension + xx);
                if (l > 1.f + e) {
                    o[0] = o[1] = o[2] = 0.f;
                    continue;
                }
                if (l >= 1.f - e) {
                    dx /= l + e;
                    dy /= l + e;
                }
                // Canopy.diskToHemisphereEquidistant (canopy.zig:189-211), then the model's axes (sky.zig:340-341)
                const float longitude = std::atan2(-dy, dx), colatitude = std::sqrt(dx * dx + dy * dy) * float(M_PI / 2.0);
                const Vec4f dir = vec(std::sin(colatitude) * std::cos(longitude), std::sin(colatitude) * std::sin(longitude), std::cos(colatitude));
                const Vec4f wi = normalize3(vec(dir[0], dir[2], -dir[1]));
                const float up = clamp_(wi[1], 0.f, 1.f), cos_gamma = clamp_(dot3(wi, to_sun), -1.f, 1.f);
                const float horizon = std::pow(1.f - up, 3.f), aureole = std::exp(-(1.f - cos_gamma) * 24.f);
                const float day = clamp_(-sin_el * 4.f + 0.15f, 0.05f, 1.f);  // dimmer as the sun goes down
                o[0] = day * (0.16f + 0.55f * horizon + 1.6f * aureole);
                o[1] = day * (0.30f + 0.50f * horizon + 1.3f * aureole);
                o[2] = day * (0.72f + 0.20f * horizon + 0.9f * aureole);
            }
        }
    }
    // SkyMaterial.setSunRadiance (sky_material.zig:66-87): the table, and its mean over the part of the disc above the horizon
    c_sun_radiance_.assign(size_t(kSkySunSamples) * 4, 0.f);
    for (uint32_t i = 0; i < kSkySunSamples; ++i)
        for (int c = 0; c < 3; ++c) c_sun_radiance_[4 * i + c] = sky_.sun_rgb[3 * i + c];
    Vec4f total = splat(0.f);
    float tw = 0.f;
    for (uint32_t i = 0; i + 1 < kSkySunSamples; ++i) {
        const Vec4f s0 = vec(c_sun_radiance_[4 * i], c_sun_radiance_[4 * i + 1], c_sun_radiance_[4 * i + 2]);
        const Vec4f s1 = vec(c_sun_radiance_[4 * i + 4], c_sun_radiance_[4 * i + 5], c_sun_radiance_[4 * i + 6]);
        const float v = (float(i) + 0.5f) / float(kSkySunSamples);
        const float wi_dot_z = sunWiDotZ(sun_elevation, v);
        const float w = std::sin(v);
        tw += w;
        if (wi_dot_z >= 0.f) total = total + (s0 + s1) * splat(0.5f * w);
    }
    const Vec4f average = total / splat(tw);
    ZygpuMaterial& sun = materials_[sky_.sun_material];
    for (int c = 0; c < 3; ++c) sun.emission[c] = average[c];
}

int Engine::compile(uint32_t frame) {
    // keyframes are relative to the frame's start (space.zig:114-125): without animations the compiled scene is the same for every frame
    if (ZYGPU_NULL == camera_entity) return -1;
    if (!camera_error.empty()) return -11;  // a camera parameter that was refused by name (loadCameraParameters)
    simulate(frame);
    warnings_.clear();
    compileSky();  // scene.compile -> sky.compile (scene.zig:196-198)

    // driver.zig:164 -> scene.propWorldPosition (scene.zig:372-380): an animated entity's first keyframe
    const Vec4f    camera_pos = frameStartTransformation(camera_entity).position;
    // animated props (space.zig:23-26): their keyframes travel with the compiled scene
    c_keyframes_.clear();
    c_prop_frames_.assign(props_.size(), ZYGPU_NULL);
    for (uint32_t p = 0; p < keyframes_.size() && p < props_.size(); ++p) {
        if (keyframes_[p].empty() || p == camera_entity) continue;
        c_prop_frames_[p] = uint32_t(c_keyframes_.size());
        for (const Transformation& k : keyframes_[p]) {
            ZygpuKeyframe kf;
            for (int i = 0; i < 4; ++i) {
                kf.position[i] = k.position[i];
                kf.scale[i] = k.scale[i];
                kf.rotation[i] = k.rotation[i];
            }
            c_keyframes_.push_back(kf);
        }
    }
    const uint32_t np = uint32_t(props_.size());

    c_props_.resize(np);
    c_trafos_.resize(np);
    c_aabbs_.resize(np);

    c_meshes_.resize(meshes_.size());
    for (size_t i = 0; i < meshes_.size(); ++i) {
        const HostMesh& m = meshes_[i];
        ZygpuMesh&      d = c_meshes_[i];
        d.nodes = m.nodes.data();
        d.triangles = m.triangles.data();
        d.triangle_parts = m.triangle_parts.data();
        d.positions = m.positions.data();
        d.normals = m.normals.data();
        d.uvs = m.uvs.data();
        d.part_areas = m.part_areas.data();
        d.primitive_mapping = nullptr;
        d.num_nodes = uint32_t(m.nodes.size());
        d.num_triangles = m.numTriangles();
        d.num_vertices = m.num_vertices;
        d.num_parts = m.numParts();
    }

    for (uint32_t p = 0; p < np; ++p) {
        const HostProp&  hp = props_[p];
        const HostShape& s = shapes_[hp.shape];
        c_props_[p].shape = s.kind;
        c_props_[p].flags = hp.flags;
        c_props_[p].parts_start = hp.parts_start;
        c_props_[p].mesh = s.mesh;

        const Transformation& t = frameStartTransformation(p);
        c_trafos_[p] = composedTrafo(t, camera_pos);

        // space.zig:60-97 (calculateWorldBounds): static, or the union over the keyframes and 3 steps between each pair
        AABB bounds = aabbTransform(shapeAabb(hp.shape), objectToWorld(t));
        if (ZYGPU_NULL != c_prop_frames_[p]) {
            c_props_[p].flags &= ~uint32_t(ZYGPU_PROP_STATIC);  // prop.zig configureAnimated
            const std::vector<Transformation>& kf = keyframes_[p];
            for (size_t i = 0; i + 1 < kf.size(); ++i) {
                for (int j = 1; j < 4; ++j) aabbMerge(bounds, aabbTransform(shapeAabb(hp.shape), objectToWorld(lerpTransformation(kf[i], kf[i + 1], 0.25f * float(j)))));
                aabbMerge(bounds, aabbTransform(shapeAabb(hp.shape), objectToWorld(kf[i + 1])));
            }
        }
        bounds.b[0] = bounds.b[0] + (-camera_pos);
        bounds.b[1] = bounds.b[1] + (-camera_pos);
        aabbCacheRadius(bounds);
        for (int i = 0; i < 4; ++i) {
            c_aabbs_[p].min[i] = bounds.b[0][i];
            c_aabbs_[p].max[i] = bounds.b[1][i];
        }
    }

    buildPropTree(finite_props_, solid_nodes_, solid_indices_);
    buildPropTree(unoccluding_props_, unoccluding_nodes_, unoccluding_indices_);
    if (!volume_props_.empty()) return -2;  // out of scope
    // Distant, Dome and Canopy props with uniform emission; sky models and image lights (emission maps) are row a20's rest

    // lights: scene.zig:402-497 (propPrepareSampling, static branch)
    mesh_lights_.clear();
    image_lights_.clear();
    portal_lights_.clear();
    light_links_.resize(lights_.size(), ZYGPU_NULL);
    std::fill(light_ids_.begin(), light_ids_.end(), ZYGPU_NULL);
    const uint32_t nl = uint32_t(lights_.size());
    c_light_aabbs_.resize(nl);
    c_light_cones_.resize(size_t(nl) * 4);
    c_light_two_sided_.resize(nl);
    c_light_infinite_.assign(nl, 0);
    c_light_prototype_.assign(nl, 0);
    // Scene.aabb() = the solid tree's bounds (scene.zig:173-175): scales the power of infinite lights (light.zig:67-77)
    AABB scene_bb = emptyAabb();
    if (!solid_nodes_.empty()) {
        const ZygpuBvhNode& r = solid_nodes_[0];
        scene_bb = {{vec(r.min[0], r.min[1], r.min[2], 0.f), vec(r.max[0], r.max[1], r.max[2], 0.f)}};
    }
    for (uint32_t li = 0; li < nl; ++li) {
        ZygpuLight&      l = lights_[li];
        const HostProp&  hp = props_[l.prop];
        const HostShape& s = shapes_[hp.shape];
        const uint32_t   p = hp.parts_start + l.part;
        light_ids_[p] = li;
        const ZygpuMaterial& m = materials_[material_ids_[p]];

        const bool is_mesh = ZYGPU_SHAPE_TRIANGLE_MESH == s.kind, is_distant = ZYGPU_SHAPE_DISTANT == s.kind;
        const bool is_sphere = ZYGPU_SHAPE_SPHERE == s.kind, is_dome = ZYGPU_SHAPE_DOME == s.kind, is_canopy = ZYGPU_SHAPE_CANOPY == s.kind;
        const bool is_infinite = is_distant || is_dome || is_canopy;
        const bool is_disk = ZYGPU_SHAPE_DISK == s.kind;
        if (ZYGPU_SHAPE_RECTANGLE != s.kind && !is_mesh && !is_infinite && !is_sphere && !is_disk) return -3;  // cube emitters: SURVEY §8(f)
        c_light_infinite_[li] = is_infinite ? 1 : 0;

        l.num_samples = m.emission_num_samples;  // light_material.zig:68-74 / substitute_material.zig:96-111
        l.two_sided = (m.flags & ZYGPU_MATF_TWO_SIDED) ? 1 : 0;
        c_light_two_sided_[li] = l.two_sided;
        l.sampler = is_mesh ? prepareMeshLight(s.mesh, l.part, material_ids_[p], 0 != l.two_sided) : ZYGPU_NULL;
        // scene.zig:430-470: an emission map makes the light a PropImage with its own sampler (light.zig:95-101)
        const bool sky_map = ZYGPU_NULL != sky_.sky && l.prop == sky_.sky;
        const bool image_light = ZYGPU_MAT_LIGHT == m.kind && ZYGPU_NULL != m.emission_image;
        l.light_class = ZYGPU_LIGHT_PROP;
        c_light_prototype_[li] = l.prototype;
        const uint32_t light_link = light_links_[li];
        if (image_light && ZYGPU_NULL != light_link) {
            // scene.zig:361-364: a linked light with a mapped emission is a Portal; Shape.samplePortalTo knows Rectangles only
            // (shape.zig:373-391: any other shape never yields a sample), and the link is a Dome (Dome.worldToImage)
            if (m.emission_image >= images_.size() || light_link >= nl) return -4;
            if (ZYGPU_SHAPE_RECTANGLE != s.kind || ZYGPU_SHAPE_DOME != shapes_[props_[lights_[light_link].prop].shape].kind) return -3;
            // the device reads the map where the ray points through the linked Dome's transformation at the frame's start
            if (ZYGPU_NULL != c_prop_frames_[lights_[light_link].prop] || ZYGPU_NULL != c_prop_frames_[l.prop]) return -3;
            l.sampler = preparePortalLight(m, frameStartTransformation(l.prop), frameStartTransformation(lights_[light_link].prop), light_link);
            l.light_class = ZYGPU_LIGHT_PORTAL_IMAGE;
        } else if (image_light) {
            if (m.emission_image >= images_.size()) return -4;
            if (is_mesh || is_distant) return -3;  // Shape.sampleMaterialTo has no case for them (shape.zig:348-371)
            l.sampler = prepareImageLight(s.kind, m, !sky_map);
            l.light_class = ZYGPU_LIGHT_PROP_IMAGE;
        }

        const Transformation& t = frameStartTransformation(l.prop);
        Transformation        rel = t;
        rel.position = t.position - camera_pos;

        // shape_sampler.zig:108-120: a mesh part brings its own box and cone, the analytic shapes use the shape's
        const AABB part_aabb = is_mesh ? mesh_lights_[l.sampler].aabb : shapeAabb(hp.shape);
        // shape.zig:115-120: Disk / Rectangle / Distant emit into a half space, the other analytic shapes everywhere
        const Vec4f part_cone = is_mesh ? mesh_lights_[l.sampler].cone : vec(0.f, 0.f, 1.f, (is_sphere || is_dome || is_canopy) ? -1.f : 1.f);
        AABB        bb;
        Vec4f       cone;
        if (ZYGPU_NULL == c_prop_frames_[l.prop]) {
            bb = aabbTransform(part_aabb, objectToWorld(rel));
            const Mat3x3 rot = quaternionToMat3x3(t.rotation);
            Mat3x3       rot_s = rot;  // rows carry the scale in w, as the reference's ComposedTransformation does
            for (int i = 0; i < 3; ++i) rot_s.r[i][3] = t.scale[i];
            const Vec4f tc = transformVector(rot_s, part_cone);
            cone = vec(tc[0], tc[1], tc[2], part_cone[3]);
        } else {
            // an animated light (scene.zig:445-488): the box bounds the part at the keyframes and at NumSteps - 1 steps between
            // them, the cone grows by merging.  Restated as written there: the keyframes' positions are taken as they are
            // (not relative to the camera), and every step merges the running cone with the running cone turned by the step's
            // rotation
            const std::vector<Transformation>& kf = keyframes_[l.prop];
            auto composed = [](const Transformation& k) { return objectToWorld(k); };
            bb = aabbTransform(part_aabb, composed(kf[0]));
            const Vec4f tc0 = transformVector(quaternionToMat3x3(kf[0].rotation), part_cone);
            cone = vec(tc0[0], tc0[1], tc0[2], part_cone[3]);
            auto step = [&](const Transformation& k) {
                aabbMerge(bb, aabbTransform(part_aabb, composed(k)));
                const Vec4f tc = transformVector(quaternionToMat3x3(k.rotation), cone);
                cone = coneMerge(cone, vec(tc[0], tc[1], tc[2], cone[3]));
            };
            for (size_t i = 0; i + 1 < kf.size(); ++i)
                for (int j = 1; j < 4; ++j) step(lerpTransformation(kf[i], kf[i + 1], 0.25f * float(j)));
            step(kf.back());
        }
        aabbCacheRadius(bb);
        for (int i = 0; i < 4; ++i) c_light_cones_[4 * li + i] = cone[i];

        // shape.zig:132-152: Rectangle area, TriangleMesh part area (triangle_mesh.zig:376-380), Distant solid angle (distant.zig:143-145)
        const float extent = is_canopy  ? 2.f * float(M_PI)
                             : is_dome  ? 4.f * float(M_PI)
                             : is_sphere ? (4.f * float(M_PI)) * ((0.5f * t.scale[0]) * (0.5f * t.scale[0]))
                             : is_disk  ? float(M_PI) * ((0.5f * t.scale[0]) * (0.5f * t.scale[0]))  // shape.zig:143
                             : is_distant ? (2.f * float(M_PI)) * (1.f - std::sqrt(1.f / (t.scale[0] * t.scale[0] + 1.f)))
                             : is_mesh  ? meshes_[s.mesh].part_areas[l.part] * (t.scale[0] * t.scale[1])
                                        : t.scale[0] * t.scale[1];
        Vec4f       emission = ZYGPU_LIGHT_PORTAL_IMAGE == l.light_class ? portal_lights_[l.sampler].average_emission
                               : image_light                             ? image_lights_[l.sampler].average_emission
                                                                         : vec(m.emission[0], m.emission[1], m.emission[2], 0.f);
        // emittance.zig:61-71 totalEmission; light.zig:67-77 power: infinite lights scale with the scene's extent
        Vec4f power = splat(0.f);
        if (extent > 0.f) power = m.emission_normalize ? emission : emission * splat(extent);
        if (is_infinite && !solid_nodes_.empty()) power = splat(squaredLength3(aabbExtent(scene_bb))) * power;
        if (l.prototype) power = splat(0.f);  // light.zig:64-67
        bb.b[0][3] = hmax3(power);

        for (int i = 0; i < 4; ++i) {
            c_light_aabbs_[li].min[i] = bb.b[0][i];
            c_light_aabbs_[li].max[i] = bb.b[1][i];
        }
    }

    LightTreeInput lin;
    lin.aabbs = c_light_aabbs_.data();
    lin.cones = c_light_cones_.data();
    lin.two_sided = c_light_two_sided_.data();
    lin.infinite = c_light_infinite_.data();
    lin.prototype = c_light_prototype_.data();
    lin.num_lights = nl;
    LightTreeBuilder ltb;
    ltb.build(lin, light_tree_);

    ZygpuSceneDesc& d = scene_desc_;
    std::memset(&d, 0, sizeof(d));
    d.props = c_props_.data();
    d.trafos = c_trafos_.data();
    d.aabbs = c_aabbs_.data();
    d.material_ids = material_ids_.data();
    d.light_ids = light_ids_.data();
    d.meshes = c_meshes_.data();
    d.materials = materials_.data();
    d.lights = lights_.data();
    d.light_aabbs = c_light_aabbs_.data();
    d.light_cones = c_light_cones_.data();
    d.solid = ZygpuPropTree{solid_nodes_.data(), solid_indices_.data(), uint32_t(solid_nodes_.size()),
                            uint32_t(solid_indices_.size())};
    d.unoccluding = ZygpuPropTree{unoccluding_nodes_.data(), unoccluding_indices_.data(),
                                  uint32_t(unoccluding_nodes_.size()), uint32_t(unoccluding_indices_.size())};
    d.light_tree = light_tree_.tree;
    c_mesh_lights_.resize(mesh_lights_.size());
    for (size_t i = 0; i < mesh_lights_.size(); ++i) {
        const HostMeshLight& ml = mesh_lights_[i];
        ZygpuMeshLight&      o = c_mesh_lights_[i];
        o.triangle_mapping = ml.triangle_mapping.data();
        o.tri_props = ml.tri_props.data();
        o.tree = ml.tree.tree;
        o.mesh = ml.mesh;
        o.part = ml.part;
        o.num_triangles = uint32_t(ml.triangle_mapping.size());
        o.two_sided = ml.two_sided ? 1 : 0;
        c_meshes_[ml.mesh].primitive_mapping = meshes_[ml.mesh].primitive_mapping.data();
    }
    d.mesh_lights = c_mesh_lights_.data();
    d.num_mesh_lights = uint32_t(c_mesh_lights_.size());
    d.infinite_props = infinite_props_.data();
    d.num_infinite_props = uint32_t(infinite_props_.size());
    c_images_.resize(images_.size());
    for (size_t i = 0; i < images_.size(); ++i)
        c_images_[i] = ZygpuImage{images_[i].pixels.data(), images_[i].width, images_[i].height, images_[i].format};
    d.images = c_images_.data();
    d.num_images = uint32_t(c_images_.size());
    ensureSrgbTable();
    d.srgb_to_float = srgb_to_float_.data();
    // the camera's aperture shape (camera/aperture.zig:22-44): its Distribution2D rides with the scene's image lights, as their last entry
    // (no light names it; ZygpuViewDesc.aperture_shape does)
    aperture_light_ = ZYGPU_NULL;
    if (ZYGPU_NULL != aperture_shape_image && lens_radius > 0.f) {
        const HostImage& img = images_[aperture_shape_image];
        image_lights_.emplace_back();
        HostImageLight& il = image_lights_.back();
        il.num_rows = uint32_t(img.height);
        il.total_weight = 1.f;
        std::vector<float> integrals(size_t(img.height)), row(size_t(img.width));
        for (int y = 0; y < img.height; ++y) {
            for (int x = 0; x < img.width; ++x) {  // Texture.image2D_1 (texture.zig:166-175)
                const size_t i = size_t(y) * size_t(img.width) + size_t(x);
                row[size_t(x)] = ZYGPU_IMAGE_BYTE1_UNORM == img.format ? float(img.pixels[i]) * (1.f / 255.f) : reinterpret_cast<const float*>(img.pixels.data())[i];
            }
            integrals[size_t(y)] = appendDistribution1D(row, il);
        }
        appendDistribution1D(integrals, il);
        aperture_light_ = uint32_t(image_lights_.size() - 1);
    }
    c_image_lights_.resize(image_lights_.size());
    for (size_t i = 0; i < image_lights_.size(); ++i) {
        const HostImageLight& il = image_lights_[i];
        c_image_lights_[i] = ZygpuImageLight{il.cdf.data(), il.lut.data(), il.rows.data(), il.num_rows, il.total_weight, il.sampler,
                                             uint32_t(il.cdf.size()), uint32_t(il.lut.size())};
    }
    d.image_lights = c_image_lights_.data();
    d.num_image_lights = uint32_t(c_image_lights_.size());
    c_portal_lights_.resize(portal_lights_.size());
    for (size_t i = 0; i < portal_lights_.size(); ++i)
        c_portal_lights_[i] = ZygpuPortalLight{portal_lights_[i].sat.data(), portal_lights_[i].func.data(), portal_lights_[i].dim, portal_lights_[i].light_link};
    d.portal_lights = c_portal_lights_.empty() ? nullptr : c_portal_lights_.data();
    d.num_portal_lights = uint32_t(c_portal_lights_.size());
    d.procedurals = procedurals_.empty() ? nullptr : procedurals_.data();
    d.num_procedurals = uint32_t(procedurals_.size());
    auto channels = [&](uint32_t image) {
        const uint32_t f = images_[image].format;
        return (ZYGPU_IMAGE_BYTE3_SRGB == f || ZYGPU_IMAGE_FLOAT3 == f) ? 3 : (ZYGPU_IMAGE_BYTE2 == f || ZYGPU_IMAGE_FLOAT2 == f) ? 2 : 1;
    };
    for (const ZygpuMaterial& m : materials_) {
        if (ZYGPU_MAT_SUBSTITUTE == m.kind || ZYGPU_MAT_GLASS == m.kind) {
            // image2D_1 reads one channel, image2D_2 (the normal maps) two (texture.zig:166-186)
            for (uint32_t t = 0; t < ZYGPU_STEX_COUNT; ++t) {
                const uint32_t image = m.tex[t].image;
                if (ZYGPU_NULL == image) continue;
                if (m.tex[t].sampler & ZYGPU_TEX_PROCEDURAL) {
                    if (image >= procedurals_.size()) return -4;
                    continue;
                }
                const int want = (ZYGPU_STEX_NORMAL == t || ZYGPU_STEX_COATING_NORMAL == t) ? 2 : 1;
                if (image >= images_.size() || want != channels(image)) return -4;
            }
        }
        if (ZYGPU_NULL != m.color_image && (m.color_sampler & ZYGPU_TEX_PROCEDURAL)) {
            if (m.color_image >= procedurals_.size()) return -4;
        } else
        if (ZYGPU_NULL != m.color_image && (m.color_image >= images_.size() || 3 != channels(m.color_image))) return -4;  // a colour map without its image
        if (ZYGPU_NULL != m.emission_image && m.emission_image < images_.size() && 3 != channels(m.emission_image)) return -4;
        if (ZYGPU_NULL != m.mask_image && (m.mask_image >= images_.size() || 1 != channels(m.mask_image))) return -4;  // image2D_1 reads one channel
    }
    for (const ZygpuProcedural& pr : procedurals_)  // a child is uniform, an image or an earlier procedural (any channel count: a fetch
        for (const ZygpuTexture& t : pr.child)      // of another width reads 0, texture.zig:166-219)
            if (ZYGPU_NULL != t.image && t.image >= ((t.sampler & ZYGPU_TEX_PROCEDURAL) ? procedurals_.size() : images_.size())) return -4;
    if (unsupported_integrator_) return -10;
    for (uint32_t mid : material_ids_) {
        const ZygpuMaterial& m = materials_[mid];
        if (m.flags & ZYGPU_MATF_UNSUPPORTED) return -9;
        // subsurface scattering (substitute_material.zig:71-74: a scattering volume inside the surface): volumes are out of scope
        if (ZYGPU_MAT_SUBSTITUTE == m.kind && m.attenuation_distance > 0.f) return -7;
        // an emitter under a clearcoat: evaluateRadiance attenuates by the coating (substitute_material.zig:283-293), not restated
        if (ZYGPU_MAT_SUBSTITUTE == m.kind && (m.flags & ZYGPU_MATF_EMISSIVE) && m.coating_thickness > 0.f) return -8;
    }
    for (const ZygpuMaterial& m : materials_) {
        if (ZYGPU_MAT_SUBSTITUTE != m.kind &&
            !(ZYGPU_MAT_GLASS == m.kind && (m.roughness > 0.f || ZYGPU_NULL != m.tex[ZYGPU_STEX_ROUGHNESS].image)))
            continue;
        const ggxlut::Tables& t = ggxlut::tables();  // computed once per process (ggx_lut.hpp)
        d.luts.e_m = t.e_m.data();
        d.luts.e_m_avg = t.e_m_avg.data();
        d.luts.e = t.e.data();
        d.luts.e_avg = t.e_avg.data();
        d.luts.e_s = t.e_s.data();
        break;
    }
    d.specular_threshold = specular_threshold;
    d.num_props = np;
    d.num_parts = uint32_t(material_ids_.size());
    d.num_meshes = uint32_t(c_meshes_.size());
    d.num_materials = uint32_t(materials_.size());
    d.num_lights = nl;
    d.keyframes = c_keyframes_.empty() ? nullptr : c_keyframes_.data();
    d.prop_frames = c_keyframes_.empty() ? nullptr : c_prop_frames_.data();
    d.num_keyframes = uint32_t(c_keyframes_.size());
    d.num_interpolation_frames = numInterpolationFrames();
    for (int i = 0; i < 4; ++i) d.origin[i] = camera_pos[i];
    d.sun_radiance = c_sun_radiance_.empty() ? nullptr : c_sun_radiance_.data();
    d.num_sun_radiance = uint32_t(c_sun_radiance_.size() / 4);

    // camera.update (camera_perspective.zig:79-122, mono / no focus point)
    ZygpuViewDesc& v = view_desc_;
    std::memset(&v, 0, sizeof(v));
    v.resolution[0] = resolution[0];
    v.resolution[1] = resolution[1];
    for (int i = 0; i < 4; ++i) v.crop[i] = crop[i];
    {
        const float fr0 = float(resolution[0]), fr1 = float(resolution[1]);
        const float ratio = fr1 / fr0;
        const float z = 1.f / std::tan(0.5f * fov);
        const Vec4f left_top = vec(-1.f, ratio, z, 0.f);
        const Vec4f right_top = vec(1.f, ratio, z, 0.f);
        const Vec4f left_bottom = vec(-1.f, -ratio, z, 0.f);
        const Vec4f d_x = (right_top - left_top) / splat(fr0);
        const Vec4f d_y = (left_bottom - left_top) / splat(fr1);
        for (int i = 0; i < 4; ++i) {
            v.left_top[i] = left_top[i];
            v.d_x[i] = d_x[i];
            v.d_y[i] = d_y[i];
            v.eye_offset[i] = 0.f;
        }
    }
    v.lens_radius = lens_radius;
    v.focus_distance = focus_distance;
    v.aperture_shape = aperture_light_;
    v.camera_trafo = c_trafos_[camera_entity];
    if (camera_entity < keyframes_.size() && !keyframes_[camera_entity].empty()) {
        const std::vector<Transformation>& kf = keyframes_[camera_entity];
        v.camera_num_keyframes = uint32_t(std::min<size_t>(kf.size(), ZYGPU_MAX_CAMERA_KEYFRAMES));
        for (int i = 0; i < 4; ++i) v.camera_origin[i] = camera_pos[i];
        for (uint32_t k = 0; k < v.camera_num_keyframes; ++k) {
            for (int i = 0; i < 4; ++i) {
                v.camera_keyframes[k].position[i] = kf[k].position[i];
                v.camera_keyframes[k].scale[i] = kf[k].scale[i];
                v.camera_keyframes[k].rotation[i] = kf[k].rotation[i];
            }
        }
    }
    v.frame_step = frame_step;
    v.frame_duration = frame_duration;
    v.shutter_configured = shutter_configured ? 1u : 0u;
    std::memcpy(v.shutter_cdf, shutter_cdf, sizeof(v.shutter_cdf));
    v.integrator = integrator;
    v.max_depth_surface = max_depth_surface;
    v.max_depth_volume = max_depth_volume;
    v.split_threshold = split_threshold;
    v.caustics_path = caustics_path ? 1u : 0u;
    v.regularize_roughness = regularize_roughness;
    v.sampler = sampler_kind;
    v.num_expected_samples = num_samples_per_pixel;
    v.filter_radius_int = sensor.filter_radius_int;
    v.filter_radius = sensor.filter_radius;
    std::memcpy(v.filter_samples, sensor.filter_samples, sizeof(v.filter_samples));
    v.clamp_emission = sensor.clamp_emission;
    v.clamp_direct = sensor.clamp_direct;
    v.clamp_indirect = sensor.clamp_indirect;
    v.exposure_factor = std::exp2(sensor.exposure);
    v.alpha_transparency = sensor.alpha_transparency ? 1 : 0;
    v.tonemapper = sensor.tonemapper;
    v.aov_slots = aov_slots;
    return 0;
}

}  // namespace zyg
