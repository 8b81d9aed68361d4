// On-disk formats of the reference, read on the host into the Engine:
//   take JSON        src/cli/take_loader.zig:27-283, src/core/take/take.zig:303-331
//   scene JSON       src/util/scene_loader.zig:115-326, 612-702 (materials by name, entities with children, shapes by
//                    type or file, transformations base/json.zig:165-245, visibility)
//   mesh JSON        src/core/scene/shape/shape_provider.zig:334-525 ("geometry": parts, vertices, indices)
//   mesh .sub        src/core/scene/shape/shape_provider.zig:527-805, 965-994 ("SUB\0", u64 json size, JSON header, binary
//                    vertex streams + indices, delta-encoded Int16 / Int32 or plain UInt16 / UInt32)
// File names resolve against the directory of the file that mentions them, then the take's directory (the mount stack of
// core/file/system.zig, reduced to these two levels).  Sky entities create the sun and the sky (engine.cpp createSky; the bake's tables are a
// procedural stand-in unless the client hands them over).  Refused by name (the load fails, nothing is skipped in silence):
// scatterers (their instances are props of the scene).  Instancers are flattened into props, animations resampled per frame by the engine.  glTF 2 scene files (.gltf + .bin,
// materials without image textures) are read as util/gltf_loader.zig reads them.  Not read: gzip-compressed files.
#pragma once

#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "engine.hpp"
#include "image_writers.hpp"

namespace zyg {

class TakeLoader {
  public:
    explicit TakeLoader(Engine& e) : engine_(e) {}

    std::string                 error;
    std::vector<imgw::Exporter> exporters;

    bool loadTake(const std::string& path) {
        JsonValue root;
        if (!readJson(path, root) || JsonValue::Object != root.type) return fail("cannot read take \"" + path + "\"");
        take_dir_ = parentDirectory(path);
        engine_.mounts.push_back(take_dir_);  // (file-backed textures: the take's directory, then the scene file's, newest first)
        const JsonValue* scene = root.get("scene");
        if (!scene || JsonValue::String != scene->type || scene->str.empty()) return fail("the take names no scene");

        const JsonValue *integrator = nullptr, *post = nullptr, *exporter = nullptr, *sensor = nullptr;
        for (const auto& kv : root.obj) {
            if (kv.first == "sensor") sensor = &kv.second;
            else if (kv.first == "aov") loadAov(kv.second);
            else if (kv.first == "camera") loadCamera(kv.second);
            else if (kv.first == "cameras" && JsonValue::Array == kv.second.type && !kv.second.arr.empty()) loadCamera(kv.second.arr[0]);
            else if (kv.first == "export") exporter = &kv.second;
            else if (kv.first == "integrator") integrator = &kv.second;
            else if (kv.first == "post") post = &kv.second;
            else if (kv.first == "sampler") loadSampler(kv.second);
        }
        if (ZYGPU_NULL == engine_.camera_entity) return fail("the take defines no Perspective camera");
        if (!error.empty()) return false;
        if (integrator) engine_.loadIntegrators(*integrator);
        {
            // the post-processor chain's tonemapper belongs to the sensor here (Engine::loadSensor reads both)
            JsonValue merged = sensor ? *sensor : JsonValue();
            if (JsonValue::Object != merged.type) {
                merged = JsonValue();
                merged.type = JsonValue::Object;
            }
            if (post && JsonValue::Array == post->type)
                for (const JsonValue& pp : post->arr)
                    if (const JsonValue* t = pp.get("tonemapper")) merged.obj.emplace_back("tonemapper", *t);
            engine_.loadSensor(merged);
        }
        if (exporter) loadExporters(*exporter);
        if (!loadSceneFile(scene->str, take_dir_, identity())) return false;
        return true;
    }

  private:
    struct Node {  // scene_graph.zig Node.world_trafo
        Transformation world;
    };

    Engine&                         engine_;
    int                             parent_node_ = -1;  // the animated graph node (Engine::createGraphNode) the entities being loaded hang below, or -1
    std::string                     take_dir_;
    std::map<std::string, uint32_t> materials_by_name_, shapes_by_file_;

    bool fail(const std::string& what) {
        error = what;
        return false;
    }
    static Transformation identity() { return {splat(0.f), splat(1.f), vec(0.f, 0.f, 0.f, 1.f)}; }
    static std::string    parentDirectory(const std::string& p) {
        const size_t at = p.find_last_of('/');
        return std::string::npos == at ? std::string() : p.substr(0, at);
    }
    static bool readFile(const std::string& path, std::string& out) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        std::ostringstream ss;
        ss << f.rdbuf();
        out = ss.str();
        return true;
    }
    static bool readJson(const std::string& path, JsonValue& v) {
        std::string text;
        return readFile(path, text) && JsonParser::parse(text.c_str(), v);
    }
    std::string resolve(const std::string& name, const std::string& dir) const {
        if (!name.empty() && '/' == name[0]) return name;
        for (const std::string& d : {dir, take_dir_}) {
            const std::string p = d.empty() ? name : d + "/" + name;
            if (std::ifstream(p)) return p;
        }
        return name;
    }

    // ---- base/json.zig:165-245
    static Vec4f readVec3(const JsonValue& v, Vec4f def) {
        if (JsonValue::Array != v.type || v.arr.size() < 3) return def;
        return vec(v.arr[0].f32(), v.arr[1].f32(), v.arr[2].f32(), 0.f);
    }
    static Mat3x3 rotationX(float a) {
        const float c = std::cos(a), s = std::sin(a);
        return {{vec(1.f, 0.f, 0.f), vec(0.f, c, -s), vec(0.f, s, c)}};
    }
    static Mat3x3 rotationY(float a) {
        const float c = std::cos(a), s = std::sin(a);
        return {{vec(c, 0.f, s), vec(0.f, 1.f, 0.f), vec(-s, 0.f, c)}};
    }
    static Mat3x3 rotationZ(float a) {
        const float c = std::cos(a), s = std::sin(a);
        return {{vec(c, -s, 0.f), vec(s, c, 0.f), vec(0.f, 0.f, 1.f)}};
    }
    static Mat3x3 mul(const Mat3x3& a, const Mat3x3& b) {  // matrix3x3.zig:99-111
        Mat3x3 o;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o.r[i][j] = a.r[i][0] * b.r[0][j] + a.r[i][1] * b.r[1][j] + a.r[i][2] * b.r[2][j];
        for (int i = 0; i < 3; ++i) o.r[i][3] = 0.f;
        return o;
    }
    static void readTransformation(const JsonValue& v, Transformation& t) {
        if (JsonValue::Object == v.type) {
            Vec4f up = vec(0.f, 1.f, 0.f), look_at = vec(0.f, 0.f, 1.f);
            bool  look = false;
            for (const auto& kv : v.obj) {
                if (kv.first == "position") t.position = readVec3(kv.second, t.position);
                else if (kv.first == "scale") t.scale = readVec3(kv.second, t.scale);
                else if (kv.first == "rotation") {
                    const Vec4f d = readVec3(kv.second, splat(0.f));
                    const float k = float(M_PI / 180.0);
                    t.rotation = quaternionFromMat3x3(mul(mul(rotationZ(d[2] * k), rotationX(d[0] * k)), rotationY(d[1] * k)));
                } else if (kv.first == "look_at") {
                    look_at = readVec3(kv.second, look_at);
                    look = true;
                } else if (kv.first == "up") {
                    up = readVec3(kv.second, up);
                }
            }
            if (look) {
                const Vec4f dir = normalize3(look_at - t.position);
                const Vec4f right = -cross3(dir, up);
                t.rotation = quaternionFromMat3x3(Mat3x3{{right, up, dir}});
            }
        } else if (JsonValue::Array == v.type && v.arr.size() >= 16) {
            Mat4x4 m;
            for (int r = 0; r < 4; ++r) m.r[r] = vec(v.arr[4 * r].f32(), v.arr[4 * r + 1].f32(), v.arr[4 * r + 2].f32(), v.arr[4 * r + 3].f32());
            Mat3x3 basis;
            decompose(m, basis, t.scale, t.position);
            t.position[3] = 0.f;
            t.rotation = quaternionFromMat3x3(basis);
        }
    }
    static Transformation transform(const Transformation& parent, const Transformation& child) { return transformBy(parent, child); }
    void setWorld(uint32_t entity, const Transformation& world) { engine_.setWorldTransformation(entity, world); }

    // View.loadAOV (take.zig:106-129): {"Albedo": true, "Depth": true, ...}, as su_aovs_create reads it
    void loadAov(const JsonValue& v) {
        static const char* const names[ZYGPU_AOV_NUM_CLASSES] = {"Albedo", "Depth", "MaterialId", "GeometricNormal", "ShadingNormal",
                                                                 "Roughness", "Emission", "Direct", "Indirect"};
        if (JsonValue::Object != v.type) return;
        for (const auto& kv : v.obj)
            for (uint32_t k = 0; k < ZYGPU_AOV_NUM_CLASSES; ++k)
                if (kv.first == names[k]) {
                    if (JsonValue::Bool == kv.second.type && kv.second.b) engine_.aov_slots |= 1u << k;
                    else engine_.aov_slots &= ~(1u << k);
                }
    }
    // ---- take_loader.zig:92-151, 232-243
    void loadCamera(const JsonValue& v) {
        if (JsonValue::Object != v.type) return;
        const JsonValue* cam = v.get("Perspective");
        if (!cam || JsonValue::Object != cam->type) return;  // Orthographic cameras are not provided
        int32_t res[2] = {0, 0};
        if (const JsonValue* r = cam->get("resolution"))
            if (JsonValue::Array == r->type && r->arr.size() >= 2) {
                res[0] = int32_t(r->arr[0].f32());
                res[1] = int32_t(r->arr[1].f32());
            }
        engine_.cameraCreate(uint32_t(res[0]), uint32_t(res[1]));
        if (const JsonValue* c = cam->get("crop"))
            if (JsonValue::Array == c->type && c->arr.size() >= 4)
                for (int i = 0; i < 4; ++i) engine_.crop[i] = int32_t(c->arr[i].f32());
        if (const JsonValue* p = cam->get("parameters"))
            if (!engine_.loadCameraParameters(*p)) error = engine_.camera_error;  // refused by name ("blades", "stereo", a shape that cannot be read)
        if (const JsonValue* sh = cam->get("shutter")) {
            // loadShutter (take_loader.zig:153-191): the frame's step (frame_step in seconds, or frames_per_second), the shutter's
            // opening, closing and slope (two cubic Bezier segments), motion_blur off = a frame of no duration.  Base.setShutter runs
            // whenever the key is there, with the defaults too: a configured distribution maps a time sample even if it is flat
            float       open = 0.f, close = 1.f, slope[8] = {};
            uint32_t    num_slope = 0;
            bool        motion_blur = true;
            const double units_per_second = 705600000.0;  // Scene.UnitsPerSecond
            if (JsonValue::Object == sh->type)
                for (const auto& kv : sh->obj) {
                    if (kv.first == "frame_step") engine_.frame_step = uint64_t(std::llround(units_per_second * (JsonValue::Number == kv.second.type ? kv.second.num : 0.0)));
                    else if (kv.first == "frames_per_second") {
                        const double fps = JsonValue::Number == kv.second.type ? kv.second.num : 0.0;
                        engine_.frame_step = 0.0 == fps ? 0u : uint64_t(std::llround(units_per_second / fps));
                    } else if (kv.first == "open") open = kv.second.f32(0.f);
                    else if (kv.first == "close") close = kv.second.f32(1.f);
                    else if (kv.first == "slope" && JsonValue::Array == kv.second.type) {
                        num_slope = uint32_t(std::min<size_t>(8, kv.second.arr.size()));
                        for (uint32_t i = 0; i < num_slope; ++i) slope[i] = kv.second.arr[i].f32(0.f);
                    } else if (kv.first == "motion_blur") motion_blur = JsonValue::Bool == kv.second.type ? kv.second.b : motion_blur;
                }
            engine_.setShutter(open, close, slope, num_slope);
            engine_.frame_duration = motion_blur ? engine_.frame_step : 0u;
            // (the view block holds ZYGPU_MAX_CAMERA_KEYFRAMES keyframes of a moving camera: frames longer than that many ticks of 1/60 s
            // are refused by name)
            if (engine_.numInterpolationFrames() > ZYGPU_MAX_CAMERA_KEYFRAMES)
                error = "a camera \"shutter\" whose frame spans more than " + std::to_string(ZYGPU_MAX_CAMERA_KEYFRAMES - 1) + " ticks of 1/60 s";
        }
        Transformation t = identity();
        if (const JsonValue* tr = cam->get("transformation")) readTransformation(*tr, t);
        setWorld(engine_.camera_entity, t);
        if (const JsonValue* av = cam->get("animation")) {  // take_loader.zig:142: the camera's keyframes, below the (identity) root
            const Transformation root = identity();
            int                  animation = -1;
            if (loadAnimation(*av, t, &root, animation) && animation >= 0 &&
                engine_.createGraphNode(engine_.camera_entity, -1, animation, t) < 0)
                fail("the camera's animation could not be created");
        }
    }
    void loadSampler(const JsonValue& v) {
        if (JsonValue::Object != v.type) return;
        engine_.sampler_kind = ZYGPU_SAMPLER_SOBOL;
        for (const auto& kv : v.obj) {
            engine_.num_samples_per_pixel = kv.second.memberU32("samples_per_pixel", 1);
            if (kv.first == "Random") {
                engine_.sampler_kind = ZYGPU_SAMPLER_RANDOM;
                return;
            }
        }
    }
    void loadExporters(const JsonValue& v) {  // take.zig:303-331
        exporters.clear();
        if (JsonValue::Object != v.type) return;
        for (const auto& kv : v.obj) {
            if (kv.first != "Image") continue;
            imgw::Exporter   e;
            const JsonValue* f = kv.second.get("format");
            const std::string format = (f && JsonValue::String == f->type) ? f->str : "PNG";
            if ("EXR" == format) {
                e.format = imgw::Exporter::EXR;
                e.half = 16 == kv.second.memberU32("bitdepth", 16);
            } else if ("RGBE" == format) {
                e.format = imgw::Exporter::RGBE;
            } else {
                e.error_diffusion = kv.second.memberBool("error_diffusion", false);
            }
            exporters.push_back(e);
        }
    }

    // ---- scene_loader.zig:132-230
    bool loadSceneFile(const std::string& name, const std::string& dir, const Transformation& parent) {
        const std::string path = resolve(name, dir);
        JsonValue         root;
        if (!readJson(path, root) || JsonValue::Object != root.type) return fail("cannot read scene \"" + path + "\"");
        const std::string                       here = parentDirectory(path);
        struct Mount {  // scene_loader.zig:167-168: fs.pushMount(parentDirectory) ... defer popMount
            Engine& e;
            Mount(Engine& e_, const std::string& d) : e(e_) { e.mounts.push_back(d); }
            ~Mount() { e.mounts.pop_back(); }
        } mount(engine_, here);
        if (isGltf(root)) return loadGltf(root, here, parent);  // scene_loader.zig:170-174: a glTF 2 document is tried first
        std::map<std::string, const JsonValue*> local_materials;
        if (const JsonValue* ms = root.get("materials"))
            if (JsonValue::Array == ms->type)
                for (const JsonValue& m : ms->arr)
                    if (const JsonValue* n = m.get("name"))
                        if (JsonValue::String == n->type) local_materials[n->str] = &m;
        if (const JsonValue* es = root.get("entities")) return loadEntities(*es, parent, local_materials, here);
        return true;
    }
    bool loadEntities(const JsonValue& v, const Transformation& parent, const std::map<std::string, const JsonValue*>& local_materials,
                      const std::string& dir) {
        if (JsonValue::Array != v.type) return true;
        for (const JsonValue& e : v.arr) {
            if (JsonValue::Object != e.type) continue;
            if (const JsonValue* file = e.get("file")) {
                if (JsonValue::String == file->type && !loadSceneFile(file->str, dir, parent)) return false;
                continue;
            }
            Transformation world;
            int            node = -1;
            scatterer = false;
            if (!loadEntity(e, parent, local_materials, dir, world, node)) {
                if (!error.empty()) return false;  // something this build cannot render: never skipped in silence
                continue;                          // scene_loader.zig:220 `catch continue` (an entity without type or shape)
            }
            const bool is_scatterer = scatterer;
            if (const JsonValue* children = e.get("entities")) {
                const int outer = parent_node_;
                parent_node_ = node;  // (-1 below a static entity of a static parent)
                const bool ok = loadEntities(*children, world, local_materials, dir);
                parent_node_ = outer;
                if (!ok) return false;
            }
            if (is_scatterer && !instantiate(e, world, local_materials, dir)) return false;  // (behind the children, scene_loader.zig:219-227)
        }
        return true;
    }
    bool scatterer = false;  // loadEntity: the entity it has just read is a "Scatterer"
    // What the reference's loader reads and this build does not render fails the load by name: an entity type (scene_loader.zig:
    // 252-264), or a key that changes the image (`animation`, scene_graph.zig:184-203).  su_ext_load_take logs the message at error
    // level and returns -1.
    bool unsupported(const std::string& what) {
        return fail("the scene uses " + what + ", which this build does not render");
    }
    bool loadEntity(const JsonValue& v, const Transformation& parent, const std::map<std::string, const JsonValue*>& local_materials,
                    const std::string& dir, Transformation& world, int& node) {
        node = -1;
        const JsonValue* type = v.get("type");
        if (!type || JsonValue::String != type->type) return false;  // Error.UndefinedType
        uint32_t entity = ZYGPU_NULL, light_link = ZYGPU_NULL;
        bool     is_light = false;
        if (type->str == "Light") {
            entity = loadProp(v, local_materials, dir, true);
            if (ZYGPU_NULL == entity) return false;  // Error.UndefinedShape
            is_light = true;
        } else if (type->str == "Prop") {
            entity = loadProp(v, local_materials, dir, false);
            if (ZYGPU_NULL == entity) return false;
        } else if (type->str == "Sky") {  // loadSky (scene_loader.zig:704-712): the entity is the sky's sun
            const int32_t sun = engine_.createSky();
            if (sun < 0) return false;
            if (const JsonValue* parameters = v.get("parameters")) engine_.setSkyParameters(*parameters);
            entity = uint32_t(sun);
        } else if (type->str == "Portal") {  // loadPortal (scene_loader.zig:371-397): a window onto light `light`
            const JsonValue* sv = v.get("shape");
            if (!sv || JsonValue::Object != sv->type) return false;  // Error.UndefinedShape
            const uint32_t shape = loadShape(*sv, dir);
            if (ZYGPU_NULL == shape) return false;
            light_link = v.memberU32("light", 0);
            const int32_t p = engine_.createPortal(shape, light_link);
            if (p < 0) return false;  // Error.UndefinedLight
            entity = uint32_t(p);
            is_light = true;
        } else if (type->str == "Instancer") {
            return loadInstancer(v, parent, local_materials, dir, world);
        } else if (type->str == "Scatterer") {
            // loadScatterer (scene_loader.zig:561-610, called once the node's children are loaded: 225-227): a node without an entity of
            // its own; its prototypes and instances are an Instancer's, the instances plain props of the scene below this node
            if (v.get("animation") || parent_node_ >= 0) return unsupported("an animated \"Scatterer\"");
            scatterer = true;
        }  // any other type is a node without an entity: its transformation applies to its children (scene_loader.zig:241-264)
        Transformation local = identity();
        if (const JsonValue* t = v.get("transformation")) readTransformation(*t, local);
        world = transform(parent, local);
        // scene_graph.zig:178-226: an entity with an "animation", and everything below it, is a node of the animated graph: its
        // world transformation at the frame's ticks = its parent node's frames transforming its own (keyframes resampled, or the
        // static transformation).  Below a static parent the keyframes take the parent's transformation in (animation_loader.zig).
        int animation = -1;
        if (const JsonValue* av = v.get("animation")) {
            if (!loadAnimation(*av, local, parent_node_ < 0 ? &parent : nullptr, animation)) return false;
        }
        if (animation >= 0 || parent_node_ >= 0) {
            if (ZYGPU_NULL != entity || v.get("entities")) {  // (a childless node without an entity moves nothing)
                node = engine_.createGraphNode(entity, parent_node_, animation, local);
                if (node < 0) return fail("the animated entity could not be created");
            }
        }
        if (ZYGPU_NULL == entity) return true;
        setWorld(entity, world);
        bool in_reflection = true;
        if (const JsonValue* vis = v.get("visibility")) {  // setVisibility (scene_loader.zig:612-619)
            in_reflection = vis->memberBool("in_reflection", true);
            engine_.setVisibility(entity, vis->memberBool("in_camera", true), in_reflection, vis->memberBool("in_sss", false),
                                  vis->memberBool("shadow_catcher_light", false));
        }
        if (v.get("shadow_catcher")) engine_.setShadowCatcher(entity);  // scene_loader.zig:312-313
        if (is_light && in_reflection) engine_.createLight(entity, light_link);
        return true;
    }
    struct PropSource {  // what loadProp reads of a Prop / Light entity (scene_loader.zig:321-369)
        uint32_t              shape = ZYGPU_NULL;
        std::vector<uint32_t> materials;
        bool                  unoccluding = false;
    };
    bool readProp(const JsonValue& v, const std::map<std::string, const JsonValue*>& local_materials, const std::string& dir,
                  bool unoccluding_default, PropSource& out) {
        const JsonValue* sv = v.get("shape");
        if (!sv || JsonValue::Object != sv->type) return false;
        out.shape = loadShape(*sv, dir);
        if (ZYGPU_NULL == out.shape) return false;
        const uint32_t num_materials = engine_.shapeNumMaterials(out.shape);
        if (const JsonValue* ms = v.get("materials"))
            if (JsonValue::Array == ms->type)
                for (size_t i = 0; i < ms->arr.size() && i < num_materials; ++i)
                    out.materials.push_back(JsonValue::String == ms->arr[i].type ? loadMaterial(ms->arr[i].str, local_materials, dir)
                                                                                  : engine_.fallbackMaterial());
        while (out.materials.size() < num_materials) out.materials.push_back(engine_.fallbackMaterial());
        out.unoccluding = !v.memberBool("occluding", !unoccluding_default);
        return true;
    }
    uint32_t createProp(const PropSource& src) {
        const int32_t p = engine_.createPropShape(src.shape, src.materials.data(), uint32_t(src.materials.size()), src.unoccluding);
        return p < 0 ? ZYGPU_NULL : uint32_t(p);
    }
    uint32_t loadProp(const JsonValue& v, const std::map<std::string, const JsonValue*>& local_materials, const std::string& dir,
                      bool unoccluding_default) {
        PropSource src;
        return readProp(v, local_materials, dir, unoccluding_default, src) ? createProp(src) : ZYGPU_NULL;
    }

    // An Instancer (scene_loader.zig:401-560, prop/instancer.zig): prototypes (entities that are not rendered themselves) and
    // instances = (prototype, transformation in the instancer's space).  The reference gives the instancer a prop tree of its own,
    // one more level under the scene's; here every instance becomes a prop of the scene with the composed transformation
    // instancer o instance (what Instancer.intersect leaves in isec.trafo), sharing the prototype's shape and materials: the
    // scene's own tree (or prop table) takes the place of the instancer's.  Prototypes are Props with a shape; `source` names a
    // file with the same members (and its own `materials`).
    bool loadInstancer(const JsonValue& v, const Transformation& parent, const std::map<std::string, const JsonValue*>& local_materials,
                       const std::string& dir, Transformation& world) {
        if (v.get("animation") || parent_node_ >= 0) return unsupported("an animated \"Instancer\"");
        Transformation local = identity();
        if (const JsonValue* t = v.get("transformation")) readTransformation(*t, local);
        world = transform(parent, local);
        if (const JsonValue* source = v.get("source")) {
            if (JsonValue::String != source->type) return false;
            const std::string path = resolve(source->str, dir);
            JsonValue         root;
            if (!readJson(path, root) || JsonValue::Object != root.type) return fail("cannot read instancer \"" + path + "\"");
            std::map<std::string, const JsonValue*> file_materials;
            if (const JsonValue* ms = root.get("materials"))
                if (JsonValue::Array == ms->type)
                    for (const JsonValue& m : ms->arr)
                        if (const JsonValue* n = m.get("name"))
                            if (JsonValue::String == n->type) file_materials[n->str] = &m;
            return instantiate(root, world, file_materials, parentDirectory(path));
        }
        return instantiate(v, world, local_materials, dir);
    }
    bool instantiate(const JsonValue& v, const Transformation& world, const std::map<std::string, const JsonValue*>& local_materials,
                     const std::string& dir) {
        struct Prototype {
            PropSource       src;
            const JsonValue* visibility = nullptr;
            bool             shadow_catcher = false;
        };
        std::vector<Prototype> prototypes;
        if (const JsonValue* ps = v.get("prototypes"))
            if (JsonValue::Array == ps->type)
                for (const JsonValue& pv : ps->arr) {
                    const JsonValue* type = pv.get("type");
                    if (!type || JsonValue::String != type->type) continue;  // loadEntity's Error.UndefinedType: `catch continue`
                    if (type->str != "Prop") return unsupported("an Instancer prototype (or a Scatterer's) of type \"" + type->str + "\"");
                    if (pv.get("animation") || pv.get("entities")) return unsupported("an Instancer prototype (or a Scatterer's) with an animation or children");
                    Prototype p;
                    if (!readProp(pv, local_materials, dir, false, p.src)) continue;  // Error.UndefinedShape
                    p.visibility = pv.get("visibility");
                    p.shadow_catcher = nullptr != pv.get("shadow_catcher");
                    prototypes.push_back(std::move(p));
                }
        const JsonValue* instances = v.get("instances");
        if (!instances || JsonValue::Object != instances->type) return true;  // an instancer without instances renders nothing
        const JsonValue *indices = instances->get("prototypes"), *trafos = instances->get("transformations");
        if (!indices || !trafos || JsonValue::Array != indices->type || JsonValue::Array != trafos->type) return true;
        if (prototypes.empty()) return true;
        const size_t n = std::min(indices->arr.size(), trafos->arr.size());
        for (size_t i = 0; i < n; ++i) {
            size_t index = size_t(std::max(0.f, indices->arr[i].f32()));
            if (index >= prototypes.size()) index = 0;
            const Prototype& proto = prototypes[index];
            Transformation   t = identity();
            readTransformation(trafos->arr[i], t);
            const uint32_t prop = createProp(proto.src);
            if (ZYGPU_NULL == prop) return fail("an Instancer's instance could not be created");
            setWorld(prop, transform(world, t));
            if (proto.visibility)
                engine_.setVisibility(prop, proto.visibility->memberBool("in_camera", true), proto.visibility->memberBool("in_reflection", true),
                                      proto.visibility->memberBool("in_sss", false), proto.visibility->memberBool("shadow_catcher_light", false));
            if (proto.shadow_catcher) engine_.setShadowCatcher(prop);
        }
        return true;
    }

    // animation_loader.zig:13-132: "keyframes" (objects with "time" in seconds and "transformation") or "transformations" + "times";
    // a keyframe without a time follows the previous one at 1 / frames_per_second.  The entity's own transformation is each
    // keyframe's default; a static parent's applies to the keyframes that name a transformation (`parent`; null below an animated parent,
    // whose frames apply at every compile instead).
    bool loadAnimation(const JsonValue& v, const Transformation& default_trafo, const Transformation* parent, int& animation) {
        animation = -1;
        if (JsonValue::Object != v.type) return true;
        const double   units_per_second = 705600000.0;  // Scene.UnitsPerSecond
        const float    fps = v.memberF32("frames_per_second", 0.f);
        const uint64_t frame_step = fps > 0.f ? uint64_t(std::llround(units_per_second / double(fps))) : 0u;
        auto           absolute = [&](const JsonValue& t) { return uint64_t(std::llround(units_per_second * (JsonValue::Number == t.type ? t.num : 0.0))); };
        std::vector<uint64_t>       times;
        std::vector<Transformation> frames;
        uint64_t                    current = 0;
        if (const JsonValue* keyframes = v.get("keyframes")) {
            if (JsonValue::Array != keyframes->type) return true;
            for (const JsonValue& k : keyframes->arr) {
                Transformation t = default_trafo;
                uint64_t       time = current;
                if (JsonValue::Object == k.type) {
                    if (const JsonValue* tv = k.get("time")) time = absolute(*tv);
                    if (const JsonValue* tr = k.get("transformation")) {
                        readTransformation(*tr, t);
                        if (parent) t = transform(*parent, t);
                    }
                }
                times.push_back(time);
                frames.push_back(t);
                current += frame_step;
            }
        } else if (const JsonValue* trafos = v.get("transformations")) {
            if (JsonValue::Array != trafos->type) return true;
            const JsonValue* tv = v.get("times");
            for (size_t i = 0; i < trafos->arr.size(); ++i) {
                Transformation t = default_trafo;
                uint64_t       time = current;
                if (tv && JsonValue::Array == tv->type && i < tv->arr.size()) time = absolute(tv->arr[i]);
                readTransformation(trafos->arr[i], t);
                if (parent) t = transform(*parent, t);
                times.push_back(time);
                frames.push_back(t);
                current += frame_step;
            }
        } else {
            return true;
        }
        if (times.empty()) return true;
        animation = engine_.createAnimation(times, frames);
        return animation >= 0 || fail("the entity's animation could not be created");
    }
    uint32_t loadMaterial(const std::string& name, const std::map<std::string, const JsonValue*>& local_materials, const std::string& dir) {
        const auto cached = materials_by_name_.find(name);
        if (materials_by_name_.end() != cached) return cached->second;
        uint32_t   id = ZYGPU_NULL;
        const auto local = local_materials.find(name);
        if (local_materials.end() != local) {
            id = engine_.createMaterial(ZYGPU_NULL, *local->second);
        } else {
            JsonValue         file;
            const std::string path = resolve(name, dir);
            if (readJson(path, file)) {
                engine_.mounts.push_back(parentDirectory(path));
                id = engine_.createMaterial(ZYGPU_NULL, file);
                engine_.mounts.pop_back();
            }
        }
        if (ZYGPU_NULL == id) id = engine_.fallbackMaterial();
        materials_by_name_[name] = id;
        return id;
    }
    uint32_t loadShape(const JsonValue& v, const std::string& dir) {  // scene_loader.zig:630-668
        if (const JsonValue* t = v.get("type")) {
            if (JsonValue::String != t->type) return ZYGPU_NULL;
            static const char* names[] = {"Canopy", "Cube", "Disk", "Distant", "Dome", "Rectangle", "Sphere"};  // manager.zig:36-44
            for (uint32_t i = 0; i < 7; ++i)
                if (t->str == names[i]) return i;
            return ZYGPU_NULL;
        }
        const JsonValue* f = v.get("file");
        if (!f || JsonValue::String != f->type) return ZYGPU_NULL;
        const std::string path = resolve(f->str, dir);
        const auto        cached = shapes_by_file_.find(path);
        if (shapes_by_file_.end() != cached) return cached->second;
        const uint32_t shape = loadMeshFile(path);
        shapes_by_file_[path] = shape;
        return shape;
    }

    // ---- util/gltf_loader.zig: a scene file that is a glTF 2 document (its "asset" has a version >= 2).  As the reference reads it:
    // every entry of "nodes" on its own (no "children"), a node with a "mesh" becomes a prop (its primitives the mesh's parts), a
    // node with a "camera" adds a camera the take does not use (the take's own comes first); translations, rotations, positions,
    // normals mirrored in z and the triangles' winding turned (right-handed -> left-handed); accessors / bufferViews / buffers with
    // external .bin files, UInt16 / UInt32 indices, strided vertex attributes; materials: pbrMetallicRoughness -> Substitute (base
    // colour, roughness, metallic, emissiveFactor, KHR_materials_ior / _clearcoat; _transmission / _volume set an attenuation
    // distance, which makes the Substitute a scattering volume: refused at compile like every other); base colour and normal
    // textures as "file" descriptors (PNG; engine.cpp loadImageFile); metallicRoughness / emissive / clearcoat textures refused by name.
    static bool isGltf(const JsonValue& root) {
        const JsonValue* asset = root.get("asset");
        if (!asset || JsonValue::Object != asset->type) return false;
        const JsonValue* version = asset->get("version");
        return version && JsonValue::String == version->type && !version->str.empty() && version->str[0] >= '2' && version->str[0] <= '9';
    }
    struct GltfAccessor {
        uint32_t buffer_view = 0, byte_offset = 0, component_type = 0, count = 0;
    };
    struct GltfView {
        uint32_t buffer = 0, byte_length = 0, byte_offset = 0, byte_stride = 0;
    };
    struct Gltf {
        std::vector<GltfAccessor> accessors;
        std::vector<GltfView>     views;
        std::vector<std::string>  buffers;  // file contents, read on first use
        std::vector<std::string>  uris;
        std::vector<uint32_t>     materials;
        std::string               dir;
    };
    static JsonValue jnum(double x) {
        JsonValue v;
        v.type = JsonValue::Number;
        v.num = x;
        return v;
    }
    static JsonValue jarr3(const Vec4f& c) {
        JsonValue v;
        v.type = JsonValue::Array;
        for (int i = 0; i < 3; ++i) v.arr.push_back(jnum(c[i]));
        return v;
    }
    static JsonValue jobj() {
        JsonValue v;
        v.type = JsonValue::Object;
        return v;
    }
    bool gltfBuffer(Gltf& g, uint32_t index, const std::string*& out) {
        if (index >= g.uris.size()) return fail("the glTF scene names a buffer it does not have");
        if (g.buffers[index].empty() && !readFile(resolve(g.uris[index], g.dir), g.buffers[index]))
            return fail("cannot read the glTF buffer \"" + g.uris[index] + "\"");
        out = &g.buffers[index];
        return true;
    }
    template <typename T>
    bool gltfRead(Gltf& g, uint32_t accessor, uint32_t elements, std::vector<T>& out) {  // loadVertexAttributes / loadIndices
        if (accessor >= g.accessors.size()) return fail("the glTF scene names an accessor it does not have");
        const GltfAccessor& a = g.accessors[accessor];
        if (a.buffer_view >= g.views.size()) return fail("the glTF scene names a bufferView it does not have");
        const GltfView&    view = g.views[a.buffer_view];
        const std::string* data = nullptr;
        if (!gltfBuffer(g, view.buffer, data)) return false;
        const size_t begin = size_t(a.byte_offset) + view.byte_offset, element_bytes = sizeof(T) * elements;
        const size_t stride = view.byte_stride > element_bytes ? view.byte_stride : element_bytes;
        if (a.count && begin + stride * (a.count - 1) + element_bytes > data->size()) return fail("a glTF accessor reaches past its buffer");
        const size_t first = out.size();
        out.resize(first + size_t(a.count) * elements);
        for (uint32_t i = 0; i < a.count; ++i) std::memcpy(out.data() + first + size_t(i) * elements, data->data() + begin + stride * i, element_bytes);
        return true;
    }
    const JsonValue* gltf_root_ = nullptr;
    bool loadGltf(const JsonValue& root, const std::string& dir, const Transformation& parent) {
        Gltf g;
        g.dir = dir;
        gltf_root_ = &root;
        auto each = [&](const char* key, auto&& f) {
            if (const JsonValue* a = root.get(key))
                if (JsonValue::Array == a->type)
                    for (const JsonValue& n : a->arr) f(n);
        };
        each("accessors", [&](const JsonValue& n) {
            g.accessors.push_back({n.memberU32("bufferView", 0), n.memberU32("byteOffset", 0), n.memberU32("componentType", 0), n.memberU32("count", 0)});
        });
        each("bufferViews", [&](const JsonValue& n) {
            g.views.push_back({n.memberU32("buffer", 0), n.memberU32("byteLength", 0), n.memberU32("byteOffset", 0), n.memberU32("byteStride", 0)});
        });
        each("buffers", [&](const JsonValue& n) {
            const JsonValue* uri = n.get("uri");
            g.uris.push_back(uri && JsonValue::String == uri->type ? uri->str : std::string());
        });
        g.buffers.resize(g.uris.size());
        bool ok = true;
        each("materials", [&](const JsonValue& n) {
            if (ok) g.materials.push_back(gltfMaterial(n, parent.scale[0], ok));
        });
        if (!ok) return false;
        const JsonValue* nodes = root.get("nodes");
        if (!nodes || JsonValue::Array != nodes->type) return fail("the glTF scene has no nodes");
        const JsonValue* meshes = root.get("meshes");
        for (const JsonValue& node : nodes->arr) {
            const JsonValue* mesh_index = node.get("mesh");
            if (!mesh_index || JsonValue::Number != mesh_index->type) continue;  // (a camera node: the take's camera is the one in use)
            if (!meshes || JsonValue::Array != meshes->type || size_t(mesh_index->num) >= meshes->arr.size()) continue;
            const Vec4f t = readVec3(node.get("translation") ? *node.get("translation") : JsonValue(), splat(0.f));
            const Vec4f sc = readVec3(node.get("scale") ? *node.get("scale") : JsonValue(), splat(1.f));
            Vec4f       q = vec(0.f, 0.f, 0.f, 1.f);
            if (const JsonValue* r = node.get("rotation"))
                if (JsonValue::Array == r->type && r->arr.size() >= 4) q = vec(r->arr[0].f32(), r->arr[1].f32(), r->arr[2].f32(), r->arr[3].f32());
            const Transformation local{vec(t[0], t[1], -t[2], 0.f), sc, vec(q[0], q[1], -q[2], q[3])};
            uint32_t             prop = ZYGPU_NULL;
            if (!gltfMesh(g, meshes->arr[size_t(mesh_index->num)], prop)) return false;
            setWorld(prop, transform(parent, local));
        }
        return true;
    }
    // loadMesh (gltf_loader.zig:279-445): the primitives' vertices one after the other, each primitive a part
    bool gltfMesh(Gltf& g, const JsonValue& mesh, uint32_t& prop) {
        const JsonValue* prims = mesh.get("primitives");
        if (!prims || JsonValue::Array != prims->type || prims->arr.empty()) return fail("a glTF mesh has no primitives");
        std::vector<uint32_t> parts, indices, materials;
        std::vector<float>    positions, normals, uvs;
        std::map<uint32_t, uint32_t> material_map;  // mappedMaterialIndex: the prop's materials in the order its parts first name them
        bool all_uvs = true;
        for (const JsonValue& p : prims->arr) {
            const JsonValue* attributes = p.get("attributes");
            const uint32_t   index_id = p.memberU32("indices", ZYGPU_NULL);
            if (ZYGPU_NULL == index_id || !attributes || JsonValue::Object != attributes->type) return fail("a glTF primitive has no indices or attributes");
            const uint32_t position_id = attributes->memberU32("POSITION", ZYGPU_NULL), normal_id = attributes->memberU32("NORMAL", ZYGPU_NULL);
            const uint32_t uv_id = attributes->memberU32("TEXCOORD_0", ZYGPU_NULL);
            if (ZYGPU_NULL == position_id || ZYGPU_NULL == normal_id) return fail("a glTF primitive without POSITION or NORMAL");
            const uint32_t first_vertex = uint32_t(positions.size() / 3), first_index = uint32_t(indices.size());
            if (index_id >= g.accessors.size()) return fail("the glTF scene names an accessor it does not have");
            if (5125 == g.accessors[index_id].component_type) {  // UNSIGNED_INT
                if (!gltfRead<uint32_t>(g, index_id, 1, indices)) return false;
            } else if (5123 == g.accessors[index_id].component_type) {  // UNSIGNED_SHORT
                std::vector<uint16_t> narrow;
                if (!gltfRead<uint16_t>(g, index_id, 1, narrow)) return false;
                indices.insert(indices.end(), narrow.begin(), narrow.end());
            } else {
                return fail("glTF indices that are neither UInt16 nor UInt32");
            }
            for (size_t i = first_index; i < indices.size(); ++i) indices[i] += first_vertex;
            const uint32_t gltf_material = p.memberU32("material", ZYGPU_NULL);
            auto           mapped = material_map.find(gltf_material);
            if (material_map.end() == mapped) {
                materials.push_back(gltf_material < g.materials.size() && ZYGPU_NULL != g.materials[gltf_material] ? g.materials[gltf_material]
                                                                                                                      : engine_.fallbackMaterial());
                mapped = material_map.emplace(gltf_material, uint32_t(materials.size() - 1)).first;
            }
            parts.insert(parts.end(), {first_index, uint32_t(indices.size()) - first_index, mapped->second});
            if (!gltfRead<float>(g, position_id, 3, positions) || !gltfRead<float>(g, normal_id, 3, normals)) return false;
            if (ZYGPU_NULL != uv_id) {
                if (!gltfRead<float>(g, uv_id, 2, uvs)) return false;
            } else {
                all_uvs = false;
            }
        }
        for (size_t i = 0; i + 2 < indices.size(); i += 3) std::swap(indices[i + 1], indices[i + 2]);
        for (size_t i = 2; i < positions.size(); i += 3) positions[i] = -positions[i];
        for (size_t i = 2; i < normals.size(); i += 3) normals[i] = -normals[i];
        if (!all_uvs) uvs.clear();
        const uint32_t shape = createMesh(parts, indices, positions, normals, uvs);
        if (ZYGPU_NULL == shape) return fail("a glTF mesh could not be created");
        const int32_t p = engine_.createPropShape(shape, materials.data(), uint32_t(materials.size()), false);
        if (p < 0) return fail("a glTF mesh's prop could not be created");
        prop = uint32_t(p);
        return true;
    }
    // loadMaterial (gltf_loader.zig:543-733): the Substitute description the reference writes as JSON text
    uint32_t gltfMaterial(const JsonValue& value, float global_scale, bool& ok) {
        const JsonValue* pbr = value.get("pbrMetallicRoughness");
        if (!pbr || JsonValue::Object != pbr->type) return ZYGPU_NULL;  // (the prop's fallback material)
        // loadTextureDescriptor (gltf_loader.zig:510-535): textures[index].source -> images[].uri, KHR_texture_transform's scale
        auto texture = [&](const JsonValue& t, JsonValue& out) {
            const uint32_t   index = t.memberU32("index", ZYGPU_NULL);
            const JsonValue *textures = gltf_root_->get("textures"), *images = gltf_root_->get("images");
            if (ZYGPU_NULL == index || !textures || JsonValue::Array != textures->type || index >= textures->arr.size()) return false;
            const uint32_t source = textures->arr[index].memberU32("source", ZYGPU_NULL);
            std::string    uri;
            if (images && JsonValue::Array == images->type && source < images->arr.size())
                if (const JsonValue* u = images->arr[source].get("uri"))
                    if (JsonValue::String == u->type) uri = u->str;
            out = jobj();
            JsonValue name;
            name.type = JsonValue::String;
            name.str = uri;
            out.obj.emplace_back("file", name);
            JsonValue scale;
            scale.type = JsonValue::Array;
            scale.arr = {jnum(1.0), jnum(1.0)};
            if (const JsonValue* ext = t.get("extensions"))
                if (const JsonValue* tt = ext->get("KHR_texture_transform"))
                    if (const JsonValue* sc = tt->get("scale"))
                        if (JsonValue::Array == sc->type && sc->arr.size() >= 2) scale.arr = {jnum(sc->arr[0].num), jnum(sc->arr[1].num)};
            out.obj.emplace_back("scale", scale);
            return true;
        };
        // "surface" (metallicRoughnessTexture: roughness and metallic from one image's Y and Z) is not a key this build's Substitute reads
        if (pbr->get("metallicRoughnessTexture")) return ok = unsupported("a glTF material with a metallicRoughnessTexture"), ZYGPU_NULL;
        if (value.get("emissiveTexture")) return ok = unsupported("a glTF material with an emissiveTexture"), ZYGPU_NULL;
        float ior = 1.5f, attenuation_distance = 0.f, coat_roughness = 0.f;
        Vec4f attenuation_color = splat(0.f);
        bool  volume = false, coat = false;
        if (const JsonValue* ext = value.get("extensions"))
            if (JsonValue::Object == ext->type)
                for (const auto& e : ext->obj) {
                    if (e.first == "KHR_materials_ior") {
                        ior = e.second.memberF32("ior", ior);
                    } else if (e.first == "KHR_materials_transmission") {
                        attenuation_distance = 1.f;
                    } else if (e.first == "KHR_materials_volume") {
                        volume = true;
                        if (const JsonValue* c = e.second.get("attenuationColor")) attenuation_color = readVec3(*c, attenuation_color);
                        attenuation_distance = e.second.memberF32("attenuationDistance", attenuation_distance);
                    } else if (e.first == "KHR_materials_clearcoat") {
                        if (e.second.get("clearcoatTexture")) return ok = unsupported("a glTF material with an image texture (clearcoatTexture)"), ZYGPU_NULL;
                        coat = true;
                        coat_roughness = e.second.memberF32("roughnessFactor", 0.f);
                    }
                }
        JsonValue sub = jobj(), tex;
        if (pbr->get("baseColorTexture") && texture(*pbr->get("baseColorTexture"), tex)) sub.obj.emplace_back("color", tex);
        else sub.obj.emplace_back("color", jarr3(pbr->get("baseColorFactor") ? readVec3(*pbr->get("baseColorFactor"), splat(0.7f)) : splat(0.7f)));
        if (value.get("normalTexture") && texture(*value.get("normalTexture"), tex)) sub.obj.emplace_back("normal", tex);
        sub.obj.emplace_back("roughness", jnum(pbr->memberF32("roughnessFactor", 0.7f)));
        sub.obj.emplace_back("metallic", jnum(pbr->memberF32("metallicFactor", 1.f)));
        const Vec4f emissive = value.get("emissiveFactor") ? readVec3(*value.get("emissiveFactor"), splat(0.f)) : splat(0.f);
        if (emissive[0] > 0.f || emissive[1] > 0.f || emissive[2] > 0.f) {
            JsonValue e = jobj();
            e.obj.emplace_back("spectrum", jarr3(emissive));
            sub.obj.emplace_back("emittance", e);
        }
        sub.obj.emplace_back("ior", jnum(ior));
        if (volume) sub.obj.emplace_back("attenuation_color", jarr3(attenuation_color));
        if (attenuation_distance > 0.f) sub.obj.emplace_back("attenuation_distance", jnum(global_scale * attenuation_distance));
        if (coat) {
            JsonValue c = jobj();
            c.obj.emplace_back("ior", jnum(1.5));
            c.obj.emplace_back("roughness", jnum(coat_roughness));
            c.obj.emplace_back("thickness", jnum(0.001));
            sub.obj.emplace_back("coating", c);
        }
        JsonValue rendering = jobj(), material = jobj();
        rendering.obj.emplace_back("Substitute", sub);
        material.obj.emplace_back("rendering", rendering);
        return engine_.createMaterial(ZYGPU_NULL, material);
    }

    // ---- shape_provider.zig:146-272: "SUB\0" -> binary, else JSON
    uint32_t loadMeshFile(const std::string& path) {
        std::string data;
        if (!readFile(path, data)) return ZYGPU_NULL;
        if (data.size() >= 12 && 0 == std::memcmp(data.data(), "SUB\0", 4)) return loadSub(data);
        JsonValue root;
        if (!JsonParser::parse(data.c_str(), root)) return ZYGPU_NULL;
        const JsonValue* g = root.get("geometry");
        if (!g || JsonValue::Object != g->type) return ZYGPU_NULL;
        std::vector<uint32_t> parts, indices;
        std::vector<float>    positions, normals, uvs;
        readParts(*g, parts);
        auto floats = [](const JsonValue* a, std::vector<float>& out) {
            if (!a || JsonValue::Array != a->type) return;
            const JsonValue* src = a;
            if (!a->arr.empty() && JsonValue::Array == a->arr[0].type) src = &a->arr[0];  // animated: the first frame
            out.reserve(src->arr.size());
            for (const JsonValue& x : src->arr) out.push_back(x.f32());
        };
        if (const JsonValue* vs = g->get("vertices")) {
            floats(vs->get("positions"), positions);
            floats(vs->get("normals"), normals);
            floats(vs->get("texture_coordinates_0"), uvs);
        }
        if (const JsonValue* is = g->get("indices"))
            if (JsonValue::Array == is->type)
                for (const JsonValue& x : is->arr) indices.push_back(uint32_t(x.f32()));
        return createMesh(parts, indices, positions, normals, uvs);
    }
    static void readParts(const JsonValue& geometry, std::vector<uint32_t>& parts) {
        if (const JsonValue* ps = geometry.get("parts"))
            if (JsonValue::Array == ps->type)
                for (const JsonValue& p : ps->arr) {
                    parts.push_back(p.memberU32("start_index", 0));
                    parts.push_back(p.memberU32("num_indices", 0));
                    parts.push_back(p.memberU32("material_index", 0));
                }
    }
    uint32_t createMesh(const std::vector<uint32_t>& parts, const std::vector<uint32_t>& indices, const std::vector<float>& positions,
                        const std::vector<float>& normals, const std::vector<float>& uvs) {
        const uint32_t num_vertices = uint32_t(positions.size() / 3);
        if (0 == num_vertices || indices.size() < 3 || normals.size() < size_t(num_vertices) * 3) return ZYGPU_NULL;
        return engine_.createMesh(ZYGPU_NULL, uint32_t(parts.size() / 3), parts.empty() ? nullptr : parts.data(), uint32_t(indices.size() / 3),
                                  indices.data(), num_vertices, positions.data(), 3, normals.data(), 3,
                                  uvs.size() >= size_t(num_vertices) * 2 ? uvs.data() : nullptr, 2);
    }
    uint32_t loadSub(const std::string& data) {
        uint64_t json_size;
        std::memcpy(&json_size, data.data() + 4, 8);
        if (12 + json_size > data.size()) return ZYGPU_NULL;
        std::string header(data.data() + 12, size_t(json_size));
        while (!header.empty() && 0 == header.back()) header.pop_back();
        JsonValue root;
        if (!JsonParser::parse(header.c_str(), root)) return ZYGPU_NULL;
        const JsonValue* g = root.get("geometry");
        if (!g || JsonValue::Object != g->type) return ZYGPU_NULL;

        std::vector<uint32_t> parts;
        readParts(*g, parts);
        uint64_t vertices_offset = 0, indices_offset = 0, indices_size = 0, index_bytes = 4;
        uint32_t num_vertices = 0, num_indices = 0, position_frames = 1, normal_frames = 1;
        bool     quat = false, has_uvs = false, has_tangents = false, delta = false, interleaved = false;
        if (const JsonValue* vs = g->get("vertices")) {
            if (const JsonValue* b = vs->get("binary")) vertices_offset = uint64_t(b->memberF64("offset", 0.0));
            num_vertices = vs->memberU32("num_vertices", 0);
            if (const JsonValue* layout = vs->get("layout"))
                if (JsonValue::Array == layout->type)
                    for (const JsonValue& l : layout->arr) {
                        const JsonValue*  sn = l.get("semantic_name");
                        const std::string name = (sn && JsonValue::String == sn->type) ? sn->str : "";
                        if ("Position" == name) position_frames = l.memberU32("num_frames", 1);
                        else if ("Normal" == name) normal_frames = l.memberU32("num_frames", 1);
                        else if ("Tangent" == name) has_tangents = true;
                        else if ("Tangent_space" == name) quat = true;
                        else if ("TextureCoordinate" == name || "Texture_coordinate" == name) has_uvs = true;
                        else if ("Bitangent_sign" == name && 0 == l.memberU32("stream", 0)) interleaved = true;
                    }
        }
        if (const JsonValue* is = g->get("indices")) {
            if (const JsonValue* b = is->get("binary")) {
                indices_offset = uint64_t(b->memberF64("offset", 0.0));
                indices_size = uint64_t(b->memberF64("size", 0.0));
            }
            num_indices = is->memberU32("num_indices", 0);
            if (const JsonValue* e = is->get("encoding"))
                if (JsonValue::String == e->type) {
                    index_bytes = ("Int16" == e->str || "UInt16" == e->str) ? 2 : 4;
                    delta = "Int16" == e->str || "Int32" == e->str;
                }
        }
        if (interleaved || 0 == num_vertices) return ZYGPU_NULL;  // legacy interleaved streams
        const size_t binary_start = size_t(json_size) + 12;
        const char*  at = data.data() + binary_start + vertices_offset;
        const char*  end = data.data() + data.size();
        auto         take = [&](size_t bytes, void* dst) {
            if (at + bytes > end) return false;
            if (dst) std::memcpy(dst, at, bytes);
            at += bytes;
            return true;
        };
        std::vector<float> positions(size_t(num_vertices) * 3), normals(size_t(num_vertices) * 3), uvs;
        bool               ok = take(positions.size() * 4, positions.data()) && take(size_t(position_frames - 1) * num_vertices * 12, nullptr);
        if (quat) {  // vertex_buffer.zig:163-174: the normal is the z axis of the tangent-space quaternion
            std::vector<float> ts(size_t(num_vertices) * 4);
            uvs.resize(size_t(num_vertices) * 2);
            ok = ok && take(ts.size() * 4, ts.data()) && take(uvs.size() * 4, uvs.data());
            for (uint32_t i = 0; ok && i < num_vertices; ++i) {
                const float q[4] = {ts[4 * i], ts[4 * i + 1], ts[4 * i + 2], std::fabs(ts[4 * i + 3])};
                const float xz = (q[2] + q[2]) * q[0], yz = (q[1] + q[1]) * q[2];
                const float w0 = (q[0] + q[0]) * q[3], w1 = (q[1] + q[1]) * q[3];
                const float t2 = (q[3] + q[0]) * (q[3] - q[0]), t3 = (q[1] + q[2]) * (q[1] - q[2]);
                normals[3 * i] = xz - w1;  // quaternion.zig:173-205 toNormal
                normals[3 * i + 1] = yz + w0;
                normals[3 * i + 2] = t2 - t3;
            }
        } else {
            ok = ok && take(normals.size() * 4, normals.data()) && take(size_t(normal_frames - 1) * num_vertices * 12, nullptr);
            if (has_uvs && has_tangents) ok = ok && take(size_t(num_vertices) * 12, nullptr);
            if (has_uvs) {
                uvs.resize(size_t(num_vertices) * 2);
                ok = ok && take(uvs.size() * 4, uvs.data());
            }
        }
        if (!ok) return ZYGPU_NULL;
        if (0 == num_indices) num_indices = uint32_t(indices_size / index_bytes);
        const char* ib = data.data() + binary_start + indices_offset;
        if (ib + size_t(num_indices) * index_bytes > end) return ZYGPU_NULL;
        std::vector<uint32_t> indices(num_indices);
        int64_t               previous = 0;
        for (uint32_t i = 0; i < num_indices; ++i) {
            int64_t v;
            if (2 == index_bytes) {
                if (delta) {
                    int16_t x;
                    std::memcpy(&x, ib + size_t(i) * 2, 2);
                    v = x;
                } else {
                    uint16_t x;
                    std::memcpy(&x, ib + size_t(i) * 2, 2);
                    v = x;
                }
            } else if (delta) {
                int32_t x;
                std::memcpy(&x, ib + size_t(i) * 4, 4);
                v = x;
            } else {
                uint32_t x;
                std::memcpy(&x, ib + size_t(i) * 4, 4);
                v = x;
            }
            if (delta) {  // shape_provider.zig:965-994: every index is relative to the one before it
                previous += v;
                v = previous;
            }
            indices[i] = uint32_t(v);
        }
        return createMesh(parts, indices, positions, normals, uvs);
    }
};

}  // namespace zyg
