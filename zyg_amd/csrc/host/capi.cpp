// The su_* C ABI (include/zyg_capi.h) on top of the host Engine and the inner zygpu_* ABI.
// Call protocol and error codes follow reference src/capi/capi.zig:57-738.
#include "../../../include/zyg_capi.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "engine.hpp"
#include "ggx_lut.hpp"
#include "image_writers.hpp"
#include "loaders.hpp"

namespace {

struct Capi {
    zyg::Engine        engine;
    ZygpuContext*      ctx = nullptr;    // the device context; with a group: its rank 0's
    ZygpuGroup*        group = nullptr;  // su_ext_set_devices: one context per device (include/zygpu.h "Device groups")
    std::vector<int>   devices;
    std::vector<uint8_t> world_id;       // su_ext_set_world: the communicator id of a world across processes
    int                first_rank = 0, world = 0;
    uint32_t           shard = 2, group_flags = 0;  // 0 iterations, 1 rows, 2: iterations when a call brings at least one per device
    int                device_id = -1;
    uint32_t           frame = 0;
    uint32_t           iteration = 0;
    std::vector<float> target;  // Float4 image (driver.zig target)
    bool               uploaded = false;
    std::vector<zyg::imgw::Exporter> exporters;  // take.exporters (take.zig:303-331)
};

std::unique_ptr<Capi> g;
su_log_func            g_log = nullptr;
su_progress_start_func g_progress_start = nullptr;
su_progress_tick_func  g_progress_tick = nullptr;

void logf(uint32_t level, const char* fmt, ...) {  // core/log.zig:9-68: <= 256 byte line, valid during the call only
    char    buffer[256];
    va_list args;
    va_start(args, fmt);
    std::vsnprintf(buffer, sizeof(buffer), fmt, args);
    va_end(args);
    if (g_log) g_log(level, buffer);
    else std::fprintf(level >= 2 ? stderr : stdout, "%s\n", buffer);
}

int defaultDevice() {
    if (const char* e = std::getenv("ZYG_DEVICE")) return std::atoi(e);
    if (const char* e = std::getenv("LOCAL_RANK")) return std::atoi(e);
    return 0;
}

bool parse(const char* text, zyg::JsonValue& v) { return text && zyg::JsonParser::parse(text, v); }

void logCompileError(int rc) {
    logf(2, -7 == rc ? "a Substitute material has an attenuation_distance: subsurface scattering is a volume, not supported"
            : -8 == rc ? "an emissive Substitute material with a coating is not supported"
            : -10 == rc ? "su_integrators_create asked for the light tracer (\"particle\"), photon mapping or the AOV integrator: only PT, PTDL and PTMIS are supported"
            : -9 == rc ? "a material in use asks for a texture feature that is not supported (a procedural texture as opacity mask or emission map "
                         "or nested more than four deep, an image file that is not PNG / Radiance RGBE / OpenEXR / IES or could not be read, Triplanar emission maps)"
            : -11 == rc ? g->engine.camera_error.c_str()
            : -4 == rc ? "a material names an image that does not exist or has the wrong number of channels for its use"
                       : "scene uses a feature outside the hot-path scope (scattering volumes, emission maps on meshes ...)");
    if (-9 == rc && !g->engine.file_error.empty()) logf(2, "%s", g->engine.file_error.c_str());
}

// su_start_frame's body: compile, upload, set view (driver.configure + startFrame, capi.zig:586-605)
int32_t prepareFrame(uint32_t frame, bool clear) {
    Capi& c = *g;
    const int rc = c.engine.compile(frame);
    if (0 != rc) {
        logCompileError(rc);
        return -1;
    }
    for (const std::string& w : c.engine.warnings()) logf(1, "%s", w.c_str());
    if (!c.ctx && !c.devices.empty()) {
        if (0 != zygpu_group_create(c.devices.data(), int(c.devices.size()), c.world_id.empty() ? nullptr : c.world_id.data(), c.first_rank, c.world,
                                    c.group_flags, &c.group)) {
            logf(2, "zygpu_group_create failed: %s", zygpu_last_error());
            c.group = nullptr;
            return -1;
        }
        c.ctx = zygpu_group_context(c.group, 0);
    }
    if (!c.ctx) {
        const int dev = c.device_id >= 0 ? c.device_id : defaultDevice();
        if (0 != zygpu_create(dev, &c.ctx)) {
            logf(2, "zygpu_create failed: %s", zygpu_last_error());
            c.ctx = nullptr;
            return -1;
        }
    }
    if (0 != (c.group ? zygpu_group_upload_scene(c.group, &c.engine.sceneDesc()) : zygpu_upload_scene(c.ctx, &c.engine.sceneDesc()))) {
        logf(2, "zygpu_upload_scene failed: %s", zygpu_last_error());
        return -1;
    }
    // Perspective.updateFocus (camera_perspective.zig:322-347): a focus point is met by one closest-hit probe ray per camera update,
    // traced on the device through the scene just uploaded (the reference's walk, precise arithmetic)
    float probe[8];
    if (c.engine.focusProbe(probe)) {
        float t = 0.f;
        const int hit = zygpu_probe_closest(c.ctx, probe, &t);
        if (hit < 0) {
            logf(2, "the focus probe failed: %s", zygpu_last_error());
            return -1;
        }
        c.engine.applyFocusProbe(1 == hit, t);
    }
    if (0 != (c.group ? zygpu_group_set_view(c.group, &c.engine.viewDesc()) : zygpu_set_view(c.ctx, &c.engine.viewDesc()))) {
        logf(2, "zygpu_set_view failed: %s", zygpu_last_error());
        return -1;
    }
    c.target.assign(size_t(c.engine.resolution[0]) * size_t(c.engine.resolution[1]) * 4, 0.f);
    c.frame = frame;
    if (clear) {
        c.iteration = 0;
        if (0 != (c.group ? zygpu_group_clear_film(c.group, 0.f) : zygpu_clear_film(c.ctx, 0.f))) return -1;
    }
    c.uploaded = true;
    return 0;
}

using zyg::imgw::linearToGamma;

}  // namespace

extern "C" {

int32_t su_init(void) {
    if (g) return -1;
    g.reset(new Capi());
    return 0;
}

int32_t su_release(void) {
    if (!g) return -1;
    if (g->group) zygpu_group_destroy(g->group);  // (its contexts with it)
    else if (g->ctx) zygpu_destroy(g->ctx);
    g.reset();
    return 0;
}

int32_t su_mount(const char* folder) {  // capi.zig: resources.fs.pushMount: where file-backed textures are looked for
    if (!g || !folder) return -1;
    g->engine.mounts.push_back(folder);
    return 0;
}

int32_t su_perspective_camera_create(uint32_t width, uint32_t height) {
    if (!g) return -1;
    g->engine.cameraCreate(width, height);
    return int32_t(g->engine.camera_entity);
}

int32_t su_camera_set_fov(float fov) {
    if (!g) return -1;
    g->engine.fov = fov;
    return 0;
}

int32_t su_camera_sensor_dimensions(int32_t* dimensions) {
    if (!g) return -1;
    dimensions[0] = g->engine.resolution[0];
    dimensions[1] = g->engine.resolution[1];
    return 0;
}

int32_t su_exporters_create(const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v) || zyg::JsonValue::Object != v.type) return -1;
    // Take.loadExporters (take.zig:303-360): image sequences; "Video" (a pipe into ffmpeg) is not provided
    g->exporters.clear();
    for (const auto& kv : v.obj) {
        if (kv.first != "Image") continue;
        zyg::imgw::Exporter e;
        const zyg::JsonValue* f = kv.second.get("format");
        const std::string     format = (f && zyg::JsonValue::String == f->type) ? f->str : "PNG";
        if ("EXR" == format) {
            e.format = zyg::imgw::Exporter::EXR;
            e.half = 16 == kv.second.memberU32("bitdepth", 16);
        } else if ("RGBE" == format) {
            e.format = zyg::imgw::Exporter::RGBE;
        } else {
            e.format = zyg::imgw::Exporter::PNG;
            e.error_diffusion = kv.second.memberBool("error_diffusion", false);
        }
        g->exporters.push_back(e);
    }
    return 0;
}

int32_t su_aovs_create(const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v)) return -1;
    // View.loadAOV (take.zig:106-129): {"Albedo": true, "Depth": true, ...}
    static const char* const names[ZYGPU_AOV_NUM_CLASSES] = {"Albedo", "Depth", "MaterialId", "GeometricNormal", "ShadingNormal",
                                                             "Roughness", "Emission", "Direct", "Indirect"};
    if (zyg::JsonValue::Object != v.type) return -1;
    for (const auto& kv : v.obj)
        for (uint32_t k = 0; k < ZYGPU_AOV_NUM_CLASSES; ++k)
            if (kv.first == names[k]) {
                if (kv.second.b) g->engine.aov_slots |= 1u << k;
                else g->engine.aov_slots &= ~(1u << k);
            }
    return 0;
}

int32_t su_sampler_create(uint32_t num_samples) {
    if (g) g->engine.num_samples_per_pixel = num_samples;
    return -1;  // capi.zig:215-221: sets the count and falls through to -1
}

int32_t su_integrators_create(const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v)) return -1;
    g->engine.loadIntegrators(v);
    return 0;
}

int32_t su_image_create(uint32_t id, uint32_t format, uint32_t num_channels, uint32_t width, uint32_t height, uint32_t depth,
                        uint32_t pixel_stride, const uint8_t* data) {
    if (!g) return -1;
    const uint32_t image = g->engine.createImage(id, format, num_channels, width, height, depth, pixel_stride, data);
    return ZYGPU_NULL == image ? -1 : int32_t(image);  // other formats / channel counts: -1, as capi.zig:268-294 for its own gaps
}

int32_t su_image_update(uint32_t id, uint32_t pixel_stride, const uint8_t* data) {
    // the next su_start_frame / su_render_frame compiles and uploads the scene again, images included
    return g ? g->engine.updateImage(id, pixel_stride, data) : -1;
}

int32_t su_material_create(uint32_t id, const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v)) return -1;
    const uint32_t m = g->engine.createMaterial(id, v);
    return ZYGPU_NULL == m ? -1 : int32_t(m);
}

int32_t su_material_update(uint32_t id, const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v)) return -1;
    return g->engine.updateMaterial(id, v);
}

int32_t su_triangle_mesh_create(uint32_t id, uint32_t num_parts, const uint32_t* parts, uint32_t num_triangles,
                                const uint32_t* indices, uint32_t num_vertices, const float* positions,
                                uint32_t positions_stride, const float* normals, uint32_t normals_stride,
                                const float* tangents, uint32_t tangents_stride, const float* uvs, uint32_t uvs_stride,
                                bool asyncr) {
    (void)tangents;
    (void)tangents_stride;
    (void)asyncr;  // the build is synchronous, so the caller's arrays are free to go on return (a superset of capi.zig:413-417)
    if (!g || !positions || !normals) return -1;
    return int32_t(g->engine.createMesh(id, num_parts, parts, num_triangles, indices, num_vertices, positions,
                                        positions_stride, normals, normals_stride, uvs, uvs ? uvs_stride : 0));
}

int32_t su_ext_prop_create(uint32_t shape, uint32_t num_materials, const uint32_t* materials, uint32_t unoccluding) {
    if (!g) return -1;
    zyg::Engine& e = g->engine;
    if (shape >= e.numShapes()) return -1;
    const uint32_t        num_expected = e.shapeNumMaterials(shape);
    std::vector<uint32_t> mats;
    for (uint32_t i = 0; i < num_materials; ++i) mats.push_back(materials[i] >= e.numMaterials() ? e.fallbackMaterial() : materials[i]);
    while (mats.size() < num_expected) mats.push_back(e.fallbackMaterial());
    return e.createPropShape(shape, mats.data(), uint32_t(mats.size()), 0 != unoccluding);
}

int32_t su_prop_create(uint32_t shape, uint32_t num_materials, const uint32_t* materials) {
    return su_ext_prop_create(shape, num_materials, materials, 0);
}

int32_t su_prop_create_instance(uint32_t entity) { return g ? g->engine.createPropInstance(entity) : -1; }

int32_t su_light_create(uint32_t prop) { return g ? g->engine.createLight(prop) : -1; }

int32_t su_prop_set_transformation(uint32_t prop, const float* trafo) {
    return g ? g->engine.setTransformation(prop, trafo) : -1;
}

int32_t su_prop_set_transformation_frame(uint32_t prop, uint32_t frame, const float* trafo) {
    return g ? g->engine.setTransformationFrame(prop, frame, trafo) : -1;
}

int32_t su_prop_set_visibility(uint32_t prop, uint32_t in_camera, uint32_t in_reflection, uint32_t in_sss) {
    return g ? g->engine.setVisibility(prop, in_camera > 0, in_reflection > 0, in_sss > 0) : -1;
}

int32_t su_start_frame(uint32_t frame) {
    if (!g) return -1;
    return prepareFrame(frame, true);
}

int32_t su_ext_render_rows(uint32_t num_steps, int32_t y_begin, int32_t y_end) {
    if (!g || !g->uploaded) return -1;
    if (g_progress_start) g_progress_start(num_steps);
    // a group's devices share the call's iterations, or its rows when there are fewer iterations than devices (a client that asks for
    // one iteration at a time, like the reference's Blender add-on, still uses every device)
    const uint32_t shard = g->group ? (2 == g->shard ? (num_steps >= uint32_t(zygpu_group_world(g->group)) ? 0u : 1u) : g->shard) : 0u;
    if (0 != (g->group ? zygpu_group_render(g->group, g->frame, g->iteration, num_steps, y_begin, y_end, shard)
                       : zygpu_render(g->ctx, g->frame, g->iteration, num_steps, y_begin, y_end))) {
        logf(2, "zygpu_render failed: %s", zygpu_last_error());
        return -1;
    }
    if (g_progress_tick)
        for (uint32_t i = 0; i < num_steps; ++i) g_progress_tick();
    g->iteration += num_steps;
    return 0;
}

int32_t su_render_iterations(uint32_t num_steps) {
    if (!g) return -1;
    return su_ext_render_rows(num_steps, 0, g->engine.resolution[1]);
}

int32_t su_render_frame(uint32_t frame) {
    if (!g) return -1;
    if (0 != prepareFrame(frame, true)) return -1;
    // Driver.render (driver.zig:115-152): one Worker.render call per tile with num_samples = the view's samples per pixel
    const uint32_t spp = g->engine.num_samples_per_pixel;
    if (g_progress_start) g_progress_start(spp);
    if (0 != (g->group ? zygpu_group_render_frame(g->group, g->frame, 0, spp, 0, g->engine.resolution[1])
                       : zygpu_render_frame(g->ctx, g->frame, 0, spp, 0, g->engine.resolution[1]))) {
        logf(2, "zygpu_render_frame failed: %s", zygpu_last_error());
        return -1;
    }
    if (g_progress_tick)
        for (uint32_t i = 0; i < spp; ++i) g_progress_tick();
    g->iteration = spp;
    if (0 != (g->group ? zygpu_group_synchronize(g->group) : zygpu_synchronize(g->ctx))) {
        logf(2, "su_render_frame: %s", zygpu_last_error());
        return -1;
    }
    return 0;
}

int32_t su_resolve_frame(uint32_t aov);

// capi.zig:556-566 -> Driver.exportFrame (driver.zig:232-262): resolve, then every exporter writes image_CC_FFFFFF.ext
// into the working directory (or $ZYG_EXPORT_DIR)
int32_t su_export_frame(void) {
    if (!g || !g->uploaded) return -1;
    if (0 != su_resolve_frame(0xFFFFFFFFu)) return -1;
    const char*          dir = std::getenv("ZYG_EXPORT_DIR");
    zyg::imgw::ImageView im;
    im.rgba = g->target.data();
    im.width = g->engine.resolution[0];
    im.height = g->engine.resolution[1];
    for (int i = 0; i < 4; ++i) im.crop[i] = g->engine.crop[i];
    const bool alpha = g->engine.sensor.alpha_transparency;
    for (const zyg::imgw::Exporter& e : g->exporters) {
        std::string path;
        if (!zyg::imgw::exportImage(e, im, alpha, 0, g->frame, dir ? dir : "", &path)) {
            logf(2, "su_export_frame: cannot write %s", path.c_str());
            return -1;
        }
        logf(0, "su_export_frame: wrote %s", path.c_str());
    }
    return 0;
}

int32_t su_resolve_frame_to_buffer(uint32_t aov, uint32_t width, uint32_t height, float* buffer) {
    if (!g || !g->uploaded) return -1;
    const uint32_t film_pixels = uint32_t(g->engine.resolution[0]) * uint32_t(g->engine.resolution[1]);
    const uint32_t num_pixels = std::min(width * height, film_pixels);
    // the group's one exchange step: the devices' films summed onto the first one's (RCCL), which is then resolved like any film
    if (g->group && 1 == zygpu_group_dirty(g->group) && 0 != zygpu_group_reduce(g->group)) {
        logf(2, "su_resolve_frame: %s", zygpu_last_error());
        return -1;
    }
    if (aov < ZYGPU_AOV_NUM_CLASSES) {  // capi.zig:637: -2 for a class the take does not record
        const int r = zygpu_resolve_aov(g->ctx, aov, buffer, num_pixels);
        if (0 != r && -2 != r) logf(2, "su_resolve_frame: %s", zygpu_last_error());
        return 0 == r ? 0 : -2 == r ? -2 : -1;
    }
    const int r = zygpu_resolve(g->ctx, buffer, num_pixels);
    if (0 != r) logf(2, "su_resolve_frame: %s", zygpu_last_error());  // e.g. the device error flags: the image is incomplete
    return 0 == r ? 0 : -1;
}

int32_t su_resolve_frame(uint32_t aov) {
    if (!g || !g->uploaded) return -1;
    return su_resolve_frame_to_buffer(aov, uint32_t(g->engine.resolution[0]), uint32_t(g->engine.resolution[1]),
                                      g->target.data());
}

int32_t su_copy_framebuffer(uint32_t format, uint32_t num_channels, uint32_t width, uint32_t height, uint8_t* destination) {
    if (!g) return -1;
    const uint32_t dw = uint32_t(g->engine.resolution[0]), dh = uint32_t(g->engine.resolution[1]);
    if (g->target.size() < size_t(dw) * dh * 4) return -1;
    const uint32_t used_height = std::min(height, dh), used_width = std::min(width, dw);
    (void)format;
    if (3 == num_channels) {
        for (uint32_t y = 0; y < used_height; ++y) {
            uint32_t o = y * width * 3;
            for (uint32_t x = 0; x < used_width; ++x) {
                const float* c = &g->target[(size_t(y) * dw + x) * 4];
                for (int k = 0; k < 3; ++k) destination[o + k] = uint8_t(linearToGamma(c[k]) * 255.f + 0.5f);
                o += 3;
            }
        }
    } else if (4 == num_channels) {
        float* dst = reinterpret_cast<float*>(destination);
        for (uint32_t y = 0; y < used_height; ++y)
            for (uint32_t x = 0; x < used_width; ++x)
                std::memcpy(dst + (size_t(y) * width + x) * 4, &g->target[(size_t(y) * dw + x) * 4], 16);
    }
    return 0;
}

int32_t su_register_log(su_log_func post) {
    g_log = post;
    return 0;
}

int32_t su_register_progress(su_progress_start_func start, su_progress_tick_func tick) {
    if (!g) return -1;
    g_progress_start = start;
    g_progress_tick = tick;
    return 0;
}

int32_t su_ext_sensor_create(const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v)) return -1;
    g->engine.loadSensor(v);
    return 0;
}

int32_t su_ext_camera_set_parameters(const char* json) {
    if (!g) return -1;
    zyg::JsonValue v;
    if (!parse(json, v)) return -1;
    if (!g->engine.loadCameraParameters(v)) {
        logf(2, "%s", g->engine.camera_error.c_str());
        return -1;
    }
    return 0;
}

int32_t su_ext_sampler_kind(uint32_t kind) {
    if (!g || kind > ZYGPU_SAMPLER_SOBOL) return -1;
    g->engine.sampler_kind = kind;
    return 0;
}

int32_t su_ext_set_devices(const int32_t* device_ids, uint32_t num_devices, uint32_t shard, uint32_t flags) {
    if (!g || g->ctx || !device_ids || 0 == num_devices || shard > 2u) return -1;
    g->devices.assign(device_ids, device_ids + num_devices);
    g->shard = shard;
    g->group_flags = flags;
    return 0;
}

int32_t su_ext_set_world(const uint8_t* unique_id, int32_t first_rank, int32_t world) {
    if (!g || g->ctx || !unique_id || first_rank < 0 || world < 1) return -1;
    g->world_id.assign(unique_id, unique_id + ZYGPU_UNIQUE_ID_BYTES);
    g->first_rank = first_rank;
    g->world = world;
    return 0;
}

ZygpuGroup* su_ext_group(void) { return g ? g->group : nullptr; }

int32_t su_ext_set_device(int32_t device_id) {
    if (!g || g->ctx) return -1;
    g->device_id = device_id;
    return 0;
}

int32_t su_ext_compile(uint32_t frame) {
    if (!g) return -1;
    const int rc = g->engine.compile(frame);
    if (0 != rc) {
        logCompileError(rc);
        return -1;
    }
    for (const std::string& w : g->engine.warnings()) logf(1, "%s", w.c_str());
    return 0;
}

const ZygpuSceneDesc* su_ext_scene_desc(void) { return g ? &g->engine.sceneDesc() : nullptr; }
const ZygpuViewDesc*  su_ext_view_desc(void) { return g ? &g->engine.viewDesc() : nullptr; }
ZygpuContext*         su_ext_context(void) { return g ? g->ctx : nullptr; }

int32_t su_ext_load_take(const char* take_filename) {
    if (!g || !take_filename) return -1;
    zyg::TakeLoader loader(g->engine);
    if (!loader.loadTake(take_filename)) {
        logf(2, "su_ext_load_take: %s", loader.error.c_str());
        return -1;
    }
    g->exporters = loader.exporters;
    return 0;
}

int32_t su_ext_portal_create(uint32_t shape, uint32_t light_id) {
    if (!g) return -1;
    return g->engine.createPortal(shape, light_id);
}

int32_t su_ext_light_create_linked(uint32_t prop, uint32_t light_id) {
    if (!g) return -1;
    return 0 == g->engine.createLight(prop, light_id) ? 0 : -1;
}

int32_t su_ext_prop_set_shadow_catcher(uint32_t prop) {
    if (!g) return -1;
    return 0 == g->engine.setShadowCatcher(prop) ? 0 : -1;
}

int32_t su_ext_prop_set_visibility(uint32_t prop, bool in_camera, bool in_reflection, bool in_sss, bool shadow_catcher_light) {
    if (!g) return -1;
    return 0 == g->engine.setVisibility(prop, in_camera, in_reflection, in_sss, shadow_catcher_light) ? 0 : -1;
}

int32_t su_ext_sky_create(const char* parameters_json) {
    if (!g) return -1;
    const int32_t sun = g->engine.createSky();
    if (sun < 0) return -1;
    if (parameters_json) {
        zyg::JsonValue v;
        if (!zyg::JsonParser::parse(parameters_json, v)) return -1;
        g->engine.setSkyParameters(v);
    }
    return sun;
}

int32_t su_ext_sky_tables(const float* sun_rgb, const float* sky_rgb, uint32_t sky_width, uint32_t sky_height) {
    if (!g) return -1;
    return 0 == g->engine.setSkyTables(sun_rgb, sky_rgb, sky_width, sky_height) ? 0 : -1;
}

int32_t su_ext_sky_model_eval(const char* dataset, double elevation, double visibility, double albedo, uint32_t what, const double* queries,
                              uint32_t count, double* out) {
    if (!dataset || !queries || !out || what > 2u) return -1;
    auto        data = std::make_shared<zyg::prague::Dataset>();
    std::string why;
    if (!data->load(dataset, why)) return -1;
    const zyg::prague::Model model(data, elevation, visibility, albedo);
    for (uint32_t i = 0; i < count; ++i) {
        const double* q = queries + 4 * size_t(i);
        out[i] = 0 == what ? model.skyRadiance(q[0], q[1], q[2], q[3]) : 1 == what ? model.solarRadiance(q[0], q[1]) : model.transmittance(q[0], q[1], q[2]);
    }
    return 0;
}

uint64_t su_ext_encode_image(uint32_t format, uint32_t flags, uint32_t width, uint32_t height, const int32_t* crop, const float* rgba,
                             uint8_t* out, uint64_t capacity) {
    if (!rgba || !crop) return 0;
    zyg::imgw::ImageView im{rgba, int(width), int(height), {crop[0], crop[1], crop[2], crop[3]}};
    const bool           alpha = 0 != (flags & 1u);
    std::vector<uint8_t> bytes;
    if (0 == format) {
        const std::vector<uint8_t> px = zyg::imgw::toSrgb(im, alpha, 0 != (flags & 2u));
        bytes = zyg::imgw::encodePng(px.data(), im.width, im.height, alpha ? 4 : 3);
    } else if (1 == format) {
        bytes = zyg::imgw::encodeExr(im, alpha, 0 == (flags & 4u));
    } else if (2 == format) {
        bytes = zyg::imgw::encodeRgbe(im);
    }
    if (out) std::memcpy(out, bytes.data(), size_t(std::min<uint64_t>(capacity, bytes.size())));
    return bytes.size();
}

uint32_t su_ext_ggx_lut(uint32_t which, float* out, uint32_t capacity) {
    const zyg::ggxlut::Tables& t = zyg::ggxlut::tables();
    const std::vector<float>*  v = 0 == which ? &t.e_m : 1 == which ? &t.e_m_avg : 2 == which ? &t.e : 3 == which ? &t.e_avg : 4 == which ? &t.e_s : nullptr;
    if (!v) return 0;
    if (out) std::copy(v->begin(), v->begin() + std::min<size_t>(capacity, v->size()), out);
    return uint32_t(v->size());
}

}  // extern "C"
