// Host side of the drop-in: resources, scene graph, scene compile, camera, view.
// Mirrors the parts of the reference that sit above the integration loop and feed it:
//   resources / built-in shapes   src/core/resource/manager.zig:36-82
//   Scene.createProp*/createLight  src/core/scene/scene.zig:254-342
//   Prop.configureShape            src/core/scene/prop/prop.zig:96-135
//   Scene.compile                  src/core/scene/scene.zig:185-223, 402-497
//   Space world bounds / trafos    src/core/scene/space.zig:60-111
//   Perspective.update             src/core/camera/camera_perspective.zig:79-122
//   Sensor.init (filter table)     src/core/rendering/sensor/sensor.zig:27-58,105-120,630-645
// Output: ZygpuSceneDesc + ZygpuViewDesc (include/zygpu.h) for zygpu_upload_scene / zygpu_set_view.
#pragma once

#include <deque>
#include <memory>
#include <string>
#include <map>
#include <vector>

#include "../../../include/zygpu.h"
#include "bvh_builder.hpp"
#include "hmath.hpp"
#include "json.hpp"
#include "light_tree_builder.hpp"
#include "prague_sky.hpp"

namespace zyg {

struct HostMesh {
    std::vector<ZygpuBvhNode> nodes;
    std::vector<uint32_t>     triangles;  // 3 per triangle, leaf order
    std::vector<uint16_t>     triangle_parts;
    std::vector<float>        positions;  // 3 * nv + 1
    std::vector<uint16_t>     normals;    // 2 * nv
    std::vector<float>        uvs;        // 2 * nv
    std::vector<uint32_t>     part_materials;
    std::vector<float>        part_areas;
    std::vector<uint32_t>     primitive_mapping;  // triangle -> index within its part; filled when a part becomes a light
    uint32_t                  num_vertices = 0;

    uint32_t numTriangles() const { return uint32_t(triangles.size() / 3); }
    uint32_t numParts() const { return uint32_t(part_materials.size()); }
    uint32_t numMaterials() const {  // triangle_mesh.zig:262-270
        uint32_t id = 0;
        for (uint32_t m : part_materials) id = m > id ? m : id;
        return id + 1;
    }
    AABB aabb() const {
        const ZygpuBvhNode& n = nodes[0];
        return {{vec(n.min[0], n.min[1], n.min[2], 0.f), vec(n.max[0], n.max[1], n.max[2], 0.f)}};
    }
    Vec4f position(uint32_t i) const { return vec(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2], 0.f); }
};

// One emissive mesh part prepared for sampling: triangle_mesh.zig:57-149 (Part.configure) + shape_sampler.zig:160-258
struct HostMeshLight {
    uint32_t              mesh = 0, part = 0, material = 0;
    bool                  two_sided = false;
    std::vector<uint32_t> triangle_mapping;
    std::vector<float>    tri_props;  // 8 per triangle
    LightTreeOutput       tree;
    AABB                  aabb;       // of the emitting triangles, object space
    Vec4f                 cone;       // power-weighted dominant axis + cos of the widest deviation
};

// An emission map prepared for importance sampling: light_material.zig:52-125 + Distribution2D (flattened, see zygpu.h)
constexpr uint32_t kSkySunSamples = 1024;      // Sky.BakeDimensionsSun (sky.zig:49)
constexpr int32_t  kSkyBakeDimension = 1024;   // Sky.BakeDimensions (sky.zig:48)

struct HostImageLight {
    std::vector<float>    cdf;
    std::vector<uint32_t> lut, rows;
    uint32_t              num_rows = 0, sampler = 0;
    float                 total_weight = 0.f;
    Vec4f                 average_emission;  // emittance.value * the uv-weighted mean of the map
};

struct HostImage {  // image/typed_image.zig: 2-D only
    uint32_t             format = ZYGPU_IMAGE_FLOAT3;
    int32_t              width = 0, height = 0;
    std::vector<uint8_t> pixels;
};

struct HostShape {
    uint32_t kind = ZYGPU_SHAPE_RECTANGLE;
    uint32_t mesh = ZYGPU_NULL;
};

struct HostProp {
    uint32_t shape = ZYGPU_NULL;  // resource id
    uint32_t flags = ZYGPU_PROP_VISIBLE_IN_CAMERA | ZYGPU_PROP_VISIBLE_IN_REFLECTION | ZYGPU_PROP_VISIBLE_IN_SHADOW |
                     ZYGPU_PROP_SOLID | ZYGPU_PROP_STATIC;
    uint32_t parts_start = 0;
};

struct SensorSettings {  // sensor.zig:60-120
    int32_t filter_radius_int = 2;
    float   filter_radius = 2.f;
    float   filter_samples[ZYGPU_FILTER_TAPS + 1] = {};
    float   clamp_emission = FLT_MAX, clamp_direct = FLT_MAX, clamp_indirect = FLT_MAX;
    float   exposure = 0.f;
    uint32_t tonemapper = ZYGPU_TONEMAP_LINEAR;
    bool    alpha_transparency = false;  // buffer class Transparent instead of Opaque
};

class Engine {
  public:
    Engine();

    // --- resources ---
    // capi.zig:236-298: UInt8 x 3 (read as sRGB) and Float32 x 3, depth 1; returns the image id or ZYGPU_NULL
    // File-backed textures (material_provider.zig:438-740 "file", texture_provider.zig:27-65, image_provider.zig:27-69): a PNG or
    // Radiance picture found through the mounts (su_mount; the loaders push the directory of the file they are reading), decoded
    // on the host and kept like an image handed over with su_image_create.  usage: 0 Color, 1 Emission, 2 Normal, 3 Opacity,
    // 4 Weight (its default swizzle); swizzle: "" or X Y Z W XY YX YZ XYZ.  ZYGPU_NULL and a message in file_error on failure.
    std::vector<std::string> mounts;
    std::string              file_error;
    std::string resolveFile(const std::string& name) const;
    uint32_t    loadImageFile(const std::string& name, uint32_t usage, const std::string& swizzle, bool invert);
    uint32_t    appendProcedural(const ZygpuProcedural& p);  // resources.procedural.append (procedural.zig:54-80)
    uint32_t    loadImageFileUnguarded(const std::string& name, uint32_t usage, const std::string& swizzle, bool invert);
    uint32_t createImage(uint32_t id, uint32_t format, uint32_t num_channels, uint32_t width, uint32_t height, uint32_t depth,
                         uint32_t pixel_stride, const uint8_t* data);
    uint32_t createMaterial(uint32_t id, const JsonValue& v);  // returns id or ZYGPU_NULL
    int      updateMaterial(uint32_t id, const JsonValue& v);
    uint32_t createMesh(uint32_t id, uint32_t num_parts, const uint32_t* parts, uint32_t num_triangles,
                        const uint32_t* indices, uint32_t num_vertices, const float* positions, uint32_t positions_stride,
                        const float* normals, uint32_t normals_stride, const float* uvs, uint32_t uvs_stride);

    // --- scene ---
    uint32_t createEntity();
    int32_t  createPropShape(uint32_t shape, const uint32_t* materials, uint32_t num_materials, bool unoccluding);
    int32_t  createPropInstance(uint32_t entity);
    // light_link (scene.zig:344-372, 707): the light a Portal is a window onto, or ZYGPU_NULL
    int      createLight(uint32_t entity, uint32_t light_link = ZYGPU_NULL);
    // A Portal entity (util/scene_loader.zig:241-248, 371-397 loadPortal): a prop of `shape` (a Rectangle) that takes the material
    // of light `light_id` (a Dome with an emission map), which it hides and turns into a prototype (never sampled itself); the
    // caller places the prop and then calls createLight(prop, light_id).  Returns the prop or -1.
    int32_t  createPortal(uint32_t shape, uint32_t light_id);
    uint32_t numLights() const { return uint32_t(lights_.size()); }
    int      setTransformation(uint32_t prop, const float* m16);
    int      setWorldTransformation(uint32_t prop, const Transformation& t);  // the loaders' path (scene_graph.zig:178-230)
    // su_prop_set_transformation_frame (capi.zig:506-533): keyframe `frame` (< numInterpolationFrames()) of the prop.  The
    // camera entity's keyframes drive the camera at every sample's time (ZygpuViewDesc.camera_keyframes); any other prop is
    // carried as keyframes in the compiled scene and interpolated at every ray's time (ZygpuSceneDesc.keyframes); an animated
    // *light* is refused by compile().
    int      setTransformationFrame(uint32_t prop, uint32_t frame, const float* m16);
    // Scene files' "animation" (util/animation.zig, animation_loader.zig): keyframes of an entity's transformation with their times
    // (in the scene's time units, 705 600 000 per second); compile(frame) resamples them at the frame's ticks (Graph.simulate,
    // scene_graph.zig:106-118) into the entity's interpolation frames, which is what setTransformationFrame sets by hand
    int      createAnimation(const std::vector<uint64_t>& times, const std::vector<Transformation>& frames);
    // A node of the animated part of the scene graph (scene_graph.zig:178-226, 250-310): `prop` or ZYGPU_NULL (a node that only
    // passes its frames on), its parent node (-1: none; the static ancestors are in the animation's keyframes already), its own
    // animation (-1: the static `local`).  Frame i of a node = its parent's frame i transforming its own.  Parents come first.
    int      createGraphNode(uint32_t prop, int parent, int animation, const Transformation& local);
    void     simulate(uint32_t frame);
    size_t   numGraphNodes() const { return graph_nodes_.size(); }
    uint32_t numInterpolationFrames() const;  // scene.zig:254-257, 724-734 with the C ABI camera's frame step / duration
    // su_image_update (capi.zig:300-338): new pixels for an existing image, same format and dimensions
    int      updateImage(uint32_t id, uint32_t pixel_stride, const uint8_t* data);
    const std::vector<std::string>& warnings() const { return warnings_; }  // of the last compile()
    int      setVisibility(uint32_t prop, bool in_camera, bool in_reflection, bool in_sss, bool shadow_catcher_light = false);
    int      setShadowCatcher(uint32_t prop);  // scene.propSetShadowCatcher (scene.zig:398-400)
    // The Sky entity (sky/sky.zig; util/scene_loader.zig:704-712): a Distant sun and a Canopy sky, both lights.  createSky
    // returns the sun's prop (Sky.configure, idempotent); the parameters are the entity's "parameters" node (Sky.setParameters).
    // The bake's products (sky.zig:172-269: the sun's radiance over its disc, kSkySunSamples rgb samples, and the canopy image)
    // come from the client with setSkyTables (what the reference reads back from its cache files, sky.zig:138-168); without
    // them compile() fills in a procedural stand-in (the Prague model's dataset does not ship with the reference).
    int32_t  createSky();
    void     setSkyParameters(const JsonValue& v);
    int      setSkyTables(const float* sun_rgb, const float* sky_rgb, uint32_t sky_width, uint32_t sky_height);
    uint32_t skySun() const { return sky_.sun; }
    uint32_t skyCanopy() const { return sky_.sky; }

    // --- view ---
    void cameraCreate(uint32_t width, uint32_t height);
    void loadIntegrators(const JsonValue& v);
    void loadSensor(const JsonValue& v);
    bool loadCameraParameters(const JsonValue& v);  // false: camera_error names what was refused
    // Perspective.updateFocus (camera_perspective.zig:322-347): whether this camera wants its focus distance from a probe ray, the ray
    // (8 floats: origin, min_t, direction, max_t; camera-relative world space as the compiled scene), and the result's way back
    bool focusProbe(float ray[8]) const;
    void applyFocusProbe(bool hit, float t);

    // Scene.compile + camera.update: fills scene_desc / view_desc (pointers stay valid until the next compile)
    int compile(uint32_t frame);

    const ZygpuSceneDesc& sceneDesc() const { return scene_desc_; }
    const ZygpuViewDesc&  viewDesc() const { return view_desc_; }

    size_t numShapes() const { return shapes_.size(); }
    size_t numProps() const { return props_.size(); }
    size_t numMaterials() const { return materials_.size(); }
    uint32_t shapeNumMaterials(uint32_t shape) const {
        const HostShape& s = shapes_[shape];
        return ZYGPU_SHAPE_TRIANGLE_MESH == s.kind ? meshes_[s.mesh].numMaterials() : 1;
    }
    uint32_t fallbackMaterial() const { return fallback_material_; }

    int32_t  resolution[2] = {0, 0};
    int32_t  crop[4] = {0, 0, 0, 0};
    float    fov = 0.f;
    float    lens_radius = 0.f;
    float    focus_distance = 0.f;
    // camera_perspective.zig:16-20, 306-347: "focus": {"point": [x, y, z]}: x, y in [0, 1] of the frame (scaled by the resolution at
    // compile, setFocus), z added to the distance the probe ray finds there (updateFocus, one closest-hit ray per camera update)
    float    focus_point[3] = {0.5f, 0.5f, 0.f};
    bool     focus_use_point = false;
    // camera/aperture.zig:22-53: "aperture": {"shape": file}: a one-channel image whose values weight the lens (Distribution2D)
    uint32_t aperture_shape_image = ZYGPU_NULL;
    std::string camera_error;  // a camera parameter this build refuses by name ("blades", "stereo"): compile() fails with it
    // The camera's shutter (take_loader.zig:153-191 loadShutter -> camera_base.zig:43-75 setShutter): the frame's step and duration
    // in ticks of 1 / 705 600 000 s (60 frames per second, motion blur on: camera_base.zig:16, 25-26) and, once configured, the cdf of
    // the Distribution1DN(63) over the shutter function that a camera sample's time is drawn through (camera_base.zig:121-127)
    uint64_t frame_step = 705600000ull / 60, frame_duration = 705600000ull / 60;
    bool     shutter_configured = false;
    float    shutter_cdf[64] = {};
    void     setShutter(float open, float close, const float* slope, uint32_t num_slope);
    uint32_t camera_entity = ZYGPU_NULL;
    uint32_t num_samples_per_pixel = 1;  // capi.zig:99
    uint32_t aov_slots = 0;              // take.view.aovs (take.zig:106-129): bit = ZygpuAovClass
    uint32_t sampler_kind = ZYGPU_SAMPLER_SOBOL;
    float    specular_threshold;

    // integrator (take/take.zig:131-271); default mirrors View's AOV placeholder by PTMIS defaults once configured
    uint32_t integrator = ZYGPU_INTEGRATOR_PTMIS;
    uint32_t max_depth_surface = 16, max_depth_volume = 256;
    float    split_threshold = 0.5f * 0.5f * 0.5f * 0.5f;
    bool     caustics_path = true;
    float    regularize_roughness = 0.f;
    bool     unsupported_integrator_ = false;  // su_integrators_create asked for the light tracer, photons or the AOV integrator

    SensorSettings sensor;

  private:
    void     parseMaterial(ZygpuMaterial& m, const JsonValue& rendering_value, uint32_t kind, bool init = true);
    void     parseMaterialBody(ZygpuMaterial& m, const JsonValue& rendering_value, uint32_t kind, bool init);
    void     commitMaterial(ZygpuMaterial& m);
    void     classifyProp(uint32_t p);
    bool     shapeFinite(uint32_t shape) const;
    AABB     shapeAabb(uint32_t shape) const;
    uint32_t shapeNumParts(uint32_t shape) const;
    void     buildPropTree(const std::vector<uint32_t>& indices, std::vector<ZygpuBvhNode>& nodes,
                           std::vector<uint32_t>& out_indices);
    void     setFilter(float radius, int kind);

    std::vector<HostShape>     shapes_;
    std::vector<HostMesh>      meshes_;
    std::vector<ZygpuMaterial> materials_;
    uint32_t                   fallback_material_ = 0;
    std::vector<ZygpuProcedural> procedurals_;  // resources.procedural's lists in one (texture/procedural.zig:37-43), append-only

    std::vector<HostProp>       props_;
    std::vector<Transformation> world_transformations_;
    std::vector<std::vector<Transformation>> keyframes_;  // per prop; empty = static (space.zig:23-26 frames / keyframes)
    std::vector<std::string>    warnings_;
    uint32_t                    aperture_light_ = ZYGPU_NULL;  // the aperture's Distribution2D among image_lights_ (compile)
    const Transformation& frameStartTransformation(uint32_t p) const {  // space.zig:99-125 at the frame's start: keyframe 0
        return (p < keyframes_.size() && !keyframes_[p].empty()) ? keyframes_[p][0] : world_transformations_[p];
    }
    std::vector<uint32_t>       material_ids_;
    std::vector<uint32_t>       light_ids_;
    std::vector<ZygpuLight>     lights_;
    std::vector<uint32_t>       finite_props_, infinite_props_, unoccluding_props_, volume_props_;

    // compiled products (backing store of scene_desc_)
    std::vector<ZygpuProp>    c_props_;
    std::vector<ZygpuTrafo>   c_trafos_;
    std::vector<ZygpuAabb>    c_aabbs_;
    std::vector<ZygpuKeyframe> c_keyframes_;    // the animated props' keyframes, flattened (ZygpuSceneDesc.keyframes)
    std::vector<uint32_t>      c_prop_frames_;  // per prop: first keyframe or ZYGPU_NULL
    std::vector<ZygpuMesh>    c_meshes_;
    std::vector<HostImage>    images_;
    std::vector<ZygpuImage>   c_images_;
    std::vector<float>        srgb_to_float_;
    std::vector<uint8_t>        c_light_infinite_;
    std::deque<HostMeshLight>   mesh_lights_;   // stable addresses: the trees point into their own vectors
    std::vector<ZygpuMeshLight> c_mesh_lights_;
    std::deque<HostImageLight>   image_lights_;
    std::vector<ZygpuImageLight> c_image_lights_;
    uint32_t prepareImageLight(uint32_t shape_kind, const ZygpuMaterial& m, bool compensate = true);
    struct HostAnimation {
        std::vector<uint64_t>       times;
        std::vector<Transformation> frames;
    };
    struct GraphNode {
        uint32_t                    prop;
        int                         parent, animation;
        Transformation              local;
        std::vector<Transformation> frames;  // simulate(): the node's world transformation at the frame's ticks
    };
    std::vector<HostAnimation> animations_;
    std::map<std::string, uint32_t> image_files_;  // path | swizzle | invert -> image
    std::vector<GraphNode>     graph_nodes_;
    struct HostPortalLight {
        std::vector<float> sat, func;
        uint32_t           dim = 0, light_link = ZYGPU_NULL;
        Vec4f              average_emission;
    };
    std::deque<HostPortalLight>   portal_lights_;
    std::vector<ZygpuPortalLight> c_portal_lights_;
    std::vector<uint32_t>         light_links_;  // per light (scene.light_links)
    std::vector<uint8_t>          c_light_prototype_;
    uint32_t preparePortalLight(const ZygpuMaterial& m, const Transformation& portal, const Transformation& dome, uint32_t light_link);
    struct HostSky {
        uint32_t sun = ZYGPU_NULL, sky = ZYGPU_NULL, sun_material = ZYGPU_NULL, sky_material = ZYGPU_NULL, image = ZYGPU_NULL;
        float    visibility = 100.f, albedo = 0.25f;                                                   // sky.zig:39-40
        Mat3x3   sun_rotation{{vec(1.f, 0.f, 0.f), vec(0.f, 0.f, 1.f), vec(0.f, -1.f, 0.f)}};          // sky.zig:42
        bool     tables_given = false;
        bool     baked = false;       // sun_rgb and the sky image are the Prague model's (host/sky_bake.hpp) for baked_key
        float    baked_key[3] = {0.f, 0.f, 0.f};  // visibility, albedo, the sun direction's y
        std::string bake_refusal;     // why the bake did not run (no dataset, or not a dataset)
        std::shared_ptr<const prague::Dataset> dataset;  // the parsed model dataset of dataset_path
        std::string dataset_path;
        std::vector<float> sun_rgb;  // kSkySunSamples x 3, as given or as generated
    } sky_;
    std::vector<float> c_sun_radiance_;  // 4 per sample
    void compileSky();
    void     ensureSrgbTable();
    uint32_t prepareMeshLight(uint32_t mesh, uint32_t part, uint32_t material, bool two_sided);
    std::vector<ZygpuAabb>    c_light_aabbs_;
    std::vector<float>        c_light_cones_;
    std::vector<uint8_t>      c_light_two_sided_;
    std::vector<ZygpuBvhNode> solid_nodes_, unoccluding_nodes_;
    std::vector<uint32_t>     solid_indices_, unoccluding_indices_;
    LightTreeOutput           light_tree_;
    std::vector<float>        luts_[5];

    ZygpuSceneDesc scene_desc_{};
    ZygpuViewDesc  view_desc_{};
};

}  // namespace zyg
