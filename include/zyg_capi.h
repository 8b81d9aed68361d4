/*
 * zyg_capi.h — outer C ABI: the `su_*` surface of the reference's libzyg.so
 * (reference src/capi/capi.zig:57-738; callers: src/capi-test/test.py:57-261,
 * src/capi-test/test_progressive.py, src/blender-plugin/engine.py:49-381).
 *
 * Same names, argument meaning and error convention (int32: 0 = ok, >= 0 = created id,
 * -1 = failure / not initialised, -2 = AOV inactive, -3/-4 = material update errors).
 * One process-global engine, single client thread, caller-owned inputs are copied.
 * The render calls run on the MI355X through the inner ABI (zygpu.h); there is no CPU path.
 *
 * `su_ext_*` entries are additions without a reference counterpart; each says why it exists.
 */
#ifndef ZYG_CAPI_H
#define ZYG_CAPI_H

#include <stdbool.h>
#include <stdint.h>

#include "zygpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void (*su_log_func)(uint32_t level, const char* text);  /* core/log.zig:42-55 */
typedef void (*su_progress_start_func)(uint32_t resolution);    /* core/progress.zig:3-77 */
typedef void (*su_progress_tick_func)(void);

int32_t su_init(void);                                                   /* capi.zig:57-114 */
int32_t su_release(void);                                                /* capi.zig:116-129 */
int32_t su_mount(const char* folder);                                    /* capi.zig:131-141 */
int32_t su_perspective_camera_create(uint32_t width, uint32_t height);  /* capi.zig:143-167; returns camera entity */
int32_t su_camera_set_fov(float fov);                                    /* capi.zig:169-176; radians */
int32_t su_camera_sensor_dimensions(int32_t* dimensions);                /* capi.zig:178-187 */
int32_t su_exporters_create(const char* json);                           /* capi.zig:189-200 */
int32_t su_aovs_create(const char* json);                                /* capi.zig:202-213, take.zig:106-129: {"Albedo": true, ...} */
int32_t su_sampler_create(uint32_t num_samples);                         /* capi.zig:215-221; always returns -1, as the reference */
int32_t su_integrators_create(const char* json);                         /* capi.zig:223-234 */
int32_t su_image_create(uint32_t id, uint32_t format, uint32_t num_channels, uint32_t width, uint32_t height,
                        uint32_t depth, uint32_t pixel_stride, const uint8_t* data); /* capi.zig:236-298 */
int32_t su_image_update(uint32_t id, uint32_t pixel_stride, const uint8_t* data);   /* capi.zig:300-340 */
int32_t su_material_create(uint32_t id, const char* json);               /* capi.zig:342-353 */
int32_t su_material_update(uint32_t id, const char* json);               /* capi.zig:355-377 */
int32_t su_triangle_mesh_create(uint32_t id, uint32_t num_parts, const uint32_t* parts, uint32_t num_triangles,
                                const uint32_t* indices, uint32_t num_vertices, const float* positions,
                                uint32_t positions_stride, const float* normals, uint32_t normals_stride,
                                const float* tangents, uint32_t tangents_stride, const float* uvs, uint32_t uvs_stride,
                                bool asyncr); /* capi.zig:379-423; parts = {start_index, num_indices, material} x num_parts */
int32_t su_prop_create(uint32_t shape, uint32_t num_materials, const uint32_t* materials); /* capi.zig:425-455 */
int32_t su_prop_create_instance(uint32_t entity);                        /* capi.zig:457-469 */
int32_t su_light_create(uint32_t prop);                                  /* capi.zig:471-483 */
int32_t su_prop_set_transformation(uint32_t prop, const float* trafo);   /* capi.zig:485-504; 16 floats, row-major, translation in [12..14] */
int32_t su_prop_set_transformation_frame(uint32_t prop, uint32_t frame, const float* trafo); /* capi.zig:506-533 */
int32_t su_prop_set_visibility(uint32_t prop, uint32_t in_camera, uint32_t in_reflection, uint32_t in_sss); /* capi.zig:535-546 */
int32_t su_render_frame(uint32_t frame);                                 /* capi.zig:548-567 */
int32_t su_export_frame(void);                                           /* capi.zig:569-579 */
int32_t su_start_frame(uint32_t frame);                                  /* capi.zig:581-600 */
int32_t su_render_iterations(uint32_t num_steps);                        /* capi.zig:602-611 */
int32_t su_resolve_frame(uint32_t aov);                                  /* capi.zig:613-624 */
int32_t su_resolve_frame_to_buffer(uint32_t aov, uint32_t width, uint32_t height, float* buffer); /* capi.zig:626-641 */
int32_t su_copy_framebuffer(uint32_t format, uint32_t num_channels, uint32_t width, uint32_t height,
                            uint8_t* destination);                       /* capi.zig:643-724 */
int32_t su_register_log(su_log_func post);                              /* capi.zig:726-729 */
int32_t su_register_progress(su_progress_start_func start, su_progress_tick_func tick); /* capi.zig:731-738 */

/* ---- additions (no reference counterpart) ---- */

/* Scene.createPropShape with `unoccluding` exposed: the reference only reaches unoccluding=true through the scene
 * loader's "Light" entity (util/scene_loader.zig:249,365-366); su_prop_create always passes false (capi.zig:449). */
int32_t su_ext_prop_create(uint32_t shape, uint32_t num_materials, const uint32_t* materials, uint32_t unoccluding);
/* The take's "sensor" node (cli/take_loader.zig:193-230): {"clamp": {...}, "filter": {"Mitchell"|"Blackman": {}}};
 * without "filter" the sensor is the box / radius-0 one.  The reference C ABI always uses Mitchell r=2 (take.zig:59-64). */
int32_t su_ext_sensor_create(const char* json);
/* Camera "parameters" node (Perspective.setParameters, camera_perspective.zig:260-304): fov (degrees), lens.radius / aperture.radius,
 * aperture.shape (a one-channel square image file, found through the mounts: the lens is sampled by its Distribution2D,
 * camera/aperture.zig:22-53), focus.distance, focus.point [x, y, z] (x, y in [0, 1] of the frame: su_start_frame traces
 * Perspective.updateFocus' probe ray on the device and focuses on its hit distance + z, camera_perspective.zig:322-347),
 * aperture.blades / roundness (more than 3 blades: the polygon rendering/shaper.zig draws into a 128 x 128 image, which then is the
 * aperture's shape, camera_perspective.zig:284-300).
 * Returns -1 and logs the key for what is refused by name: stereo (two layers), a shape file that cannot be read. */
int32_t su_ext_camera_set_parameters(const char* json);
/* The take's sampler class (take_loader.zig:232-244: {"Random": ...} | {"Sobol": ...}): 0 = Random (PCG only), 1 = Sobol
 * (the C ABI's default; su_sampler_create only sets the sample count, capi.zig:215-221). */
int32_t su_ext_sampler_kind(uint32_t kind);
/* HIP device the engine renders on (default: env ZYG_DEVICE, else LOCAL_RANK, else 0). */
int32_t su_ext_set_device(int32_t device_id);
/* Several HIP devices behind the same su_* calls (before the first su_start_frame / su_render_frame): the engine renders with a
 * ZygpuGroup (include/zygpu.h "Device groups": one context and one host thread per device, the scene replicated, the films summed
 * onto the first device's with one ncclReduce over xGMI before a resolve).  Stands where the reference's Driver hands tiles to its
 * worker threads (driver.zig:115-152, 293-348).  shard: how su_render_iterations splits a call: 0 = its iterations, 1 = the frame's
 * rows, 2 = iterations when the call brings at least one per device, rows otherwise; su_render_frame always splits rows.
 * flags: ZygpuGroupFlags.  A device may be listed twice (testing on one GPU).  su_ext_group: the group (NULL before the first
 * frame, or without this call). */
int32_t su_ext_set_devices(const int32_t* device_ids, uint32_t num_devices, uint32_t shard, uint32_t flags);
/* The devices of su_ext_set_devices as ranks [first_rank, first_rank + num_devices) of a world of `world` ranks across processes (one
 * process per GPU under a launcher): unique_id = the ZYGPU_UNIQUE_ID_BYTES that zygpu_group_unique_id produced in rank 0's process.
 * su_render_iterations(n) then names the WORLD's iterations, of which this process renders its ranks' shares, and the resolve's
 * reduce is collective: every process makes the same calls; rank 0's process holds the frame. */
int32_t su_ext_set_world(const uint8_t* unique_id, int32_t first_rank, int32_t world);
ZygpuGroup* su_ext_group(void);
/* Scene.compile + camera.update only (driver.zig:154-180), no device work; fills the descriptors below. */
int32_t su_ext_compile(uint32_t frame);
const ZygpuSceneDesc* su_ext_scene_desc(void);
const ZygpuViewDesc*  su_ext_view_desc(void);
/* su_render_iterations restricted to film rows [y_begin, y_end): the unit ranks shard. */
int32_t su_ext_render_rows(uint32_t num_steps, int32_t y_begin, int32_t y_end);
/* Inner-ABI context of the engine (valid after su_start_frame / su_render_frame). */
ZygpuContext* su_ext_context(void);
/* The GGX albedo tables Substitute scenes are compiled with (replaces the constants of material/ggx_integral.zig:3,166,
 * 175,979, 1048): which = 0 E_m[32*32], 1 E_m_avg[32], 2 E[16*16*16], 3 E_avg[16*16], 4 E_s[16*16*16].  Copies up to capacity floats,
 * returns the table's length (0 for an unknown table).  Host-only: needs no device. */
uint32_t su_ext_ggx_lut(uint32_t which, float* out, uint32_t capacity);
/* The image encoders behind su_export_frame on a caller-provided Float4 image (core/exporting/image_sequence.zig,
 * image/encoding/{srgb,png/png_writer,exr/exr_writer,rgbe/rgbe_writer}.zig): format 0 PNG (flags bit 0 = alpha,
 * bit 1 = error diffusion), 1 EXR (bit 0 = alpha, bit 2 = 32-bit float instead of half), 2 RGBE; crop = x0, y0, x1, y1.
 * Writes up to capacity bytes to out, returns the encoded size.  Host-only: needs no device. */
/* Loads a take file and the scene it names into the (initialised, empty) engine: camera, sampler, integrator, sensor,
 * post-processor and exporters of cli/take_loader.zig:27-283, materials / entities / shapes of util/scene_loader.zig:115-702,
 * JSON and "SUB\0" binary meshes of shape_provider.zig:334-805.  0, or -1 with the reason in the log.  Host-only. */
int32_t su_ext_load_take(const char* take_filename);
/* Portal lights (the reference's scene loader only: util/scene_loader.zig:247-248, 371-397; su_light_create passes no link,
 * capi.zig:477): su_ext_portal_create makes a prop of the built-in shape `shape` (a Rectangle) with the material of light
 * `light_id` (a Dome with an emission map), hides that light's prop and turns the light into a prototype; place the prop, then
 * su_ext_light_create_linked(prop, light_id) makes it the Portal light.  Returns the prop id / 0, or -1. */
int32_t su_ext_portal_create(uint32_t shape, uint32_t light_id);
int32_t su_ext_light_create_linked(uint32_t prop, uint32_t light_id);
/* Shadow catchers (the reference's scene loader only: util/scene_loader.zig:312-313 "shadow_catcher" -> Scene.propSetShadowCatcher,
 * and 612-619 visibility.shadow_catcher_light; su_prop_set_visibility passes false for the latter, capi.zig:545-560).
 * PathtracerMIS treats a hit on a shadow catcher as a hole onto the backplate, darkened where the shadow-catcher lights are
 * occluded (pathtracer_mis.zig:56-70, 108-115, 248-276, 296-337; vertex.zig:252-266).  A light is created with the flag its prop
 * has at that moment (scene.zig:344-372), so set the visibility before su_light_create. */
int32_t su_ext_prop_set_shadow_catcher(uint32_t prop);
int32_t su_ext_prop_set_visibility(uint32_t prop, bool in_camera, bool in_reflection, bool in_sss, bool shadow_catcher_light);
/* The Sky entity of the reference's scene loader (util/scene_loader.zig:262-263, 704-712 -> Scene.createSky; the reference's C
 * ABI has no entry for it): creates the sun (a Distant light) and the sky (a Canopy light) once, applies `parameters_json`
 * (Sky.setParameters, sky.zig:78-93: {"sun": {"rotation": [x, y, z]}, "turbidity" | "visibility", "albedo"}; NULL: the
 * defaults) and returns the sun's prop id, or -1.  Scene.compile places both from the sun's rotation (sky.zig:95-134). */
int32_t su_ext_sky_create(const char* parameters_json);
/* The products of the sky bake (sky.zig:172-269; what the reference keeps in its cache files, sky.zig:138-168): the sun's
 * radiance over its disc, 1024 rgb samples (Sky.BakeDimensionsSun), and the canopy image, sky_width x sky_height rgb floats
 * (1024 x 1024 in the reference).  Without this call compile bakes them as the reference does (csrc/host/sky_bake.hpp over this
 * library's own evaluator of the Prague model, csrc/host/prague_sky.hpp) when sky/SkyModelDataset.dat.gz is found through the
 * mounts; otherwise it fills in a procedural stand-in and says so, with the reason, in the log: the model's dataset does not ship
 * with the reference. */
int32_t su_ext_sky_tables(const float* sun_rgb, const float* sky_rgb, uint32_t sky_width, uint32_t sky_height);
/* The Prague sky model's evaluator by itself (csrc/host/prague_sky.hpp; what the reference gets from the C it links,
 * thirdparty/arpraguesky/ArPragueSkyModelGround.h:170-216), so that it can be checked query by query: `count` queries of four doubles
 * against the dataset file `dataset` (gzip or plain) for the state (solar elevation in radians, visibility in km, ground albedo).
 * what = 0: sky radiance of (theta, gamma, shadow, wavelength); 1: solar radiance of (theta, wavelength, -, -); 2: transmittance of
 * (theta, wavelength, distance, -).  out = count doubles.  0, or -1 when the dataset cannot be read.  Host-only: needs no device. */
int32_t su_ext_sky_model_eval(const char* dataset, double elevation, double visibility, double albedo, uint32_t what, const double* queries,
                              uint32_t count, double* out);
uint64_t su_ext_encode_image(uint32_t format, uint32_t flags, uint32_t width, uint32_t height, const int32_t* crop, const float* rgba,
                             uint8_t* out, uint64_t capacity);

#ifdef __cplusplus
}
#endif

#endif /* ZYG_CAPI_H */
