/*
 * zygpu.h — inner C ABI of the MI355X wavefront path-tracing core.
 *
 * This is the seam at which the reference hands tiles to its CPU workers:
 * after `Scene.compile` + `camera.update` (reference src/core/rendering/driver.zig:154-180)
 * and instead of `threads.runParallel(self, renderTiles, 0)` (driver.zig:293-348).
 * The reference has no plugin interface at this point (integrator dispatch is a closed
 * Zig tagged union, src/core/rendering/integrator/surface/integrator.zig:10-21), so the
 * entry points below are what a Zig host would bind with `extern fn`.
 *
 * Everything is plain C: pointers + sizes, caller-owned host memory, int return
 * (0 = ok, negative = error).  All arrays describe the *compiled* scene of the reference
 * (SURVEY.md Appendix B), camera-relative (origin already subtracted,
 * src/core/scene/space.zig:99-108), static transforms only.
 */
#ifndef ZYGPU_H
#define ZYGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZYGPU_NULL 0xFFFFFFFFu

/* reference src/core/scene/bvh/node.zig:9-20 — 32 B BVH2 node; children are (a, a+1); leaf iff count != 0 */
typedef struct ZygpuBvhNode {
    float    min[3];
    uint32_t children_or_first;
    float    max[3];
    uint32_t count;
} ZygpuBvhNode;

/* reference src/core/scene/composed_transformation.zig:11-29 — rotation rows, row.w = scale component; position */
typedef struct ZygpuTrafo {
    float r[3][4];
    float position[4];
} ZygpuTrafo;

/* reference src/base/math/aabb.zig:11 + cacheRadius (137-144): max[3] = cached radius, min[3] = 0
 * (light AABBs: min[3] = light power, src/core/scene/scene.zig:496) */
typedef struct ZygpuAabb {
    float min[4];
    float max[4];
} ZygpuAabb;

/* shape classes: built-in resource ids of the reference (src/core/resource/manager.zig:36-44) */
enum ZygpuShape {
    ZYGPU_SHAPE_CANOPY = 0,
    ZYGPU_SHAPE_CUBE = 1,
    ZYGPU_SHAPE_DISK = 2,
    ZYGPU_SHAPE_DISTANT = 3,
    ZYGPU_SHAPE_DOME = 4,
    ZYGPU_SHAPE_RECTANGLE = 5,
    ZYGPU_SHAPE_SPHERE = 6,
    ZYGPU_SHAPE_TRIANGLE_MESH = 7 /* prop.mesh indexes ZygpuSceneDesc.meshes */
};

/* reference src/core/scene/prop/prop.zig:23-49 — Properties bits, same order */
enum ZygpuPropFlags {
    ZYGPU_PROP_VISIBLE_IN_CAMERA = 1u << 0,
    ZYGPU_PROP_VISIBLE_IN_REFLECTION = 1u << 1,
    ZYGPU_PROP_VISIBLE_IN_SHADOW = 1u << 2,
    ZYGPU_PROP_VISIBLE_IN_SSS = 1u << 3,
    ZYGPU_PROP_EVALUATE_VISIBILITY = 1u << 4,
    ZYGPU_PROP_UNOCCLUDING = 1u << 5,
    ZYGPU_PROP_SOLID = 1u << 6,
    ZYGPU_PROP_VOLUME = 1u << 7,
    ZYGPU_PROP_CAUSTIC = 1u << 8,
    ZYGPU_PROP_STATIC = 1u << 9,
    ZYGPU_PROP_INSTANCER = 1u << 10,
    ZYGPU_PROP_SHADOW_CATCHER = 1u << 11,
    ZYGPU_PROP_SHADOW_CATCHER_LIGHT = 1u << 12
};

/* reference src/core/scene/prop/prop.zig:51-53 + scene.zig:84 (prop_parts) */
typedef struct ZygpuProp {
    uint32_t shape;       /* enum ZygpuShape */
    uint32_t flags;       /* enum ZygpuPropFlags */
    uint32_t parts_start; /* index into material_ids / light_ids (scene.zig:539-556) */
    uint32_t mesh;        /* index into meshes, or ZYGPU_NULL */
} ZygpuProp;

/* reference src/core/scene/shape/triangle/triangle_tree.zig:25-27 + triangle_data.zig:22-29 */
typedef struct ZygpuMesh {
    const ZygpuBvhNode* nodes;
    const uint32_t*     triangles; /* 3 vertex indices per triangle, leaf order (builder reorders) */
    const uint16_t*     triangle_parts;
    const float*        positions; /* 3*num_vertices + 1 floats (unaligned 16 B reads, triangle_data.zig:62-64) */
    const uint16_t*     normals;   /* 2 per vertex, oct + snorm16 (base/encoding.zig:82-108) */
    const float*        uvs;       /* 2 per vertex */
    const float*        part_areas;
    const uint32_t*     primitive_mapping; /* triangle -> its index among the triangles of its part (triangle_mesh.zig:729-746);
                                              NULL unless a part of the mesh is a light */
    uint32_t num_nodes;
    uint32_t num_triangles;
    uint32_t num_vertices;
    uint32_t num_parts;
} ZygpuMesh;

/* material classes: union order of the reference (src/core/scene/material/material.zig:35-43) */
enum ZygpuMaterialKind {
    ZYGPU_MAT_DEBUG = 0,
    ZYGPU_MAT_GLASS = 1,
    ZYGPU_MAT_HAIR = 2,
    ZYGPU_MAT_LIGHT = 3,
    ZYGPU_MAT_SKY = 4,
    ZYGPU_MAT_SUBSTITUTE = 5,
    ZYGPU_MAT_VOLUMETRIC = 6
};

/* reference src/core/scene/material/material_base.zig:13-22 */
enum ZygpuMaterialFlags {
    ZYGPU_MATF_TWO_SIDED = 1u << 0,
    ZYGPU_MATF_EVALUATE_VISIBILITY = 1u << 1,
    ZYGPU_MATF_CAUSTIC = 1u << 2,
    ZYGPU_MATF_EMISSIVE = 1u << 3,
    /* Substitute with anything beyond the uniform opaque base: a texture in one of `tex`, a normal map, a coating,
     * translucency, flakes or a rotated tangent frame (set by the host; the device runs its complete closure for it) */
    ZYGPU_MATF_EXTENDED = 1u << 4,
    /* one of the material's textures (the colour map or an entry of `tex`) is procedural (ZYGPU_TEX_PROCEDURAL; set by the host):
     * the device evaluates them per vertex in wf_textures before the closure is built */
    ZYGPU_MATF_PROCEDURAL = 1u << 5,
    /* the description the material was made from asked for something that is not built (a procedural texture as a mask or an
     * emission map, an image file that cannot be read ...): zygpu_upload_scene refuses a scene that uses it */
    ZYGPU_MATF_UNSUPPORTED = 1u << 31
};

/* Uniform-texture subset of the reference materials (all colours ACES AP1, base/json.zig:247-277).
 * Debug: color_a/color_b = front/back albedo (debug_material.zig:11-12);
 * Substitute: color_a = color, see substitute_material.zig:32-60; emittance: light/emittance.zig:18-27;
 * Glass: dielectric, thick (smooth or rough: roughness; dispersive with an Abbe number `abbe`) or a thin pane (thickness > 0):
 * glass_material.zig:46-115, glass_sample.zig:33-538 */

/* A texture of the reference (texture/texture.zig:14-60) that is either uniform (image == ZYGPU_NULL: the material's uniform
 * field of the same name) or a 2-D image read at the fragment's uv */
typedef struct ZygpuTexture {
    uint32_t image;   /* index into ZygpuSceneDesc.images */
    uint32_t sampler; /* ZYGPU_TEX_* bits */
    float    scale[2];
} ZygpuTexture;

/* Substitute's textures beyond the colour map (substitute_material.zig:37-48; Glass uses ROUGHNESS, SPECULAR and NORMAL,
 * glass_material.zig:22-24; one channel each: Byte1 / Float1 images; the
 * two normal maps: two channels, Byte2 read as snorm or Float2, material_helper.zig:16-20) */
enum ZygpuSubstituteTexture {
    ZYGPU_STEX_ROUGHNESS = 0,
    ZYGPU_STEX_METALLIC,
    ZYGPU_STEX_SPECULAR,
    ZYGPU_STEX_ROTATION,
    ZYGPU_STEX_TRANSLUCENCY,
    ZYGPU_STEX_COATING_SCALE,
    ZYGPU_STEX_COATING_ROUGHNESS,
    ZYGPU_STEX_FLAKES_COVERAGE,
    ZYGPU_STEX_NORMAL,
    ZYGPU_STEX_COATING_NORMAL,
    ZYGPU_STEX_COUNT
};

typedef struct ZygpuMaterial {
    uint32_t kind;
    uint32_t flags;
    float    color_a[3];
    float    color_b[3];
    float    roughness;
    float    metallic;
    float    specular;
    float    ior;
    float    anisotropy;
    float    emission[3]; /* emittance.value */
    float    cos_a;
    float    camera_weight;
    uint32_t emission_normalize;
    uint32_t emission_num_samples;
    uint32_t color_image;   /* Substitute colour map: index into ZygpuSceneDesc.images, ZYGPU_NULL = the uniform color_a */
    uint32_t color_sampler; /* ZYGPU_TEX_* bits (texture/sampler_mode.zig:7-68) */
    float    color_scale[2];/* texture.data.image.scale */
    /* Glass (glass_material.zig:19-47): the absorption coefficient ccoef.attenuationCoefficient(attenuation_color,
     * attenuation_distance) (collision_coefficients.zig:35-44); ior / roughness / specular above */
    float    absorption[3];
    float    thickness;
    float    abbe;
    int32_t  priority;      /* material_base.zig:27: nested dielectrics, the higher priority wins (sample_base.zig:44) */
    /* Light: emittance.emission_map as an image (emittance.zig:19, 49): radiance = emission * texel */
    uint32_t emission_image;   /* index into ZygpuSceneDesc.images, ZYGPU_NULL = uniform */
    uint32_t emission_sampler; /* ZYGPU_TEX_* bits */
    float    emission_scale[2];
    /* material_base.zig:29 `mask`: a one-channel image of opacities, evaluated stochastically by the closest-hit traversal
     * (stochasticOpacity, material_base.zig:44-55) and as a transmittance by shadow rays (material.zig:238-245) */
    uint32_t mask_image;   /* index into ZygpuSceneDesc.images, ZYGPU_NULL = opaque */
    uint32_t mask_sampler; /* ZYGPU_TEX_* bits */
    float    mask_scale[2];
    /* Substitute beyond the uniform opaque base (substitute_material.zig:37-60, 114-221; material_provider.zig:262-342).
     * Subsurface scattering (attenuation_distance > 0: a scattering volume) is refused by the host */
    ZygpuTexture tex[ZYGPU_STEX_COUNT];
    float    rotation;          /* "anisotropy_rotation", in turns */
    float    translucency;
    float    coating_scale;     /* 1 */
    float    coating_roughness; /* 0.2 */
    float    coating_thickness; /* 0 = uncoated */
    float    coating_ior;       /* 1.5 */
    float    coating_absorption[3]; /* setCoatingAttenuation: ccoef.attenuationCoefficient(color, distance) */
    float    flakes_coverage;   /* 0 = none */
    float    flakes_color[3];   /* 0.8 */
    float    flakes_alpha;      /* setFlakesRoughness: clampRoughness(r)^2, 0.01 */
    float    flakes_res;        /* setFlakesSize: max(4, ceil(sqrt(1.5396 / size^2 / 4))) */
    float    attenuation_distance; /* > 0: subsurface scattering, refused (compile returns -7) */
    /* emittance.profile (emittance.zig:20, 45-52): a one-channel image over the octahedral map of the directions from the light (in its
     * object space) to the point it lights: an IES photometric file resampled by the host (ies_reader.zig); ZYGPU_NULL = none */
    uint32_t profile_image;
    uint32_t profile_sampler; /* ZYGPU_TEX_* bits */
    float    profile_scale[2];
} ZygpuMaterial;

/* ZYGPU_TEX_SNORM: a Byte2 image read as snorm (byte / 128 - 1, base/encoding.zig:18-20): what usage Normal makes of it
 * (texture_provider.zig:70) */
/* ZYGPU_TEX_TRIPLANAR / OBJECTPOS: the sampler's tex_coord (sampler_mode.zig, material_provider.zig:664-674): Triplanar reads
 * the image at the fragment's object-space position projected along the dominant axis of its geometric normal
 * (renderstate.zig:44-49), for a normal map both also choose the frame the mapped normal lives in (material_helper.zig:24-41);
 * Substitute's and Glass' textures only (masks take the uv, material_base.zig:35-41) */
/* ZYGPU_TEX_PROCEDURAL: the texture is procedural (texture.zig:60-63, 91-93): `image` indexes ZygpuSceneDesc.procedurals, the
 * address and tex_coord bits are the procedural's mode; Substitute's and Glass' colour, weight and normal textures only */
enum { ZYGPU_TEX_LINEAR = 1u, ZYGPU_TEX_REPEAT_U = 2u, ZYGPU_TEX_REPEAT_V = 4u, ZYGPU_TEX_SNORM = 8u, ZYGPU_TEX_TRIPLANAR = 16u,
       ZYGPU_TEX_OBJECTPOS = 32u, ZYGPU_TEX_PROCEDURAL = 64u };

/* Procedural textures (texture/procedural.zig:27-35, the order of Procedural.Type; material_provider.zig:465-578).
 * Checker (procedural_checker.zig): value[0] / value[1] = the two colours, checker_scale; filtered with the texture-space
 * footprint of a pixel (context.zig:139-200).  Noise (procedural_noise.zig): `levels` octaves of gradient (Perlin) or cellular
 * (Worley) noise over the uv, the triplanar st or the object-space position, remapped by ratio / transition; as a normal map its
 * finite differences.  ChannelMix: value[k] = the row that makes output channel k of the source's colour.  Max / Mul / Mix / DetailNormal
 * combine their children (Mix picks a or b stochastically by its weight: one sampler dimension per evaluation).
 * Children: child[0] = a / base / source, child[1] = b / detail, child[2] = Mix's weight; a child with image == ZYGPU_NULL is the
 * uniform child_uniform[k]; a child may be an image or (ZYGPU_TEX_PROCEDURAL) another procedural, ZYGPU_PROCEDURAL_MAX_DEPTH deep */
enum ZygpuProceduralKind { ZYGPU_PROC_CHANNEL_MIX = 0, ZYGPU_PROC_CHECKER = 1, ZYGPU_PROC_DETAIL_NORMAL = 2, ZYGPU_PROC_MAX = 3,
                           ZYGPU_PROC_MIX = 4, ZYGPU_PROC_MUL = 5, ZYGPU_PROC_NOISE = 6 };
enum { ZYGPU_NOISE_CELLULAR = 1u, ZYGPU_NOISE_ABSOLUTE = 2u, ZYGPU_NOISE_INVERT = 4u };
#define ZYGPU_PROCEDURAL_MAX_DEPTH 4
typedef struct ZygpuProcedural {
    uint32_t kind;  /* ZygpuProceduralKind */
    uint32_t flags; /* ZYGPU_NOISE_* */
    uint32_t levels;
    float    checker_scale;
    float    attenuation, ratio, transition, pad;
    float    scale[4];
    float    period[4];
    float    value[3][4];
    ZygpuTexture child[3];
    float    child_uniform[3][4];
} ZygpuProcedural;

/* 2-D images behind textures (capi.zig:236-298, texture_provider.zig:67-84): Byte3 is read as sRGB and converted to AP1 per
 * fetch (texture.zig:196-199), Float3 is taken as is (213-216); the one-channel formats serve masks: Byte1 is unorm
 * (byte / 255, texture.zig:170, encoding.zig:23), Float1 as is (172) */
enum ZygpuImageFormat { ZYGPU_IMAGE_BYTE3_SRGB = 0, ZYGPU_IMAGE_FLOAT3 = 1, ZYGPU_IMAGE_BYTE1_UNORM = 2, ZYGPU_IMAGE_FLOAT1 = 3,
                        ZYGPU_IMAGE_BYTE2 = 4, ZYGPU_IMAGE_FLOAT2 = 5 }; /* two channels: normal maps (texture.zig:177-186) */
typedef struct ZygpuImage {
    const void* pixels; /* row-major, 1 to 3 bytes or floats per pixel */
    int32_t     width, height;
    uint32_t    format;
} ZygpuImage;

/* reference src/core/scene/light/light.zig:32-47 */
enum ZygpuLightClass { ZYGPU_LIGHT_PORTAL_IMAGE = 0, ZYGPU_LIGHT_PROP = 1, ZYGPU_LIGHT_PROP_IMAGE = 2,
                       ZYGPU_LIGHT_VOLUME = 3, ZYGPU_LIGHT_VOLUME_IMAGE = 4 };

typedef struct ZygpuLight {
    uint32_t prop;
    uint32_t part;
    uint32_t num_samples; /* shape_sampler.num_samples (shape/shape_sampler.zig:28-41) */
    uint8_t  light_class;
    uint8_t  two_sided;
    uint8_t  shadow_catcher_light;
    uint8_t  prototype;
    uint32_t sampler;     /* index into ZygpuSceneDesc.mesh_lights for TriangleMesh emitters, else ZYGPU_NULL (light.zig:41-47) */
} ZygpuLight;

/* reference src/core/scene/light/light_tree.zig:25-35 — 32 B compressed node */
typedef struct ZygpuLightNode {
    uint16_t center[4]; /* unorm16 of (centre - bounds.min)/extent; [3] = radius / bounds radius */
    uint16_t cone[4];   /* snorm16 axis.xyz, cos */
    float    power;
    float    variance;
    uint32_t meta;      /* bit0 has_children, bit1 two_sided, bits 2.. children_or_light */
    uint32_t num_lights;
} ZygpuLightNode;

/* reference src/core/scene/light/light_tree.zig:247-273 (Tree) */
typedef struct ZygpuLightTree {
    const ZygpuLightNode* nodes;
    const uint32_t*       node_middles;
    const uint32_t*       light_orders;
    const uint32_t*       light_mapping;
    const float*          infinite_cdf; /* Distribution1D cdf over the infinite lights' powers, num_infinite_lights + 1 entries
                                           (light_tree.zig:260, base/math/distribution_1d.zig:87-124); NULL without infinite lights */
    ZygpuAabb bounds;
    float     infinite_weight;
    float     infinite_guard;
    uint32_t  infinite_end;
    uint32_t  max_split_depth;
    uint32_t  num_lights;
    uint32_t  num_infinite_lights;
    uint32_t  num_nodes;
} ZygpuLightTree;

/* An emissive part of a triangle mesh, in the mesh's object space: shape_sampler.zig:160-258 (MeshImpl) +
 * triangle_mesh.zig:46-150 (Part.configure).  The primitive tree is light_tree.zig:520-720 (PrimitiveTree: split depth 6,
 * no infinite lights), stored in the scene tree's layout. */
typedef struct ZygpuMeshLight {
    const uint32_t* triangle_mapping; /* part-local triangle -> triangle of the mesh (leaf order) */
    const float*    tri_props;        /* 8 per part-local triangle: centre.xyz, radius, normal.xyz, relative power
                                         (MeshImpl.lightProperties, shape_sampler.zig:199-223) */
    ZygpuLightTree  tree;
    uint32_t mesh;
    uint32_t part;
    uint32_t num_triangles;
    uint32_t two_sided;
} ZygpuMeshLight;

/* Importance sampling of an emission map (light_material.zig:52-125 prepareSampling -> shape_sampler.zig:128-152 ImageImpl):
 * a Distribution2D (base/math/distribution_2d.zig) over the texels' MIS-compensated luminance x shape.uvWeight.
 * `rows`: 4 words per Distribution1D -- offset into cdf, size (entries of its cdf), offset into lut, lut_range (float bits);
 * the conditional rows first, the marginal last (distribution_1d.zig:19-27) */
typedef struct ZygpuImageLight {
    const float*    cdf;
    const uint32_t* lut;
    const uint32_t* rows;
    uint32_t num_rows;      /* conditional distributions = image height */
    float    total_weight;  /* sum of the uv weights */
    uint32_t sampler;       /* ZYGPU_TEX_* address bits of the emission map (ImageImpl.mode) */
    uint32_t num_cdf, num_lut;
} ZygpuImageLight;

/* A Portal light (scene/shape/portal.zig; shape_sampler.zig:263-299 PortalImpl; light_material.zig:142-199
 * preparePortalSampling): a Rectangle that is a window onto an environment map.  Its material is the linked light's (a Dome with
 * an emission map); `func` tabulates the map's luminance as the portal sees it, in the portal's own rectified image space
 * (dim x dim, dim = the map's height), `sat` is its summed-area table ((dim + 1)^2, first row and column zero): a
 * WindowedDistribution2D (base/math/windowed_distribution_2d.zig), sampled inside the window a shading point sees the
 * portal through.  `light_link`: the linked (prototype) light, whose prop's rotation maps a direction to the map's uv. */
typedef struct ZygpuPortalLight {
    const float* sat;
    const float* func;
    uint32_t dim;
    uint32_t light_link;
} ZygpuPortalLight;

/* top-level BVH over props (reference src/core/scene/prop/prop_tree.zig:22-27) */
typedef struct ZygpuPropTree {
    const ZygpuBvhNode* nodes;
    const uint32_t*     indices;
    uint32_t num_nodes;
    uint32_t num_indices;
} ZygpuPropTree;

/* albedo LUTs of the GGX energy compensation (reference material/ggx_integral.zig:3,166,175,979,1048) */
typedef struct ZygpuLuts {
    const float* e_m;     /* 32 x 32 */
    const float* e_m_avg; /* 32 */
    const float* e;       /* 16 x 16 x 16 */
    const float* e_avg;   /* 16 x 16 */
    const float* e_s;     /* 16 x 16 x 16 */
} ZygpuLuts;

/* reference base/math/transformation.zig: one keyframe of an animated entity (rotation is a quaternion x, y, z, w) */
typedef struct ZygpuKeyframe {
    float position[4];
    float scale[4];
    float rotation[4];
} ZygpuKeyframe;

typedef struct ZygpuSceneDesc {
    const ZygpuProp*     props;
    const ZygpuTrafo*    trafos; /* per prop, camera-relative (space.zig:99-108) */
    const ZygpuAabb*     aabbs;  /* per prop, camera-relative world bounds (space.zig:60-97) */
    const uint32_t*      material_ids; /* per (prop.parts_start + part) */
    const uint32_t*      light_ids;    /* per (prop.parts_start + part), ZYGPU_NULL if none */
    const ZygpuMesh*     meshes;
    const ZygpuMaterial* materials;
    const ZygpuLight*    lights;
    const ZygpuAabb*     light_aabbs; /* min[3] = power, max[3] = radius (scene.zig:654-676) */
    const float*         light_cones; /* 4 per light: axis.xyz, cos */
    ZygpuPropTree        solid;       /* finite occluding props (scene.zig:77) */
    ZygpuPropTree        unoccluding; /* finite, purely emissive `Light` entities (scene.zig:78) */
    ZygpuLightTree       light_tree;
    const ZygpuMeshLight* mesh_lights;
    const ZygpuImageLight* image_lights; /* ZygpuLight.sampler of the ZYGPU_LIGHT_PROP_IMAGE lights */
    const uint32_t*      infinite_props; /* props with an infinite shape (scene.zig:76), tested when a ray escapes */
    const ZygpuImage*    images;
    const float*         srgb_to_float; /* 256 entries: spectrum/srgb.zig:28-38 of i/255 (texture/encoding.zig:11,41-51) */
    ZygpuLuts            luts;
    float    specular_threshold; /* resources.specular_threshold (resource/manager.zig:57) */
    uint32_t num_props;
    uint32_t num_parts;  /* length of material_ids / light_ids */
    uint32_t num_meshes;
    uint32_t num_materials;
    uint32_t num_lights;
    uint32_t num_mesh_lights;
    uint32_t num_images;
    uint32_t num_infinite_props;
    uint32_t num_image_lights;
    /* Animated props (su_prop_set_transformation_frame; space.zig:23-26): prop_frames[p] = the prop's first keyframe in
     * `keyframes`, or ZYGPU_NULL for a static prop (NULL pointer: no prop is animated).  num_interpolation_frames keyframes per
     * animated prop (scene.zig:254-257), one tick (1/60 s) apart from the frame's start, in world space; `origin` is
     * subtracted after the interpolation (space.zig:114-125).  An animated prop's flags lack ZYGPU_PROP_STATIC, its `trafos`
     * entry is the transformation at the frame's start and its `aabbs` entry bounds the whole motion (space.zig:60-97).
     * Lights must be static. */
    const ZygpuKeyframe* keyframes;
    const uint32_t*      prop_frames;
    uint32_t             num_keyframes;
    uint32_t             num_interpolation_frames;
    float                origin[4];
    /* The Sky entity's sun (sky/sky.zig:36-76, sky_material.zig:40-168): a Distant light whose material is of kind ZYGPU_MAT_SKY
     * with no emission map.  Its radiance varies over the sun's disc: Material.evaluateRadiance = sun_radiance.eval(sunV(
     * trafo.rotation, wi)), an InterpolatedFunction1D over [0, 1] (interpolated_function.zig:11-47) of num_sun_radiance samples
     * (Sky.BakeDimensionsSun = 1024), 4 floats each (rgb, 0): the baked table of sky.zig:214-221 (or what stands in for it).
     * ZygpuMaterial.emission of that material is the table's average_emission (sky_material.zig:66-87), which the light's
     * power is computed from.  The sky itself is a Canopy with an ordinary emission map (the baked 1024^2 image).
     * NULL / 0: the scene has no Sky. */
    const float*         sun_radiance;
    uint32_t             num_sun_radiance;
    /* ZygpuLight.sampler of the ZYGPU_LIGHT_PORTAL_IMAGE lights */
    uint32_t             num_portal_lights;
    const ZygpuPortalLight* portal_lights;
    /* what the ZYGPU_TEX_PROCEDURAL textures index */
    const ZygpuProcedural* procedurals;
    uint32_t             num_procedurals;
} ZygpuSceneDesc;

enum ZygpuIntegrator { ZYGPU_INTEGRATOR_PT = 1, ZYGPU_INTEGRATOR_PTDL = 2, ZYGPU_INTEGRATOR_PTMIS = 3 };
enum ZygpuSampler { ZYGPU_SAMPLER_RANDOM = 0, ZYGPU_SAMPLER_SOBOL = 1 };

#define ZYGPU_FILTER_TAPS 30 /* reference sensor.zig:72: InterpolatedFunction1DN(30) */

/* camera (camera_perspective.zig:79-150), integrator (take/take.zig:131-271) and sensor (sensor.zig:27-120) state */
/* reference rendering/sensor/aov/aov_value.zig:11-21; resolved per class as aov_buffer.zig:48-78 says */
enum ZygpuAovClass { ZYGPU_AOV_ALBEDO = 0, ZYGPU_AOV_DEPTH = 1, ZYGPU_AOV_MATERIAL_ID = 2, ZYGPU_AOV_GEOMETRIC_NORMAL = 3,
                     ZYGPU_AOV_SHADING_NORMAL = 4, ZYGPU_AOV_ROUGHNESS = 5, ZYGPU_AOV_EMISSION = 6, ZYGPU_AOV_DIRECT = 7,
                     ZYGPU_AOV_INDIRECT = 8, ZYGPU_AOV_NUM_CLASSES = 9 };

#ifndef ZYGPU_MAX_CAMERA_KEYFRAMES
#define ZYGPU_MAX_CAMERA_KEYFRAMES 64 /* frames of up to 63 ticks of 1/60 s: a 1-second exposure and more */
#endif

typedef struct ZygpuViewDesc {
    int32_t  resolution[2];
    int32_t  crop[4]; /* x0, y0, x1, y1 (exclusive end), camera_base.zig:31-41 */
    float    left_top[4];
    float    d_x[4];
    float    d_y[4];
    float    eye_offset[4];
    float    lens_radius;
    float    focus_distance;
    ZygpuTrafo camera_trafo; /* camera prop transformation, camera-relative */
    uint64_t frame_step;     /* ticks, camera_base.zig:25 */
    uint64_t frame_duration; /* ticks, camera_base.zig:26 */
    uint32_t integrator;     /* enum ZygpuIntegrator */
    uint32_t max_depth_surface;
    uint32_t max_depth_volume;
    float    split_threshold; /* already st^4 (take.zig:263-271) */
    uint32_t caustics_path;
    float    regularize_roughness;
    uint32_t sampler;         /* enum ZygpuSampler */
    uint32_t num_expected_samples; /* view.num_samples_per_pixel (driver.zig:298) */
    int32_t  filter_radius_int;    /* sensor.filter_radius_int */
    float    filter_radius;        /* range end of the tabulated filter */
    float    filter_samples[ZYGPU_FILTER_TAPS + 1]; /* normalised table (base/math/interpolated_function.zig) */
    float    clamp_emission, clamp_direct, clamp_indirect; /* sensor.zig:60-70 */
    float    exposure_factor; /* 2^exposure, tonemapper.zig:15 */
    uint32_t alpha_transparency; /* sensor buffer class Transparent (take_loader.zig:193-196, buffer_transparent.zig) */
    uint32_t tonemapper;      /* enum ZygpuTonemapper (tonemapper.zig:9) */
    /* AOVs (sensor/aov/aov_value.zig:11-48): bit k = class k is recorded; ZygpuAovClass */
    uint32_t aov_slots;
    /* Animated camera (su_prop_set_transformation_frame on the camera entity; space.zig:23-26, 99-137): keyframes one tick
     * (1/60 s, space.zig:19) apart, the first at the frame's start (scene.zig:186-187), in world space.  A camera sample's
     * shutter time picks the interval and the weight (camera_base.zig:121-136, space.zig:127-137), the two keyframes are
     * interpolated (transformation.zig:36-44: lerp, lerp, slerp) and the scene origin (= keyframe 0's position, everything
     * else in the scene is relative to it) is subtracted from the result (space.zig:121-124).  0 = static: camera_trafo. */
    uint32_t      camera_num_keyframes;
    float         camera_origin[4];
    ZygpuKeyframe camera_keyframes[ZYGPU_MAX_CAMERA_KEYFRAMES];
    /* The aperture's shape (camera/aperture.zig:22-53, camera_perspective.zig:270-283): the entry of the scene's `image_lights` that
     * holds the Distribution2D over the shape image's values (no uv weight, no compensation, total_weight 1), or ZYGPU_NULL: the
     * lens is sampled as a disc.  A lens sample is 2 * distribution.sampleContinuous(lens_uv).uv - 1, times lens_radius. */
    uint32_t      aperture_shape;
    /* The shutter (take_loader.zig:153-191 loadShutter -> camera_base.zig:43-75 setShutter): 1 = a camera sample's time goes
     * through the Distribution1DN(63) whose cdf follows (camera_base.zig:121-127 sampleShutterTime, distribution_1d.zig:188-205:
     * linear search, offset + (cdf[o + 1] - r) / (cdf[o + 1] - cdf[o]), over 63); 0 = the time is the sample itself. */
    uint32_t      shutter_configured;
    float         shutter_cdf[64];
} ZygpuViewDesc;

/* tonemapper.zig:9; AgX with the look the take loader picks (take_loader.zig:268-270: Substitute) */
enum ZygpuTonemapper { ZYGPU_TONEMAP_LINEAR = 0, ZYGPU_TONEMAP_ACES = 1, ZYGPU_TONEMAP_AGX = 2, ZYGPU_TONEMAP_PBR_NEUTRAL = 3 };

/* traversal / path statistics (no reference counterpart; they define the algorithmic bytes of SURVEY §8(d)) */
typedef struct ZygpuStats {
    uint64_t camera_samples;
    uint64_t extension_rays;
    uint64_t shadow_rays;
    uint64_t nodes_fetched;      /* both children of every inner step count (triangle_tree.zig:76-80) */
    uint64_t triangles_tested;
    uint64_t instances_tested;
} ZygpuStats;

typedef struct ZygpuContext ZygpuContext;

/* Creates a device context on HIP device `device_id`.  Replaces Driver.init (driver.zig:60-72). */
int zygpu_create(int device_id, ZygpuContext** out_ctx);
int zygpu_destroy(ZygpuContext* ctx);

/* Copies the compiled scene to HBM (re-laid out for wave64 traversal).  Replaces what
 * Scene.compile leaves in host memory for the workers (scene.zig:185-223). */
int zygpu_upload_scene(ZygpuContext* ctx, const ZygpuSceneDesc* scene);

/* 1 when a surface of a Glass mesh lies in the plane of another solid surface (within 1e-5 of the scene's extent; a glass box standing
 * on the floor): the condition under which option "pool_relaxed" -1 keeps a scene in the precise units (see zygpu_set_option).  What
 * zygpu_upload_scene computes for the scenes of the POOL kernels; host-only: needs no device. */
int zygpu_scene_has_coincident_dielectric(const ZygpuSceneDesc* scene);

/* Camera/integrator/sensor state.  Replaces Driver.configure + camera.update (driver.zig:86-113,167-169). */
int zygpu_set_view(ZygpuContext* ctx, const ZygpuViewDesc* view);

/* buffer.clear(weight) (buffer_opaque.zig:26-30) on the device film. */
int zygpu_clear_film(ZygpuContext* ctx, float weight);

/* Driver.renderIterations semantics in progressive form (driver.zig:182-187,338-348): renders iterations
 * [iteration, iteration + num_iterations), one camera sample per pixel and iteration, each (pixel, iteration)
 * reseeded as worker.zig:141-149, and accumulates into the device film.  Rows [y_begin, y_end) of the crop
 * window only (whole frame: 0, resolution[1]); this is what lets ranks shard the frame.
 * Asynchronous on the context's stream. */
int zygpu_render(ZygpuContext* ctx, uint32_t frame, uint32_t iteration, uint32_t num_iterations,
                 int32_t y_begin, int32_t y_end);

/* One-shot rendering: Worker.render with num_samples > 1 (reference src/core/rendering/worker.zig:141-165), what
 * Driver.render runs for su_render_frame (driver.zig:115-152).  The samples of a pixel are the progressive ones except that
 * the pixel's PCG stream (bounces >= 3, worker.zig:201-207) is started once, for `iteration`, and runs on across the samples.
 * Same film layout and accumulation as zygpu_render. */
int zygpu_render_frame(ZygpuContext* ctx, uint32_t frame, uint32_t iteration, uint32_t num_samples, int32_t y_begin, int32_t y_end);

/* Waits for all queued work on the context's stream. */
int zygpu_synchronize(ZygpuContext* ctx);

/* Raw film {sum w*r, sum w*g, sum w*b, sum w} (buffer_opaque.zig:12-13), resolution[0]*resolution[1]*4 floats. */
int zygpu_download_film(ZygpuContext* ctx, float* host_rgbw);
int zygpu_upload_film(ZygpuContext* ctx, const float* host_rgbw);

/* Device pointer of the film (for an RCCL reduce by the caller); not owned by the caller. */
int zygpu_film_device_ptr(ZygpuContext* ctx, void** out_ptr, uint64_t* out_num_floats);
/* Makes the kernels accumulate into a caller-owned device buffer of resolution[0]*resolution[1]*4 floats
 * (e.g. a torch tensor that is then reduced with RCCL in place); NULL switches back to the internal film.
 * Call after zygpu_set_view. */
int zygpu_set_external_film(ZygpuContext* ctx, void* device_rgbw);

/* The alpha plane's device pointer (NULL and 0 floats for an opaque sensor), the context's HIP stream (a hipStream_t: what a caller's
 * own collective on the film must be ordered behind) and its HIP device. */
int zygpu_film_alpha_device_ptr(ZygpuContext* ctx, void** out_ptr, uint64_t* out_num_floats);
int zygpu_stream(ZygpuContext* ctx, void** out_stream);
int zygpu_device(ZygpuContext* ctx);

/* Transparent sensor (ZygpuViewDesc.alpha_transparency): the second plane of buffer_transparent.zig:11-59, one float per
 * pixel = sum of weight * alpha (the film's float4 keeps {sum w r, sum w g, sum w b, sum w}).  download returns -1 for
 * an opaque sensor; an external plane (device pointer, W*H floats) follows the same rules as zygpu_set_external_film. */
int zygpu_download_film_alpha(ZygpuContext* ctx, float* host_alpha);
int zygpu_set_external_film_alpha(ZygpuContext* ctx, void* device_alpha);

/* resolveTonemap (buffer_opaque.zig:73-79 / buffer_transparent.zig:84-94, tonemapper.zig:18-46, Linear): |rgb / w| * exposure
 * -> AP1 -> sRGB; alpha 1, or |sum w alpha / w| for a Transparent sensor; writes min(num_pixels, film pixels) * 4 floats. */
int zygpu_resolve(ZygpuContext* ctx, float* host_rgba, uint32_t num_pixels);

/* Options of the shipped library (twelve):
 * "precise" (default 1): the reference's arithmetic bit for bit (the verification mode: the film is the oracle's); 0: the
 *     relaxed-arithmetic kernel units (native divide / sqrt, contraction, OCML f32 functions, the wide BLAS walks), inside the contract
 *     RMS <= 1e-3 against the reference; bench.py measures these;
 * "pool_relaxed" (default -1): whether the scenes of the POOL kernels (Glass, shadow catchers) follow "precise" 0 into the relaxed
 *     units.  -1: they do unless zygpu_upload_scene finds a dielectric surface lying in the plane of another surface (a glass box
 *     standing on the floor): which of two coincident surfaces a ray meets first hangs on the last bit of two formulas, only the precise
 *     units decide that as the reference does, and such a scene measures 1.3e-3 .. 2e-3 in the relaxed ones; then
 *     it gets the precise units and zygpu_get_option "relaxed_units" says 0.  1: always follow; 0: POOL scenes always precise;
 * "kernel_timing" (default 1): a HIP event pair around every wavefront launch (zygpu_kernel_time);
 * "batch_mpaths" (default 32, 1..512): paths per batch in units of 2^20 (about 450 B of pool per path; POOL scenes: twice as many camera
 *     samples of about 2.9 KB each; bounded by the device's free memory: zygpu_render halves a batch that does not fit);
 * "tick_sync" (default 0): reads the next tick's queue length back after every tick (exact launch sizes, a batch ends with its last
 *     living path) instead of keeping a batch asynchronous with every launch sized for the whole batch;
 * "nee_in_place" (default 1): next-event estimation, shadow rays and the bxdf sample of a vertex in one kernel (wf_scatter); 0: as
 *     wf_nee / wf_shadow rounds + wf_extend: the same film (scenes with multi-sample or TriangleMesh lights always run in place);
 * "fuse_trace_shade" (default 1): closest hit and first shading pass in one kernel (wf_trace_shade); 0: wf_intersect + wf_shade;
 * "uniform_tlas" (default 1): scenes of at most 112 Rectangle / mesh props are walked wave-wide from a table in constant memory;
 *     0: the top-level trees lane by lane;
 * "mesh_kernels" (default 1): in the relaxed units the walks of meshes of 4096 triangles or more run in persistent kernels whose
 *     lanes fetch their own rays (wf_scatter_h suspends the vertices whose shadow ray enters such a mesh's bounds);
 * "bvh8" (default 1): those walkers read the meshes' eight-wide compressed trees (csrc/hip/bvh8.hpp: wf_shadow_u8, wf_mesh_closest8,
 *     the closest hits of the later ticks' rays included); 0, or a mesh the eight-wide form cannot hold: the four-wide wf_shadow_u;
 * "sort_alive" (default -1: the library's choice, on in the scenes of the mesh kernels; 1 / 0): the paths that survive a tick's
 *     shading in one sub-queue per hit prop (16), so that the waves of the kernel that scatters them stand on one surface each;
 * "shadow_u_blocks" (default 4, 1-10): blocks per CU of the persistent mesh walkers.
 * Within the precise mode the film is the same bits for every setting.
 *
 * Variant builds only (python -m zyg_amd.build --variant all --units 4,5,6,7,-1 -DZYG_VARIANTS=1 -> variants/lib_all.so; the shipped
 * library answers -1): the kernels that measured slower on every BASELINE scene and the switches that turn a default off, kept for
 * their tests and A/B tools (DESIGN.md section 4 has the numbers): "persistent" (wf_persistent), "fuse_tick" (wf_tick), "regenerate" /
 * "regen_kslots" (wf_rscatter / wf_rgenerate / wf_rtrace), "mesh_closest" (the four-wide wf_mesh_closest), "compact_blocks";
 * "mesh_closest8", "pack_prev", "hit_by_queue", "lazy_head", "pair_small", "split_materials", "split_nee", "fuse_pool",
 * "pool_tick_sync" (all default 1; 0 = the layout of an earlier round). */
int zygpu_set_option(ZygpuContext* ctx, const char* name, int value);
/* Reads an option back, and what the library derived from the options and the uploaded scene (valid after zygpu_upload_scene):
 * "relaxed_units" 1 when the relaxed-arithmetic kernel units render this scene (option "precise" 0 and a scene they are inside
 * the RMS contract for), 0 when the precise ones do whatever the option says; "pool_mode" 1 when the scene runs the POOL
 * instantiations (Glass / shadow catchers: split paths, medium stack); "prop_tables" / "wide_blas" / "mesh_kernels_active":
 * the prop-table walk, the four-wide BLAS and the persistent mesh kernels are what this scene renders with; "table_complete" 1
 * when the complete kernels (scenes with Spheres, image lights, textures, Glass ...) may walk the prop table wave-wide in their
 * relaxed units: at most 88 static solid props, Spheres / Disks / Cubes among them, no opacity mask and no thin Glass
 * (ZYG_TABLE_COMPLETE=0 in the environment of zygpu_create keeps the per-lane walk of the top-level tree: an A/B switch).
 * bench.py reports `config.arithmetic` from it. -1: unknown name. */
int zygpu_get_option(ZygpuContext* ctx, const char* name, int* out_value);

/* Ray-level test hooks: n rays of 8 floats (o.xyz, min_t, d.xyz, max_t) against the uploaded scene, in the arithmetic option
 * "precise" selects (the hooks live in kernel unit 7 of either mode; -1 when that unit is not in the build).  Modes 1-3 are
 * compared with the oracle ray for ray only with precise = 1; the relaxed units' dump is bounded, not bit-equal.
 * mode 0: the reference's walk (top-level tree, BVH2); 1: the prop tables + BVH2; 2: the prop tables + the four-wide BLAS
 * re-layout; 3: the relaxed units' walk (2 without the tie rule between props, small meshes item by item from the
 * constant block, in their unit-triangle form) (modes 1-3 need a scene the tables hold: -1 otherwise).  hits: 8 words per ray (t, u, v as float bits,
 * primitive, prop or ZYGPU_NULL, 3 x 0), the layout of the oracle's zyg_oracle_intersect; visible: 1 = unoccluded. */
int zygpu_test_intersect(ZygpuContext* ctx, uint32_t n, const float* rays, uint32_t depth_surface, uint32_t mode, uint32_t* hits);
int zygpu_test_visibility(ZygpuContext* ctx, uint32_t n, const float* rays, uint32_t mode, uint32_t* visible);
/* One closest-hit ray (8 floats as above) walked as the reference's Scene.intersect does (mode 0), in precise arithmetic (a development
 * library built without the precise kernel units answers with the relaxed units' walk of the same trees), at the frame's start time:
 * the camera's focus probe (Perspective.updateFocus, camera_perspective.zig:322-347), which the host traces between
 * zygpu_upload_scene and zygpu_set_view.  Returns 1 and the hit distance in *out_t, 0 for a miss, -1 on error. */
int zygpu_probe_closest(ZygpuContext* ctx, const float* ray8, float* out_t);

/* Measurement support (tools/calibrate_fetch.py): `repeats` launches of a kernel whose lane i reads the 16 B record at
 * i * stride_bytes (stride 16 / 64 / 128 / 256 / 512) of a zero-filled buffer of num_records records; average ms per launch.
 * Run under `rocprofv3 --pmc FETCH_SIZE` it calibrates that counter on the access shape of the path-state blocks. */
int zygpu_calibrate_gather(ZygpuContext* ctx, uint32_t stride_bytes, uint32_t num_records, uint32_t repeats, double* out_ms_per_launch);
/* The copy rate this device reaches (bench.py's measured ceiling beside the nominal 8 TB/s): `repeats` launches of a float4
 * grid-stride copy kernel (8 blocks of 256 threads per CU, four 16 B loads in flight per thread) over two buffers of `bytes` each
 * on the context's stream, timed with HIP events; out_gbs = read + written bytes per second / 1e9. */
int zygpu_measure_copy(ZygpuContext* ctx, uint64_t bytes, uint32_t repeats, double* out_gbs);

/* Counters accumulated since the last reset (only maintained when enabled). */
int zygpu_enable_stats(ZygpuContext* ctx, int enabled);
int zygpu_get_stats(ZygpuContext* ctx, ZygpuStats* out_stats);
/* The same counters split by kernel, ZYGPU_NUM_KERNELS entries in the order below: traversal happens in wf_intersect
 * (extension rays), wf_shade (unoccluding-light walk) and wf_shadow or wf_scatter (shadow rays); the rest stay zero. */
int zygpu_get_kernel_stats(ZygpuContext* ctx, ZygpuStats* out_stats);
int zygpu_reset_stats(ZygpuContext* ctx);
/* Profiling builds only (-DZYG_PROFILE_SECTIONS, tools/sections.py): shader cycles per code section summed over waves,
 * [0..15] wf_shade, [16..31] wf_scatter; all zero in the shipped build.  Cleared by zygpu_reset_stats. */
int zygpu_get_section_cycles(ZygpuContext* ctx, uint64_t* out32);

/* Sums of the durations (ms, HIP events on the context's stream, one pair per launch) and launch counts of the
 * wavefront kernels since the last reset; both out arrays hold ZYGPU_NUM_KERNELS entries:
 * 0 wf_intersect, 1 wf_shade, 2 wf_nee, 3 wf_shadow, 4 wf_extend, 5 wf_scatter (only with option "nee_in_place"). */
#define ZYGPU_NUM_KERNELS 6
int zygpu_kernel_time(ZygpuContext* ctx, double* out_ms, uint64_t* out_launches, int reset);

/* Unit-test hook: evaluates small device functions (op: 0 sin/cos/acos, 1 SphQuad, 2 Sobol stream, 3 PCG,
 * 4 offsetRay, 6 exp / atan2, 7 tan; strides in floats per item) on host arrays; see tests/test_gpu_units.py. */
int zygpu_test_eval(ZygpuContext* ctx, uint32_t op, uint32_t n, const float* in, uint32_t in_stride, float* out,
                    uint32_t out_stride);

/* AOVs: Sensor.resolveAov (sensor.zig:491-494, aov_buffer.zig:48-78) of one class into RGBA floats; -2 = the view does not
 * record that class (capi.zig:620, 637). zygpu_download_aov: the class' raw Pack4f plane (tests). */
int zygpu_resolve_aov(ZygpuContext* ctx, uint32_t aov_class, float* host_rgba, uint32_t num_pixels);
int zygpu_download_aov(ZygpuContext* ctx, uint32_t aov_class, float* host_plane);
/* A class' plane on the device (NULL and 0 floats when the view does not record it) and its clear (aov_buffer.zig:36-46: zero; Depth:
 * f32 max), on the context's stream: what a ZygpuGroup reduces and resets. */
int zygpu_aov_device_ptr(ZygpuContext* ctx, uint32_t aov_class, void** out_ptr, uint64_t* out_num_floats);
int zygpu_clear_aov(ZygpuContext* ctx, uint32_t aov_class);

const char* zygpu_last_error(void);

/* ---- Device groups: the GPUs of a node behind this ABI (csrc/hip/group.hip) -----------------------------------
 * Replaces, for N devices, what Driver.render / Driver.renderIterations do with the reference's worker threads (driver.zig:115-152,
 * 293-348: threads.runParallel(self, renderTiles, 0)): one ZygpuContext and one host thread per device, the compiled scene
 * replicated, a call's iterations (ZYGPU_SHARD_ITERATIONS) or rows (ZYGPU_SHARD_ROWS) split into one contiguous share per rank, each
 * rank accumulating into its own film, and one exchange step, zygpu_group_reduce: ncclReduce(sum, float, 4 W H, root 0) of the films
 * over xGMI (RCCL), in stream order behind each rank's kernels; rank 0's context then holds the frame (zygpu_resolve,
 * zygpu_download_film on zygpu_group_context(g, 0)).  (pixel, iteration) samples are independent and the film is additive
 * (buffer_opaque.zig:39-45), so nothing else is exchanged.  Rows: the film equals the single-device film bit for bit with the
 * radius-0 sensor; iterations: up to the order of the float sums.  A view that records AOVs always splits rows (Depth keeps a pixel's
 * minimum and MaterialId its heaviest sample over the pixel's own samples, aov_buffer.zig:101-119: with its rows on one device a pixel's
 * planes are complete there), and the reduce sums the planes too (Depth: ncclMin).
 *
 * zygpu_group_create: n contexts on the listed HIP devices = ranks [first_rank, first_rank + n) of a world of `world` ranks.
 * unique_id NULL: this process is the whole world (first_rank and world are ignored, world = n).  Otherwise the 128 bytes that
 * zygpu_group_unique_id produced in the process of rank 0 and the launcher carried to the others (one process per GPU under
 * torch.distributed.run, an MPI-style host): the call returns when every rank of the world has joined.  A device may be listed
 * twice (how a box with one GPU exercises the group): such members are driven one after the other and their films are added by a
 * kernel instead of RCCL.  flags: ZYGPU_GROUP_ALWAYS_RCCL builds the communicator for a world of one as well.
 * Every process of a world calls zygpu_group_reduce the same number of times.  Failures: -1, text in zygpu_last_error(). */
typedef struct ZygpuGroup ZygpuGroup;
#define ZYGPU_UNIQUE_ID_BYTES 128
enum ZygpuShard { ZYGPU_SHARD_ITERATIONS = 0, ZYGPU_SHARD_ROWS = 1 };
enum ZygpuGroupFlags { ZYGPU_GROUP_ALWAYS_RCCL = 1 };
int zygpu_group_unique_id(uint8_t* out128);
int zygpu_group_create(const int* device_ids, int n, const uint8_t* unique_id, int first_rank, int world, uint32_t flags, ZygpuGroup** out_group);
int zygpu_group_destroy(ZygpuGroup* group);
int zygpu_group_size(const ZygpuGroup* group);       /* members in this process */
int zygpu_group_world(const ZygpuGroup* group);      /* ranks in the world */
int zygpu_group_first_rank(const ZygpuGroup* group);
int zygpu_group_uses_rccl(const ZygpuGroup* group);  /* 1: the reduce is ncclReduce; 0: a world of one, or members sharing a device */
ZygpuContext* zygpu_group_context(ZygpuGroup* group, int member); /* options, statistics, and (rank 0) the film */
int zygpu_group_upload_scene(ZygpuGroup* group, const ZygpuSceneDesc* scene);
int zygpu_group_set_view(ZygpuGroup* group, const ZygpuViewDesc* view); /* with AOVs a world > 1 splits rows whatever `shard` says (below) */
int zygpu_group_set_option(ZygpuGroup* group, const char* name, int value);
int zygpu_group_clear_film(ZygpuGroup* group, float weight);
/* zygpu_render / zygpu_render_frame for the world: [iteration, iteration + num_iterations) and [y_begin, y_end) name the whole job,
 * each rank renders its share.  One-shot rendering always splits rows (a pixel's PCG stream runs on across its samples). */
int zygpu_group_render(ZygpuGroup* group, uint32_t frame, uint32_t iteration, uint32_t num_iterations, int32_t y_begin, int32_t y_end, uint32_t shard);
int zygpu_group_render_frame(ZygpuGroup* group, uint32_t frame, uint32_t iteration, uint32_t num_samples, int32_t y_begin, int32_t y_end);
int zygpu_group_reduce(ZygpuGroup* group);      /* asynchronous with RCCL: zygpu_group_synchronize waits */
/* The split itself (a pure function: needs no device): rank `rank` of `world` takes [first, first + length) of `count` iterations or
 * rows, contiguous shares in rank order, the first count % world of them one longer. */
int zygpu_group_share(uint32_t count, int world, int rank, uint32_t* out_first, uint32_t* out_length);
int zygpu_group_dirty(const ZygpuGroup* group); /* 1: something was rendered since the last reduce (world > 1) */
int zygpu_group_synchronize(ZygpuGroup* group);

#ifdef __cplusplus
}
#endif

#endif /* ZYGPU_H */
