"""zyg_amd — MI355X wavefront path-tracing core behind zyg's `su_*` C ABI.

Python host mirror of the reference's ctypes clients (reference src/capi-test/test.py:45-261,
src/blender-plugin/engine.py:25-381): a thin ctypes layer over `libzyg_amd.so`, which exports the
`su_*` surface (include/zyg_capi.h) and the inner `zygpu_*` ABI (include/zygpu.h).  There is no
CPU fallback: rendering needs the HIP library and a GPU and fails loudly otherwise.
"""
import ctypes
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZYG_AMD_LIB", os.path.join(_HERE, "libzyg_amd.so"))

NULL = 0xFFFFFFFF
FILTER_TAPS = 30
MAX_CAMERA_KEYFRAMES = 64  # ZYGPU_MAX_CAMERA_KEYFRAMES


class BvhNode(ctypes.Structure):
    _fields_ = [("min", ctypes.c_float * 3), ("children_or_first", ctypes.c_uint32),
                ("max", ctypes.c_float * 3), ("count", ctypes.c_uint32)]


class Trafo(ctypes.Structure):
    _fields_ = [("r", (ctypes.c_float * 4) * 3), ("position", ctypes.c_float * 4)]


class Aabb(ctypes.Structure):
    _fields_ = [("min", ctypes.c_float * 4), ("max", ctypes.c_float * 4)]


class Prop(ctypes.Structure):
    _fields_ = [("shape", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("parts_start", ctypes.c_uint32),
                ("mesh", ctypes.c_uint32)]


class Mesh(ctypes.Structure):
    _fields_ = [("nodes", ctypes.POINTER(BvhNode)), ("triangles", ctypes.POINTER(ctypes.c_uint32)),
                ("triangle_parts", ctypes.POINTER(ctypes.c_uint16)), ("positions", ctypes.POINTER(ctypes.c_float)),
                ("normals", ctypes.POINTER(ctypes.c_uint16)), ("uvs", ctypes.POINTER(ctypes.c_float)),
                ("part_areas", ctypes.POINTER(ctypes.c_float)), ("primitive_mapping", ctypes.POINTER(ctypes.c_uint32)),
                ("num_nodes", ctypes.c_uint32),
                ("num_triangles", ctypes.c_uint32), ("num_vertices", ctypes.c_uint32), ("num_parts", ctypes.c_uint32)]


class Texture(ctypes.Structure):
    _fields_ = [("image", ctypes.c_uint32), ("sampler", ctypes.c_uint32), ("scale", ctypes.c_float * 2)]


STEX_COUNT = 10  # enum ZygpuSubstituteTexture
(STEX_ROUGHNESS, STEX_METALLIC, STEX_SPECULAR, STEX_ROTATION, STEX_TRANSLUCENCY, STEX_COATING_SCALE, STEX_COATING_ROUGHNESS,
 STEX_FLAKES_COVERAGE, STEX_NORMAL, STEX_COATING_NORMAL) = range(STEX_COUNT)
# ZYGPU_TEX_* sampler bits, ZYGPU_MATF_* material flags (include/zygpu.h)
TEX_LINEAR, TEX_REPEAT_U, TEX_REPEAT_V, TEX_SNORM, TEX_TRIPLANAR, TEX_OBJECTPOS, TEX_PROCEDURAL = 1, 2, 4, 8, 16, 32, 64
MATF_TWO_SIDED, MATF_EVALUATE_VISIBILITY, MATF_CAUSTIC, MATF_EMISSIVE, MATF_EXTENDED, MATF_PROCEDURAL = 1, 2, 4, 8, 16, 32


class Material(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("color_a", ctypes.c_float * 3),
                ("color_b", ctypes.c_float * 3), ("roughness", ctypes.c_float), ("metallic", ctypes.c_float),
                ("specular", ctypes.c_float), ("ior", ctypes.c_float), ("anisotropy", ctypes.c_float),
                ("emission", ctypes.c_float * 3), ("cos_a", ctypes.c_float), ("camera_weight", ctypes.c_float),
                ("emission_normalize", ctypes.c_uint32), ("emission_num_samples", ctypes.c_uint32),
                ("color_image", ctypes.c_uint32), ("color_sampler", ctypes.c_uint32), ("color_scale", ctypes.c_float * 2),
                ("absorption", ctypes.c_float * 3), ("thickness", ctypes.c_float), ("abbe", ctypes.c_float),
                ("priority", ctypes.c_int32), ("emission_image", ctypes.c_uint32), ("emission_sampler", ctypes.c_uint32),
                ("emission_scale", ctypes.c_float * 2),
                ("mask_image", ctypes.c_uint32), ("mask_sampler", ctypes.c_uint32), ("mask_scale", ctypes.c_float * 2),
                ("tex", Texture * STEX_COUNT), ("rotation", ctypes.c_float), ("translucency", ctypes.c_float),
                ("coating_scale", ctypes.c_float), ("coating_roughness", ctypes.c_float), ("coating_thickness", ctypes.c_float),
                ("coating_ior", ctypes.c_float), ("coating_absorption", ctypes.c_float * 3), ("flakes_coverage", ctypes.c_float),
                ("flakes_color", ctypes.c_float * 3), ("flakes_alpha", ctypes.c_float), ("flakes_res", ctypes.c_float),
                ("attenuation_distance", ctypes.c_float), ("profile_image", ctypes.c_uint32), ("profile_sampler", ctypes.c_uint32),
                ("profile_scale", ctypes.c_float * 2)]


class Image(ctypes.Structure):
    _fields_ = [("pixels", ctypes.c_void_p), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("format", ctypes.c_uint32)]


class Light(ctypes.Structure):
    _fields_ = [("prop", ctypes.c_uint32), ("part", ctypes.c_uint32), ("num_samples", ctypes.c_uint32),
                ("light_class", ctypes.c_uint8), ("two_sided", ctypes.c_uint8),
                ("shadow_catcher_light", ctypes.c_uint8), ("prototype", ctypes.c_uint8), ("sampler", ctypes.c_uint32)]


class LightNode(ctypes.Structure):
    _fields_ = [("center", ctypes.c_uint16 * 4), ("cone", ctypes.c_uint16 * 4), ("power", ctypes.c_float),
                ("variance", ctypes.c_float), ("meta", ctypes.c_uint32), ("num_lights", ctypes.c_uint32)]


class LightTree(ctypes.Structure):
    _fields_ = [("nodes", ctypes.POINTER(LightNode)), ("node_middles", ctypes.POINTER(ctypes.c_uint32)),
                ("light_orders", ctypes.POINTER(ctypes.c_uint32)), ("light_mapping", ctypes.POINTER(ctypes.c_uint32)),
                ("infinite_cdf", ctypes.POINTER(ctypes.c_float)), ("bounds", Aabb), ("infinite_weight", ctypes.c_float), ("infinite_guard", ctypes.c_float),
                ("infinite_end", ctypes.c_uint32), ("max_split_depth", ctypes.c_uint32),
                ("num_lights", ctypes.c_uint32), ("num_infinite_lights", ctypes.c_uint32),
                ("num_nodes", ctypes.c_uint32)]


class MeshLight(ctypes.Structure):
    _fields_ = [("triangle_mapping", ctypes.POINTER(ctypes.c_uint32)), ("tri_props", ctypes.POINTER(ctypes.c_float)),
                ("tree", LightTree), ("mesh", ctypes.c_uint32), ("part", ctypes.c_uint32),
                ("num_triangles", ctypes.c_uint32), ("two_sided", ctypes.c_uint32)]


class ImageLight(ctypes.Structure):
    _fields_ = [("cdf", ctypes.POINTER(ctypes.c_float)), ("lut", ctypes.POINTER(ctypes.c_uint32)),
                ("rows", ctypes.POINTER(ctypes.c_uint32)), ("num_rows", ctypes.c_uint32), ("total_weight", ctypes.c_float),
                ("sampler", ctypes.c_uint32), ("num_cdf", ctypes.c_uint32), ("num_lut", ctypes.c_uint32)]


class PortalLight(ctypes.Structure):
    _fields_ = [("sat", ctypes.POINTER(ctypes.c_float)), ("func", ctypes.POINTER(ctypes.c_float)), ("dim", ctypes.c_uint32),
                ("light_link", ctypes.c_uint32)]


class PropTree(ctypes.Structure):
    _fields_ = [("nodes", ctypes.POINTER(BvhNode)), ("indices", ctypes.POINTER(ctypes.c_uint32)),
                ("num_nodes", ctypes.c_uint32), ("num_indices", ctypes.c_uint32)]


class Luts(ctypes.Structure):
    _fields_ = [("e_m", ctypes.POINTER(ctypes.c_float)), ("e_m_avg", ctypes.POINTER(ctypes.c_float)),
                ("e", ctypes.POINTER(ctypes.c_float)), ("e_avg", ctypes.POINTER(ctypes.c_float)),
                ("e_s", ctypes.POINTER(ctypes.c_float))]


class Keyframe(ctypes.Structure):
    _fields_ = [("position", ctypes.c_float * 4), ("scale", ctypes.c_float * 4), ("rotation", ctypes.c_float * 4)]


class Procedural(ctypes.Structure):  # include/zygpu.h ZygpuProcedural
    _fields_ = [("kind", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("levels", ctypes.c_uint32), ("checker_scale", ctypes.c_float),
                ("attenuation", ctypes.c_float), ("ratio", ctypes.c_float), ("transition", ctypes.c_float), ("pad", ctypes.c_float),
                ("scale", ctypes.c_float * 4), ("period", ctypes.c_float * 4), ("value", (ctypes.c_float * 4) * 3),
                ("child", Texture * 3), ("child_uniform", (ctypes.c_float * 4) * 3)]


class SceneDesc(ctypes.Structure):
    _fields_ = [("props", ctypes.POINTER(Prop)), ("trafos", ctypes.POINTER(Trafo)), ("aabbs", ctypes.POINTER(Aabb)),
                ("material_ids", ctypes.POINTER(ctypes.c_uint32)), ("light_ids", ctypes.POINTER(ctypes.c_uint32)),
                ("meshes", ctypes.POINTER(Mesh)), ("materials", ctypes.POINTER(Material)),
                ("lights", ctypes.POINTER(Light)), ("light_aabbs", ctypes.POINTER(Aabb)),
                ("light_cones", ctypes.POINTER(ctypes.c_float)), ("solid", PropTree), ("unoccluding", PropTree),
                ("light_tree", LightTree), ("mesh_lights", ctypes.POINTER(MeshLight)), ("image_lights", ctypes.POINTER(ImageLight)),
                ("infinite_props", ctypes.POINTER(ctypes.c_uint32)), ("images", ctypes.POINTER(Image)),
                ("srgb_to_float", ctypes.POINTER(ctypes.c_float)), ("luts", Luts),
                ("specular_threshold", ctypes.c_float),
                ("num_props", ctypes.c_uint32), ("num_parts", ctypes.c_uint32), ("num_meshes", ctypes.c_uint32),
                ("num_materials", ctypes.c_uint32), ("num_lights", ctypes.c_uint32), ("num_mesh_lights", ctypes.c_uint32),
                ("num_images", ctypes.c_uint32), ("num_infinite_props", ctypes.c_uint32), ("num_image_lights", ctypes.c_uint32),
                ("keyframes", ctypes.POINTER(Keyframe)), ("prop_frames", ctypes.POINTER(ctypes.c_uint32)),
                ("num_keyframes", ctypes.c_uint32), ("num_interpolation_frames", ctypes.c_uint32), ("origin", ctypes.c_float * 4),
                ("sun_radiance", ctypes.POINTER(ctypes.c_float)), ("num_sun_radiance", ctypes.c_uint32),
                ("num_portal_lights", ctypes.c_uint32), ("portal_lights", ctypes.POINTER(PortalLight)),
                ("procedurals", ctypes.POINTER(Procedural)), ("num_procedurals", ctypes.c_uint32)]


class ViewDesc(ctypes.Structure):
    _fields_ = [("resolution", ctypes.c_int32 * 2), ("crop", ctypes.c_int32 * 4), ("left_top", ctypes.c_float * 4),
                ("d_x", ctypes.c_float * 4), ("d_y", ctypes.c_float * 4), ("eye_offset", ctypes.c_float * 4),
                ("lens_radius", ctypes.c_float), ("focus_distance", ctypes.c_float), ("camera_trafo", Trafo),
                ("frame_step", ctypes.c_uint64), ("frame_duration", ctypes.c_uint64), ("integrator", ctypes.c_uint32),
                ("max_depth_surface", ctypes.c_uint32), ("max_depth_volume", ctypes.c_uint32),
                ("split_threshold", ctypes.c_float), ("caustics_path", ctypes.c_uint32),
                ("regularize_roughness", ctypes.c_float), ("sampler", ctypes.c_uint32),
                ("num_expected_samples", ctypes.c_uint32), ("filter_radius_int", ctypes.c_int32),
                ("filter_radius", ctypes.c_float), ("filter_samples", ctypes.c_float * (FILTER_TAPS + 1)),
                ("clamp_emission", ctypes.c_float), ("clamp_direct", ctypes.c_float),
                ("clamp_indirect", ctypes.c_float), ("exposure_factor", ctypes.c_float),
                ("alpha_transparency", ctypes.c_uint32), ("tonemapper", ctypes.c_uint32),
                ("aov_slots", ctypes.c_uint32), ("camera_num_keyframes", ctypes.c_uint32), ("camera_origin", ctypes.c_float * 4),
                ("camera_keyframes", Keyframe * MAX_CAMERA_KEYFRAMES), ("aperture_shape", ctypes.c_uint32), ("shutter_configured", ctypes.c_uint32),
                ("shutter_cdf", ctypes.c_float * 64)]


class Stats(ctypes.Structure):
    _fields_ = [("camera_samples", ctypes.c_uint64), ("extension_rays", ctypes.c_uint64),
                ("shadow_rays", ctypes.c_uint64), ("nodes_fetched", ctypes.c_uint64),
                ("triangles_tested", ctypes.c_uint64), ("instances_tested", ctypes.c_uint64)]

    def algorithmic_bytes(self, filter_radius_int=0):
        """B = B_film + sum_ext (224 + B_trav) + sum_shadow (96 + B_trav)."""
        b_trav = 32 * self.nodes_fetched + 48 * self.triangles_tested + 104 * self.instances_tested
        taps = (2 * filter_radius_int + 1) ** 2
        return 32 * taps * self.camera_samples + 224 * self.extension_rays + 96 * self.shadow_rays + b_trav


_LOG_FUNC = ctypes.CFUNCTYPE(None, ctypes.c_uint32, ctypes.c_char_p)
LOG_CALLBACK = _LOG_FUNC  # su_register_log's callback type: (level, text); level 2 = error (core/log.zig)

_lib = None


def load_library(path=None):
    """Loads libzyg_amd.so (built by `python -m zyg_amd.build` / __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ZYG_AMD_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build the HIP library first (python -m zyg_amd.build); "
                           "there is no CPU fallback for the render path")
    lib = ctypes.CDLL(p)
    lib.su_ext_scene_desc.restype = ctypes.POINTER(SceneDesc)
    lib.su_ext_view_desc.restype = ctypes.POINTER(ViewDesc)
    lib.su_ext_context.restype = ctypes.c_void_p
    lib.zygpu_last_error.restype = ctypes.c_char_p
    lib.su_camera_set_fov.argtypes = [ctypes.c_float]
    lib.zygpu_clear_film.argtypes = [ctypes.c_void_p, ctypes.c_float]
    for name in ("zygpu_destroy", "zygpu_synchronize", "zygpu_reset_stats"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.zygpu_render.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                 ctypes.c_int32]
    lib.zygpu_download_film.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.zygpu_upload_film.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.zygpu_set_external_film.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.zygpu_resolve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.zygpu_enable_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zygpu_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    lib.zygpu_get_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    lib.zygpu_measure_copy.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
    lib.zygpu_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
    lib.zygpu_get_kernel_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
    lib.zygpu_kernel_time.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.zygpu_upload_scene.argtypes = [ctypes.c_void_p, ctypes.POINTER(SceneDesc)]
    lib.zygpu_set_view.argtypes = [ctypes.c_void_p, ctypes.POINTER(ViewDesc)]
    lib.zygpu_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    # device groups (include/zygpu.h)
    lib.su_ext_group.restype = ctypes.c_void_p
    lib.su_ext_set_devices.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.zygpu_group_context.restype = ctypes.c_void_p
    lib.zygpu_group_context.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zygpu_group_unique_id.argtypes = [ctypes.c_void_p]
    lib.zygpu_group_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                       ctypes.POINTER(ctypes.c_void_p)]
    for name in ("zygpu_group_destroy", "zygpu_group_size", "zygpu_group_world", "zygpu_group_first_rank", "zygpu_group_uses_rccl", "zygpu_group_reduce",
                 "zygpu_group_dirty", "zygpu_group_synchronize"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.zygpu_group_share.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    lib.zygpu_group_upload_scene.argtypes = [ctypes.c_void_p, ctypes.POINTER(SceneDesc)]
    lib.zygpu_group_set_view.argtypes = [ctypes.c_void_p, ctypes.POINTER(ViewDesc)]
    lib.zygpu_group_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    lib.zygpu_group_clear_film.argtypes = [ctypes.c_void_p, ctypes.c_float]
    lib.zygpu_group_render.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32]
    lib.zygpu_group_render_frame.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32]
    if path is None:
        _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


class Engine:
    """The `su_*` protocol of the reference's clients (capi-test/test.py), one global engine per process."""

    def __init__(self, lib=None):
        self.lib = lib or load_library()
        self._log = None
        if self.lib.su_init() != 0:
            raise RuntimeError("su_init failed (engine already initialised?)")

    def release(self):
        if self.lib is not None:
            if self._log is not None:  # the callback dies with this object: the library must not keep its address
                self.lib.su_register_log.argtypes = [LOG_CALLBACK]
                self.lib.su_register_log(LOG_CALLBACK())
                self._log = None
            self.lib.su_release()
            self.lib = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.release()

    # --- view ---
    def camera(self, width, height, fov_deg=None):
        cam = self.lib.su_perspective_camera_create(width, height)
        if cam < 0:
            raise RuntimeError("su_perspective_camera_create failed")
        if fov_deg is not None:
            self.lib.su_camera_set_fov(ctypes.c_float(np.float32(np.radians(fov_deg))))
        return cam

    def samples_per_pixel(self, n):
        self.lib.su_sampler_create(n)  # returns -1 by design (capi.zig:215-221)

    def integrators(self, json_text):
        if self.lib.su_integrators_create(json_text.encode()) != 0:
            raise RuntimeError("su_integrators_create failed")

    def aovs(self, names):
        """su_aovs_create (capi.zig:202-213): e.g. ["Albedo", "Depth"]."""
        if self.lib.su_aovs_create(json.dumps({n: True for n in names}).encode()) != 0:
            raise RuntimeError("su_aovs_create failed")

    def download_aov(self, aov_class, width, height):
        """The raw {sum w v, sum w} plane of one AOV class, or None when the view does not record it."""
        out = np.zeros((height, width, 4), dtype=np.float32)
        self.lib.zygpu_download_aov.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        rc = self.lib.zygpu_download_aov(self.context(), aov_class, out.ctypes.data)
        if rc == -2:
            return None
        if rc != 0:
            raise RuntimeError("zygpu_download_aov failed")
        return out

    def resolve_aov(self, aov_class, width, height):
        """su_resolve_frame_to_buffer(aov < 9, ...): the resolved AOV, or None for rc -2 (class not recorded)."""
        out = np.zeros((height, width, 4), dtype=np.float32)
        self.lib.su_resolve_frame_to_buffer.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        rc = self.lib.su_resolve_frame_to_buffer(aov_class, width, height, out.ctypes.data)
        if rc == -2:
            return None
        if rc != 0:
            raise RuntimeError("su_resolve_frame_to_buffer failed")
        return out

    def sensor(self, json_text):
        if self.lib.su_ext_sensor_create(json_text.encode()) != 0:
            raise RuntimeError("su_ext_sensor_create failed")

    def camera_parameters(self, json_text):
        """The take's camera "parameters" object (camera_perspective.zig:260-270): fov (degrees), lens / aperture radius,
        focus distance.  The reference's C ABI only has su_camera_set_fov; this is an su_ext_* addition."""
        if self.lib.su_ext_camera_set_parameters(json_text.encode()) != 0:
            raise RuntimeError("su_ext_camera_set_parameters failed")

    def set_option(self, name, value):
        """zygpu_set_option on the engine's context (after start_frame): e.g. "precise", "nee_in_place"."""
        g = self.group()
        if (self.lib.zygpu_group_set_option(g, name.encode(), int(value)) if g else self.lib.zygpu_set_option(self.context(), name.encode(), int(value))) != 0:
            raise RuntimeError("zygpu_set_option failed: " + (self.lib.zygpu_last_error() or b"").decode())

    def log_to(self, stream):
        """su_register_log (capi.zig:128-141): the library's log lines (the reference prints them on stdout) go to `stream` instead,
        e.g. sys.stderr for a program whose stdout is data (bench.py's one JSON line)."""
        def forward(level, text):
            stream.write(("Error: " if level >= 2 else "Warning: " if 1 == level else "") + (text or b"").decode(errors="replace") + "\n")
        self._log = LOG_CALLBACK(forward)  # kept alive with the engine
        self.lib.su_register_log.argtypes = [LOG_CALLBACK]
        self.lib.su_register_log(self._log)

    def get_option(self, name):
        """zygpu_get_option: an option's value, or what the library derived for the uploaded scene ("relaxed_units", "pool_mode",
        "prop_tables", "wide_blas", "mesh_kernels_active")."""
        v = ctypes.c_int(0)
        if self.lib.zygpu_get_option(self.context(), name.encode(), ctypes.byref(v)) != 0:
            raise RuntimeError("zygpu_get_option failed: " + (self.lib.zygpu_last_error() or b"").decode())
        return v.value

    # --- resources / scene ---
    def material(self, json_text, id=NULL):
        m = self.lib.su_material_create(id, json_text.encode())
        if m < 0:
            raise RuntimeError("su_material_create failed: " + json_text)
        return m

    def image(self, pixels):
        """su_image_create (capi.zig:236-298) for an (H, W, 3) array: uint8 (read as sRGB) or float32 (taken as is); an
        (H, W) or (H, W, 1) array makes a one-channel image (masks, weights: uint8 = unorm), an (H, W, 2) array a two-channel
        one (normal maps: uint8 = snorm)."""
        a = np.ascontiguousarray(pixels)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.ndim != 3 or a.shape[2] not in (1, 2, 3) or a.dtype not in (np.uint8, np.float32):
            raise ValueError("image: (H, W, 3), (H, W, 2) or (H, W) uint8 or float32")
        fmt, bpc = (0, 1) if a.dtype == np.uint8 else (4, 4)  # capi.zig:31-37 Format
        self.lib.su_image_create.argtypes = [ctypes.c_uint32] * 7 + [ctypes.c_void_p]
        i = self.lib.su_image_create(NULL, fmt, a.shape[2], a.shape[1], a.shape[0], 1, a.shape[2] * bpc, a.ctypes.data)
        if i < 0:
            raise RuntimeError("su_image_create failed")
        return i

    def triangle_mesh(self, indices, positions, normals, uvs=None, parts=None):
        indices = _u32(indices).reshape(-1)
        positions = _f32(positions).reshape(-1, 3)
        normals = _f32(normals).reshape(-1, 3)
        num_parts = 0
        parts_p = None
        if parts is not None:
            parts = _u32(parts).reshape(-1, 3)
            num_parts = parts.shape[0]
            parts_p = parts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        uvs_p = None
        if uvs is not None:
            uvs = _f32(uvs).reshape(-1, 2)
            uvs_p = uvs.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        fp = ctypes.POINTER(ctypes.c_float)
        m = self.lib.su_triangle_mesh_create(
            NULL, num_parts, parts_p, indices.size // 3, indices.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
            positions.shape[0], positions.ctypes.data_as(fp), 3, normals.ctypes.data_as(fp), 3, None, 0, uvs_p, 2,
            False)
        if m < 0:
            raise RuntimeError("su_triangle_mesh_create failed")
        return m

    def prop(self, shape, materials, unoccluding=False):
        mats = _u32(materials)
        p = self.lib.su_ext_prop_create(shape, mats.size, mats.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                        1 if unoccluding else 0)
        if p < 0:
            raise RuntimeError("su_prop_create failed")
        return p

    def light(self, prop):
        if self.lib.su_light_create(prop) != 0:
            raise RuntimeError("su_light_create failed")

    def transformation(self, prop, m16):
        m = _f32(m16).reshape(16)
        if self.lib.su_prop_set_transformation(prop, m.ctypes.data_as(ctypes.POINTER(ctypes.c_float))) != 0:
            raise RuntimeError("su_prop_set_transformation failed")

    def sky(self, parameters=None, sun_table=None, sky_image=None):
        """The reference's Sky entity (su_ext_sky_create): a Distant sun + a Canopy sky; parameters as the scene file's
        {"sun": {"rotation": [x, y, z]}, "turbidity": t, ...}.  sun_table (1024 x 3) and sky_image (h x w x 3), both float32,
        are the sky bake's products (su_ext_sky_tables); without them compile() generates a stand-in.  Returns the sun's prop."""
        self.lib.su_ext_sky_create.argtypes = [ctypes.c_char_p]
        sun = self.lib.su_ext_sky_create(json.dumps(parameters).encode() if parameters is not None else None)
        if sun < 0:
            raise RuntimeError("su_ext_sky_create failed")
        if sun_table is not None:
            sun_table = np.ascontiguousarray(sun_table, dtype=np.float32)
            sky_image = np.ascontiguousarray(sky_image, dtype=np.float32)
            assert sun_table.shape == (1024, 3) and 3 == sky_image.ndim and 3 == sky_image.shape[2]
            self.lib.su_ext_sky_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
            if 0 != self.lib.su_ext_sky_tables(sun_table.ctypes.data, sky_image.ctypes.data, sky_image.shape[1], sky_image.shape[0]):
                raise RuntimeError("su_ext_sky_tables failed")
        return sun

    def portal(self, light_id, shape=5):
        """A Portal (su_ext_portal_create): a Rectangle prop with the material of light `light_id` (a Dome with an emission map),
        which becomes a hidden prototype; place it with transformation(), then portal_light(prop, light_id)."""
        self.lib.su_ext_portal_create.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        p = self.lib.su_ext_portal_create(shape, light_id)
        if p < 0:
            raise RuntimeError("su_ext_portal_create failed")
        return p

    def portal_light(self, prop, light_id):
        self.lib.su_ext_light_create_linked.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        if self.lib.su_ext_light_create_linked(prop, light_id) != 0:
            raise RuntimeError("su_ext_light_create_linked failed")

    def shadow_catcher(self, prop):
        """scene.propSetShadowCatcher (su_ext_prop_set_shadow_catcher): PathtracerMIS sees the backplate through the prop, darkened
        where the shadow-catcher lights are occluded."""
        self.lib.su_ext_prop_set_shadow_catcher.argtypes = [ctypes.c_uint32]
        if self.lib.su_ext_prop_set_shadow_catcher(prop) != 0:
            raise RuntimeError("su_ext_prop_set_shadow_catcher failed")

    def shadow_catcher_light(self, prop, in_camera=True, in_reflection=True):
        """The prop's visibility with shadow_catcher_light = true (scene_loader.zig:612-619); call before light(prop)."""
        self.lib.su_ext_prop_set_visibility.argtypes = [ctypes.c_uint32] + [ctypes.c_bool] * 4
        if self.lib.su_ext_prop_set_visibility(prop, in_camera, in_reflection, False, True) != 0:
            raise RuntimeError("su_ext_prop_set_visibility failed")

    def visibility(self, prop, in_camera=True, in_reflection=True, in_sss=False):
        self.lib.su_prop_set_visibility(prop, int(in_camera), int(in_reflection), int(in_sss))

    # --- compile / render ---
    def compile(self, frame=0):
        """Scene.compile + camera.update only; returns (SceneDesc, ViewDesc) views valid until the next compile."""
        if self.lib.su_ext_compile(frame) != 0:
            raise RuntimeError("su_ext_compile failed (feature outside the hot-path scope?)")
        return self.lib.su_ext_scene_desc().contents, self.lib.su_ext_view_desc().contents

    def set_device(self, device_id):
        self.lib.su_ext_set_device(device_id)

    def set_devices(self, device_ids, shard=2, flags=0):
        """Several devices behind the same calls (su_ext_set_devices: a ZygpuGroup; shard 0 iterations, 1 rows, 2 the library's choice)."""
        ids = (ctypes.c_int32 * len(device_ids))(*device_ids)
        if self.lib.su_ext_set_devices(ids, len(device_ids), shard, flags) != 0:
            raise RuntimeError("su_ext_set_devices failed (after the first frame?)")

    def set_world(self, unique_id, first_rank, world):
        """The devices of set_devices as ranks of a world across processes (su_ext_set_world; unique_id: bytes of unique_id())."""
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        if self.lib.su_ext_set_world(buf, first_rank, world) != 0:
            raise RuntimeError("su_ext_set_world failed")

    def unique_id(self):
        buf = (ctypes.c_uint8 * 128)()
        if self.lib.zygpu_group_unique_id(buf) != 0:
            raise RuntimeError("zygpu_group_unique_id failed: " + (self.lib.zygpu_last_error() or b"").decode())
        return bytes(buf)

    def group(self):
        g = self.lib.su_ext_group()
        return ctypes.c_void_p(g) if g else None

    def start_frame(self, frame=0):
        if self.lib.su_start_frame(frame) != 0:
            raise RuntimeError("su_start_frame failed: " + (self.lib.zygpu_last_error() or b"").decode())

    def render_iterations(self, n):
        if self.lib.su_render_iterations(n) != 0:
            raise RuntimeError("su_render_iterations failed: " + (self.lib.zygpu_last_error() or b"").decode())

    def render_rows(self, n, y_begin, y_end):
        if self.lib.su_ext_render_rows(n, y_begin, y_end) != 0:
            raise RuntimeError("su_ext_render_rows failed: " + (self.lib.zygpu_last_error() or b"").decode())

    def render_frame(self, frame=0):
        if self.lib.su_render_frame(frame) != 0:
            raise RuntimeError("su_render_frame failed: " + (self.lib.zygpu_last_error() or b"").decode())

    def context(self):
        return ctypes.c_void_p(self.lib.su_ext_context())

    def synchronize(self):
        """Waits for the device and raises on a device error flag (e.g. more crossed unoccluding emitters than slots)."""
        g = self.group()
        rc = self.lib.zygpu_group_synchronize(g) if g else self.lib.zygpu_synchronize(self.context())
        if rc != 0:
            raise RuntimeError(f"zygpu_synchronize failed ({rc}): " + (self.lib.zygpu_last_error() or b"").decode())

    def resolve(self, width, height):
        out = np.zeros((height, width, 4), dtype=np.float32)
        rc = self.lib.su_resolve_frame_to_buffer(NULL, width, height, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        if rc != 0:
            raise RuntimeError("su_resolve_frame_to_buffer failed")
        return out

    def download_film(self, width, height):
        out = np.zeros((height, width, 4), dtype=np.float32)
        g = self.group()
        if g and 1 == self.lib.zygpu_group_dirty(g) and 0 != self.lib.zygpu_group_reduce(g):  # the devices' films summed onto the first one's
            raise RuntimeError("zygpu_group_reduce failed: " + (self.lib.zygpu_last_error() or b"").decode())
        if self.lib.zygpu_download_film(self.context(), out.ctypes.data) != 0:
            raise RuntimeError("zygpu_download_film failed")
        return out

    def stats(self):
        s = Stats()
        self.lib.zygpu_get_stats(self.context(), ctypes.byref(s))
        return s
