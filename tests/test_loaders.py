"""Take / scene / mesh file loaders (reference cli/take_loader.zig, util/scene_loader.zig, shape_provider.zig): a scene
loaded from disk compiles to the same arrays as the scene built through the su_* calls."""
import ctypes
import json
import os

import numpy as np
import pytest

from util import write_png as _png

import zyg_amd
from zyg_amd import scenes


def _compiled(e):
    sd, vd = e.compile()
    props = [(sd.props[i].shape, sd.props[i].flags) for i in range(sd.num_props)]
    trafos = np.array([[list(sd.trafos[i].r[k]) for k in range(3)] + [list(sd.trafos[i].position)[:4]] for i in range(sd.num_props)],
                      dtype=np.float32)
    mats = [(m.kind, tuple(m.color_a), m.roughness, tuple(m.emission)) for m in (sd.materials[sd.material_ids[i]] for i in range(sd.num_parts))]
    meshes = [(sd.meshes[i].num_triangles, sd.meshes[i].num_vertices, sd.meshes[i].num_nodes) for i in range(sd.num_meshes)]
    nodes = [bytes(ctypes.string_at(sd.meshes[i].nodes, 32 * sd.meshes[i].num_nodes)) for i in range(sd.num_meshes)]
    view = (tuple(vd.resolution), tuple(vd.left_top), tuple(vd.d_x), vd.integrator, vd.max_depth_surface, vd.num_expected_samples,
            vd.filter_radius_int, vd.tonemapper)
    return props, trafos, mats, meshes, nodes, view, sd.num_lights


def test_take_file_equals_api_scene(product_lib, tmp_path):
    w, h, spp = 48, 32, 4
    take = scenes.write_cornell_take(str(tmp_path), w, h, spp)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(take.encode()) == 0
        loaded = _compiled(e)
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, w, h, spp, material="substitute")
        e.sensor(json.dumps({}))
        built = _compiled(e)
    assert loaded[0] == built[0] and loaded[2] == built[2] and loaded[3] == built[3] and loaded[6] == built[6] == 2
    assert np.array_equal(loaded[1], built[1])
    assert loaded[4] == built[4]  # the BLAS of the JSON mesh and of the .sub mesh, byte for byte
    assert loaded[5] == built[5]


def test_transformations_children_and_files(product_lib, tmp_path):
    """Euler rotations (Rz Rx Ry, degrees), look_at, parent-child composition (transformation.zig:30-36), nested scene
    files, material files, visibility, UInt32 and delta-encoded .sub indices."""
    idx, pos, nrm, uv = scenes.box_mesh((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    scenes.write_sub_mesh(str(tmp_path / "cube_delta.sub"), idx, pos, nrm, uv, delta=True)
    scenes.write_sub_mesh(str(tmp_path / "cube_plain.sub"), idx, pos, nrm, uv, delta=False)
    os.mkdir(tmp_path / "sub")
    with open(tmp_path / "sub" / "gold.material", "w") as f:
        json.dump({"rendering": {"Substitute": {"metal_preset": "Gold", "roughness": 0.3}}}, f)
    with open(tmp_path / "sub" / "inner.scene", "w") as f:
        json.dump({"entities": [{"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["gold.material"],
                                 "transformation": {"position": [0, 2, 0], "scale": [0.5, 0.5, 0.5]}}]}, f)
    scene = {"materials": [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}},
                           {"name": "lamp", "rendering": {"Light": {"emittance": {"spectrum": [5, 5, 5]}}}}],
             "entities": [
                 {"type": "Dummy", "transformation": {"position": [1, 0, 0], "rotation": [0, 90, 0], "scale": [2, 2, 2]},
                  "entities": [{"type": "Prop", "shape": {"file": "cube_delta.sub"}, "materials": ["grey"],
                                "transformation": {"position": [0, 0, 1]}, "visibility": {"in_camera": False}},
                               {"file": "sub/inner.scene"}]},
                 {"type": "Prop", "shape": {"file": "cube_plain.sub"}, "materials": ["grey"],
                  "transformation": {"position": [-2, 0, 0], "rotation": [30, 0, 45]}},
                 {"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"],
                  "transformation": {"position": [0, 3, 0], "look_at": [0, 0, 0], "up": [0, 0, 1]}}]}
    with open(tmp_path / "t.scene", "w") as f:
        json.dump(scene, f)
    take = {"scene": "t.scene", "camera": {"Perspective": {"parameters": {"fov": 60.0}, "resolution": [32, 16], "crop": [2, 1, 30, 15],
                                                           "transformation": {"position": [0, 1, -6]}}},
            "sampler": {"Random": {"samples_per_pixel": 9}}, "integrator": {"surface": {"PTDL": {"depth": {"surface": 5}}}},
            "sensor": {"filter": {"Mitchell": {}}}, "post": [{"tonemapper": {"ACES": {"exposure": 1.0}}}],
            "export": {"Image": {"format": "EXR", "bitdepth": 32}}}
    with open(tmp_path / "t.take", "w") as f:
        json.dump(take, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "t.take").encode()) == 0
        sd, vd = e.compile()
        assert (vd.integrator, vd.max_depth_surface, vd.sampler, vd.num_expected_samples) == (2, 5, 0, 9)
        assert tuple(vd.crop) == (2, 1, 30, 15) and vd.filter_radius_int == 2 and vd.tonemapper == 1 and abs(vd.exposure_factor - 2.0) < 1e-6
        cam = np.array(list(vd.camera_trafo.position))
        shapes = [sd.props[i].shape for i in range(sd.num_props)]
        assert shapes.count(scenes.SPHERE) == 1 and shapes.count(7) == 2 and shapes.count(scenes.RECTANGLE) == 1 and sd.num_lights == 1
        assert sd.meshes[0].num_triangles == sd.meshes[1].num_triangles == 12
        assert bytes(ctypes.string_at(sd.meshes[0].nodes, 32 * sd.meshes[0].num_nodes)) == bytes(ctypes.string_at(sd.meshes[1].nodes, 32 * sd.meshes[1].num_nodes))

        def world(i):  # camera-relative compiled trafo -> rows (scaled basis), position
            t = sd.trafos[i]
            rows = np.array([list(t.r[k])[:3] for k in range(3)]) * np.array([t.r[k][3] for k in range(3)])[:, None]
            return rows, np.array(list(t.position)[:3]) + np.array([0, 1, -6])

        by_shape = {}
        for i in range(sd.num_props):
            by_shape.setdefault(sd.props[i].shape, []).append(i)
        # parent: rotation about y by 90 degrees, scale 2, position (1,0,0); child at local (0,0,1)
        ry = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=np.float64)  # rows of Mat3x3.initRotationY(90 deg)
        child = [i for i in by_shape[7] if not (sd.props[i].flags & 1)][0]  # the one hidden from the camera
        rows, p = world(child)
        np.testing.assert_allclose(rows, 2.0 * ry, atol=1e-6)
        np.testing.assert_allclose(p, np.array([1, 0, 0]) + 2.0 * (np.array([0, 0, 1]) @ ry), atol=1e-6)
        rows, p = world(by_shape[scenes.SPHERE][0])  # nested file under the same parent
        np.testing.assert_allclose(p, [1, 4, 0], atol=1e-6)
        np.testing.assert_allclose(rows, 1.0 * ry, atol=1e-6)
        other = [i for i in by_shape[7] if i != child][0]
        rows, p = world(other)
        a, c = np.radians(30.0), np.radians(45.0)
        rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        np.testing.assert_allclose(rows, rz @ rx, atol=1e-6)
        rows, p = world(by_shape[scenes.RECTANGLE][0])  # look_at: z axis towards the target, y = up
        np.testing.assert_allclose(rows[2], [0, -1, 0], atol=1e-6)
        np.testing.assert_allclose(rows[1], [0, 0, 1], atol=1e-6)
        gold = sd.materials[sd.material_ids[sd.props[by_shape[scenes.SPHERE][0]].parts_start]]
        assert gold.metallic == 1.0 and abs(gold.roughness - 0.3) < 1e-7
        assert np.allclose(cam, 0.0)  # camera-relative scene


def test_missing_files_fail_loudly(product_lib, tmp_path):
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "nope.take").encode()) == -1
    with open(tmp_path / "a.take", "w") as f:
        json.dump({"scene": "missing.scene", "camera": {"Perspective": {"resolution": [8, 8]}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "a.take").encode()) == -1


_REFUSED = {
    "animated \"Scatterer\"": {"type": "Scatterer", "animation": {"keyframes": [{"time": 0, "transformation": {"position": [0, 0, 0]}}]}, "prototypes": []},
    "animated \"Instancer\"": {"type": "Instancer", "animation": {"keyframes": [{"time": 0, "transformation": {"position": [0, 0, 0]}}]},
                              "prototypes": [{"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["grey"]}],
                              "instances": {"prototypes": [0], "transformations": [{}]}},
    "Instancer prototype": {"type": "Instancer", "prototypes": [{"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"]}],
                            "instances": {"prototypes": [0], "transformations": [{}]}},
}


@pytest.mark.parametrize("what", sorted(_REFUSED))
def test_unrendered_entities_and_keys_fail_the_load(product_lib, tmp_path, what):
    """What the reference's scene loader reads (scene_loader.zig:241-326, 612-628) and this build does not render is refused by
    name, logged at error level: a take is never rendered with a part of its scene dropped in silence.  Nested below a node of an
    unknown type, which the reference treats as a plain transformation node (its children are loaded)."""
    scene = {"materials": [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}},
                           {"name": "lamp", "rendering": {"Light": {"emittance": {"spectrum": [5, 5, 5]}}}}],
             "entities": [{"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"], "transformation": {"position": [0, 3, 0]}},
                          {"type": "Group", "transformation": {"position": [1, 0, 0]}, "entities": [_REFUSED[what]]}]}
    with open(tmp_path / "s.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / "s.take", "w") as f:
        json.dump({"scene": "s.scene", "camera": {"Perspective": {"resolution": [8, 8]}}}, f)
    messages = []
    callback = zyg_amd.LOG_CALLBACK(lambda level, text: messages.append((level, text.decode())))
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_register_log.argtypes = [zyg_amd.LOG_CALLBACK]
        assert e.lib.su_register_log(callback) == 0
        try:
            assert e.lib.su_ext_load_take(str(tmp_path / "s.take").encode()) == -1
        finally:
            e.lib.su_register_log(zyg_amd.LOG_CALLBACK())
    assert any(level == 2 and what in text and "does not render" in text for level, text in messages), messages


def _write_take(tmp_path, scene, name="s"):
    with open(tmp_path / f"{name}.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / f"{name}.take", "w") as f:
        json.dump({"scene": f"{name}.scene", "camera": {"Perspective": {"resolution": [16, 12]}, "transformation": {"position": [0, 1, -6]}}}, f)
    return str(tmp_path / f"{name}.take").encode()


_MATERIALS = [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}},
              {"name": "red", "rendering": {"Substitute": {"color": [0.7, 0.1, 0.1]}}},
              {"name": "lamp", "rendering": {"Light": {"emittance": {"spectrum": [5, 5, 5]}}}}]
_LAMP = {"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"], "transformation": {"position": [0, 3, 0], "rotation": [90, 0, 0]}}


def test_instancer_is_flattened_into_props(product_lib, tmp_path):
    """An Instancer (scene_loader.zig:401-560, prop/instancer.zig): prototypes that are not rendered themselves, instances = (prototype,
    transformation in the instancer's space).  The loader makes every instance a prop with the transformation instancer o instance and
    the prototype's shape, materials and visibility: the scene compiles to the same arrays as one that names those props directly; an
    index past the prototypes falls back to prototype 0 (scene_loader.zig:454-457); `source` reads the same members from a file."""
    idx, pos, nrm, uv = scenes.box_mesh((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    scenes.write_sub_mesh(str(tmp_path / "cube.sub"), idx, pos, nrm, uv, delta=True)
    instancer_t = {"position": [0.5, 0, 1], "rotation": [0, 30, 0], "scale": [2, 2, 2]}
    instances = [{"position": [-1, 0.5, 0]}, {"position": [1, 0.5, 0], "rotation": [0, 45, 0], "scale": [0.5, 1, 0.5]}, {"position": [0, 2, 0]}]
    members = {"prototypes": [{"type": "Prop", "shape": {"file": "cube.sub"}, "materials": ["grey"], "transformation": {"position": [9, 9, 9]}},
                              {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["red"], "visibility": {"in_camera": False}}],
               "instances": {"prototypes": [0, 1, 7], "transformations": instances}}
    inline = {"materials": _MATERIALS, "entities": [_LAMP, dict({"type": "Instancer", "transformation": instancer_t}, **members)]}
    with open(tmp_path / "forest.instancer", "w") as f:
        json.dump(dict({"materials": _MATERIALS[:2]}, **members), f)
    from_file = {"materials": _MATERIALS, "entities": [_LAMP, {"type": "Instancer", "transformation": instancer_t, "source": "forest.instancer"}]}
    kinds = [({"shape": {"file": "cube.sub"}, "materials": ["grey"]}), ({"shape": {"type": "Sphere"}, "materials": ["red"], "visibility": {"in_camera": False}}),
             ({"shape": {"file": "cube.sub"}, "materials": ["grey"]})]
    flat = {"materials": _MATERIALS,
            "entities": [_LAMP, {"type": "Group", "transformation": instancer_t,
                                 "entities": [dict({"type": "Prop", "transformation": t}, **k) for k, t in zip(kinds, instances)]}]}
    compiled = {}
    for name, scene in (("inline", inline), ("file", from_file), ("flat", flat)):
        with zyg_amd.Engine(product_lib) as e:
            assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, name)) == 0, name
            compiled[name] = _compiled(e)
    for name in ("inline", "file"):
        a, b = compiled[name], compiled["flat"]
        assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and a[6] == b[6], name
        assert np.array_equal(a[1], b[1]), name
    assert 5 == len(compiled["flat"][0])  # camera, lamp, three instances: no prop for a prototype


def test_scatterer_instances_are_props_of_the_scene(product_lib, tmp_path):
    """A Scatterer (scene_loader.zig:225-227, 260-262, 561-610): a node without an entity of its own whose children are loaded like any
    node's; then its prototypes (not rendered themselves) are instantiated: instance i = a prop with prototype[indices[i]]'s shape and
    materials and the transformation scatterer o transformations[i], created behind the node's children.  The scene compiles to the
    arrays of one that names those props directly below a Group with the scatterer's transformation."""
    scatter_t = {"position": [-0.5, 0, 2], "rotation": [0, -20, 0], "scale": [1.5, 1.5, 1.5]}
    instances = [{"position": [-1, 0.5, 0]}, {"position": [1, 0.5, 0], "rotation": [0, 45, 0], "scale": [0.5, 1, 0.5]}, {"position": [0, 2, 0]}]
    child = {"type": "Prop", "shape": {"type": "Disk"}, "materials": ["red"], "transformation": {"position": [0, 0.1, 0], "rotation": [90, 0, 0]}}
    scatterer = {"type": "Scatterer", "transformation": scatter_t, "entities": [child],
                 "prototypes": [{"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["grey"], "transformation": {"position": [9, 9, 9]}},
                                {"type": "Prop", "shape": {"type": "Cube"}, "materials": ["red"]}],
                 "instances": {"prototypes": [0, 1, 5], "transformations": instances}}
    kinds = [{"shape": {"type": "Sphere"}, "materials": ["grey"]}, {"shape": {"type": "Cube"}, "materials": ["red"]}, {"shape": {"type": "Sphere"}, "materials": ["grey"]}]
    flat = {"materials": _MATERIALS,
            "entities": [_LAMP, {"type": "Group", "transformation": scatter_t,
                                 "entities": [child] + [dict({"type": "Prop", "transformation": t}, **k) for k, t in zip(kinds, instances)]}]}
    compiled = {}
    for name, scene in (("scatterer", {"materials": _MATERIALS, "entities": [_LAMP, scatterer]}), ("flat", flat)):
        with zyg_amd.Engine(product_lib) as e:
            assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, name)) == 0, name
            compiled[name] = _compiled(e)
    a, b = compiled["scatterer"], compiled["flat"]
    assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and a[6] == b[6]
    assert np.array_equal(a[1], b[1])
    assert 6 == len(b[0])  # camera, lamp, the child, three instances: no prop for a prototype


def test_animation_keyframes_become_interpolation_frames(product_lib, tmp_path):
    """An entity's "animation" (animation_loader.zig, animation.zig): keyframes with times in seconds; every compile resamples them
    at the frame's two ticks (1/60 s apart: Graph.simulate, scene_graph.zig:106-118): position on the centripetal Catmull-Rom
    spline through the neighbouring keyframes (ends extrapolated), scale lerped, rotation slerped; the static parent applies to the
    keyframes.  Here: four keyframes 1/30 s apart on a straight line at constant speed, on which the spline is the straight line, so
    frame f's two interpolation frames are known in closed form; the "transformations" / "times" form reads the same."""
    keys = [{"time": k / 30.0, "transformation": {"position": [0.3 * k, 1.0, 0.0], "scale": [1.0 + 0.5 * k] * 3}} for k in range(4)]
    forms = {"keyframes": {"keyframes": keys},
             "arrays": {"transformations": [k["transformation"] for k in keys], "times": [k["time"] for k in keys]},
             "fps": {"frames_per_second": 30, "keyframes": [{"transformation": k["transformation"]} for k in keys]}}
    for name, animation in forms.items():
        scene = {"materials": _MATERIALS,
                 "entities": [_LAMP, {"type": "Group", "transformation": {"position": [10, 0, 0]},
                                      "entities": [{"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["grey"], "animation": animation}]}]}
        with zyg_amd.Engine(product_lib) as e:
            assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, name)) == 0
            for frame in (0, 1, 3, 5):
                assert e.lib.su_ext_compile(frame) == 0
                sd = e.lib.su_ext_scene_desc().contents
                sphere = [p for p in range(sd.num_props) if sd.props[p].shape == scenes.SPHERE]
                assert 1 == len(sphere) and 2 == sd.num_interpolation_frames
                first = sd.prop_frames[sphere[0]]
                assert first != 0xFFFFFFFF, "the animated prop has no keyframes"
                for i in range(2):
                    k = sd.keyframes[first + i]
                    t = (frame + i) / 60.0  # seconds
                    s = min(t * 30.0, 3.0)  # keyframe units; past the last keyframe the last one holds
                    if t * 30.0 >= 3.0:
                        s = 3.0
                    np.testing.assert_allclose(list(k.position)[:3], [10.0 + 0.3 * s, 1.0, 0.0], rtol=0, atol=2e-5, err_msg=f"{name} frame {frame}.{i}")
                    np.testing.assert_allclose(list(k.scale)[:3], [1.0 + 0.5 * s] * 3, rtol=0, atol=2e-5, err_msg=f"{name} frame {frame}.{i}")


def test_animated_parents_move_their_children(product_lib, tmp_path):
    """scene_graph.zig:178-226, 250-310: everything below an animated node inherits its frames: a child's world transformation at
    tick i is the parent's frame i transforming the child's own (static, or its own resampled keyframes, which then do not take the
    parent in).  A Group translating along x at constant speed carries a static Sphere and a Disk that itself moves up; a Group
    turning about y (slerp: the angle is linear in time) carries a Sphere one unit off its axis."""
    line = [{"time": k / 30.0, "transformation": {"position": [0.6 * k, 0.0, 0.0]}} for k in range(4)]
    rise = [{"time": k / 30.0, "transformation": {"position": [0.0, 0.3 * k, 2.0]}} for k in range(4)]
    turn = [{"time": k / 30.0, "transformation": {"rotation": [0, 20.0 * k, 0]}} for k in range(4)]
    scene = {"materials": _MATERIALS,
             "entities": [_LAMP,
                          {"type": "Group", "animation": {"keyframes": line}, "entities": [
                              {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["grey"], "transformation": {"position": [0, 1, 0]}},
                              {"type": "Prop", "shape": {"type": "Disk"}, "materials": ["red"], "animation": {"keyframes": rise}}]},
                          {"type": "Group", "transformation": {"position": [0, 0, 5]}, "entities": [
                              {"type": "Group", "animation": {"keyframes": turn}, "entities": [
                                  {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["red"],
                                   "transformation": {"position": [1, 0, 0], "scale": [0.5, 0.5, 0.5]}}]}]}]}
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, "graph")) == 0
        for frame in (0, 1, 4):
            assert e.lib.su_ext_compile(frame) == 0
            sd = e.lib.su_ext_scene_desc().contents
            spheres = [p for p in range(sd.num_props) if sd.props[p].shape == scenes.SPHERE]
            disks = [p for p in range(sd.num_props) if sd.props[p].shape == 2]  # ZYGPU_SHAPE_DISK
            assert 2 == len(spheres) and 1 == len(disks)
            for i in range(2):
                s = min((frame + i) / 2.0, 3.0)  # keyframe units (two ticks per keyframe interval)
                carried, risen, turned = (sd.keyframes[sd.prop_frames[p] + i] for p in (spheres[0], disks[0], spheres[1]))
                np.testing.assert_allclose(list(carried.position)[:3], [0.6 * s, 1.0, 0.0], atol=2e-5)
                np.testing.assert_allclose(list(risen.position)[:3], [0.6 * s, 0.3 * s, 2.0], atol=2e-5)
                angle = np.radians(20.0 * s)
                # base/json.zig's rotation about y by a turns +x towards -z for a positive angle?  Only the radius, the height and the
                # angle between consecutive ticks are asserted, not the sense of rotation
                p = np.array(list(turned.position)[:3]) - np.array([0.0, 0.0, 5.0])
                np.testing.assert_allclose([np.hypot(p[0], p[2]), p[1]], [1.0, 0.0], atol=2e-5)
                np.testing.assert_allclose(abs(np.arctan2(p[2], p[0])), angle, atol=2e-5)
                np.testing.assert_allclose(list(turned.scale)[:3], [0.5, 0.5, 0.5], atol=1e-6)


def test_camera_animation_in_the_take(product_lib, tmp_path):
    """take_loader.zig:142: a camera's "animation" makes it an animated entity like any other; the view carries its two interpolation
    frames for the frame being compiled."""
    keys = [{"time": k / 30.0, "transformation": {"position": [0.4 * k, 1.0, -6.0]}} for k in range(4)]
    with open(tmp_path / "c.scene", "w") as f:
        json.dump({"materials": _MATERIALS, "entities": [_LAMP]}, f)
    with open(tmp_path / "c.take", "w") as f:
        json.dump({"scene": "c.scene", "camera": {"Perspective": {"resolution": [16, 12], "transformation": {"position": [0, 1, -6]},
                                                                 "animation": {"keyframes": keys}}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "c.take").encode()) == 0
        for frame in (0, 3):
            assert e.lib.su_ext_compile(frame) == 0
            vd = e.lib.su_ext_view_desc().contents
            assert 2 == vd.camera_num_keyframes
            for i in range(2):
                s = (frame + i) / 2.0
                np.testing.assert_allclose(list(vd.camera_keyframes[i].position)[:3], [0.4 * s, 1.0, -6.0], atol=2e-5)


def _shutter_cdf(open_, close, slope):
    """Base.setShutter + Distribution1DN(63).configure (camera_base.zig:43-118, distribution_1d.zig:167-186) in numpy float32: the
    test's own statement of them."""
    f32 = np.float32

    def lerp(a, b, t):  # util.zig:3-8: fma(1 - t, a, t * b); the product in double and rounded once is that fma
        return f32(np.float64(f32(1) - t) * np.float64(a) + np.float64(f32(t * b)))

    def bezier(cp, u):
        l = [tuple(lerp(cp[i][k], cp[i + 1][k], u) for k in range(2)) for i in range(3)]
        m = [tuple(lerp(l[i][k], l[i + 1][k], u) for k in range(2)) for i in range(2)]
        return tuple(lerp(m[0][k], m[1][k], u) for k in range(2))

    def search(cp, x):
        u, step, c = f32(0.5), f32(0.25), (f32(0), f32(0))
        for _ in range(16):
            c = bezier(cp, u)
            if x < c[0]:
                u = f32(u - step)
            elif x > c[0]:
                u = f32(u + step)
            else:
                break
            step = f32(step * f32(0.5))
        return c[1]

    open_, close, slope = f32(open_), f32(close), [f32(v) for v in slope]
    function = []
    for i in range(63):
        t = f32(f32(f32(i) + f32(0.5)) / f32(63))
        if t < open_:
            f = search([(f32(0), f32(0)), (slope[0], slope[1]), (slope[2], slope[3]), (open_, f32(1))], t) if len(slope) >= 4 \
                else lerp(f32(0), f32(1), f32(t / open_))
        elif t > close:
            f = search([(close, f32(1)), (slope[4], slope[5]), (slope[6], slope[7]), (f32(1), f32(0))], t) if len(slope) >= 8 \
                else lerp(f32(1), f32(0), f32(f32(t - close) / f32(f32(1) - close)))
        else:
            f = f32(1)
        function.append(f)
    integral = f32(0)
    for f in function:
        integral = f32(integral + f)
    ii = f32(f32(1) / integral)
    cdf, p = [f32(0)], f32(0)
    for f in function[:-1]:
        p = f32(np.float64(f) * np.float64(ii) + np.float64(p))
        cdf.append(p)
    cdf.append(f32(1))
    return np.asarray(cdf, dtype=np.float32)


def test_take_aov_and_shutter_keys(product_lib, tmp_path):
    """The take's "aov" (take.zig:106-129) reaches the view's AOV slots.  A camera "shutter" (take_loader.zig:153-191 loadShutter ->
    camera_base.zig:43-75 setShutter) sets the frame's step and duration (frames_per_second / frame_step, motion_blur) and configures
    the distribution a camera sample's time goes through, flat when only the key is there; the scene's interpolation frames follow the
    frame's length (scene.zig:254-257, 724-734).  A frame longer than the view block's keyframes is refused by name."""
    tick = 705600000 // 60

    def load(shutter, aov=None):
        lamp = dict(_LAMP, animation={"keyframes": [{"time": 0, "transformation": {"position": [0, 3, 0]}},
                                                    {"time": 1, "transformation": {"position": [1, 3, 0]}}]})
        with open(tmp_path / "c.scene", "w") as f:
            json.dump({"materials": _MATERIALS, "entities": [lamp]}, f)
        camera = {"resolution": [16, 12]}
        if shutter is not None:
            camera["shutter"] = shutter
        take = {"scene": "c.scene", "camera": {"Perspective": camera}}
        if aov:
            take["aov"] = aov
        with open(tmp_path / "c.take", "w") as f:
            json.dump(take, f)
        with zyg_amd.Engine(product_lib) as e:
            rc = e.lib.su_ext_load_take(str(tmp_path / "c.take").encode())
            if 0 != rc:
                return rc, None, None
            assert e.lib.su_ext_compile(0) == 0
            vd = e.lib.su_ext_view_desc().contents
            sd = e.lib.su_ext_scene_desc().contents
            return rc, sd.num_interpolation_frames, (vd.aov_slots, vd.frame_step, vd.frame_duration, vd.shutter_configured,
                                                     np.array(list(vd.shutter_cdf), dtype=np.float32))

    rc, nif, (slots, step, duration, configured, cdf) = load(None, {"Depth": True, "MaterialId": True, "Albedo": False})
    assert (0, 2, 0b110, tick, tick, 0) == (rc, nif, slots, step, duration, configured)
    rc, nif, (_, step, duration, configured, cdf) = load({"frames_per_second": 60, "motion_blur": True})
    assert (0, 2, tick, tick, 1) == (rc, nif, step, duration, configured)
    assert np.array_equal(cdf, _shutter_cdf(0.0, 1.0, []))  # flat, and configured: the time sample is mirrored inside its bin
    rc, nif, (_, step, duration, configured, cdf) = load({"frames_per_second": 24, "open": 0.25, "close": 0.75})
    assert (0, 705600000 // 24, 705600000 // 24, 1) == (rc, step, duration, configured)
    assert 5 == nif  # 2 whole ticks + the step and the duration each off the tick grid + 1
    assert np.array_equal(cdf, _shutter_cdf(0.25, 0.75, [])) and cdf[0] == 0 and cdf[63] == 1 and np.all(np.diff(cdf) > 0)
    slope = [0.1, 0.0, 0.2, 1.0, 0.8, 1.0, 0.9, 0.0]
    rc, nif, (_, step, duration, configured, cdf) = load({"frame_step": 1 / 30, "open": 0.3, "close": 0.7, "slope": slope, "motion_blur": False})
    assert (0, 2 * tick, 0, 1) == (rc, step, duration, configured) and 2 == nif
    assert np.array_equal(cdf, _shutter_cdf(0.3, 0.7, slope))
    rc, nif, (_, step, duration, configured, cdf) = load({"frames_per_second": 4})  # 15 ticks per frame
    assert (0, 705600000 // 4, 705600000 // 4) == (rc, step, duration) and 16 == nif
    assert 0 == load({"frames_per_second": 1})[0]       # a one-second exposure: 61 keyframes
    assert -1 == load({"frames_per_second": 0.9})[0]    # 66.7 ticks per frame: more keyframes than the view block holds


@pytest.mark.gpu
@pytest.mark.parametrize("shutter", [{"frames_per_second": 30, "open": 0.3, "close": 0.6, "slope": [0.1, 0.0, 0.2, 1.0, 0.7, 1.0, 0.9, 0.0]},
                                     {"motion_blur": False}, {}, {"frames_per_second": 24, "open": 0.5},
                                     {"frames_per_second": 4, "open": 0.1, "close": 0.9}])  # a quarter of a second: 16 camera keyframes
def test_render_take_with_a_shutter(product_lib, oracle, tmp_path, shutter):
    """A moving prop and a moving camera under a configured shutter: the frame's step and duration, the time samples drawn through
    the shutter's distribution (camera_base.zig:121-136).  The device's film is the oracle's bit for bit, frame 1 differs from
    frame 0 (the scene moves), and with motion_blur off every sample of a frame sees the same instant."""
    from util import oracle_render
    keys = [{"time": k / 20.0, "transformation": {"position": [-1.5 + 0.9 * k, 1.2, 0.4 * k], "scale": [0.7, 0.7, 0.7]}} for k in range(6)]
    scene = {"materials": _MATERIALS,
             "entities": [{"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"],
                           "transformation": {"position": [0, 4, 0], "rotation": [-90, 0, 0], "scale": [2, 2, 1]}},
                          {"type": "Prop", "shape": {"type": "Rectangle"}, "materials": ["grey"], "transformation": {"rotation": [90, 0, 0], "scale": [12, 12, 1]}},
                          {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["red"], "animation": {"keyframes": keys}}]}
    with open(tmp_path / "m.scene", "w") as f:
        json.dump(scene, f)
    camera_keys = [{"time": k / 10.0, "transformation": {"position": [0.3 * k, 1.0, -6.0]}} for k in range(3)]
    with open(tmp_path / "m.take", "w") as f:
        json.dump({"scene": "m.scene", "camera": {"Perspective": {"resolution": [24, 16], "shutter": shutter}, "transformation": {"position": [0, 1, -6]},
                                                   "animation": {"keyframes": camera_keys}}}, f)
    films = {}
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "m.take").encode()) == 0
        for frame in (0, 1):
            e.start_frame(frame)
            e.render_iterations(8)
            e.synchronize()
            films[frame] = e.download_film(24, 16)
            sd = e.lib.su_ext_scene_desc().contents
            vd = e.lib.su_ext_view_desc().contents
            assert 1 == vd.shutter_configured
            assert np.isfinite(films[frame]).all() and films[frame][..., :3].max() > 0
            assert np.array_equal(films[frame], oracle_render(oracle, sd, vd, 8, frame=frame)), f"frame {frame}"
            assert e.lib.zygpu_clear_film(e.context(), ctypes.c_float(0.0)) == 0
    assert not np.array_equal(films[0], films[1])


def test_gltf_scene(product_lib, tmp_path):
    """A take whose scene is a glTF 2 document (util/gltf_loader.zig; scene_loader.zig:170-174 tries it first): every node with a mesh
    becomes a prop, the mesh's primitives its parts, pbrMetallicRoughness materials become Substitutes; right-handed -> left-handed:
    z of translations, positions, normals and the rotation quaternion's z mirrored, the triangles' winding turned.  The same geometry
    written as a JSON mesh with those conversions applied, in a scene file with the same materials and transformation, compiles to the
    same arrays.  UInt16 and UInt32 indices, a strided bufferView, a node without a mesh (a camera) ignored."""
    import struct
    idx, pos, nrm, uv = scenes.box_mesh((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    idx = np.asarray(idx, dtype=np.uint32).reshape(-1, 3)
    pos, nrm, uv = (np.asarray(a, dtype=np.float32) for a in (pos, nrm, uv))
    half = len(idx) // 2
    # glTF side: the mesh as a right-handed one (z mirrored, winding turned back), two primitives sharing the vertex streams' layout
    g_pos, g_nrm = pos.reshape(-1, 3) * [1, 1, -1], nrm.reshape(-1, 3) * [1, 1, -1]
    g_idx = idx[:, [0, 2, 1]]
    blob = bytearray()
    views, accessors = [], []

    def add(data, component, count, stride=0):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data), **({"byteStride": stride} if stride else {})})
        accessors.append({"bufferView": len(views) - 1, "byteOffset": 0, "componentType": component, "count": count})
        blob.extend(data)
        return len(accessors) - 1
    interleaved = np.concatenate([g_pos, g_nrm], axis=1).astype(np.float32)  # POSITION + NORMAL in one strided view
    a_pos = add(interleaved.tobytes(), 5126, len(g_pos), stride=24)
    accessors.append({"bufferView": views.index(views[-1]), "byteOffset": 12, "componentType": 5126, "count": len(g_nrm)})
    a_nrm = len(accessors) - 1
    a_uv = add(uv.astype(np.float32).ravel().tobytes(), 5126, len(g_pos))
    a_i16 = add(g_idx[:half].astype(np.uint16).tobytes(), 5123, 3 * half)
    a_i32 = add(g_idx[half:].astype(np.uint32).tobytes(), 5125, 3 * (len(idx) - half))
    with open(tmp_path / "box.bin", "wb") as f:
        f.write(bytes(blob))
    attributes = {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv}
    gltf = {"asset": {"version": "2.0"}, "buffers": [{"uri": "box.bin", "byteLength": len(blob)}], "bufferViews": views, "accessors": accessors,
            "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.2, 0.1, 1.0], "roughnessFactor": 0.4, "metallicFactor": 0.0},
                           "extensions": {"KHR_materials_ior": {"ior": 1.4}}},
                          {"pbrMetallicRoughness": {"roughnessFactor": 0.25}, "emissiveFactor": [2.0, 1.5, 1.0]}],
            "cameras": [{"perspective": {"yfov": 0.6}}],
            "meshes": [{"primitives": [{"attributes": attributes, "indices": a_i16, "material": 0},
                                       {"attributes": attributes, "indices": a_i32, "material": 1}]}],
            "nodes": [{"camera": 0, "translation": [0, 1, 5]},
                      {"mesh": 0, "translation": [0.5, 1.0, 2.0], "rotation": [0.0, 0.38268343, 0.0, 0.92387953], "scale": [1.0, 2.0, 1.0]}]}
    with open(tmp_path / "g.gltf", "w") as f:
        json.dump(gltf, f)
    with open(tmp_path / "g.take", "w") as f:
        json.dump({"scene": "g.gltf", "camera": {"Perspective": {"resolution": [16, 12]}, "transformation": {"position": [0, 1, -6]}}}, f)
    # the same prop through the scene-file loader: the two primitives repeat the vertices (as loadMesh concatenates them)
    nv = len(g_pos)
    mesh = {"geometry": {"parts": [{"start_index": 0, "num_indices": 3 * half, "material_index": 0},
                                   {"start_index": 3 * half, "num_indices": 3 * (len(idx) - half), "material_index": 1}],
                         "vertices": {"positions": np.concatenate([pos.ravel(), pos.ravel()]).tolist(),
                                      "normals": np.concatenate([nrm.ravel(), nrm.ravel()]).tolist(),
                                      "texture_coordinates_0": np.concatenate([uv.ravel(), uv.ravel()]).tolist()},
                         "indices": np.concatenate([idx[:half].ravel(), idx[half:].ravel() + nv]).tolist()}}
    with open(tmp_path / "box.json", "w") as f:
        json.dump(mesh, f)
    GLTF_YAW = 45.0  # the quaternion (0, sin 22.5, 0, cos 22.5) of the node, as the scene files' Euler angles in degrees
    scene = {"materials": [{"name": "a", "rendering": {"Substitute": {"color": [0.8, 0.2, 0.1], "roughness": 0.4, "metallic": 0.0, "ior": 1.4}}},
                           {"name": "b", "rendering": {"Substitute": {"color": [0.7, 0.7, 0.7], "roughness": 0.25, "metallic": 1.0,
                                                                      "emittance": {"spectrum": [2.0, 1.5, 1.0]}, "ior": 1.5}}}],
             "entities": [{"type": "Prop", "shape": {"file": "box.json"}, "materials": ["a", "b"],
                           "transformation": {"position": [0.5, 1.0, -2.0], "rotation": [0, GLTF_YAW, 0], "scale": [1.0, 2.0, 1.0]}}]}
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "g.take").encode()) == 0
        from_gltf = _compiled(e)
    with open(tmp_path / "j.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / "j.take", "w") as f:
        json.dump({"scene": "j.scene", "camera": {"Perspective": {"resolution": [16, 12]}, "transformation": {"position": [0, 1, -6]}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "j.take").encode()) == 0
        from_json = _compiled(e)
    assert from_gltf[0] == from_json[0] and from_gltf[2] == from_json[2] and from_gltf[3] == from_json[3] and from_gltf[4] == from_json[4]
    np.testing.assert_allclose(from_gltf[1], from_json[1], rtol=0, atol=1e-6)
    assert 0 == from_gltf[6] == from_json[6]  # (gltf_loader.zig creates props, never lights: an emissive part is only found by rays)
    # a base colour texture: textures[].source -> images[].uri, read as a file texture (PNG) with KHR_texture_transform's scale
    texels = np.arange(4 * 4 * 3, dtype=np.uint8).reshape(4, 4, 3) * 5
    _png(tmp_path / "albedo.png", texels, [0, 4])
    gltf["images"] = [{"uri": "albedo.png"}]
    gltf["textures"] = [{"source": 0}]
    gltf["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0, "extensions": {"KHR_texture_transform": {"scale": [2.0, 3.0]}}}
    with open(tmp_path / "g.gltf", "w") as f:
        json.dump(gltf, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "g.take").encode()) == 0
        assert e.lib.su_ext_compile(0) == 0
        sd = e.lib.su_ext_scene_desc().contents
        assert 1 == sd.num_images and (0, 4, 4, texels.tobytes()) == _image_bytes(sd, 0)
        textured = [sd.materials[i] for i in range(sd.num_materials) if sd.materials[i].color_image == 0]
        assert 1 == len(textured) and [2.0, 3.0] == list(textured[0].color_scale)
    # what the Substitute of this build has no key for is refused by name
    gltf["materials"][0]["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": 0}
    with open(tmp_path / "g.gltf", "w") as f:
        json.dump(gltf, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "g.take").encode()) == -1


def test_portal_entity_loads(product_lib, tmp_path):
    """{"type": "Portal", "shape": ..., "light": n} (scene_loader.zig:247-248, 371-397): the prop takes light n's material, light n is
    hidden and becomes a prototype, and the Portal's own light is created with the link (scene_loader.zig:316-318)."""
    scene = {"materials": [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}},
                           {"name": "env", "rendering": {"Light": {"emittance": {"spectrum": [1, 1, 1], "emission_map": {"usage": "Emission", "id": 0}}}}}],
             "entities": [{"type": "Light", "shape": {"type": "Dome"}, "materials": ["env"]},
                          {"type": "Portal", "shape": {"type": "Rectangle"}, "light": 0,
                           "transformation": {"position": [0, 2, 0], "rotation": [-90, 0, 0], "scale": [1.5, 1.5, 1]}},
                          {"type": "Prop", "shape": {"type": "Rectangle"}, "materials": ["grey"], "transformation": {"rotation": [90, 0, 0], "scale": [4, 4, 1]}}]}
    with open(tmp_path / "s.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / "s.take", "w") as f:
        json.dump({"scene": "s.scene", "camera": {"Perspective": {"resolution": [8, 8], "transformation": {"position": [0, 1, -4]}}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.image(scenes.sky_image(32, 16)) == 0
        assert e.lib.su_ext_load_take(str(tmp_path / "s.take").encode()) == 0
        sd, _ = e.compile()
        assert sd.num_lights == 2 and sd.num_portal_lights == 1
        assert sd.lights[0].prototype == 1 and sd.lights[1].light_class == 0 and sd.portal_lights[0].light_link == 0
        assert sd.props[sd.lights[1].prop].shape == scenes.RECTANGLE and not (sd.props[sd.lights[0].prop].flags & 3)


def test_shadow_catcher_keys(product_lib, tmp_path):
    """"shadow_catcher" on an entity (scene_loader.zig:312-313 -> Prop.setShadowCatcher: the prop casts no shadow) and
    visibility.shadow_catcher_light (612-619), which a light carries from the moment it is created (scene.zig:344-372)."""
    scene = {"materials": [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}},
                           {"name": "lamp", "rendering": {"Light": {"emittance": {"spectrum": [5, 5, 5]}}}}],
             "entities": [{"type": "Prop", "shape": {"type": "Rectangle"}, "materials": ["grey"], "shadow_catcher": {},
                           "transformation": {"rotation": [90, 0, 0], "scale": [8, 8, 1]}},
                          {"type": "Light", "shape": {"type": "Distant"}, "materials": ["lamp"], "visibility": {"shadow_catcher_light": True},
                           "transformation": {"rotation": [-60, 0, 0], "scale": [0.05, 0.05, 0.05]}},
                          {"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"], "transformation": {"position": [0, 3, 0]}}]}
    with open(tmp_path / "s.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / "s.take", "w") as f:
        json.dump({"scene": "s.scene", "camera": {"Perspective": {"resolution": [8, 8], "transformation": {"position": [0, 1, -4]}}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "s.take").encode()) == 0
        sd, _ = e.compile()
        catchers = [p for p in range(sd.num_props) if sd.props[p].flags & (1 << 11)]
        assert len(catchers) == 1 and sd.props[catchers[0]].shape == scenes.RECTANGLE
        flags = sorted((sd.props[sd.lights[i].prop].shape, sd.lights[i].shadow_catcher_light) for i in range(sd.num_lights))
        assert flags == [(scenes.DISTANT, 1), (scenes.RECTANGLE, 0)]


def test_sky_entity_loads(product_lib, tmp_path):
    """{"type": "Sky", "parameters": {...}} (scene_loader.zig:262-263, 704-712): the sun and the sky as two lights, placed by
    Scene.compile from the sun's rotation; the entity's visibility is the sun's, and the sky follows it (sky.zig:100)."""
    scene = {"materials": [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}}],
             "entities": [{"type": "Sky", "parameters": {"sun": {"rotation": [-40, 30, 0]}, "turbidity": 3.0}, "visibility": {"in_camera": False}},
                          {"type": "Prop", "shape": {"type": "Rectangle"}, "materials": ["grey"], "transformation": {"rotation": [90, 0, 0], "scale": [8, 8, 1]}}]}
    with open(tmp_path / "s.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / "s.take", "w") as f:
        json.dump({"scene": "s.scene", "camera": {"Perspective": {"resolution": [8, 8], "transformation": {"position": [0, 1, -4]}}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "s.take").encode()) == 0
        sd, _ = e.compile()
        assert sd.num_lights == 2 and sd.num_sun_radiance == 1024 and sd.num_infinite_props == 2
        light_props = sorted(sd.lights[i].prop for i in range(2))
        assert sorted(sd.props[p].shape for p in light_props) == [scenes.CANOPY, scenes.DISTANT]
        for p in light_props:
            assert not (sd.props[p].flags & 1) and (sd.props[p].flags & 2)  # hidden from the camera, seen in reflections
        sun = [p for p in light_props if sd.props[p].shape == scenes.DISTANT][0]
        assert abs(sd.trafos[sun].r[2][1] - np.sin(np.radians(-40.0))) < 1e-6


def test_unknown_entity_type_is_a_transformation_node(product_lib, tmp_path):
    """scene_loader.zig:241-264: a type that names no entity class creates none, but its transformation and children count."""
    scene = {"materials": [{"name": "grey", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5]}}}],
             "entities": [{"type": "Group", "transformation": {"position": [0, 5, 0]},
                           "entities": [{"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["grey"], "transformation": {"position": [1, 0, 0]}}]},
                          {"type": "Prop", "materials": ["grey"]}]}  # no shape: skipped (`catch continue`)
    with open(tmp_path / "s.scene", "w") as f:
        json.dump(scene, f)
    with open(tmp_path / "s.take", "w") as f:
        json.dump({"scene": "s.scene", "camera": {"Perspective": {"resolution": [8, 8], "transformation": {"position": [0, 0, -4]}}}}, f)
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(str(tmp_path / "s.take").encode()) == 0
        sd, _ = e.compile()
        spheres = [i for i in range(sd.num_props) if sd.props[i].shape == scenes.SPHERE]
        assert len(spheres) == 1 and sd.num_props == 2  # the camera is a prop too
        np.testing.assert_allclose(np.array(list(sd.trafos[spheres[0]].position)[:3]) + np.array([0, 0, -4]), [1, 5, 0], atol=1e-6)


@pytest.mark.gpu
def test_render_take_and_export(product_lib, tmp_path):
    """The whole drop-in chain on files: take -> scene -> meshes -> render -> image_00_000000.png, the film being the one the
    API-built scene renders."""
    w, h, spp = 48, 32, 4
    take = scenes.write_cornell_take(str(tmp_path), w, h, spp)
    os.environ["ZYG_EXPORT_DIR"] = str(tmp_path)
    try:
        with zyg_amd.Engine(product_lib) as e:
            assert e.lib.su_ext_load_take(take.encode()) == 0
            e.start_frame(0)
            e.render_iterations(spp)
            e.synchronize()
            film = e.download_film(w, h)
            assert e.lib.su_export_frame() == 0
        assert (tmp_path / "image_00_000000.png").stat().st_size > w * h * 3
    finally:
        del os.environ["ZYG_EXPORT_DIR"]
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, w, h, spp, material="substitute")
        e.sensor(json.dumps({}))
        e.start_frame(0)
        e.render_iterations(spp)
        e.synchronize()
        assert np.array_equal(film, e.download_film(w, h))


@pytest.mark.gpu
def test_render_take_with_instancer_and_animation(product_lib, oracle, tmp_path):
    """A take whose scene has an Instancer (flattened into props) and an animated entity (its keyframes resampled into the frame's
    interpolation frames): the device renders the oracle's film bit for bit, at frame 0 and at frame 2 (other interpolation frames),
    and the moving sphere blurs (frame 2's film differs from frame 0's)."""
    from util import oracle_render
    idx, pos, nrm, uv = scenes.box_mesh((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    scenes.write_sub_mesh(str(tmp_path / "cube.sub"), idx, pos, nrm, uv, delta=True)
    keys = [{"time": k / 30.0, "transformation": {"position": [-1.5 + 1.2 * k, 1.6, 0.5 * k], "scale": [0.6, 0.6, 0.6]}} for k in range(4)]
    scene = {"materials": _MATERIALS,
             "entities": [{"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"],
                           "transformation": {"position": [0, 4, 0], "rotation": [-90, 0, 0], "scale": [2, 2, 1]}},
                          {"type": "Prop", "shape": {"type": "Rectangle"}, "materials": ["grey"],
                           "transformation": {"rotation": [90, 0, 0], "scale": [12, 12, 1]}},
                          {"type": "Instancer", "transformation": {"position": [0, 0.5, 1]},
                           "prototypes": [{"type": "Prop", "shape": {"file": "cube.sub"}, "materials": ["red"]}],
                           "instances": {"prototypes": [0, 0, 0], "transformations": [{"position": [-1.5, 0, 0]}, {"position": [0, 0, 0.5], "rotation": [0, 30, 0]},
                                                                                     {"position": [1.5, 0, 0], "scale": [0.5, 1.5, 0.5]}]}},
                          {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["grey"], "animation": {"keyframes": keys}}]}
    take = _write_take(tmp_path, scene, "moving")
    films = {}
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(take) == 0
        for frame in (0, 2):
            e.start_frame(frame)
            e.render_iterations(8)
            e.synchronize()
            films[frame] = e.download_film(16, 12)
            sd = e.lib.su_ext_scene_desc().contents
            vd = e.lib.su_ext_view_desc().contents
            assert np.isfinite(films[frame]).all() and films[frame][..., :3].max() > 0
            assert np.array_equal(films[frame], oracle_render(oracle, sd, vd, 8, frame=frame)), f"frame {frame}"
            assert e.lib.zygpu_clear_film(e.context(), ctypes.c_float(0.0)) == 0
    assert not np.array_equal(films[0], films[2])


def _image_bytes(sd, i):
    img = sd.images[i]
    bpp = {0: 3, 1: 12, 2: 1, 3: 4, 4: 2, 5: 8}[img.format]
    return img.format, img.width, img.height, bytes(ctypes.string_at(img.pixels, img.width * img.height * bpp))


def test_file_textures(product_lib, tmp_path):
    """Textures named by "file" (material_provider.zig:438-740, texture_provider.zig:27-65): PNG (png_reader.zig: 8 bits, grey / RGB /
    RGBA, all five scanline filters, several IDAT chunks) and Radiance RGBE (rgbe_reader.zig: run-length encoded scanlines) files found
    through the mounts, channels picked by the usage's default swizzle or the "swizzle" key, "invert" for one-channel images.  The
    compiled images are the ones su_image_create gets for the same pixels; a file that cannot be read fails the compile by name."""
    rng = np.random.default_rng(5)
    rgba = rng.integers(0, 256, size=(9, 12, 4), dtype=np.uint8)
    grey = rng.integers(0, 256, size=(5, 8, 1), dtype=np.uint8)
    os.mkdir(tmp_path / "tex")
    _png(tmp_path / "tex" / "rgba.png", rgba, [0, 1, 2, 3, 4])
    _png(tmp_path / "tex" / "rgb.png", rgba[..., :3], [4, 3, 1])
    _png(tmp_path / "tex" / "grey.png", grey, [2, 4])
    # a Radiance picture: 16 x 4, run-length encoded scanlines with runs and literal spans
    hdr = rng.integers(0, 256, size=(4, 16, 4), dtype=np.uint8)
    hdr[..., 3] = rng.integers(120, 136, size=(4, 16), dtype=np.uint8)
    hdr[:, 4:12, :] = hdr[:, 4:5, :]  # a run of 8 equal pixels per scanline
    hdr[1, 2, 3] = 0  # a zero pixel
    with open(tmp_path / "tex" / "sky.hdr", "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 16\n")
        for y in range(4):
            f.write(bytes([2, 2, 0, 16]))
            for c in range(4):
                ch = hdr[y, :, c]
                f.write(bytes([4]) + ch[0:4].tobytes() + bytes([128 + 8, int(ch[4])]) + bytes([4]) + ch[12:16].tobytes())
    materials = [
        {"name": "colour", "rendering": {"Substitute": {"color": {"file": "tex/rgba.png"}, "roughness": {"file": "tex/rgba.png", "swizzle": "Y"},
                                                        "metallic": {"file": "tex/grey.png", "invert": True}, "normal": {"file": "tex/rgb.png"}}}},
        {"name": "lamp", "rendering": {"Light": {"emittance": {"spectrum": [1, 1, 1], "emission_map": {"file": "tex/sky.hdr"}}}}},
        {"name": "masked", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5], "mask": {"file": "tex/rgba.png"}}}}]
    scene = {"materials": materials,
             "entities": [{"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"], "transformation": {"position": [0, 3, 0]}},
                          {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["colour"]},
                          {"type": "Prop", "shape": {"type": "Rectangle"}, "materials": ["masked"], "transformation": {"position": [2, 0, 0]}}]}
    take = _write_take(tmp_path, scene, "textured")
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(take) == 0
        assert e.lib.su_ext_compile(0) == 0
        sd = e.lib.su_ext_scene_desc().contents
        got = [_image_bytes(sd, i) for i in range(sd.num_images)]
    expected = {"colour": (0, 12, 9, rgba[..., :3].tobytes()), "roughness": (2, 12, 9, rgba[..., 1].tobytes()),
                "metallic": (2, 8, 5, (255 - grey[..., 0]).tobytes()), "normal": (4, 12, 9, rgba[..., :2].tobytes()),
                "mask": (2, 12, 9, rgba[..., 3].tobytes())}
    for name, image in expected.items():
        assert image in got, name
    # the Radiance picture: (mantissa + 0.5) * 2^(e - 136), sRGB -> AP1, rounded to f16
    srgb = (hdr[..., :3].astype(np.float32) + 0.5) * np.exp2(hdr[..., 3:4].astype(np.float32) - 136.0).astype(np.float32)
    srgb[hdr[..., 3] == 0] = 0.0
    m = np.array([[0.61309732, 0.33952285, 0.04737928], [0.07019422, 0.91635557, 0.01345259], [0.02061560, 0.10956983, 0.86981512]], dtype=np.float32)
    ap1 = (srgb.reshape(-1, 3) @ m.T).astype(np.float16).astype(np.float32).reshape(4, 16, 3)
    floats = [np.frombuffer(b, dtype=np.float32).reshape(h, w, 3) for f, w, h, b in got if 1 == f]
    assert 1 == len(floats)
    np.testing.assert_allclose(floats[0], ap1, rtol=1e-3, atol=0)  # (the matrix product's rounding against numpy's; both round to f16)
    # a missing file: the compile fails and names it
    materials[0]["rendering"]["Substitute"]["color"] = {"file": "tex/nowhere.png"}
    messages = []
    callback = zyg_amd.LOG_CALLBACK(lambda level, text: messages.append((level, text.decode())))
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_register_log.argtypes = [zyg_amd.LOG_CALLBACK]
        assert e.lib.su_register_log(callback) == 0
        try:
            assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, "missing")) == 0
            e.lib.zygpu_last_error.restype = ctypes.c_char_p
            assert e.lib.su_ext_compile(0) != 0
        finally:
            e.lib.su_register_log(zyg_amd.LOG_CALLBACK())


def _exr(path, planes, half, data_window=None, display=None):
    """A ZIP-compressed scan-line OpenEXR file: planes = {channel name: (H, W) float array}; written in alphabetical channel order,
    16-line blocks with the predictor (interleave the halves of the byte stream, delta-code), a block that does not shrink stored raw."""
    import struct
    import zlib
    names = sorted(planes)
    h, w = planes[names[0]].shape
    dw = data_window or (0, 0, w - 1, h - 1)
    disp = display or dw
    dtype = np.float16 if half else np.float32
    header = b"\x76\x2f\x31\x01" + struct.pack("<I", 2)

    def attribute(name, kind, body):
        return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<I", len(body)) + body
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<IBBBBII", 1 if half else 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    header += attribute("channels", "chlist", chlist) + attribute("compression", "compression", bytes([3]))
    header += attribute("dataWindow", "box2i", struct.pack("<4i", *dw)) + attribute("displayWindow", "box2i", struct.pack("<4i", *disp))
    header += attribute("lineOrder", "lineOrder", bytes([0])) + attribute("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += attribute("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attribute("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    blocks = []
    for y0 in range(0, h, 16):
        rows = min(16, h - y0)
        raw = b"".join(planes[n][y].astype(dtype).tobytes() for y in range(y0, y0 + rows) for n in names)
        a = np.frombuffer(raw, dtype=np.uint8)
        split = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
        delta = np.concatenate([split[:1], (split[1:] - split[:-1] + 128 + 256) & 255]).astype(np.uint8).tobytes()
        packed = zlib.compress(delta, 6)
        body = packed if len(packed) < 16 * w * len(names) * (2 if half else 4) else raw
        blocks.append(struct.pack("<iI", dw[1] + y0, len(body)) + body)
    offset, table = len(header) + 8 * len(blocks), b""
    for b in blocks:
        table += struct.pack("<Q", offset)
        offset += len(b)
    with open(path, "wb") as f:
        f.write(header + table + b"".join(blocks))


def test_exr_file_textures(product_lib, tmp_path):
    """OpenEXR textures as exr_reader.zig reads them: ZIP-compressed scan-line files, all channels half or all float, the channels in
    alphabetical order (R last); a colour usage converts sRGB -> AP1 (a half image keeps f16 values); a Weight usage takes R; a data
    window inside a larger display window lands at its place; other compressions are refused by name."""
    rng = np.random.default_rng(11)
    h, w = 20, 12  # two scan-line blocks
    rgb = {c: rng.random((h, w)).astype(np.float32) * 4.0 for c in "RGB"}
    _exr(tmp_path / "env_half.exr", rgb, half=True)
    _exr(tmp_path / "weights.exr", {"B": rgb["B"], "G": rgb["G"], "R": rgb["R"]}, half=False)
    _exr(tmp_path / "inset.exr", {"R": rgb["R"][:4, :6]}, half=False, data_window=(2, 0, 7, 3), display=(0, 0, 9, 3))
    materials = [{"name": "lamp", "rendering": {"Light": {"emittance": {"spectrum": [1, 1, 1], "emission_map": {"file": "env_half.exr"}}}}},
                 {"name": "rough", "rendering": {"Substitute": {"color": [0.5, 0.5, 0.5], "roughness": {"file": "weights.exr"},
                                                                "metallic": {"file": "inset.exr"}}}}]
    scene = {"materials": materials,
             "entities": [{"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"], "transformation": {"position": [0, 3, 0]}},
                          {"type": "Prop", "shape": {"type": "Sphere"}, "materials": ["rough"]}]}
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, "exr")) == 0
        assert e.lib.su_ext_compile(0) == 0
        sd = e.lib.su_ext_scene_desc().contents
        got = [_image_bytes(sd, i) for i in range(sd.num_images)]
    colour = [np.frombuffer(b, dtype=np.float32).reshape(hh, ww, 3) for f, ww, hh, b in got if 1 == f]
    assert 1 == len(colour)
    m = np.array([[0.61309732, 0.33952285, 0.04737928], [0.07019422, 0.91635557, 0.01345259], [0.02061560, 0.10956983, 0.86981512]], dtype=np.float32)
    srgb = np.stack([rgb[c].astype(np.float16).astype(np.float32) for c in "RGB"], axis=-1)
    np.testing.assert_allclose(colour[0], (srgb.reshape(-1, 3) @ m.T).astype(np.float16).astype(np.float32).reshape(h, w, 3), rtol=1e-3)
    singles = {(ww, hh): np.frombuffer(b, dtype=np.float32).reshape(hh, ww) for f, ww, hh, b in got if 3 == f}
    assert np.array_equal(singles[w, h], rgb["R"])
    inset = np.zeros((4, 10), dtype=np.float32)
    inset[:, 2:8] = rgb["R"][:4, :6]
    assert np.array_equal(singles[10, 4], inset)
    # an uncompressed file: the reference reads ZIP only, and so does this
    data = bytearray(open(tmp_path / "weights.exr", "rb").read())
    at = data.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    data[at] = 0
    open(tmp_path / "weights.exr", "wb").write(bytes(data))
    with zyg_amd.Engine(product_lib) as e:
        assert e.lib.su_ext_load_take(_write_take(tmp_path, scene, "exr")) == 0
        assert e.lib.su_ext_compile(0) != 0
