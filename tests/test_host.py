"""Host-side logic without a GPU: the C-ABI library loads and exports every declared symbol, the su_* protocol keeps
the reference's error convention (src/capi/capi.zig), and the scene-build products (SAH BVH with spatial splits,
top-level trees, light tree) satisfy their invariants and agree with brute force."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:su|zygpu)_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["zygpu.h", "zyg_capi.h"])
def test_library_exports_every_declared_symbol(product_lib, header):
    names = _declared_functions(header)
    assert len(names) >= (15 if header == "zygpu.h" else 30)
    for n in names:
        assert hasattr(product_lib, n), n


def test_reference_c_abi_has_30_entries():
    names = [n for n in _declared_functions("zyg_capi.h") if n.startswith("su_") and not n.startswith("su_ext_")]
    assert len(names) == 30


def test_su_protocol_error_codes(product_lib):
    lib = product_lib
    assert lib.su_release() == -1  # not initialised
    assert lib.su_init() == 0
    try:
        assert lib.su_init() == -1  # capi.zig:58-60
        assert lib.su_sampler_create(16) == -1  # capi.zig:215-221: always -1
        assert lib.su_material_update(12345, b'{"roughness": 0.2}') == -3
        assert lib.su_material_create(zyg_amd.NULL, b"not json") == -1
        assert lib.su_material_create(zyg_amd.NULL, b'{"rendering": {"Hair": {}}}') == -1
        m = lib.su_material_create(zyg_amd.NULL, b'{"rendering": {"Substitute": {"color": [1, 0, 0], "roughness": 0.5}}}')
        assert m >= 1  # id 0 is the fallback material
        assert lib.su_material_update(m, b'{"roughness": 0.25}') == 0
        assert lib.su_prop_create(999, 0, None) == -1
        assert lib.su_prop_create_instance(999) == -1
        assert lib.su_light_create(999) == -1
        assert lib.su_prop_set_visibility(999, 1, 1, 0) == -1
        cam = lib.su_perspective_camera_create(64, 32)
        assert cam == 0
        dims = (ctypes.c_int32 * 2)()
        assert lib.su_camera_sensor_dimensions(dims) == 0 and list(dims) == [64, 32]
        p = lib.su_prop_create(scenes.RECTANGLE, 0, None)  # materials padded with the fallback (capi.zig:441-447)
        assert p == 1
        inst = lib.su_prop_create_instance(p)
        assert inst == 2
        assert lib.su_render_iterations(1) == -1  # before su_start_frame
        assert lib.su_resolve_frame(zyg_amd.NULL) == -1
        assert lib.su_image_create(zyg_amd.NULL, 0, 3, 4, 4, 1, 3, None) == -1  # image textures are out of scope
    finally:
        assert lib.su_release() == 0


def test_cornell_compile_products(engine):
    ids = scenes.cornell(engine, 64, 48, 4)
    sd, vd = engine.compile()
    assert sd.num_props == 10 and sd.num_lights == 2 and sd.num_meshes == 2
    # lights are unoccluding `Light` entities: not in the solid tree (scene.zig:324-342)
    solid = {sd.solid.indices[i] for i in range(sd.solid.num_indices)}
    unocc = {sd.unoccluding.indices[i] for i in range(sd.unoccluding.num_indices)}
    assert unocc == set(ids["lights"]) and solid == set(ids["walls"]) | set(ids["meshes"])
    # camera-relative world: the camera prop sits at the origin (space.zig:99-108)
    assert list(vd.camera_trafo.position)[:3] == [0.0, 0.0, 0.0]
    # light power = max(emission) * area (emittance.zig:61-71, light.zig:67-77)
    mat = sd.materials[sd.material_ids[sd.props[ids["lights"][0]].parts_start]]
    assert sd.light_aabbs[0].min[3] == pytest.approx(max(mat.emission) * 0.25, rel=1e-6)
    assert vd.integrator == 3 and vd.max_depth_surface == 16 and vd.split_threshold == pytest.approx(0.5 ** 4)
    assert vd.filter_radius_int == 0 and vd.num_expected_samples == 4


def test_default_sensor_is_mitchell_radius_2(engine):
    engine.camera(8, 8)
    sd, vd = engine.compile()
    assert vd.filter_radius_int == 2 and vd.filter_radius == 2.0  # take.zig:59-64
    taps = np.array(vd.filter_samples[:30])
    # normalised so that 2 * sum(midpoint rule over [0, r]) = 1 (sensor.zig:105-120, 630-645)
    xs = (np.arange(64) + 0.5) * (2.0 / 64)
    vals = np.interp(xs, np.linspace(0, 2, 30), taps)
    assert 2 * vals.sum() * (2.0 / 64) == pytest.approx(1.0, rel=1e-4)
    assert taps[0] > 0 and taps[-1] == pytest.approx(0.0, abs=1e-6)


def _mesh_desc(engine, n=40):
    idx, pos, nrm, uv = scenes.heightfield_mesh(n=n, size=1.5, height=0.6)
    engine.camera(16, 16)
    shape = engine.triangle_mesh(idx, pos, nrm, uv)
    p = engine.prop(shape, [0])
    engine.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 0)))
    sd, vd = engine.compile()
    return sd, idx, pos, p


def test_triangle_bvh_invariants(engine):
    sd, idx, pos, _ = _mesh_desc(engine)
    m = sd.meshes[0]
    ntri_in = idx.size // 3
    assert m.num_triangles >= ntri_in  # spatial splits may duplicate references
    tris = np.ctypeslib.as_array(m.triangles, shape=(m.num_triangles, 3))
    # every input triangle survives (as a vertex-index triple) in at least one leaf
    have = {tuple(t) for t in tris.tolist()}
    assert {tuple(t) for t in idx.reshape(-1, 3).tolist()} <= have
    P = np.ctypeslib.as_array(m.positions, shape=(m.num_vertices * 3 + 1,))[:-1].reshape(-1, 3)
    covered = np.zeros(m.num_triangles, dtype=bool)
    leaves = 0
    stack = [0]
    while stack:
        n = m.nodes[stack.pop()]
        lo, hi = np.array(n.min[:]), np.array(n.max[:])
        assert np.all(lo <= hi)
        if n.count:
            leaves += 1
            assert n.count <= 0x2FF
            first = n.children_or_first
            assert not covered[first:first + n.count].any()
            covered[first:first + n.count] = True
            for t in tris[first:first + n.count]:
                tri_lo, tri_hi = P[t].min(axis=0), P[t].max(axis=0)
                # the (possibly clipped) reference overlaps the leaf box
                assert np.all(tri_lo <= hi + 1e-6) and np.all(tri_hi >= lo - 1e-6)
        else:
            a = n.children_or_first
            assert a % 2 == 1  # children are adjacent pairs (a, a + 1) allocated right after the root
            for c in (a, a + 1):
                cn = m.nodes[c]
                assert np.all(np.array(cn.min[:]) >= lo - 1e-6) and np.all(np.array(cn.max[:]) <= hi + 1e-6)
            stack += [a, a + 1]
    assert covered.all() and leaves > ntri_in // 8
    areas = np.ctypeslib.as_array(m.part_areas, shape=(m.num_parts,))
    e1, e2 = P[tris[:, 1]] - P[tris[:, 0]], P[tris[:, 2]] - P[tris[:, 0]]
    assert areas[0] == pytest.approx(0.5 * np.linalg.norm(np.cross(e1, e2), axis=1).sum(), rel=1e-4)


def test_oct_normals_round_trip(engine):
    sd, idx, pos, _ = _mesh_desc(engine, n=24)
    m = sd.meshes[0]
    enc = np.ctypeslib.as_array(m.normals, shape=(m.num_vertices, 2)).astype(np.float64)
    o = enc / 32768.0 - 1.0
    v = np.stack([o[:, 0], o[:, 1], -1.0 + np.abs(o[:, 0]) + np.abs(o[:, 1])], axis=1)
    t = np.maximum(v[:, 2], 0.0)
    v[:, 0] += np.where(v[:, 0] > 0, -t, t)
    v[:, 1] += np.where(v[:, 1] > 0, -t, t)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    _, _, nrm, _ = scenes.heightfield_mesh(n=24, size=1.5, height=0.6)
    assert np.max(np.abs(v - nrm)) < 3e-4  # snorm16 quantisation of the octahedral map


def test_oracle_traversal_matches_brute_force(engine, oracle):
    """Closest hits through TLAS + BLAS (oracle) against an O(N) float64 Moeller-Trumbore over the input triangles."""
    sd, idx, pos, prop = _mesh_desc(engine, n=32)
    rng = np.random.default_rng(11)
    n = 3000
    o = np.stack([rng.uniform(-1, 1, n), rng.uniform(0.8, 2.0, n), rng.uniform(-1, 1, n)], axis=1)
    target = np.stack([rng.uniform(-0.9, 0.9, n), np.zeros(n), rng.uniform(-0.9, 0.9, n)], axis=1)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 0.0, d, 2.14748313e9
    hits = np.zeros((n, 8), dtype=np.uint32)
    oracle.zyg_oracle_intersect(ctypes.byref(sd), n, rays.ctypes.data_as(ctypes.c_void_p), 0, hits.ctypes.data_as(ctypes.c_void_p))
    t_oracle = hits[:, 0].view(np.float32)
    hit_oracle = hits[:, 4] != zyg_amd.NULL

    tri = pos[idx.reshape(-1, 3)].astype(np.float64)
    a, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    ro, rd = rays[:, 0:3].astype(np.float64), rays[:, 4:7].astype(np.float64)
    best = np.full(n, np.inf)
    for i in range(n):
        pvec = np.cross(rd[i], e2)
        det = np.einsum("ij,ij->i", e1, pvec)
        inv = 1.0 / np.where(det == 0, np.nan, det)
        tvec = ro[i] - a
        u = np.einsum("ij,ij->i", tvec, pvec) * inv
        qvec = np.cross(tvec, e1)
        v = (qvec @ rd[i]) * inv
        t = np.einsum("ij,ij->i", e2, qvec) * inv
        ok = (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t >= 0)
        if ok.any():
            best[i] = t[ok].min()
    hit_bf = np.isfinite(best)
    disagree = hit_bf != hit_oracle
    assert disagree.mean() < 0.002  # edge-grazing rays only
    both = hit_bf & hit_oracle
    assert both.sum() > n // 2
    np.testing.assert_allclose(t_oracle[both], best[both], rtol=2e-4)


def test_light_tree_pick_pdf_consistency(engine, oracle):
    """randomLight without splitting returns one pick whose pdf equals Tree.pdf of that light (light_tree.zig:346-517),
    and the pdfs over all lights sum to 1."""
    scenes.cornell(engine, 16, 16, 1)
    sd, vd = engine.compile()
    rng = np.random.default_rng(2)
    offs = (ctypes.c_uint32 * 64)()
    pdfs = (ctypes.c_float * 64)()
    for _ in range(200):
        p = rng.uniform(-0.9, 0.9, 3).astype(np.float32)
        p[1] = rng.uniform(0.05, 1.5)
        p[2] += 3.9  # camera-relative
        p[1] -= 1.0
        nrm = np.array([0, 1, 0], dtype=np.float32)
        r = float(rng.uniform())
        vp = lambda v: v.ctypes.data_as(ctypes.c_void_p)
        c = oracle.zyg_oracle_random_light(ctypes.addressof(sd), vp(p), vp(nrm), 0, r, 0.0, offs, pdfs)
        assert c == 1
        pdf = oracle.zyg_oracle_light_pdf(ctypes.addressof(sd), vp(p), vp(nrm), 0, 0.0, offs[0])
        assert pdf == pytest.approx(pdfs[0], rel=1e-6)
        total = sum(oracle.zyg_oracle_light_pdf(ctypes.addressof(sd), vp(p), vp(nrm), 0, 0.0, l) for l in range(sd.num_lights))
        assert total == pytest.approx(1.0, rel=1e-5)
        # with splitting enabled every light with non-zero importance is returned with pdf 1 (max_split_depth >= 1)
        c2 = oracle.zyg_oracle_random_light(ctypes.addressof(sd), vp(p), vp(nrm), 0, r, 1.0, offs, pdfs)
        assert 1 <= c2 <= sd.num_lights


def test_many_lights_tree(engine, oracle):
    """A 5 x 5 grid of quad lights: the light BVH covers every light exactly once and pdfs still sum to one."""
    engine.camera(8, 8)
    lamp = engine.material(scenes.light_material())
    for i in range(5):
        for j in range(5):
            p = engine.prop(scenes.RECTANGLE, [lamp], unoccluding=True)
            engine.transformation(p, scenes.trafo((1, 0, 0), (0, 0, 1), (0.2 + 0.02 * i, 0.2, 1), (i - 2.0, 2.0, j + 1.0)))
            engine.light(p)
    sd, vd = engine.compile()
    lt = sd.light_tree
    assert lt.num_lights == 25 and lt.num_nodes >= 25
    mapping = sorted(lt.light_mapping[i] for i in range(25))
    assert mapping == list(range(25))
    orders = sorted(lt.light_orders[i] for i in range(25))
    assert orders == list(range(25))
    p = np.array([0.1, 0.0, 3.0], dtype=np.float32)
    nrm = np.array([0, 1, 0], dtype=np.float32)
    vp = lambda v: v.ctypes.data_as(ctypes.c_void_p)
    total = sum(oracle.zyg_oracle_light_pdf(ctypes.addressof(sd), vp(p), vp(nrm), 0, 0.0, l) for l in range(25))
    assert total == pytest.approx(1.0, rel=1e-4)


def test_unsupported_features_fail_loudly(engine):
    """A Cube light is outside the hot-path scope: compile reports it instead of silently rendering something else."""
    engine.camera(8, 8)
    lamp = engine.material(scenes.light_material())
    p = engine.prop(1, [lamp])  # built-in Cube (resource/manager.zig:36-44)
    engine.light(p)
    with pytest.raises(RuntimeError):
        engine.compile()


def test_render_without_gpu_fails_loudly(engine):
    """No CPU fallback in the product path: without a HIP device su_start_frame reports an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    scenes.cornell(engine, 16, 16, 1)
    with pytest.raises(RuntimeError, match="su_start_frame failed"):
        engine.start_frame(0)


def test_ggx_albedo_tables_match_reference_entries(product_lib):
    """The scene-compile-time LUT generator (ggx_lut.hpp) against a sample of the entries the reference ships in
    material/ggx_integral.zig (tests/golden/ggx_lut_golden.json; 8 printed decimals, so 1e-6 absolute)."""
    import json
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ggx_lut_golden.json")))
    product_lib.su_ext_ggx_lut.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    product_lib.su_ext_ggx_lut.restype = ctypes.c_uint32
    for name, g in golden.items():
        n = g["length"]
        table = np.zeros(n, dtype=np.float32)
        assert product_lib.su_ext_ggx_lut(g["which"], table.ctypes.data, n) == n, name
        got = table[np.array(g["index"])]
        np.testing.assert_allclose(got, np.array(g["value"], dtype=np.float32), rtol=0, atol=1e-6, err_msg=name)
        assert np.all((table >= 0.0) & (table <= 1.0)), name
    assert product_lib.su_ext_ggx_lut(9, None, 0) == 0


def test_mesh_light_products(engine):
    """Part.configure / buildPrimitive products for an 8 x 8 emissive quad mesh: mappings are permutations, the relative powers
    sum to 1, the primitive tree's leaves hold at most 4 triangles and cover every triangle once."""
    idx, pos, nrm, uv = scenes.quad_mesh(8, 0.5)
    engine.camera(8, 8)
    lamp = engine.material(scenes.light_material())
    p = engine.prop(engine.triangle_mesh(idx, pos, nrm, uv), [lamp])
    engine.transformation(p, scenes.trafo((1, 0, 0), (0, 0, 1), (1, 1, 1), (0, 1.98, 0)))
    engine.light(p)
    sd, _ = engine.compile()
    assert sd.num_mesh_lights == 1 and sd.lights[0].sampler == 0
    ml = sd.mesh_lights[0]
    n = ml.num_triangles
    assert n == 128
    mapping = np.array([ml.triangle_mapping[i] for i in range(n)])
    assert sorted(mapping.tolist()) == list(range(n))
    props = np.array([ml.tri_props[i] for i in range(8 * n)], dtype=np.float32).reshape(n, 8)
    np.testing.assert_allclose(props[:, 7].sum(), 1.0, rtol=1e-5)
    np.testing.assert_allclose(props[:, 4:7], np.tile([[0, 0, 1]], (n, 1)), atol=1e-6)
    t = ml.tree
    assert t.max_split_depth == 6 and t.num_lights == n and t.num_infinite_lights == 0
    seen = []
    for i in range(t.num_nodes):
        node = t.nodes[i]
        if not (node.meta & 1):
            assert 1 <= node.num_lights <= 4
            first = node.meta >> 2
            seen += [t.light_mapping[j] for j in range(first, first + node.num_lights)]
    assert sorted(seen) == list(range(n))
    orders = sorted(t.light_orders[i] for i in range(n))
    assert orders == list(range(n))
    pm = sd.meshes[ml.mesh].primitive_mapping
    assert sorted(pm[i] for i in range(n)) == list(range(n))


def test_images_and_color_maps(engine):
    """su_image_create accepts tightly packed UInt8 x 3 and Float32 x 3 2-D images (capi.zig:236-298 accepts more; the rest
    reports -1) and a Substitute colour map refers to them by id."""
    img = engine.image(scenes.checker_image(4, np.uint8))
    img2 = engine.image(scenes.checker_image(5, np.float32))
    assert (img, img2) == (0, 1)
    lib = engine.lib
    lib.su_image_create.argtypes = [ctypes.c_uint32] * 7 + [ctypes.c_void_p]
    buf = np.zeros((2, 2, 4), dtype=np.float32)
    assert lib.su_image_create(0xFFFFFFFF, 4, 4, 2, 2, 1, 16, buf.ctypes.data) == -1  # 4 channels: not supported here
    assert lib.su_image_create(0xFFFFFFFF, 4, 3, 2, 2, 1, 16, buf.ctypes.data) == -1  # padded pixels are not copied
    engine.camera(8, 8)
    m = engine.material(scenes.substitute_textured(img2, sampler={"filter": "Nearest", "address": "Clamp"}, scale=[2, 3]))
    p = engine.prop(scenes.RECTANGLE, [m])
    engine.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 2)))
    sd, _ = engine.compile()
    mat = sd.materials[m]
    assert mat.color_image == img2 and mat.color_sampler == 0 and tuple(mat.color_scale) == (2.0, 3.0)
    assert sd.num_images == 2 and sd.images[1].width == 5 and sd.images[0].format == 0
    table = np.array([sd.srgb_to_float[i] for i in range(256)])
    assert table[0] == 0.0 and table[255] == 1.0 and np.all(np.diff(table) > 0)
    np.testing.assert_allclose(table[128], ((128 / 255 + 0.055) / 1.055) ** 2.4, rtol=1e-6)
    bad = engine.material(scenes.substitute_textured(7))
    q = engine.prop(scenes.RECTANGLE, [bad])
    engine.transformation(q, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 3)))
    with pytest.raises(RuntimeError):
        engine.compile()


def test_metal_preset_is_a_metallic_substitute(engine):
    """{"metal_preset": name} -> Substitute with metallic 1 and colour = fresnel.conductor(eta, k, 1) of the metal's optical
    constants (material_provider.zig:277-280, fresnel.zig:45-57); unknown names take eta 1.5, k 1."""
    engine.camera(8, 8)
    gold = engine.material(json.dumps({"rendering": {"Substitute": {"metal_preset": "Gold", "roughness": 0.2}}}))
    other = engine.material(json.dumps({"rendering": {"Substitute": {"metal_preset": "Unobtainium"}}}))
    p = engine.prop(scenes.RECTANGLE, [gold])
    engine.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 2)))
    sd, _ = engine.compile()

    def conductor(eta, k):
        eta, k = np.float32(eta), np.float32(k)
        f = eta * eta + k * k
        a = np.float32(2.0) * eta
        one = np.float32(1.0)
        return np.float32(0.5) * ((f - a + one) / (f + a + one) + (f - a + one) / (f + a + one))

    g = sd.materials[gold]
    assert g.metallic == 1.0 and abs(g.roughness - 0.2) < 1e-7
    np.testing.assert_allclose(list(g.color_a)[:3], conductor([0.18267, 0.49447, 1.3761], [3.1178, 2.3515, 1.8324]), rtol=2e-6)
    assert g.color_a[0] > g.color_a[1] > g.color_a[2]  # gold reflects red most
    np.testing.assert_allclose(list(sd.materials[other].color_a)[:3], conductor([1.5] * 3, [1.0] * 3), rtol=2e-6)


def test_malformed_meshes_are_refused(product_lib):
    """su_triangle_mesh_create validates client data: an index beyond the vertex count, a part range beyond the index
    buffer, more than 65535 parts (part ids are u16) or an absurd material slot return -1 instead of reading out of bounds."""
    import ctypes

    import numpy as np
    lib = product_lib
    assert lib.su_init() == 0
    try:
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
        nrm = np.array([[0, 0, 1]] * 3, dtype=np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        up = ctypes.POINTER(ctypes.c_uint32)

        def create(indices, parts=None, num_parts=None):
            idx = np.asarray(indices, dtype=np.uint32)
            pr = None if parts is None else np.asarray(parts, dtype=np.uint32)
            n = (0 if pr is None else len(pr) // 3) if num_parts is None else num_parts
            return lib.su_triangle_mesh_create(0xFFFFFFFF, n, None if pr is None else pr.ctypes.data_as(up), len(idx) // 3,
                                               idx.ctypes.data_as(up), 3, pos.ctypes.data_as(fp), 3, nrm.ctypes.data_as(fp), 3,
                                               None, 0, None, 0, False)

        assert create([0, 1, 2]) >= 7
        assert create([0, 1, 2], [0, 3, 0]) >= 7
        assert create([0, 1, 3]) == -1             # index >= num_vertices
        assert create([0, 1, 2], [0, 6, 0]) == -1  # part range beyond the index buffer
        assert create([0, 1, 2], [3, 3, 0]) == -1
        assert create([0, 1, 2], [0, 3, 70000]) == -1  # material slot
        assert create([0, 1, 2], [0, 3, 0] * 70000) == -1  # 70000 parts
    finally:
        lib.su_release()


@pytest.mark.parametrize("material", [
    '{"rendering": {"Substitute": {"color": [0.5, 0.5, 0.5], "mask": {"Checker": {"scale": 4}}}}}',   # a procedural opacity mask
    '{"rendering": {"Substitute": {"roughness": {"file": "rough.png"}}}}',                        # image file
    '{"rendering": {"Light": {"emittance": {"spectrum": [1, 1, 1], "emission_map": {"id": 0, "sampler": {"texcoord": "Triplanar"}}}}}}',
    '{"rendering": {"Light": {"emittance": {"spectrum": [1, 1, 1], "emission_map": {"Noise": {}}}}}}',  # a procedural emission map
    '{"rendering": {"Substitute": {"roughness": {"Turbulence": {"levels": 3}}}}}',                # a procedural the reference does not have
    '{"rendering": {"Light": {"emittance": {"spectrum": [1, 1, 1], "profile": {"file": "lamp.ies"}}}}}',
])
def test_unsupported_texture_features_do_not_compile(product_lib, material):
    """A material description that asks for something outside the scope is accepted (su_material_create has no way to say
    no to part of a description) but a scene that uses it does not compile: nothing renders with a silent default."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 8, 8, 1, material="substitute")
        e.image(scenes.checker_image(4, np.float32))
        p = e.prop(scenes.RECTANGLE, [e.material(material)])
        e.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1, 0)))
        assert e.lib.su_ext_compile(0) == -1


def test_file_loader_options_on_an_image_id_are_ignored(product_lib):
    """"swizzle" / "invert" belong to the image-file loader; createTexture (material_provider.zig:700-736) does not look at them
    for an image id."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 8, 8, 1, material="substitute")
        one = e.image(scenes.weight_image(4))
        p = e.prop(scenes.RECTANGLE, [e.material('{"rendering": {"Substitute": {"roughness": {"id": %d, "swizzle": "Y", "invert": true}}}}' % one)])
        e.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1, 0)))
        assert e.lib.su_ext_compile(0) == 0


def test_unused_unsupported_material_is_harmless(product_lib):
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 8, 8, 1, material="substitute")
        e.material('{"rendering": {"Substitute": {"color": [0.5, 0.5, 0.5], "mask": {"Checker": {"scale": 4}}}}}')
        assert e.lib.su_ext_compile(0) == 0


@pytest.mark.parametrize("integrators", ['{"surface": {"PTMIS": {}}, "particle": {"depth": {"surface": 4}}}',
                                         '{"surface": {"PTMIS": {}}, "photon": {"num_photons": 1000}}',
                                         '{"surface": {"AOV": {"value": "AO"}}}'])
def test_other_integrators_are_refused(product_lib, integrators):
    """The light tracer, photon mapping and the AOV surface integrator (take.zig:131-237) are not the path in scope: the frame
    does not start, it is not rendered without them."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 8, 8, 1)
        e.integrators(integrators)
        assert e.lib.su_ext_compile(0) == -1
        e.integrators('{"surface": {"PTMIS": {}}, "photon": {"num_photons": 0}}')
        assert e.lib.su_ext_compile(0) == 0


def test_material_update_runs_the_kinds_parser(product_lib):
    """su_material_update (capi.zig:352-380 -> material_provider.zig:108-125): the update goes through the same parser as a new
    material, on top of the material as it is: anything a description can say can be updated, the rest stays."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 8, 8, 1, material="substitute")
        m = e.material(scenes.substitute_full((0.2, 0.3, 0.4), roughness=0.3, metallic=0.5))
        g = e.material(scenes.glass(1.5, (0.5, 0.6, 0.7), 2.0))
        for mat in (m, g):
            p = e.prop(scenes.RECTANGLE, [mat])
            e.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (0.1, 0.1, 1), (0, 1, 0)))
        lib = e.lib
        assert lib.su_material_update(m, b'{"coating": {"thickness": 0.05, "ior": 1.7}, "translucency": 0.25, "two_sided": true}') == 0
        assert lib.su_material_update(g, b'{"abbe": 30.0, "attenuation_distance": 4.0}') == 0
        e.compile(0)
        mats = e.lib.su_ext_scene_desc().contents.materials
        sub, gl = mats[m], mats[g]
        assert abs(sub.coating_thickness - 0.05) < 1e-7 and abs(sub.coating_ior - 1.7) < 1e-6 and abs(sub.translucency - 0.25) < 1e-7
        assert abs(sub.roughness - 0.3) < 1e-7 and abs(sub.metallic - 0.5) < 1e-7  # untouched
        assert sub.flags & 1 and sub.flags & 16  # two-sided, extended
        assert abs(gl.abbe - 30.0) < 1e-6
        # updateGlass' colour is a local that starts white again (material_provider.zig:173): the absorption is that of
        # ccoef.attenuationCoefficient(1, 4) = -log(0.991102) / 4
        assert np.allclose(list(gl.absorption), [-np.log(np.float32(0.991102)) / 4.0] * 3, rtol=1e-5)
        assert lib.su_material_update(m, b'{"two_sided": false}') == 0
        e.compile(0)
        assert not (e.lib.su_ext_scene_desc().contents.materials[m].flags & 1)


def test_sky_entity_compile_products(engine, oracle):
    """The Sky entity (sky/sky.zig, sky_material.zig; su_ext_sky_create): a Distant sun whose material reads its radiance from a
    table over the sun's disc, a Canopy with the baked image as its emission map.  Scene.compile places both from the sun's rotation
    (sky.zig:95-134); SkyMaterial.setSunRadiance's mean (sky_material.zig:66-87) is the sun's emission; the canopy's sampling
    distribution is the map's luminance without the Light material's MIS compensation (sky_material.zig:89-133, 170-215)."""
    sun_table, sky_image = scenes.sky_tables(size=32)
    ids = scenes.outdoor(engine, 32, 18, 1, material="substitute", sky_entity={"parameters": {"sun": {"rotation": [-55, 20, 0]}},
                                                                                "tables": (sun_table, sky_image)})
    sd, _ = engine.compile()
    sun = ids["sun"]
    canopy = [p for p in range(sd.num_props) if sd.props[p].shape == scenes.CANOPY][0]
    assert sd.props[sun].shape == scenes.DISTANT and sd.num_infinite_props == 2 and sd.num_lights == 2
    # json.createRotationMatrix: Rz * Rx * Ry of the angles in degrees; its third row is the direction the sun shines in
    a, b = np.radians(-55.0), np.radians(20.0)
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rot = rx @ ry
    t = sd.trafos[sun]
    np.testing.assert_allclose([list(t.r[k])[:3] for k in range(3)], rot, atol=2e-6)
    radius = np.tan(np.radians(0.2667))  # Sky.Radius (sky.zig:46)
    assert abs(t.r[0][3] - radius) < 1e-9 and abs(t.r[1][3] - radius) < 1e-9
    # the canopy: X(90 degrees) * Y(azimuth), azimuth = -atan2(sun.x, sun.z): its z axis points up
    tc = sd.trafos[canopy]
    np.testing.assert_allclose(list(tc.r[2])[:3], [0, 1, 0], atol=2e-6)
    # the sun's table and its weighted mean over the part of the disc above the horizon
    assert sd.num_sun_radiance == 1024
    table = np.ctypeslib.as_array(sd.sun_radiance, shape=(1024, 4))
    assert np.array_equal(table[:, :3], sun_table) and not table[:, 3].any()
    elevation = -np.arcsin(rot[2, 1])
    v = (np.arange(1023) + 0.5) / 1024.0
    wz = np.sin(elevation + (2 * v - 1) * radius)
    wgt = np.sin(v)
    mean = ((sun_table[:-1].astype(np.float64) + sun_table[1:]) * (0.5 * wgt * (wz >= 0))[:, None]).sum(0) / wgt.sum()
    mat = sd.materials[sd.material_ids[sd.props[sun].parts_start]]
    assert mat.kind == 4 and (mat.flags & 8)  # ZYGPU_MAT_SKY, emissive
    np.testing.assert_allclose(list(mat.emission), mean, rtol=2e-5)
    # the canopy's distribution: rows of hmax3(uv_weight * texel), no compensation
    il = sd.image_lights[0]
    n = 32
    g = (np.arange(n) + 0.5) / n * 2 - 1
    inside = (g[None, :] ** 2 + g[:, None] ** 2) <= 1.0
    lum = sky_image.max(axis=2) * inside
    rows = np.ctypeslib.as_array(il.rows, shape=(n + 1, 4))
    cdf = np.ctypeslib.as_array(il.cdf, shape=(il.num_cdf,))
    for y in (3, 11, 16):
        assert rows[y, 1] == n + 1
        expect = np.concatenate([[0.0], np.cumsum(lum[y] / lum[y].sum())])
        np.testing.assert_allclose(cdf[rows[y, 0]:rows[y, 0] + n + 1], expect, atol=3e-6)
    # and the oracle renders it: the sky is seen, the sun lights the ground
    from util import oracle_render, oracle_resolve
    img = oracle_resolve(oracle, engine.lib.su_ext_view_desc().contents, oracle_render(oracle, sd, engine.lib.su_ext_view_desc().contents, 4, threads=4))
    assert np.isfinite(img).all() and img[2, :, 2].mean() > 0.3 and img[-2, :, :3].mean() > 0.05


def test_sky_entity_without_tables_gets_the_stand_in_and_says_so(product_lib):
    messages = []
    callback = zyg_amd.LOG_CALLBACK(lambda level, text: messages.append((level, text.decode())))
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_register_log.argtypes = [zyg_amd.LOG_CALLBACK]
        e.lib.su_register_log(callback)
        try:
            scenes.outdoor(e, 16, 9, 1, sky_entity={"parameters": {"sun": {"rotation": [-30, 0, 0]}}})
            sd, _ = e.compile()
            assert sd.num_sun_radiance == 1024 and sd.images[sd.num_images - 1].width == 1024
            table = np.ctypeslib.as_array(sd.sun_radiance, shape=(1024, 4))
            assert table[:, :3].min() >= 0 and table[512, 0] > 1e3
        finally:
            e.lib.su_register_log(zyg_amd.LOG_CALLBACK())
    assert any("stand-in" in text for _, text in messages), messages
    # under the horizon: the sun's radius is zero (sky.zig:102-105)
    with zyg_amd.Engine(product_lib) as e:
        ids = scenes.outdoor(e, 16, 9, 1, sky_entity={"parameters": {"sun": {"rotation": [20, 0, 0]}}})
        sd, _ = e.compile()
        assert sd.trafos[ids["sun"]].r[0][3] == 0.0


def test_portal_light_compile_products(engine, oracle):
    """A Portal (scene/shape/portal.zig, light_material.zig:142-199, summed_area_table.zig, scene_loader.zig:371-397): a Rectangle in
    the ceiling's place takes the Dome's material; the Dome's light becomes a hidden prototype (not in the light tree, power 0); the
    portal's sampler tabulates the map's brightest channel per texel of the portal's rectified image, times the mapping's weight."""
    sky = scenes.sky_image(32, 16)
    ids = scenes.cornell(engine, 32, 18, 1, portal_map=sky)
    sd, vd = engine.compile()
    assert sd.num_lights == 2 and sd.num_portal_lights == 1 and sd.light_tree.num_lights == 1
    dome, portal = sd.lights[0], sd.lights[1]
    assert dome.prototype == 1 and dome.light_class == 2 and sd.props[dome.prop].shape == scenes.DOME
    assert not (sd.props[dome.prop].flags & 3)  # hidden from the camera and from reflections
    assert portal.prototype == 0 and portal.light_class == 0 and sd.props[portal.prop].shape == scenes.RECTANGLE
    assert sd.light_aabbs[0].min[3] == 0.0 and sd.light_aabbs[1].min[3] > 0.0  # the lights' powers
    assert sd.material_ids[sd.props[portal.prop].parts_start] == sd.material_ids[sd.props[dome.prop].parts_start]
    pl = sd.portal_lights[0]
    dim = pl.dim
    assert dim == 16 and pl.light_link == 0
    func = np.ctypeslib.as_array(pl.func, shape=(dim, dim)).copy()
    sat = np.ctypeslib.as_array(pl.sat, shape=(dim + 1, dim + 1)).copy()
    assert not sat[0].any() and not sat[:, 0].any()
    np.testing.assert_allclose(sat[1:, 1:], func.astype(np.float64).cumsum(0).cumsum(1), rtol=2e-5)
    # one texel restated: Portal.imageToWorld of the 16 subsamples, the direction through the Dome's lat-long map (identity rotation),
    # nearest texel, brightest channel, times the weight.  The portal's axes: r0 = x, r1 = z, r2 = -y (it looks down).
    rot = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float64)
    for (px, py) in ((8, 8), (3, 12)):
        acc = 0.0
        for yy in range(4):
            v = (py + 0.125 + yy / 4.0) / dim
            for xx in range(4):
                u = (px + 0.125 + xx / 4.0) / dim
                w = np.array([np.tan(-np.pi / 2 + u * np.pi), np.tan(-np.pi / 2 + v * np.pi), 1.0])
                w /= np.linalg.norm(w)
                weight = np.pi ** 2 * (1 - w[0] ** 2) * (1 - w[1] ** 2) / w[2]
                d = -(w @ rot)
                du = np.arctan2(d[0], d[2]) / (2 * np.pi) + 0.5
                dv = np.arccos(d[1]) / np.pi
                tx, ty = min(int(du * 32), 31), min(int(dv * 16), 15)
                acc += sky[ty, tx].max() * weight
        assert abs(func[py, px] - acc / 16.0) <= 2e-4 * max(1.0, acc / 16.0), (px, py, func[py, px], acc / 16.0)
    # the oracle renders the room lit through the window
    from util import oracle_render, oracle_resolve
    img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, 8, threads=4))
    assert np.isfinite(img).all() and img[12:, 8:24, :3].mean() > 0.02


def _aperture_image(n=16):
    """A one-channel aperture shape: a triangle of weights (not symmetric, so a flipped or transposed read shows), a few zero texels."""
    y, x = np.mgrid[0:n, 0:n]
    img = np.where(x <= y, 40 + 13 * ((x * 3 + y * 5) % 16), 0).astype(np.uint8)
    img[n - 1, 0] = 255
    return img


def test_aperture_shape_distribution(product_lib, tmp_path):
    """"aperture": {"shape": file} (camera_perspective.zig:270-283 -> aperture.zig:22-44): the file is read as a one-channel (Opacity)
    texture, one Distribution1D per row over its values, the marginal over the rows' integrals; the view names the distribution, which
    rides with the scene's image lights.  Checked against a numpy restatement of Distribution1D.configure (distribution_1d.zig:104-171)."""
    from util import write_png
    img = _aperture_image()
    write_png(tmp_path / "iris.png", img[..., None], [0, 2])
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_mount(str(tmp_path).encode())
        scenes.cornell(e, 32, 24, 1, lens={"aperture": {"radius": 0.1, "shape": "iris.png"}, "focus": {"distance": 3.0}})
        sd, vd = e.compile()
        assert vd.aperture_shape == sd.num_image_lights - 1 and vd.lens_radius == np.float32(0.1)
        il = sd.image_lights[vd.aperture_shape]
        n = img.shape[0]
        assert il.num_rows == n and il.total_weight == 1.0
        rows = np.ctypeslib.as_array(il.rows, shape=((n + 1) * 4,)).reshape(n + 1, 4)
        cdf = np.ctypeslib.as_array(il.cdf, shape=(il.num_cdf,))
        integrals = []
        for y in range(n + 1):
            data = (img[y].astype(np.float32) * np.float32(1.0 / 255.0)) if y < n else np.array(integrals, dtype=np.float32)
            integral = np.float32(0)
            for d in data:
                integral = np.float32(integral + d)
            integrals.append(integral)
            offset, size = int(rows[y, 0]), int(rows[y, 1])
            if integral == 0:
                assert size == 2 and cdf[offset] == 1.0 and cdf[offset + 1] == 1.0
                continue
            assert size == len(data) + 1
            ii = np.float32(1.0) / integral
            acc, want = np.float64(0), [np.float32(0)]
            for d in data[:-1]:  # p = @mulAdd(data[i], ii, p): one rounding
                acc = np.float64(np.float32(np.float64(d) * np.float64(ii) + np.float64(want[-1])))
                want.append(np.float32(acc))
            want.append(np.float32(1))
            assert np.array_equal(cdf[offset:offset + size], np.array(want, dtype=np.float32)), y


def _shaper_aperture(blades, roundness, dim=128):
    """rendering/shaper.zig:156-301 drawAperture(color 1, centre (0.5, 0.5), blades, radius 0.5, roundness, rotation pi) + resolve(Byte1),
    in numpy float32 (a second implementation: vectorised over the 4 x 4 sub-samples of every pixel)."""
    f = np.float32
    pi = f(np.pi)
    delta = f(f(2.0) * pi) / f(blades)
    ss = f(1.0) / f(4)
    so, step = f(0.5) * ss, ss / f(dim)
    base = (np.arange(dim, dtype=np.float32) + so) / f(dim)
    coords = np.stack([base + f(k) * step for k in range(4)], axis=1)  # [pixel][sub-sample]; the reference adds the step k times
    for k in range(1, 4):  # u += ssx, repeatedly (not k * ssx)
        coords[:, k] = coords[:, k - 1] + step
    u = coords[None, :, None, :]  # [y][x][sy][sx]
    v = coords[:, None, :, None]
    cx, cy = (u - f(0.5)) + f(0) * v, (v - f(0.5)) + f(0) * u
    lc = np.sqrt(cx * cx + cy * cy)
    mt = np.zeros_like(lc)
    bx, by = np.sin(pi), np.cos(pi)
    for i in range(blades):
        angle = f(i + 1) * delta + pi
        ccx, ccy = np.sin(angle), np.cos(angle)
        ex, ey = f(ccx - bx), f(ccy - by)
        length = np.sqrt(f(ey * ey + ex * ex))
        nx, ny = f(ey / length), f(-ex / length)
        d = f(nx * bx + ny * by)
        mt = np.maximum(mt, (nx * cx + ny * cy) / d)
        bx, by = ccx, ccy
    inside = (lc <= f(0.5)) & ((f(1.0) - f(roundness)) * mt + f(roundness) * lc <= f(0.5))
    w = inside.sum(axis=(2, 3)).astype(np.float32) * f(1.0 / 16.0)
    return np.floor(np.minimum(w, f(1.0)) * f(255.0) + f(0.5)).astype(np.uint8)


@pytest.mark.parametrize("blades,roundness", [(5, 0.0), (6, 0.35), (8, 1.0), (11, 0.2)])
def test_aperture_blades_are_drawn_like_the_shaper(product_lib, blades, roundness):
    """camera_perspective.zig:284-300: `blades` > 3 draws a regular polygon (rendering/shaper.zig) into a 128 x 128 one-channel image,
    which becomes the aperture's shape (its Distribution2D rides with the image lights, like a shape given as a file)."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 32, 24, 1, lens={"aperture": {"radius": 0.05, "blades": blades, "roundness": roundness}, "focus": {"distance": 3.0}})
        sd, vd = e.compile()
        assert vd.aperture_shape != zyg_amd.NULL and vd.lens_radius == np.float32(0.05)
        img = sd.images[sd.num_images - 1]
        assert (img.width, img.height) == (128, 128)
        got = np.ctypeslib.as_array(ctypes.cast(img.pixels, ctypes.POINTER(ctypes.c_uint8)), shape=(128, 128)).copy()
    want = _shaper_aperture(blades, roundness)
    # sin / cos of the host's libm against numpy's: a sub-sample exactly on an edge may fall the other way
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 16 and (got != want).mean() < 0.01
    assert got[64, 64] == 255 and got[0, 0] == 0 and got[127, 127] == 0 and got[10, 10] == 0
    area = got.astype(np.float64).sum() / 255.0 / (128 * 128)
    polygon = 0.5 * blades * 0.25 * np.sin(2 * np.pi / blades)  # a regular polygon inscribed in the circle of radius 0.5
    circle = np.pi * 0.25
    assert polygon - 0.004 <= area <= circle + 0.004
    if roundness == 0.0:
        assert abs(area - polygon) < 0.004
    if roundness == 1.0:
        assert abs(area - circle) < 0.004


@pytest.mark.parametrize("parameters,word", [({"aperture": {"radius": 0.1, "blades": 65}}, "blades"), ({"stereo": {"ipd": 0.062}}, "stereo"),
                                             ({"aperture": {"radius": 0.1, "shape": "missing.png"}}, "missing.png")])
def test_camera_parameters_refused_by_name(product_lib, parameters, word):
    """camera_perspective.zig:284-303: a stereo camera is not provided, and a shape file that cannot be read (or an absurd number of
    blades) is no aperture: su_ext_camera_set_parameters fails and the log names the key; a frame does not start with the parameter
    silently ignored."""
    lines = []
    callback = zyg_amd.LOG_CALLBACK(lambda level, text: lines.append((level, text.decode())))
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_register_log.argtypes = [zyg_amd.LOG_CALLBACK]
        e.lib.su_register_log(callback)
        try:
            scenes.cornell(e, 32, 24, 1)
            assert e.lib.su_ext_camera_set_parameters(json.dumps(parameters).encode()) == -1
            assert any(level == 2 and word in text for level, text in lines), lines
            assert e.lib.su_ext_compile(0) != 0  # and the engine stays refused until the parameters are set right
        finally:
            e.lib.su_register_log(zyg_amd.LOG_CALLBACK())


def test_focus_point_is_read(product_lib):
    """"focus": {"point": [x, y, z]} (camera_perspective.zig:348-361): kept for the probe ray su_start_frame traces (the -m gpu tests
    check the distance); without a lens radius nothing is probed and the pinhole view is what it was."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 32, 24, 1, lens={"focus": {"point": [0.25, 0.75, 0.5], "distance": 2.5}})
        _, vd = e.compile()
        assert vd.lens_radius == 0.0 and vd.focus_distance == np.float32(2.5) and vd.aperture_shape == zyg_amd.NULL


def test_coincident_dielectric_surfaces_are_found(product_lib):
    """zygpu_scene_has_coincident_dielectric (what option pool_relaxed -1 decides with): a glass box standing ON the floor
    (its bottom face in the floor Rectangle's plane), a glass box against the other box (mesh against mesh) are found; the same boxes
    1 mm apart, a scene without Glass, a glass Sphere resting on the floor (a point of contact) are not."""
    from zyg_amd import scenes
    lib = product_lib
    lib.zygpu_scene_has_coincident_dielectric.argtypes = [ctypes.POINTER(zyg_amd.SceneDesc)]
    glass = {"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)}

    def found(**kw):
        with zyg_amd.Engine(lib) as e:
            scenes.cornell(e, 32, 18, 1, **kw)
            sd, _ = e.compile()
            return lib.zygpu_scene_has_coincident_dielectric(ctypes.byref(sd))

    assert 1 == found(glass_objects=glass)
    assert 0 == found(glass_objects=glass, box_lift=1e-3)
    assert 0 == found()
    assert 0 == found(glass_objects={"sphere": scenes.glass(1.5)}, box_lift=1e-3)
    with zyg_amd.Engine(lib) as e:  # the outdoor scene's glass box on the ground Rectangle
        scenes.outdoor(e, 32, 18, 1, integrator="PTMIS", material="substitute", showcase=True)
        sd, _ = e.compile()
        assert 1 == lib.zygpu_scene_has_coincident_dielectric(ctypes.byref(sd))
    with zyg_amd.Engine(lib) as e:
        scenes.outdoor(e, 32, 18, 1, integrator="PTMIS", material="substitute", showcase=True, box_lift=1e-3)
        sd, _ = e.compile()
        assert 0 == lib.zygpu_scene_has_coincident_dielectric(ctypes.byref(sd))


def test_device_hash_follows_the_kernel_units_flags(monkeypatch):
    """bench.py quotes a committed rocprofv3 profile only for the library it was taken with (build.device_hash): the hash covers the
    kernel sources and the flags the kernel units are compiled with, and leaves the host side alone."""
    import zyg_amd.build as zbuild
    mine = zbuild.device_hash()
    assert mine == zbuild.device_hash()
    monkeypatch.setattr(zbuild, "RELAXED", zbuild.RELAXED + ["-DZYG_SOME_OTHER_BUILD=1"])
    assert zbuild.device_hash() != mine
    monkeypatch.undo()
    monkeypatch.setenv("ZYG_SCHED_DEFAULT", "1")  # (the A/B's other side is another library)
    assert zbuild.device_hash() != mine
    monkeypatch.undo()
    assert zbuild.device_hash() == mine
