"""Ray-level parity of the device's three ways through a scene against the CPU oracle's walk (zyg_oracle_intersect /
zyg_oracle_visibility): the reference's own traversal (top-level tree lane by lane, BVH2), the wave-wide walk of the prop tables
(dscene.hpp sceneIntersectUniform), and the same with the four-wide re-layout of the meshes' trees (bvh4.hpp).  All in
precise arithmetic, so (t, u, v, primitive, prop) must be the oracle's bits ray for ray; only the four-wide walk may resolve an
exact tie between two triangles (a ray through a shared edge, bit-equal t) the other way round, which is counted and bounded.

Also: the kernels built on these walks give the same film as the ones they replace (options uniform_tlas, persistent)."""
import ctypes

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    """origins inside the Cornell box, directions uniform on the sphere; a few axis-parallel and grazing ones"""
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(-0.95, 0.95, n), rng.uniform(0.05, 1.95, n), rng.uniform(-0.95, 0.95, n)], axis=1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:64] = np.eye(3)[rng.integers(0, 3, 64)] * rng.choice([-1.0, 1.0], (64, 1))
    d[64:128, 1] *= 1e-4
    d[64:128] /= np.linalg.norm(d[64:128], axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = o
    rays[:, 4:7] = d
    rays[:, 7] = 2.14748313e+09
    rays[n // 2:, 7] = rng.uniform(0.2, 3.0, n - n // 2)  # shadow-ray-like finite extents
    return rays


def _camera_relative(engine, rays):
    """the compiled scene is relative to the camera position (space.zig:99-108)"""
    out = rays.copy()
    out[:, 0:3] -= np.array([0.0, 1.0, -3.9], dtype=np.float32)
    return out


def _both(engine, oracle, rays, depth=1, precise=1):
    lib, ctx = engine.lib, engine.context()
    n = rays.shape[0]
    sd = lib.su_ext_scene_desc().contents
    ref_hits = np.zeros((n, 8), dtype=np.uint32)
    ref_vis = np.zeros(n, dtype=np.uint32)
    oracle.zyg_oracle_intersect.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    oracle.zyg_oracle_visibility.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    assert oracle.zyg_oracle_intersect(ctypes.addressof(sd), n, rays.ctypes.data, depth, ref_hits.ctypes.data) == 0
    assert oracle.zyg_oracle_visibility(ctypes.addressof(sd), n, rays.ctypes.data, ref_vis.ctypes.data) == 0
    lib.zygpu_test_intersect.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    lib.zygpu_test_visibility.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    out = {}
    engine.set_option("precise", precise)  # the hooks run in the kernel unit of the context's arithmetic mode
    for mode in (0, 1, 2, 3):
        hits = np.zeros((n, 8), dtype=np.uint32)
        vis = np.zeros(n, dtype=np.uint32)
        assert lib.zygpu_test_intersect(ctx, n, rays.ctypes.data, depth, mode, hits.ctypes.data) == 0, lib.zygpu_last_error()
        assert lib.zygpu_test_visibility(ctx, n, rays.ctypes.data, mode, vis.ctypes.data) == 0
        out[mode] = (hits, vis)
    return ref_hits, ref_vis, out


@pytest.mark.parametrize("scene", ["boxes", "mesh"])
def test_ray_dump_against_the_oracle(engine, oracle, scene):
    n = 1 << 18
    if scene == "boxes":
        scenes.cornell(engine, 64, 48, 1)
    else:
        scenes.cornell(engine, 64, 48, 1, boxes=False, mesh=scenes.heightfield_mesh(n=400), mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
    engine.start_frame(0)
    rays = _camera_relative(engine, _rays(n, 11))
    ref_hits, ref_vis, out = _both(engine, oracle, rays)
    assert (ref_hits[:, 4] != zyg_amd.NULL).mean() > 0.5 and 0.1 < ref_vis.mean() < 0.99
    if scene == "mesh":
        sd = engine.lib.su_ext_scene_desc().contents
        mesh_props = [p for p in range(sd.num_props) if sd.props[p].shape == 7]
        assert np.isin(ref_hits[:, 4], mesh_props).mean() > 0.03  # the mesh is really hit (it lies low in the box)
    for mode in (0, 1):  # the reference's walk and the prop tables: the oracle's bits, every ray
        hits, vis = out[mode]
        assert np.array_equal(hits, ref_hits), (mode, int(np.any(hits != ref_hits, axis=1).sum()))
        assert np.array_equal(vis, ref_vis), mode
    # four-wide BLAS: same boolean, same hit except exact ties between triangles
    hits, vis = out[2]
    assert np.array_equal(vis, ref_vis)
    bad = np.any(hits != ref_hits, axis=1)
    assert bad.mean() <= 1e-5, int(bad.sum())
    # where it differs the distance is the same: it is the other triangle of a shared edge
    assert np.array_equal(hits[bad, 0], ref_hits[bad, 0])
    # the relaxed units' walk in this unit's arithmetic: the small meshes are tested in their unit-triangle form (dscene.hpp
    # smallTriangleIntersect), which rounds differently from the reference's test: bounded like the relaxed units' dump
    _bounded(out[3], ref_hits, ref_vis, 3)


def _bounded(result, ref_hits, ref_vis, mode):
    hits, vis = result
    assert (vis != ref_vis).mean() <= 2e-4, (mode, int((vis != ref_vis).sum()))
    # another primitive or prop: where two surfaces coincide (the Cornell boxes stand on the floor: a ray from inside a box hits
    # its bottom face and the floor at the same distance) an ulp decides which one is reported; never a hit against a miss,
    # and the distance is the same
    # (mode 3 tests the Rectangles and the small meshes in their unit frames: the floor and a box's bottom face no longer round alike)
    other = (hits[:, 3] != ref_hits[:, 3]) | (hits[:, 4] != ref_hits[:, 4])
    assert other.mean() <= (5e-3 if 3 == mode else 2e-3), (mode, int(other.sum()))
    assert np.array_equal(hits[:, 4] == zyg_amd.NULL, ref_hits[:, 4] == zyg_amd.NULL) or \
        ((hits[:, 4] == zyg_amd.NULL) != (ref_hits[:, 4] == zyg_amd.NULL)).mean() <= 1e-5, mode
    hit_both = (hits[:, 4] != zyg_amd.NULL) & (ref_hits[:, 4] != zyg_amd.NULL)
    t, rt = hits[hit_both, 0].view(np.float32), ref_hits[hit_both, 0].view(np.float32)
    assert np.all(np.abs(t - rt) <= 1e-4 * np.maximum(1.0, np.abs(rt))), mode
    same = ~other & hit_both
    uv, ruv = hits[same, 1:3].view(np.float32), ref_hits[same, 1:3].view(np.float32)
    assert np.all(np.abs(uv - ruv) <= 2e-3)


@pytest.mark.parametrize("scene", ["boxes", "boxes18", "boxes48", "boxes108", "mesh", "sphere"])
def test_relaxed_ray_dump_against_the_oracle(engine, oracle, scene):
    """The same rays through the walks as the relaxed kernel units compile them (native divide, contraction: the hooks of
    kernels.hip built with -DZYG_FAST): an occlusion answer or a hit may flip only where a ray grazes an edge or a box within
    an ulp, so the differing rays are bounded and, where both sides hit the same primitive, t / u / v agree to float accuracy."""
    n = 1 << 18
    if scene == "boxes":
        scenes.cornell(engine, 64, 48, 1)
    elif scene == "boxes18":  # one face of each box as a 1 x 4 grid: 18 triangles, 9 items of the small-mesh table
        scenes.cornell(engine, 64, 48, 1, box_first_face=(1, 4))
    elif scene == "boxes48":  # 48 triangles that pair up into 24 items: still in the table (its limit is 32 items)
        scenes.cornell(engine, 64, 48, 1, box_subdivide=2)
    elif scene == "boxes108":  # 54 items: past the table, through the four-wide BLAS
        scenes.cornell(engine, 64, 48, 1, box_subdivide=3)
    elif scene == "mesh":
        scenes.cornell(engine, 64, 48, 1, boxes=False, mesh=scenes.heightfield_mesh(n=400), mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
    else:
        scenes.cornell(engine, 64, 48, 1, boxes=False, mesh=scenes.displaced_sphere_mesh(n=400), mesh_position=scenes.MESH_CENTRE,
                       mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
    engine.start_frame(0)
    rays = _camera_relative(engine, _rays(n, 17))
    ref_hits, ref_vis, out = _both(engine, oracle, rays, precise=0)
    if scene == "sphere":
        sd = engine.lib.su_ext_scene_desc().contents
        mesh_props = [p for p in range(sd.num_props) if sd.props[p].shape == 7]
        assert np.isin(ref_hits[:, 4], mesh_props).mean() > 0.15  # the sphere hangs in the middle of the box
    for mode in (0, 1, 2, 3):
        _bounded(out[mode], ref_hits, ref_vis, mode)


def _film(product_lib, build, w, h, spp, **options):
    with zyg_amd.Engine(product_lib) as e:
        build(e)
        e.start_frame(0)
        for k, v in options.items():
            e.set_option(k, v)
        e.render_iterations(spp)
        e.synchronize()
        return e.download_film(w, h)


@pytest.mark.parametrize("scene", ["lambert", "substitute", "mesh", "three_lights"])
def test_prop_table_and_persistent_kernels_give_the_same_film(product_lib, variant_lib_or_none, scene):
    """precise arithmetic: the wave-wide walk of the prop tables (default) == the per-lane tree walk (uniform_tlas = 0) ==
    the persistent kernel with the path state in LDS (persistent = 1: variant builds, conftest.py), bit for bit; with regeneration
    (more samples than resident lanes) for the persistent one."""
    w, h, spp = 320, 180, 4
    build = {
        "lambert": lambda e: scenes.cornell(e, w, h, spp),
        "substitute": lambda e: scenes.cornell(e, w, h, spp, material="substitute", integrator="PTDL"),
        "mesh": lambda e: scenes.cornell(e, w, h, spp, boxes=False, mesh=scenes.heightfield_mesh(n=96), mesh_material=scenes.substitute((0.5, 0.5, 0.5))),
        "three_lights": lambda e: scenes.cornell(e, w, h, spp, many_lights=2),
    }[scene]
    base = _film(product_lib, build, w, h, spp, precise=1, uniform_tlas=0)
    assert np.isfinite(base).all() and base[..., :3].max() > 0
    assert np.array_equal(base, _film(product_lib, build, w, h, spp, precise=1, uniform_tlas=1))
    if variant_lib_or_none is None:
        return
    assert np.array_equal(base, _film(variant_lib_or_none, build, w, h, spp, precise=1, persistent=1))
    # relaxed arithmetic: the same samples, but the compiler is free to fuse multiply-adds differently in the two kernel
    # families, so the films agree closely, not bit for bit
    r0 = _film(variant_lib_or_none, build, w, h, spp, precise=0)
    r1 = _film(variant_lib_or_none, build, w, h, spp, precise=0, persistent=1)
    assert np.isfinite(r1).all() and np.array_equal(r0[..., 3], r1[..., 3])
    close = np.isclose(r0, r1, rtol=1e-3, atol=1e-4).all(axis=-1)
    assert close.mean() > 0.98, close.mean()


@pytest.mark.parametrize("material", ["lambert", "substitute"])
def test_fused_tick_gives_the_same_film(variant_lib, material):
    """Option fuse_tick (wf_tick: wf_scatter of vertex k and wf_trace_shade of vertex k + 1 in one launch) on a scene with a mesh
    above the wide-BLAS threshold (4096 triangles): in the relaxed units one block runs the any-hit and the closest-hit walk of the
    four-wide nodes back to back without a barrier; the two walks' parked children must not share LDS (kernels.hip: unit 5 keeps
    them in scratch).  Precise: bit for bit the film of the separate launches; relaxed: the same samples, close (another kernel
    family, other contractions), and deterministic."""
    w, h, spp = 480, 270, 4
    build = lambda e: scenes.cornell(e, w, h, spp, material=material, boxes=False, mesh=scenes.displaced_sphere_mesh(n=96),
                                     mesh_position=scenes.MESH_CENTRE)
    base = _film(variant_lib, build, w, h, spp, precise=1)
    assert np.array_equal(base, _film(variant_lib, build, w, h, spp, precise=1, fuse_tick=1))
    r0 = _film(variant_lib, build, w, h, spp, precise=0)
    r1 = _film(variant_lib, build, w, h, spp, precise=0, fuse_tick=1)
    assert np.isfinite(r1).all() and np.array_equal(r0[..., 3], r1[..., 3])
    assert np.array_equal(r1, _film(variant_lib, build, w, h, spp, precise=0, fuse_tick=1))
    close = np.isclose(r0, r1, rtol=1e-3, atol=1e-4).all(axis=-1)
    assert close.mean() > 0.98, close.mean()


@pytest.mark.parametrize("scene", ["sphere_lambert", "sphere_substitute", "mound", "four_lights"])
def test_mesh_kernels_give_the_film_of_the_in_place_kernel(product_lib, variant_lib_or_none, scene):
    """Option mesh_kernels (default on; relaxed units, scenes of the prop tables with a mesh above the wide-BLAS threshold):
    wf_scatter_h suspends the vertices whose shadow ray has to walk the large mesh, wf_shadow_u (persistent, its lanes fetch their
    own rays) walks it, the resume launches take the vertices up again; wf_mesh_closest (option mesh_closest) does the same for the
    closest hits.  The same samples as wf_scatter's with the walk in place: the same film up to the contractions of another kernel
    family, deterministic, and with four lights (up to four suspensions per vertex) too."""
    w, h, spp = 480, 270, 4
    mesh = scenes.heightfield_mesh(n=96) if "mound" == scene else scenes.displaced_sphere_mesh(n=96)
    kw = dict(boxes=False, mesh=mesh, material="substitute" if "substitute" in scene else "lambert")
    if "mound" != scene:
        kw["mesh_position"] = scenes.MESH_CENTRE
    if "four_lights" == scene:
        kw["many_lights"] = 2
    build = lambda e: scenes.cornell(e, w, h, spp, **kw)
    base = _film(product_lib, build, w, h, spp, precise=0, mesh_kernels=0)
    assert np.isfinite(base).all() and base[..., :3].max() > 0
    # (bvh8 / mesh_closest8, both default on: the persistent walkers over the eight-wide compressed trees, bvh8.hpp)
    # the four-wide closest-hit pass (mesh_closest) and the eight-wide walkers without theirs (mesh_closest8 = 0): variant builds
    cases = [(product_lib, {"mesh_kernels": 1, "bvh8": 0}), (product_lib, {"mesh_kernels": 1})]
    if variant_lib_or_none is not None:
        cases += [(variant_lib_or_none, {"mesh_kernels": 1, "bvh8": 0, "mesh_closest": 1}), (variant_lib_or_none, {"mesh_kernels": 1, "mesh_closest8": 0})]
    for lib, options in cases:
        film = _film(lib, build, w, h, spp, precise=0, **options)
        assert np.isfinite(film).all() and np.array_equal(base[..., 3], film[..., 3])
        again = _film(lib, build, w, h, spp, precise=0, **options)
        assert np.array_equal(film, again), (options, int((film != again).any(axis=-1).sum()))
        close = np.isclose(base, film, rtol=1e-3, atol=1e-4).all(axis=-1)
        assert close.mean() > 0.98, (options, close.mean())
        assert abs(float(film[..., :3].mean()) - float(base[..., :3].mean())) <= 2e-4 * float(base[..., :3].mean())


def _patchwork_mesh(seed=5):
    """A small mesh (27 triangles) that exercises the small-triangle table's items (dscene.hpp SmallItem): ten parallelograms in
    random planes, split along either diagonal, every triangle's vertex order turned and flipped at random and the triangles
    shuffled (so that the halves of a pair are apart, in either order, and the pair's frame starts at any corner), five lone
    triangles, and a quad whose fourth corner is off the parallelogram's (its halves must stay two items)."""
    rng = np.random.default_rng(seed)
    pos, tris = [], []

    def corner(p):
        pos.append(np.asarray(p, dtype=np.float32))
        return len(pos) - 1

    def triangle(i, j, k):
        t = [i, j, k]
        if rng.integers(0, 2):
            t = [t[0], t[2], t[1]]
        r = int(rng.integers(0, 3))
        tris.append(t[r:] + t[:r])

    for q in range(11):
        a = rng.uniform(-0.6, 0.6, 3) + np.array([0.0, 1.0, 0.0])
        e1, e2 = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.5, 0.5, 3)
        # corners on a 1/64 grid: b + c - a is exact in f32, as it is for a modelling tool's boxes
        a, e1, e2 = (np.round(x * 64) / 64 for x in (a, e1, e2))
        d = a + e1 + e2 + (np.array([0.0, 0.125, 0.0]) if 10 == q else 0.0)
        i, j, k, m = corner(a), corner(a + e1), corner(d), corner(a + e2)
        if rng.integers(0, 2):
            triangle(i, j, k), triangle(i, k, m)
        else:
            triangle(i, j, m), triangle(j, k, m)
    for _ in range(5):
        a = rng.uniform(-0.6, 0.6, 3) + np.array([0.0, 1.0, 0.0])
        triangle(corner(a), corner(a + rng.uniform(-0.5, 0.5, 3)), corner(a + rng.uniform(-0.5, 0.5, 3)))
    order = rng.permutation(len(tris))
    idx = np.asarray([tris[i] for i in order], dtype=np.uint32)
    pos = np.asarray(pos, dtype=np.float32)
    nrm = np.tile(np.array([[0, 1, 0]], dtype=np.float32), (pos.shape[0], 1))
    return idx.ravel(), pos, nrm, np.zeros((pos.shape[0], 2), dtype=np.float32)


@pytest.mark.parametrize("precise", [1, 0])
def test_small_mesh_items_against_the_oracle(engine, oracle, precise, variant_lib_or_none):
    """The relaxed units' test of the small meshes (unit-triangle frames, parallelogram halves as one item: dscene.hpp
    smallItemIntersect) reports the reference's primitive and barycentric coordinates whichever corner an item's frame starts at;
    option pair_small = 0 (every triangle its own item) gives the same answers."""
    n = 1 << 18
    mesh = _patchwork_mesh()
    assert mesh[0].size // 3 == 27
    scenes.cornell(engine, 64, 48, 1, boxes=False, mesh=mesh, mesh_position=(0, 0, 0))
    engine.start_frame(0)
    rays = _camera_relative(engine, _rays(n, 23))
    ref_hits, ref_vis, out = _both(engine, oracle, rays, precise=precise)
    sd = engine.lib.su_ext_scene_desc().contents
    mesh_props = [p for p in range(sd.num_props) if sd.props[p].shape == 7]
    on_mesh = np.isin(ref_hits[:, 4], mesh_props)
    assert on_mesh.mean() > 0.1 and len(np.unique(ref_hits[on_mesh, 3])) == 27  # every triangle is hit
    _bounded(out[3], ref_hits, ref_vis, 3)
    paired = out[3]
    if variant_lib_or_none is None:  # (option pair_small: variant builds)
        return
    engine.release()
    engine.__init__(variant_lib_or_none)
    scenes.cornell(engine, 64, 48, 1, boxes=False, mesh=mesh, mesh_position=(0, 0, 0))
    engine.start_frame(0)
    engine.set_option("pair_small", 0)
    engine.start_frame(0)  # the table is built when the scene is uploaded
    _, _, out = _both(engine, oracle, rays, precise=precise)
    engine.set_option("pair_small", 1)
    _bounded(out[3], ref_hits, ref_vis, 3)
    # the same primitive with and without pairs except where an ulp decides, the same barycentrics to float accuracy
    same = (paired[0][:, 3] == out[3][0][:, 3]) & (paired[0][:, 4] == out[3][0][:, 4])
    assert same.mean() >= 1 - 1e-4
    hit = same & (paired[0][:, 4] != zyg_amd.NULL)
    assert np.all(np.abs(paired[0][hit, 1:3].view(np.float32) - out[3][0][hit, 1:3].view(np.float32)) <= 1e-4)
