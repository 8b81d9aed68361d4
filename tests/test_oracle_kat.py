"""Pins the CPU oracle (test infrastructure) against known answers: the reference ships no tests or golden vectors
(SURVEY §4), so these are the vectors SURVEY §8(c) lists — public PCG32 demo vector, sobol5/hash answers derived from
the reference's direction table and hash constants (tests/golden), curve/geometry properties and closed-form
radiometry."""
import ctypes
import json
import os

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

from util import oracle_render, oracle_resolve

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampler_golden.json")))


def test_pcg32_demo_vector(oracle):
    g = GOLDEN["pcg32_demo"]
    u = (ctypes.c_uint32 * 6)()
    f = (ctypes.c_float * 6)()
    oracle.zyg_oracle_pcg(g["state"], g["sequence"], 6, u, f)
    assert list(u) == g["out"]
    for bits, fl in zip(g["out"], f):  # generator.zig:25-33 mantissa trick
        expect = np.array([(bits & 0x007FFFFF) | 0x3F800000], dtype=np.uint32).view(np.float32)[0] - np.float32(1.0)
        assert fl == expect and 0.0 <= fl < 1.0


def test_hash_golden(oracle):
    for e in GOLDEN["hash"]:
        assert oracle.zyg_oracle_hash(e["in"]) == e["out"]


def test_sobol5_golden(oracle):
    out = (ctypes.c_uint32 * 5)()
    for e in GOLDEN["sobol5"]:
        oracle.zyg_oracle_sobol5(ctypes.c_uint32(e["index"]), out)
        assert list(out) == e["out"], e["index"]
        # dimension 0 is the van der Corput sequence (sobol.zig:195-204)
        assert out[0] == int(f"{e['index']:032b}"[::-1], 2)


def test_sobol_stratification(oracle):
    """Owen scrambling keeps the net property: 2^k consecutive samples of one pixel hit each of 2^k strata once, per
    dimension (first draw of each sample = dimension 0 of the first refill)."""
    k = 6
    n = 1 << k
    firsts = np.zeros((n, 5), dtype=np.float32)
    buf = (ctypes.c_float * 5)()
    for i in range(n):
        oracle.zyg_oracle_sobol_stream(ctypes.c_uint32(i), ctypes.c_uint32(1234), 5, 0, buf)
        firsts[i] = list(buf)
    for d in range(5):
        strata = np.floor(firsts[:, d].astype(np.float64) * n).astype(int)
        assert sorted(np.clip(strata, 0, n - 1)) == list(range(n)), d


def test_sobol_padding_forces_refill(oracle):
    a = (ctypes.c_float * 10)()
    b = (ctypes.c_float * 10)()
    oracle.zyg_oracle_sobol_stream(7, 99, 10, 0, a)
    oracle.zyg_oracle_sobol_stream(7, 99, 10, 2, b)  # padding after every 2 draws: dims 2..4 of each refill are dropped
    assert list(a)[:2] == list(b)[:2]
    assert list(a)[5:7] == list(b)[2:4]  # second refill starts right after the padding


@pytest.mark.parametrize("dim", [(60, 34), (16, 16), (7, 13), (1, 9)])
def test_gilbert_curve_is_a_unit_step_bijection(oracle, dim):
    w, h = dim
    xy = (ctypes.c_int32 * 2)()
    pts = []
    for i in range(w * h):
        oracle.zyg_oracle_gilbert(i, w, h, xy)
        pts.append((xy[0], xy[1]))
    assert sorted(pts) == sorted((x, y) for y in range(h) for x in range(w))
    steps = [abs(a[0] - b[0]) + abs(a[1] - b[1]) for a, b in zip(pts, pts[1:])]
    # generalized Hilbert curve: unit steps, except that odd x odd-ish shapes may contain one diagonal step
    assert max(steps, default=1) <= 2 and sum(s != 1 for s in steps) <= 1


def test_offset_ray_properties(oracle):
    """Waechter-Binder offset (ray_offset.zig:14-31): moves along n, by >= 1 ulp-scaled step, float path near 0."""
    rng = np.random.default_rng(5)
    out = (ctypes.c_float * 3)()
    for _ in range(200):
        p = rng.uniform(-50, 50, 3).astype(np.float32)
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        oracle.zyg_oracle_offset_ray(p.ctypes.data_as(ctypes.c_void_p), n.ctypes.data_as(ctypes.c_void_p), out)
        q = np.array(list(out), dtype=np.float32)
        d = (q.astype(np.float64) - p.astype(np.float64))
        assert np.dot(d, n.astype(np.float64)) > 0
        assert np.linalg.norm(d) < 1e-2
    p = np.array([0.01, -0.02, 0.0], dtype=np.float32)
    n = np.array([0.0, 0.0, 1.0], dtype=np.float32)
    oracle.zyg_oracle_offset_ray(p.ctypes.data_as(ctypes.c_void_p), n.ctypes.data_as(ctypes.c_void_p), out)
    assert out[2] == np.float32(1.0 / 65536.0) and out[0] == p[0] and out[1] == p[1]


def test_trig_accuracy(oracle):
    """The restated sin / cos / acos (musl algorithms) stay within their published error bounds."""
    rng = np.random.default_rng(1)
    x = rng.uniform(-7, 7, 200000).astype(np.float32)
    s, c, a = (np.zeros_like(x) for _ in range(3))
    vp = lambda v: v.ctypes.data_as(ctypes.c_void_p)
    oracle.zyg_oracle_trig(x.size, vp(x), vp(s), vp(c), vp(a))

    def ulp(got, ref):
        return np.max(np.abs(got.astype(np.float64) - ref) / np.spacing(np.abs(ref.astype(np.float32))).astype(np.float64))

    assert ulp(s, np.sin(x.astype(np.float64))) < 0.51
    assert ulp(c, np.cos(x.astype(np.float64))) < 0.51
    xa = rng.uniform(-1, 1, 200000).astype(np.float32)
    oracle.zyg_oracle_trig(xa.size, vp(xa), vp(s), vp(c), vp(a))
    assert ulp(a, np.arccos(xa.astype(np.float64))) < 1.0


def test_exp_accuracy(oracle):
    """The restated expf (msun e_expf.c) stays under 1 ulp, and agrees with libm's correctly rounded result almost always."""
    rng = np.random.default_rng(2)
    x = np.concatenate([-rng.exponential(3.0, 150000), rng.uniform(-87, 5, 50000)]).astype(np.float32)
    e = np.zeros_like(x)
    oracle.zyg_oracle_exp(x.size, x.ctypes.data_as(ctypes.c_void_p), e.ctypes.data_as(ctypes.c_void_p))
    ref = np.exp(x.astype(np.float64))
    err = np.abs(e.astype(np.float64) - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert err.max() < 1.0
    assert np.mean(e == ref.astype(np.float32)) > 0.9
    one = np.array([0.0, -0.0], dtype=np.float32)
    oracle.zyg_oracle_exp(2, one.ctypes.data_as(ctypes.c_void_p), e.ctypes.data_as(ctypes.c_void_p))
    assert e[0] == 1.0 and e[1] == 1.0


def _quad_trafo(size, pos):
    t = zyg_amd.Trafo()
    rows = [(1, 0, 0, size[0]), (0, 1, 0, size[1]), (0, 0, 1, 1.0)]
    for i, r in enumerate(rows):
        for j in range(4):
            t.r[i][j] = r[j]
    for j in range(3):
        t.position[j] = pos[j]
    return t


def test_sphquad_solid_angle_closed_form(oracle):
    """SphQuad.S (rectangle.zig:199-262) vs the closed form for a point on the axis of an a x b rectangle at height d:
    Omega = 4 atan(a b / (2 d sqrt(4 d^2 + a^2 + b^2)))."""
    sa = ctypes.c_float()
    pdf = ctypes.c_float()
    for a, b, d in [(1.0, 1.0, 1.0), (0.5, 0.5, 1.98), (2.0, 0.7, 0.3), (0.3, 3.0, 5.0)]:
        t = _quad_trafo((a, b), (0, 0, 0))
        p = (ctypes.c_float * 3)(0.0, 0.0, d)
        oracle.zyg_oracle_sphquad(ctypes.byref(t), p, ctypes.byref(sa), ctypes.byref(pdf))
        omega = 4.0 * np.arctan(a * b / (2.0 * d * np.sqrt(4.0 * d * d + a * a + b * b)))
        assert sa.value == pytest.approx(omega, rel=2e-5)
        assert pdf.value == pytest.approx(1.0 / omega, rel=2e-5)


def _irradiance_from_rect(a, b, d):
    """Irradiance on a plane parallel to an a x b Lambertian rectangle of radiance 1 at distance d, below its centre:
    4 x the corner formula E = 1/2 [ X/sqrt(1+X^2) atan(Y/sqrt(1+X^2)) + Y/sqrt(1+Y^2) atan(X/sqrt(1+Y^2)) ], X=a/2d, Y=b/2d."""
    x, y = a / (2 * d), b / (2 * d)
    return 4 * 0.5 * (x / np.sqrt(1 + x * x) * np.arctan(y / np.sqrt(1 + x * x)) + y / np.sqrt(1 + y * y) * np.arctan(x / np.sqrt(1 + y * y)))


@pytest.mark.parametrize("integrator", ["PT", "PTDL", "PTMIS"])
def test_quad_light_irradiance_closed_form(engine, oracle, integrator):
    """One emissive quad above a Lambert floor, depth 1: radiance towards the camera = albedo/pi * E (closed form)."""
    w = h = 8
    spp = 2048 if integrator == "PT" else 256
    cam = engine.camera(w, h, fov_deg=0.2)  # a thin pencil of rays around the floor point below the light
    engine.samples_per_pixel(spp)
    engine.integrators(json.dumps({"surface": {integrator: {"depth": {"surface": 1}}}}))
    engine.sensor("{}")
    albedo = 0.6
    floor = engine.material(scenes.lambert([albedo, albedo, albedo]))
    lamp = engine.material(scenes.light_material(spectrum=(1.0, 1.0, 1.0)))
    p = engine.prop(scenes.RECTANGLE, [floor])
    engine.transformation(p, scenes.trafo((1, 0, 0), (0, 0, -1), (50, 50, 1), (0, 0, 0)))
    lp = engine.prop(scenes.RECTANGLE, [lamp], unoccluding=True)
    a, b, d = 1.0, 0.6, 0.8
    engine.transformation(lp, scenes.trafo((1, 0, 0), (0, 0, 1), (a, b, 1), (0, d, 0)))
    engine.light(lp)
    # camera above the light's side, looking at the origin: +z is the view direction
    eye = np.array([0.0, 3.0, -4.0])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross([0, 1, 0], fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3], m[3, :3], m[3, 3] = right, up, fwd, eye, 1
    engine.transformation(cam, m)
    sd, vd = engine.compile()
    film = oracle_render(oracle, sd, vd, spp)
    rgb = film[..., :3].sum(axis=(0, 1)) / film[..., 3].sum()
    # json colours are sRGB -> AP1 on load (base/json.zig:247-277); grey stays (almost) grey, so compare luminance-free means
    albedo_ap1 = np.array(sd.materials[floor].color_a[:], dtype=np.float64)
    emission_ap1 = np.array(sd.materials[lamp].emission[:], dtype=np.float64)
    expect = albedo_ap1 / np.pi * _irradiance_from_rect(a, b, d) * emission_ap1
    np.testing.assert_allclose(rgb, expect, rtol=0.03 if integrator == "PT" else 0.01)


def test_integrators_agree_on_cornell(product_lib, oracle):
    """PT, PTDL and PTMIS estimate the same integral: equal means within Monte-Carlo noise."""
    w, h, spp = 48, 36, 96
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator=integ)
            sd, vd = e.compile()
            film = oracle_render(oracle, sd, vd, spp)
            means[integ] = oracle_resolve(oracle, vd, film)[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.01)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.04)


def test_progressive_iterations_are_independent(engine, oracle):
    """Rendering iterations [0,2) then [2,4) equals rendering [0,4) (worker.zig:141-149 reseeds per (pixel, iteration))."""
    scenes.cornell(engine, 40, 30, 4)
    sd, vd = engine.compile()
    a = oracle_render(oracle, sd, vd, 2, threads=2) + 0
    b = oracle_render(oracle, sd, vd, 2, threads=3, iteration0=2)
    c = oracle_render(oracle, sd, vd, 4, threads=1)
    # film sums: (s0 + s1) + (s2 + s3) vs ((s0 + s1) + s2) + s3 differ only by float association
    np.testing.assert_allclose(a + b, c, rtol=1e-6, atol=1e-7)


def test_row_ranges_partition_the_frame(engine, oracle):
    scenes.cornell(engine, 40, 30, 2)
    sd, vd = engine.compile()
    full = oracle_render(oracle, sd, vd, 2)
    parts = oracle_render(oracle, sd, vd, 2, y_range=(0, 11)) + oracle_render(oracle, sd, vd, 2, y_range=(11, 30))
    assert np.array_equal(full, parts)


def test_empty_scene_and_zero_iterations(engine, oracle):
    """Edge cases: a camera looking at nothing gives a black film with exact weights; zero iterations leave it untouched."""
    cam = engine.camera(16, 8)
    engine.samples_per_pixel(1)
    engine.integrators(json.dumps({"surface": {"PTMIS": {}}}))
    engine.sensor("{}")
    sd, vd = engine.compile()
    assert sd.solid.num_nodes == 0 and sd.num_lights == 0
    film = oracle_render(oracle, sd, vd, 3)
    assert np.all(film[..., :3] == 0) and np.all(film[..., 3] == 3)
    assert np.all(oracle_render(oracle, sd, vd, 0) == 0)


def test_substitute_integrators_agree_and_conserve_energy(product_lib, oracle):
    """Substitute closure (substitute_sample.zig:38-410 with the compile-time GGX albedo tables): PTDL and PTMIS estimate
    the same image (evaluate and sample are consistent, pdfs match), and a rough dielectric Substitute wall set stays
    within a few percent of the Lambert box of the same albedo (the base lobe is energy-compensated Lambert-like)."""
    w, h, spp = 48, 36, 128
    means = {}
    for key, kw in (("dl", dict(integrator="PTDL", material="substitute")),
                    ("mis", dict(integrator="PTMIS", material="substitute")),
                    ("pt", dict(integrator="PT", material="substitute")),
                    ("lambert", dict(integrator="PTMIS", material="lambert"))):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, **kw)
            sd, vd = e.compile()
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[key] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["dl"], means["mis"], rtol=0.015)
    np.testing.assert_allclose(means["pt"], means["mis"], rtol=0.05)
    np.testing.assert_allclose(means["mis"], means["lambert"], rtol=0.25)
    assert np.all(means["mis"] > 0.02)


def test_mesh_lights_agree_with_rectangle_lights(product_lib, oracle):
    """The two emissive quads as TriangleMesh lights (primitive tree + spherical-triangle sampling, triangle_mesh.zig:391-703)
    render the same image as the Rectangle lights they replace, for the three integrators (PT has no light sampling at all,
    so it pins the MIS weights through Mesh.pdf)."""
    w, h, spp = 48, 36, 128
    means = {}
    for key, kw in (("rect", dict(integrator="PTMIS")), ("mesh_mis", dict(integrator="PTMIS", mesh_lights=True, light_grid=4)),
                    ("mesh_dl", dict(integrator="PTDL", mesh_lights=True, light_grid=4)),
                    ("mesh_pt", dict(integrator="PT", mesh_lights=True, light_grid=4)),
                    ("mesh_one", dict(integrator="PTMIS", mesh_lights=True, light_grid=1))):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, unoccluding_lights=False, **kw)
            sd, vd = e.compile()
            assert sd.num_mesh_lights == (0 if key == "rect" else 2)
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[key] = img[..., :3].mean(axis=(0, 1))
    for key in ("mesh_mis", "mesh_dl", "mesh_one"):
        np.testing.assert_allclose(means[key], means["rect"], rtol=0.01)
    np.testing.assert_allclose(means["mesh_pt"], means["rect"], rtol=0.04)


def test_distant_light_integrators_agree(product_lib, oracle):
    """Infinite (Distant) light: light sampling (PTDL), MIS (PTMIS) and plain path tracing (PT) estimate the same image.  The
    sun disc is kept small: Distant.intersect accepts sin(angle) <= radius while sampleTo and the solid angle use
    tan(angle) <= radius (distant.zig:33, 84-97, 143-145), which only agree for small discs."""
    w, h, spp = 40, 30, 512
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.outdoor(e, w, h, spp, integrator=integ, sun_radius=0.12, extra_quad_light=True, max_depth=5)
            sd, vd = e.compile()
            assert sd.light_tree.num_infinite_lights == 1 and sd.num_infinite_props == 1
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[integ] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.01)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.06)


@pytest.mark.parametrize("sky", ["dome", "canopy"])
def test_sky_light_integrators_agree(product_lib, oracle, sky):
    """Uniform Dome / Canopy environment lights (dome.zig, canopy.zig) next to the sun: PTDL, PTMIS and PT agree, and the
    image is brighter than with the sun alone."""
    w, h, spp = 40, 30, 256
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.outdoor(e, w, h, spp, integrator=integ, sun_radius=0.12, sky=sky, max_depth=5)
            sd, vd = e.compile()
            assert sd.light_tree.num_infinite_lights == 2 and sd.num_infinite_props == 2
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[integ] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.01)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.06)
    with zyg_amd.Engine(product_lib) as e:
        scenes.outdoor(e, w, h, spp, integrator="PTMIS", sun_radius=0.12, max_depth=5)
        sd, vd = e.compile()
        sun_only = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))[..., :3].mean(axis=(0, 1))
    assert (means["PTMIS"] > 1.2 * sun_only).all()


def test_sphere_lights_integrators_agree(product_lib, oracle):
    """Sphere lights and Sphere geometry: PT (no light sampling), PTDL and PTMIS estimate the same image."""
    w, h, spp = 40, 30, 256
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator=integ, spheres=True, sphere_lights=True, unoccluding_lights=False, boxes=False)
            sd, vd = e.compile()
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[integ] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.01)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.04)


def test_glass_integrators_agree(product_lib, oracle):
    """Glass: PT and PTDL (one random choice between reflection and refraction) and PTMIS (both, as split paths weighted
    by the Fresnel term) estimate the same image; a dielectric with the ior of its surroundings is invisible."""
    w, h, spp = 40, 30, 256
    objs = {"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)}
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator=integ, glass_objects=objs, max_depth=12)
            sd, vd = e.compile()
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[integ] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.015)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.04)
    plain = {}
    for key, objs in (("none", None), ("ghost", {"sphere": scenes.glass(1.0), "inner": scenes.glass(1.0)})):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator="PTMIS", glass_objects=objs, max_depth=12)
            sd, vd = e.compile()
            plain[key] = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(plain["ghost"], plain["none"], rtol=0.01)
    assert abs(means["PTMIS"][0] - plain["none"][0]) > 0.005  # real glass does change the picture


@pytest.mark.parametrize("sky", ["dome", "canopy"])
def test_image_light_matches_uniform_light(product_lib, oracle, sky):
    """An emission map of constant value c is the uniform light of radiance c: the importance-sampled image light
    (Distribution2D over luminance x uv weight, dome.zig:105-158 / canopy.zig:94-133) and the uniformly sampled
    Dome / Canopy estimate the same image."""
    w, h, spp = 40, 30, 256
    means = {}
    for key, kw in (("uniform", dict(sky_spectrum=(0.7, 1.0, 1.6))),
                    ("map", dict(sky_spectrum=(0.35, 0.5, 0.8), sky_map=scenes.sky_image(32, 16, uniform=2.0)))):
        with zyg_amd.Engine(product_lib) as e:
            scenes.outdoor(e, w, h, spp, integrator="PTMIS", sun=False, sky=sky, max_depth=5, **kw)
            sd, vd = e.compile()
            assert sd.num_image_lights == (1 if key == "map" else 0)
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[key] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["map"], means["uniform"], rtol=0.015)


@pytest.mark.parametrize("sky", ["dome", "canopy"])
def test_image_light_integrators_agree(product_lib, oracle, sky):
    """A non-uniform environment map (gradient + a small bright blob): light sampling by the map (PTDL), MIS and plain path
    tracing agree; so does splitting the light into 4 samples."""
    w, h, spp = 40, 30, 512
    means = {}
    for key, integ, n in (("PT", "PT", 1), ("PTDL", "PTDL", 1), ("PTMIS", "PTMIS", 1), ("PTMIS4", "PTMIS", 4)):
        with zyg_amd.Engine(product_lib) as e:
            scenes.outdoor(e, w, h, spp, integrator=integ, sun=False, sky=sky, max_depth=4, sky_map=scenes.sky_image(), sky_samples=n,
                           sky_spectrum=(1.0, 1.0, 1.0))
            sd, vd = e.compile()
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[key] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.02)
    np.testing.assert_allclose(means["PTMIS4"], means["PTMIS"], rtol=0.02)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.1)


def test_tonemapper_known_answers(product_lib, oracle):
    """Tonemapper.tonemap (tonemapper.zig:18-71): Linear is AP1 -> sRGB; PbrNeutral leaves small values alone but for the
    0.04 toe offset; ACES maps black to ~0 and compresses large values below 1; the exposure is 2^e."""
    film = np.zeros((1, 4, 4), dtype=np.float32)
    film[0, :, 3] = 1.0
    film[0, 1, :3] = 0.18
    film[0, 2, :3] = 4.0
    film[0, 3, :3] = (0.5, 0.05, 0.01)
    out = {}
    for name, exposure in (("Linear", 0.0), ("Linear", 1.0), ("ACES", 0.0), ("PbrNeutral", 0.0), ("AgX", 0.0)):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, 4, 1, 1)
            e.sensor(json.dumps({"tonemapper": {name: {"exposure": exposure}}}))
            _, vd = e.compile()
            out[(name, exposure)] = oracle_resolve(oracle, vd, film)[0, :, :3]
    lin = out[("Linear", 0.0)]
    m = np.array([[1.70505155, -0.13025714, -0.02400328], [-0.62179068, 1.14080289, -0.12896877], [-0.08325840, -0.01054853, 1.15297171]])
    np.testing.assert_allclose(lin[3], np.array([0.5, 0.05, 0.01]) @ m, rtol=1e-6)
    np.testing.assert_allclose(out[("Linear", 1.0)], 2.0 * lin, rtol=1e-6)
    aces = out[("ACES", 0.0)]
    assert abs(aces[0]).max() < 1e-3 and 0.9 < aces[2].max() <= 1.01 and aces[1].max() < aces[2].max()
    pbr = out[("PbrNeutral", 0.0)]
    x = lin[1].min()
    np.testing.assert_allclose(pbr[1], lin[1] - 0.04, rtol=1e-5)  # below the compression start, toe offset 0.04 (x >= 0.08)
    assert x >= 0.08 and pbr[2].max() < 1.0
    agx = out[("AgX", 0.0)]
    assert agx.min() >= 0.0 and agx.max() <= 1.1 and agx[2].mean() > agx[1].mean() > agx[0].mean()


def test_rough_glass_integrators_agree(product_lib, oracle):
    """Rough dielectrics: path tracing with and without light sampling through the interface (Glass.evaluate) and with the
    split paths estimate the same image; energy stays close to the smooth case (the E_s compensation, ggx.zig:30-32)."""
    w, h, spp = 40, 30, 512
    means = {}
    for integ in ("PT", "PTMIS"):
        for rough in (0.0, 0.3):
            objs = {"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3, roughness=rough), "sphere": scenes.glass(1.5, roughness=rough)}
            with zyg_amd.Engine(product_lib) as e:
                scenes.cornell(e, w, h, spp, integrator=integ, glass_objects=objs, max_depth=12)
                sd, vd = e.compile()
                img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
                assert np.isfinite(img).all()
                means[(integ, rough)] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means[("PT", 0.3)], means[("PTMIS", 0.3)], rtol=0.02)
    np.testing.assert_allclose(means[("PTMIS", 0.3)], means[("PTMIS", 0.0)], rtol=0.03)


def test_thin_glass_integrators_agree(product_lib, oracle):
    """Thin glass: light sampling attenuated through the panes (Scene.visibility's transmittance) against plain path
    tracing, which passes them as .Straight events; both darker than the scene without glass absorption would be."""
    w, h, spp = 40, 30, 512
    objs = {"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3, thickness=0.02), "sphere": scenes.glass(1.5, (0.9, 0.7, 0.6), 0.1, thickness=0.01)}
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator=integ, glass_objects=objs, max_depth=12)
            sd, vd = e.compile()
            flags = [sd.props[i].flags for i in range(sd.num_props)]
            assert sum(1 for f in flags if f & 16) == 2  # ZYGPU_PROP_EVALUATE_VISIBILITY on the two glass props
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[integ] = img[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.01)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.02)


def test_textured_rectangle_light_matches_uniform(product_lib, oracle):
    """A Rectangle light with a constant emission map c (Rectangle.sampleMaterialTo: area sampling by the map's
    distribution, rectangle.zig:399-456, 577-595) estimates the same image as the solid-angle-sampled uniform light of
    radiance c x spectrum; PT agrees with PTMIS for a patterned map."""
    w, h, spp = 40, 30, 256
    means = {}
    for key, kw in (("uniform", dict(light_spectrum=(8.5, 6.0, 2.0))),
                    ("map", dict(light_spectrum=(17.0, 12.0, 4.0), light_map=np.full((8, 8, 3), 0.5, dtype=np.float32)))):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator="PTMIS", unoccluding_lights=False, **kw)
            sd, vd = e.compile()
            assert sd.num_image_lights == (2 if key == "map" else 0)
            means[key] = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["map"], means["uniform"], rtol=0.02)
    pattern = scenes.checker_image(8, np.float32)
    for integ in ("PT", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, 512, integrator=integ, light_map=pattern, light_samples=2 if integ == "PTMIS" else 1)
            sd, vd = e.compile()
            means[integ] = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, 512))[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.04)


def test_opacity_mask_limits_and_integrators_agree(product_lib, oracle):
    """Opacity masks (material_base.zig:29-55, material.zig:238-245): a mask of zeros removes the masked objects (the image
    of the scene without them, up to noise), a half-opaque mask lies between that and the solid objects; light sampling
    through masks (PTDL / PTMIS: transmittance 1 - o) estimates what plain path tracing does (stochastic hits)."""
    w, h, spp = 40, 30, 384
    means = {}

    def render(key, integ="PTMIS", **kw):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator=integ, material="substitute", max_depth=8, **kw)
            sd, vd = e.compile()
            if "masked" in kw:
                flags = [sd.props[i].flags for i in range(sd.num_props)]
                assert sum(1 for f in flags if f & 16) == 4  # ZYGPU_PROP_EVALUATE_VISIBILITY: Rectangle, Sphere, Disk, tall box
                masks = {sd.materials[i].mask_image for i in range(sd.num_materials)} - {zyg_amd.NULL}
                assert len(masks) == 1 and sd.images[masks.pop()].format in (2, 3)  # one-channel formats
            means[key] = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))[..., :3].mean(axis=(0, 1))

    render("zeros", masked={"image": np.zeros((4, 4), dtype=np.uint8)})
    render("half", masked={"image": np.full((4, 4), 0.5, dtype=np.float32)})
    render("ones", masked={"image": np.full((4, 4), 255, dtype=np.uint8)})
    with zyg_amd.Engine(product_lib) as e:  # the scene without the masked objects and without the tall box
        scenes.cornell(e, w, h, spp, integrator="PTMIS", material="substitute", max_depth=8, boxes=False)
        idx, pos, nrm, uv = scenes.box_mesh((0.15, 0.0, -0.65), (0.75, 0.6, -0.05), -18.0)
        p = e.prop(e.triangle_mesh(idx, pos, nrm, uv), [e.material(scenes.substitute(scenes.WHITE))])
        e.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 0)))
        sd, vd = e.compile()
        means["without"] = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["zeros"], means["without"], rtol=0.02)
    assert np.all(means["ones"] < means["half"]) and np.all(means["half"] < means["zeros"])  # the objects shade the box
    pattern = {"image": scenes.mask_image(16, np.uint8)}
    for integ in ("PT", "PTDL", "PTMIS"):
        render(integ, integ=integ, masked=pattern)
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.02)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.04)


def test_mask_needs_a_one_channel_image(product_lib):
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 8, 8, 1, material="substitute")
        rgb = e.image(scenes.checker_image(4, np.float32))
        p = e.prop(scenes.RECTANGLE, [e.material(scenes.with_mask(scenes.substitute(scenes.RED), rgb))])
        e.transformation(p, scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1, 0)))
        assert e.lib.su_ext_compile(0) != 0


def test_disk_lights_match_their_closed_form_and_integrators_agree(product_lib, oracle):
    """Disk emitters (disk.zig:252-359, 492-533): the three integrators estimate the same image, and a small Disk light lights
    the floor like the Rectangle light of the same area and radiance (far-field limit)."""
    w, h, spp = 40, 30, 384
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, integrator=integ, disk_lights=True, unoccluding_lights=False, max_depth=8, light_samples=2 if integ == "PTMIS" else 1)
            sd, vd = e.compile()
            assert [sd.props[sd.lights[i].prop].shape for i in range(sd.num_lights)] == [2, 2]
            means[integ] = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))[..., :3].mean(axis=(0, 1))
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.02)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.04)


def test_aov_known_answers(product_lib, oracle):
    """AOVs of the open Cornell box seen from the front (worker.zig:209-241, aov_buffer.zig:48-78): the back wall's pixels
    carry its normal (0, 0, -1), its depth along the ray, its albedo and material id; Emission + Direct + Indirect resolve to
    the beauty image (Linear tonemapper, exposure 0); inactive classes stay out."""
    from util import AOV_NAMES, oracle_render_aovs, oracle_resolve_aov
    w, h, spp = 48, 36, 16
    with zyg_amd.Engine(product_lib) as e:
        e.aovs(AOV_NAMES)
        ids = scenes.cornell(e, w, h, spp, integrator="PTMIS", boxes=False)
        sd, vd = e.compile()
        assert vd.aov_slots == 0x1FF
        film, planes = oracle_render_aovs(oracle, sd, vd, spp)
        beauty = oracle_resolve(oracle, vd, film)
        res = {k: oracle_resolve_aov(oracle, vd, k, planes[k]) for k in range(9)}
        back_material = sd.material_ids[sd.props[ids["walls"][2]].parts_start]
        ap1 = np.array(list(sd.materials[back_material].color_a))
    cy, cx = h // 2, w // 2  # the back wall, z = +1, seen from z = -3.9 at height 1
    np.testing.assert_allclose(res[3][cy, cx, :3], [0, 0, -1], atol=1e-6)
    np.testing.assert_allclose(res[4][cy, cx, :3], [0, 0, -1], atol=1e-6)
    assert abs(res[1][cy, cx, 0] - 4.9) < 0.02  # the nearest of the pixel's samples
    np.testing.assert_allclose(res[5][cy, cx, 0], 1.0, atol=1e-6)  # Debug material: alpha 1
    assert res[2][cy, cx, 0] == 1 + back_material
    # albedo = the wall's colour (AP1 -> sRGB primaries on resolve)
    to_srgb = np.array([[1.70505155, -0.62179068, -0.08325840], [-0.13025714, 1.14080289, -0.01054853], [-0.02400328, -0.12896877, 1.15297171]])
    np.testing.assert_allclose(res[0][cy, cx, :3], to_srgb @ ap1, rtol=1e-5)
    np.testing.assert_allclose((res[6] + res[7] + res[8])[..., :3], beauty[..., :3], rtol=2e-4, atol=1e-5)
    with zyg_amd.Engine(product_lib) as e:
        e.aovs(["Depth", "Albedo"])
        scenes.cornell(e, 16, 12, 1)
        _, vd = e.compile()
        assert vd.aov_slots == 0b11


def test_portal_light_integrators_agree(product_lib, oracle):
    """A Portal (rectangle.zig:458-513, windowed_distribution_2d.zig) only changes how PathtracerDL / PathtracerMIS *sample* the
    environment map behind the window; Pathtracer finds the same light by letting its rays leave through the window.  All three
    must estimate the same image: the windowed distribution's density, the rectified mapping's Jacobian and lightPortalUvw's lookup
    are only consistent with each other if they do.  One and four samples per pick agree too."""
    w, h, spp = 40, 30, 768
    sky = scenes.sky_image(64, 32)
    means = {}
    for key, kw in (("PT", dict(integrator="PT")), ("PTDL", dict(integrator="PTDL")), ("PTMIS", dict(integrator="PTMIS")),
                    ("PTMIS4", dict(integrator="PTMIS", light_samples=4))):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, w, h, spp, portal_map=sky, **kw)
            sd, vd = e.compile()
            assert 1 == sd.num_portal_lights
            img = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(img).all()
            means[key] = img[8:, :, :3].mean(axis=(0, 1))  # (below the window's rows, which every integrator sees directly)
    assert (means["PTMIS"] > 0.01).all()
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.03)
    np.testing.assert_allclose(means["PTMIS4"], means["PTMIS"], rtol=0.03)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.08)


def test_sky_entity_integrators_agree(product_lib, oracle):
    """The Sky entity's sun is a Distant light whose radiance varies across its disc (sky_material.zig:150-168: a table looked up by
    sunV of the direction).  Light sampling (PTDL) and MIS must agree on what it sends to the ground, and plain path tracing, which
    only finds the sun by chance, within its noise.  The sky here is a dim uniform image: the reference gives a sample of a Canopy
    map the density pdf_uv / 2 pi (canopy.zig:94-133), which the equidistant disc only has at its rim (at the zenith the solid
    angle per uv area is pi / 2 times larger), so PathtracerDL under-weights the zenith of a textured canopy by up to 36 %; the
    restatement keeps that, and a map whose colours differ between zenith and horizon shows it as a colour shift between PTDL and
    the other two (measured: blue -6 %, red +-0 for scenes.sky_tables' image).  A uniform map is unaffected in expectation."""
    w, h, spp = 40, 30, 512
    sun, _ = scenes.sky_tables(elevation_deg=50.0)
    dim = np.full((32, 32, 3), 0.02, dtype=np.float32)
    entity = {"parameters": {"sun": {"rotation": [-50, 150, 0]}}, "tables": (sun, dim)}
    means = {}
    for integ in ("PT", "PTDL", "PTMIS"):
        with zyg_amd.Engine(product_lib) as e:
            scenes.outdoor(e, w, h, spp, integrator=integ, sky_entity=entity, max_depth=5)
            sd, vd = e.compile()
            assert 1024 == sd.num_sun_radiance
            out = oracle_resolve(oracle, vd, oracle_render(oracle, sd, vd, spp))
            assert np.isfinite(out).all()
            means[integ] = out[h // 2:, :, :3].mean(axis=(0, 1))  # the ground and the boxes
    assert (means["PTMIS"] > 0.05).all()
    np.testing.assert_allclose(means["PTDL"], means["PTMIS"], rtol=0.02)
    np.testing.assert_allclose(means["PT"], means["PTMIS"], rtol=0.25)  # (a sun of 0.27 degrees: plain path tracing rarely finds it)
