"""Procedural textures (material_provider.zig:465-578, texture/procedural*.zig, texture/noise/*.zig).

CPU part: the host's parse of the descriptions into ZygpuSceneDesc.procedurals, what it refuses, and the oracle's noise against a
second implementation written here from the published algorithms the reference's files name (Perlin gradient noise with Bob Jenkins'
lookup3 final mix as MaterialX / OSL hash it; Worley noise over a PCG hash).  The reference ships no vectors for these and cannot be
run here: parity unpinned, as for the rest of the path.  GPU part (tests/test_gpu_parity.py style): device == oracle, bit for bit."""
import ctypes
import json
import math
import struct

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

from util import oracle_render, oracle_resolve, rms

S = scenes.substitute_full


def checker(scale=4.0, a=(0.9, 0.1, 0.1), b=(0.1, 0.1, 0.9), **more):
    t = {"Checker": {"scale": scale, "colors": [list(a), list(b)]}}
    t.update(more)
    return t


def noise(**kw):
    sampler = kw.pop("sampler", None)
    t = {"Noise": kw}
    if sampler:
        t["sampler"] = sampler
    return t


def _procedurals(engine):
    sd = engine.lib.su_ext_scene_desc().contents
    return [sd.procedurals[i] for i in range(sd.num_procedurals)], sd


def test_descriptions_become_entries_of_the_scene(product_lib):
    e = zyg_amd.Engine(product_lib)
    try:
        mix = {"Mix": {"a": checker(2.0), "b": [0.2, 0.8, 0.2], "weight": noise(levels=3, scale=[4, 4, 4], attenuation=0.7)}}
        ov = {"floor": S(checker(8.0, sampler={"texcoord": "Triplanar", "address": "Clamp"}), roughness=noise(**{"class": "Cellular", "invert": True, "ratio": 0.3, "transition": 0.2})),
              "back": S(mix, normal={"DetailNormal": {"base": noise(scale=[9, 9, 9]), "detail": noise(scale=[31, 31, 31], absolute=True)}}),
              "tall": S({"ChannelMix": {"source": checker(3.0), "channels": [[0, 0, 1], [0, 1, 0], [1, 0, 0]]}},
                        metallic={"Max": {"a": 0.25, "b": noise(period=[4, 4, 0], scale=[4, 4, 1])}},
                        roughness={"Mul": {"a": 0.5, "b": noise()}})}
        scenes.cornell(e, 32, 24, 1, material="substitute", overrides=ov)
        e.compile()
        pr, sd = _procedurals(e)
        kinds = [p.kind for p in pr]
        # children are appended before the procedural that owns them (the device relies on it: no cycles)
        for k, p in enumerate(pr):
            for c in p.child:
                if c.image != zyg_amd.NULL and (c.sampler & zyg_amd.TEX_PROCEDURAL):
                    assert c.image < k
        names = {0: "ChannelMix", 1: "Checker", 2: "DetailNormal", 3: "Max", 4: "Mix", 5: "Mul", 6: "Noise"}
        assert sorted(set(kinds)) == [0, 1, 2, 3, 4, 5, 6], [names[k] for k in kinds]
        mats = [sd.materials[i] for i in range(sd.num_materials)]
        floor = [m for m in mats if m.color_image != zyg_amd.NULL and (m.color_sampler & zyg_amd.TEX_PROCEDURAL) and (m.color_sampler & zyg_amd.TEX_TRIPLANAR)]
        assert len(floor) == 1
        m = floor[0]
        assert m.flags & zyg_amd.MATF_PROCEDURAL and m.flags & zyg_amd.MATF_EXTENDED
        assert not (m.color_sampler & (zyg_amd.TEX_REPEAT_U | zyg_amd.TEX_REPEAT_V))
        c = pr[m.color_image]
        assert c.kind == 1 and c.checker_scale == 8.0 and list(c.value[0])[:3] != list(c.value[1])[:3]
        r = pr[m.tex[zyg_amd.STEX_ROUGHNESS].image]
        assert r.kind == 6 and r.flags == 1 | 4 and r.levels == 1 and abs(r.ratio - 0.3) < 1e-7 and abs(r.transition - 0.2) < 1e-7
        mixes = [p for p in pr if p.kind == 4]
        assert len(mixes) == 1
        assert mixes[0].child[1].image == zyg_amd.NULL and abs(mixes[0].child_uniform[1][1] - 0.8) < 0.2   # a uniform child (AP1 of sRGB)
        w = pr[mixes[0].child[2].image]
        assert w.kind == 6 and w.levels == 3 and abs(w.attenuation - 0.7) < 1e-7 and list(w.scale)[:3] == [4.0, 4.0, 4.0]
    finally:
        e.release()


@pytest.mark.parametrize("what", ["mask", "emission", "deep"])
def test_what_is_refused(product_lib, what):
    """A procedural opacity mask or emission map (evaluated as images by the kernels that see them: the traversal, the light samples)
    and a nesting deeper than ZYGPU_PROCEDURAL_MAX_DEPTH flag the material: the scene does not compile."""
    e = zyg_amd.Engine(product_lib)
    try:
        if what == "mask":
            ov = {"tall": json.dumps({"rendering": {"Substitute": {"color": [0.5, 0.5, 0.5], "mask": noise()}}})}
        elif what == "emission":
            ov = {"tall": json.dumps({"rendering": {"Substitute": {"color": [0.5, 0.5, 0.5], "emittance": {"spectrum": [1, 1, 1], "emission_map": checker()}}}})}
        else:
            t = noise()
            for _ in range(5):
                t = {"Mul": {"a": t, "b": 0.5}}
            ov = {"tall": S((0.5, 0.5, 0.5), roughness=t)}
        with pytest.raises(Exception):
            scenes.cornell(e, 32, 24, 1, material="substitute", overrides=ov)
            e.compile()
    finally:
        e.release()


# ---- a second implementation of the noise functions (pure Python, IEEE single precision through struct round trips)
def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def _rotl(x, k):
    return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF


def _bjfinal(a, b, c):  # lookup3's final()
    M = 0xFFFFFFFF
    c ^= b; c = (c - _rotl(b, 14)) & M
    a ^= c; a = (a - _rotl(c, 11)) & M
    b ^= a; b = (b - _rotl(a, 25)) & M
    c ^= b; c = (c - _rotl(b, 16)) & M
    a ^= c; a = (a - _rotl(c, 4)) & M
    b ^= a; b = (b - _rotl(a, 14)) & M
    c ^= b; c = (c - _rotl(b, 24)) & M
    return c


def _fade(t):
    return t * t * t * (t * (t * 6.0 - 15.0) + 10.0)


def _grad2(h, x, y):
    h &= 7
    u, v = (x, 2.0 * y) if h < 4 else (y, 2.0 * x)
    return (-u if h & 1 else u) + (-v if h & 2 else v)


def _perlin2(x, y, period=(0, 0)):
    X, Y = math.floor(x), math.floor(y)
    fx, fy = x - X, y - Y
    start = (0xdeadbeef + (2 << 2) + 13) & 0xFFFFFFFF
    def cell(ix, iy, gx, gy):
        if period[0] > 0:
            ix %= period[0]
        if period[1] > 0:
            iy %= period[1]
        return _grad2(_bjfinal((start + ix) & 0xFFFFFFFF, (start + iy) & 0xFFFFFFFF, start), gx, gy)
    c = [cell(X, Y, fx, fy), cell(X + 1, Y, fx - 1, fy), cell(X, Y + 1, fx, fy - 1), cell(X + 1, Y + 1, fx - 1, fy - 1)]
    u, v = _fade(fx), _fade(fy)
    return 0.6616 * ((1 - v) * ((1 - u) * c[0] + u * c[1]) + v * ((1 - u) * c[2] + u * c[3]))


def _pcg2d(a, b):
    M = 0xFFFFFFFF
    a = (a * 1664525 + 1013904223) & M
    b = (b * 1664525 + 1013904223) & M
    a = (a + b * 1664525) & M
    b = (b + a * 1664525) & M
    a ^= a >> 16
    b ^= b >> 16
    a = (a + b * 1664525) & M
    b = (b + a * 1664525) & M
    a ^= a >> 16
    b ^= b >> 16
    return a, b


def _to01(bits):
    return struct.unpack("f", struct.pack("I", (bits & 0x007FFFFF) | 0x3F800000))[0] - 1.0


def _worley2(x, y):
    X, Y = math.floor(x), math.floor(y)
    lx, ly = x - X, y - Y
    best = 1e6
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            a, b = _pcg2d((X + dx) & 0xFFFFFFFF, (Y + dy) & 0xFFFFFFFF)
            ox, oy = _to01(a), _to01(b)
            d = (dx + ox - lx) ** 2 + (dy + oy - ly) ** 2
            best = min(best, d)
    return best


def _oracle_noise(oracle, what, pts):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.zeros(len(pts), dtype=np.float32)
    oracle.zyg_oracle_noise(ctypes.c_uint32(len(pts)), pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(what), out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_gradient_noise_against_a_second_implementation(oracle):
    rng = np.random.default_rng(7)
    pts = np.zeros((4096, 4), dtype=np.float32)
    pts[:, :2] = rng.uniform(-37.0, 41.0, (4096, 2)).astype(np.float32)
    got = _oracle_noise(oracle, 0, pts)
    want = np.array([_perlin2(float(p[0]), float(p[1])) for p in pts])
    assert np.abs(got - want).max() < 2e-6
    assert got.min() < -0.5 and got.max() > 0.5 and np.abs(got).max() <= 1.0 + 1e-6   # the scale the reference's constants normalise to
    # gradient noise is 0 on the lattice
    lattice = np.zeros((64, 4), dtype=np.float32)
    lattice[:, 0] = np.arange(64) - 30
    lattice[:, 1] = (np.arange(64) * 7) % 13 - 6
    assert np.array_equal(_oracle_noise(oracle, 0, lattice), np.zeros(64, dtype=np.float32))
    # a period repeats the pattern
    pts[:, 2:] = [4.0, 8.0]
    a = _oracle_noise(oracle, 0, pts)
    shifted = pts.copy()
    shifted[:, 0] += 4.0
    shifted[:, 1] -= 8.0
    b = _oracle_noise(oracle, 0, shifted)
    assert np.abs(a - b).max() < 1e-4
    want = np.array([_perlin2(float(p[0]), float(p[1]), (4, 8)) for p in pts])
    assert np.abs(a - want).max() < 2e-6


def test_cellular_noise_against_a_second_implementation(oracle):
    rng = np.random.default_rng(11)
    pts = np.zeros((4096, 4), dtype=np.float32)
    pts[:, :2] = rng.uniform(-19.0, 23.0, (4096, 2)).astype(np.float32)
    got = _oracle_noise(oracle, 2, pts)
    want = np.array([_worley2(float(p[0]), float(p[1])) for p in pts])
    assert np.abs(got - want).max() < 2e-6
    assert got.min() >= 0.0 and got.max() < 2.0
    # three dimensions: the squared distance to the nearest feature point is bounded by the cell's diagonal and continuous
    p3 = np.zeros((2048, 4), dtype=np.float32)
    p3[:, :3] = rng.uniform(-9.0, 9.0, (2048, 3)).astype(np.float32)
    d = _oracle_noise(oracle, 3, p3)
    assert d.min() >= 0.0 and d.max() < 3.0
    q3 = p3.copy()
    q3[:, :3] += 1e-4
    assert np.abs(_oracle_noise(oracle, 3, q3) - d).max() < 2e-3
    g3 = _oracle_noise(oracle, 1, p3)
    assert np.abs(g3).max() <= 1.0 + 1e-6 and g3.std() > 0.1
    assert np.abs(_oracle_noise(oracle, 1, q3) - g3).max() < 2e-3


def test_octave_weights(oracle):
    """std.math.pow as Noise.evaluate1 calls it (procedural_noise.zig:64, 80): amplitude^attenuation for amplitudes 1, 1/2, 1/4 ..."""
    pts = np.zeros((64, 4), dtype=np.float32)
    k = 0
    for level in range(8):
        for att in (0.0, 1.0, 0.5, 2.0, 0.7, 1.3, 3.0, -1.0):
            pts[k, 0] = 0.5 ** level
            pts[k, 1] = att
            k += 1
    got = _oracle_noise(oracle, 4, pts)
    want = np.power(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64))
    assert np.allclose(got, want, rtol=3e-7, atol=0)
    exact = (pts[:, 1] == 0.0) | (pts[:, 1] == 1.0) | (pts[:, 1] == 2.0) | (pts[:, 1] == 3.0) | (pts[:, 1] == -1.0) | (pts[:, 0] == 1.0)
    assert np.array_equal(got[exact], want[exact].astype(np.float32))


def _oracle_image(oracle, engine, w, h, spp):
    sd = engine.lib.su_ext_scene_desc().contents
    vd = engine.lib.su_ext_view_desc().contents
    film = oracle_render(oracle, sd, vd, spp)
    return oracle_resolve(oracle, vd, film)


def test_the_oracle_renders_them(product_lib, oracle):
    """A checker floor far from the camera tends to the mean of its two colours (the analytic filter), near the camera it shows both;
    noise moves a surface's colour around its mean; Mix with weight 0 / 1 is one of its children."""
    def image(ov, spp=4):
        e = zyg_amd.Engine(product_lib)
        try:
            scenes.cornell(e, 64, 48, spp, material="substitute", overrides=ov, integrator="PTMIS")
            e.compile()
            return _oracle_image(oracle, e, 64, 48, spp)
        finally:
            e.release()
    base = image({"floor": S((0.5, 0.5, 0.5))})
    coarse = image({"floor": S(checker(2.0, (0.95, 0.95, 0.95), (0.05, 0.05, 0.05)))})
    fine = image({"floor": S(checker(4000.0, (0.95, 0.95, 0.95), (0.05, 0.05, 0.05)))})
    assert np.isfinite(coarse).all() and np.isfinite(fine).all()
    assert rms(coarse, base) > 3e-3
    # 4000 squares across the floor are far below a pixel: the filtered checker is the 0.5 grey of `base`
    assert rms(fine, base) < rms(coarse, base) * 0.35
    red, blue = image({"floor": S((0.9, 0.2, 0.2))}), image({"floor": S((0.2, 0.2, 0.9))})
    mix0 = image({"floor": S({"Mix": {"a": [0.9, 0.2, 0.2], "b": [0.2, 0.2, 0.9], "weight": 0.0}})})
    mix1 = image({"floor": S({"Mix": {"a": [0.9, 0.2, 0.2], "b": [0.2, 0.2, 0.9], "weight": 1.0}})})
    # weight < r picks a: a weight of 0 is a (wherever r > 0), 1 is b; the draw shifts the later sampler dimensions, so the
    # pictures agree in the mean only
    d = lambda im: float((im[..., 0] - im[..., 2]).mean())  # noqa: E731
    gap = abs(d(red) - d(blue))
    assert gap > 0.01 and abs(d(mix0) - d(red)) < 0.25 * gap and abs(d(mix1) - d(blue)) < 0.25 * gap
    n = image({"back": S(noise(scale=[6, 6, 6], levels=3, attenuation=1.0)), "floor": S(noise(sampler={"texcoord": "ObjectPos"}, scale=[3, 3, 3], **{"class": "Cellular"}))})
    assert np.isfinite(n).all() and rms(n, base) > 3e-3


# ---------------------------------------------------------------- on the device
def _both(engine, oracle, w, h, spp, **kw):
    from test_gpu_parity import _check, _render_both
    out = _render_both(engine, oracle, w, h, spp, **kw)
    _check(*out, spp)
    return out


_TRI = {"texcoord": "Triplanar"}
_OBJ = {"texcoord": "ObjectPos"}


def _surfaces():
    """Every procedural kind, over the uv of Rectangles and meshes, triplanar and in object space, as colour, weight and normal map."""
    detail = {"DetailNormal": {"base": noise(scale=[5, 5, 5], levels=2), "detail": noise(scale=[23, 23, 23], absolute=True, sampler=_TRI)}}
    return {
        "floor": S(checker(6.0, (0.9, 0.85, 0.8), (0.15, 0.1, 0.1)), roughness=noise(scale=[7, 5, 1], levels=3, attenuation=0.6, ratio=0.45, transition=0.3)),
        "back": S({"Mix": {"a": checker(3.0, (0.1, 0.6, 0.2), (0.8, 0.8, 0.2)), "b": [0.25, 0.3, 0.7], "weight": noise(scale=[3, 3, 3], transition=0.1)}},
                  roughness=0.4, normal=noise(scale=[11, 11, 11], levels=2, attenuation=1.0)),
        "left": S(checker(5.0, (0.8, 0.2, 0.2), (0.9, 0.7, 0.6), sampler=_TRI), metallic={"Max": {"a": 0.1, "b": noise(scale=[4, 4, 4], sampler=_TRI, **{"class": "Cellular"})}}),
        "right": S({"ChannelMix": {"source": noise(scale=[9, 9, 9], sampler=_OBJ), "channels": [[0.2, 0, 0], [0.9, 0, 0], [0.3, 0, 0]]}},
                   roughness={"Mul": {"a": 0.8, "b": noise(scale=[2, 6, 1], period=[2, 6, 0])}}),
        "tall": S(noise(scale=[5, 5, 5], levels=4, attenuation=1.7, sampler=_OBJ), roughness=0.25, metallic=1.0, normal=noise(scale=[13, 13, 13], sampler=_OBJ)),
        "short": S(checker(4.0, (0.9, 0.9, 0.9), (0.2, 0.3, 0.8)), roughness=0.5, normal=detail,
                   coating={"thickness": 0.05, "normal": noise(scale=[17, 17, 17], sampler=_TRI), "roughness": noise(scale=[3, 3, 3], invert=True),
                            "scale": noise(scale=[2, 2, 2], ratio=0.3, transition=0.05)}),
        "sphere_a": S(checker(8.0, (0.9, 0.5, 0.1), (0.1, 0.1, 0.1)), roughness=noise(scale=[6, 6, 6], sampler=_TRI), normal=noise(scale=[9, 9, 9], **{"class": "Cellular"})),
        "sphere_b": S(noise(scale=[4, 4, 4], sampler=_OBJ, **{"class": "Cellular"}), roughness=0.6,
                      anisotropy_rotation=noise(scale=[3, 3, 3]), anisotropy=0.5, translucency={"Mix": {"a": 0.0, "b": 0.7, "weight": 0.3}}),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["PTMIS", "PTDL", "PT"])
def test_procedural_textures_on_the_device(engine, oracle, integrator):
    """wf_textures (dprocedural.hpp) against the oracle, bit for bit: Checker with its analytic filter (the pixel's footprint through
    the camera's ray differentials and the surface's position differentials: Rectangles, meshes, Spheres, the triplanar form), gradient and
    cellular Noise in two and three dimensions with octaves, as a normal map by finite differences, ChannelMix / Max / Mul / DetailNormal
    over children, and Mix, whose sampler draws sit between the vertex' stochastic_r and its light samples."""
    out = _both(engine, oracle, 96, 72, 4, integrator=integrator, material="substitute", spheres=True, overrides=_surfaces())
    engine.release()
    engine.__init__(engine.lib)
    plain = _both(engine, oracle, 96, 72, 4, stats=False, integrator=integrator, material="substitute", spheres=True)
    assert rms(out[1], plain[1]) > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["PTMIS", "PT"])
def test_procedural_textures_next_to_glass(engine, oracle, integrator):
    """The POOL kernels (split paths: the vertices of a camera sample share its samplers) read the same values: wf_textures runs per
    round between wf_shade and wf_light; Glass takes a procedural roughness, specular weight and normal."""
    glass = {"box": scenes.glass(1.5, (0.8, 0.9, 1.0), 0.5, roughness=noise(scale=[5, 5, 5], ratio=0.7, transition=0.3), normal=noise(scale=[8, 8, 8], sampler=_TRI)),
             "sphere": scenes.glass(1.45, roughness=0.1, specular={"Mix": {"a": 0.2, "b": 1.0, "weight": noise(scale=[4, 4, 4], sampler=_OBJ)}},
                                    normal=noise(scale=[6, 6, 6], sampler=_OBJ, **{"class": "Cellular"}))}
    ov = {k: v for k, v in _surfaces().items() if k in ("floor", "back", "tall", "left")}
    _both(engine, oracle, 80, 60, 4, integrator=integrator, material="substitute", glass_objects=glass, overrides=ov)


@pytest.mark.gpu
@pytest.mark.parametrize("glass", [False, True])
def test_procedural_textures_with_a_moving_camera(product_lib, oracle, glass):
    """The footprint comes from the camera where it was at the vertex' time (context.zig:140): the path carries its time when the
    camera has keyframes and the scene a procedural texture; next to Glass the POOL kernels' camera samples do."""
    from ctypes import POINTER, c_float
    with zyg_amd.Engine(product_lib) as e:
        ov = {"floor": S(checker(9.0, (0.9, 0.9, 0.9), (0.1, 0.1, 0.1))), "back": S((0.5, 0.5, 0.5), normal=noise(scale=[12, 12, 12]))}
        more = {"glass_objects": {"sphere": scenes.glass(1.5, roughness=noise(scale=[5, 5, 5]))}} if glass else {}
        ids = scenes.cornell(e, 96, 64, 4, integrator="PTMIS", material="substitute", overrides=ov, **more)
        base = scenes.trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0.0, 1.0, -3.9))
        for k, off in enumerate((-0.3, 0.3)):
            m = np.array(base, dtype=np.float32).reshape(16).copy()
            m[12] += off
            assert e.lib.su_prop_set_transformation_frame(ids["camera"], k, m.ctypes.data_as(POINTER(c_float))) == 0
        e.start_frame(0)
        e.render_iterations(4)
        e.synchronize()
        film = e.download_film(96, 64)
        sd = e.lib.su_ext_scene_desc().contents
        vd = e.lib.su_ext_view_desc().contents
        assert np.array_equal(film, oracle_render(oracle, sd, vd, 4))


@pytest.mark.gpu
def test_procedural_textures_in_the_relaxed_units(engine, oracle):
    """The benched arithmetic (option precise = 0) within north_star's tolerance of the oracle: RMS <= 1e-3 of the resolved image."""
    w, h, spp = 256, 192, 256
    scenes.cornell(engine, w, h, spp, integrator="PTMIS", material="substitute", spheres=True, overrides=_surfaces())
    engine.start_frame(0)
    engine.set_option("precise", 0)
    assert 1 == engine.get_option("relaxed_units")
    engine.render_iterations(spp)
    engine.synchronize()
    film = engine.download_film(w, h)
    assert np.isfinite(film).all()
    sd = engine.lib.su_ext_scene_desc().contents
    vd = engine.lib.su_ext_view_desc().contents
    cpu = oracle_render(oracle, sd, vd, spp, threads=16)
    r = rms(oracle_resolve(oracle, vd, film), oracle_resolve(oracle, vd, cpu))
    print(f"procedural textures, relaxed units vs oracle: RMS {r:.3e}")
    assert r <= 1e-3
