"""The Sky entity's bake (reference core/sky/sky.zig:172-269, sky_model.zig:32-137; zyg_amd/csrc/host/sky_bake.hpp over the product's own
evaluator of the Prague model, prague_sky.hpp): with a dataset in the model's file layout among the mounts, the sun's radiance table
and the canopy image are the model's, not the procedural stand-in.  The real dataset does not ship with the reference:
tools/make_sky_dataset.py writes a small synthetic one.  The baked tables are checked against a numpy restatement of the bake that asks
the REFERENCE's library for the model's values (oracle/_ref/libarpraguesky.so: the C the reference links, compiled where it lies by
oracle/Makefile; the product does not load it: tests/test_prague_sky.py)."""
import ctypes
import gzip
import os
import re
import sys

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
PRAGUE = os.path.join(ROOT, "oracle", "_ref", "libarpraguesky.so")
needs_library = pytest.mark.skipif(not os.path.exists(PRAGUE), reason="oracle/_ref/libarpraguesky.so is compiled from the reference's checkout (make -C oracle ref)")
SKY = {"parameters": {"sun": {"rotation": [-55, 20, 0]}, "visibility": 80.0, "albedo": 0.3}}
f32 = np.float32


class Pcg:  # base/random/generator.zig:1-47
    def __init__(self, state, sequence):
        self.state, self.inc = 0, ((sequence << 1) | 1) & (2 ** 64 - 1)
        self.uint()
        self.state = (self.state + state) & (2 ** 64 - 1)
        self.uint()

    def uint(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & (2 ** 64 - 1)
        xrs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xrs >> rot) | (xrs << ((-rot) & 31))) & 0xFFFFFFFF

    def float(self):
        return np.array([(self.uint() & 0x007FFFFF) | 0x3F800000], dtype=np.uint32).view(np.float32)[0] - f32(1.0)


def _cie_bins():
    """discrete.zig:21-66 over the tabulated matching functions (the generated header's numbers): the mean of each over a 30 nm bin"""
    text = open(os.path.join(ROOT, "zyg_amd", "csrc", "host", "cie_xyz_1931.hpp")).read()
    tables = [np.array([float(v) for v in re.findall(r"([0-9.e+-]+)f,", m)], dtype=np.float64) for m in re.findall(r"\[kNum\] = \{(.*?)\};", text, re.S)]
    assert [len(t) for t in tables] == [471, 471, 471]
    wl = np.arange(360, 831, dtype=np.float64)
    bins = np.zeros((12, 3))
    for i in range(12):
        a, b = 380.0 + 30.0 * i, 410.0 + 30.0 * i
        sel = (wl >= a) & (wl <= b)
        for c in range(3):
            bins[i, c] = np.trapezoid(tables[c][sel], wl[sel]) / 30.0
    return bins


def _model(dataset, elevation, visibility, albedo):
    lib = ctypes.CDLL(PRAGUE)
    raw = dataset + ".raw"
    open(raw, "wb").write(gzip.open(dataset).read())
    lib.arpragueskymodelground_state_alloc_init.restype = ctypes.c_void_p
    lib.arpragueskymodelground_state_alloc_init.argtypes = [ctypes.c_char_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.arpragueskymodelground_sky_radiance.restype = lib.arpragueskymodelground_solar_radiance.restype = ctypes.c_double
    lib.arpragueskymodelground_sky_radiance.argtypes = [ctypes.c_void_p] + [ctypes.c_double] * 4
    lib.arpragueskymodelground_solar_radiance.argtypes = [ctypes.c_void_p] + [ctypes.c_double] * 2
    state = lib.arpragueskymodelground_state_alloc_init(raw.encode(), float(elevation), float(visibility), float(albedo))
    assert state
    return lib, state


def _ap1(radiance_bins, cie):
    xyz = (30.0 * radiance_bins[:, None] * cie).sum(axis=0)
    m = np.array([[1.66058533, -0.31529556, -0.24150933], [-0.65992606, 1.60839147, 0.01729859], [0.00900257, -0.00356688, 0.91364331]])
    return m @ xyz


def _spectral(evaluate, rng):
    """sky_model.zig:88-107: per bin 24 golden-ratio wavelengths from one random offset"""
    out = np.zeros(12)
    for i in range(12):
        x = rng.float()
        acc = 0.0
        for _ in range(24):
            acc += evaluate(380.0 + (i + float(x)) * 30.0) / 24.0
            x = f32(x + f32(0.618033988749894))
            if x >= 1.0:
                x = f32(x - f32(1.0))
        out[i] = acc
    return out


@needs_library
def test_bake_against_a_numpy_restatement(product_lib, tmp_path):
    import make_sky_dataset
    dataset = make_sky_dataset.write(str(tmp_path))
    lines = []
    callback = zyg_amd.LOG_CALLBACK(lambda level, text: lines.append(text.decode()))
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_register_log.argtypes = [zyg_amd.LOG_CALLBACK]
        e.lib.su_register_log(callback)
        try:
            e.lib.su_mount(str(tmp_path).encode())
            scenes.outdoor(e, 32, 18, 1, integrator="PTMIS", material="substitute", sky_entity=SKY)
            sd, _ = e.compile()
            assert not any("stand-in" in l for l in lines), lines
            sun = np.ctypeslib.as_array(sd.sun_radiance, shape=(sd.num_sun_radiance * 4,)).reshape(-1, 4).copy()
            img = [sd.images[i] for i in range(sd.num_images) if sd.images[i].width == 1024 and sd.images[i].height == 1024]
            assert 1 == len(img)
            sky = np.ctypeslib.as_array(ctypes.cast(img[0].pixels, ctypes.POINTER(ctypes.c_float)), shape=(1024, 1024, 3)).copy()
            props = [sd.trafos[i] for i in range(sd.num_props)]
        finally:
            e.lib.su_register_log(zyg_amd.LOG_CALLBACK())
    # the sun's rotation [-55, 20, 0] degrees: its z axis' y component is the sine of the (negative) elevation angle
    sin_el = [t.r[2][1] for t in props if abs(t.r[0][3] - 0.0046548266) < 1e-9][0]
    elevation = -np.arcsin(f32(sin_el))
    lib, state = _model(dataset, elevation, SKY["parameters"]["visibility"], SKY["parameters"]["albedo"])
    cie = _cie_bins()
    # the sun: one generator runs through all 1024 entries, 12 draws each (sky.zig:203-214)
    rng = Pcg(0, 0)
    for i in range(1024):
        v = f32(i) / f32(1023)
        wi_dot_z = np.sin(f32(elevation) + (f32(2) * v - f32(1)) * f32(0.0046548266))
        theta = float(np.arccos(np.clip(f32(wi_dot_z), 0, 1)))
        if i in (0, 1, 511, 1023):
            want = _ap1(_spectral(lambda wl: lib.arpragueskymodelground_solar_radiance(state, theta, wl), rng), cie)
            np.testing.assert_allclose(sun[i, :3], want, rtol=3e-5)
        else:
            for _ in range(12):
                rng.uint()
    assert np.all(sun[:, :3] > 0) and np.all(sun[:, 3] == 0)
    # the canopy image: texel (x, y) of the left half from its own generator, the right half its mirror image (sky.zig:284-322)
    sun_dir = np.array([0.0, sin_el, np.sqrt(1.0 - sin_el * sin_el)])
    d = np.sqrt(sun_dir[0] ** 2 + sun_dir[2] ** 2)
    shadow_dir = np.array([(-sun_dir[0] / d) * sun_dir[1], np.sqrt(1.0 - sun_dir[1] ** 2), (-sun_dir[2] / d) * sun_dir[1]])
    for x, y in ((300, 512), (511, 200), (40, 600), (100, 100)):
        u, v = (x + 0.5) / 1024.0, (y + 0.5) / 1024.0
        dx, dy = 2 * u - 1, 2 * v - 1
        l, eps = np.hypot(dx, dy), 3.5 / 1024.0
        if l > 1 + eps:
            assert np.all(sky[y, x] == 0) and np.all(sky[y, 1023 - x] == 0)
            continue
        if l >= 1 - eps:
            dx, dy = dx / (l + eps), dy / (l + eps)
        lon, col = np.arctan2(-dy, dx), np.hypot(dx, dy) * np.pi / 2
        direction = np.array([np.sin(col) * np.cos(lon), np.sin(col) * np.sin(lon), np.cos(col)])
        wi = np.array([direction[0], direction[2], -direction[1]])
        wi /= np.linalg.norm(wi)
        theta = float(np.arccos(np.clip(wi[1], -1, 1)))
        gamma = float(np.arccos(np.clip(-np.dot(wi, sun_dir), -1, 1)))
        shadow = float(np.arccos(np.clip(np.dot(wi, shadow_dir), -1, 1)))
        want = _ap1(_spectral(lambda wl: lib.arpragueskymodelground_sky_radiance(state, theta, gamma, shadow, wl), Pcg(0, y * 1024 + x)), cie)
        np.testing.assert_allclose(sky[y, x], want, rtol=2e-4, atol=1e-7)
        assert np.array_equal(sky[y, x], sky[y, 1023 - x])
    assert sky[512, 300, 2] > sky[512, 300, 0] > 0  # a blue sky (the synthetic dataset's Rayleigh term)


def test_without_the_dataset_the_stand_in_is_flagged(product_lib, tmp_path):
    lines = []
    callback = zyg_amd.LOG_CALLBACK(lambda level, text: lines.append((level, text.decode())))
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_register_log.argtypes = [zyg_amd.LOG_CALLBACK]
        e.lib.su_register_log(callback)
        try:
            e.lib.su_mount(str(tmp_path).encode())  # (a mount without the dataset)
            scenes.outdoor(e, 32, 18, 1, integrator="PTMIS", material="substitute", sky_entity=SKY)
            assert e.lib.su_start_frame(0) in (0, -1)  # the warning is posted by su_start_frame's compile (-1 here: no GPU)
        finally:
            e.lib.su_register_log(zyg_amd.LOG_CALLBACK())
    assert any(1 == level and "stand-in" in text and "SkyModelDataset" in text for level, text in lines), lines


@pytest.mark.gpu
def test_render_with_the_baked_sky(engine, oracle, tmp_path):
    """The device consumes the baked tables like any others: a small frame of the outdoor scene lit by the synthetic dataset's sky
    and sun, device film == the oracle's."""
    import make_sky_dataset
    from util import oracle_render
    make_sky_dataset.write(str(tmp_path))
    engine.lib.su_mount(str(tmp_path).encode())
    w, h, spp = 96, 54, 4
    scenes.outdoor(engine, w, h, spp, integrator="PTMIS", material="substitute", showcase=True, max_depth=8, sky_entity=SKY, box_lift=1e-3)
    engine.start_frame(0)
    engine.render_iterations(spp)
    engine.synchronize()
    film = engine.download_film(w, h)
    sd = engine.lib.su_ext_scene_desc().contents
    vd = engine.lib.su_ext_view_desc().contents
    assert np.isfinite(film).all() and film[..., :3].max() > 0
    assert np.array_equal(film, oracle_render(oracle, sd, vd, spp))
