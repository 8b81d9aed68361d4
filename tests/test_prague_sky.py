"""The Prague sky model's own evaluator (zyg_amd/csrc/host/prague_sky.hpp, through su_ext_sky_model_eval) against the reference's:
oracle/_ref/libarpraguesky.so is the authors' C that the reference links (thirdparty/arpraguesky/ArPragueSkyModelGround.c:738-825,
1094-1134; interface ArPragueSkyModelGround.h:170-216), compiled where it lies by oracle/Makefile (`make -C oracle ref`): the one
oracle of this repository that IS the reference.  Queries: random (theta, gamma, shadow, wavelength, distance) for model states over,
between, at and beyond the dataset's grid values (negative solar elevations take the shadow angle), on the synthetic dataset of
tools/make_sky_dataset.py and on a second one with other grid sizes (rank 12 like the real file, two altitudes)."""
import ctypes
import gzip
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = os.path.join(ROOT, "oracle", "_ref", "libarpraguesky.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libarpraguesky.so is compiled from the reference's checkout (make -C oracle ref)")

STATES = [  # solar elevation (degrees), visibility (km), albedo
    (35.0, 80.0, 0.3), (0.0, 20.0, 0.0), (-3.0, 45.0, 1.0), (-4.2, 60.0, 0.5), (90.0, 150.0, 0.25), (10.0, 10.0, 0.7), (62.5, 200.0, 0.999999),
    (29.9999999, 59.9999999, 1e-7), (-10.0, 131.8, 0.1), (5.0, 27.6, 0.62),
]


def _reference(path_raw, elevation, visibility, albedo):
    lib = ctypes.CDLL(REF)
    lib.arpragueskymodelground_state_alloc_init.restype = ctypes.c_void_p
    lib.arpragueskymodelground_state_alloc_init.argtypes = [ctypes.c_char_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.arpragueskymodelground_state_free.argtypes = [ctypes.c_void_p]
    for name, n in (("sky_radiance", 4), ("solar_radiance", 2), ("transmittance", 3)):
        f = getattr(lib, "arpragueskymodelground_" + name)
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_void_p] + [ctypes.c_double] * n
    state = lib.arpragueskymodelground_state_alloc_init(path_raw.encode(), elevation, visibility, albedo)
    assert state
    return lib, state


def _own(product_lib, dataset, state, what, queries):
    product_lib.su_ext_sky_model_eval.restype = ctypes.c_int32
    product_lib.su_ext_sky_model_eval.argtypes = [ctypes.c_char_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p,
                                                  ctypes.c_uint32, ctypes.c_void_p]
    q = np.ascontiguousarray(queries, dtype=np.float64)
    out = np.zeros(len(q), dtype=np.float64)
    assert 0 == product_lib.su_ext_sky_model_eval(dataset.encode(), state[0], state[1], state[2], what, q.ctypes.data, len(q), out.ctypes.data)
    return out


def _second_dataset(directory):
    """Another file of the model's layout: other grid sizes, rank 12, random smooth-ish numbers (half-precision values, zenith scales != 1)"""
    rng = np.random.default_rng(7)
    out = bytearray()
    axes = ([20.0, 27.6, 40.0, 59.4, 90.0, 131.8], [0.0, 0.5, 1.0], [0.0, 500.0], [-4.2, -2.0, 0.0, 5.0, 20.0, 45.0, 90.0])
    for values in axes:
        out += struct.pack("<i", len(values)) + np.asarray(values, "<f8").tobytes()
    channels, comps = 11, 5
    out += struct.pack("<idd", channels, 320.0, 40.0) + struct.pack("<i", comps)
    breaks = [np.sort(np.concatenate(([0.0, np.pi], rng.uniform(0.01, 3.1, n)))) for n in (12, 9, 5)]
    for b in breaks:
        out += struct.pack("<i", len(b)) + np.asarray(b, "<f8").tobytes()
    points = 6 * 3 * 2 * 7 * channels
    for _ in range(points):
        for _c in range(comps):
            out += rng.uniform(-0.2, 1.5, len(breaks[0])).astype("<f2").tobytes()
            out += struct.pack("<d", rng.uniform(0.5, 40.0))
            out += rng.uniform(0.0, 30.0, len(breaks[1])).astype("<f2").tobytes()
        out += rng.uniform(0.1, 1.2, len(breaks[2])).astype("<f2").tobytes()
    n_d, n_a, nvis, nalt, rank = 15, 7, 5, 2, 12
    out += struct.pack("<5i", n_d, n_a, nvis, nalt, rank)
    out += np.asarray([0.0, 1000.0], "<f4").tobytes() + np.asarray([20.0, 40.0, 60.0, 100.0, 131.8], "<f4").tobytes()
    out += rng.uniform(-0.3, 0.4, nalt * n_d * n_a * rank).astype("<f4").tobytes()
    out += rng.uniform(-0.3, 0.6, nvis * nalt * 11 * rank).astype("<f4").tobytes()
    path = os.path.join(directory, "second.dat.gz")
    with gzip.open(path, "wb", compresslevel=1) as f:
        f.write(bytes(out))
    return path


@pytest.mark.parametrize("which", ["synthetic", "second"])
def test_own_evaluator_against_the_reference_library(product_lib, tmp_path, which):
    import make_sky_dataset
    dataset = make_sky_dataset.write(str(tmp_path)) if "synthetic" == which else _second_dataset(str(tmp_path))
    raw = dataset + ".raw"
    open(raw, "wb").write(gzip.open(dataset).read())
    rng = np.random.default_rng(11)
    per_state = 1500
    total = 0
    for state_deg in STATES:
        state = (np.radians(state_deg[0]), state_deg[1], state_deg[2])
        lib, ref = _reference(raw, *state)
        theta = rng.uniform(0.0, np.pi, per_state)
        theta[:8] = [0.0, np.pi / 2, np.pi / 2 - 1e-9, np.pi / 2 + 1e-9, 1e-12, 3.14, 1.5707963705062866, 0.78539816]
        gamma, shadow = rng.uniform(0.0, np.pi, per_state), rng.uniform(0.0, np.pi, per_state)
        wavelength = rng.uniform(300.0, 780.0, per_state)  # (the model answers 320..760 nm, zero outside)
        wavelength[:6] = [320.0, 359.999999, 360.0, 759.9999, 760.0, 319.9999]
        distance = np.where(rng.uniform(size=per_state) < 0.3, 5.78960446186580977117855e76, 10.0 ** rng.uniform(0.0, 7.0, per_state))
        q_sky = np.stack([theta, gamma, shadow, wavelength], axis=1)
        q_sun = np.stack([theta, wavelength, np.zeros(per_state), np.zeros(per_state)], axis=1)
        q_tr = np.stack([theta, wavelength, distance, np.zeros(per_state)], axis=1)
        want_sky = np.array([lib.arpragueskymodelground_sky_radiance(ref, *q[:4]) for q in q_sky])
        want_sun = np.array([lib.arpragueskymodelground_solar_radiance(ref, q[0], q[1]) for q in q_sun])
        want_tr = np.array([lib.arpragueskymodelground_transmittance(ref, q[0], q[1], q[2]) for q in q_tr])
        lib.arpragueskymodelground_state_free(ref)
        for what, queries, want in ((0, q_sky, want_sky), (1, q_sun, want_sun), (2, q_tr, want_tr)):
            got = _own(product_lib, dataset, state, what, queries)
            assert np.isfinite(got).all()
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-300, err_msg=f"{which} state {state_deg} function {what}")
            total += len(want)
        assert want_sky.max() > 0 and want_sun.max() > 0 and 0 < want_tr.max() <= 1
    assert total >= 10000


def test_a_file_that_is_not_a_dataset_is_refused(product_lib, tmp_path):
    bad = os.path.join(str(tmp_path), "bad.dat.gz")
    with gzip.open(bad, "wb") as f:
        f.write(b"\x03\x00\x00\x00" + b"\x00" * 20)
    q = np.zeros((1, 4))
    out = np.zeros(1)
    product_lib.su_ext_sky_model_eval.restype = ctypes.c_int32
    product_lib.su_ext_sky_model_eval.argtypes = [ctypes.c_char_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p,
                                                  ctypes.c_uint32, ctypes.c_void_p]
    assert -1 == product_lib.su_ext_sky_model_eval(bad.encode(), 0.5, 50.0, 0.5, 0, q.ctypes.data, 1, out.ctypes.data)
    assert -1 == product_lib.su_ext_sky_model_eval(os.path.join(str(tmp_path), "absent").encode(), 0.5, 50.0, 0.5, 0, q.ctypes.data, 1, out.ctypes.data)


def test_nothing_of_the_reference_is_loaded_by_the_product():
    """the product executed a library built from the reference's thirdparty.  No dlopen / dlsym under zyg_amd/, no
    reference-built library beside libzyg_amd.so, and libzyg_amd.so does not link libdl for it."""
    for directory, _, files in os.walk(os.path.join(ROOT, "zyg_amd")):
        for f in files:
            if f.endswith((".hpp", ".cpp", ".hip", ".h", ".py")):
                text = open(os.path.join(directory, f), errors="replace").read()
                assert "dlopen" not in text and "dlsym" not in text, os.path.join(directory, f)
            assert "arpraguesky" not in f.lower(), os.path.join(directory, f)
