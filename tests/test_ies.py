"""IES photometric profiles (reference image/encoding/ies/ies_reader.zig:16-394, emittance.zig:20, 45-52, 73-103, material_provider.zig:412-433):
a Light material's `"profile": {"file": "lamp.ies"}`.  The host resamples the file's candela table into a square one-channel image over
the octahedral map of directions (zyg_amd/csrc/host/image_readers.hpp readIes); the light's radiance is multiplied by the image at the
direction from the light, in its own space, to the point it lights; the cone angle defaults to the widest lit texel's.

Without a GPU: the resampled image and the angle against a Python restatement of the reader, for the table forms the reader tells apart
(type C with one horizontal angle: rotationally symmetric; a quadrant mirrored three times; a half plane; a full turn, interpolated with
Catmull-Rom splines over 4 x 4 neighbours).  On the GPU: a Cornell box lit through such a profile, device film == the oracle's."""
import math

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

from util import oracle_render

f32 = np.float32


def _ies_text(vertical, horizontal, candela, gonio=1):
    lines = ["IESNA:LM-63-2002", "[TEST] synthetic", "[MANUFAC] none", "TILT=NONE",
             f"1 1000 1 {len(vertical)} {len(horizontal)} {gonio} 2 0.1 0.1 0.1", "1 1 50",
             " ".join(f"{v:g}" for v in vertical), " ".join(f"{h:g}" for h in horizontal)]
    for row in candela:
        lines.append(" ".join(f"{c:g}" for c in row))
    return "\r\n".join(lines) + "\r\n"


def _tables():
    v = [0, 10, 20, 30, 45, 60, 75, 90]
    sym = ([0, 15, 30, 45, 60, 90], [0], [[900, 850, 600, 300, 80, 0]])
    quad_h = [0, 30, 60, 90]
    quad = (v, quad_h, [[1000 - 9 * a - 2 * h for a in v] for h in quad_h])
    half_h = [0, 45, 90, 135, 180]
    half = (v, half_h, [[max(0.0, 700 - 7 * a + 1.5 * h) for a in v] for h in half_h])
    full_h = [0, 40, 80, 120, 160, 200, 240, 280, 320]
    full = ([0, 20, 40, 60, 80, 100, 120, 140, 160, 180], full_h,
            [[max(0.0, 500 + 300 * math.cos(math.radians(h)) - 3 * a) for a in range(0, 181, 20)] for h in full_h])
    return {"symmetric": sym, "quadrant": quad, "half": half, "full": full}


def _catmull(c, t):
    c = [f32(x) for x in c]
    t = f32(t)
    t2 = t * t
    a0 = f32(-0.5) * c[0] + f32(1.5) * c[1] - f32(1.5) * c[2] + f32(0.5) * c[3]
    a1 = c[0] - f32(2.5) * c[1] + f32(2.0) * c[2] - f32(0.5) * c[3]
    a2 = f32(-0.5) * c[0] + f32(0.5) * c[2]
    return a0 * t * t2 + a1 * t2 + a2 * t + c[1]


def _sample(vertical, horizontal, candela, gonio, rad_phi, rad_theta):
    """Data.sample (ies_reader.zig:31-206)"""
    va, ha = [f32(x) for x in vertical], [f32(x) for x in horizontal]
    flat = [f32(c) for row in candela for c in row]
    phi, theta = f32(math.degrees(rad_phi)), f32(math.degrees(rad_theta))
    vf, vb, hf, hb = va[0], va[-1], ha[0], ha[-1]
    vsym = nonsym = False
    if 1 == gonio:
        if 0.0 == hb:
            phi, vsym = f32(0.0), True
        elif 90.0 == hb:
            if 90.0 < phi < 180.0:
                phi = f32(180.0) - phi
            if 180.0 < phi < 270.0:
                phi = phi - f32(180.0)
            if 270.0 < phi < 360.0:
                phi = f32(360.0) - phi
            if phi < 0.0 or phi > 90.0 or theta < 0.0 or theta > 180.0:
                return f32(0.0)
        elif 180.0 == hb:
            if phi > 180.0:
                phi = f32(360.0) - phi
            if phi < 0.0 or phi > 180.0 or theta < 0.0 or theta > 180.0:
                return f32(0.0)
        elif 90.0 == hf and 270.0 == hb:
            if phi < 90.0 or phi > 270.0:
                phi = f32(180.0) - phi
            if phi < 90.0 or phi > 270.0:
                return f32(0.0)
        else:
            nonsym = True
    if vsym and (phi < hf or phi > hb):
        return f32(0.0)
    if theta < vf or theta > vb:
        return f32(0.0)
    below = lambda a, x: max(int(np.searchsorted(np.array(a, dtype=np.float32), x, side="left")) - 1, 0)
    hit, vit = below(ha, phi), below(va, theta)
    nv, nh = len(va), len(ha)
    if nonsym:
        if vit >= nv - 1:
            return f32(0.0)
        next_phi = ha[hit + 1] if hit < nh - 1 else ha[(hit + 1) % nh]
    else:
        if not vsym and (hit >= nh - 1 or vit >= nv - 1):
            return f32(0.0)
        next_phi = ha[hit + 1] if hit < nh - 1 else ha[hit]
    next_theta = va[vit + 1] if vit < nv - 1 else va[vit]
    d_theta = (theta - va[vit]) / (next_theta - va[vit])
    if vsym:
        return f32(np.float32(1.0) - d_theta) * flat[vit] + d_theta * flat[(vit + 1) % nv]
    d_phi = (phi - ha[hit]) / (next_phi - ha[hit])

    def off(x, i, b):
        y = x + i
        return 1 - y if y < 0 else (b - i if y >= b else y)
    rows = []
    for j in (-1, 0, 1, 2):
        h = off(hit, j, nh) * nv
        rows.append(_catmull([flat[off(vit, k, nv) + h] for k in (-1, 0, 1, 2)], d_theta))
    return _catmull(rows, d_phi)


def _oct_decode(ox, oy):
    v = [ox, oy, -1.0 + abs(ox) + abs(oy)]
    t = max(v[2], 0.0)
    v[0] += -t if v[0] > 0 else t
    v[1] += -t if v[1] > 0 else t
    n = math.sqrt(sum(x * x for x in v))
    return [x / n for x in v]


def _restated_image(vertical, horizontal, candela, gonio=1):
    steps = [abs(b - a) for a, b in zip(vertical, vertical[1:])] + [abs(b - a) for a, b in zip(horizontal, horizontal[1:])]
    res = int(360.0 / min(steps) + 0.5)
    most = max(c for row in candela for c in row)
    img = np.zeros((res, res), dtype=np.float32)
    for y in range(res):
        for x in range(res):
            d = _oct_decode(2.0 * ((x + 0.5) / res - 0.5), 2.0 * ((y + 0.5) / res - 0.5))
            lx, ly, lz = d[0], -d[2], -d[1]
            phi = math.atan2(lx, lz)
            phi = 2.0 * math.pi + phi if phi < 0 else phi
            value = float(_sample(vertical, horizontal, candela, gonio, phi, math.acos(max(-1.0, min(1.0, ly)))))
            img[y, x] = np.float16(min(max(value / most, 0.0), 1.0))
    return img


def _profile_material(path, angle=None):
    e = {"spectrum": [17.0, 12.0, 4.0], "value": 4.0, "profile": {"file": path}}
    if angle is not None:
        e["angle"] = angle
    import json
    return json.dumps({"rendering": {"Light": {"emittance": e}}})


@pytest.mark.parametrize("kind", ["symmetric", "quadrant", "half", "full"])
def test_profile_image_against_a_restatement(product_lib, tmp_path, kind):
    vertical, horizontal, candela = _tables()[kind]
    path = tmp_path / f"{kind}.ies"
    path.write_text(_ies_text(vertical, horizontal, candela), newline="")
    with zyg_amd.Engine(product_lib) as e:
        e.lib.su_mount(str(tmp_path).encode())
        scenes.cornell(e, 16, 12, 1)
        lamp = e.prop(scenes.RECTANGLE, [e.material(_profile_material(f"{kind}.ies"))])
        e.transformation(lamp, scenes.trafo((1, 0, 0), (0, 0, 1), (0.3, 0.3, 1), (0, 1.9, 0)))
        e.light(lamp)
        sd, _ = e.compile()
        mats = [sd.materials[i] for i in range(sd.num_materials) if sd.materials[i].profile_image != zyg_amd.NULL]
        assert 1 == len(mats)
        img = sd.images[mats[0].profile_image]
        got = np.ctypeslib.as_array(np.ctypeslib.ctypes.cast(img.pixels, np.ctypeslib.ctypes.POINTER(np.ctypeslib.ctypes.c_float)), shape=(img.height, img.width)).copy()
        cos_a = mats[0].cos_a
    want = _restated_image(vertical, horizontal, candela)
    assert got.shape == want.shape and got.max() > 0.9
    # (f32 trigonometry against Python's f64 may move a texel across an interval's end: all but a few texels agree to an f16 step)
    close = np.abs(got - want) <= 2e-3
    assert close.mean() >= 0.98, (kind, close.mean(), np.abs(got - want).max())
    # angleFromProfile (emittance.zig:73-103)
    res = want.shape[0]
    c = 1.0
    for y in range(res):
        for x in range(res):
            if got[y, x] > 0:
                c = min(c, -_oct_decode(2.0 * ((x + 0.5) / res - 0.5), 2.0 * ((y + 0.5) / res - 0.5))[2])
    assert abs(cos_a - math.cos(math.acos(max(c - 1.0 / res, -1.0)))) < 1e-5


def test_files_that_are_not_profiles_are_refused(product_lib, tmp_path):
    (tmp_path / "tilted.ies").write_text("IESNA91\r\nTILT=INCLUDE\r\n1 2 3\r\n", newline="")
    (tmp_path / "short.ies").write_text("IESNA91\r\nTILT=NONE\r\n1 1000 1 4 1 1 2 0 0 0\r\n1 1 50\r\n0 30 60\r\n", newline="")
    for name in ("tilted.ies", "short.ies", "absent.ies"):
        with zyg_amd.Engine(product_lib) as e:
            e.lib.su_mount(str(tmp_path).encode())
            scenes.cornell(e, 16, 12, 1)
            lamp = e.prop(scenes.RECTANGLE, [e.material(_profile_material(name))])
            e.transformation(lamp, scenes.trafo((1, 0, 0), (0, 0, 1), (0.3, 0.3, 1), (0, 1.9, 0)))
            assert e.lib.su_ext_compile(0) == -1, name


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["PTMIS", "PTDL", "PT"])
def test_light_through_a_profile(engine, oracle, tmp_path, integrator):
    """A small Rectangle lamp with a quadrant-table profile hangs in the Cornell box (whose own lights stay): light samples and rays
    that hit the lamp read the profile at the direction towards the lit point / the previous vertex; device film == the oracle's."""
    vertical, horizontal, candela = _tables()["quadrant"]
    (tmp_path / "lamp.ies").write_text(_ies_text(vertical, horizontal, candela), newline="")
    engine.lib.su_mount(str(tmp_path).encode())
    w, h, spp = 96, 72, 4
    scenes.cornell(engine, w, h, spp, integrator=integrator)
    lamp = engine.prop(scenes.RECTANGLE, [engine.material(_profile_material("lamp.ies"))])
    engine.transformation(lamp, scenes.trafo((1, 0, 0), (0, 0, 1), (0.4, 0.4, 1), (0.2, 1.6, 0.1)))
    engine.light(lamp)
    engine.start_frame(0)
    engine.render_iterations(spp)
    engine.synchronize()
    film = engine.download_film(w, h)
    sd = engine.lib.su_ext_scene_desc().contents
    vd = engine.lib.su_ext_view_desc().contents
    want = oracle_render(oracle, sd, vd, spp)
    assert np.isfinite(film).all() and film[..., :3].max() > 0
    assert np.array_equal(film, want)
