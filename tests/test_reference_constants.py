"""The oracle's and the product's constants against the reference's, by machine.

tests/golden/reference_constants.json and ggx_lut_full.json are the literals of the cited reference line ranges, parsed by
tests/golden/audit_constants.py.  Every one of them must appear as a literal in the oracle's and in the product's source
that implements that piece (so a mistyped digit on either side fails here), the GGX tables the product generates are compared
entry by entry with all 9 504 reference entries, and a few are checked through behaviour (metal presets through the material
compiler, the tonemappers through the oracle's resolve)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = json.load(open(os.path.join(GOLD, "reference_constants.json")))

NUM = re.compile(r"(?<![\w.])(?:\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+|\.\d+(?:[eE][-+]?\d+)?)")

ORACLE, PRODUCT = "oracle", "zyg_amd/csrc"
# group -> the files of the oracle and of the product that implement it (None: that side takes it ready-made from the other:
# the filter table and the compiled materials reach the oracle through the view / scene descriptors the host fills)
WHERE = {
    "aces_srgb_to_ap1": (["oracle/texture.hpp"], ["zyg_amd/csrc/hip/dscene.hpp", "zyg_amd/csrc/host/hmath.hpp"]),
    "aces_ap1_to_srgb": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "aces_ap1_to_rrt_sat": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "aces_odt_sat_to_srgb": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "aces_rrt_and_odt": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "agx_inset": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "agx_range": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "agx_outset": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "agx_contrast": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "pbr_neutral": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]),
    "blackman": (None, ["zyg_amd/csrc/host/engine.cpp"]),
    "mitchell": (None, ["zyg_amd/csrc/host/engine.cpp"]),
    "ray_offset": (["oracle/scene.hpp"], ["zyg_amd/csrc/hip/dmath.hpp"]),
    # the oracle reads max_split_depth from the compiled tree (ZygpuLightTree), which the host builder fills
    "light_tree": (None, ["zyg_amd/csrc/hip/dlight.hpp", "zyg_amd/csrc/host/light_tree_builder.hpp"]),
    "primitive_tree": (None, ["zyg_amd/csrc/host/light_tree_builder.hpp"]),
    "low_threshold": (["oracle/light.hpp"], ["zyg_amd/csrc/hip/dlight.hpp"]),
    "russian_roulette": (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/wavefront.hpp"]),
    "ggx_min_roughness": (["oracle/material.hpp"], ["zyg_amd/csrc/hip/dmaterial.hpp"]),
    "metal_presets": (None, ["zyg_amd/csrc/host/engine.cpp"]),
    "dot_min": (["oracle/omath.hpp"], ["zyg_amd/csrc/hip/dmath.hpp"]),
    "rainbow": (["oracle/material.hpp"], ["zyg_amd/csrc/hip/dmaterial.hpp"]),
    "glass_dispersion": (["oracle/material.hpp"], ["zyg_amd/csrc/hip/dmaterial.hpp"]),
    "sample_normal": (["oracle/material.hpp"], ["zyg_amd/csrc/hip/dmaterial.hpp"]),
    # the host compiles the material: the oracle and the device read ZygpuMaterial
    "flakes": (None, ["zyg_amd/csrc/host/engine.cpp"]),
    "coating_defaults": (None, ["zyg_amd/csrc/host/engine.cpp"]),
}


def _literals(paths):
    vals = set()
    for p in paths:
        text = open(os.path.join(ROOT, p)).read()
        text = "\n".join(line.split("//")[0] for line in text.split("\n"))
        for m in NUM.finditer(text):
            vals.add(np.float32(abs(float(m.group(0).rstrip("f")))))
        for m in re.finditer(r"(?<![\w.])(\d+)(?![\w.])", text):  # plain integers (MaxSplitDepth = 10 ...)
            vals.add(np.float32(int(m.group(1))))
    return vals


def test_every_group_has_a_home():
    assert set(WHERE) | {"agx_look"} == set(REF)  # agx_look: only the Substitute look is implemented (checked below)
    agx_look = REF["agx_look"]["values"]
    assert 1.1 in agx_look and 1.05 in agx_look  # the Substitute look the take loader picks (take_loader.zig:268-270)


@pytest.mark.parametrize("group", sorted(WHERE))
def test_constants_are_the_references(group):
    vals = [np.float32(abs(v)) for v in REF[group]["values"]]
    oracle_files, product_files = WHERE[group]
    for side, files in (("oracle", oracle_files), ("product", product_files)):
        if files is None:
            continue
        have = _literals(files)
        missing = [float(v) for v in vals if v not in have]
        assert not missing, f"{group} ({REF[group]['source']}): the {side} ({files}) lacks {missing}"


@pytest.mark.parametrize("path", ["oracle/material.hpp", "zyg_amd/csrc/hip/dmaterial.hpp"])
def test_rainbow_table_in_order(path):
    """The 32 x 3 entries of the Rainbow spectrum (rainbow_integral.zig:8-41), in the reference's order, on both sides."""
    ref = REF["rainbow"]["values"][2:]  # after Wavelength_start / Wavelength_end
    assert 128 == len(ref)
    ref_rgb = [v for i, v in enumerate(ref) if i % 4 != 3]
    text = open(os.path.join(ROOT, path)).read()
    body = re.search(r"kRainbow\[kRainbowBands\]\[3\] = \{(.*?)\}\};", text, re.S).group(1)
    have = [float(x) for x in re.findall(r"(\d+\.\d+)f", body)]
    assert have == ref_rgb


def test_agx_looks_other_than_substitute_are_not_claimed():
    """only the Substitute look (slope 1, power 1.1, saturation 1.05) is implemented: take_loader.zig:268-270 picks it"""
    look = REF["agx_look"]["values"]
    assert look[:4] == [0.2126, 0.7152, 0.072, 0.0] and look[4:7] == [1.0, 1.1, 1.05]
    for files in (["oracle/oracle.cpp"], ["zyg_amd/csrc/hip/zygpu.hip"]):
        have = _literals(files)
        for v in look[:7]:
            assert np.float32(v) in have


def test_ggx_tables_entry_by_entry(product_lib):
    """all 9 504 entries of material/ggx_integral.zig against the tables the product generates (ggx_lut.hpp integrates them
    itself instead of embedding the reference's numbers): the reference prints 8 decimals of an f32 Monte Carlo / quadrature
    result, the product integrates in double precision: they agree to a few 1e-7"""
    full = json.load(open(os.path.join(GOLD, "ggx_lut_full.json")))["tables"]
    lib = product_lib
    lib.su_ext_ggx_lut.restype = ctypes.c_uint32
    lib.su_ext_ggx_lut.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    worst = {}
    for which, name in enumerate(("E_m", "E_m_avg", "E", "E_avg", "E_s")):
        ref = np.asarray(full[name], dtype=np.float64)
        n = lib.su_ext_ggx_lut(which, None, 0)
        assert n == ref.size, name
        mine = np.zeros(n, dtype=np.float32)
        assert lib.su_ext_ggx_lut(which, mine.ctypes.data, n) == n
        worst[name] = float(np.abs(mine.astype(np.float64) - ref).max())
        assert worst[name] <= 6e-7, (name, worst[name], int(np.abs(mine - ref).argmax()))
    assert max(worst.values()) > 0.0  # generated, not copied


def test_metal_presets_through_the_material_compiler(product_lib):
    """Substitute {"metal_preset": name}: the compiled material's colour is the Fresnel conductor reflectance at normal
    incidence of the reference's ior / absorption triples (material_provider.zig:277-280, fresnel.zig:45-57)"""
    import zyg_amd
    vals = REF["metal_presets"]["values"]
    names = ["Aluminium", "Gold", "Silver", "Iron"]
    with zyg_amd.Engine(product_lib) as e:
        e.camera(16, 16)
        for k, name in enumerate(names):
            ior = np.array(vals[8 * k:8 * k + 3], dtype=np.float32)
            absorb = np.array(vals[8 * k + 4:8 * k + 7], dtype=np.float32)
            m = e.material(json.dumps({"rendering": {"Substitute": {"metal_preset": name, "roughness": 0.2}}}))
            e.compile()
            mat = e.lib.su_ext_scene_desc().contents.materials[m]
            # fresnel.conductor at wo.h = 1 (fresnel.zig:45-57)
            expect = ((ior - 1.0) ** 2 + absorb ** 2) / ((ior + 1.0) ** 2 + absorb ** 2)
            np.testing.assert_allclose(np.array(list(mat.color_a)), expect, rtol=2e-6, err_msg=name)
            assert mat.metallic == 1.0
