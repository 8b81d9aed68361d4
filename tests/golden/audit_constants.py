#!/usr/bin/env python3
"""Pins the numeric constants of the hot path to the reference by machine: every literal of the cited line ranges of
/root/reference is parsed into tests/golden/reference_constants.json (data: numbers, with the file:lines they come from),
and all 9 504 entries of the GGX albedo tables (material/ggx_integral.zig) into tests/golden/ggx_lut_full.json.
tests/test_reference_constants.py then checks that the oracle and the product carry exactly these numbers.

    python3 tests/golden/audit_constants.py            (needs /root/reference; the fixtures are committed)
"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))

NUM = re.compile(r"(?<![\w.])[-+]?(?:\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+|\.\d+(?:[eE][-+]?\d+)?)")

# group -> (file, first line, last line, what it is)
GROUPS = {
    "aces_srgb_to_ap1": ("base/spectrum/aces.zig", 9, 17, "sRGB -> AP1 matrix"),
    "aces_ap1_to_srgb": ("base/spectrum/aces.zig", 19, 27, "AP1 -> sRGB matrix"),
    "aces_ap1_to_rrt_sat": ("base/spectrum/aces.zig", 29, 36, "AP1 -> RRT_SAT matrix"),
    "aces_odt_sat_to_srgb": ("base/spectrum/aces.zig", 49, 57, "ODT_SAT -> sRGB matrix"),
    "aces_rrt_and_odt": ("base/spectrum/aces.zig", 59, 64, "RRT and ODT fit"),
    "agx_inset": ("core/rendering/sensor/agx.zig", 14, 21, "sRGB -> AgX inset matrix"),
    "agx_range": ("core/rendering/sensor/agx.zig", 23, 37, "AgX log2 range"),
    "agx_look": ("core/rendering/sensor/agx.zig", 39, 70, "AgX looks (luma weights, Substitute / Golden / Punchy)"),
    "agx_outset": ("core/rendering/sensor/agx.zig", 72, 79, "AgX -> sRGB outset matrix"),
    "agx_contrast": ("core/rendering/sensor/agx.zig", 81, 101, "AgX EOTF exponent and sigmoid polynomial"),
    "pbr_neutral": ("core/rendering/sensor/tonemapper.zig", 52, 71, "PbrNeutral tonemapper"),
    "blackman": ("core/rendering/sensor/sensor.zig", 27, 41, "Blackman filter"),
    "mitchell": ("core/rendering/sensor/sensor.zig", 43, 58, "Mitchell filter"),
    "ray_offset": ("core/scene/ray_offset.zig", 5, 12, "RayMaxT, offset scales"),
    "light_tree": ("core/scene/light/light_tree.zig", 248, 249, "MaxSplitDepth, MaxLights"),
    "primitive_tree": ("core/scene/light/light_tree.zig", 523, 523, "PrimitiveTree.MaxSplitDepth"),
    "low_threshold": ("core/rendering/integrator/helper.zig", 27, 39, "LightSampling.LowThreshold"),
    "russian_roulette": ("core/rendering/integrator/helper.zig", 75, 89, "russianRoulette"),
    "ggx_min_roughness": ("core/scene/material/ggx.zig", 14, 15, "MinRoughness"),
    "metal_presets": ("core/scene/material/metal_presets.zig", 6, 40, "ior and absorption of the metal presets"),
    "dot_min": ("base/math/safe.zig", 5, 5, "safe.DotMin"),
    "rainbow": ("core/scene/material/rainbow_integral.zig", 3, 41, "wavelength range, band count and the Rainbow spectrum table (w = 1 per entry)"),
    "glass_dispersion": ("core/scene/material/glass/glass_sample.zig", 190, 191, "Cauchy / Abbe dispersion constants"),
    "sample_normal": ("core/scene/material/material_helper.zig", 19, 64, "normal-map reconstruction and shading-normal adaption thresholds"),
    "flakes": ("core/scene/material/substitute/substitute_material.zig", 90, 95, "flake density constants (setFlakesSize)"),
    "coating_defaults": ("core/scene/material/substitute/substitute_material.zig", 37, 60, "Substitute parameter defaults"),
}


def literals(path, first, last):
    lines = open(os.path.join(REF, path)).read().split("\n")[first - 1:last]
    out = []
    for line in lines:
        code = line.split("//")[0]
        for m in NUM.finditer(code):
            out.append(float(m.group(0)))
    # plain integers that matter (MaxSplitDepth = 10 ...)
    if not out:
        for line in lines:
            out += [float(m) for m in re.findall(r"=\s*(\d+)\s*;", line.split("//")[0])]
    return out


def table(text, name):
    m = re.search(r"pub const " + name + r" = \[[^\]]*\]f32\{(.*?)\};", text, re.S)
    body = "\n".join(line.split("//")[0] for line in m.group(1).split("\n"))
    return [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]


def main():
    out = {}
    for name, (path, a, b, what) in GROUPS.items():
        vals = literals(path, a, b)
        assert vals, name
        out[name] = {"source": f"src/{path}:{a}-{b}", "what": what, "values": vals}
    json.dump(out, open(os.path.join(HERE, "reference_constants.json"), "w"), indent=1)
    text = open(os.path.join(REF, "core/scene/material/ggx_integral.zig")).read()
    luts = {n: table(text, n) for n in ("E_m", "E_m_avg", "E", "E_avg", "E_s")}
    assert [len(luts[n]) for n in ("E_m", "E_m_avg", "E", "E_avg", "E_s")] == [1024, 32, 4096, 256, 4096]
    json.dump({"source": "src/core/scene/material/ggx_integral.zig:3,166,175,979,1048 (all entries)", "tables": luts},
              open(os.path.join(HERE, "ggx_lut_full.json"), "w"))
    print({k: len(v["values"]) for k, v in out.items()}, {k: len(v) for k, v in luts.items()})


if __name__ == "__main__":
    main()
