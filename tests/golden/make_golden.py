#!/usr/bin/env python3
"""Generates tests/golden/sampler_golden.json: known-answer vectors for the integer core of the hot path.

Inputs and expected outputs only (no reference source text):
  * sobol5(i) for a list of indices, evaluated here in Python with the direction words parsed out of the
    reference's src/core/sampler/sobol.zig:194-245 (run in the build container, where /root/reference exists);
  * hash(x) of sobol.zig:107-124, evaluated with the constants read from that file;
  * the public pcg-c-basic demo vector (initstate 42, initseq 54), which generator.zig:13-47 reproduces.
"""
import json
import os
import re

REF = "/root/reference/src/core/sampler/sobol.zig"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sampler_golden.json")

src = open(REF).read()
table = src[src.index("const Directions"):]
words = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{8}", table)]
assert len(words) == 160
D = [words[d * 32:(d + 1) * 32] for d in range(5)]

hash_src = src[src.index("fn hash(i: u32) u32"):src.index("fn hashCombine")]
active = [l for l in hash_src.splitlines() if not l.strip().startswith("//")]
m1, m2 = [int(x, 16) for x in re.findall(r"\*%= (0x[0-9a-fA-F]+)", "\n".join(active))]
s1, s2, s3 = [int(x) for x in re.findall(r">> (\d+)", "\n".join(active))]


def hash32(i):
    x = i ^ (i >> s1)
    x = (x * m1) & 0xFFFFFFFF
    x ^= x >> s2
    x = (x * m2) & 0xFFFFFFFF
    x ^= x >> s3
    return x


def sobol5(index):
    out = [0] * 5
    for bit in range(32):
        if (index >> bit) & 1:
            for d in range(5):
                out[d] ^= D[d][bit]
    return out


indices = [0, 1, 2, 3, 5, 7, 8, 15, 16, 31, 64, 255, 256, 1023, 4097, 65535, 65536, 1 << 20, (1 << 24) + 12345,
           0x7FFFFFFF, 0x80000000, 0xDEADBEEF, 0xFFFFFFFF]
golden = {
    "sobol5": [{"index": i, "out": sobol5(i)} for i in indices],
    "hash": [{"in": i, "out": hash32(i)} for i in [0, 1, 2, 42, 0xFFFFFFFF, 0x12345678, 1 << 31]],
    "pcg32_demo": {"state": 42, "sequence": 54,
                   "out": [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]},
}
json.dump(golden, open(OUT, "w"), indent=1)
print("wrote", OUT)

# ---- GGX albedo tables: a sample of the entries the reference carries in material/ggx_integral.zig (data the
# reference's own integrator printed, {d:.8}); zyg_amd/csrc/host/ggx_lut.hpp has to reproduce them.
LUT_REF = "/root/reference/src/core/scene/material/ggx_integral.zig"
LUT_OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ggx_lut_golden.json")
lut_src = open(LUT_REF).read()


def lut_table(name):
    body = re.search(r"pub const " + name + r" = \[[^\]]*\]f32\{(.*?)\};", lut_src, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    return [float(x) for x in body.replace("\n", " ").split(",") if x.strip()]


lut_golden = {}
for which, (name, stride) in enumerate([("E_m", 13), ("E_m_avg", 3), ("E", 41), ("E_avg", 7), ("E_s", 37)]):
    t = lut_table(name)
    idx = sorted(set(list(range(0, len(t), stride)) + [len(t) - 1]))
    lut_golden[name] = {"which": which, "length": len(t), "index": idx, "value": [t[i] for i in idx]}
json.dump(lut_golden, open(LUT_OUT, "w"), indent=1)
print("wrote", LUT_OUT)
