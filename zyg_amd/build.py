"""Builds libzyg_amd.so in-tree: host C++ (scene compile + su_* ABI) and the gfx950 HIP kernels.

    python -m zyg_amd.build

-ffp-contract=off everywhere: the reference (Zig) fuses a*b+c only at its explicit @mulAdd sites, which the
sources reproduce with fmaf / __fmaf_rn; correctly rounded f32 divide/sqrt is hipcc's default.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "libzyg_amd.so")

SOURCES = [
    os.path.join(HERE, "csrc", "hip", "zygpu.hip"),
    os.path.join(HERE, "csrc", "host", "engine.cpp"),
    os.path.join(HERE, "csrc", "host", "capi.cpp"),
]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(verbose=True, extra=()):
    # -disable-promote-alloca-to-vector: the per-lane traversal stacks stay in (sparsely touched) scratch instead of
    # being turned into ~110 VGPRs with waterfall indexing, which is what keeps wf_scatter / wf_nee at 2 waves/SIMD
    cmd = [hipcc(), "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
           "-mllvm", "-disable-promote-alloca-to-vector",
           "-Wall", "-Wno-unused-function", "-x", "hip", *SOURCES, "-o", OUT, *extra]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = list(SOURCES)
    for d in (os.path.join(HERE, "csrc", "hip"), os.path.join(HERE, "csrc", "host"), os.path.join(ROOT, "include")):
        deps += [os.path.join(d, f) for f in os.listdir(d)]
    return all(os.path.getmtime(p) <= t for p in deps)


if __name__ == "__main__":
    build(extra=sys.argv[1:])
