"""Builds libzyg_amd.so in-tree: host C++ (scene compile + su_* ABI) and the gfx950 HIP kernels.

    python -m zyg_amd.build [--mode both|precise|relaxed] [--units 0,4,6] [--jobs N] [extra hipcc flags]

The device code is compiled as translation units in parallel (csrc/hip/kernel_table.hpp): zygpu.hip (film kernels + the
zygpu_* ABI) and kernels.hip once per unit and arithmetic mode.  Objects are kept in zyg_amd/build/ and reused while they
carry the tag of the flags and of the sources they were compiled from (a sidecar <object>.tag, written after a successful
compile: contents are hashed, not mtimes); `--units` recompiles only the named kernel units (development: the others are
linked as they are, and the library is marked partial, so up_to_date() is false and bench.py / the tests rebuild it in full).

Precise units: -ffp-contract=off (the reference, Zig, fuses a*b+c only at its explicit @mulAdd sites, which the sources
reproduce with fmaf / __fmaf_rn); correctly rounded f32 divide/sqrt is hipcc's default.  Relaxed units (-DZYG_FAST=1): native
divide / sqrt, contraction allowed, f32 denormals flushed, OCML f32 transcendentals (dmath.hpp); no -ffast-math: the slab tests
rely on inf and NaN.  The relaxed units 0-6 are scheduled with -mllvm -amdgpu-sched-strategy=max-ilp (_objects; ZYG_SCHED_DEFAULT=1 in the
environment of a --variant build compiles the named units with the default scheduler instead: the A/B's other side).
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "libzyg_amd.so")
OBJ = os.path.join(HERE, "build")
HIP = os.path.join(HERE, "csrc", "hip")
HOST = os.path.join(HERE, "csrc", "host")
NUM_UNITS = 8  # kernel_table.hpp ZYG_NUM_UNITS

SOURCES = [os.path.join(HIP, "zygpu.hip"), os.path.join(HIP, "kernels.hip"), os.path.join(HIP, "group.hip"), os.path.join(HOST, "engine.cpp"),
           os.path.join(HOST, "capi.cpp")]

# -disable-promote-alloca-to-vector: the per-lane traversal stacks stay in (sparsely touched) scratch instead of
# being turned into ~110 VGPRs with waterfall indexing, which is what keeps wf_scatter / wf_nee at 2 waves/SIMD
COMMON = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-mllvm", "-disable-promote-alloca-to-vector", "-Wall",
          "-Wno-unused-function"]
PRECISE = ["-ffp-contract=off"]
# -fgpu-flush-denormals-to-zero: with f32 denormals on, LLVM wraps every native divide and sqrt in frexp / ldexp scaling (a
# normalize = 24 instructions instead of 5): wf_scatter's Lambert instantiation 6467 -> 5790 instructions, configs[1] +1 %
RELAXED = ["-DZYG_FAST=1", "-Dzd=zdf", "-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps():
    deps = []
    for d in (HIP, HOST, os.path.join(ROOT, "include")):
        deps += [os.path.join(d, f) for f in os.listdir(d)]
    deps.append(os.path.abspath(__file__))
    return deps


def source_hash(source=None):
    """Hash of the contents of every source and header the library is compiled from (csrc/hip, csrc/host, include/).
    source: only what that translation unit can include: itself and the headers of its own directory and of include/ (the hip
    and the host side never include each other, and no .hip / .cpp includes another), so that a change in zygpu.hip or on the
    host side does not recompile the kernel units."""
    h = hashlib.sha256()
    for p in sorted(_deps()):
        if p.endswith("build.py") or not os.path.isfile(p):
            continue
        if source is not None:
            header = p.endswith((".hpp", ".h"))
            if p != source and not (header and os.path.dirname(p) in (os.path.dirname(source), os.path.join(ROOT, "include"))):
                continue
        h.update(os.path.relpath(p, ROOT).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def device_hash():
    """Hash of what the device code is compiled from (csrc/hip and include/) and with (the kernel units' flags): what a committed
    rocprofv3 profile of the kernels is valid for (bench.py committed_profile); a change on the host side (loaders, scene compile)
    leaves it alone."""
    h = hashlib.sha256()
    for name, _, flags in _objects("both"):
        if name.startswith("unit"):
            h.update(" ".join([name] + COMMON + list(flags)).encode())
    for d in (HIP, os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if os.path.isfile(p):
                h.update(os.path.relpath(p, ROOT).encode())
                h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def _flags_hash(flags, extra=()):
    return hashlib.sha1(" ".join(COMMON + list(flags) + list(extra)).encode()).hexdigest()[:12]


def _library_tag(extra=(), mode="both"):
    """What a complete build of the current sources with these flags is tagged with (libzyg_amd.so.tag)."""
    return f"{_flags_hash(PRECISE + RELAXED, extra)}:{mode}:{source_hash()}"


def _read(path):
    try:
        return open(path).read()
    except OSError:
        return None


def _objects(mode):
    """(object path, source, flags) of every translation unit"""
    objs = [("zygpu.o", os.path.join(HIP, "zygpu.hip"), PRECISE), ("group.o", os.path.join(HIP, "group.hip"), PRECISE), ("engine.o", os.path.join(HOST, "engine.cpp"), PRECISE),
            ("capi.o", os.path.join(HOST, "capi.cpp"), PRECISE)]
    for u in range(NUM_UNITS):
        if mode in ("both", "precise"):
            # the units of the kernels that know every shape / light / closure call the restated libm instead of expanding it at
            # several hundred sites (dmath.hpp ZYG_LIBM_CALLS: wf_scatter<.., MESHL, POOL> 594 -> 285 KB of code, cornell_glass +13 %,
            # the outdoor scene +9 % in precise arithmetic); units 4 and 6 hold the small kernels of the plain scenes
            libm = ["-DZYG_LIBM_CALLS=1"] if u not in (4, 6) else []
            objs.append((f"unit{u}.o", os.path.join(HIP, "kernels.hip"), PRECISE + libm + [f"-DZYG_TU={u}"]))
        if mode in ("both", "relaxed"):
            # the scheduler that favours instruction-level parallelism over occupancy (the kernels' waves per SIMD are pinned anyway,
            # wavefront.hpp ZYG_WAVES): configs[1] +2.0 %, outdoor_mound at 3840x2160 +2.1 %, cornell_glass +1.2 %, configs[2] +0.5 %
            # (tools/ab_lib_args.sh, two pairs each); unit 7 (the mesh walkers, wf_scatter_h) gains nothing from it and keeps the default
            # (ZYG_SCHED_DEFAULT=1 in the environment of a --variant build: the A/B's other side)
            sched = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if 7 != u and not os.environ.get("ZYG_SCHED_DEFAULT") else []
            objs.append((f"unit{u}f.o", os.path.join(HIP, "kernels.hip"), RELAXED + sched + [f"-DZYG_TU={u}"]))
    return objs


def _compile(job):
    name, src, flags, extra, verbose = job
    out = os.path.join(OBJ, name)
    cmd = [hipcc(), *COMMON, *flags, *extra, "-x", "hip", "-c", src, "-o", out]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        sys.stderr.write(f"[{time.time() - t0:6.1f} s] {name}\n{r.stderr}")
    if r.returncode:
        if os.path.exists(out + ".tag"):
            os.remove(out + ".tag")
        raise RuntimeError(f"hipcc failed for {name}: {' '.join(cmd)}")
    open(out + ".tag", "w").write(_flags_hash(flags, extra) + ":" + _compile.sources[src])
    return out


_compile.sources = {}


def build_variant(name, units, extra, mode="both", verbose=True):
    """Development: variants/lib_<name>.so = the main build's objects with the kernel units `units` (and zygpu.o when
    `units` holds -1) recompiled with `extra` flags into build/<name>/ (mode "relaxed" / "precise": only the units of that
    arithmetic mode; every other object is the main build's).  Load it with ZYG_AMD_LIB=variants/lib_<name>.so."""
    vdir = os.path.join(OBJ, name)
    os.makedirs(vdir, exist_ok=True)
    _compile.sources = {src: source_hash(src) for src in SOURCES}
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    todo, objs = [], []
    for oname, src, flags in _objects("both"):
        mine = (oname.startswith("unit") and int(oname[4:].rstrip("f.o")) in units) or (oname == "zygpu.o" and -1 in units)
        if oname.startswith("unit") and mode != "both" and ("relaxed" == mode) != oname.endswith("f.o"):
            mine = False
        if mine:
            todo.append((os.path.join(name, oname), src, flags, list(extra), verbose))
            objs.append(os.path.join(vdir, oname))
        else:
            objs.append(os.path.join(OBJ, oname))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(_compile, todo))
    out = os.path.join(ROOT, "variants", f"lib_{name}.so")
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-lrccl", "-o", out], check=True)
    return out


def build(verbose=True, extra=(), mode="both", units=None, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    extra = list(extra)
    sources = _compile.sources = {src: source_hash(src) for src in SOURCES}
    todo, objs, stale_left = [], [], []
    for name, src, flags in _objects(mode):
        out = os.path.join(OBJ, name)
        objs.append(out)
        # an object is reused only when it was compiled from these sources with these flags
        fresh = os.path.exists(out) and _read(out + ".tag") == _flags_hash(flags, extra) + ":" + sources[src]
        if units is not None and name.startswith("unit"):
            unit = int(name[4:].rstrip("f.o"))
            if unit in units or not os.path.exists(out):
                todo.append((name, src, flags, extra, verbose))
            elif not fresh:
                stale_left.append(name)
        elif not fresh:
            todo.append((name, src, flags, extra, verbose))
    # the longest compiles first (measured seconds per unit, precise / relaxed; kernel_table.hpp says what each holds)
    weight = {"unit5.o": 40, "unit5f.o": 34, "unit7f.o": 32, "unit4.o": 29, "unit7.o": 26, "unit3f.o": 26, "unit3.o": 22, "unit1.o": 22, "unit1f.o": 24,
              "unit6.o": 27, "unit6f.o": 30, "unit2.o": 21, "unit2f.o": 23, "unit0.o": 28, "unit0f.o": 27}
    todo.sort(key=lambda j: (-weight.get(j[0], 18 if j[0].startswith("unit") else 5), j[0]))
    t0 = time.time()
    if os.path.exists(OUT + ".tag"):
        os.remove(OUT + ".tag")
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(_compile, todo))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-lrccl", "-o", OUT]  # libz: the PNG reader (host/image_readers.hpp)
    if verbose:
        print(f"compiled {len(todo)} of {len(objs)} translation units in {time.time() - t0:.0f} s; linking", flush=True)
    subprocess.run(cmd, check=True)
    if stale_left:
        # development only: the library mixes objects of different source revisions; never taken for a complete build
        print(f"PARTIAL build: {', '.join(stale_left)} were not recompiled for the current sources", file=sys.stderr, flush=True)
        open(OUT + ".tag", "w").write("partial")
    else:
        open(OUT + ".tag", "w").write(_library_tag(extra, mode))
    return OUT


def up_to_date():
    """The library is a complete build (both arithmetic modes, the default flags) of the sources as they are now."""
    return os.path.exists(OUT) and _read(OUT + ".tag") == _library_tag()


def ensure_built(verbose=False):
    """Builds the library unless it is up to date, under a file lock: ranks of one launch may all call this, the first one
    builds, the others wait for it and find the library fresh."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, "build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not up_to_date():
                build(verbose=verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == "__main__":
    args = sys.argv[1:]
    kw = {}
    variant = None
    while args and args[0] in ("--mode", "--units", "--jobs", "--variant"):
        key, val = args[0][2:], args[1]
        if "variant" == key:
            variant = val
        else:
            kw[key] = val if "mode" == key else int(val) if "jobs" == key else {int(v) for v in val.split(",")}
        args = args[2:]
    if variant:
        print(build_variant(variant, kw.get("units", set(range(NUM_UNITS))), args, kw.get("mode", "both")))
    else:
        build(extra=args, **kw)
