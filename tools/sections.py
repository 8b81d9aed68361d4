#!/usr/bin/env python3
"""Section-cycle attribution of wf_shade (marks: SectionTimer in wavefront.hpp; add more where needed).  Needs a library built with -DZYG_PROFILE_SECTIONS
(ZYG_AMD_LIB=variants/lib_prof.so python3 tools/sections.py); prints each section's share of the summed wave cycles."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zyg_amd
from zyg_amd import scenes

w, h, iters = 1920, 1080, 16
material = sys.argv[1] if len(sys.argv) > 1 else "lambert"
eng = zyg_amd.Engine()
kw = {}
if os.environ.get("ZYG_SCENE") == "mesh":  # BASELINE.json configs[2]
    kw = dict(boxes=False, mesh=scenes.displaced_sphere_mesh(), mesh_position=scenes.MESH_CENTRE, mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
elif os.environ.get("ZYG_SCENE") == "mound":
    kw = dict(boxes=False, mesh=scenes.heightfield_mesh(), mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
if os.environ.get("ZYG_SCENE") == "proc":  # bench.py --material procedural: the complete closures outside the POOL kernels
    import bench
    bench.build_scene(eng, scenes, "cornell", "procedural", w, h, iters)
elif os.environ.get("ZYG_SCENE") == "outdoor":  # BASELINE.json configs[4]'s scene (POOL kernels)
    scenes.outdoor(eng, w, h, iters, integrator="PTMIS", material="substitute", showcase=True, mesh=scenes.heightfield_mesh(), max_depth=16,
                   sky_entity={"parameters": {"sun": {"rotation": [-55, 20, 0]}}}, box_lift=1e-3)
else:
    if os.environ.get("ZYG_SCENE") == "glass":
        kw = dict(glass_objects={"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)}, box_lift=1e-3)
    scenes.cornell(eng, w, h, iters, integrator="PTMIS", material=material, **kw)
eng.start_frame(0)
lib, ctx = eng.lib, eng.context()
lib.zygpu_get_section_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
lib.zygpu_reset_stats(ctx)
assert 0 == lib.zygpu_set_option(ctx, b"pool_relaxed", 1)
assert 0 == lib.zygpu_set_option(ctx, b"precise", int(os.environ.get("ZYG_PRECISE", "0")))  # the relaxed units unless asked otherwise
lib.zygpu_render(ctx, 0, 0, iters, 0, h)
lib.zygpu_synchronize(ctx)
out = (ctypes.c_uint64 * 32)()
lib.zygpu_get_section_cycles(ctx, out)
names = {0: "shade: state loads + fragment", 1: "shade: hit emission (evaluateRadiance)", 2: "shade: unoccluding-light candidates (evaluateRadiance)",
         3: "shade: IValue add, RR, record write", 4: "shade: finalize", 5: "shade: queue push",
         8: "trace: queue read (+ camera ray generation in the first tick)", 9: "trace: closest hit (TLAS + BLAS)",
         10: "trace: unoccluding-light walk", 11: "trace: queue pushes",
         16: "scatter: state loads + fragment + closure", 17: "scatter: light tree picks", 18: "scatter: light sample (SphQuad) + shadow ray set-up",
         19: "scatter: shadow ray traversal", 20: "scatter: radiance, bsdf evaluate, MIS", 21: "scatter: IValue add",
         22: "scatter: bxdf sample, state update, stores", 23: "scatter: queue push",
         }
for base, label in ((0, "wf_trace_shade + wf_shade"), (16, "wf_scatter")):
    tot = sum(out[base:base + 16]) or 1
    print(f"{label}: {tot / 1e9:.2f} G wave-cycles")
    if 16 == base:  # wf_scatter: sections 24-31 = cycles x lanes at the mark of sections 16-23
        tot = sum(out[16:24]) or 1
    for i in range(base, base + (8 if 16 == base else 16)):
        if out[i]:
            lanes = f"  lanes at its mark {out[i + 8] / out[i]:5.1f} / 64" if 16 == base else ""
            print(f"  {100.0 * out[i] / tot:5.1f} %  {names.get(i, i)}{lanes}")
eng.release()
