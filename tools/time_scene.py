"""Times render_iterations on a scenes.py scene at 1080p: python3 tools/time_scene.py glass|rough|showcase|cornell [iterations] [batch_mpaths]"""
import sys, time
sys.path.insert(0, '.')
import zyg_amd
from zyg_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "glass"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w, h = 1920, 1080
with zyg_amd.Engine() as e:
    if which == "glass":
        scenes.cornell(e, w, h, 256, glass_objects={"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)})
    elif which == "rough":
        scenes.cornell(e, w, h, 256, glass_objects={"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3, roughness=0.3), "sphere": scenes.glass(1.5, roughness=0.1)})
    elif which == "showcase":
        scenes.outdoor(e, w, h, 256, sky="canopy", sky_map=scenes.sky_image(256, 256), showcase=True, material="substitute", max_depth=16)
    else:
        scenes.cornell(e, w, h, 256)
    e.start_frame(0)
    if len(sys.argv) > 3:
        assert e.lib.zygpu_set_option(e.context(), b"batch_mpaths", int(sys.argv[3])) == 0
    e.render_iterations(iters)  # same batch size as the timed call: (re)allocating the path pool takes seconds
    e.synchronize()
    import ctypes
    ms = (ctypes.c_double * 6)()
    ln = (ctypes.c_uint64 * 6)()
    e.lib.zygpu_kernel_time.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    e.lib.zygpu_kernel_time(e.context(), ms, ln, 1)
    t0 = time.perf_counter()
    e.render_iterations(iters)
    e.synchronize()
    dt = time.perf_counter() - t0
    e.lib.zygpu_kernel_time(e.context(), ms, ln, 0)
    print(which, f"{w * h * iters / dt / 1e6:.1f} Msamples/s", f"{dt * 1e3 / iters:.2f} ms/iteration",
          {n: (round(ms[i], 1), int(ln[i])) for i, n in enumerate(("intersect", "shade", "nee", "shadow", "extend", "scatter")) if ln[i]})
