#!/usr/bin/env python3
"""What the GPU box's host really gives this process, and how the oracle scales with its thread count."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, zyg_amd
from zyg_amd import scenes
info = bench.host_cpu()
print(json.dumps(info))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "-", e)
lib = zyg_amd.load_library()
orc = bench.Oracle(zyg_amd)
w, h = 1920, 1080
with zyg_amd.Engine(lib) as e:
    scenes.cornell(e, w, h, 256)
    e.compile(0)
    film = np.zeros((h, w, 4), np.float32)
    orc.render(e, film, 0, 1, 0, h, info["usable"])
    for t in (8, 16, 32, 64, 96, 128, 192, 256):
        if t > 2 * info["logical"]: break
        dt = orc.render(e, film, 0, 2, 0, h, t)
        print(f"threads {t:4d}: {w * h * 2 / dt / 1e6:8.3f} Msamples/s  ({w * h * 2 / dt / 1e6 / t:.4f} per thread)", flush=True)
