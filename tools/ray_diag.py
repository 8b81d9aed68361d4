#!/usr/bin/env python3
"""Which rays the relaxed units' walks answer differently from the oracle (tests/test_gpu_traversal.py's ray dump, precise = 0)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zyg_amd
from zyg_amd import scenes
import test_gpu_traversal as tt

oracle = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
n = 1 << 18
for scene in ("boxes", "sphere"):
    with zyg_amd.Engine() as e:
        if scene == "boxes":
            scenes.cornell(e, 64, 48, 1)
        else:
            scenes.cornell(e, 64, 48, 1, boxes=False, mesh=scenes.displaced_sphere_mesh(n=400), mesh_position=scenes.MESH_CENTRE,
                           mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
        e.start_frame(0)
        rays = tt._camera_relative(e, tt._rays(n, 17))
        for precise in (1, 0):
            ref_hits, ref_vis, out = tt._both(e, oracle, rays, precise=precise)
            for mode in (0, 1, 2, 3):
                hits, vis = out[mode]
                dv = vis != ref_vis
                dh = (hits[:, 3] != ref_hits[:, 3]) | (hits[:, 4] != ref_hits[:, 4])
                cls = np.zeros(n, dtype=int); cls[:64] = 1; cls[64:128] = 2
                print(f"{scene} precise={precise} mode {mode}: visibility differs {int(dv.sum())} (axis {int(dv[cls == 1].sum())}, grazing {int(dv[cls == 2].sum())}, "
                      f"finite-extent half {int(dv[n // 2:].sum())}; oracle says visible for {int(ref_vis[dv].sum())}), hit differs {int(dh.sum())}")
                if dv.any() and 0 == precise and mode == 0:
                    idx = np.nonzero(dv)[0][:8]
                    for i in idx:
                        print("   ray", i, rays[i], "oracle vis", ref_vis[i], "oracle hit prop", ref_hits[i, 4], "t", ref_hits[i, 0:1].view(np.float32))
