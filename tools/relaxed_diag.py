#!/usr/bin/env python3
"""Relaxed vs precise arithmetic on the same samples (the precise film is the oracle's, bit for bit): bias and RMS as a
function of spp, NaNs.  usage: relaxed_diag.py [cornell|glass|outdoor|mesh] [w h]
If the difference is the noise of paths that took another branch on an ulp (chaotic through glass), the RMS falls like
1 / sqrt(spp) and the mean difference stays at the noise of the mean; a bias would stay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zyg_amd
from zyg_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 360)


def build(e, spp):
    if "glass" == name:
        scenes.cornell(e, w, h, spp, glass_objects={"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)})
    elif "outdoor" == name:
        scenes.outdoor(e, w, h, spp, integrator="PTMIS", material="substitute", sky="canopy", sky_map=scenes.sky_image(1024, 1024),
                       sky_spectrum=(1.0, 1.0, 1.0), showcase=True, mesh=scenes.heightfield_mesh(n=256), max_depth=16)
    elif "mesh" == name:
        scenes.cornell(e, w, h, spp, boxes=False, mesh=scenes.displaced_sphere_mesh(n=256), mesh_position=scenes.MESH_CENTRE,
                       mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
    else:
        scenes.cornell(e, w, h, spp)


for spp in (16, 64, 256, 1024):
    films = {}
    for precise in (1, 0):
        with zyg_amd.Engine() as e:
            build(e, spp)
            e.start_frame(0)
            e.set_option("precise", precise)
            e.render_iterations(spp)
            e.synchronize()
            films[precise] = (e.download_film(w, h), e.resolve(w, h))
    a, b = films[1][1][..., :3].astype(np.float64), films[0][1][..., :3].astype(np.float64)
    nan = int(np.isnan(films[0][0]).sum()), int(np.isnan(films[1][0]).sum())
    ac, bc = np.clip(np.nan_to_num(a), 0, 1), np.clip(np.nan_to_num(b), 0, 1)
    d = bc - ac
    same = np.all(films[0][0] == films[1][0], axis=-1).mean()
    print(f"{name} spp {spp:4d}: rms {np.sqrt(np.mean(d * d)):.3e}  mean diff {d.mean():+.3e}  mean {ac.mean():.4f}  nan(relaxed, precise) {nan}"
          f"  max|d| {np.abs(d).max():.3f}  frac |d|>0.01: {(np.abs(d) > 0.01).mean():.4f}  identical pixels {same:.3f}", flush=True)
