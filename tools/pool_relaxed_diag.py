#!/usr/bin/env python3
"""Relaxed vs precise arithmetic on the POOL scenes (Glass): RMS of the resolved images on the same samples, with the boxes standing
on the floor (their bottom faces coincide with it: which surface a ray meets first hangs on the last bit) and lifted off it.
usage: pool_relaxed_diag.py glass|outdoor [w h] [spp ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zyg_amd
from zyg_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "glass"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 360)
spps = [int(a) for a in sys.argv[4:]] or [64, 256]


def build(e, spp, lift):
    if "glass" == name:
        scenes.cornell(e, w, h, spp, glass_objects={"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5)}, box_lift=lift)
    else:
        scenes.outdoor(e, w, h, spp, integrator="PTMIS", material="substitute", showcase=True, mesh=scenes.heightfield_mesh(n=256), max_depth=16,
                       sky_entity={"parameters": {"sun": {"rotation": [-55, 20, 0]}}}, box_lift=lift)


for lift in (0.0, 1e-3):
    for spp in spps:
        films = {}
        for precise in (1, 0):
            with zyg_amd.Engine() as e:
                build(e, spp, lift)
                e.start_frame(0)
                e.set_option("pool_relaxed", 1)
                e.set_option("precise", precise)
                e.render_iterations(spp)
                e.synchronize()
                films[precise] = (e.download_film(w, h), e.resolve(w, h))
        a, b = films[1][1][..., :3].astype(np.float64), films[0][1][..., :3].astype(np.float64)
        ac, bc = np.clip(np.nan_to_num(a), 0, 1), np.clip(np.nan_to_num(b), 0, 1)
        d = bc - ac
        same = np.all(films[0][0] == films[1][0], axis=-1).mean()
        rows = np.sqrt(np.mean(d * d, axis=(1, 2)))
        print(f"{name} lift {lift:g} spp {spp:4d}: rms {np.sqrt(np.mean(d * d)):.3e}  mean diff {d.mean():+.3e}  frac |d|>0.01: {(np.abs(d) > 0.01).mean():.4f}"
              f"  identical pixels {same:.3f}  worst rows {np.argsort(rows)[-3:].tolist()} {np.sort(rows)[-3:].round(4).tolist()}", flush=True)
