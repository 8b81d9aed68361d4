#!/usr/bin/env python3
"""Relaxed vs precise arithmetic on the same samples: bias, RMS as a function of spp, NaNs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zyg_amd
from zyg_amd import scenes
w, h = 640, 360
res = {}
for spp in (4, 16, 64, 256):
    films = {}
    for precise in (1, 0):
        with zyg_amd.Engine() as e:
            scenes.cornell(e, w, h, spp)
            e.start_frame(0)
            e.set_option("precise", precise)
            e.render_iterations(spp)
            e.synchronize()
            films[precise] = (e.download_film(w, h), e.resolve(w, h))
    a, b = films[1][1][..., :3].astype(np.float64), films[0][1][..., :3].astype(np.float64)
    nan = int(np.isnan(films[0][0]).sum()), int(np.isnan(films[1][0]).sum())
    ac, bc = np.clip(np.nan_to_num(a), 0, 1), np.clip(np.nan_to_num(b), 0, 1)
    d = bc - ac
    print(f"spp {spp:4d}: rms {np.sqrt(np.mean(d * d)):.3e}  mean diff {d.mean():+.3e}  mean |a| {ac.mean():.4f}  nan(relaxed, precise) {nan}"
          f"  max|d| {np.abs(d).max():.3f}  frac |d|>0.01: {(np.abs(d) > 0.01).mean():.4f}")
    if spp == 256:
        # where: per 8x8 grid cell mean signed difference
        g = d.mean(axis=2).reshape(8, h // 8, 8, w // 8).mean(axis=(1, 3))
        np.set_printoptions(precision=4, suppress=True, linewidth=200)
        print(g)
