import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zyg_amd
from zyg_amd import scenes
for (w, h, spp) in ((128, 72, 1), (128, 72, 8), (480, 270, 1), (480, 270, 2), (480, 270, 8)):
    films = {}
    for persistent in (0, 1):
        with zyg_amd.Engine() as e:
            scenes.cornell(e, w, h, spp)
            e.start_frame(0)
            e.set_option("precise", 1)
            e.set_option("persistent", persistent)
            e.render_iterations(spp)
            e.synchronize()
            films[persistent] = e.download_film(w, h)
    a, b = films[0], films[1]
    diff = np.any(a != b, axis=-1)
    print(w, h, spp, "paths", w * h * spp, "differing pixels", int(diff.sum()), "mean a", float(a[..., :3].mean()), "mean b", float(b[..., :3].mean()))
    ys, xs = np.nonzero(diff)
    for y, x in list(zip(ys, xs))[:3]:
        print("   ", y, x, a[y, x], b[y, x])
