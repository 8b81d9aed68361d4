#!/usr/bin/env python3
"""wf_persistent against the wavefront kernels: same film bit for bit (precise), timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zyg_amd
from zyg_amd import scenes
for name, build in (("cornell", lambda e, w, h, s: scenes.cornell(e, w, h, s)), ("substitute", lambda e, w, h, s: scenes.cornell(e, w, h, s, material="substitute")),
                    ("mesh", lambda e, w, h, s: scenes.cornell(e, w, h, s, boxes=False, mesh=scenes.heightfield_mesh(n=200), mesh_material=scenes.substitute((0.5, 0.5, 0.5))))):
    w, h, spp = 480, 270, 8
    films = {}
    for precise in (1, 0):
        for persistent in (0, 1):
            with zyg_amd.Engine() as e:
                build(e, w, h, spp)
                e.start_frame(0)
                e.set_option("precise", precise)
                e.set_option("persistent", persistent)
                e.render_iterations(spp)
                e.synchronize()
                films[(precise, persistent)] = e.download_film(w, h)
    for precise in (1, 0):
        a, b = films[(precise, 0)], films[(precise, 1)]
        print(name, "precise" if precise else "relaxed", "identical:", bool(np.array_equal(a, b)), "max diff", float(np.abs(a - b).max()), "finite", bool(np.isfinite(b).all()))
