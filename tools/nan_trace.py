#!/usr/bin/env python3
"""Development: with a library built with -DZYG_NAN_TRACE, which sites saw non-finite values (Control::error bits)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zyg_amd
from zyg_amd import scenes
w, h, spp = 640, 360, 128
SCENES = {
    "cornell": lambda e: scenes.cornell(e, w, h, spp),
    "substitute": lambda e: scenes.cornell(e, w, h, spp, material="substitute"),
    "mesh": lambda e: scenes.cornell(e, w, h, spp, boxes=False, mesh=scenes.heightfield_mesh(n=256), mesh_material=scenes.substitute((0.5, 0.5, 0.5), roughness=0.3, metallic=1.0)),
    "glass": lambda e: scenes.cornell(e, w, h, spp, glass_objects={"box": scenes.glass(1.5, (0.6, 0.9, 0.7), 0.3), "sphere": scenes.glass(1.5, roughness=0.2)}),
    "outdoor": lambda e: scenes.outdoor(e, w, h, spp, integrator="PTMIS", material="substitute", sky="canopy", sky_map=scenes.sky_image(256, 256), sky_spectrum=(1.0, 1.0, 1.0), showcase=True, mesh=scenes.heightfield_mesh(n=128), max_depth=16),
    "lights": lambda e: scenes.cornell(e, w, h, spp, mesh_lights=True, light_grid=3, spheres=True, sphere_lights=False),
    "disks": lambda e: scenes.cornell(e, w, h, spp, disk_lights=True, disk_and_cube=True, material="substitute"),
    "spherel": lambda e: scenes.cornell(e, w, h, spp, sphere_lights=True, spheres=True),
}
for name, build in SCENES.items():
  for precise in (0,):
    with zyg_amd.Engine() as e:
        build(e)
        print(name, end=": ")
        e.start_frame(0)
        e.set_option("precise", precise)
        e.render_iterations(spp)
        try:
            e.synchronize()
            print("precise", precise, "no flags")
        except RuntimeError as ex:
            print("precise", precise, ex)
        film = np.zeros((h, w, 4), np.float32)
        e.lib.zygpu_download_film(e.context(), film.ctypes.data)
        print("nan values", int(np.isnan(film).sum()))
