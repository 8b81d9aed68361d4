"""The wide re-layout of a mesh's BLAS (zyg_amd/csrc/hip/bvh4.hpp: collapse4, compress4) checked on the CPU: a small C++ harness
(tests/cpp/bvh4_check.cpp) runs both on the BVH2 the host builder made for a heightfield mesh and checks that every leaf survives
exactly once under boxes that contain it, and that the quantised child boxes contain the exact ones under the device's decode."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [24, 160, -128])
def test_collapse_and_compress(product_lib, tmp_path, n):
    exe = tmp_path / "bvh4_check"
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "bvh4_check.cpp"), "-o", str(exe)], check=True)
    with zyg_amd.Engine(product_lib) as e:
        # n < 0: the closed displaced sphere of configs[2] with -n x -n vertices
        mesh = scenes.heightfield_mesh(n=n) if n > 0 else scenes.displaced_sphere_mesh(n=-n)
        n = abs(n)
        scenes.cornell(e, 8, 8, 1, boxes=False, mesh=mesh, mesh_position=scenes.MESH_CENTRE)
        e.compile(0)
        sd = e.lib.su_ext_scene_desc().contents
        mesh = sd.meshes[sd.num_meshes - 1]
        assert mesh.num_triangles >= 2 * (n - 1) * (n - 1) * 0.9
        raw = ctypes.string_at(mesh.nodes, mesh.num_nodes * 32)
    path = tmp_path / "nodes.bin"
    path.write_bytes(raw)
    r = subprocess.run([str(exe), str(path)], capture_output=True, text=True)
    assert 0 == r.returncode, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("OK")
    # the walks' stack (dscene.hpp kStack4 = 48 entries) holds three parked children per level
    assert 3 * int(r.stdout.split()[-1]) <= 48, r.stdout
