"""The eight-wide compressed BLAS of the relaxed units' mesh walkers (zyg_amd/csrc/hip/bvh8.hpp: build8) checked on the CPU: a C++
harness (tests/cpp/bvh8_check.cpp) builds it from the BVH2 the host builder made for a mesh and checks that every triangle is stored
once under quantised boxes that contain it, that the group encoding is consistent (consecutive inner children, running triangle
offsets), and that the host restatement of the device's group walk (walk8: octant-ordered hit bits, popcount child indexing, stack
of groups) reaches the brute-force closest hit of random rays, axis-parallel ones included."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [6, 40, -96])
def test_build_and_walk(product_lib, tmp_path, n):
    exe = tmp_path / "bvh8_check"
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "bvh8_check.cpp"), "-o", str(exe)], check=True)
    with zyg_amd.Engine(product_lib) as e:
        mesh = scenes.heightfield_mesh(n=n) if n > 0 else scenes.displaced_sphere_mesh(n=-n)
        scenes.cornell(e, 8, 8, 1, boxes=False, mesh=mesh, mesh_position=scenes.MESH_CENTRE)
        e.compile(0)
        sd = e.lib.su_ext_scene_desc().contents
        m = sd.meshes[sd.num_meshes - 1]
        nodes = ctypes.string_at(m.nodes, m.num_nodes * 32)
        tris = np.ctypeslib.as_array(m.triangles, shape=(m.num_triangles, 3)).copy()
        pos = np.ctypeslib.as_array(m.positions, shape=(int(tris.max()) + 1, 3)).copy()
    verts = np.zeros((len(tris), 3, 4), dtype=np.float32)
    verts[:, :, :3] = pos[tris]
    (tmp_path / "nodes.bin").write_bytes(nodes)
    (tmp_path / "verts.bin").write_bytes(verts.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "nodes.bin"), str(tmp_path / "verts.bin"), "400"], capture_output=True, text=True)
    assert 0 == r.returncode, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("OK"), r.stdout
    print(r.stdout)
    # the walkers' LDS stack (dscene.hpp kStack8) holds two groups per level
    assert 2 * int(r.stdout.split()[-1]) <= 32, r.stdout
