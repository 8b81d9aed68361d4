"""A second, independent implementation of the scene-compile products, because the device and the oracle
both read the scene the product's host compiled: a wrong SAH bin or light-tree cone would be the same on both sides.

* BVH: a naive median-split BVH2 built here in numpy (no SAH, no spatial splits, its own triangle order) replaces the host
  builder's tree of a mesh in a copy of the compiled scene; the oracle renders both.  An acceleration structure changes
  which triangles are tested, never which one is hit: the two films must agree up to the rare exact tie.
* Light tree: one flat leaf holding every light (no builder, no cones, no variance: the picks come from the per-light
  importance alone) replaces the builder's tree; the oracle renders both with PTMIS.  Both are unbiased estimators of
  the same image: the means must agree within noise, and both must agree with the tree-free PT render.
"""
import ctypes

import numpy as np
import pytest

import zyg_amd
from zyg_amd import scenes

from util import oracle_render


def _median_split_bvh(tri_positions, leaf_size=4):
    """tri_positions: (n, 3, 3).  Returns (nodes as (m, 8) float32 rows in the reference's 32 B layout, triangle order)."""
    n = tri_positions.shape[0]
    lo, hi = tri_positions.min(axis=1), tri_positions.max(axis=1)
    centroid = 0.5 * (lo + hi)
    order = np.arange(n)
    nodes = []

    def new_node():
        nodes.append(np.zeros(8, dtype=np.float32))
        return len(nodes) - 1

    def fill(index, ids, children_or_first, count):
        row = nodes[index]
        row[0:3] = lo[ids].min(axis=0)
        row[4:7] = hi[ids].max(axis=0)
        row.view(np.uint32)[3] = children_or_first
        row.view(np.uint32)[7] = count

    out_order = []
    root = new_node()
    stack = [(root, order)]
    while stack:
        index, ids = stack.pop()
        if len(ids) <= leaf_size:
            fill(index, ids, len(out_order), len(ids))
            out_order.extend(ids.tolist())
            continue
        extent = centroid[ids].max(axis=0) - centroid[ids].min(axis=0)
        axis = int(np.argmax(extent))
        ids = ids[np.argsort(centroid[ids, axis], kind="stable")]
        half = len(ids) // 2
        left, right = new_node(), new_node()  # children adjacent (node.zig:9-20)
        assert right == left + 1
        fill(index, ids, left, 0)
        stack.append((right, ids[half:]))
        stack.append((left, ids[:half]))
    return np.stack(nodes), np.asarray(out_order)


def _with_rebuilt_mesh(sd, mesh_index):
    """a copy of the compiled scene whose mesh `mesh_index` has the naive BVH; returns (scene copy, keep-alive list)"""
    m = sd.meshes[mesh_index]
    nt, nv = m.num_triangles, m.num_vertices
    tris = np.ctypeslib.as_array(m.triangles, (nt * 3,)).reshape(nt, 3).copy()
    parts = np.ctypeslib.as_array(m.triangle_parts, (nt,)).copy()
    pos = np.ctypeslib.as_array(m.positions, (nv * 3 + 1,))[: nv * 3].reshape(nv, 3)
    # the host builder may have duplicated triangles (spatial splits): the naive tree takes every distinct one once
    _, first = np.unique(np.sort(tris, axis=1), axis=0, return_index=True)
    first.sort()
    tris, parts = tris[first], parts[first]
    nodes, order = _median_split_bvh(pos[tris])
    new_tris = np.ascontiguousarray(tris[order].astype(np.uint32))
    new_parts = np.ascontiguousarray(parts[order].astype(np.uint16))
    nodes = np.ascontiguousarray(nodes)
    meshes = (zyg_amd.Mesh * sd.num_meshes)()
    for i in range(sd.num_meshes):
        ctypes.memmove(ctypes.addressof(meshes[i]), ctypes.addressof(sd.meshes[i]), ctypes.sizeof(zyg_amd.Mesh))
    nm = meshes[mesh_index]
    nm.nodes = ctypes.cast(nodes.ctypes.data, ctypes.POINTER(zyg_amd.BvhNode))
    nm.triangles = new_tris.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    nm.triangle_parts = new_parts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))
    nm.num_nodes = nodes.shape[0]
    nm.num_triangles = new_tris.shape[0]
    copy = zyg_amd.SceneDesc.from_buffer_copy(sd)
    copy.meshes = ctypes.cast(meshes, ctypes.POINTER(zyg_amd.Mesh))
    return copy, [meshes, nodes, new_tris, new_parts]


def test_naive_bvh_gives_the_same_image_as_the_sah_builder(oracle, engine):
    w, h, spp = 96, 64, 2
    scenes.cornell(engine, w, h, spp, integrator="PTMIS", boxes=False, mesh=scenes.heightfield_mesh(n=40),
                   mesh_material=scenes.substitute((0.5, 0.5, 0.5)))
    sd, vd = engine.compile()
    mesh_index = max(range(sd.num_meshes), key=lambda i: sd.meshes[i].num_triangles)
    assert sd.meshes[mesh_index].num_triangles >= 2 * 39 * 39
    a = oracle_render(oracle, sd, vd, spp, threads=8)
    sd2, keep = _with_rebuilt_mesh(sd, mesh_index)
    assert sd2.meshes[mesh_index].num_nodes != sd.meshes[mesh_index].num_nodes  # really another tree
    b = oracle_render(oracle, sd2, vd, spp, threads=8)
    assert np.array_equal(a[..., 3], b[..., 3])
    same = np.all(a == b, axis=-1).mean()
    assert same >= 0.999, same  # a ray through a shared edge may take the other triangle
    np.testing.assert_allclose(a[..., :3].mean(axis=(0, 1)), b[..., :3].mean(axis=(0, 1)), rtol=2e-3)


def test_naive_bvh_on_the_cornell_boxes(oracle, engine):
    w, h, spp = 64, 48, 2
    scenes.cornell(engine, w, h, spp, integrator="PTMIS")
    sd, vd = engine.compile()
    a = oracle_render(oracle, sd, vd, spp, threads=8)
    cur, keep = sd, []
    for i in range(sd.num_meshes):
        cur, k = _with_rebuilt_mesh(cur, i)
        keep.append(k)
    b = oracle_render(oracle, cur, vd, spp, threads=8)
    assert np.all(a == b, axis=-1).mean() >= 0.999


def _with_flat_light_list(sd):
    lt = sd.light_tree
    n = lt.num_lights
    assert lt.num_infinite_lights == 0
    node = (zyg_amd.LightNode * 1)()
    node[0].meta = 0  # no children, first light 0
    node[0].num_lights = n
    node[0].power = 1.0
    ident = (ctypes.c_uint32 * n)(*range(n))
    middles = (ctypes.c_uint32 * 1)(0)
    copy = zyg_amd.SceneDesc.from_buffer_copy(sd)
    copy.light_tree.nodes = ctypes.cast(node, ctypes.POINTER(zyg_amd.LightNode))
    copy.light_tree.node_middles = ctypes.cast(middles, ctypes.POINTER(ctypes.c_uint32))
    copy.light_tree.light_orders = ctypes.cast(ident, ctypes.POINTER(ctypes.c_uint32))
    copy.light_tree.light_mapping = ctypes.cast(ident, ctypes.POINTER(ctypes.c_uint32))
    copy.light_tree.num_nodes = 1
    return copy, [node, ident, middles]


@pytest.mark.parametrize("grid", [2, 4])
def test_flat_light_list_agrees_with_the_light_tree(oracle, engine, grid):
    """grid x grid small quad lights under the ceiling: the builder's tree (cones, variance, splitting) against one flat leaf
    and against the path tracer without light sampling: three estimators of one image"""
    w, h, spp = 48, 32, 48
    scenes.cornell(engine, w, h, spp, integrator="PTMIS", many_lights=grid, split_threshold=0.5)
    sd, vd = engine.compile()
    assert sd.light_tree.num_lights == grid * grid and sd.light_tree.num_nodes > 1
    tree = oracle_render(oracle, sd, vd, spp, threads=8)
    sd2, keep = _with_flat_light_list(sd)
    flat = oracle_render(oracle, sd2, vd, spp, threads=8)
    mean_tree = tree[..., :3].sum(axis=(0, 1)) / tree[..., 3].sum()
    mean_flat = flat[..., :3].sum(axis=(0, 1)) / flat[..., 3].sum()
    assert not np.array_equal(tree, flat)
    np.testing.assert_allclose(mean_tree, mean_flat, rtol=0.02)
    # block means (8 x 8 pixels): no region is systematically brighter under one of the two
    bt = (tree[..., :3].reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3)) / spp).reshape(-1, 3)
    bf = (flat[..., :3].reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3)) / spp).reshape(-1, 3)
    rel = np.abs(bt - bf) / np.maximum(bf, 0.05)
    assert np.median(rel) < 0.05 and rel.max() < 0.35, (float(np.median(rel)), float(rel.max()))
