"""Device groups (include/zygpu.h "Device groups", zyg_amd/csrc/hip/group.hip): several GPUs behind the same su_* calls.
Stands where the reference's Driver hands tiles to its worker threads (driver.zig:115-152, 293-348).

Without a GPU: the split of a job into the ranks' shares, and that the group fails loudly (no device: no group).
On the GPU box (one MI355X): a group of two contexts on device 0 (members that share a device are driven one after the other and summed
by the group's film_add kernel), a group of one through RCCL (ncclCommInitRank + ncclReduce from C++), and a world of one joined through
a communicator id, all through su_* and compared with the CPU oracle / the single-context film."""
import ctypes

import numpy as np
import pytest

import zyg_amd
from zyg_amd import distributed, scenes

from util import oracle_render

W, H, SPP = 160, 90, 8


def test_shares_partition_the_job(product_lib):
    """zygpu_group_share: contiguous shares in rank order that cover [0, count) once, sizes differing by at most one; the same split as
    zyg_amd.distributed.row_shard (what the torch path of rounds 1-4 sharded rows with)."""
    first, length = ctypes.c_uint32(), ctypes.c_uint32()
    for count in (0, 1, 7, 8, 9, 1080, 2160, 4096):
        for world in (1, 2, 3, 4, 8):
            at = 0
            sizes = []
            for rank in range(world):
                assert 0 == product_lib.zygpu_group_share(count, world, rank, ctypes.byref(first), ctypes.byref(length))
                assert first.value == at
                assert (first.value, first.value + length.value) == distributed.row_shard(rank, world, count)
                at += length.value
                sizes.append(length.value)
            assert at == count and max(sizes) - min(sizes) <= 1
    assert -1 == product_lib.zygpu_group_share(8, 2, 2, ctypes.byref(first), ctypes.byref(length))
    assert -1 == product_lib.zygpu_group_share(8, 0, 0, ctypes.byref(first), ctypes.byref(length))


def test_group_fails_loudly_without_a_device(product_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    group = ctypes.c_void_p()
    ids = (ctypes.c_int * 2)(0, 1)
    assert -1 == product_lib.zygpu_group_create(ids, 2, None, 0, 0, 0, ctypes.byref(group))
    assert not group.value and b"hipGetDeviceCount" in product_lib.zygpu_last_error()
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, 32, 18, 1)
        e.set_devices([0, 0])
        assert -1 == e.lib.su_start_frame(0)  # no CPU fallback behind the group either


def _single(engine_factory, build, spp, one_shot=False):
    with engine_factory() as e:
        build(e)
        if one_shot:
            e.render_frame(0)
        else:
            e.start_frame(0)
            e.render_iterations(spp)
        e.synchronize()
        return e.download_film(W, H)


@pytest.mark.gpu
@pytest.mark.parametrize("shard", [0, 1, 2])
def test_two_contexts_behind_su_render_the_oracles_film(product_lib, oracle, shard):
    """su_ext_set_devices([0, 0]): su_start_frame uploads to both contexts, su_render_iterations splits the call (its iterations, the
    frame's rows, or the library's choice), the download / resolve sums the films onto the first context's.  Rows: the oracle's film bit
    for bit (precise arithmetic is a new context's default); iterations: the same samples summed in another order."""
    def build(e):
        scenes.cornell(e, W, H, SPP, integrator="PTMIS")

    with zyg_amd.Engine(product_lib) as e:
        build(e)
        e.set_devices([0, 0], shard=shard)
        e.start_frame(0)
        group = e.group()
        assert group and 2 == e.lib.zygpu_group_world(group) and 2 == e.lib.zygpu_group_size(group) and 0 == e.lib.zygpu_group_uses_rccl(group)
        e.render_iterations(SPP // 2)      # shard 2: 4 iterations for 2 devices -> iterations
        e.render_iterations(1)             # shard 2: fewer iterations than devices -> rows
        e.render_iterations(SPP - SPP // 2 - 1)
        e.synchronize()
        assert 1 == e.lib.zygpu_group_dirty(group)
        film = e.download_film(W, H)       # (reduces)
        assert 0 == e.lib.zygpu_group_dirty(group)
        again = e.download_film(W, H)      # a second reduce adds nothing: the other films were cleared
        resolved = e.resolve(W, H)
        sd, vd = e.lib.su_ext_scene_desc().contents, e.lib.su_ext_view_desc().contents
        want = oracle_render(oracle, sd, vd, SPP)
    assert np.array_equal(film, again)
    assert np.array_equal(film[..., 3], want[..., 3]) and np.all(film[..., 3] == SPP)
    if 1 == shard:
        assert np.array_equal(film, want)
    else:
        np.testing.assert_allclose(film[..., :3], want[..., :3], rtol=2e-6, atol=1e-7)
    assert np.isfinite(resolved).all() and resolved[..., :3].max() > 0


@pytest.mark.gpu
def test_group_one_shot_frame_and_glass(product_lib, oracle):
    """su_render_frame through a group of two contexts: one-shot rendering splits rows (a pixel's PCG stream runs on across its samples),
    so the film is the oracle's one-shot film bit for bit; a scene with Glass (the POOL kernels: host-synchronous ticks) likewise."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, W, H, 4, integrator="PTMIS", glass_objects={"sphere": scenes.glass(1.5)})
        e.set_devices([0, 0])
        e.render_frame(0)
        e.synchronize()
        film = e.download_film(W, H)
        sd, vd = e.lib.su_ext_scene_desc().contents, e.lib.su_ext_view_desc().contents
        want = oracle_render(oracle, sd, vd, 1, samples_per_call=4)
    assert np.array_equal(film, want)


@pytest.mark.gpu
def test_group_renders_procedural_textures(product_lib, oracle):
    """Every context of a group has its own pool, the values of its vertices' procedural textures (Params::tex_values) included:
    two contexts, the frame's rows split between them, a Checker floor, a Mix on the back wall and a noise normal map render the
    oracle's film bit for bit."""
    S = scenes.substitute_full
    ov = {"floor": S({"Checker": {"scale": 6.0, "colors": [[0.9, 0.9, 0.9], [0.1, 0.1, 0.1]]}}, roughness={"Noise": {"scale": [5, 5, 5]}}),
          "back": S({"Mix": {"a": [0.8, 0.2, 0.2], "b": [0.2, 0.2, 0.8], "weight": {"Noise": {"scale": [3, 3, 3]}}}}, normal={"Noise": {"scale": [9, 9, 9]}})}
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, W, H, SPP, integrator="PTMIS", material="substitute", overrides=ov)
        e.set_devices([0, 0], shard=1)
        e.start_frame(0)
        e.render_iterations(SPP)
        e.synchronize()
        film = e.download_film(W, H)
        sd, vd = e.lib.su_ext_scene_desc().contents, e.lib.su_ext_view_desc().contents
        assert sd.num_procedurals >= 5
        want = oracle_render(oracle, sd, vd, SPP)
    assert np.array_equal(film, want)


@pytest.mark.gpu
def test_group_of_one_reduces_through_rccl(product_lib, oracle):
    """ZYGPU_GROUP_ALWAYS_RCCL: a world of one still builds its communicator (ncclGetUniqueId + ncclCommInitRank on the member's thread)
    and the reduce is ncclReduce on the member's stream: the RCCL path of group.hip on the hardware there is.  Film == oracle."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, W, H, SPP, integrator="PTMIS")
        e.set_devices([0], flags=1)
        e.start_frame(0)
        group = e.group()
        assert 1 == e.lib.zygpu_group_uses_rccl(group) and 1 == e.lib.zygpu_group_world(group)
        e.render_iterations(SPP)
        assert 0 == e.lib.zygpu_group_reduce(group)
        e.synchronize()
        film = e.download_film(W, H)
        sd, vd = e.lib.su_ext_scene_desc().contents, e.lib.su_ext_view_desc().contents
        want = oracle_render(oracle, sd, vd, SPP)
    assert np.array_equal(film, want)


@pytest.mark.gpu
def test_world_joined_through_a_communicator_id(product_lib, oracle):
    """The multi-process form with the one process this box allows: rank 0 makes the id (zygpu_group_unique_id), su_ext_set_world hands
    it to the group, whose member joins the communicator with ncclCommInitRank(id, rank 0 of 1); su_render_iterations names the world's
    iterations."""
    with zyg_amd.Engine(product_lib) as e:
        scenes.cornell(e, W, H, SPP, integrator="PTMIS")
        ident = e.unique_id()
        assert 128 == len(ident) and any(ident)
        e.set_devices([0], shard=0)
        e.set_world(ident, 0, 1)
        e.start_frame(0)
        group = e.group()
        assert 1 == e.lib.zygpu_group_uses_rccl(group) and 0 == e.lib.zygpu_group_first_rank(group)
        e.render_iterations(SPP)
        e.synchronize()
        film = e.download_film(W, H)
        sd, vd = e.lib.su_ext_scene_desc().contents, e.lib.su_ext_view_desc().contents
        want = oracle_render(oracle, sd, vd, SPP)
    assert np.array_equal(film, want)


@pytest.mark.gpu
def test_group_reduces_the_aov_planes(product_lib):
    """A view that records AOVs splits rows whatever the sharding asked for (Depth keeps a pixel's minimum and MaterialId its heaviest
    sample over the pixel's OWN samples: with its rows on one device a pixel's planes are complete there), and the reduce sums the planes
    (Depth: a minimum): every plane of a two-context group is the single context's, bit for bit."""
    names = ["Albedo", "Depth", "MaterialId", "ShadingNormal", "Direct"]

    def planes(devices):
        with zyg_amd.Engine(product_lib) as e:
            scenes.cornell(e, W, H, 4, integrator="PTMIS")
            e.aovs(names)
            if devices:
                e.set_devices(devices, shard=0)  # (iterations asked for: rows are what runs)
            e.start_frame(0)
            e.render_iterations(3)
            e.render_iterations(1)
            e.synchronize()
            film = e.download_film(W, H)  # (reduces)
            return film, {k: e.download_aov(k, W, H) for k in range(9)}

    film1, one = planes(None)
    film2, two = planes([0, 0])
    assert np.array_equal(film1, film2)
    recorded = [k for k in range(9) if one[k] is not None]
    assert len(recorded) == len(names)
    for k in recorded:
        assert two[k] is not None and np.array_equal(one[k], two[k]), k
