"""The N > 1 path on CPU: world_size-2 `gloo` processes shard the iterations (and, separately, the rows) of one frame,
render their shard with the CPU oracle standing in for the device, and reduce the films; rank 0 compares the sum with
the single-process film.  This is the same sharding + reduce bench.py runs with RCCL."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import zyg_amd
    from zyg_amd import distributed, scenes
    from util import oracle_render

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import conftest
    oracle = ctypes.CDLL(conftest._build_oracle())
    oracle.zyg_oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(zyg_amd.Stats)]
    w, h, k = 40, 30, 2
    with zyg_amd.Engine() as e:
        scenes.cornell(e, w, h, world * k)
        sd, vd = e.compile()
        # the same ShardedFrame bench.py drives the device with; here the oracle renders the shard
        frame = distributed.ShardedFrame(rank, world, dist, mode, h)
        film = np.zeros((h, w, 4), dtype=np.float32)

        def render(first, count, y0, y1):
            film[...] += oracle_render(oracle, sd, vd, count, threads=2, iteration0=first, y_range=(y0, y1))

        steps, ips = (1, k) if mode == "iterations" else (1, world * k)  # rows: every rank renders all world * k iterations
        for s in range(steps):
            frame.render_step(render, s, ips, steps)
        t = torch.from_numpy(film)
        frame.reduce(t)
        if rank == 0:
            full = oracle_render(oracle, sd, vd, world * k, threads=2)
            np.savez(out_path, reduced=t.numpy(), full=full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["iterations", "rows"])
def test_two_rank_sharding_matches_single_process(tmp_path, mode):
    import torch.multiprocessing as mp
    conftest = __import__("conftest")
    conftest._build_oracle()
    conftest._build_product()
    out = str(tmp_path / "films.npz")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    d = np.load(out)
    if mode == "rows":
        assert np.array_equal(d["reduced"], d["full"])  # disjoint pixels: exact
    else:
        assert np.array_equal(d["reduced"][..., 3], d["full"][..., 3])
        np.testing.assert_allclose(d["reduced"], d["full"], rtol=1e-6, atol=1e-7)  # float association only


def test_shard_helpers():
    from zyg_amd import distributed
    assert [distributed.row_shard(r, 3, 10) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [distributed.iteration_shard(r, 4, 16) for r in range(4)] == [(0, 16), (16, 16), (32, 16), (48, 16)]


def _gpu_worker(rank, world, port, mode, out_path):
    """one rank of a two-process run on the one GPU: its shard through the HIP path (zygpu_render), films summed over gloo"""
    sys.path.insert(0, ROOT)
    import torch

    import zyg_amd
    from zyg_amd import distributed, scenes

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist = distributed.init_process_group("gloo", rank, world)
    w, h, steps, ips = 160, 90, 2, 2
    total = world * steps * ips if mode == "iterations" else steps * ips
    with zyg_amd.Engine() as e:
        scenes.cornell(e, w, h, total)
        e.start_frame(0)
        e.set_option("precise", 1)
        frame = distributed.ShardedFrame(rank, world, dist, mode, h)

        def render(first, count, y0, y1):
            assert e.lib.zygpu_render(e.context(), 0, first, count, y0, y1) == 0, e.lib.zygpu_last_error()

        elapsed = frame.timed(lambda: [frame.render_step(render, s, ips, steps) for s in range(steps)] and e.synchronize(), e.synchronize)
        assert elapsed > 0.0
        film = torch.from_numpy(e.download_film(w, h))
        frame.reduce(film)
        if 0 == rank:
            e.start_frame(0)
            e.set_option("precise", 1)
            e.render_iterations(total)
            e.synchronize()
            np.savez(out_path, reduced=film.numpy(), full=e.download_film(w, h))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["iterations", "rows"])
def test_two_processes_on_one_gpu_shard_the_frame(tmp_path, mode):
    """The N > 1 path with the device in it: two fresh processes (children, no re-exec of a process that touched the GPU),
    each with its own engine and HIP context on the one GPU, shard the frame with zyg_amd.distributed.ShardedFrame exactly as
    bench.py does and sum their films over gloo; rank 0 compares with its own single-process render."""
    import torch.multiprocessing as mp
    __import__("conftest")._build_product()
    out = str(tmp_path / "films.npz")
    mp.spawn(_gpu_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    d = np.load(out)
    assert np.array_equal(d["reduced"][..., 3], d["full"][..., 3])
    if mode == "rows":
        assert np.array_equal(d["reduced"], d["full"])  # disjoint pixels: exact
    else:
        np.testing.assert_allclose(d["reduced"][..., :3], d["full"][..., :3], rtol=1e-6, atol=1e-7)  # float association only


def _bench_two_ranks(launcher, extra=()):
    import json
    import subprocess
    env = dict(os.environ, ZYG_BENCH_BACKEND="gloo", ZYG_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180", "--iters-per-step", "4",
             *extra]
    if "torchrun" == launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + bench
    else:
        cmd = [sys.executable] + bench
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_py_two_ranks_end_to_end():
    """bench.py's own N > 1 path on the one GPU of this box: both ranks on device 0 and gloo instead of RCCL (ZYG_BENCH_*),
    everything else the measured code: ShardedFrame's shards, timing protocol and reduce, rank 0's one JSON line.  Launched both
    ways: the way the driver launches it (torch.distributed.run, one process per rank, 127.0.0.1) and plainly
    (`python bench.py --gpus 2`: it starts its own ranks as fresh child processes before touching the GPU); the same line but
    for the measured times."""
    outs = {"torchrun": _bench_two_ranks("torchrun"), "self": _bench_two_ranks("self", ("--spawn", "1"))}
    for out in outs.values():
        assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
        assert out["config"]["parallelism"] == "iterations x2 (weak)" and out["config"]["collective"] == "gloo (torch.distributed.reduce)"
        # two ranks x 2 steps x 4 iterations, summed into rank 0's film
        assert out["film_weight_min_max"] == [16.0, 16.0]
        assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    a, b = outs["torchrun"], outs["self"]
    assert a.keys() == b.keys() and a["config"] == b["config"] and a["rays_per_sample"] == b["rays_per_sample"]
    assert a["roofline"]["kernel_bytes"] == b["roofline"]["kernel_bytes"]


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["iterations", "rows"])
def test_bench_py_one_process_drives_a_group(shard):
    """`python bench.py --gpus 2` without a launcher: ONE process, the devices behind a ZygpuGroup (su_ext_set_devices: a context and a
    host thread per device, the films summed from C++).  On this box both members sit on device 0 (ZYG_BENCH_ONE_DEVICE), which RCCL
    refuses inside one communicator, so the sum is the group's film_add kernel; with two GPUs the same line says ncclReduce."""
    out = _bench_two_ranks("self", ("--shard", shard))
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["collective"].startswith("film_add kernel (ZygpuGroup")
    if "iterations" == shard:
        assert out["scaling"] == "weak" and out["film_weight_min_max"] == [16.0, 16.0]  # 2 devices x 2 steps x 4 iterations
        samples = 320 * 180 * 16
    else:
        assert out["scaling"] == "strong" and out["film_weight_min_max"] == [8.0, 8.0]  # every row rendered once, by its device
        samples = 320 * 180 * 8
    assert abs(out["value"] * 1e6 * out["ms_per_step"] * 1e-3 * out["steps"] / samples - 1.0) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["iterations", "rows"])
def test_bench_py_strong_scaling_modes(shard):
    """BASELINE.json configs[3] as written: a fixed number of samples per pixel split over the ranks (`--spp-total 16`), by iterations
    (every rank renders 8 iterations of all rows) or by rows (every rank renders all 16 iterations of half the rows): the line says
    "strong", names the sharding, the film holds the job's 16 samples per pixel everywhere, and `value` counts the job's samples
    once."""
    out = _bench_two_ranks("torchrun", ("--spp-total", "16", "--shard", shard))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["parallelism"].startswith(f"{shard} x2, 16 spp in all (strong)")
    assert out["config"]["spp_per_rank"] == (8 if "iterations" == shard else 16)
    assert out["film_weight_min_max"] == [16.0, 16.0]
    samples = 320 * 180 * 16
    assert abs(out["value"] * 1e6 * out["ms_per_step"] * 1e-3 * out["steps"] / samples - 1.0) < 0.02


def _rccl_worker(port, out_path):
    """a fresh process: a world of one over RCCL on cuda:0, the frame's collectives on the device film"""
    sys.path.insert(0, ROOT)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch

    import zyg_amd
    from zyg_amd import distributed, scenes
    device = torch.device("cuda", 0)
    dist = distributed.init_process_group("nccl", 0, 1, device)
    torch.cuda.set_device(device)
    assert "nccl" == dist.get_backend()
    w, h, steps, ips = 160, 90, 2, 2
    with zyg_amd.Engine() as e:
        scenes.cornell(e, w, h, steps * ips)
        e.start_frame(0)
        e.set_option("precise", 1)
        film = torch.zeros((h, w, 4), dtype=torch.float32, device=device)
        assert e.lib.zygpu_set_external_film(e.context(), ctypes.c_void_p(film.data_ptr())) == 0
        frame = distributed.ShardedFrame(0, 1, dist, "iterations", h)

        def render(first, count, y0, y1):
            assert e.lib.zygpu_render(e.context(), 0, first, count, y0, y1) == 0, e.lib.zygpu_last_error()

        def body():
            for s in range(steps):
                frame.render_step(render, s, ips, steps)
            e.synchronize()
            frame.reduce(film)  # dist.reduce over RCCL on the tensor the kernels accumulated into

        elapsed = frame.timed(body, lambda: torch.cuda.synchronize(device))  # barrier + all_reduce(MAX) over RCCL
        assert elapsed > 0.0
        reduced = film.cpu().numpy()
        e.lib.zygpu_set_external_film(e.context(), None)
        e.start_frame(0)
        e.set_option("precise", 1)
        e.render_iterations(steps * ips)
        e.synchronize()
        np.savez(out_path, reduced=reduced, full=e.download_film(w, h))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_world_of_one_reduces_the_external_film(tmp_path):
    """The RCCL code path on the hardware there is: init_process_group("nccl", world_size = 1, device_id = cuda:0) in a fresh
    child process, the engine accumulating into a torch tensor (zygpu_set_external_film), ShardedFrame's barrier, reduce and
    max-over-ranks running as RCCL collectives on it; the film is the single-process film bit for bit."""
    import torch.multiprocessing as mp
    __import__("conftest")._build_product()
    out = str(tmp_path / "films.npz")
    # spawn: the child must not inherit an initialised GPU
    p = mp.get_context("spawn").Process(target=_rccl_worker, args=(_free_port(), out))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    d = np.load(out)
    assert np.array_equal(d["reduced"], d["full"]) and np.all(d["full"][..., 3] == 4.0)


@pytest.mark.gpu
@pytest.mark.parametrize("reduce", ["group", "torch"])
def test_bench_py_runs_its_collectives_over_rccl_on_one_gpu(reduce):
    """bench.py itself as one rank of a launcher's world (ZYG_BENCH_GROUP=1: the N > 1 protocol with WORLD_SIZE 1, what one GPU allows):
    the process group before the device is set, barrier and max-over-ranks in the timed protocol, and the film reduce either as
    ncclReduce from C++ over a communicator whose id rank 0 made and torch.distributed carried (--reduce group, the default:
    zygpu_group_unique_id -> su_ext_set_world -> zygpu_group_reduce), or as torch.distributed.reduce on the film tensor."""
    import json
    import subprocess
    env = dict(os.environ, ZYG_BENCH_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180",
           "--iters-per-step", "4", "--no-cpu-baseline", "--no-extra", "--reduce", reduce]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][0])
    want = "ncclReduce from C++ (ZygpuGroup), one process per GPU" if "group" == reduce else "nccl (torch.distributed.reduce)"
    assert out["config"]["collective"] == want and out["film_weight_min_max"] == [8.0, 8.0] and out["value"] > 0
