"""Multi-GPU sharding of the path: every (pixel, iteration) is independent and the film stores additive
sums {sum w*rgb, sum w} (reference src/core/rendering/sensor/buffer_opaque.zig:39-45), so ranks render disjoint
iteration ranges (or row ranges) of the same frame with a replicated scene and the films are summed once with a
single reduce (RCCL over xGMI on the GPU box, gloo in the tests).  No collective on the data path.

`ShardedFrame` is the one implementation of that protocol: bench.py drives the device with it over RCCL, the tests drive
the device (two processes on one GPU, gloo) and the CPU oracle (gloo) with it.  It knows nothing about who renders: the
caller passes `render(first_iteration, count, y_begin, y_end)`.
"""
import os
import time


def iteration_shard(rank, world, iterations_per_rank):
    """Rank r renders iterations [r * k, (r + 1) * k): weak scaling, total spp = world * k."""
    return rank * iterations_per_rank, iterations_per_rank


def row_shard(rank, world, height):
    """Rank r renders film rows [y0, y1): strong scaling of one frame (exact for the radius-0 sensor)."""
    base, rem = divmod(height, world)
    y0 = rank * base + min(rank, rem)
    return y0, y0 + base + (1 if rank < rem else 0)


def reduce_film(film, dist, dst=0):
    """Sums the per-rank films into rank `dst` (torch.distributed; backend "nccl" is RCCL on ROCm)."""
    dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def init_process_group(backend, rank=None, world=None, device=None):
    """torch.distributed rendezvous on 127.0.0.1 (the container's hostname may not resolve).  `device` (a torch.device)
    binds the RCCL communicator to this rank's GPU up front.  Returns the torch.distributed module."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # dmabuf IPC is the only mode the host driver supports.  HSA reads the flag once, when the process makes its first HIP call,
    # so a caller that has touched the GPU already must have set it itself (bench.py does, before it imports torch); here it
    # only helps a process that has not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    kw = {"device_id": device} if (device is not None and "nccl" == backend) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


class ShardedFrame:
    """One frame rendered by `world` ranks.

    mode "iterations": the run's `steps` x `iterations_per_step` iterations per rank are rank r's block
    [r * steps * ips, (r + 1) * steps * ips) of the frame's world * steps * ips samples per pixel, over all rows.
    mode "rows": every rank renders all steps * ips iterations of its own rows.

    A frame with a process group runs its collectives whatever the world size (a world of one still goes through the backend:
    that is how the RCCL path is exercised on a box with one GPU); without one (dist = None, world = 1) there is nothing to run.
    """

    def __init__(self, rank=0, world=1, dist=None, mode="iterations", height=0):
        if mode not in ("iterations", "rows"):
            raise ValueError(mode)
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs an initialised torch.distributed")
        self.rank, self.world, self.dist, self.mode, self.height = rank, world, dist, mode, height

    def step_range(self, step, iterations_per_step, steps, height=None):
        """(first_iteration, count, y_begin, y_end) of this rank's part of step `step`; `height`: of the frame rendered, when it is
        not the one the object was made for (bench.py measures scenes of other sizes with the same ranks)."""
        height = self.height if height is None else height
        if "iterations" == self.mode:
            first, _ = iteration_shard(self.rank, self.world, steps * iterations_per_step)
            return first + step * iterations_per_step, iterations_per_step, 0, height
        y0, y1 = row_shard(self.rank, self.world, height)
        return step * iterations_per_step, iterations_per_step, y0, y1

    def render_step(self, render, step, iterations_per_step, steps, height=None):
        render(*self.step_range(step, iterations_per_step, steps, height))

    def reduce(self, film, dst=0):
        """The one exchange step: the ranks' films summed into rank `dst` (in place on `dst`)."""
        if self.dist is not None:
            reduce_film(film, self.dist, dst)
        return film

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def timed(self, body, synchronize):
        """bench.py's timing protocol: barrier + device synchronize on both sides, maximum over the ranks."""
        self.barrier()
        synchronize()
        t0 = time.perf_counter()
        body()
        synchronize()
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64)
            if "nccl" == self.dist.get_backend():
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed
