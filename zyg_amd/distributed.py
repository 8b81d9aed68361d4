"""Multi-GPU sharding of the path: every (pixel, iteration) is independent and the film stores additive
sums {sum w*rgb, sum w} (reference src/core/rendering/sensor/buffer_opaque.zig:39-45), so ranks render disjoint
iteration ranges (or row ranges) of the same frame with a replicated scene and the films are summed once with a
single reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests).  No collective on the data path."""


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
