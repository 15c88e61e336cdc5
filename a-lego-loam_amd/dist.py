"""Multi-GPU sharding of the hot path: one process per GPU, independent streams per rank.

SURVEY.md §8e: a stream is a sequential chain (LO/LM state), streams are independent, so the path shards
across streams with NO data-path collective ("weak" scaling).  torch.distributed (RCCL on GPUs, gloo in the
CPU tests) is used only for the timing barrier and the max-over-ranks of the step time.
"""
import os


def env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_ids(rank: int, streams_per_gpu: int):
    """Global stream ids owned by `rank` (contiguous block; stream s starts 70*s scans along the T0 lap)."""
    return list(range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu))


def init(backend: str, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def shard_registration(handle, dist, rank: int, world: int):
    """BASELINE config 5: every rank runs the same stream; the scan-to-map registration is split over the ranks and its
    normal equations are summed with RCCL (alego_dist_init).  The unique id travels through torch.distributed."""
    from . import binding
    box = [binding.dist_unique_id() if rank == 0 else None]
    if dist is not None and world > 1:
        dist.broadcast_object_list(box, src=0)
    handle.dist_init(rank, world, box[0])


def max_over_ranks(seconds: float, dist, device="cpu") -> float:
    """The job's step time is the slowest rank's."""
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_scans_per_s(world: int, streams_per_gpu: int, steps: int, seconds: float) -> float:
    """Whole-job throughput: every rank advanced `streams_per_gpu` streams by `steps` scans in `seconds`."""
    return world * streams_per_gpu * steps / seconds


def gather_floats(value: float, dist, device="cpu"):
    """[value of rank 0, value of rank 1, ...] on every rank (all_gather of one double)."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
