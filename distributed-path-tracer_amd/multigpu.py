"""Multi-GPU host logic for the one exchange step the path has: the framebuffer sum.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU in the tests).
The scene is replicated on every GPU (it is small next to 288 GB of HBM); the SAMPLE space is sharded: rank r
renders samples [r*spp, (r+1)*spp) of every pixel. Because random numbers are keyed by (pixel, global sample index),
the union over ranks is exactly the sample set a single GPU would have traced for N*spp samples, and the only
communication is ONE sum-reduce of the float32 accumulation buffer (W*H*4 floats: 33.2 MB at 1080p) to rank 0.
This replaces the reference's planned (never implemented) SNS/SQS fan-in — path-tracer-core/src/models/work_info.hpp:22-23,
intersection_worker.cpp:78-110 — see SURVEY.md §8e.
"""


def sample_range(rank: int, world: int, spp_per_rank: int):
    """First sample index and count for `rank` (weak scaling: every rank traces spp_per_rank samples per pixel)."""
    if not (0 <= rank < world) or spp_per_rank < 0:
        raise ValueError("bad rank / world / spp")
    return rank * spp_per_rank, spp_per_rank


def split_samples(rank: int, world: int, spp_total: int):
    """Strong-scaling split of a fixed sample budget: contiguous, disjoint, complete, sizes differ by at most 1."""
    if not (0 <= rank < world) or spp_total < 0:
        raise ValueError("bad rank / world / spp")
    base, extra = divmod(spp_total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def reduce_accum(accum, dst: int = 0):
    """Sum-reduce the accumulation buffer (a torch tensor, on the GPU for nccl / on the CPU for gloo) onto rank dst."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def render_sharded(scene, W, H, spp_per_rank, bounces, accum, rank, world, **kw):
    """Render this rank's sample range into `accum` (sums) and reduce onto rank 0. Returns the stats of the local render.
    `scene` is a distributed-path-tracer_amd.Scene (anything with the same .render signature)."""
    s0, n = sample_range(rank, world, spp_per_rank)
    _, stats = scene.render(W, H, n, bounces, accum=accum, sample0=s0, **kw)
    reduce_accum(accum, 0)
    return stats
