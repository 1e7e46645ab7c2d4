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


def tile_rows(rank: int, world: int, H: int):
    """Strong-scaling split of ONE frame into horizontal bands (the "image tiles shard across the GPUs" of BASELINE.json's
    configs 4-5): contiguous, disjoint, complete, heights differ by at most 1. -> (first row, row count)."""
    if not (0 <= rank < world) or H < 0:
        raise ValueError("bad rank / world / H")
    base, extra = divmod(H, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


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


def render_tiles(scene, W, H, spp, bounces, accum, rank, world, **kw):
    """Tile sharding: this rank renders ALL `spp` samples of its band of rows straight into that band of the full-frame
    `accum` ([H,W,4], zero elsewhere); the same sum-reduce then assembles the frame on rank 0 (x + 0 == x: the result is
    bitwise the single-GPU frame). Returns the stats of the local render."""
    y0, h = tile_rows(rank, world, H)
    stats = {"rays": 0, "samples": 0, "passes": 0, "kernel_ms": 0.0}
    if h > 0:
        _, stats = scene.render(W, H, spp, bounces, accum=accum[y0:y0 + h], tile=(0, y0, W, h), **kw)
    reduce_accum(accum, 0)
    return stats
