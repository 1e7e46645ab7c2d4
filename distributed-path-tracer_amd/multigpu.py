"""Multi-GPU host logic for the one exchange step the path has: the framebuffer sum.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU in the tests).
The scene is replicated on every GPU (it is small next to 288 GB of HBM). ONE frame is sharded in one of two ways
(SURVEY.md section 8e), both ending in ONE sum-reduce of the float32 accumulation buffer (W*H*4 floats: 33.2 MB at 1080p,
132.7 MB at 4K) onto rank 0:

  * samples (`render_samples`, strong scaling, the default): rank r traces the r-th of `world` contiguous slices of the
    frame's sample indices, of every pixel — perfectly balanced whatever the image looks like. Random numbers are keyed by
    (pixel, GLOBAL sample index), so the union over ranks is exactly the sample set of a single-GPU render.
  * tiles (`render_tiles`, strong scaling; "image tiles shard across the GPUs" of BASELINE configs 4-5): the frame is cut into
    64 x 64 tiles, tile t (row-major) belongs to rank t % world — interleaved, so that expensive and cheap image regions are
    dealt out evenly; each rank renders all samples of its tiles in ONE launch sequence (ptx_render_cfg.shard_*: a pixel list
    inside the library) straight into a zeroed full-frame buffer; the sum (x + 0 = x) is bitwise the single-GPU frame.

`render_sharded` is the weak-scaling variant (every rank adds `spp` samples of its own: the frame gets world * spp).
This replaces the reference's planned (never implemented) SNS/SQS fan-in — path-tracer-core/src/models/work_info.hpp:22-23,
intersection_worker.cpp:78-110.
"""

TILE = 64   # interleaved 64 x 64 tiles


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


def tile_owner(x: int, y: int, W: int, world: int, tile: int = TILE):
    """Rank that renders pixel (x, y): the row-major index of its tile, modulo the number of ranks (ptx_render_cfg.shard_*)."""
    tiles_x = (W + tile - 1) // tile
    return ((y // tile) * tiles_x + (x // tile)) % world


def tile_mask(rank: int, world: int, W: int, H: int, tile: int = TILE):
    """Boolean [H, W] mask of the pixels `rank` renders under interleaved tile sharding."""
    import numpy as np
    if not (0 <= rank < world):
        raise ValueError("bad rank / world")
    tiles_x = (W + tile - 1) // tile
    ty, tx = np.meshgrid(np.arange(H) // tile, np.arange(W) // tile, indexing="ij")
    return (ty * tiles_x + tx) % world == rank


def _order_before_reduce(scene):
    """The library launches on its context's own non-blocking stream; the collective runs on torch's. Wait for the render
    before the buffer is handed to the collective (the host-side sync costs microseconds next to a frame)."""
    ctx = getattr(scene, "ctx", None)
    if ctx is not None:
        ctx.synchronize()


def reduce_accum(accum, dst: int = 0):
    """Sum-reduce the accumulation buffer (a torch tensor, on the GPU for nccl / on the CPU for gloo) onto rank dst."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def render_sharded(scene, W, H, spp_per_rank, bounces, accum, rank, world, **kw):
    """WEAK scaling: render this rank's own `spp_per_rank` samples (global indices [rank*spp, (rank+1)*spp)) into `accum` (sums)
    and reduce onto rank 0: the frame receives world * spp_per_rank samples. Returns the stats of the local render.
    `scene` is a distributed-path-tracer_amd.Scene (anything with the same .render signature). `accum` must be ready (zeroed)
    before the call as far as the library's stream is concerned: synchronise torch's stream after preparing it."""
    s0, n = sample_range(rank, world, spp_per_rank)
    _, stats = scene.render(W, H, n, bounces, accum=accum, sample0=s0, **kw)
    _order_before_reduce(scene)
    reduce_accum(accum, 0)
    return stats


def render_samples(scene, W, H, spp_total, bounces, accum, rank, world, **kw):
    """STRONG scaling by samples: ONE frame of `spp_total` samples per pixel; this rank traces its contiguous share of the
    sample indices of every pixel, then the buffers are sum-reduced onto rank 0."""
    s0, n = split_samples(rank, world, spp_total)
    stats = {"rays": 0, "samples": 0, "passes": 0, "kernel_ms": 0.0}
    if n > 0:
        _, stats = scene.render(W, H, n, bounces, accum=accum, sample0=s0, **kw)
    _order_before_reduce(scene)
    reduce_accum(accum, 0)
    return stats


def render_tiles(scene, W, H, spp, bounces, accum, rank, world, tile=TILE, **kw):
    """STRONG scaling by interleaved tiles: this rank renders ALL `spp` samples of the 64 x 64 image tiles t with
    t % world == rank into the full-frame `accum` ([H,W,4], zero elsewhere); the sum-reduce then assembles the frame on rank 0
    (x + 0 == x: bitwise the single-GPU frame). Returns the stats of the local render."""
    _, stats = scene.render(W, H, spp, bounces, accum=accum, shard=(rank, world, tile), **kw)
    _order_before_reduce(scene)
    reduce_accum(accum, 0)
    return stats if stats is not None else {"rays": 0, "samples": 0, "passes": 0, "kernel_ms": 0.0}
