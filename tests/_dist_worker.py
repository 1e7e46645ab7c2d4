"""Worker for tests/test_multigpu_gloo.py: one rank of a world_size-N gloo job (CPU).
The product's multi-GPU host logic (multigpu.render_sharded / reduce_accum) runs unchanged; the GPU scene is
replaced by an oracle-backed stand-in with the same .render signature (test infrastructure, CPU)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pt_oracle as ora  # noqa: E402

mg = importlib.import_module("distributed-path-tracer_amd.multigpu")
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "cornell.gltf")


class OracleScene:
    def __init__(self, many_surfaces=False):
        if many_surfaces:   # a Sponza-class stand-in: one model of 24 surfaces under the sun (BASELINE configs 4 / 5 shard by tiles)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from conftest import oracle_from_dict
            self.s = oracle_from_dict(ora, importlib.import_module("distributed-path-tracer_amd.procedural").atrium_scene(1))
        else:
            self.s = ora.OracleScene(ora.load_gltf(CORNELL))

    def render(self, W, H, spp, bounces, accum=None, sample0=0, tile=None, shard=None, **kw):
        smp = self.s.render_samples(ora.make_cfg(W, H, spp, bounces, sample0=sample0, tile=tile), threads=2)   # [h,w,spp,3]
        a = accum.numpy()
        # ptx_render_cfg.shard_*: only the pixels of this shard's interleaved tiles are touched
        m = mg.tile_mask(shard[0], shard[1], W, H, shard[2]) if shard else np.ones((H, W), bool)
        for k in range(spp):                       # sums in sample order, like k_resolve
            a[..., :3][m] += smp[:, :, k][m]
            a[..., 3][m] += 1.0
        return accum, {"rays": 0, "samples": int(m.sum()) * spp}


def main():
    out = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "samples"
    W, H, spp, b = 40, 24, 3, 4
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    accum = torch.zeros((H, W, 4), dtype=torch.float32)
    if mode == "atrium":
        W, H = 48, 32
        accum = torch.zeros((H, W, 4), dtype=torch.float32)
        st = mg.render_tiles(OracleScene(many_surfaces=True), W, H, 2, 5, accum, rank, world, tile=8)
        assert st["samples"] == int(mg.tile_mask(rank, world, W, H, 8).sum()) * 2
    elif mode == "tiles":
        mg.render_tiles(OracleScene(), W, H, 2 * spp, b, accum, rank, world, tile=8)   # 5 x 3 tiles of 8 x 8 on the 40 x 24 frame
    elif mode == "strong":
        mg.render_samples(OracleScene(), W, H, 2 * spp, b, accum, rank, world)
    else:
        mg.render_sharded(OracleScene(), W, H, spp, b, accum, rank, world)
    if rank == 0:
        np.save(out, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
