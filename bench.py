#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json config 2).

A "step" = one pass of the integrator over one batch of synthetic input: the Cornell-box scene
(scenes/cornell-box, the reference's own asset) at 1920x1080, 256 spp, 8 bounces on each GPU,
entirely through the C ABI (ptx_render). Inputs (scene) are resident in HBM before the timed region.
With N > 1 ranks (one process per GPU, torch.distributed / RCCL) every rank renders its own 256-sample
range of the same frame (weak scaling: the frame gets N*256 spp) and the float32 accumulation buffers
are sum-reduced to rank 0 over xGMI inside the timed region — the path's only exchange step.

Prints ONE JSON line on rank 0 (see the bench contract): value = whole-job Msamples/s; plus
  roofline    — dominant kernel (k_render_pass): algorithmic bytes per launch / HIP-event kernel time
  cpu_baseline — the UNMODIFIED reference renderer (oracle/_ref/ref_harness) timed on this box's host cores
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "cornell.gltf")

W, H, SPP, BOUNCES = 1920, 1080, 256, 8
# algorithmic bytes per ray on the reference-topology Cornell trees
#   188 (ray/hit/path-state streams) + 8*3.44 (KD nodes) + 40*11.82 (leaf triangles) + 192 (hit attributes)
B_RAY_CORNELL = 188 + 8 * 3.44 + 40 * 11.82 + 192
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak


def measured_traffic(rays_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (FETCH_SIZE / WRITE_SIZE are collected
    in separate rocprofv3 --pmc passes — tools/pmc_passes.sh — and cannot be read live here): the measured bytes per ray
    times the rays of this run's launches (a launch covers more samples than the profiled one; bytes per ray do not
    depend on that). None if no profile."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        per_ray = json.load(fh).get("traffic_bytes_per_ray")
    return None if per_ray is None else round(per_ray * rays_per_launch)


def measured_valu():
    """VALU issue-slot occupancy of the dominant kernel from the committed PMC profile (the resource that actually binds it:
    DESIGN.md "Roofline accounting"). None if no profile."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_valu.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        v = json.load(fh)
    return {"valu_issue_frac": v.get("valu_issue_frac"), "lane_utilisation": v.get("lane_utilisation")}


def cpu_baseline():
    """Time the reference's own renderer::render on the host cores: 1080p, 8 bounces, 1 spp (~10 s)."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        # the reference's thread pool stops scaling early (allocator / refcount contention): on the
        # GPU box (256 hardware threads, 16-core share per GPU) 16 threads is its best setting (8: 0.23,
        # 16: 0.65, 32: 0.20, 64: 0.13, 128: 0.11, 256: 0.08 Msamples/s), so that is what is timed
        threads = str(min(16, os.cpu_count() or 1))
        out = subprocess.run([harness, "render", CORNELL, str(W), str(H), "1", str(BOUNCES), threads],
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        return {"value": round(r["msamples_per_s"], 4), "unit": "Msamples/s", "cores": r["threads"], "kind": "reference",
                "sample": f"Cornell {W}x{H}, 1 spp, {BOUNCES} bounces (2.07 M camera paths), renderer::render of the "
                          f"unmodified reference, {r['seconds']:.1f} s"}
    # the compiled reference is absent: time the oracle (my CPU restatement) instead
    from oracle import pt_oracle as ora
    sc = ora.OracleScene(ora.load_gltf(CORNELL))
    t = time.time()
    _, st = sc.render(ora.make_cfg(W, H, 1, BOUNCES), threads=0)
    dt = time.time() - t
    return {"value": round(W * H / dt / 1e6, 4), "unit": "Msamples/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"Cornell {W}x{H}, 1 spp, {BOUNCES} bounces, oracle restatement, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel per GPU per step (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=("samples", "tiles"), default="samples",
                    help="samples: every rank traces --spp samples of every pixel (weak scaling, the default); "
                         "tiles: ONE --spp frame split into row bands across ranks (strong scaling)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # BENCH_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ptx = importlib.import_module("distributed-path-tracer_amd")
    mg = importlib.import_module("distributed-path-tracer_amd.multigpu")
    ctx = ptx.Context(local_rank)
    scene = ptx.Scene.load_gltf(ctx, CORNELL)      # scene is uploaded to HBM here, outside the timed region
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    spp = args.spp

    def step(collect):
        accum.zero_()
        torch.cuda.synchronize()
        # this rank's sample range -> ptx_render (syncs the ctx stream) -> RCCL sum-reduce of the framebuffer onto rank 0
        if args.shard == "tiles":
            st = mg.render_tiles(scene, W, H, spp, BOUNCES, accum, rank, world, want_stats=True)
        else:
            st = mg.render_sharded(scene, W, H, spp, BOUNCES, accum, rank, world, want_stats=True)
        if collect is not None:
            collect.append(st)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(None)
    stats = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(stats)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([sum(s["rays"] for s in stats)], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = float(r.item())
    else:
        total_rays = float(sum(s["rays"] for s in stats))

    if rank == 0:
        weak = args.shard == "samples"
        samples = float(W) * H * spp * (world if weak else 1) * args.steps
        launches = sum(s["passes"] for s in stats)
        kernel_ms = sum(s["kernel_ms"] for s in stats) / max(launches, 1)     # average launch of k_render_pass (HIP events)
        rays_per_launch = sum(s["rays"] for s in stats) / max(launches, 1)
        achieved = rays_per_launch * B_RAY_CORNELL / (kernel_ms * 1e-3) / 1e9  # GB/s, algorithmic
        out = {
            "metric": "Msamples/sec, Cornell box 1920x1080, 256 spp, 8 bounces (camera paths traced per second)",
            "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Cornell box (scenes/cornell-box/cornell.gltf) {W}x{H}, {spp} spp per {'GPU' if weak else 'frame'}, {BOUNCES} bounces, "
                                   f"1xMI355X per rank (BASELINE.json configs[1])",
                       "spp_total": spp * world if weak else spp,
                       "sharding": "none" if world == 1 else ("sample ranges per rank" if weak else "row bands per rank") + " + RCCL sum-reduce of the accumulation buffer"},
            "mrays_per_s": round(total_rays / dt / 1e6, 2),
            "rays_per_sample": round(total_rays / samples, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(rays_per_launch),
                         "algorithmic_bytes_per_launch": round(rays_per_launch * B_RAY_CORNELL),
                         "kernel": "k_render_pass<LDS>", "avg_launch_ms": round(kernel_ms, 4), "launches": launches,
                         "rays_per_launch": round(rays_per_launch), "bytes_per_ray": round(B_RAY_CORNELL, 2),
                         # algorithmic bytes count the geometry fetches that LDS serves, so `frac` can exceed what HBM sees
                         # (`traffic`); the binding resource is VALU issue — from the same PMC profile:
                         "binding": measured_valu()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
