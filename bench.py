#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json configs[1]).

A "step" = one pass of the integrator over one batch of synthetic input: ONE frame of the Cornell-box scene
(scenes/cornell-box, the reference's own asset) at 1920x1080, 256 spp, 8 bounces, entirely through the C ABI (ptx_render).
The scene is resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene cornell|mesh|atrium|atrium4k|jack]

N = 1: one process, one GPU. N > 1: one process per GPU over RCCL (torch.distributed "nccl"); when this script is started from a
plain shell (no WORLD_SIZE) with --gpus N > 1 it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
CHILD process (before anything here touches a GPU), relays rank 0's JSON line and exits with the child's code. Under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE set, as the driver launches it) it is one rank of the job.
Default for N > 1 is STRONG scaling: the same 256-spp frame, its 64x64 image tiles dealt round-robin to the ranks (--shard samples:
each rank traces a contiguous share of the sample indices of every pixel instead), then ONE RCCL sum-reduce of the float32
accumulation buffer onto rank 0 inside the timed region. --weak: every rank traces 256 spp of its own (the frame gets N*256).

--scene picks the frame that is timed (default: the Cornell frame of configs[1]); the other scene classes of BASELINE.json — the 82 k-triangle
mesh (config 3), the 262 k-triangle 24-surface atrium at 1080p / 8 bounces (config 4) and at 4K / 16 bounces (config 5), stand-ins for
the bunny and Sponza the reference does not ship — shard over N ranks the same way (interleaved tiles + one RCCL reduce).

Prints ONE JSON line on rank 0 (bench contract): value = whole-job Msamples/s; plus
  roofline     — dominant kernel (k_render_pass), bound by the resource that limits it (VALU issue), with the HBM view beside it
  psnr_db      — GPU vs CPU oracle on a fixed 1080p tile at equal spp and RNG keys (outside the timed region)
  cpu_baseline — the UNMODIFIED reference renderer (oracle/_ref/ref_harness) timed on this box's host cores: median of 5 runs at 2 spp
  configs      — (N = 1, default scene) configs 3 / 4 / 5 and jack-of-blades rendered after the timed loop, outside it: Msamples/s, Mrays/s,
                 dominant kernel, its launch time and roofline view (tools/bench_configs.py)
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "cornell.gltf")
KERNEL_SRCS = [os.path.join(ROOT, "distributed-path-tracer_amd", "csrc", f) for f in ("kernels.hip", "device_core.hpp")]   # what k_render_pass is compiled from

W, H, SPP, BOUNCES = 1920, 1080, 256, 8
# --scene: (scene key of tools/bench_configs.py, W, H, bounces, default spp, BASELINE config it stands for)
SCENES = {"cornell": ("cornell", 1920, 1080, 8, 256, "configs[1]"),
          "mesh": ("mesh6", 1920, 1080, 8, 64, "configs[2] class: Cornell + 81 920-triangle mesh (stand-in geometry)"),
          "atrium": ("atrium", 1920, 1080, 8, 64, "configs[3] class: 262 176 triangles in 24 surfaces (stand-in geometry)"),
          "atrium4k": ("atrium", 3840, 2160, 16, 16, "configs[4] class: the same scene at 4K, 16 bounces (stand-in geometry)"),
          "jack": ("jack", 1920, 1080, 8, 64, "the reference's textured asset")}
# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E spec peak
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
MAX_CLOCK_GHZ = 2.4            # "Max clock 2400 MHz" (chip-level parameters)
# SURVEY.md §8(d), HBM part only: ray / hit / path-state streams 188 B per ray + 192 B of hit attributes per hit; the KD nodes and
# triangle records (8*3.44 + 40*11.82 = 500 B per ray on the Cornell trees) are served by LDS in this kernel and never reach HBM
B_STREAM, B_ATTR = 188.0, 192.0
B_RAY_CORNELL_ALL = 188 + 8 * 3.44 + 40 * 11.82 + 192   # the full §8(d) figure (880.3), reported for reference


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SRCS:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_profile():
    """Per-ray hardware counts of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/round*_kernel_pmc.json:
    they need separate profiler passes — tools/pmc_passes.sh — and cannot be collected inside this process). Deterministic per
    (binary, workload): instruction and byte counts per ray do not depend on the run; only the TIME is measured live here.
    `stale` says whether kernels.hip changed since the profile was taken."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_kernel_pmc.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        p = json.load(fh)
    p["file"] = os.path.relpath(files[-1], ROOT)
    p["stale"] = kernel_source_hash() != p.get("kernels_hip_sha256_16")
    return p


def cpu_baseline():
    """Time the reference's own renderer::render on the host cores: Cornell 1080p, 8 bounces, 2 spp (4.1 M camera paths, ~15 s each at
    the reference's speed), 5 runs: the median is the value, min / max and every run are listed."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    spp, n_runs = 2, 5
    W, H, BOUNCES = 1920, 1080, 8
    if os.path.exists(harness):
        # the reference's thread pool stops scaling early (allocator / refcount contention): on the GPU box (16-core share
        # per GPU) 16 threads is its best setting (8: 0.23, 16: 0.65, 32: 0.20, 64: 0.13 Msamples/s at 960x540), so that is what is timed
        threads = str(min(16, os.cpu_count() or 1))
        runs = []
        t_all = time.time()
        for _ in range(n_runs):
            out = subprocess.run([harness, "render", CORNELL, str(W), str(H), str(spp), str(BOUNCES), threads],
                                 capture_output=True, text=True, timeout=900)
            runs.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
            if time.time() - t_all > 150 and len(runs) >= 3:     # a slow host: three runs are enough to name a median
                break
        vals = sorted(r["msamples_per_s"] for r in runs)
        med = vals[len(vals) // 2]
        return {"value": round(med, 4), "unit": "Msamples/s", "cores": runs[0]["threads"], "host_cores": os.cpu_count(), "kind": "reference",
                "stat": f"median of {len(runs)} runs", "min": round(vals[0], 4), "max": round(vals[-1], 4),
                "runs": [round(r["msamples_per_s"], 4) for r in runs],
                "sample": f"Cornell {W}x{H}, {spp} spp, {BOUNCES} bounces ({W * H * spp / 1e6:.2f} M camera paths per run), renderer::render of the "
                          f"unmodified reference on {runs[0]['threads']} threads ({', '.join('%.1f s' % r['seconds'] for r in runs)})"}
    # the compiled reference is absent: time the oracle (my CPU restatement) instead
    from oracle import pt_oracle as ora
    sc = ora.OracleScene(ora.load_gltf(CORNELL))
    t = time.time()
    sc.render(ora.make_cfg(W, H, spp, BOUNCES), threads=0)
    dt = time.time() - t
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": os.cpu_count(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"Cornell {W}x{H}, {spp} spp, {BOUNCES} bounces, oracle restatement, {dt:.1f} s"}


def psnr_vs_oracle(ptx, scene, ctx):
    """PSNR (8-bit, after tonemap + sRGB: the reference's output space) of the GPU path against the CPU oracle on a fixed tile of the
    benchmark frame, equal spp, same RNG keys. The oracle is the checker here, never the thing measured."""
    from oracle import pt_oracle as ora
    tile, spp = (832, 420, 192, 108), 8
    mean, _ = ora.OracleScene(ora.load_gltf(CORNELL)).render(ora.make_cfg(W, H, spp, BOUNCES, tile=tile), threads=0)
    accum, _ = scene.render(W, H, spp, BOUNCES, tile=tile)
    return {"psnr_db": round(min(ora.psnr8(ctx.tonemap_encode(accum, tile[2], tile[3], spp), ora.tonemap_write(mean)), 99.0), 2),
            "psnr_sample": f"tile {tile[2]}x{tile[3]} at ({tile[0]},{tile[1]}) of the {W}x{H} frame, {spp} spp, {BOUNCES} bounces, "
                           f"8-bit RGB after tonemap, GPU vs oracle/pt_oracle.cpp"}


def spawn_ranks(args):
    """--gpus N > 1 from a plain shell: run the N ranks as a fresh child job; this process never touches a GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in r.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif r.returncode == 0:
        print("bench.py: the ranks produced no result line", file=sys.stderr)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of the frame (default: 256 for the Cornell frame of the BASELINE config, 16-64 for the others)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 3 / 4 / 5 block after the timed loop")
    ap.add_argument("--scene", choices=sorted(SCENES), default="cornell", help="the frame that is timed (default: BASELINE configs[1])")
    ap.add_argument("--shard", choices=("tiles", "samples"), default="tiles",
                    help="how ONE frame is split over N ranks: interleaved 64x64 image tiles (default; BASELINE.json: \"image tiles shard naturally "
                         "across the 8 GPUs\") or contiguous shares of the sample indices of every pixel")
    ap.add_argument("--spp-per-pass", type=int, default=0, help="samples of every pixel per kernel launch (0 = the library's default)")
    ap.add_argument("--weak", action="store_true", help="weak scaling instead: every rank traces --spp samples of its own")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a box with fewer GPUs than ranks (not a measurement): BENCH_REHEARSAL=1 puts every rank on GPU 0 and reduces over
    # gloo instead of RCCL (RCCL refuses two ranks on one device) — the whole multi-process flow except the collective's transport.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU for local rank {local_rank} (this benchmark has no CPU path)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # BENCH_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ptx = importlib.import_module("distributed-path-tracer_amd")
    mg = importlib.import_module("distributed-path-tracer_amd.multigpu")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs
    ctx = ptx.Context(local_rank)
    scene_key, W, H, BOUNCES, default_spp, stands_for = SCENES[args.scene]
    scene_cache = {}
    scene = bench_configs.build_scene(ptx, ctx, scene_key, scene_cache)[0]      # built and uploaded to HBM here, outside the timed region
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    spp = args.spp or default_spp
    headline = args.scene == "cornell"
    mode = "weak" if args.weak else args.shard

    def step(collect):
        accum.zero_()
        torch.cuda.synchronize()
        # this rank's share -> ptx_render -> (N > 1) RCCL sum-reduce of the framebuffer onto rank 0
        if mode == "weak":
            st = mg.render_sharded(scene, W, H, spp, BOUNCES, accum, rank, world, want_stats=True, spp_per_pass=args.spp_per_pass)
        elif mode == "tiles":
            st = mg.render_tiles(scene, W, H, spp, BOUNCES, accum, rank, world, want_stats=True, spp_per_pass=args.spp_per_pass)
        else:
            st = mg.render_samples(scene, W, H, spp, BOUNCES, accum, rank, world, want_stats=True, spp_per_pass=args.spp_per_pass)
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
    my_rays = float(sum(s["rays"] for s in stats))
    my_kernel_ms = float(sum(s["kernel_ms"] for s in stats)) / max(args.steps, 1)     # this rank's integrator time per frame: the tile load balance
    rank_kernel_ms = [my_kernel_ms]
    if world > 1:
        k = torch.zeros(world, dtype=torch.float64, device="cpu" if rehearsal else accum.device)
        k[rank] = my_kernel_ms
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        rank_kernel_ms = [round(float(v), 3) for v in k.tolist()]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([my_rays], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = float(r.item())
    else:
        total_rays = my_rays

    if rank == 0:
        spp_total = spp * world if mode == "weak" else spp
        samples = float(W) * H * spp_total * args.steps
        launches = sum(s["passes"] for s in stats)
        kernel_ms = sum(s["kernel_ms"] for s in stats) / max(launches, 1)     # average launch of k_render_pass on rank 0 (HIP events on the ctx stream)
        rays_per_launch = my_rays / max(launches, 1)
        bray, bray_file = bench_configs.load_json("round*_bray.json")            # the oracle's per-scene counters
        hit_frac = next((r["hit_fraction"] for r in (bray or {}).get("scenes", []) if r["scene"].startswith("cornell (")), 1.0)
        prof = kernel_profile() if headline else None
        # --- binding resource: VALU issue. Peak = one wave64 VALU instruction per SIMD every 2 cycles (measured: v_add_f32 2.00 cycles per
        # instruction per SIMD at >= 2 waves per SIMD, profiles/round2_valu_issue.txt) at the clock the kernel held; achieved = the kernel's
        # wave64 VALU instructions per second (count per ray from the PMC profile x this run's rays / this run's kernel time).
        roof = {"bound": "valu", "kernel": "k_render_pass<LDS>" if headline else "see the configs block of the default run (tools/bench_configs.py)",
                "avg_launch_ms": round(kernel_ms, 4), "launches": launches,
                "rays_per_launch": round(rays_per_launch)}
        hbm_alg = rays_per_launch * (B_STREAM + hit_frac * B_ATTR)
        hbm = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "algorithmic_bytes_per_ray": round(B_STREAM + hit_frac * B_ATTR, 1), "hit_fraction": hit_frac, "hit_fraction_source": bray_file,
               "algorithmic_note": "SURVEY §8(d) without the geometry term: streams 188 B + 192 B x hit fraction; KD nodes / triangle records "
                                   f"(500 B per ray of the {B_RAY_CORNELL_ALL:.0f} B figure) are LDS-resident and never reach HBM",
               "achieved": round(hbm_alg / (kernel_ms * 1e-3) / 1e9, 1)}
        hbm["frac"] = round(hbm["achieved"] / HBM_PEAK_GBS, 4)
        if prof:
            valu_per_ray = prof["valu_insts_per_ray"]
            achieved = valu_per_ray * rays_per_launch / (kernel_ms * 1e-3) / 1e9          # G wave-instructions / s
            peak = N_SIMD * MAX_CLOCK_GHZ / 2.0                                            # at the chip's maximum clock: the fraction can only be understated
            traffic = prof["hbm_bytes_per_ray"] * rays_per_launch
            roof.update({"achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G wave64-VALU-instr/s", "frac": round(achieved / peak, 4),
                         "peak_note": f"{N_SIMD} SIMDs x {MAX_CLOCK_GHZ} GHz (max clock) / 2 cycles per wave64 instruction; in cycles of the profiled launch "
                                      f"({prof.get('shader_clock_ghz')} GHz under the profiler) the same ratio is {prof.get('valu_issue_frac_2cycle')}",
                         "busy_frac_mix_weighted": prof.get("valu_busy_frac_mix_weighted"),
                         "lane_utilisation": prof.get("lane_utilisation"),
                         "traffic": round(traffic), "traffic_source": prof["file"], "counters_stale": prof["stale"]})
            hbm.update({"traffic_bytes_per_launch": round(traffic), "traffic_gbs": round(traffic / (kernel_ms * 1e-3) / 1e9, 1),
                        "traffic_frac_of_peak": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        else:
            roof.update({"achieved": None, "peak": None, "unit": "G wave64-VALU-instr/s", "frac": None, "traffic": None,
                         "traffic_source": None})
        roof["hbm"] = hbm
        what = {"weak": f"{spp} spp per GPU (weak scaling: {spp_total} spp per frame)",
                "samples": f"{spp} spp per frame" + (f", sample indices split over {world} GPUs (strong scaling)" if world > 1 else ""),
                "tiles": f"{spp} spp per frame" + (f", interleaved 64x64 tiles over {world} GPUs (strong scaling)" if world > 1 else "")}[mode]
        out = {
            "metric": f"Msamples/sec, {'Cornell box' if headline else args.scene} {W}x{H}, {what}, {BOUNCES} bounces (camera paths traced per second)",
            "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if mode == "weak" else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" — REHEARSAL: all ranks share GPU 0, gloo reduce; not a measurement" if rehearsal else ""),
            "config": {"workload": (f"Cornell box (scenes/cornell-box/cornell.gltf) {W}x{H}, {spp_total} spp per frame, {BOUNCES} bounces "
                                    f"(BASELINE.json configs[1]) on {world} x MI355X") if headline else
                                   f"{args.scene}: {stands_for}; {W}x{H}, {spp_total} spp per frame, {BOUNCES} bounces on {world} x MI355X",
                       "spp_total": spp_total,
                       "sharding": "none" if world == 1 else
                                   {"weak": "a sample range of its own per rank", "samples": "contiguous shares of the frame's sample indices per rank",
                                    "tiles": "interleaved 64x64 tiles per rank"}[mode] + " + ONE RCCL sum-reduce of the accumulation buffer"},
            "mrays_per_s": round(total_rays / dt / 1e6, 2),
            "rays_per_sample": round(total_rays / samples, 4),
            "roofline": roof,
        }
        if world > 1:
            out["rank_kernel_ms_per_frame"] = rank_kernel_ms      # integrator time per rank (HIP events): how evenly the tiles / samples were dealt
            out["rank_kernel_ms_spread"] = round(max(rank_kernel_ms) / max(min(rank_kernel_ms), 1e-9), 4)
        if world == 1 and headline and not args.no_configs:
            # configs 3 / 4 / 5 + the textured asset: same process, after the timed loop and outside it (<= 60 s with the scene builds)
            t_cfg = time.time()
            del accum
            torch.cuda.empty_cache()
            out["configs"] = bench_configs.run_all(ptx, ctx)
            out["configs_seconds"] = round(time.time() - t_cfg, 1)
        if not args.no_psnr and headline:
            out.update(psnr_vs_oracle(ptx, scene, ctx))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
