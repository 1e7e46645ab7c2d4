#!/usr/bin/env python3
"""The scene classes of BASELINE.json configs 3-5 (+ the reference's textured asset) through ptx_render on one GPU: Msamples/s, Mrays/s,
the dominant kernel's launch time (HIP events) and its roofline view — algorithmic bytes per ray from BASELINE.md section 7
(profiles/round3_bray.json), hardware counts per ray replayed from the committed PMC profile (profiles/round3_configs_pmc.json, made by
tools/pmc_configs.sh: rocprofv3 --pmc needs separate passes), time measured live. Used by bench.py (its "configs" block) and alone:
   python tools/bench_configs.py [--only config4_atrium_1080p_8b,...] [--spp-scale 1]
Configs 3-5 run on STAND-IN geometry (no bunny in the reference, sponza.bin missing): seeded procedural scenes of the same size class."""
import argparse, glob, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS, N_SIMD, MAX_CLOCK_GHZ = 8000.0, 1024, 2.4   # /opt/skills/guides/MI355X_MICROARCH.md
CONFIGS = {
    "config3_mesh82k_1080p_8b": dict(scene="mesh6", W=1920, H=1080, bounces=8, spp=32, bray="cornell + 81 920",
                                     what="Cornell + 81 920-triangle displaced icosphere (stand-in for the ~70k-triangle bunny), 1920x1080, 8 bounces"),
    "config4_atrium_1080p_8b": dict(scene="atrium", W=1920, H=1080, bounces=8, spp=64, bray="config 4 stand-in",
                                    what="atrium: 262 176 triangles in 24 surfaces of one model, sun (stand-in for Sponza), 1920x1080, 8 bounces"),
    "config5_atrium_4k_16b": dict(scene="atrium", W=3840, H=2160, bounces=16, spp=16, bray="config 5 stand-in",
                                  what="the same scene, 3840x2160, 16 bounces"),
    "jack_of_blades_1080p_8b": dict(scene="jack", W=1920, H=1080, bounces=8, spp=64, bray="jack-of-blades",
                                    what="jack-of-blades (the reference's asset: 58 740 triangles, 17 textures, normal maps, alpha, sun), 1920x1080, 8 bounces"),
}


def load_json(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(files[-1], ROOT)


def build_scene(ptx, ctx, name, cache):
    if name in cache:
        return cache[name]
    proc = importlib.import_module("distributed-path-tracer_amd.procedural")
    t = time.time()
    if name == "jack":
        sc = ptx.Scene.load_gltf(ctx, os.path.join(ROOT, "scenes/jack-of-blades/jack-of-blades.gltf"))
    elif name == "cornell":
        sc = ptx.Scene.load_gltf(ctx, os.path.join(ROOT, "scenes/cornell-box/cornell.gltf"))
    else:
        if name == "atrium":
            d = proc.atrium_scene(5)
        else:
            cornell = build_scene(ptx, ctx, "cornell", cache)[0]
            c = {k: cornell.array(getattr(ptx, "ARR_" + k.upper())) for k in ("model_xform", "model_surf", "surf_range", "vertices", "triangles", "materials", "camera")}
            d = proc.cornell_with_mesh(c, level=6)
        sc = ptx.Scene.from_arrays(ctx, d["model_xform"], d["model_surf"], d["surf_range"], d["vertices"], d["triangles"], d["materials"], d["camera"], d.get("sun"))
    cache[name] = (sc, round(time.time() - t, 2))
    return cache[name]


def run_config(ptx, ctx, name, cache, spp_scale=1.0, timing=True):
    import torch
    cfg = CONFIGS[name]
    scene, build_s = build_scene(ptx, ctx, cfg["scene"], cache)
    W, H, b, spp = cfg["W"], cfg["H"], cfg["bounces"], max(1, int(cfg["spp"] * spp_scale))
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{ctx.device}")
    ctx.set_timing(False)
    scene.render(W, H, spp, b, accum=accum, want_stats=True)     # warm-up: same size, so that every workspace has its final size
    dt = None
    for _ in range(2):      # two timed frames, the faster one: a frame that had to resize a workspace (first frames of a scene) is not what is quoted
        accum.zero_(); torch.cuda.synchronize()
        t = time.perf_counter()
        _, st = scene.render(W, H, spp, b, accum=accum, want_stats=True)
        d1 = time.perf_counter() - t
        dt = d1 if dt is None else min(dt, d1)
    tm = ctx.timing()
    if timing and tm["pipeline"] == 1:     # the per-kernel split of the queue-based pipeline: a run of its own (four event records per step)
        ctx.set_timing(True)
        scene.render(W, H, spp, b, accum=accum, want_stats=True)
        tm = ctx.timing()
        ctx.set_timing(False)
    samples = float(W) * H * spp
    info = scene.info()
    out = {"workload": cfg["what"] + f", {spp} spp in this run", "stand_in_geometry": cfg["scene"] != "jack", "triangles": info["n_triangles"], "kd_nodes": info["n_kd_nodes"],
           "scene_build_s": build_s, "msamples_per_s": round(samples / dt / 1e6, 1), "mrays_per_s": round(st["rays"] / dt / 1e6, 1),
           "rays_per_sample": round(st["rays"] / samples, 4), "seconds": round(dt, 4), "rays": int(st["rays"]),
           "renders_in_process": 3 + (1 if (timing and tm["pipeline"] == 1) else 0)}
    if tm["pipeline"] == 1:
        k_ms = tm["classify_ms"] + tm["traverse_ms"] + tm["shade_ms"]
        out.update({"pipeline": "queue-based (classify / traverse / shade per step of a slab)", "dominant_kernel": "k_wf_traverse2",
                    "kernel_ms": round(tm["traverse_ms"] / max(tm["steps"], 1), 4), "kernel_launches": tm["steps"], "kernel_total_ms": round(tm["traverse_ms"], 3),
                    "kernel_share_of_gpu_time": round(tm["traverse_ms"] / max(k_ms, 1e-9), 4),
                    "classify_total_ms": round(tm["classify_ms"], 3), "shade_total_ms": round(tm["shade_ms"], 3),
                    "slab_paths": tm["slab_paths"], "pool_pairs": tm["pool_pairs"], "peak_pairs": tm["peak_pairs"], "workspace_gb": round(tm["workspace_bytes"] / 1e9, 2),
                    "traverse_drain_frac": round(tm["traverse_drain_frac"], 4)})
        kernel_s = tm["traverse_ms"] * 1e-3
    else:
        out.update({"pipeline": "fused persistent-wave kernel", "dominant_kernel": "k_render_pass", "kernel_ms": round(st["kernel_ms"] / max(st["passes"], 1), 4),
                    "kernel_launches": st["passes"], "kernel_total_ms": round(st["kernel_ms"], 3), "kernel_share_of_gpu_time": 1.0})
        kernel_s = st["kernel_ms"] * 1e-3
    # ---- roofline view of the dominant kernel
    bray, bray_file = load_json("round*_bray.json")
    pmc, pmc_file = load_json("round*_configs_pmc.json")
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    if bray:
        row = next((r for r in bray["scenes"] if cfg["bray"] in r["scene"]), None)
        if row:
            roof.update({"B_ray": row["B_ray"], "B_ray_source": bray_file, "N_branch": row["N_branch"], "N_tri": row["N_tri"], "hit_fraction": row["hit_fraction"],
                         "achieved_algorithmic": round(row["B_ray"] * st["rays"] / max(kernel_s, 1e-9) / 1e9, 1)})
            roof["hbm_frac_algorithmic"] = round(roof["achieved_algorithmic"] / HBM_PEAK_GBS, 4)
    p = (pmc or {}).get("configs", {}).get(name)
    if p:
        traffic = p["hbm_bytes_per_ray"] * st["rays"]
        valu = p["valu_insts_per_ray"] * st["rays"] / max(kernel_s, 1e-9) / 1e9
        roof.update({"traffic": round(traffic), "hbm_frac_counter": round(traffic / max(kernel_s, 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
                     "hbm_bytes_per_ray_counter": round(p["hbm_bytes_per_ray"], 1), "valu_issue_frac": round(valu / (N_SIMD * MAX_CLOCK_GHZ / 2.0), 4),
                     "lanes_on": p.get("lanes_on"), "wait_frac": p.get("wait_frac"), "l2_hit_rate": p.get("l2_hit_rate"), "traffic_source": pmc_file,
                     "counters_stale": p.get("source_sha256_16") != source_hash()})
    else:
        roof.update({"traffic": None, "hbm_frac_counter": None, "valu_issue_frac": None, "lanes_on": None, "traffic_source": None})
    out["roofline"] = roof
    return out


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "wavefront.hip", "device_core.hpp"):
        with open(os.path.join(ROOT, "distributed-path-tracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def run_all(ptx, ctx, only=None, spp_scale=1.0):
    cache, out = {}, {}
    for name in CONFIGS:
        if only and name not in only:
            continue
        out[name] = run_config(ptx, ctx, name, cache, spp_scale)
    for sc, _ in cache.values():
        sc.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--only", default=""); ap.add_argument("--spp-scale", type=float, default=1.0)
    ap.add_argument("--no-timing", action="store_true", help="skip the per-kernel timing run (profiler passes)")
    a = ap.parse_args()
    ptx = importlib.import_module("distributed-path-tracer_amd")
    ctx = ptx.Context(0)
    only = set(filter(None, a.only.split(",")))
    cache = {}
    for name in CONFIGS:
        if only and name not in only:
            continue
        print(json.dumps({name: run_config(ptx, ctx, name, cache, a.spp_scale, timing=not a.no_timing)}), flush=True)
