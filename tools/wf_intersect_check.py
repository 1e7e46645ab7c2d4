#!/usr/bin/env python3
"""Queue-based intersection (wavefront.hip) against the fused kernel on the 262k-triangle atrium: bitwise comparison of every output
and rays per second, for camera rays of the 1080p frame and for two generations of bounce rays off the hit points.
   python tools/wf_intersect_check.py [--detail 5]"""
import argparse, ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser(); ap.add_argument("--detail", type=int, default=5); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--scene", default="atrium"); ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--only-wavefront", action="store_true", help="skip the fused kernel (profiling runs)")
ap.add_argument("--stride", type=int, default=1, help="use every stride-th ray (small launches)")
ap.add_argument("--sort", default="none", help="order of the rays of a batch: none | octant (direction octant, stable) | cell (8x8x8 grid cell of the origin, then octant) — what ray coherence is worth")
args = ap.parse_args()
ptx = importlib.import_module("distributed-path-tracer_amd")
proc = importlib.import_module("distributed-path-tracer_amd.procedural")
ctx = ptx.Context(0)
dev = "cuda:0"
if args.scene == "atrium":
    d = proc.atrium_scene(args.detail)
    scene = ptx.Scene.from_arrays(ctx, d["model_xform"], d["model_surf"], d["surf_range"], d["vertices"], d["triangles"], d["materials"], d["camera"], d["sun"])
else:
    scene = ptx.Scene.load_gltf(ctx, os.path.join(ROOT, "scenes", args.scene, args.scene + ".gltf"))
print(json.dumps(scene.info()), flush=True)
cam = scene.array(ptx.ARR_CAMERA)
org, basis, tanh = torch.tensor(cam[:3], device=dev), torch.tensor(cam[3:12], device=dev).reshape(3, 3).T, float(cam[13])
W, H = 1920, 1080
gen = torch.Generator(device=dev); gen.manual_seed(7)
ss, ys, xs = torch.meshgrid(torch.arange(args.spp, device=dev, dtype=torch.float32), torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")     # sample-major, as a render pass orders its paths
dx = ((xs + torch.rand(xs.shape, device=dev, generator=gen)) / W * 2 - 1) * tanh * (W / H)
dy = -((ys + torch.rand(xs.shape, device=dev, generator=gen)) / H * 2 - 1) * tanh
dl = torch.stack([dx, dy, -torch.ones_like(dx)], -1).reshape(-1, 3)
dl = dl / dl.norm(dim=1, keepdim=True)
dirs = dl @ basis.T
dirs = dirs / dirs.norm(dim=1, keepdim=True)
orgs = org.expand_as(dirs)

KEYS = ("distance", "surface", "triangle", "b0", "b1", "b2", "px", "py", "pz", "nx", "ny", "nz", "u", "v")


def intersect(o, d, wavefront):
    os.environ["PTX_WAVEFRONT"] = "1" if wavefront else "0"
    n = o.shape[0]
    oT, dT = o.T.contiguous(), d.T.contiguous()
    out = {k: torch.zeros(n, dtype=torch.int32 if k in ("surface", "triangle") else torch.float32, device=dev) for k in KEYS}
    r = ptx.Rays(*[oT[k].data_ptr() for k in range(3)], *[dT[k].data_ptr() for k in range(3)])
    hh = ptx.Hits(*[out[k].data_ptr() for k, _ in ptx.Hits._fields_])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(args.reps):
        t = time.perf_counter()
        ptx._check(ptx.lib().ptx_intersect_batch(scene.h, C.byref(r), n, C.byref(hh)))
        ctx.synchronize()
        best = min(best, time.perf_counter() - t)
    return out, best


def compare(a, b):
    bad = 0
    for k in KEYS:
        bad += int((a[k].view(torch.int32) != b[k].view(torch.int32)).sum())
    return bad


o, dd = orgs[::args.stride].contiguous(), dirs[::args.stride].contiguous()
def reorder(o, d):
    if args.sort == "none":
        return o, d
    octant = ((d[:, 0] < 0).long() << 2) | ((d[:, 1] < 0).long() << 1) | (d[:, 2] < 0).long()
    key = octant
    if args.sort == "cell":
        lo, hi = o.min(0).values, o.max(0).values
        c = ((o - lo) / (hi - lo + 1e-9) * 8).clamp(0, 7).long()
        key = (((c[:, 0] << 6) | (c[:, 1] << 3) | c[:, 2]) << 3) | octant
    idx = torch.sort(key, stable=True).indices
    return o[idx].contiguous(), d[idx].contiguous()


for generation in range(3):
    o, dd = reorder(o, dd)
    got, t_w = intersect(o, dd, True)
    ref, t_f = (got, t_w) if args.only_wavefront else intersect(o, dd, False)
    n = o.shape[0]
    hit = ref["surface"] >= 0
    print(json.dumps({"rays": "camera" if generation == 0 else "bounce %d" % generation, "n": n, "hit_fraction": round(float(hit.float().mean()), 4),
                      "mismatching_words": compare(ref, got), "fused_mrays_s": round(n / t_f / 1e6, 1), "wavefront_mrays_s": round(n / t_w / 1e6, 1),
                      "speedup": round(t_f / t_w, 2)}), flush=True)
    idx = torch.nonzero(hit).squeeze(1)
    p = torch.stack([ref["px"], ref["py"], ref["pz"]], 1)[idx]
    nn = torch.stack([ref["nx"], ref["ny"], ref["nz"]], 1)[idx]
    inc = dd[idx]
    nn = torch.where((nn * inc).sum(1, keepdim=True) > 0, -nn, nn)     # face the incoming ray
    r = torch.randn(p.shape, device=dev, generator=gen)
    r = r / r.norm(dim=1, keepdim=True)
    r = torch.where((r * nn).sum(1, keepdim=True) < 0, -r, r)
    o, dd = (p + nn * 1e-4).contiguous(), r.contiguous()
