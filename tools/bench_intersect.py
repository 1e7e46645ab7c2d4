#!/usr/bin/env python3
"""Measurement: throughput of ptx_intersect_batch (the INTERSECT stage-queue unit) on device-resident SoA rays.
   python tools/bench_intersect.py [n_rays]"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
ptx = importlib.import_module("distributed-path-tracer_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
ctx = ptx.Context(0)
s = ptx.Scene.load_gltf(ctx, os.path.join(ROOT, "scenes/cornell-box/cornell.gltf"))
g = torch.Generator(device="cuda:0").manual_seed(1)
# rays from inside the room in random directions
o = torch.stack([torch.rand(n, generator=g, device="cuda:0") * 1.6 - 0.8, torch.rand(n, generator=g, device="cuda:0") * 1.6 + 0.1,
                 torch.rand(n, generator=g, device="cuda:0") * 1.6 - 0.8]).contiguous()
d = torch.randn(3, n, generator=g, device="cuda:0")
d = (d / d.norm(dim=0, keepdim=True)).contiguous()
outf = {k: torch.zeros(n, dtype=torch.float32, device="cuda:0") for k in ("distance", "b0", "b1", "b2", "px", "py", "pz", "nx", "ny", "nz", "u", "v")}
outi = {k: torch.zeros(n, dtype=torch.int32, device="cuda:0") for k in ("surface", "triangle")}
r = ptx.Rays(*[o[k].data_ptr() for k in range(3)], *[d[k].data_ptr() for k in range(3)])
h = ptx.Hits(*[(outf[k].data_ptr() if k in outf else outi[k].data_ptr()) for k, _ in ptx.Hits._fields_])
L = ptx.lib()
for _ in range(2):
    assert L.ptx_intersect_batch(s.h, C.byref(r), n, C.byref(h)) == 0
ctx.synchronize(); torch.cuda.synchronize()
t = time.perf_counter()
K = 5
for _ in range(K):
    assert L.ptx_intersect_batch(s.h, C.byref(r), n, C.byref(h)) == 0
ctx.synchronize()
dt = (time.perf_counter() - t) / K
print(f"ptx_intersect_batch: {n / dt / 1e9:.2f} Grays/s ({n} rays, hit fraction {(outi['surface'] >= 0).float().mean().item():.3f}, with attributes)")
