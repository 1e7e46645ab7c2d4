#!/usr/bin/env python3
"""Measurement: headline workload at different samples-per-launch (ptx_render_cfg.spp_per_pass). python tools/pass_size.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ptx = importlib.import_module("distributed-path-tracer_amd")
ctx = ptx.Context(0)
s = ptx.Scene.load_gltf(ctx, os.path.join(ROOT, "scenes/cornell-box/cornell.gltf"))
W, H, SPP, B = 1920, 1080, 256, 8
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
for pp in (8, 16, 64, 128, 256):
    s.render(W, H, pp, B, accum=acc, spp_per_pass=pp)
    torch.cuda.synchronize()
    t = time.perf_counter()
    _, st = s.render(W, H, SPP, B, accum=acc, spp_per_pass=pp)
    dt = time.perf_counter() - t
    print(f"spp_per_pass {pp:4d}: {W * H * SPP / dt / 1e6:8.1f} Msamples/s  kernel {st['kernel_ms']:.1f} ms of {dt * 1e3:.1f}", flush=True)
