#!/usr/bin/env python3
"""CPU measurement: node steps and triangle tests of a ray's walks on the atrium under four policies (tools/prune_potential.cpp).
   g++ -O2 -std=c++17 -shared -fPIC -o tools/bin/libprune_potential.so tools/prune_potential.cpp && python tools/prune_potential.py"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pt_oracle as ora
from conftest import oracle_from_dict
proc = importlib.import_module("distributed-path-tracer_amd.procedural")
L = C.CDLL(os.path.join(ROOT, "tools/bin/libprune_potential.so"))
L.prune_potential.restype = C.c_uint64
sc = oracle_from_dict(ora, proc.atrium_scene(int(os.environ.get("DETAIL", "5"))))
W, H = 160, 90
ndc = np.array([[(x + 0.5) / W * 2 - 1, 1 - (y + 0.5) / H * 2, W / H] for y in range(H) for x in range(W)], np.float32)
rays = sc.camera_rays(ndc)
rng = np.random.default_rng(3)
for gen in range(3):
    out = np.zeros(12, np.uint64)
    diff = L.prune_potential(sc.h, C.c_size_t(len(rays)), rays.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    o = out.reshape(4, 3).astype(float)
    n = len(rays)
    names = ["no bound (reference)", "final nearest hit known to every walk", "surface order, earlier results known", "box-entry order, earlier results known"]
    for k in range(4):
        print(json.dumps({"rays": ["camera", "bounce 1", "bounce 2"][gen], "n": n, "policy": names[k], "walks_per_ray": round(o[k, 2] / n, 2), "node_steps_per_ray": round(o[k, 0] / n, 1),
                          "tri_tests_per_ray": round(o[k, 1] / n, 1), "lane_loads_vs_reference": round((o[k, 0] + 3 * o[k, 1]) / (o[0, 0] + 3 * o[0, 1]), 3), "rays_with_other_result": int(diff)}))
    if os.environ.get("SURFACES"):
        so = np.zeros(5 * sc.n_surf, np.uint64)
        L.surface_work(sc.h, C.c_size_t(n), rays.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p))
        for u, (w, nn, tt, mx, kd) in enumerate(so.reshape(-1, 5)):
            print(json.dumps({"surface": u, "kd_nodes": int(kd), "pairs_per_ray": round(w / n, 3), "steps_per_pair": round((nn + tt) / max(w, 1), 1), "share_of_steps": round(float(nn + tt) / float(so.reshape(-1, 5)[:, 1:3].sum()), 3), "longest_walk": int(mx)}))
    hit, idx = sc.intersect(rays)
    m = idx >= 0
    pos, nrm = hit[m, 0:3], hit[m, 5:8]
    # cosine-weighted direction around the geometric normal, flipped to the side the ray came from
    d_in = rays[m, 3:6]
    nrm = np.where((np.sum(nrm * d_in, 1) > 0)[:, None], -nrm, nrm)
    u1, u2 = rng.random(len(pos)), rng.random(len(pos))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t1 = np.cross(nrm, a); t1 /= np.linalg.norm(t1, axis=1, keepdims=True); t2 = np.cross(nrm, t1)
    d = t1 * (r * np.cos(phi))[:, None] + t2 * (r * np.sin(phi))[:, None] + nrm * np.sqrt(1 - u1)[:, None]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([pos + d * 1e-4, d], 1).astype(np.float32)
