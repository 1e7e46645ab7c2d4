#!/usr/bin/env python3
"""the algorithmic bytes per ray of each benchmark scene, B_ray = 188 + 8 N_branch + 40 N_tri + 192 h, from the
ORACLE's traversal counters (oracle/pt_oracle.cpp: branches visited, triangle tests and hits per renderer::intersect call on the
reference-topology trees). CPU only; the oracle is the checker here, nothing of the product runs.
   python tools/freeze_bray.py            -> prints the table of BASELINE.md and writes profiles/round3_bray.json"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pt_oracle as ora
from conftest import oracle_from_dict
proc = importlib.import_module("distributed-path-tracer_amd.procedural")

CORNELL = os.path.join(ROOT, "scenes/cornell-box/cornell.gltf")
JACK = os.path.join(ROOT, "scenes/jack-of-blades/jack-of-blades.gltf")


def cornell_dict():
    a = ora.load_gltf(CORNELL)
    return {"model_xform": a.model_xform, "model_surf": a.model_surf, "surf_range": a.surf_range, "vertices": a.vertices, "triangles": a.triangles,
            "materials": a.materials, "camera": a.camera}


def measure(name, scene, W, H, bounces, spp, tile):
    t = time.time()
    cfg = ora.make_cfg(W, H, spp, bounces, tile=tile)
    _, st = scene.render(cfg, threads=0, stats=True)
    rays, branches, tris, hits = float(st[0]), float(st[3]), float(st[5]), float(st[7])
    n_branch, n_tri, h = branches / rays, tris / rays, hits / rays
    b_ray = 188 + 8 * n_branch + 40 * n_tri + 192 * h
    samples = tile[2] * tile[3] * spp
    out = {"scene": name, "frame": f"{W}x{H}, {bounces} bounces", "sample": f"tile {tile[2]}x{tile[3]} at ({tile[0]},{tile[1]}), {spp} spp = {samples} camera paths, {int(rays)} rays",
           "rays_per_sample": round(rays / samples, 4), "mesh_tests_per_ray": round(float(st[2]) / rays, 3), "N_branch": round(n_branch, 3), "N_tri": round(n_tri, 3),
           "hit_fraction": round(h, 4), "B_ray": round(b_ray, 1), "hbm_roofline_grays_per_s": round(8e12 / b_ray / 1e9, 2), "oracle_seconds": round(time.time() - t, 1)}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    rows = []
    # whole-frame coverage at reduced density: every 4th pixel block is not possible through the tile interface, so a centred tile
    # of 1/4 x 1/4 of the frame plus the four corners would bias; the frame is rendered at 1/4 resolution instead (same camera, same
    # field of view: the ray distribution over the scene is the frame's)
    c = ora.OracleScene(ora.load_gltf(CORNELL))
    rows.append(measure("cornell (configs 1, 2)", c, 480, 270, 8, 4, (0, 0, 480, 270)))
    d6 = proc.cornell_with_mesh(cornell_dict(), level=6)
    rows.append(measure("cornell + 81 920-triangle displaced icosphere (config 3 stand-in)", oracle_from_dict(ora, d6), 480, 270, 8, 2, (0, 0, 480, 270)))
    atr = oracle_from_dict(ora, proc.atrium_scene(5))
    rows.append(measure("atrium, 262 176 triangles in 24 surfaces (config 4 stand-in)", atr, 480, 270, 8, 2, (0, 0, 480, 270)))
    rows.append(measure("atrium, 262 176 triangles in 24 surfaces (config 5 stand-in)", atr, 480, 270, 16, 2, (0, 0, 480, 270)))
    rows.append(measure("jack-of-blades (the reference's asset)", ora.OracleScene(ora.load_gltf(JACK)), 480, 270, 8, 4, (0, 0, 480, 270)))
    with open(os.path.join(ROOT, "profiles", "round3_bray.json"), "w") as fh:
        json.dump({"definition": "SURVEY.md section 8(d): B_ray = 188 (ray / hit / path-state streams) + 8 N_branch (8-byte KD nodes visited) + 40 N_tri (36-byte vertex triple + 4-byte id per "
                                 "triangle test) + 192 h (hit attributes); N_branch, N_tri, h per renderer::intersect call from the oracle's counters on the reference-topology trees",
                   "scenes": rows}, fh, indent=1)
