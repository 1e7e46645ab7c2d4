#!/usr/bin/env python3
"""Derive the test asset scenes/jack-of-blades/ from the reference's own scene
(/root/reference/path-tracer-core/scenes/jack-of-blades): same glTF and geometry buffer, the 17 PNG textures
box-filtered from 1024x1024 (256x256 for the two Glow maps) down to 256x256 (64x64) so that the repository and the
GPU-box snapshot stay small (23 MB -> 4 MB). Sizes stay powers of two (the reference's bilinear wrap, quirk Q3,
is only well-defined for those). Golden vectors are produced by the compiled reference on THIS derived asset.
Deterministic (PIL box reduce by an integer factor)."""
import os
import shutil

from PIL import Image

SRC = "/root/reference/path-tracer-core/scenes/jack-of-blades"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "jack-of-blades")
os.makedirs(os.path.join(DST, "textures"), exist_ok=True)
for f in ("jack-of-blades.gltf", "jack-of-blades.bin"):
    shutil.copyfile(os.path.join(SRC, f), os.path.join(DST, f))
for f in sorted(os.listdir(os.path.join(SRC, "textures"))):
    im = Image.open(os.path.join(SRC, "textures", f))
    out = im.reduce(4)                       # integer box filter, keeps the mode (RGB / RGBA)
    out.save(os.path.join(DST, "textures", f), optimize=True)
    print(f, im.size, im.mode, "->", out.size, os.path.getsize(os.path.join(DST, "textures", f)) // 1024, "KB")
