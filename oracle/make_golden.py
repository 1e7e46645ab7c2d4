#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerate tests/golden/*.npz from the UNMODIFIED reference.

Runs oracle/_ref/ref_harness (the reference library compiled from /root/reference by
`make -C oracle ref`; harness source: oracle/ref_harness.cpp) on the Cornell scene and packs its
.npy outputs into compressed .npz fixtures. Only possible where /root/reference exists; the
fixtures themselves are committed so the tests run anywhere.

    python oracle/make_golden.py            # all fixtures (about 2 minutes)
    python oracle/make_golden.py --no-mean  # skip the converged mean images
"""
import argparse
import glob
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "cornell.gltf")
JACK = os.path.join(ROOT, "scenes", "jack-of-blades", "jack-of-blades.gltf")
ENV_PNG = os.path.join(ROOT, "scenes", "jack-of-blades", "textures", "TORSO_baseColor.png")   # any PNG serves as an environment map   # derived asset, see tools/make_jack_asset.py


def kd_stream(t, axis, split, left, right, first, count, refs):
    """Canonical uint32 stream of a pre-order KD dump (one surface): branch = (0, axis, split bits, has_left, has_right),
    leaf = (1, count, mesh-local triangle ids...). tests/conftest.py builds the same stream from the oracle and the product."""
    out = []
    sb = np.ascontiguousarray(split, np.float32).view(np.uint32)
    for i in range(len(t)):
        if t[i] == 1:
            f, c = int(first[i]), int(count[i])
            out.append(np.concatenate([[1, c], refs[f:f + c]]).astype(np.uint32))
        else:
            out.append(np.array([0, axis[i], sb[i], left[i] >= 0, right[i] >= 0], np.uint32))
    return np.concatenate(out) if out else np.zeros(0, np.uint32)


def sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def pack_big_scene(src_dir, dst):
    """Scene too large to commit array by array: small arrays verbatim, SHA-256 digests of the big ones."""
    a = {os.path.basename(f)[:-4]: np.load(f) for f in sorted(glob.glob(os.path.join(src_dir, "*.npy")))}
    out = {k: a[k] for k in ("model_xform", "model_aabb", "model_surf", "mesh_aabb", "surf_range", "materials", "material_tex",
                             "camera", "sun", "model_names", "kd_depth")}
    out["sha_vertices"] = sha(a["vertices"]); out["sha_triangles"] = sha(a["triangles"])
    kd = []
    for srow in a["surf_range"]:
        k0, nk, r0, nr = (int(v) for v in srow[4:8])
        sl = slice(k0, k0 + nk)
        refs = a["kd_refs"][r0:r0 + nr]
        kd.append(sha(kd_stream(a["kd_type"][sl], a["kd_axis"][sl], a["kd_split"][sl], a["kd_left"][sl], a["kd_right"][sl],
                                a["kd_first"][sl] - r0, a["kd_count"][sl], refs)))
    out["sha_kd"] = np.stack(kd)
    np.savez_compressed(dst, **out)
    print(f"{dst}: {os.path.getsize(dst) / 1024:.0f} KiB")


def pack(src_dir, dst):
    arrs = {os.path.basename(f)[:-4]: np.load(f) for f in sorted(glob.glob(os.path.join(src_dir, "*.npy")))}
    np.savez_compressed(dst, **arrs)
    print(f"{dst}: {len(arrs)} arrays, {os.path.getsize(dst) / 1024:.0f} KiB")


SPONZA_TEX = "/root/reference/path-tracer-core/scenes/sponza-new/textures"


def make_jpeg_fixtures(env):
    """JPEG textures (image::image::load -> stb_image v2.30): decoded pixels and bilinear lookups of the compiled reference on
    (a) two of the reference's own Sponza textures, copied as data, (b) small synthetic files written here with Pillow that cover what
    the 63 Sponza files (all baseline 4:4:4) do not: 4:2:0 / 4:2:2 / 4:1:1 chroma, progressive, restart intervals, grey, odd sizes,
    extreme quality; (c) SHA-256 digests of the decoded pixels of ALL 63 Sponza JPEGs (checked where the reference tree is present)."""
    import hashlib
    import shutil
    from PIL import Image
    jd = os.path.join(GOLD, "jpeg")
    os.makedirs(jd, exist_ok=True)
    rng = np.random.default_rng(3)

    def img(w, h):
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([127 + 120 * np.sin(x / 7.0) * np.cos(y / 5.0), 255 * x / max(w - 1, 1), 255 * y / max(h - 1, 1)], -1) + rng.normal(0, 12, (h, w, 3))
        return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8), "RGB")
    cases = {"s444": dict(size=(67, 45), subsampling=0, quality=90), "s420": dict(size=(67, 45), subsampling=2, quality=85),
             "s422": dict(size=(64, 48), subsampling=1, quality=75), "s411": dict(size=(70, 50), subsampling="4:1:1", quality=80),
             "s420_1x1": dict(size=(1, 1), subsampling=2, quality=90), "s420_odd": dict(size=(17, 9), subsampling=2, quality=50),
             "prog444": dict(size=(67, 45), subsampling=0, quality=88, progressive=True), "prog420": dict(size=(70, 41), subsampling=2, quality=80, progressive=True),
             "opt420": dict(size=(96, 64), subsampling=2, quality=95, optimize=True), "q10": dict(size=(80, 56), subsampling=2, quality=10),
             "q100": dict(size=(40, 40), subsampling=0, quality=100), "rst_blocks": dict(size=(100, 75), subsampling=2, quality=80, restart_marker_blocks=3),
             "rst_rows": dict(size=(100, 75), subsampling=0, quality=80, restart_marker_rows=1),
             "rst_prog": dict(size=(90, 70), subsampling=2, quality=80, restart_marker_rows=1, progressive=True)}
    for n, c in cases.items():
        sz = c.pop("size")
        img(*sz).save(os.path.join(jd, n + ".jpg"), **c)
    img(61, 47).convert("L").save(os.path.join(jd, "gray.jpg"), quality=80)
    img(61, 47).convert("L").save(os.path.join(jd, "gray_prog.jpg"), quality=80, progressive=True)
    for k, name in enumerate(("16885566240357350108.jpg", "8503262930880235456.jpg")):   # the two smallest Sponza textures (12 KB, 102 KB)
        shutil.copyfile(os.path.join(SPONZA_TEX, name), os.path.join(jd, f"sponza_{k}.jpg"))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(glob.glob(os.path.join(jd, "*.jpg"))):
            tag = os.path.basename(f)[:-4]
            subprocess.check_call([HARNESS, "image", f, tmp, "21", "192"], env=env)
            px = np.load(os.path.join(tmp, "pixels.npy"))
            out[tag + "_shape"] = np.array(px.shape, np.int32)
            out[tag + "_sha"] = sha(px)
            if px.size <= 40000:
                out[tag + "_pixels"] = px
            for k in ("uv", "sample_linear", "sample_srgb"):
                out[f"{tag}_{k}"] = np.load(os.path.join(tmp, k + ".npy"))
        names, digests = [], []
        for f in sorted(glob.glob(os.path.join(SPONZA_TEX, "*.jpg"))):
            subprocess.check_call([HARNESS, "image", f, tmp, "1", "1"], env=env)
            names.append(os.path.basename(f)); digests.append(sha(np.load(os.path.join(tmp, "pixels.npy"))))
        out["sponza_names"] = np.array(names)
        out["sponza_sha"] = np.stack(digests)
    np.savez_compressed(os.path.join(GOLD, "jpeg_vectors.npz"), **out)
    print(f"jpeg_vectors.npz written ({os.path.getsize(os.path.join(GOLD, 'jpeg_vectors.npz')) / 1024:.0f} KiB), {len(names)} Sponza digests")


def write_hdr(path, img, rle):
    """Radiance RGBE writer for the fixtures (img float32 [H, W, 3]): new-style run-length scanlines when `rle` (needs 8 <= W < 32768),
    flat quadruples otherwise. What the bytes decode to is taken from the reference, not from this encoder."""
    h, w = img.shape[:2]
    v = img.max(-1)
    m, e = np.frexp(v)
    scale = np.where(v > 1e-32, m * 256.0 / np.maximum(v, 1e-38), 0.0)
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(v > 1e-32, e + 128, 0).astype(np.uint8)
    out = bytearray(b"#?RADIANCE\n# synthetic fixture\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
    if not rle:
        out += rgbe.tobytes()
    else:
        for j in range(h):
            out += bytes([2, 2, w >> 8, w & 255])
            for k in range(4):
                row = rgbe[j, :, k]
                i = 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and row[i + run] == row[i]:
                        run += 1
                    if run >= 4:
                        out += bytes([128 + run, int(row[i])]); i += run
                    else:
                        n = 1
                        while i + n < w and n < 128 and not (i + n + 3 < w and row[i + n] == row[i + n + 1] == row[i + n + 2] == row[i + n + 3]):
                            n += 1
                        out += bytes([n]) + row[i:i + n].tobytes(); i += n
    with open(path, "wb") as fh:
        fh.write(bytes(out))


def make_hdr_fixtures(env):
    """Radiance .hdr images (image::image::load's HDR branch, stbi_loadf): decoded floats, bilinear lookups (linear and sRGB) and
    environment-map lookups + trace() on misses of the compiled reference, on synthetic files written here."""
    hd = os.path.join(GOLD, "hdr")
    os.makedirs(hd, exist_ok=True)
    rng = np.random.default_rng(8)
    y, x = np.mgrid[0:32, 0:64]
    sky = np.stack([0.4 + 0.3 * np.sin(x / 9.0), 0.5 + 0.4 * y / 31.0, 0.9 - 0.5 * y / 31.0], -1).astype(np.float32)
    sky[4:8, 40:46] = (900.0, 700.0, 350.0)             # a small very bright "sun": the point of an HDR map
    sky[20:, :] = np.round(sky[20:, :] * 4) / 4           # flat areas: long runs
    sky += (rng.random(sky.shape) * 0.02).astype(np.float32) * (y[..., None] < 20)
    write_hdr(os.path.join(hd, "sky_rle.hdr"), sky, True)
    write_hdr(os.path.join(hd, "sky_flat.hdr"), sky, False)                    # W >= 8 but not run-length encoded: the first-pixel fallback
    write_hdr(os.path.join(hd, "tiny.hdr"), (rng.random((5, 6, 3)) * np.float32(3)).astype(np.float32), False)   # W < 8: always flat
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(glob.glob(os.path.join(hd, "*.hdr"))):
            tag = os.path.basename(f)[:-4]
            subprocess.check_call([HARNESS, "image", f, tmp, "31", "256"], env=env)
            for k in ("pixels", "uv", "sample_linear", "sample_srgb"):
                out[f"{tag}_{k}"] = np.load(os.path.join(tmp, k + ".npy"))
        for srgb in (0, 1):
            d = os.path.join(tmp, f"env{srgb}")
            subprocess.check_call([HARNESS, "envmap", CORNELL, os.path.join(hd, "sky_rle.hdr"), str(srgb), d, "9", "384"], env=env)
            for k in ("env_in", "env_uv", "env_out", "env_trace"):
                out[f"srgb{srgb}_{k}"] = np.load(os.path.join(d, k + ".npy"))
    np.savez_compressed(os.path.join(GOLD, "hdr_vectors.npz"), **out)
    print(f"hdr_vectors.npz written ({os.path.getsize(os.path.join(GOLD, 'hdr_vectors.npz')) / 1024:.0f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-mean", action="store_true")
    ap.add_argument("--only-trace", action="store_true", help="regenerate trace_vectors.npz only")
    ap.add_argument("--only-jpeg", action="store_true", help="regenerate tests/golden/jpeg/* and jpeg_vectors.npz only")
    ap.add_argument("--only-hdr", action="store_true", help="regenerate tests/golden/hdr/* and hdr_vectors.npz only")
    ap.add_argument("--n", type=int, default=1024)
    args = ap.parse_args()
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    env = dict(os.environ, ORACLE_SEED="20261004")
    if args.only_jpeg:
        make_jpeg_fixtures(env)
        return
    if args.only_hdr:
        make_hdr_fixtures(env)
        return
    with tempfile.TemporaryDirectory() as tmp:
        # ---- renderer::trace itself: n rays traced one after the other by ONE thread on ONE seeded mt19937 stream. Pins the
        # integrator's COMPOSITION (draw order, lobe choice, sun block, BRDF / PDF combine, clamp, emissive x 10, opacity
        # pass-through) bit for bit: the oracle replays the same stream (ora_trace_mt).
        out = {}
        for tag, gltf, seed, pcg, n, b in (("cornell", CORNELL, "424242", "9", 2000, 8), ("jack", JACK, "777", "10", 4000, 6)):
            d = os.path.join(tmp, "trace_" + tag)
            subprocess.check_call([HARNESS, "trace", gltf, d, pcg, str(n), str(b)], env=dict(env, ORACLE_SEED=seed))
            out[tag + "_rays"] = np.load(os.path.join(d, "trace_rays.npy"))
            out[tag + "_out"] = np.load(os.path.join(d, "trace_out.npy"))
            out[tag + "_meta"] = np.load(os.path.join(d, "trace_meta.npy"))   # mt19937 seed, bounces, random_device calls (must be 1)
        np.savez_compressed(os.path.join(GOLD, "trace_vectors.npz"), **out)
        print(f"trace_vectors.npz written ({os.path.getsize(os.path.join(GOLD, 'trace_vectors.npz')) / 1024:.0f} KiB)")
        if args.only_trace:
            return
        d = os.path.join(tmp, "scene")
        subprocess.check_call([HARNESS, "scene", CORNELL, d], env=env)
        pack(d, os.path.join(GOLD, "cornell_scene.npz"))
        d = os.path.join(tmp, "vec")
        subprocess.check_call([HARNESS, "vectors", CORNELL, d, "1", str(args.n)], env=env)
        pack(d, os.path.join(GOLD, "cornell_vectors.npz"))
        if not args.no_mean:
            # converged float32 mean images straight from renderer::trace (two independent halves each,
            # so tests can derive the noise bound from the reference itself)
            out = {}
            for tag, (W, H, spp, b) in {"b4": (64, 64, 2048, 4), "b8": (48, 48, 1536, 8)}.items():
                for half, seed in (("a", "111"), ("b", "222")):
                    f = os.path.join(tmp, f"mean_{tag}{half}.npy")
                    subprocess.check_call([HARNESS, "mean", CORNELL, f, str(W), str(H), str(spp), str(b), "8"],
                                          env=dict(env, ORACLE_SEED=seed))
                    out[f"{tag}_{half}"] = np.load(f)
                out[f"{tag}_cfg"] = np.array([W, H, spp, b], np.int32)
            np.savez_compressed(os.path.join(GOLD, "cornell_mean.npz"), **out)
            print("cornell_mean.npz written")
        # ---- the textured, sun-lit asset (58 740 triangles, 17 textures): scene digests, material / intersection vectors, mean image
        d = os.path.join(tmp, "jscene")
        subprocess.check_call([HARNESS, "scene", JACK, d], env=env)
        pack_big_scene(d, os.path.join(GOLD, "jack_scene.npz"))
        d = os.path.join(tmp, "jvec")
        subprocess.check_call([HARNESS, "vectors", JACK, d, "2", "192"], env=env)
        subprocess.check_call([HARNESS, "materials", JACK, d, "5", "256"], env=env)
        keep = ("mesh_in", "mesh_out", "mesh_idx", "world_rays", "model_out", "model_idx", "scene_out", "scene_idx", "cam_in", "cam_out",
                "mat_in", "mat_out")
        np.savez_compressed(os.path.join(GOLD, "jack_vectors.npz"), **{k: np.load(os.path.join(d, k + ".npy")) for k in keep})
        print("jack_vectors.npz written")
        if not args.no_mean:
            out = {}
            W, H, spp, b = 64, 36, 384, 4
            for half, seed in (("a", "333"), ("b", "444")):
                f = os.path.join(tmp, f"jmean_{half}.npy")
                subprocess.check_call([HARNESS, "mean", JACK, f, str(W), str(H), str(spp), str(b), "8"], env=dict(env, ORACLE_SEED=seed))
                out[f"b4_{half}"] = np.load(f)
            out["b4_cfg"] = np.array([W, H, spp, b], np.int32)
            np.savez_compressed(os.path.join(GOLD, "jack_mean.npz"), **out)
            print("jack_mean.npz written")
        # environment map: equirectangular_proj + image_texture::sample + trace() on misses (renderer.cpp:443-449)
        d = os.path.join(tmp, "env")
        subprocess.check_call([HARNESS, "envmap", CORNELL, ENV_PNG, "1", d, "7", "512"], env=env)
        pack(d, os.path.join(GOLD, "env_vectors.npz"))
        make_jpeg_fixtures(env)
        make_hdr_fixtures(env)
        # a small deterministic PNG from renderer::render itself (single thread + fixed seed => reproducible)
        png = os.path.join(GOLD, "cornell_ref_64x64_16spp_4b.png")
        r = subprocess.check_output([HARNESS, "render", CORNELL, "64", "64", "16", "4", "1", png], env=env)
        print(json.loads(r.decode().strip().splitlines()[-1]))


if __name__ == "__main__":
    main()
