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


def pack(src_dir, dst):
    arrs = {os.path.basename(f)[:-4]: np.load(f) for f in sorted(glob.glob(os.path.join(src_dir, "*.npy")))}
    np.savez_compressed(dst, **arrs)
    print(f"{dst}: {len(arrs)} arrays, {os.path.getsize(dst) / 1024:.0f} KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-mean", action="store_true")
    ap.add_argument("--n", type=int, default=1024)
    args = ap.parse_args()
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    env = dict(os.environ, ORACLE_SEED="20261004")
    with tempfile.TemporaryDirectory() as tmp:
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
        # a small deterministic PNG from renderer::render itself (single thread + fixed seed => reproducible)
        png = os.path.join(GOLD, "cornell_ref_64x64_16spp_4b.png")
        r = subprocess.check_output([HARNESS, "render", CORNELL, "64", "64", "16", "4", "1", png], env=env)
        print(json.loads(r.decode().strip().splitlines()[-1]))


if __name__ == "__main__":
    main()
