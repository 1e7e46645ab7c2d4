import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "cornell.gltf")
JACK = os.path.join(ROOT, "scenes", "jack-of-blades", "jack-of-blades.gltf")   # derived from the reference's asset (tools/make_jack_asset.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold_scene():
    return dict(np.load(os.path.join(GOLD, "cornell_scene.npz")))


@pytest.fixture(scope="session")
def gold_vec():
    return dict(np.load(os.path.join(GOLD, "cornell_vectors.npz")))


@pytest.fixture(scope="session")
def gold_mean():
    return dict(np.load(os.path.join(GOLD, "cornell_mean.npz")))


@pytest.fixture(scope="session")
def ptx():
    import importlib
    return importlib.import_module("distributed-path-tracer_amd")


@pytest.fixture(scope="session")
def ora():
    from oracle import pt_oracle
    pt_oracle.lib()
    return pt_oracle


@pytest.fixture(scope="session")
def cornell_arrays(ora):
    return ora.load_gltf(CORNELL)


@pytest.fixture(scope="session")
def cornell_oracle(ora, cornell_arrays):
    return ora.OracleScene(cornell_arrays)


def ulp_diff(a, b):
    """Distance in units-in-the-last-place between two float32 arrays (NaN == NaN, +0 == -0)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def oracle_from_dict(ora, d):
    """Oracle scene from the flat-array dict of distributed-path-tracer_amd.procedural (same arrays the product takes)."""
    a = ora.SceneArrays()
    a.model_xform = np.ascontiguousarray(d["model_xform"], np.float32)
    a.model_surf = np.ascontiguousarray(d["model_surf"], np.int32)
    a.surf_range = np.ascontiguousarray(np.asarray(d["surf_range"])[:, :4], np.int32)
    a.vertices = np.ascontiguousarray(d["vertices"], np.float32)
    a.triangles = np.ascontiguousarray(d["triangles"], np.uint32)
    a.materials = np.ascontiguousarray(d["materials"], np.float32)
    cam = np.zeros(14, np.float32)
    cam[:13] = np.asarray(d["camera"], np.float32)[:13]
    a.camera = cam
    a.sun = None if d.get("sun") is None else np.ascontiguousarray(d["sun"], np.float32)
    return ora.OracleScene(a)


def product_from_dict(ptx, ctx, d):
    return ptx.Scene.from_arrays(ctx, d["model_xform"], d["model_surf"], d["surf_range"], d["vertices"], d["triangles"],
                                 d["materials"], d["camera"], d.get("sun"))


@pytest.fixture(scope="session")
def gold_jack():
    g = dict(np.load(os.path.join(GOLD, "jack_scene.npz")))
    g.update(np.load(os.path.join(GOLD, "jack_vectors.npz")))
    return g


@pytest.fixture(scope="session")
def jack_arrays(ora):
    return ora.load_gltf(JACK)


@pytest.fixture(scope="session")
def jack_oracle(ora, jack_arrays):
    return ora.OracleScene(jack_arrays)


def sha_u8(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def kd_stream_preorder(kd):
    """Canonical uint32 stream (same definition as oracle/make_golden.py:kd_stream) from the oracle's pre-order dump."""
    out = []
    sb = np.ascontiguousarray(kd["split"], np.float32).view(np.uint32)
    for i in range(len(kd["type"])):
        if kd["type"][i] == 1:
            f, c = int(kd["first"][i]), int(kd["count"][i])
            out.append(np.concatenate([[1, c], kd["refs"][f:f + c]]).astype(np.uint32))
        else:
            out.append(np.array([0, kd["axis"][i], sb[i], kd["left"][i] >= 0, kd["right"][i] >= 0], np.uint32))
    return np.concatenate(out)


def kd_stream_packed(nodes, refs, root, tri_base):
    """The same stream from the product's packed breadth-first nodes (iterative pre-order walk)."""
    out = []
    stack = [int(root)]
    while stack:
        i = stack.pop()
        w0, w1 = int(nodes[i, 0]), int(nodes[i, 1])
        kind = w1 & 3
        if kind == 3:
            c = w1 >> 2
            out.append(np.concatenate([[1, c], refs[w0:w0 + c].astype(np.int64) - tri_base]).astype(np.uint32))
            continue
        hl, hr, first = bool(w1 & 4), bool(w1 & 8), w1 >> 4
        out.append(np.array([0, kind, w0, hl, hr], np.uint32))
        if hr:
            stack.append(first + (1 if hl else 0))
        if hl:
            stack.append(first)
    return np.concatenate(out)
