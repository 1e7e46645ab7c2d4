import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(ROOT, "scenes", "cornell-box", "cornell.gltf")


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
