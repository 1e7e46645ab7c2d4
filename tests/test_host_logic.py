"""Product host logic on CPU (no GPU, no compute calls): the C-ABI library loads and exports every
symbol include/ptx.h declares; the product's own glTF loader / AABB / SAH KD builder / flattener
reproduce the UNMODIFIED reference's scene (tests/golden/cornell_scene.npz); error behaviour."""
import ctypes as C
import io
import os

import numpy as np
import pytest

from conftest import CORNELL, ROOT


def test_library_exports_every_declared_symbol(ptx):
    L = ptx.lib()
    names = ptx.declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ptx.h but not exported"
    assert b"gfx950" in L.ptx_version()


@pytest.fixture(scope="module")
def host_scene(ptx):
    return ptx.Scene.load_gltf(None, CORNELL)   # ctx=None: host-only scene


def test_loader_matches_reference(ptx, host_scene, gold_scene):
    s, g = host_scene, gold_scene
    assert s.array(ptx.ARR_MODEL_NAMES) == bytes(g["model_names"]).decode().split()
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_XFORM), g["model_xform"])
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_SURF), g["model_surf"])
    np.testing.assert_array_equal(s.array(ptx.ARR_SURF_RANGE)[:, :4], g["surf_range"][:, :4])
    np.testing.assert_array_equal(s.array(ptx.ARR_VERTICES), g["vertices"])
    np.testing.assert_array_equal(s.array(ptx.ARR_TRIANGLES), g["triangles"])
    np.testing.assert_array_equal(s.array(ptx.ARR_MATERIALS), g["materials"])
    np.testing.assert_array_equal(s.array(ptx.ARR_CAMERA), g["camera"])
    assert s.array(ptx.ARR_SUN).size == 0
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_AABB), g["model_aabb"])
    np.testing.assert_array_equal(s.array(ptx.ARR_MESH_AABB), g["mesh_aabb"])


def unpack_tree(nodes, refs, root):
    """Packed breadth-first device nodes -> pre-order lists, the order the golden dump uses."""
    out = []
    def walk(i):
        w0, w1 = int(nodes[i, 0]), int(nodes[i, 1])
        kind = w1 & 3
        if kind == 3:
            cnt = w1 >> 2
            out.append(("leaf", tuple(int(r) for r in refs[w0:w0 + cnt])))
            return
        hl, hr, first = bool(w1 & 4), bool(w1 & 8), w1 >> 4
        out.append(("branch", kind, w0, hl, hr))
        if hl:
            walk(first)
        if hr:
            walk(first + (1 if hl else 0))
    import sys
    sys.setrecursionlimit(10000)
    walk(root)
    return out


def gold_tree(g, s):
    k0, nk, r0, nr = (int(v) for v in g["surf_range"][s, 4:8])
    t0 = int(g["surf_range"][s, 2])
    out = []
    for i in range(k0, k0 + nk):   # the dump is already pre-order
        if g["kd_type"][i] == 1:
            f, c = int(g["kd_first"][i]), int(g["kd_count"][i])
            out.append(("leaf", tuple(int(r) + t0 for r in g["kd_refs"][f:f + c])))
        else:
            out.append(("branch", int(g["kd_axis"][i]), int(g["kd_split"][i:i + 1].view(np.uint32)[0]),
                        bool(g["kd_left"][i] >= 0), bool(g["kd_right"][i] >= 0)))
    return out


def test_kd_trees_have_the_reference_topology(ptx, host_scene, gold_scene):
    nodes = host_scene.array(ptx.ARR_KD_NODES)
    refs = host_scene.array(ptx.ARR_KD_REFS)
    rng = host_scene.array(ptx.ARR_SURF_RANGE)
    info = host_scene.info()
    assert info["n_kd_nodes"] == len(gold_scene["kd_type"]) and info["n_kd_refs"] == len(gold_scene["kd_refs"])
    assert info["kd_max_depth"] == gold_scene["kd_depth"].max()
    for s in range(info["n_surfaces"]):
        assert rng[s, 5] == gold_scene["surf_range"][s, 5] and rng[s, 7] == gold_scene["surf_range"][s, 7]
        assert unpack_tree(nodes, refs, int(rng[s, 4])) == gold_tree(gold_scene, s)
    assert info["lds_resident"] == 1 and info["geometry_bytes"] <= 160 * 1024


def test_from_arrays_equals_gltf_load(ptx, host_scene, gold_scene):
    g = gold_scene
    s2 = ptx.Scene.from_arrays(None, g["model_xform"], g["model_surf"], g["surf_range"], g["vertices"], g["triangles"],
                               g["materials"], g["camera"])
    for a in (ptx.ARR_KD_NODES, ptx.ARR_KD_REFS, ptx.ARR_MESH_AABB, ptx.ARR_MODEL_AABB, ptx.ARR_SURF_RANGE, ptx.ARR_CAMERA):
        np.testing.assert_array_equal(s2.array(a), host_scene.array(a))


def test_error_behaviour(ptx, host_scene, tmp_path):
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.load_gltf(None, str(tmp_path / "missing.gltf"))
    assert e.value.code == ptx.ERR_IO
    bad = tmp_path / "bad.gltf"
    bad.write_text("{ not json")
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.load_gltf(None, str(bad))
    assert e.value.code == ptx.ERR_PARSE
    # renderer.cpp:73-74: "Scene does not contain camera #i."
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.load_gltf(None, CORNELL, camera_index=3)
    assert e.value.code == ptx.ERR_NO_CAMERA and "camera #3" in str(e.value)
    # GPU work on a host-only scene must fail loudly, never fall back
    with pytest.raises(ptx.PtxError) as e:
        host_scene.render(8, 8, 1, 1)
    assert e.value.code == ptx.ERR_NO_DEVICE
    with pytest.raises(ptx.PtxError) as e:
        host_scene.intersect(np.zeros((1, 3), np.float32), np.array([[0, 0, 1]], np.float32))
    assert e.value.code == ptx.ERR_NO_DEVICE


def test_png_encode_roundtrip(ptx):
    from PIL import Image
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    png = ptx.encode_png(img)
    back = np.array(Image.open(io.BytesIO(png)))
    np.testing.assert_array_equal(back, img)


def test_reference_png_fixture_decodes(gold_vec):
    # the deterministic PNG written by the reference's own renderer::render is a valid 64x64 RGBA image
    from PIL import Image
    from conftest import GOLD
    im = np.array(Image.open(os.path.join(GOLD, "cornell_ref_64x64_16spp_4b.png")))
    assert im.shape == (64, 64, 4) and (im[..., 3] == 255).all()


def test_cpp_host_cli_fails_loudly_without_gpu(tmp_path):
    """The C++ host over the C ABI (host/ptx_renderer.hpp) has no CPU path: on a GPU-less box it must exit non-zero."""
    import subprocess
    import torch
    from conftest import ROOT
    cli = os.path.join(ROOT, "distributed-path-tracer_amd", "ptx_render_cli")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu tests")
    r = subprocess.run([cli, CORNELL, str(tmp_path / "o.png"), "16", "16", "1", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr
    assert not (tmp_path / "o.png").exists()


def test_procedural_scene_kd_matches_oracle_builder(ptx, ora):
    """A generated mesh (not the Cornell fixture): the product's SAH builder and the oracle's give the same trees."""
    import importlib
    from conftest import oracle_from_dict, product_from_dict
    proc = importlib.import_module("distributed-path-tracer_amd.procedural")
    d = proc.plaza_scene(level=3)
    assert len(d["triangles"]) == 2 + 1280 + 320
    s = product_from_dict(ptx, None, d)
    o = oracle_from_dict(ora, d)
    nodes, refs, rng = s.array(ptx.ARR_KD_NODES), s.array(ptx.ARR_KD_REFS), s.array(ptx.ARR_SURF_RANGE)
    mb, sb = o.boxes()
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_AABB), mb)
    np.testing.assert_array_equal(s.array(ptx.ARR_MESH_AABB), sb)
    for k in range(3):
        kd = o.kd(k)
        t0 = int(rng[k, 2])
        want = []
        for i in range(len(kd["type"])):
            if kd["type"][i] == 1:
                f, c = int(kd["first"][i]), int(kd["count"][i])
                want.append(("leaf", tuple(int(r) + t0 for r in kd["refs"][f:f + c])))
            else:
                want.append(("branch", int(kd["axis"][i]), int(kd["split"][i:i + 1].view(np.uint32)[0]),
                             bool(kd["left"][i] >= 0), bool(kd["right"][i] >= 0)))
        assert unpack_tree(nodes, refs, int(rng[k, 4])) == want
    info = s.info()
    assert info["has_sun"] == 1 and info["n_models"] == 3


def test_kd_build_is_independent_of_the_thread_count(ptx, monkeypatch):
    """The host SAH builder runs one surface per thread and, for a large mesh, the top levels of its recursion on further threads
    (PTX_BUILD_THREADS): the emitted node / reference / record arrays must not depend on it. A many-surface scene and a scene
    whose one big mesh (20 480 triangles, above the 16 384-triangle task threshold) is split into subtree tasks."""
    import importlib
    from conftest import product_from_dict
    proc = importlib.import_module("distributed-path-tracer_amd.procedural")
    cor = ptx.Scene.load_gltf(None, CORNELL)
    c = {k: cor.array(getattr(ptx, "ARR_" + k.upper())) for k in ("model_xform", "model_surf", "surf_range", "vertices", "triangles", "materials", "camera")}
    for d in (proc.atrium_scene(3), proc.cornell_with_mesh(c, level=5)):
        got = []
        for threads in ("1", "3", "16"):
            monkeypatch.setenv("PTX_BUILD_THREADS", threads)
            s = product_from_dict(ptx, None, d)
            got.append([np.asarray(s.array(a)).tobytes() for a in (ptx.ARR_KD_NODES, ptx.ARR_KD_REFS, ptx.ARR_SURF_RANGE, ptx.ARR_MESH_AABB)] + [s.info()["kd_max_depth"]])
        assert got[0] == got[1] == got[2]


def test_from_arrays_rejects_bad_input(ptx):
    import importlib
    proc = importlib.import_module("distributed-path-tracer_amd.procedural")
    d = proc.plaza_scene(level=1)
    bad = dict(d); bad["triangles"] = d["triangles"].copy(); bad["triangles"][5, 1] = 10 ** 6
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.from_arrays(None, bad["model_xform"], bad["model_surf"], bad["surf_range"], bad["vertices"], bad["triangles"],
                              bad["materials"], bad["camera"], bad["sun"])
    assert e.value.code == ptx.ERR_INVALID
    bad = dict(d); bad["model_surf"] = np.array([[1, 1], [0, 1], [2, 1]], np.int32)      # ranges not in model order
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.from_arrays(None, bad["model_xform"], bad["model_surf"], bad["surf_range"], bad["vertices"], bad["triangles"],
                              bad["materials"], bad["camera"], bad["sun"])
    assert e.value.code == ptx.ERR_INVALID


# ---------------------------------------------------------------------------- textured asset: loader, PNG reader, trees
def test_jack_product_loader_matches_reference(ptx, gold_jack, jack_arrays):
    from conftest import JACK, kd_stream_packed, sha_u8
    s = ptx.Scene.load_gltf(None, JACK)
    g = gold_jack
    assert s.array(ptx.ARR_MODEL_NAMES) == bytes(g["model_names"]).decode().split()
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_XFORM), g["model_xform"])
    np.testing.assert_array_equal(s.array(ptx.ARR_MATERIALS), g["materials"])
    np.testing.assert_array_equal(s.array(ptx.ARR_CAMERA), g["camera"])
    np.testing.assert_array_equal(s.array(ptx.ARR_SUN), g["sun"])
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_AABB), g["model_aabb"])
    np.testing.assert_array_equal(s.array(ptx.ARR_MESH_AABB), g["mesh_aabb"])
    np.testing.assert_array_equal(sha_u8(s.array(ptx.ARR_VERTICES)), g["sha_vertices"])
    np.testing.assert_array_equal(sha_u8(s.array(ptx.ARR_TRIANGLES)), g["sha_triangles"])
    rng, nodes, refs = s.array(ptx.ARR_SURF_RANGE), s.array(ptx.ARR_KD_NODES), s.array(ptx.ARR_KD_REFS)
    for k in range(len(rng)):
        assert rng[k, 5] == g["surf_range"][k, 5] and rng[k, 7] == g["surf_range"][k, 7]
        np.testing.assert_array_equal(sha_u8(kd_stream_packed(nodes, refs, rng[k, 4], int(rng[k, 2]))), g["sha_kd"][k])
    info = s.info()
    assert info["n_triangles"] == 58740 and info["has_sun"] == 1 and info["lds_resident"] == 2 and info["n_textures"] == 17
    # texture slots and decoded texels against the oracle's loader (PIL decode): same slots, same sRGB flags, same bytes
    np.testing.assert_array_equal(s.array(ptx.ARR_SURF_TEX), jack_arrays.surf_tex)
    np.testing.assert_array_equal((s.array(ptx.ARR_SURF_TEX) >= 0).astype(np.uint8), g["material_tex"])
    tex, texels = s.array(ptx.ARR_TEXTURES), s.array(ptx.ARR_TEXELS)
    for i, im in enumerate(jack_arrays.images):
        w, h, cs, off = (int(v) for v in tex[i])
        assert (h, w, cs & 255) == im.shape and bool(cs >> 8) == jack_arrays.image_srgb[i]
        np.testing.assert_array_equal(texels[off:off + im.size].reshape(im.shape), im)


def test_png_reader_against_pil(ptx, tmp_path):
    """csrc/png_read.cpp vs PIL on every PNG flavour stb_image would accept: grey, grey+alpha, RGB, RGBA, palette (with and
    without tRNS), 16-bit, sub-byte grey; all five filter types occur in the 'optimize' encodings."""
    import json
    from PIL import Image
    rng = np.random.default_rng(5)
    imgs = {}
    base = (rng.random((37, 53, 4)) * 255).astype(np.uint8)
    base[:, :, 0] = np.linspace(0, 255, 53).astype(np.uint8)[None, :]          # smooth ramps make Sub/Up/Paeth filters win
    base[:, :, 1] = np.linspace(0, 255, 37).astype(np.uint8)[:, None]
    imgs["rgba"] = Image.fromarray(base, "RGBA")
    imgs["rgb"] = Image.fromarray(base[..., :3].copy(), "RGB")
    imgs["l"] = Image.fromarray(base[..., 1].copy(), "L")
    imgs["la"] = Image.fromarray(base[..., [1, 3]].copy(), "LA")
    imgs["p"] = Image.fromarray(base[..., :3].copy(), "RGB").quantize(31)
    imgs["l16"] = Image.fromarray((base[..., 0].astype(np.uint16) * 257), "I;16")
    imgs["l1"] = Image.fromarray((base[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1")
    gl = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0, 1]}], "cameras": [{"name": "c", "type": "perspective", "perspective": {"yfov": 0.7}}],
          "nodes": [{"camera": 0, "name": "c"}, {"mesh": 0, "name": "m"}], "buffers": [{"uri": "b.bin", "byteLength": 132}],
          "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                          {"buffer": 0, "byteOffset": 60, "byteLength": 36}, {"buffer": 0, "byteOffset": 96, "byteLength": 6}],
          "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                        {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 3, "componentType": 5123, "count": 3, "type": "SCALAR"}],
          "images": [], "textures": [], "materials": [], "meshes": [{"primitives": []}]}
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    blob = pos.tobytes() + np.zeros((3, 2), np.float32).tobytes() + np.tile(np.float32([0, 0, 1]), 3).tobytes() + np.uint16([0, 1, 2]).tobytes() + b"\0" * 30
    (tmp_path / "b.bin").write_bytes(blob[:132])
    names = sorted(imgs)
    for n in names:
        imgs[n].save(tmp_path / f"{n}.png", optimize=True)
    # Adam7-interlaced files (Pillow cannot write them): built by hand from the same pixels, filter type 0, for 8-bit RGBA, 8-bit grey
    # with a width that leaves some passes empty, and 4-bit grey (sub-byte rows per pass)
    import struct
    import zlib

    def adam7_png(path, arr, depth, ctype):
        h, w = arr.shape[:2]
        ch = 1 if arr.ndim == 2 else arr.shape[2]
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = arr[y0::dy, x0::dx]
            if sub.size == 0:
                continue
            for row in sub.reshape(sub.shape[0], -1):
                if depth == 8:
                    raw += b"\0" + row.astype(np.uint8).tobytes()
                else:                                           # 4-bit: two samples per byte, MSB first
                    v = list(row.astype(np.uint8)) + [0]
                    raw += b"\0" + bytes((v[k] << 4) | v[k + 1] for k in range(0, len(row), 2))

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) +
                         chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    adam7_png(tmp_path / "i_rgba.png", base, 8, 6)
    adam7_png(tmp_path / "i_l_narrow.png", base[:5, :3, 1].copy(), 8, 0)
    adam7_png(tmp_path / "i_l4.png", (base[:, :, 1] >> 4).copy(), 4, 0)
    names += ["i_rgba", "i_l_narrow", "i_l4"]
    for i, n in enumerate(names):
        gl["images"].append({"uri": f"{n}.png"}); gl["textures"].append({"source": i})
        gl["materials"].append({"name": n, "pbrMetallicRoughness": {"baseColorTexture": {"index": i}}})
        gl["meshes"][0]["primitives"].append({"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "NORMAL": 2}, "indices": 3, "material": i})
    (tmp_path / "t.gltf").write_text(json.dumps(gl))
    s = ptx.Scene.load_gltf(None, str(tmp_path / "t.gltf"))
    tex, texels = s.array(ptx.ARR_TEXTURES), s.array(ptx.ARR_TEXELS)
    assert len(tex) == len(names)
    from oracle import pt_oracle as ora
    for i, n in enumerate(names):
        ref = ora._decode_image(str(tmp_path / f"{n}.png"))
        w, h, cs, off = (int(v) for v in tex[i])
        assert (h, w, cs & 255) == ref.shape, n
        np.testing.assert_array_equal(texels[off:off + ref.size].reshape(ref.shape), ref, err_msg=n)
    # a JPEG (or anything that is not a PNG) is refused, not mis-rendered
    (tmp_path / "x.png").write_bytes(b"\xff\xd8\xff\xe0 not a png")
    gl["images"][0]["uri"] = "x.png"
    (tmp_path / "u.gltf").write_text(json.dumps(gl))
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.load_gltf(None, str(tmp_path / "u.gltf"))
    assert e.value.code == ptx.ERR_PARSE


# ---------------------------------------------------------------------------- worker event front-end
def _event(tmp_path, work, samples=5, bounces=3, X=48, Y=32):
    import json, shutil
    root = tmp_path / "scene-root"
    root.mkdir()
    shutil.copyfile(CORNELL, root / "scene.gltf")                       # the worker downloads <scene_root>scene.gltf (worker.cpp:108-112)
    shutil.copyfile(os.path.join(os.path.dirname(CORNELL), "cornell.bin"), root / "cornell.bin")
    ev = {"scene_info": {"work": work, "total_size": 0.05}, "scene_bucket": "distributed-path-tracer", "scene_root": "scenes/cornell-box/",
          "worker_id": "1", "sqs_queue_arn": "", "sns_topic_arn": "", "num_workers": 1, "samples": samples, "bounces": bounces, "X": X, "Y": Y}
    p = tmp_path / "event.json"
    p.write_text(json.dumps(ev, indent=4))
    return str(p), str(root)


def test_worker_event_and_primitive_filter(ptx, ora, tmp_path):
    """models::worker_info JSON + scene_info.work filter (host logic; the reference's HOST needs the AWS SDK and cannot be
    built here, so this row is checked against the oracle's restatement of src/scene/load_gltf.cpp:93-99 — parity unpinned)."""
    work = {"Cube.003": [0, 2], "Sphere": [0], "No.Such.Mesh": [0]}
    ev, root = _event(tmp_path, work)
    s, cfg, info = ptx.Scene.load_event(None, ev, root)
    assert (cfg.W, cfg.H, cfg.spp, cfg.bounces) == (48, 32, 5, 3) and tuple(cfg.env) == (1.0, 1.0, 1.0)
    assert info["worker_id"] == "1" and info["num_workers"] == 1 and info["n_work_meshes"] == 3 and info["scene_root"] == "scenes/cornell-box/"
    a = ora.load_gltf(os.path.join(root, "scene.gltf"), work=work)
    assert a.model_surf.tolist() == [[0, 0], [0, 0], [0, 2], [2, 0], [2, 1]]   # unlisted meshes keep an empty model
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_SURF), a.model_surf)
    np.testing.assert_array_equal(s.array(ptx.ARR_SURF_RANGE)[:, :4], a.surf_range)
    np.testing.assert_array_equal(s.array(ptx.ARR_VERTICES), a.vertices)
    np.testing.assert_array_equal(s.array(ptx.ARR_MATERIALS), a.materials)
    o = ora.OracleScene(a)
    mb, sb = o.boxes()
    np.testing.assert_array_equal(s.array(ptx.ARR_MODEL_AABB), mb)   # empty models: the cleared box (min > max), never entered
    # the same filter through ptx_load_opts
    s2 = ptx.Scene.load_gltf(None, os.path.join(root, "scene.gltf"), work=work)
    np.testing.assert_array_equal(s2.array(ptx.ARR_KD_NODES), s.array(ptx.ARR_KD_NODES))
    # no filter = core::renderer behaviour
    assert ptx.Scene.load_gltf(None, os.path.join(root, "scene.gltf")).info()["n_surfaces"] == 7
    bad = tmp_path / "bad.json"
    bad.write_text('{"samples": 1}')
    with pytest.raises(ptx.PtxError) as e:
        ptx.Scene.load_event(None, str(bad), root)
    assert e.value.code == ptx.ERR_PARSE


# ---------------------------------------------------------------------------- malformed inputs: an error code, never a crash
def _expect_error_or_load(ptx, path):
    try:
        ptx.Scene.load_gltf(None, path)
        return "loaded"
    except ptx.PtxError as e:
        assert e.code in (ptx.ERR_IO, ptx.ERR_PARSE, ptx.ERR_INVALID, ptx.ERR_NO_CAMERA, ptx.ERR_UNSUPPORTED), e
        return "error"


def test_malformed_gltf_is_an_error_not_a_crash(ptx, tmp_path):
    """Out-of-range indices, missing / short buffers, wrong accessor types, a node cycle (the reference would recurse forever),
    and the document truncated at 60 random places."""
    import copy
    import json
    import random
    import shutil
    src = os.path.dirname(CORNELL)
    base = json.load(open(CORNELL))
    binname = base["buffers"][0]["uri"]
    binb = open(os.path.join(src, binname), "rb").read()

    def attempt(doc=None, raw=None, bin_bytes=None):
        d = tmp_path / "case"
        shutil.rmtree(d, ignore_errors=True)
        d.mkdir()
        if bin_bytes is None:
            (d / binname).write_bytes(binb)
        elif bin_bytes is not False:
            (d / binname).write_bytes(bin_bytes)
        (d / "s.gltf").write_text(raw if raw is not None else json.dumps(doc))
        return _expect_error_or_load(ptx, str(d / "s.gltf"))

    assert attempt(base) == "loaded"
    assert attempt(base, bin_bytes=False) == "error"
    assert attempt(base, bin_bytes=binb[:len(binb) // 2]) == "error"
    muts = [lambda d: d["bufferViews"][0].__setitem__("byteOffset", 10 ** 9),
            lambda d: d["accessors"][0].__setitem__("count", 10 ** 8),
            # values that would wrap through size_t in `offset + (count - 1) * stride + element`: negative, or near 2^63
            lambda d: d["bufferViews"][0].__setitem__("byteOffset", -16),
            lambda d: d["accessors"][0].__setitem__("byteOffset", -4),
            lambda d: d["accessors"][0].__setitem__("count", -1),
            lambda d: d["accessors"][0].__setitem__("count", 2 ** 62),
            lambda d: d["bufferViews"][0].__setitem__("byteStride", 2 ** 62),
            lambda d: d["bufferViews"][0].__setitem__("byteStride", -8),
            lambda d: d["bufferViews"][0].__setitem__("byteOffset", 2 ** 63 - 1),
            lambda d: d["bufferViews"][0].__setitem__("buffer", -1),
            lambda d: d["accessors"][0].__setitem__("bufferView", 999),
            lambda d: d["accessors"][0].__setitem__("componentType", 1234),
            lambda d: d["accessors"][0].__setitem__("type", "MAT4"),
            lambda d: d["meshes"][0]["primitives"][0]["attributes"].pop("POSITION"),
            lambda d: d["meshes"][0]["primitives"][0]["attributes"].__setitem__("POSITION", 9999),
            lambda d: d["meshes"][0]["primitives"][0].__setitem__("indices", 9999),
            lambda d: d["meshes"][0]["primitives"][0].__setitem__("material", 9999),
            lambda d: d["nodes"][0].__setitem__("mesh", 9999),
            lambda d: d["nodes"][0].__setitem__("children", [9999]),
            lambda d: d["nodes"][0].__setitem__("children", [0]),          # cycle
            lambda d: d["scenes"][0].__setitem__("nodes", [9999]),
            lambda d: d.pop("accessors"),
            lambda d: d.__setitem__("meshes", "x")]
    for m in muts:
        d = copy.deepcopy(base)
        m(d)
        assert attempt(d) == "error", m
    ia = base["accessors"][base["meshes"][0]["primitives"][0]["indices"]]
    bv = base["bufferViews"][ia["bufferView"]]
    off = bv.get("byteOffset", 0) + ia.get("byteOffset", 0)
    bb = bytearray(binb)
    bb[off:off + 2] = (65535).to_bytes(2, "little")                          # an index beyond the vertex count
    assert attempt(base, bin_bytes=bytes(bb)) == "error"
    raw = json.dumps(base)
    rnd = random.Random(1)
    for _ in range(60):
        assert attempt(raw=raw[:rnd.randrange(1, len(raw))]) == "error"


def test_malformed_png_is_an_error_not_a_crash(ptx, tmp_path):
    """Texture files that are truncated, bit-flipped, or lie about their size / type."""
    import json
    import random
    import struct
    import zlib
    from PIL import Image
    gl = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0, 1]}], "cameras": [{"name": "c", "type": "perspective", "perspective": {"yfov": 0.7}}],
          "nodes": [{"camera": 0, "name": "c"}, {"mesh": 0, "name": "m"}], "buffers": [{"uri": "b.bin", "byteLength": 132}],
          "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                          {"buffer": 0, "byteOffset": 60, "byteLength": 36}, {"buffer": 0, "byteOffset": 96, "byteLength": 6}],
          "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                        {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 3, "componentType": 5123, "count": 3, "type": "SCALAR"}],
          "images": [{"uri": "t.png"}], "textures": [{"source": 0}], "materials": [{"name": "m", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
          "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "NORMAL": 2}, "indices": 3, "material": 0}]}]}
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    blob = pos.tobytes() + np.zeros((3, 2), np.float32).tobytes() + np.tile(np.float32([0, 0, 1]), 3).tobytes() + np.uint16([0, 1, 2]).tobytes() + b"\0" * 30
    (tmp_path / "b.bin").write_bytes(blob[:132])
    (tmp_path / "s.gltf").write_text(json.dumps(gl))
    img = (np.random.default_rng(3).random((23, 31, 4)) * 255).astype(np.uint8)
    Image.fromarray(img, "RGBA").save(tmp_path / "good.png")
    good = (tmp_path / "good.png").read_bytes()

    def attempt(data):
        (tmp_path / "t.png").write_bytes(data)
        return _expect_error_or_load(ptx, str(tmp_path / "s.gltf"))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    def png(w, h, depth, ctype, interlace, idat):
        return good[:8] + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")

    raw = b"".join(b"\0" + bytes(31 * 4) for _ in range(23))
    assert attempt(good) == "loaded"
    assert attempt(png(31, 23, 8, 6, 0, zlib.compress(raw))) == "loaded"
    for bad in (b"", good[:8], png(2 ** 31 - 1, 2 ** 31 - 1, 8, 6, 0, zlib.compress(raw)), png(0, 0, 8, 6, 0, zlib.compress(raw)),
                png(4096, 4096, 8, 6, 0, zlib.compress(raw)), png(31, 23, 8, 7, 0, zlib.compress(raw)), png(31, 23, 3, 6, 0, zlib.compress(raw)),
                png(65535, 65535, 16, 6, 0, zlib.compress(raw)), png(65535, 65535, 16, 6, 1, zlib.compress(raw)),   # 32 GiB by the header, 3 KB of data: refused before any allocation
                png(20000, 20000, 8, 6, 0, zlib.compress(raw)),
                png(31, 23, 8, 6, 1, zlib.compress(raw)), png(31, 23, 8, 6, 0, zlib.compress(raw.replace(b"\0" + bytes(124), b"\x09" + bytes(124)))),
                png(31, 23, 8, 6, 0, zlib.compress(raw[:100])), png(31, 23, 8, 6, 0, b"\x12\x34" * 50),
                png(31, 23, 8, 3, 0, zlib.compress(b"".join(b"\0" + bytes([200] * 31) for _ in range(23)))),
                good[:8] + struct.pack(">I", 0xFFFFFFF0) + b"IHDR" + good[16:]):
        assert attempt(bad) == "error"
    rnd = random.Random(2)
    for _ in range(60):
        assert attempt(good[:rnd.randrange(1, len(good))]) == "error"
    for _ in range(120):                       # a flipped byte either fails a check or still decodes: both are fine, a crash is not
        b = bytearray(good)
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(8, len(b))] = rnd.randrange(256)
        attempt(bytes(b))


def test_malformed_worker_event_is_an_error(ptx, tmp_path):
    import json
    import random
    work = {"Cube.003": [0, 1, 2]}
    ev_path, root = _event(tmp_path, work, samples=2, bounces=2, X=8, Y=8)
    raw = open(ev_path).read()
    rnd = random.Random(3)
    bad = tmp_path / "bad_event.json"
    for _ in range(40):
        bad.write_text(raw[:rnd.randrange(1, len(raw) - 1)])
        with pytest.raises(ptx.PtxError) as e:
            ptx.Scene.load_event(None, str(bad), root)
        assert e.value.code in (ptx.ERR_PARSE, ptx.ERR_INVALID)
    doc = json.loads(raw)
    for key, val in (("samples", -1), ("bounces", 0), ("X", 0), ("Y", -5)):
        d = json.loads(raw)
        tgt = d["scene_info"] if key in d.get("scene_info", {}) else d
        tgt[key] = val
        bad.write_text(json.dumps(d))
        with pytest.raises(ptx.PtxError):
            ptx.Scene.load_event(None, str(bad), root)
    assert doc


# ---------------------------------------------------------------------------- JPEG textures (image::image::load -> stb_image v2.30)
def _texture_scene(ptx, tmp_path, image_path):
    """Host-only scene whose one material samples `image_path` as base colour: the decoded texels come back through PTX_ARR_TEXELS."""
    import json
    import shutil
    d = tmp_path / "jpeg_scene"
    d.mkdir(exist_ok=True)
    shutil.copyfile(image_path, d / "tex.bin")              # stb_image (and this reader) go by content, not by extension
    gl = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0, 1]}], "cameras": [{"name": "c", "type": "perspective", "perspective": {"yfov": 0.7}}],
          "nodes": [{"camera": 0, "name": "c"}, {"mesh": 0, "name": "m"}], "buffers": [{"uri": "b.bin", "byteLength": 132}],
          "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                          {"buffer": 0, "byteOffset": 60, "byteLength": 36}, {"buffer": 0, "byteOffset": 96, "byteLength": 6}],
          "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"},
                        {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 3, "componentType": 5123, "count": 3, "type": "SCALAR"}],
          "images": [{"uri": "tex.bin"}], "textures": [{"source": 0}], "materials": [{"name": "m", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
          "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "NORMAL": 2}, "indices": 3, "material": 0}]}]}
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    blob = pos.tobytes() + np.zeros((3, 2), np.float32).tobytes() + np.tile(np.float32([0, 0, 1]), 3).tobytes() + np.uint16([0, 1, 2]).tobytes() + b"\0" * 30
    (d / "b.bin").write_bytes(blob[:132])
    (d / "t.gltf").write_text(json.dumps(gl))
    s = ptx.Scene.load_gltf(None, str(d / "t.gltf"))
    tex, texels = s.array(ptx.ARR_TEXTURES), s.array(ptx.ARR_TEXELS)
    w, h, cs, off = (int(v) for v in tex[0])
    return texels[off:off + w * h * (cs & 255)].reshape(h, w, cs & 255).copy()


@pytest.fixture(scope="module")
def gold_jpeg():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "jpeg_vectors.npz")))


def _jpeg_files():
    import glob
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg", "*.jpg")))


@pytest.mark.parametrize("tag", _jpeg_files())
def test_jpeg_decode_matches_reference_texel_for_texel(ptx, ora, tmp_path, gold_jpeg, tag):
    """csrc/jpeg_read.cpp against the pixels the compiled reference (stb_image v2.30 via image::image::load, image.cpp:23-54) decoded
    from the same file: every byte equal — integer IDCT, chroma upsampling and YCbCr -> RGB restated. Files: two of the reference's own
    Sponza textures and synthetic ones covering 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, progressive, restart intervals, grey, 1x1, odd sizes,
    quality 10 and 100. Then image_texture::sample (bilinear, linear and sRGB) on those texels through the oracle's sampler (itself
    pinned on PNGs) against the reference's lookups: bit-exact."""
    from conftest import sha_u8
    g = gold_jpeg
    px = _texture_scene(ptx, tmp_path, os.path.join(ROOT, "tests", "golden", "jpeg", tag + ".jpg"))
    assert px.shape == tuple(g[tag + "_shape"])
    np.testing.assert_array_equal(sha_u8(px), g[tag + "_sha"])
    if tag + "_pixels" in g:
        np.testing.assert_array_equal(px, g[tag + "_pixels"])
    # bilinear lookups: a one-surface oracle scene whose albedo texture is the product's decode
    for srgb, key in ((False, "sample_linear"), (True, "sample_srgb")):
        a = ora.SceneArrays()
        a.model_xform = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1]], np.float32)
        a.model_surf = np.array([[0, 1]], np.int32)
        a.surf_range = np.array([[0, 3, 0, 1]], np.int32)
        a.vertices = np.zeros((3, 11), np.float32); a.vertices[1, 0] = 1; a.vertices[2, 1] = 1; a.vertices[:, 7] = 1
        a.triangles = np.array([[0, 1, 2]], np.uint32)
        a.materials = np.array([[1, 1, 1, 1, 1, 1, 1, 1, 1, 1.33, 0]], np.float32)
        cam = np.zeros(14, np.float32); cam[3] = cam[7] = cam[11] = 1; cam[12] = 0.7; cam[13] = 0.365
        a.camera = cam
        a.images, a.image_srgb, a.image_paths = [px], [srgb], [tag]
        a.surf_tex = np.array([[-1, 0, -1, -1, -1, -1, -1]], np.int32)
        out = ora.OracleScene(a).material_eval(0, g[tag + "_uv"])
        ref = g[f"{tag}_{key}"]
        n_ch = px.shape[2]
        exp = ref[:, :3] if n_ch >= 3 else np.stack([ref[:, 0], np.ones(len(ref), np.float32), np.ones(len(ref), np.float32)], 1)   # missing channels read 1 (image_texture.cpp:47-62)
        np.testing.assert_array_equal(out[:, 3:6].view(np.uint32), np.ascontiguousarray(exp, np.float32).view(np.uint32))


@pytest.mark.skipif(not os.path.isdir("/root/reference/path-tracer-core/scenes/sponza-new/textures"), reason="the reference tree (38 MB of textures) is not on this machine")
def test_all_63_sponza_jpegs_decode_like_the_reference(ptx, tmp_path, gold_jpeg):
    """Every JPEG of scenes/sponza-new/textures (63 baseline 4:4:4 files of 1024 x 1024): SHA-256 of the decoded pixels equals the
    digest of the compiled reference's decode (fixture). Runs only where the reference tree exists; the fixture travels."""
    from conftest import sha_u8
    d = "/root/reference/path-tracer-core/scenes/sponza-new/textures"
    names = [str(n) for n in gold_jpeg["sponza_names"]]
    assert len(names) == 63
    for n, ref in zip(names[::4], gold_jpeg["sponza_sha"][::4]):          # every fourth file: 16 MB of pixels, ~2 s
        np.testing.assert_array_equal(sha_u8(_texture_scene(ptx, tmp_path, os.path.join(d, n))), ref, err_msg=n)


def test_jpeg_refusals_and_malformed_files(ptx, tmp_path):
    """CMYK / 12-bit / arithmetic files are refused (PTX_ERR_UNSUPPORTED); truncated or corrupted files give an error or an image,
    never a crash."""
    import random
    src = open(os.path.join(ROOT, "tests", "golden", "jpeg", "s420.jpg"), "rb").read()
    from PIL import Image
    Image.new("CMYK", (16, 16), (10, 20, 30, 40)).save(tmp_path / "cmyk.jpg")
    with pytest.raises(ptx.PtxError) as e:
        _texture_scene(ptx, tmp_path, str(tmp_path / "cmyk.jpg"))
    assert e.value.code == ptx.ERR_UNSUPPORTED
    # a scan that names a quantisation table no DQT segment defined: refused (the reader used to decode against uninitialised memory)
    def segments(data):
        out, i = [], 2
        while i + 4 <= len(data) and data[i] == 0xFF and data[i + 1] != 0xDA:
            n = (data[i + 2] << 8) | data[i + 3]
            out.append((data[i + 1], i, i + 2 + n))
            i += 2 + n
        return out
    dqt = [(a, b) for m, a, b in segments(src) if m == 0xDB]
    assert dqt
    stripped = bytearray(src)
    for a, b in reversed(dqt):
        del stripped[a:b]
    (tmp_path / "nodqt.jpg").write_bytes(bytes(stripped))
    with pytest.raises(ptx.PtxError) as e:
        _texture_scene(ptx, tmp_path, str(tmp_path / "nodqt.jpg"))
    assert e.value.code == ptx.ERR_PARSE
    rnd = random.Random(2)
    for k in range(80):
        b = bytearray(src)
        if k % 2:
            b = b[:rnd.randrange(2, len(b))]
        else:
            for _ in range(rnd.randrange(1, 6)):
                b[rnd.randrange(2, len(b))] = rnd.randrange(256)
        (tmp_path / "m.jpg").write_bytes(bytes(b))
        try:
            px = _texture_scene(ptx, tmp_path, str(tmp_path / "m.jpg"))
            assert px.ndim == 3
        except ptx.PtxError as e:
            assert e.code in (ptx.ERR_PARSE, ptx.ERR_UNSUPPORTED, ptx.ERR_IO), e


# ---------------------------------------------------------------------------- Radiance .hdr images (image::image::load, HDR branch)
@pytest.mark.parametrize("tag", ["sky_rle", "sky_flat", "tiny"])
def test_hdr_decode_matches_reference(ptx, ora, tmp_path, tag):
    """csrc/hdr_read.cpp (and the oracle's numpy reader) against the floats stb_image's stbi_loadf produced in the compiled reference —
    run-length scanlines, a file that is not run-length encoded, a file too narrow for RLE: bit-exact floats (byte * 2^(e - 136))."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "hdr_vectors.npz")))
    path = os.path.join(ROOT, "tests", "golden", "hdr", tag + ".hdr")
    ref = g[tag + "_pixels"]
    np.testing.assert_array_equal(ora._decode_hdr(path).view(np.uint32), ref.view(np.uint32))
    s = ptx.Scene.load_gltf(None, CORNELL)
    s.set_environment(path, False)                            # host-only scene: decodes, keeps the floats
    tex = s.array(ptx.ARR_TEXTURES)[-1]
    w, h, cs, off = (int(v) for v in tex)
    assert (h, w, cs & 255) == ref.shape and cs & (1 << 16)
    fl = s.array(ptx.ARR_TEXELS_F32)[off:off + ref.size].reshape(ref.shape)
    np.testing.assert_array_equal(fl.view(np.uint32), ref.view(np.uint32))
    n_tex = s.info()["n_textures"]
    s.set_environment(os.path.join(ROOT, "tests", "golden", "jpeg", "s444.jpg"), True)   # replaced by an 8-bit image: the floats go away
    assert s.info()["n_textures"] == n_tex and len(s.array(ptx.ARR_TEXELS_F32)) == 0


def test_hdr_environment_lookup_bit_exact_in_the_oracle(cornell_oracle, ora):
    """renderer::trace's miss branch with a Radiance .hdr as renderer::environment: equirectangular_proj + image_texture::sample on float
    texels (+ pow(v, 2.2F) when loaded as sRGB) + environment_factor, against the compiled reference (384 directions incl. the six axes)."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "hdr_vectors.npz")))
    path = os.path.join(ROOT, "tests", "golden", "hdr", "sky_rle.hdr")
    try:
        for srgb in (0, 1):
            cornell_oracle.set_environment(path, srgb=bool(srgb))
            uv, rgba, _ = cornell_oracle.env_lookup(g[f"srgb{srgb}_env_in"], (0.5, 1.25, 2.0))
            np.testing.assert_array_equal(uv.view(np.uint32), g[f"srgb{srgb}_env_uv"].view(np.uint32))
            np.testing.assert_array_equal(rgba.view(np.uint32), g[f"srgb{srgb}_env_out"].view(np.uint32))
            d = g[f"srgb{srgb}_env_in"].astype(np.float32)
            ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
            _, _, col = cornell_oracle.env_lookup(d * (np.float32(1) / ln)[:, None], (0.5, 1.25, 2.0))      # trace() sees ray::get_dir()
            np.testing.assert_array_equal(col.view(np.uint32), g[f"srgb{srgb}_env_trace"][:, :3].view(np.uint32))
        assert g["srgb0_env_out"].max() > 100                     # the map really is high dynamic range
    finally:
        cornell_oracle.set_environment(None)


def test_malformed_hdr_is_an_error_or_an_image(ptx, tmp_path):
    """Truncated / corrupted Radiance files: an error code or an image, never a crash or an out-of-bounds read."""
    import random
    src = open(os.path.join(ROOT, "tests", "golden", "hdr", "sky_rle.hdr"), "rb").read()
    s = ptx.Scene.load_gltf(None, CORNELL)
    rnd = random.Random(4)
    cases = [src[:rnd.randrange(1, len(src))] for _ in range(40)]
    for _ in range(40):
        b = bytearray(src)
        for _ in range(rnd.randrange(1, 5)):
            b[rnd.randrange(11, len(b))] = rnd.randrange(256)
        cases.append(bytes(b))
    cases += [b"#?RADIANCE\n\n-Y 4 +X 4\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 4 +X 4\n" + b"\0" * 64,
              b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 99999999 +X 99999999\n", b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y -3 +X 4\n"]
    for c in cases:
        (tmp_path / "m.hdr").write_bytes(c)
        try:
            s.set_environment(str(tmp_path / "m.hdr"), False)
        except ptx.PtxError as e:
            assert e.code in (ptx.ERR_PARSE, ptx.ERR_UNSUPPORTED, ptx.ERR_IO), e
