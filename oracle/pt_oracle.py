"""TEST INFRASTRUCTURE — ctypes glue for the CPU oracle (oracle/libpt_oracle.so) plus a numpy
restatement of the reference's glTF scene loader.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Loader restated from (paths relative to /root/reference/path-tracer-core/path_tracer_lib/path_tracer/):
  core/renderer.cpp:61-99   load_gltf       (camera / sun picked by index, matched to nodes by NAME)
  core/renderer.cpp:101-174 process_node    (TRS only — a node `matrix` is ignored; entity named after
                                             its camera / light; roots kept in an unordered_map)
  core/renderer.cpp:177-263 get_mesh        (quirk Q1: TANGENT unpacked as count*3 floats of the xyzw stream)
  core/renderer.cpp:265-331 get_material    (factors; texture presence is recorded, lookups are a later row)
  scene/transform.cpp:14-31 transform::make, math/quat.cpp:95-113 quat::to_basis
  scene/entity.cpp:72-85    get_global_transform (parent * local)
All float arithmetic is done on np.float32 scalars in the reference's operation order.
"""
import ctypes as C
import json
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpt_oracle.so")

f32 = np.float32
_lib = None


def build(force=False):
    """Compile oracle/libpt_oracle.so (g++, a few seconds). Also builds oracle/_ref when the reference is present."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "pt_oracle.cpp")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/path-tracer-core/path_tracer_lib"):
        subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.ora_scene_create.restype = C.c_void_p
        _lib.ora_tan_half_fov.restype = C.c_float
    return _lib


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


class RenderCfg(C.Structure):
    _fields_ = [("W", C.c_uint32), ("H", C.c_uint32), ("spp", C.c_uint32), ("bounces", C.c_uint32),
                ("env", C.c_float * 3), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("sample0", C.c_uint32), ("integrator", C.c_uint32)]


def make_cfg(W, H, spp, bounces, env=(1.0, 1.0, 1.0), seed=0x5EED, tile=None, sample0=0, integrator=0):
    x0, y0, w, h = tile if tile else (0, 0, W, H)
    c = RenderCfg(W, H, spp, bounces, (C.c_float * 3)(*env), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
                  x0, y0, w, h, sample0, integrator)
    return c


# ------------------------------------------------------------------------------------------ glTF loader
@dataclass
class SceneArrays:
    """Flat description of a loaded scene; models are in the order renderer::intersect visits them."""
    model_names: list = field(default_factory=list)
    model_xform: np.ndarray = None   # [n_models, 12] origin(3) basis.x(3) basis.y(3) basis.z(3)
    model_surf: np.ndarray = None    # [n_models, 2]  first surface, surface count
    surf_range: np.ndarray = None    # [n_surf, 4]    v0, nv, t0, nt
    vertices: np.ndarray = None      # [nv, 11]       pos3 uv2 normal3 tangent3
    triangles: np.ndarray = None     # [nt, 3]        mesh-local vertex ids (uint32)
    materials: np.ndarray = None     # [n_surf, 11]   albedo3 opacity rough metal emissive3 ior shadow_catcher
    material_tex: np.ndarray = None  # [n_surf, 7]    has normal/albedo/opacity/occlusion/roughness/metallic/emissive tex
    camera: np.ndarray = None        # [14]           origin3 basis9 fov tan_half_fov
    sun: np.ndarray = None           # [13] basis9 energy3 angular_radius, or None
    images: list = field(default_factory=list)       # decoded 8-bit images [H, W, C] (as stb_image returns them)
    image_srgb: list = field(default_factory=list)   # sRGB flag each image was FIRST loaded with (renderer.cpp:33-51 cache)
    image_paths: list = field(default_factory=list)
    surf_tex: np.ndarray = None      # [n_surf, 7]    image index per material slot or -1


_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}
_CTYPE = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}


def _accessor(g, bufs, idx):
    a = g["accessors"][idx]
    bv = g["bufferViews"][a["bufferView"]]
    dt = np.dtype(_CTYPE[a["componentType"]])
    nc = _NCOMP[a["type"]]
    off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    raw = bufs[bv["buffer"]]
    out = np.ndarray((a["count"], nc), dtype=dt, buffer=raw, offset=off, strides=(stride, dt.itemsize))
    return np.array(out), a


def _unpack_floats(g, bufs, idx, float_count):
    """cgltf_accessor_unpack_floats(accessor, out, float_count) — custom_cgltf.h (whole elements only)."""
    arr, a = _accessor(g, bufs, idx)
    if arr.dtype != np.float32:
        if a.get("normalized"):
            info = np.iinfo(arr.dtype)
            arr = np.maximum(arr.astype(np.float32) / f32(info.max), f32(-1.0)) if info.min < 0 else arr.astype(np.float32) / f32(info.max)
        else:
            arr = arr.astype(np.float32)
    nc = arr.shape[1]
    avail = arr.shape[0] * nc
    float_count = min(avail, float_count)
    n_el = float_count // nc
    return arr[:n_el].reshape(-1)


def _quat_to_basis(w, x, y, z):  # math/quat.cpp:95-113 (columns x, y, z)
    one, two = f32(1), f32(2)
    bx = (one - two * (y * y + z * z), two * (x * y + z * w), two * (x * z - y * w))
    by = (two * (x * y - z * w), one - two * (x * x + z * z), two * (y * z + x * w))
    bz = (two * (x * z + y * w), two * (y * z - x * w), one - two * (x * x + y * y))
    return [list(bx), list(by), list(bz)]


def _mat_vec(b, v):  # mat3.inl:219-224
    return [b[0][i] * v[0] + b[1][i] * v[1] + b[2][i] * v[2] for i in range(3)]


def _compose(p, c):  # transform::operator* — transform.cpp:110-115 ; mat3*mat3 mat3.inl:144-152
    po, pb = p
    co, cb = c
    o = [a + b for a, b in zip(_mat_vec(pb, co), po)]
    def col(rc):
        return [pb[0][i] * rc[0] + pb[1][i] * rc[1] + pb[2][i] * rc[2] for i in range(3)]
    return (o, [col(cb[0]), col(cb[1]), col(cb[2])])


def _root_order(names):
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    out = (C.c_int * len(names))()
    k = lib().ora_root_order(len(names), arr, out)
    return [out[i] for i in range(k)]


def load_gltf(path, camera_index=0, sun_light_index=0, work=None) -> SceneArrays:
    # work: None, or {mesh name: [primitive indices]} — distributed_scene's scene_work filter (src/scene/load_gltf.cpp:93-99)
    with open(path) as fh:
        g = json.load(fh)
    base = os.path.dirname(path)
    bufs = [np.fromfile(os.path.join(base, b["uri"].replace("%20", " ")), dtype=np.uint8) for b in g.get("buffers", [])]
    if len(g.get("cameras", [])) < camera_index + 1:
        raise RuntimeError(f"Scene does not contain camera #{camera_index}.")
    cam_name = g["cameras"][camera_index].get("name")
    lights = g.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    sun_def = None
    if len(lights) >= sun_light_index + 1 and lights[sun_light_index].get("type") == "directional":
        sun_def = lights[sun_light_index]

    nodes = g["nodes"]
    ents = []  # dicts: name, local, parent, children, surfaces
    tex_cache = {}   # path -> image index: get_cached_texture (renderer.cpp:33-51) caches by path, first srgb flag wins
    out_images, out_srgb, out_paths = [], [], []

    def texture(ref, srgb):
        if ref is None:
            return -1
        uri = g["images"][g["textures"][ref["index"]]["source"]]["uri"]
        path = os.path.join(base, uri).replace("%20", " ")
        if path not in tex_cache:
            tex_cache[path] = len(out_images)
            out_images.append(_decode_image(path)); out_srgb.append(bool(srgb)); out_paths.append(path)
        return tex_cache[path]

    def make_entity(ni, parent):
        n = nodes[ni]
        light_idx = n.get("extensions", {}).get("KHR_lights_punctual", {}).get("light")
        if "camera" in n:
            name = g["cameras"][n["camera"]].get("name")
        elif light_idx is not None:
            name = lights[light_idx].get("name")
        else:
            name = n.get("name")
        r = n.get("rotation")
        q = (f32(r[3]), f32(r[0]), f32(r[1]), f32(r[2])) if r else (f32(0), f32(0), f32(0), f32(0))
        s = [f32(v) for v in n["scale"]] if "scale" in n else [f32(1)] * 3
        t = [f32(v) for v in n["translation"]] if "translation" in n else [f32(0)] * 3
        b = _quat_to_basis(*q)
        b = [[c * s[k] for c in b[k]] for k in range(3)]  # transform::make_basis — transform.cpp:21-31
        e = {"name": name, "local": (t, b), "parent": parent, "children": [], "surfaces": None,
             "is_cam": name == cam_name, "is_sun": sun_def is not None and name == sun_def.get("name")}
        ents.append(e)
        if "mesh" in n:
            mesh = g["meshes"][n["mesh"]]
            listed = None if work is None else work.get(mesh.get("name", ""), [])
            e["surfaces"] = [(_get_mesh(g, bufs, p), _get_material(g, p, texture)) for k, p in enumerate(mesh["primitives"])
                             if listed is None or k in listed]
        for ci in n.get("children", []):
            c = make_entity(ci, e)
            e["children"].append(c)
        return e

    roots = [make_entity(ni, None) for ni in g["scenes"][0]["nodes"]]

    def global_xf(e):
        return _compose(global_xf(e["parent"]), e["local"]) if e["parent"] else e["local"]

    # camera / sun: assigned in process_node pre-order, the LAST match wins (renderer.cpp:145-160)
    cam_e = [e for e in ents if e["is_cam"]]
    if not cam_e:
        raise RuntimeError("Scene is missing a camera.")
    sun_e = [e for e in ents if e["is_sun"]]

    # root container: entities[name] = entity (a repeated name replaces the earlier entity)
    last = {}
    for i, e in enumerate(roots):
        last[e["name"]] = i
    order = _root_order([e["name"] for e in roots])
    stack = [roots[last[roots[i]["name"]]] for i in order]   # pushed in iteration order
    visit = []
    while stack:                                             # renderer.cpp:653-671
        e = stack.pop()
        stack.extend(e["children"])
        if e["surfaces"] is not None:
            visit.append(e)

    out = SceneArrays()
    xf, msurf, srange, verts, tris, mats, mtex = [], [], [], [], [], [], []
    ns = nv = nt = 0
    for e in visit:
        o, b = global_xf(e)
        xf.append(o + b[0] + b[1] + b[2])
        msurf.append([ns, len(e["surfaces"])])
        out.model_names.append(e["name"])
        for (v, t), (m, tx) in e["surfaces"]:
            srange.append([nv, len(v), nt, len(t)])
            verts.append(v); tris.append(t); mats.append(m); mtex.append(tx)
            nv += len(v); nt += len(t); ns += 1
    out.model_xform = np.array(xf, dtype=np.float32).reshape(-1, 12)
    out.model_surf = np.array(msurf, dtype=np.int32).reshape(-1, 2)
    out.surf_range = np.array(srange, dtype=np.int32).reshape(-1, 4)
    out.vertices = np.concatenate(verts).astype(np.float32) if verts else np.zeros((0, 11), np.float32)
    out.triangles = np.concatenate(tris).astype(np.uint32) if tris else np.zeros((0, 3), np.uint32)
    out.materials = np.array(mats, dtype=np.float32).reshape(-1, 11)
    out.surf_tex = np.array(mtex, dtype=np.int32).reshape(-1, 7)
    out.material_tex = (out.surf_tex >= 0).astype(np.uint8)
    out.images, out.image_srgb, out.image_paths = out_images, out_srgb, out_paths
    co, cb = global_xf(cam_e[-1])
    fov = f32(g["cameras"][camera_index]["perspective"]["yfov"])
    out.camera = np.array(co + cb[0] + cb[1] + cb[2] + [fov, f32(lib().ora_tan_half_fov(C.c_float(fov)))], dtype=np.float32)
    if sun_e:
        _, sb = global_xf(sun_e[-1])
        col = [f32(c) for c in sun_def.get("color", [1, 1, 1])]
        inten = f32(sun_def.get("intensity", 1))
        out.sun = np.array(sb[0] + sb[1] + sb[2] + [c * inten for c in col] + [f32(0.004732)], dtype=np.float32)
    return out


def _get_mesh(g, bufs, prim):  # renderer.cpp:177-263
    pos = uv = nrm = tan = None
    for name, idx in prim["attributes"].items():   # JSON order; a later TEXCOORD_n overwrites an earlier one
        cnt = g["accessors"][idx]["count"]
        if name == "POSITION":
            pos = _unpack_floats(g, bufs, idx, cnt * 3)
        elif name.startswith("TEXCOORD"):
            uv = _unpack_floats(g, bufs, idx, cnt * 2)
        elif name == "NORMAL":
            nrm = _unpack_floats(g, bufs, idx, cnt * 3)
        elif name == "TANGENT":
            t = _unpack_floats(g, bufs, idx, cnt * 3)   # Q1: VEC4 accessor, count*3 floats requested
            tan = np.zeros(cnt * 3, np.float32)
            tan[:len(t)] = t
    n = len(pos) // 3
    v = np.zeros((n, 11), np.float32)
    v[:, 0:3] = pos.reshape(-1, 3)
    if uv is not None:
        v[:, 3:5] = uv[:2 * n].reshape(-1, 2)
    if nrm is not None:
        v[:, 5:8] = nrm[:3 * n].reshape(-1, 3)
    if tan is not None:
        v[:, 8:11] = tan[:3 * n].reshape(-1, 3)
    idx, _ = _accessor(g, bufs, prim["indices"])
    t = idx.reshape(-1).astype(np.uint32)
    t = t[:(len(t) // 3) * 3].reshape(-1, 3)
    return v, t


def _decode_hdr(path):
    """Radiance .hdr as stb_image's stbi_loadf decodes it: float32 [H, W, 3], rgb = byte * 2^(e - 136), black when e == 0.
    Header "#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w"; new-style RLE scanlines or flat RGBE quadruples."""
    b = open(path, "rb").read()
    lines, p = [], 0
    while True:
        q = b.index(b"\n", p)
        lines.append(b[p:q].decode("latin1")); p = q + 1
        if lines[-1] == "":
            break
    if lines[0] not in ("#?RADIANCE", "#?RGBE") or "FORMAT=32-bit_rle_rgbe" not in lines:
        raise RuntimeError("unsupported HDR")
    q = b.index(b"\n", p)
    dim = b[p:q].decode("latin1").split(); p = q + 1
    assert dim[0] == "-Y" and dim[2] == "+X"
    h, w = int(dim[1]), int(dim[3])
    rgbe = np.zeros((h, w, 4), np.uint8)
    def flat(first):
        n = h * w - first
        rgbe.reshape(-1, 4)[first:] = np.frombuffer(b, np.uint8, n * 4, p).reshape(-1, 4)
    if w < 8 or w >= 32768:
        flat(0)
    else:
        for j in range(h):
            c1, c2, l1 = b[p], b[p + 1], b[p + 2]
            if c1 != 2 or c2 != 2 or (l1 & 0x80):
                assert j == 0
                flat(0)
                break
            assert ((l1 << 8) | b[p + 3]) == w
            p += 4
            for k in range(4):
                i = 0
                while i < w:
                    cnt = b[p]; p += 1
                    if cnt > 128:
                        rgbe[j, i:i + cnt - 128, k] = b[p]; p += 1; i += cnt - 128
                    else:
                        rgbe[j, i:i + cnt, k] = np.frombuffer(b, np.uint8, cnt, p); p += cnt; i += cnt
    e = rgbe[..., 3].astype(np.int32)
    f1 = np.where(e != 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0)).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) * f1[..., None]).astype(np.float32)


def _is_hdr(path):
    with open(path, "rb") as fh:
        s = fh.read(11)
    return s.startswith(b"#?RADIANCE\n") or s.startswith(b"#?RGBE\n")


def _decode_image(path):
    """8-bit pixels with the channel count stb_image reports (req_comp = 0): L=1, LA=2, RGB=3, RGBA=4, palette -> RGB(A)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode == "P":
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    elif im.mode == "1":
        im = im.convert("L")                      # 1-bit grey: stb_image scales it to 0 / 255, one channel
    elif im.mode in ("I;16", "I"):
        return (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)[..., None]
    elif im.mode not in ("L", "LA", "RGB", "RGBA"):
        im = im.convert("RGBA")
    a = np.asarray(im, dtype=np.uint8)
    return np.ascontiguousarray(a if a.ndim == 3 else a[..., None])


def _get_material(g, prim, texture):  # renderer.cpp:265-331, defaults core/material.hpp:11-17
    if "material" not in prim:
        return [1, 1, 1, 1, 1, 1, 1, 1, 1, 1.33, 0], [-1] * 7
    m = g["materials"][prim["material"]]
    pbr = m.get("pbrMetallicRoughness", {})
    bc = pbr.get("baseColorFactor", [1, 1, 1, 1])
    em = m.get("emissiveFactor", [0, 0, 0])
    name = m.get("name", "")
    sc = 1.0 if ("shadow" in name and "catcher" in name) else 0.0
    mat = [bc[0], bc[1], bc[2], bc[3], pbr.get("roughnessFactor", 1.0), pbr.get("metallicFactor", 1.0),
           em[0], em[1], em[2], 1.33, sc]
    # texture loads in the order of renderer.cpp:297-324 (normal, albedo[/opacity], occlusion, roughness+metallic, emissive)
    t_n = texture(m.get("normalTexture"), False)
    t_a = texture(pbr.get("baseColorTexture"), True)
    t_op = t_a if (t_a >= 0 and m.get("alphaMode", "OPAQUE") != "OPAQUE") else -1
    t_oc = texture(m.get("occlusionTexture"), False)
    t_mr = texture(pbr.get("metallicRoughnessTexture"), False)
    t_e = texture(m.get("emissiveTexture"), True)
    return mat, [t_n, t_a, t_op, t_oc, t_mr, t_mr, t_e]


# ------------------------------------------------------------------------------------------ oracle scene
class OracleScene:
    def __init__(self, arrays: SceneArrays):
        self.a = arrays
        L = lib()
        a = arrays
        self._keep = [np.ascontiguousarray(x) for x in (a.model_xform, a.model_surf, a.surf_range, a.vertices,
                                                        a.triangles, a.materials, a.camera)]
        sun = np.ascontiguousarray(a.sun) if a.sun is not None else None
        self.h = C.c_void_p(L.ora_scene_create(len(a.model_xform), _p(self._keep[0]), _p(self._keep[1]),
                                               len(a.surf_range), _p(self._keep[2]), _p(self._keep[3]),
                                               _p(self._keep[4]), _p(self._keep[5]), _p(self._keep[6]),
                                               _p(sun) if sun is not None else None))
        self.n_models = len(a.model_xform)
        self.n_surf = len(a.surf_range)
        if a.images:
            self._img = [np.ascontiguousarray(im, np.uint8) for im in a.images]
            meta = np.array([[im.shape[1], im.shape[0], im.shape[2], int(sr)] for im, sr in zip(self._img, a.image_srgb)], np.int32)
            ptrs = (C.c_void_p * len(self._img))(*[im.ctypes.data for im in self._img])
            st = np.ascontiguousarray(a.surf_tex, np.int32)
            L.ora_scene_set_textures(self.h, len(self._img), _p(meta), ptrs, _p(st))

    def set_environment(self, png_path, srgb=True):
        """renderer::environment = image_texture::load(path, srgb) (renderer.hpp:28); None removes it."""
        if png_path is None:
            lib().ora_scene_set_environment(self.h, 0, 0, 0, 0, None)
            return
        if _is_hdr(png_path):                                 # image::hdr: float texels
            self._env = np.ascontiguousarray(_decode_hdr(png_path), np.float32)
            h, w, c = self._env.shape
            lib().ora_scene_set_environment_f32(self.h, w, h, c, int(bool(srgb)), _p(self._env))
            return
        self._env = np.ascontiguousarray(_decode_image(png_path), np.uint8)
        h, w, c = self._env.shape
        lib().ora_scene_set_environment(self.h, w, h, c, int(bool(srgb)), _p(self._env))

    def env_lookup(self, dirs, env_factor=(1.0, 1.0, 1.0)):
        d = np.ascontiguousarray(dirs, np.float32)
        uv = np.zeros((len(d), 2), np.float32); rgba = np.zeros((len(d), 4), np.float32); col = np.zeros((len(d), 3), np.float32)
        f = np.asarray(env_factor, np.float32)
        lib().ora_env_lookup(self.h, C.c_size_t(len(d)), _p(d), _p(f), _p(uv), _p(rgba), _p(col))
        return uv, rgba, col

    def __del__(self):
        try:
            lib().ora_scene_destroy(self.h)
        except Exception:
            pass

    def boxes(self):
        mb = np.zeros((self.n_models, 6), np.float32)
        sb = np.zeros((self.n_surf, 6), np.float32)
        lib().ora_scene_boxes(self.h, _p(mb), _p(sb))
        return mb, sb

    def kd(self, surf):
        nn, nr = C.c_int(), C.c_int()
        lib().ora_kd_counts(self.h, surf, C.byref(nn), C.byref(nr))
        nn, nr = nn.value, nr.value
        d = dict(type=np.zeros(nn, np.uint8), axis=np.zeros(nn, np.uint8), split=np.zeros(nn, np.float32),
                 left=np.zeros(nn, np.int32), right=np.zeros(nn, np.int32), first=np.zeros(nn, np.int32),
                 count=np.zeros(nn, np.int32), refs=np.zeros(nr, np.uint32))
        lib().ora_kd_get(self.h, surf, *[_p(d[k]) for k in ("type", "axis", "split", "left", "right", "first", "count", "refs")])
        return d

    def material_eval(self, surf, uv):
        uv = np.ascontiguousarray(uv, np.float32)
        out = np.zeros((len(uv), 12), np.float32)
        lib().ora_material_eval(self.h, surf, C.c_size_t(len(uv)), _p(uv), _p(out))
        return out

    def mesh_intersect(self, surf, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        n = len(rays)
        out = np.zeros((n, 4), np.float32); idx = np.zeros(n, np.int32)
        lib().ora_mesh_intersect(self.h, surf, C.c_size_t(n), _p(rays), _p(out), _p(idx))
        return out, idx

    def model_intersect(self, mdl, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        n = len(rays)
        out = np.zeros((n, 4), np.float32); idx = np.zeros((n, 2), np.int32)
        lib().ora_model_intersect(self.h, mdl, C.c_size_t(n), _p(rays), _p(out), _p(idx))
        return out, idx

    def intersect(self, rays, stats=False):
        rays = np.ascontiguousarray(rays, np.float32)
        n = len(rays)
        out = np.zeros((n, 14), np.float32); idx = np.zeros(n, np.int32)
        st = np.zeros(38, np.uint64)
        lib().ora_scene_intersect(self.h, C.c_size_t(n), _p(rays), _p(out), _p(idx), _p(st) if stats else None)
        return (out, idx, st) if stats else (out, idx)

    def camera_rays(self, ndc_ratio):
        a = np.ascontiguousarray(ndc_ratio, np.float32)
        out = np.zeros((len(a), 6), np.float32)
        lib().ora_camera_rays(self.h, C.c_size_t(len(a)), _p(a), _p(out))
        return out

    def primary_rays(self, cfg, sample):
        out = np.zeros((cfg.h, cfg.w, 6), np.float32)
        lib().ora_primary_rays(self.h, C.byref(cfg), C.c_uint32(sample), _p(out))
        return out

    def render(self, cfg, threads=0, stats=False):
        """-> (mean_rgba [h,w,4] float32, stats uint64[8] = rays, model tests, mesh tests, branches, leaves, tris, pushes, hits)"""
        img = np.zeros((cfg.h, cfg.w, 4), np.float32)
        st = np.zeros(8, np.uint64)
        lib().ora_render(self.h, C.byref(cfg), _p(img), C.c_int(threads), _p(st), C.c_int(1 if stats else 0))
        return img, st

    def trace_mt(self, rays, bounces, seed, env=(1.0, 1.0, 1.0), args_rtl=True):
        """renderer::trace(bounces, ray) for every ray in sequence on one std::mt19937(seed) stream, drawn in the reference's
        order (what oracle/_ref/ref_harness `trace` computes). -> (rgba [n,4] float32, number of rand() calls)"""
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros((len(rays), 4), np.float32)
        nd = C.c_uint64(0)
        e = np.asarray(env, np.float32)
        lib().ora_trace_mt(self.h, C.c_size_t(len(rays)), _p(rays), C.c_uint32(bounces), _p(e), C.c_uint32(seed), C.c_int(1 if args_rtl else 0),
                           _p(out), C.byref(nd))
        return out, int(nd.value)

    def render_samples(self, cfg, threads=0):
        out = np.zeros((cfg.h, cfg.w, cfg.spp, 3), np.float32)
        lib().ora_render_samples(self.h, C.byref(cfg), _p(out), C.c_int(threads))
        return out


def tri_intersect(inp):
    inp = np.ascontiguousarray(inp, np.float32); out = np.zeros((len(inp), 4), np.float32)
    lib().ora_tri_intersect(C.c_size_t(len(inp)), _p(inp), _p(out)); return out


def aabb_intersect(inp):
    inp = np.ascontiguousarray(inp, np.float32); out = np.zeros((len(inp), 3), np.float32)
    lib().ora_aabb_intersect(C.c_size_t(len(inp)), _p(inp), _p(out)); return out


def pbr(inp):
    inp = np.ascontiguousarray(inp, np.float32); out = np.zeros((len(inp), 15), np.float32)
    lib().ora_pbr(C.c_size_t(len(inp)), _p(inp), _p(out)); return out


def tonemap_write(rgba):
    a = np.ascontiguousarray(rgba, np.float32).reshape(-1, 4); out = np.zeros((len(a), 4), np.uint8)
    lib().ora_tonemap_write(C.c_size_t(len(a)), _p(a), _p(out)); return out.reshape(np.shape(rgba))


def srgb8_scan(threads=0):
    """image::write's sRGB quantiser over every float of [0, 1]: -> (first bits [256] uint32, number of places where the byte decreases)."""
    first = np.zeros(256, np.uint32)
    lib().ora_srgb8_scan.restype = C.c_uint64
    bad = lib().ora_srgb8_scan(_p(first), C.c_int(threads))
    return first, int(bad)


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 4); key = np.ascontiguousarray(key, np.uint32).reshape(-1, 2)
    out = np.zeros_like(ctr)
    lib().ora_philox(C.c_size_t(len(ctr)), _p(ctr), _p(key), _p(out)); return out


def draws(pixel, sample, seed, depth, pas, block):
    out = np.zeros(4, np.float32)
    lib().ora_draws(C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(seed & 0xFFFFFFFF), C.c_uint32(seed >> 32),
                    C.c_uint32(depth), C.c_uint32(pas), C.c_uint32(block), _p(out))
    return out


def psnr8(a, b):
    """PSNR in dB between two uint8 images (RGB channels)."""
    a = np.asarray(a, np.float64)[..., :3]; b = np.asarray(b, np.float64)[..., :3]
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
