"""distributed-path-tracer_amd — MI355X-native ray-intersection + Monte-Carlo shading integrator.

Thin ctypes binding of the C ABI in include/ptx.h (libptx_hip.so, built from csrc/ by
`__graft_entry__.build()` / `make -C distributed-path-tracer_amd/csrc`) plus a host-side mirror of the
reference's renderer interface (core::renderer, path_tracer_lib/path_tracer/core/renderer.hpp:15-36):
same field names, same defaults, `load_gltf(path)` and `render() -> PNG bytes`.

There is no CPU implementation in this package: every compute entry point goes to the HIP library
and raises PtxError when the library or a GPU is missing.

The directory name contains '-', so import it with importlib:
    ptx = importlib.import_module("distributed-path-tracer_amd")
"""
import atexit
import ctypes as C
import os
import re
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTX_LIB") or os.path.join(_HERE, "libptx_hip.so")  # PTX_LIB: experiment builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ptx.h")

OK, ERR_INVALID, ERR_IO, ERR_PARSE, ERR_NO_CAMERA, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED = range(8)
NO_SUN_LIGHT = 0xFFFFFFFF  # core::renderer::no_sun_light, renderer.hpp:19

(ARR_MODEL_XFORM, ARR_MODEL_AABB, ARR_MODEL_SURF, ARR_SURF_RANGE, ARR_MESH_AABB, ARR_VERTICES, ARR_TRIANGLES,
 ARR_MATERIALS, ARR_KD_NODES, ARR_KD_REFS, ARR_CAMERA, ARR_SUN, ARR_MODEL_NAMES, ARR_TEXTURES, ARR_TEXELS, ARR_SURF_TEX, ARR_TEXELS_F32) = range(17)


class PtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ptx error {code}: {msg}")
        self.code = code


class WorkItem(C.Structure):
    _fields_ = [("mesh_name", C.c_char_p), ("primitives", C.POINTER(C.c_int32)), ("n_primitives", C.c_uint32)]


class LoadOpts(C.Structure):
    _fields_ = [("camera_index", C.c_uint32), ("sun_light_index", C.c_uint32), ("filter_primitives", C.c_uint32),
                ("n_work", C.c_uint32), ("work", C.POINTER(WorkItem))]


class WorkerEvent(C.Structure):
    _fields_ = [("num_workers", C.c_int32), ("n_work_meshes", C.c_uint32), ("worker_id", C.c_char * 64),
                ("scene_root", C.c_char * 256), ("scene_bucket", C.c_char * 128)]


class SceneDesc(C.Structure):
    _fields_ = [("n_models", C.c_uint32), ("model_xform", C.c_void_p), ("model_surf", C.c_void_p),
                ("n_surfaces", C.c_uint32), ("surf_range", C.c_void_p), ("vertices", C.c_void_p),
                ("triangles", C.c_void_p), ("materials", C.c_void_p), ("camera", C.c_void_p), ("sun", C.c_void_p)]


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_models", "n_surfaces", "n_vertices", "n_triangles", "n_kd_nodes", "n_kd_refs",
                                          "kd_max_depth", "has_sun", "geometry_bytes", "lds_resident", "n_textures")]


class RenderCfg(C.Structure):
    _fields_ = [("W", C.c_uint32), ("H", C.c_uint32), ("spp", C.c_uint32), ("bounces", C.c_uint32),
                ("env", C.c_float * 3), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("sample0", C.c_uint32), ("spp_per_pass", C.c_uint32), ("integrator", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("shard_tile", C.c_uint32)]


INTEGRATOR_LIB, INTEGRATOR_WORKER = 0, 1   # ptx_integrator


class RenderStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("passes", C.c_uint64), ("kernel_ms", C.c_double)]


class KernelTiming(C.Structure):
    _fields_ = [("pipeline", C.c_uint32), ("steps", C.c_uint32), ("classify_ms", C.c_double), ("traverse_ms", C.c_double), ("shade_ms", C.c_double),
                ("fused_ms", C.c_double), ("fused_launches", C.c_uint32), ("pool_overflows", C.c_uint32), ("pool_pairs", C.c_uint64),
                ("peak_pairs", C.c_uint64), ("slab_paths", C.c_uint64), ("workspace_bytes", C.c_uint64), ("traverse_drain_frac", C.c_double)]


class Rays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz")]


class Hits(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("distance", "surface", "triangle", "b0", "b1", "b2", "px", "py", "pz",
                                          "nx", "ny", "nz", "u", "v")]


_lib = None
_live = weakref.WeakSet()   # scenes and contexts still open; closed before interpreter teardown (atexit)
                            # so that no HIP call is made after the HIP runtime's own static destructors ran


@atexit.register
def _close_all():
    for o in sorted(list(_live), key=lambda x: isinstance(x, Context)):   # scenes first, then contexts
        try:
            o.close()
        except Exception:
            pass


def declared_symbols():
    """Entry points declared in include/ptx.h."""
    with open(HEADER_PATH) as fh:
        txt = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z_0-9]+)\s*\(", txt)))


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (+ HSA runtime) and a second
    copy from /opt/rocm in the same process leaves whichever initialises second without a GPU. When torch is
    installed, load ITS runtime first so that libptx_hip.so's DT_NEEDED libamdhip64.so.7 binds to the same one,
    whatever the import order. (torch itself is not imported here.)"""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load libptx_hip.so. Raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtxError(ERR_NO_DEVICE, f"{LIB_PATH} is not built; run __graft_entry__.build() — there is no fallback path")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.ptx_last_error.restype = C.c_char_p
        L.ptx_version.restype = C.c_char_p
        L.ptx_ctx_stream.restype = C.c_void_p
        L.ptx_ctx_stream.argtypes = [C.c_void_p]
        L.ptx_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.ptx_ctx_destroy.argtypes = [C.c_void_p]
        L.ptx_ctx_synchronize.argtypes = [C.c_void_p]
        L.ptx_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.ptx_ctx_get_timing.argtypes = [C.c_void_p, C.POINTER(KernelTiming)]
        L.ptx_scene_load_gltf.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(LoadOpts), C.POINTER(C.c_void_p)]
        L.ptx_worker_event_load.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(RenderCfg), C.POINTER(WorkerEvent)]
        L.ptx_scene_from_arrays.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        L.ptx_scene_destroy.argtypes = [C.c_void_p]
        L.ptx_scene_set_environment.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.ptx_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        L.ptx_scene_get_array.restype = C.c_int64
        L.ptx_scene_get_array.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.ptx_render.argtypes = [C.c_void_p, C.POINTER(RenderCfg), C.c_void_p, C.POINTER(RenderStats)]
        L.ptx_intersect_batch.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits)]
        L.ptx_tonemap_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ptx_pbr_eval_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ptx_camera_rays_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ptx_reduce_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.ptx_encode_png.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ptx_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        raise PtxError(rc, lib().ptx_last_error().decode(errors="replace"))


def _ptr(x):
    """Device pointer of a torch tensor, host pointer of a numpy array, or a raw int."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    return x.data_ptr()  # torch.Tensor


class Context:
    """One per GPU (ptx_ctx): HIP device, stream and workspace."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().ptx_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device
        _live.add(self)

    @property
    def stream(self):
        return lib().ptx_ctx_stream(self.h)

    def synchronize(self):
        _check(lib().ptx_ctx_synchronize(self.h))

    def set_timing(self, on=True):
        """ptx_ctx_set_timing: collect per-kernel HIP-event times of the queue-based pipeline in renders that ask for stats."""
        _check(lib().ptx_ctx_set_timing(self.h, int(bool(on))))

    def timing(self):
        """ptx_ctx_get_timing: where the time of the last render with stats went, and the workspace it used."""
        t = KernelTiming()
        _check(lib().ptx_ctx_get_timing(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in KernelTiming._fields_}

    def reduce_framebuffer(self, nccl_comm, accum, root=0):
        """ptx_reduce_framebuffer: in-place RCCL sum-reduce of a device-resident accumulation buffer onto `root`, on this
        context's stream. `nccl_comm`: the raw ncclComm_t (an int / ctypes pointer) of a communicator the caller created."""
        n = int(accum.numel()) if hasattr(accum, "numel") else int(accum.size)
        comm = nccl_comm if isinstance(nccl_comm, C.c_void_p) else C.c_void_p(int(nccl_comm))
        _check(lib().ptx_reduce_framebuffer(self.h, comm, _ptr(accum), n, root))

    def pbr_eval(self, records):
        """ptx_pbr_eval_batch: records [n,14] float32 (normal, outcoming, incoming, u1, u2, roughness, cos_theta, ior) ->
        [n,15] float32 (rand_cone_vec, importance_diffuse, importance_specular, pdf_diffuse, pdf_specular, fresnel, reflect)."""
        a = np.ascontiguousarray(records, np.float32).reshape(-1, 14)
        out = np.zeros((len(a), 15), np.float32)
        _check(lib().ptx_pbr_eval_batch(self.h, a.ctypes.data, len(a), out.ctypes.data))
        return out

    def tonemap_encode(self, accum, W, H, spp, out=None):
        """accum: [H,W,4] float32 sums (numpy or torch-on-GPU). Returns/filles RGBA8 [H,W,4]."""
        ret = out if out is not None else np.zeros((H, W, 4), np.uint8)
        _check(lib().ptx_tonemap_encode(self.h, _ptr(accum), W, H, spp, _ptr(ret)))
        return ret

    def close(self):
        if getattr(self, "h", None):
            lib().ptx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_ARR_DTYPE = {ARR_MODEL_XFORM: (np.float32, 12), ARR_MODEL_AABB: (np.float32, 6), ARR_MODEL_SURF: (np.int32, 2),
              ARR_SURF_RANGE: (np.int32, 8), ARR_MESH_AABB: (np.float32, 6), ARR_VERTICES: (np.float32, 11),
              ARR_TRIANGLES: (np.uint32, 3), ARR_MATERIALS: (np.float32, 11), ARR_KD_NODES: (np.uint32, 2),
              ARR_KD_REFS: (np.uint32, 1), ARR_CAMERA: (np.float32, 1), ARR_SUN: (np.float32, 1),
              ARR_TEXTURES: (np.uint32, 4), ARR_TEXELS: (np.uint8, 1), ARR_SURF_TEX: (np.int32, 7), ARR_TEXELS_F32: (np.float32, 1)}


class Scene:
    """Flattened immutable scene (ptx_scene). ctx=None gives a host-only scene (inspection, no GPU work)."""

    def __init__(self, handle, ctx):
        self.h = handle
        self.ctx = ctx
        _live.add(self)

    @classmethod
    def load_gltf(cls, ctx, path, camera_index=0, sun_light_index=0, work=None):
        """work: None = every primitive (core::renderer::load_gltf); a dict {mesh name: [primitive indices]} = the host's
        per-worker filter (models::work_info::work)."""
        h = C.c_void_p()
        opts = LoadOpts(camera_index, sun_light_index, 0, 0, None)
        keep = []
        if work is not None:
            items = (WorkItem * max(len(work), 1))()
            for k, (name, prims) in enumerate(work.items()):
                arr = (C.c_int32 * max(len(prims), 1))(*prims)
                keep.append(arr)
                items[k] = WorkItem(name.encode(), arr, len(prims))
            opts = LoadOpts(camera_index, sun_light_index, 1, len(work), items)
        _check(lib().ptx_scene_load_gltf(ctx.h if ctx else None, os.fsencode(path), C.byref(opts), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def load_event(cls, ctx, event_json, local_scene_root):
        """The reference worker's Lambda event (events/event.json shape) -> (scene, RenderCfg, event info dict)."""
        h, cfg, ev = C.c_void_p(), RenderCfg(), WorkerEvent()
        _check(lib().ptx_worker_event_load(ctx.h if ctx else None, os.fsencode(event_json), os.fsencode(local_scene_root),
                                           C.byref(h), C.byref(cfg), C.byref(ev)))
        info = dict(num_workers=ev.num_workers, n_work_meshes=ev.n_work_meshes, worker_id=ev.worker_id.decode(),
                    scene_root=ev.scene_root.decode(), scene_bucket=ev.scene_bucket.decode())
        return cls(h, ctx), cfg, info

    def render_cfg(self, cfg, accum=None, want_stats=True):
        """ptx_render with a ready RenderCfg (e.g. from load_event)."""
        W, H = cfg.W, cfg.H
        w, h = (cfg.w, cfg.h) if cfg.w and cfg.h else (W, H)
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        st = RenderStats()
        _check(lib().ptx_render(self.h, C.byref(cfg), _ptr(accum), C.byref(st) if want_stats else None))
        return accum, (dict(rays=st.rays, samples=st.samples, passes=st.passes, kernel_ms=st.kernel_ms) if want_stats else None)

    @classmethod
    def from_arrays(cls, ctx, model_xform, model_surf, surf_range, vertices, triangles, materials, camera, sun=None):
        keep = [np.ascontiguousarray(model_xform, np.float32), np.ascontiguousarray(model_surf, np.int32),
                np.ascontiguousarray(np.asarray(surf_range)[:, :4], np.int32), np.ascontiguousarray(vertices, np.float32),
                np.ascontiguousarray(triangles, np.uint32), np.ascontiguousarray(materials, np.float32),
                np.ascontiguousarray(np.asarray(camera)[:13], np.float32)]
        sun_a = np.ascontiguousarray(sun, np.float32) if sun is not None and len(sun) else None
        d = SceneDesc(len(keep[0]), keep[0].ctypes.data, keep[1].ctypes.data, len(keep[2]), keep[2].ctypes.data,
                      keep[3].ctypes.data, keep[4].ctypes.data, keep[5].ctypes.data, keep[6].ctypes.data,
                      sun_a.ctypes.data if sun_a is not None else None)
        h = C.c_void_p()
        _check(lib().ptx_scene_from_arrays(ctx.h if ctx else None, C.byref(d), C.byref(h)))
        return cls(h, ctx)

    def info(self):
        i = SceneInfo()
        _check(lib().ptx_scene_get_info(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in SceneInfo._fields_}

    def array(self, which):
        n = lib().ptx_scene_get_array(self.h, which, None, 0)
        if n < 0:
            _check(ERR_INVALID)
        if which == ARR_MODEL_NAMES:
            buf = C.create_string_buffer(int(n) + 1)
            lib().ptx_scene_get_array(self.h, which, buf, n)
            return buf.raw[:n].decode().split("\n")[:-1]
        dt, cols = _ARR_DTYPE[which]
        out = np.zeros(int(n), dt)
        if n:
            got = lib().ptx_scene_get_array(self.h, which, out.ctypes.data, out.nbytes)
            assert got == n
        return out.reshape(-1, cols) if cols > 1 else out

    def render(self, W, H, spp, bounces, accum=None, env=(1.0, 1.0, 1.0), seed=0x5EED, tile=None, sample0=0,
               spp_per_pass=0, want_stats=True, integrator=INTEGRATOR_LIB, shard=None):
        """Adds radiance SUMS of samples [sample0, sample0+spp) into accum ([h,w,4] float32; numpy or torch-on-GPU).
        integrator: INTEGRATOR_LIB = core::renderer::trace, INTEGRATOR_WORKER = the HOST worker's stage pipeline.
        shard: None, or (index, count[, tile size = 64]): only the pixels in image tiles t with t % count == index are rendered.
        Returns (accum, stats dict or None)."""
        x0, y0, w, h = tile if tile else (0, 0, W, H)
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        cfg = RenderCfg(W, H, spp, bounces, (C.c_float * 3)(*env), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
                        x0, y0, w, h, sample0, spp_per_pass, integrator, *((tuple(shard) + (0,))[:3] if shard else (0, 0, 0)))
        st = RenderStats()
        _check(lib().ptx_render(self.h, C.byref(cfg), _ptr(accum), C.byref(st) if want_stats else None))
        stats = dict(rays=st.rays, samples=st.samples, passes=st.passes, kernel_ms=st.kernel_ms) if want_stats else None
        return accum, stats

    def set_environment(self, png_path, srgb=True):
        """renderer::environment = image_texture::load(png_path, srgb): the miss colour becomes map(direction) * environment_factor.
        None removes the map."""
        _check(lib().ptx_scene_set_environment(self.h, os.fsencode(png_path) if png_path is not None else None, int(bool(srgb))))

    def camera_rays(self, ndc_ratio):
        """ptx_camera_rays_batch: [n,3] float32 (ndc.x, ndc.y, aspect ratio) -> [n,6] float32 (origin, direction) = scene::camera::get_ray."""
        a = np.ascontiguousarray(ndc_ratio, np.float32).reshape(-1, 3)
        out = np.zeros((len(a), 6), np.float32)
        _check(lib().ptx_camera_rays_batch(self.h, a.ctypes.data, len(a), out.ctypes.data))
        return out

    def intersect(self, origins, dirs, attributes=True):
        """Batch closest-hit. origins/dirs: [n,3] float32 numpy. Returns dict of numpy arrays."""
        o = np.ascontiguousarray(np.asarray(origins, np.float32).T)
        d = np.ascontiguousarray(np.asarray(dirs, np.float32).T)
        n = o.shape[1]
        out = {k: np.zeros(n, np.float32) for k in ("distance", "b0", "b1", "b2")}
        out["surface"] = np.zeros(n, np.int32)
        out["triangle"] = np.zeros(n, np.int32)
        if attributes:
            out.update({k: np.zeros(n, np.float32) for k in ("px", "py", "pz", "nx", "ny", "nz", "u", "v")})
        r = Rays(*[o[k].ctypes.data for k in range(3)], *[d[k].ctypes.data for k in range(3)])
        hh = Hits(*[out[k].ctypes.data if k in out else None for k, _ in Hits._fields_])
        _check(lib().ptx_intersect_batch(self.h, C.byref(r), n, C.byref(hh)))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().ptx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_png(rgba8):
    """image::image::save_to_memory_png — RGBA8 [H,W,4] numpy -> PNG bytes."""
    a = np.ascontiguousarray(rgba8, np.uint8)
    H, W = a.shape[:2]
    p, n = C.c_void_p(), C.c_size_t()
    _check(lib().ptx_encode_png(a.ctypes.data, W, H, C.byref(p), C.byref(n)))
    try:
        return C.string_at(p, n.value)
    finally:
        lib().ptx_free(p)


class Renderer:
    """Host-side mirror of core::renderer (renderer.hpp:15-36): same public fields and defaults,
    load_gltf(path), render() -> PNG bytes (RGBA8, tonemapped, sRGB). Work happens on the GPU."""

    no_sun_light = NO_SUN_LIGHT

    def __init__(self, device=0):
        self.resolution = (1920, 1080)
        self.thread_count = 0            # kept for interface parity; the GPU grid replaces the thread pool
        self.sample_count = 10000
        self.bounce_count = 4
        self.environment = None          # path of a PNG environment map (the reference holds a loaded image::texture), sRGB-decoded
        self.environment_factor = (1.0, 1.0, 1.0)
        self.transparent_background = False
        self.camera_index = 0
        self.sun_light_index = 0
        self.visualize_kd_tree_depth = 0
        self.seed = 0x5EED
        self._ctx = Context(device)
        self._scene = None
        self.last_stats = None

    def load_gltf(self, path):
        self._scene = Scene.load_gltf(self._ctx, path, self.camera_index, self.sun_light_index)

    def render_accum(self):
        if self._scene is None:
            raise PtxError(ERR_INVALID, "render() before load_gltf()")
        if self.transparent_background or self.visualize_kd_tree_depth:
            raise PtxError(ERR_UNSUPPORTED, "transparent_background / visualize_kd_tree_depth are debug paths that are not built")
        W, H = self.resolution
        if self.environment != getattr(self, "_env_set", None):
            self._scene.set_environment(self.environment)
            self._env_set = self.environment
        accum, self.last_stats = self._scene.render(W, H, self.sample_count, self.bounce_count,
                                                    env=self.environment_factor, seed=self.seed)
        return accum

    def render(self):
        W, H = self.resolution
        accum = self.render_accum()
        return encode_png(self._ctx.tonemap_encode(accum, W, H, self.sample_count))
