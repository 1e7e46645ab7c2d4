"""Seeded procedural scenes (numpy only) for the BASELINE configs whose assets the reference does not ship
(no Stanford bunny anywhere, `sponza.bin` missing): displaced icospheres of a chosen triangle
count, placed either in the reference's Cornell room or on an open sun-lit plaza. Every generator is
deterministic, so the GPU box regenerates identical geometry. Output = the flat arrays both
`Scene.from_arrays` (product) and the oracle's `SceneArrays` take."""
import numpy as np

_T = (1.0 + 5.0 ** 0.5) / 2.0


def icosphere(level: int):
    """Unit icosphere: 20 * 4**level triangles (level 6 -> 81 920, level 7 -> 327 680)."""
    v = np.array([[-1, _T, 0], [1, _T, 0], [-1, -_T, 0], [1, -_T, 0], [0, -1, _T], [0, 1, _T], [0, -1, -_T], [0, 1, -_T],
                  [_T, 0, -1], [_T, 0, 1], [-_T, 0, -1], [-_T, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7],
                  [9, 8, 1]], np.int64)
    for _ in range(level):
        n = len(v)
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.sort(e, axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid])
        m = (n + inv.reshape(3, -1).T)            # midpoint ids per face: ab, bc, ca
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        ab, bc, ca = m[:, 0], m[:, 1], m[:, 2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v, f


def displaced_icosphere(level: int, seed: int = 1, amplitude: float = 0.15):
    """-> vertices [n,11] float32 (position, uv, normal, tangent), triangles [m,3] uint32.
    Displacement: a few seeded low-frequency lobes along the normal (keeps the mesh a star-shaped solid)."""
    v, f = icosphere(level)
    rng = np.random.default_rng(seed)
    r = np.ones(len(v))
    for _ in range(6):
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        k = rng.integers(2, 7)
        r += amplitude / 6 * np.cos(k * np.arccos(np.clip(v @ d, -1, 1)) + rng.uniform(0, 6.28))
    p = v * r[:, None]
    # area-weighted vertex normals
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    nrm = np.zeros_like(p)
    for k in range(3):
        np.add.at(nrm, f[:, k], fn)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    up = np.where(np.abs(nrm[:, 1:2]) < 0.99, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    tan = np.cross(up, nrm)
    tan /= np.maximum(np.linalg.norm(tan, axis=1, keepdims=True), 1e-20)
    uv = np.stack([np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi) + 0.5, np.arcsin(np.clip(v[:, 1], -1, 1)) / np.pi + 0.5], 1)
    out = np.concatenate([p, uv, nrm, tan], 1).astype(np.float32)
    return out, f.astype(np.uint32)


def _quad(y, half, normal_up=True):
    p = np.array([[-half, y, -half], [half, y, -half], [half, y, half], [-half, y, half]], np.float32)
    n = np.array([0, 1, 0], np.float32)
    v = np.zeros((4, 11), np.float32)
    v[:, 0:3] = p
    v[:, 3:5] = [[0, 0], [1, 0], [1, 1], [0, 1]]
    v[:, 5:8] = n
    v[:, 8:11] = [1, 0, 0]
    t = np.array([[0, 2, 1], [0, 3, 2]], np.uint32)   # counter-clockwise seen from +y
    return v, t


def _look_at(eye, target):
    """Camera basis columns (x, y, z) with -z pointing at the target, as scene::camera expects (camera.cpp:10-21)."""
    eye, target = np.asarray(eye, np.float64), np.asarray(target, np.float64)
    z = eye - target
    z /= np.linalg.norm(z)
    x = np.cross([0, 1, 0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.concatenate([eye, x, y, z]).astype(np.float32)


def plaza_scene(level: int = 3, sun: bool = True, alpha: bool = True, seed: int = 3):
    """Open scene: ground quad (shadow catcher when `alpha`), a displaced icosphere (scaled + translated model, glossy
    metal), a second small sphere (half-transparent when `alpha`), one directional light when `sun`.
    -> dict of arrays: model_xform, model_surf, surf_range, vertices, triangles, materials, camera[13], sun[13] or None."""
    gv, gt = _quad(0.0, 6.0)
    s1v, s1t = displaced_icosphere(level, seed, 0.2)
    s2v, s2t = displaced_icosphere(max(level - 1, 0), seed + 1, 0.05)
    verts = np.concatenate([gv, s1v, s2v])
    tris = np.concatenate([gt, s1t, s2t])
    surf_range = np.array([[0, len(gv), 0, len(gt)], [len(gv), len(s1v), len(gt), len(s1t)],
                           [len(gv) + len(s1v), len(s2v), len(gt) + len(s1t), len(s2t)]], np.int32)
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    model_xform = np.array([[0, 0, 0] + ident,
                            [0.3, 1.25, -0.2, 1.1, 0, 0, 0, 1.1, 0, 0, 0, 1.1],
                            [-1.9, 0.62, 1.1, 0.6, 0, 0, 0, 0.6, 0, 0, 0, 0.6]], np.float32)
    model_surf = np.array([[0, 1], [1, 1], [2, 1]], np.int32)
    #            albedo            opacity rough metal emissive     ior   shadow_catcher
    materials = np.array([[0.75, 0.72, 0.68, 1.0, 0.6, 0.0, 0, 0, 0, 1.33, 1.0 if alpha else 0.0],
                          [0.95, 0.64, 0.54, 1.0, 0.25, 1.0, 0, 0, 0, 1.33, 0.0],
                          [0.2, 0.5, 0.9, 0.55 if alpha else 1.0, 0.4, 0.0, 0.05, 0.05, 0.1, 1.33, 0.0]], np.float32)
    cam = np.concatenate([_look_at([4.5, 3.2, 6.5], [0.0, 0.9, 0.0]), [np.float32(0.7)]]).astype(np.float32)
    sun13 = None
    if sun:
        d = np.array([0.35, 0.8, 0.45])   # direction TOWARDS the sun = basis * (0,0,1) (renderer.cpp:499)
        d /= np.linalg.norm(d)
        x = np.cross([0, 1, 0], d)
        x /= np.linalg.norm(x)
        y = np.cross(d, x)
        sun13 = np.concatenate([x, y, d, [3.0, 2.7, 2.2], [0.004732]]).astype(np.float32)
    return dict(model_xform=model_xform, model_surf=model_surf, surf_range=surf_range, vertices=verts.astype(np.float32),
                triangles=tris.astype(np.uint32), materials=materials, camera=cam, sun=sun13)


def cornell_with_mesh(cornell: dict, level: int = 6, seed: int = 7):
    """The reference's Cornell room (arrays as loaded from scenes/cornell-box) with its 960-triangle sphere replaced by
    a displaced icosphere of 20*4**level triangles: the "~70k-triangle mesh, KD-tree traversal stress" of BASELINE
    config 3 (level 6 = 81 920 triangles) and, at level 7 (327 680), the "~250k-triangle" class of configs 4-5."""
    sr = np.asarray(cornell["surf_range"])[:, :4]
    keep = len(sr) - 1                                  # the sphere is the last surface / model
    v0, nv, t0, nt = sr[keep]
    mv, mt = displaced_icosphere(level, seed, 0.12)
    verts = np.concatenate([np.asarray(cornell["vertices"])[:v0], mv])
    tris = np.concatenate([np.asarray(cornell["triangles"])[:t0], mt])
    surf_range = np.concatenate([sr[:keep], [[v0, len(mv), t0, len(mt)]]]).astype(np.int32)
    return dict(model_xform=np.asarray(cornell["model_xform"], np.float32), model_surf=np.asarray(cornell["model_surf"], np.int32),
                surf_range=surf_range, vertices=verts.astype(np.float32), triangles=tris.astype(np.uint32),
                materials=np.asarray(cornell["materials"], np.float32), camera=np.asarray(cornell["camera"], np.float32)[:13], sun=None)


def _grid(n: int, origin, du, dv):
    """n x n quads (2 n^2 triangles) spanning origin + s*du + t*dv, s,t in [0,1]; normal = normalize(du x dv)."""
    origin, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), indexing="ij")
    p = origin + s[..., None] * du + t[..., None] * dv
    nrm = np.cross(du, dv)
    nrm /= np.linalg.norm(nrm)
    tan = du / np.linalg.norm(du)
    v = np.zeros(((n + 1) ** 2, 11), np.float32)
    v[:, 0:3] = p.reshape(-1, 3)
    v[:, 3:5] = np.stack([s, t], -1).reshape(-1, 2)
    v[:, 5:8] = nrm
    v[:, 8:11] = tan
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).ravel()
    b, c, d = a + (n + 1), a + (n + 1) + 1, a + 1
    tri = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.uint32)
    return v, tri


def _place(v, scale, offset):
    """Scale + translate a vertex array in place of a node transform (one model, many surfaces); normals by the inverse-transpose."""
    out = v.copy()
    sc = np.asarray(scale, np.float32)
    out[:, 0:3] = v[:, 0:3] * sc + np.asarray(offset, np.float32)
    n = v[:, 5:8] / sc
    out[:, 5:8] = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    t = v[:, 8:11] * sc
    out[:, 8:11] = t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-20)
    return out


def atrium_scene(detail: int = 5, seed: int = 11):
    """Sponza-class stand-in for BASELINE configs 4-5 (`sponza.bin` is missing from the reference): ONE model with 24 surfaces
    (as Sponza's single mesh has 24 primitives) — a gridded floor and three walls, two rows of five tall columns, eight
    ornaments, a lintel and a plinth — lit by one directional light through the open roof. detail = 5 gives 262 176 triangles
    (Sponza's accessors total 262 267); every level down divides the sphere-derived part by four."""
    parts = []
    g = max(detail - 1, 0)
    parts.append(_grid(4 << g, [-6, 0, -14], [0, 0, 28], [12, 0, 0]))                 # floor, normal +y
    parts.append(_grid(2 << g, [-6, 0, -14], [0, 9, 0], [0, 0, 28]))                  # left wall, normal +x
    parts.append(_grid(2 << g, [6, 0, 14], [0, 9, 0], [0, 0, -28]))                   # right wall, normal -x
    parts.append(_grid(2 << g, [-6, 0, -14], [12, 0, 0], [0, 9, 0]))                  # back wall, normal +z
    col_v, col_t = displaced_icosphere(detail, seed, 0.10)
    for k in range(10):
        x = -3.2 if k % 2 == 0 else 3.2
        z = -11.0 + 5.0 * (k // 2)
        parts.append((_place(col_v, (0.55, 3.6, 0.55), (x, 3.6, z)), col_t))
    orn_v, orn_t = displaced_icosphere(max(detail - 1, 0), seed + 1, 0.25)
    for k in range(8):
        x = -1.6 + 3.2 * (k % 2)
        z = -9.0 + 5.5 * (k // 2)
        parts.append((_place(orn_v, (0.7, 0.7, 0.7), (x, 0.75 + 0.5 * (k % 3), z)), orn_t))
    parts.append(_grid(1 << g, [-6, 8.2, -14], [12, 0, 0], [0, 0, 6]))                # lintel over the far end, normal -y
    parts.append(_grid((7 << g) // 4 or 1, [-2, 0.02, 6], [0, 0, 4], [4, 0, 0]))      # plinth, normal +y
    assert len(parts) == 24
    verts = np.concatenate([p[0] for p in parts]).astype(np.float32)
    tris = np.concatenate([p[1] for p in parts]).astype(np.uint32)
    sr, v0, t0 = [], 0, 0
    for v, t in parts:
        sr.append([v0, len(v), t0, len(t)])
        v0 += len(v); t0 += len(t)
    rng = np.random.default_rng(seed)
    mats = np.zeros((24, 11), np.float32)
    mats[:, 0:3] = 0.35 + 0.55 * rng.random((24, 3))
    mats[:, 3] = 1.0
    mats[:, 4] = 0.25 + 0.7 * rng.random(24)
    mats[4:14, 5] = (rng.random(10) < 0.3)            # a few metallic columns
    mats[:, 9] = 1.45
    d = np.array([0.25, 0.85, 0.35])                  # towards the sun
    d /= np.linalg.norm(d)
    x = np.cross([0, 1, 0], d); x /= np.linalg.norm(x)
    y = np.cross(d, x)
    sun13 = np.concatenate([x, y, d, [4.0, 3.7, 3.2], [0.004732]]).astype(np.float32)
    cam = np.concatenate([_look_at([0.3, 2.2, 13.0], [0.0, 2.6, -6.0]), [np.float32(0.9)]]).astype(np.float32)
    ident = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    return dict(model_xform=np.array([[0, 0, 0] + ident], np.float32), model_surf=np.array([[0, 24]], np.int32),
                surf_range=np.array(sr, np.int32), vertices=verts, triangles=tris, materials=mats, camera=cam, sun=sun13)
