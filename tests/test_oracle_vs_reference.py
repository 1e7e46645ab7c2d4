"""The oracle (oracle/pt_oracle.cpp, my CPU restatement) against golden vectors produced by the
UNMODIFIED reference (oracle/_ref, generator oracle/make_golden.py). CPU only.

Bar: bit-exact for everything built from IEEE +,-,*,/,sqrt (geometry, KD build, traversal,
hit attributes); <= 2 ulp where libm sin/cos/acos/pow are involved.
"""
import numpy as np
import pytest

from conftest import ulp_diff


def test_loader_matches_reference_scene(cornell_arrays, gold_scene):
    a, g = cornell_arrays, gold_scene
    assert a.model_names == bytes(g["model_names"]).decode().split()
    np.testing.assert_array_equal(a.model_xform, g["model_xform"])
    np.testing.assert_array_equal(a.model_surf, g["model_surf"])
    np.testing.assert_array_equal(a.surf_range, g["surf_range"][:, :4])
    np.testing.assert_array_equal(a.vertices, g["vertices"])      # includes the Q1 tangent mis-stride
    np.testing.assert_array_equal(a.triangles, g["triangles"])
    np.testing.assert_array_equal(a.materials, g["materials"])
    np.testing.assert_array_equal(a.material_tex, g["material_tex"])
    np.testing.assert_array_equal(a.camera, g["camera"])
    assert a.sun is None and g["sun"].size == 0


def test_aabbs_match_reference(cornell_oracle, gold_scene):
    mb, sb = cornell_oracle.boxes()
    np.testing.assert_array_equal(mb, gold_scene["model_aabb"])
    np.testing.assert_array_equal(sb, gold_scene["mesh_aabb"])   # includes the Q2 aabb::clear quirk (red wall max.x = 1e-4)
    assert sb[3, 3] == np.float32(1e-4)


def test_kd_trees_match_reference(cornell_oracle, gold_scene):
    g = gold_scene
    for s in range(cornell_oracle.n_surf):
        k0, nk, r0, nr = g["surf_range"][s, 4:8]
        kd = cornell_oracle.kd(s)
        assert len(kd["type"]) == nk and len(kd["refs"]) == nr
        sl = slice(k0, k0 + nk)
        np.testing.assert_array_equal(kd["type"], g["kd_type"][sl])
        br = kd["type"] == 0
        np.testing.assert_array_equal(kd["axis"][br], g["kd_axis"][sl][br])
        np.testing.assert_array_equal(kd["split"][br], g["kd_split"][sl][br])
        gl, gr = g["kd_left"][sl], g["kd_right"][sl]
        np.testing.assert_array_equal(kd["left"], np.where(gl < 0, -1, gl - k0))
        np.testing.assert_array_equal(kd["right"], np.where(gr < 0, -1, gr - k0))
        lf = ~br
        np.testing.assert_array_equal(kd["first"][lf], g["kd_first"][sl][lf] - r0)
        np.testing.assert_array_equal(kd["count"][lf], g["kd_count"][sl][lf])
        np.testing.assert_array_equal(kd["refs"], g["kd_refs"][r0:r0 + nr])
    assert g["kd_depth"].max() <= 26


def test_triangle_intersect_bit_exact(ora, gold_vec):
    out = ora.tri_intersect(gold_vec["tri_in"])
    ref = gold_vec["tri_out"]
    assert (ref[:, 0] >= 0).sum() > 200            # the fixture does exercise hits
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_aabb_intersect_bit_exact(ora, gold_vec):
    out = ora.aabb_intersect(gold_vec["aabb_in"])
    ref = gold_vec["aabb_out"]
    assert 100 < (ref[:, 0] > 0).sum() < len(ref)
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_mesh_intersect_bit_exact(cornell_oracle, gold_vec):
    rays, ref, idx = gold_vec["mesh_in"], gold_vec["mesh_out"], gold_vec["mesh_idx"]
    for s in range(cornell_oracle.n_surf):
        m = idx[:, 1] == s
        out, oi = cornell_oracle.mesh_intersect(s, rays[m])
        assert (idx[m, 0] >= 0).sum() > 50
        np.testing.assert_array_equal(oi, idx[m, 0])
        np.testing.assert_array_equal(out.view(np.uint32), ref[m].view(np.uint32))


def test_model_intersect_bit_exact(cornell_oracle, gold_vec):
    rays = gold_vec["world_rays"]
    for mdl in range(cornell_oracle.n_models):
        out, oi = cornell_oracle.model_intersect(mdl, rays)
        np.testing.assert_array_equal(oi, gold_vec["model_idx"][:, mdl])
        np.testing.assert_array_equal(out.view(np.uint32), gold_vec["model_out"][:, mdl].view(np.uint32))


def test_scene_intersect_bit_exact(cornell_oracle, gold_vec):
    out, oi = cornell_oracle.intersect(gold_vec["world_rays"])
    assert (gold_vec["scene_idx"] >= 0).mean() > 0.5
    np.testing.assert_array_equal(oi, gold_vec["scene_idx"])
    np.testing.assert_array_equal(out.view(np.uint32), gold_vec["scene_out"].view(np.uint32))


def test_pbr_functions(ora, gold_vec):
    out = ora.pbr(gold_vec["pbr_in"])
    ref = gold_vec["pbr_out"]
    # columns: rand_cone_vec(3) importance_diffuse(3) importance_specular(3) pdf_d pdf_s fresnel reflect(3)
    # The oracle calls the same libm (glibc) in the same order, so it is bit-exact here.
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_camera_rays_bit_exact(cornell_oracle, gold_vec):
    out = cornell_oracle.camera_rays(gold_vec["cam_in"])
    np.testing.assert_array_equal(out.view(np.uint32), gold_vec["cam_out"].view(np.uint32))


def test_tonemap_write_bytes_exact(ora, gold_vec):
    out = ora.tonemap_write(gold_vec["tone_in"])
    np.testing.assert_array_equal(out, gold_vec["tone_out"])


# ---------------------------------------------------------------------------- textured, sun-lit asset (jack-of-blades)
def test_jack_loader_and_trees_match_reference(jack_arrays, jack_oracle, gold_jack):
    from conftest import kd_stream_preorder, sha_u8
    a, g = jack_arrays, gold_jack
    assert a.model_names == bytes(g["model_names"]).decode().split()
    np.testing.assert_array_equal(a.model_xform, g["model_xform"])
    np.testing.assert_array_equal(a.surf_range, g["surf_range"][:, :4])
    np.testing.assert_array_equal(a.materials, g["materials"])
    np.testing.assert_array_equal(a.material_tex, g["material_tex"])
    np.testing.assert_array_equal(a.camera, g["camera"])
    np.testing.assert_array_equal(a.sun, g["sun"])                       # KHR_lights_punctual directional light, renderer.cpp:154-160
    np.testing.assert_array_equal(sha_u8(a.vertices), g["sha_vertices"])  # includes the mis-strided tangents (Q1) used by normal maps
    np.testing.assert_array_equal(sha_u8(a.triangles), g["sha_triangles"])
    mb, sb = jack_oracle.boxes()
    np.testing.assert_array_equal(mb, g["model_aabb"])
    np.testing.assert_array_equal(sb, g["mesh_aabb"])
    for s in range(jack_oracle.n_surf):                                   # 58 740 triangles, 515 135 nodes, depth 26
        kd = jack_oracle.kd(s)
        assert len(kd["type"]) == g["surf_range"][s, 5] and len(kd["refs"]) == g["surf_range"][s, 7]
        np.testing.assert_array_equal(sha_u8(kd_stream_preorder(kd)), g["sha_kd"][s])


def test_jack_material_texture_lookups_bit_exact(jack_oracle, gold_jack):
    """material::get_* over image_texture::sample: bilinear taps, unsigned-wrap of negative coordinates (Q3), sRGB pow 2.2."""
    for s in range(jack_oracle.n_surf):
        out = jack_oracle.material_eval(s, gold_jack["mat_in"][s])
        np.testing.assert_array_equal(out.view(np.uint32), gold_jack["mat_out"][s].view(np.uint32))


def test_jack_intersections_bit_exact(jack_oracle, gold_jack):
    g = gold_jack
    for s in range(jack_oracle.n_surf):
        m = g["mesh_idx"][:, 1] == s
        out, oi = jack_oracle.mesh_intersect(s, g["mesh_in"][m])
        np.testing.assert_array_equal(oi, g["mesh_idx"][m, 0])
        np.testing.assert_array_equal(out.view(np.uint32), g["mesh_out"][m].view(np.uint32))
    out, oi = jack_oracle.intersect(g["world_rays"])
    np.testing.assert_array_equal(oi, g["scene_idx"])
    np.testing.assert_array_equal(out.view(np.uint32), g["scene_out"].view(np.uint32))   # shading normal goes through the normal maps
    np.testing.assert_array_equal(jack_oracle.camera_rays(g["cam_in"]).view(np.uint32), g["cam_out"].view(np.uint32))


def test_jack_mean_image_statistics(jack_oracle, ora):
    """Sun NEE + textures + alpha in trace(): the oracle's image agrees with the reference's to within the reference's own noise."""
    import os
    from conftest import GOLD
    g = dict(np.load(os.path.join(GOLD, "jack_mean.npz")))
    W, H, spp, b = (int(v) for v in g["b4_cfg"])
    img, _ = jack_oracle.render(ora.make_cfg(W, H, spp, b), threads=0)
    o, ra, rb = img[..., :3], g["b4_a"], g["b4_b"]
    rl2 = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm((x + y) / 2)
    noise = rl2(ra, rb)
    assert rl2(o, ra) < 1.25 * noise and rl2(o, rb) < 1.25 * noise, (rl2(o, ra), rl2(o, rb), noise)
    assert abs(o.mean() / ((ra.mean() + rb.mean()) / 2) - 1) < 0.02


def test_environment_map_lookup_bit_exact(cornell_oracle, ora):
    """renderer::trace's miss branch with renderer::environment set (renderer.cpp:443-449): core::equirectangular_proj +
    image_texture::sample + environment_factor, against the compiled reference (512 directions incl. the poles and the seam)."""
    import os
    from conftest import GOLD, ROOT
    g = np.load(os.path.join(GOLD, "env_vectors.npz"))
    cornell_oracle.set_environment(os.path.join(ROOT, "scenes", "jack-of-blades", "textures", "TORSO_baseColor.png"), srgb=True)
    try:
        uv, rgba, col = cornell_oracle.env_lookup(g["env_in"], (0.5, 1.25, 2.0))
        np.testing.assert_array_equal(uv.view(np.uint32), g["env_uv"].view(np.uint32))
        np.testing.assert_array_equal(rgba.view(np.uint32), g["env_out"].view(np.uint32))
        # trace() sees ray::get_dir(): the constructor normalises the direction once more (ray.cpp:6-8, vec3.inl:250-253: v * (1 / len))
        d = g["env_in"].astype(np.float32)
        ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
        dn = d * (np.float32(1) / ln)[:, None]
        _, _, col = cornell_oracle.env_lookup(dn, (0.5, 1.25, 2.0))
        np.testing.assert_array_equal(col.view(np.uint32), g["env_trace"][:, :3].view(np.uint32))
        assert (g["env_trace"][:, 3] == 1).all()
    finally:
        cornell_oracle.set_environment(None)


# ---------------------------------------------------------------------------- the integrator's composition, bit for bit
@pytest.fixture(scope="module")
def gold_trace():
    import os
    from conftest import GOLD
    return dict(np.load(os.path.join(GOLD, "trace_vectors.npz")))


@pytest.mark.parametrize("tag", ["cornell", "jack"])
def test_trace_composition_bit_exact(tag, gold_trace, cornell_oracle, jack_oracle):
    """renderer::trace as a whole (renderer.cpp:437-643: opacity pass-through :466-472, lobe choice :490-492, sun block + shadow
    ray :498-564, BRDF / PDF combine :579-606, clamp :617-620, emissive x 10 :462, recursion) against the compiled reference run
    on ONE thread with ONE seeded mt19937 (oracle/ref_harness.cpp `trace`): the oracle replays the same std::mt19937 /
    uniform_real_distribution<float> stream in the reference's draw order and must return the same BITS for every ray —
    2000 Cornell rays x 8 bounces (emissive quad, no sun), 4000 jack-of-blades rays x 6 bounces (sun NEE, alpha, 17 textures,
    normal maps). The stream is sequential over all rays, so one wrong draw anywhere desynchronises everything after it."""
    sc = cornell_oracle if tag == "cornell" else jack_oracle
    rays, ref, meta = gold_trace[tag + "_rays"], gold_trace[tag + "_out"], gold_trace[tag + "_meta"]
    assert meta[2] == 1                                   # the reference seeded exactly one mt19937 (renderer.cpp's core::rand)
    out, n_draws = sc.trace_mt(rays, int(meta[1]), int(meta[0]))
    assert n_draws > len(rays) // 4 and (ref[:, :3].max(1) > 0).sum() > 50    # the fixture does shade surfaces
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    # renderer.cpp:500 / :572 pass two rand() calls as arguments of one call: g++ evaluates them right to left; the other order
    # visibly disagrees, i.e. the fixture is sensitive to the draw order
    other, _ = sc.trace_mt(rays, int(meta[1]), int(meta[0]), args_rtl=False)
    assert (other.view(np.uint32) != ref.view(np.uint32)).any(1).mean() > 0.02


def test_cornell_mean_image_statistics(cornell_oracle, ora, gold_mean):
    """The counter-based (Philox) stream the product shares with the oracle against the reference's mt19937 renders: converged mean
    images agree within the reference's own run-to-run noise (two independent reference halves per config), means within 0.5 %."""
    rl2 = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm((x + y) / 2)
    for tag in ("b4", "b8"):
        W, H, spp, b = (int(v) for v in gold_mean[tag + "_cfg"])
        img, _ = cornell_oracle.render(ora.make_cfg(W, H, spp, b), threads=0)
        o, ra, rb = img[..., :3], gold_mean[tag + "_a"], gold_mean[tag + "_b"]
        noise = rl2(ra, rb)
        assert rl2(o, ra) < 1.1 * noise and rl2(o, rb) < 1.1 * noise, (tag, rl2(o, ra), rl2(o, rb), noise)
        assert abs(o.mean() / ((ra.mean() + rb.mean()) / 2) - 1) < 0.005, tag


def test_srgb_quantiser_is_a_monotone_step_function(ora, gold_vec):
    """image::write (image.cpp:143-154): byte = uint8(powf(v, 1/2.2f) * 255 + 0.5f). Over EVERY float of [0, 1] (1.07e9 values, glibc
    powf) the byte never decreases, so the product may evaluate it as a 255-threshold step function (kernels.hip srgb8) and be exact
    against the reference's bytes. The thresholds found here reproduce the reference's fixture bytes."""
    first, decreasing = ora.srgb8_scan()
    assert decreasing == 0
    assert first[0] == 0 and (np.diff(first.astype(np.int64)) > 0).all() and first[255] <= 0x3F800000
    thr = first.view(np.float32)
    tin = gold_vec["tone_in"].reshape(-1, 4)
    x = tin[:, :3]
    v = np.clip((x * (np.float32(2.51) * x + np.float32(0.03))) / (x * (np.float32(2.43) * x + np.float32(0.59)) + np.float32(0.14)), 0, 1).astype(np.float32)
    got = (np.searchsorted(thr[1:], v.reshape(-1), side="right")).reshape(v.shape)        # number of thresholds <= v
    np.testing.assert_array_equal(got, gold_vec["tone_out"].reshape(-1, 4)[:, :3])
