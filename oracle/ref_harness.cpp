// TEST INFRASTRUCTURE — drives the UNMODIFIED reference library (compiled from
// /root/reference by oracle/Makefile, target `ref`) to produce golden vectors and
// to serve as the timed CPU baseline ("cpu_baseline.kind = reference" in bench.py).
//
// Nothing in here is reference code: it only *calls* the reference's public (and,
// through the `#define private public` include trick, two private) entry points:
//   core::renderer::load_gltf / render / trace / intersect   (core/renderer.hpp:35-53)
//   scene::model::intersect (scene/model.cpp:20), core::mesh::intersect (core/mesh.cpp:300)
//   geometry::triangle::intersect (geometry/triangle.cpp:120), geometry::aabb::intersect (aabb.cpp:41)
//   core::pbr::* (core/pbr.cpp), util::rand_cone_vec, scene::camera::get_ray, image::image::write
//
// Sub-commands (all paths are given on the command line; nothing is read implicitly):
//   scene   <gltf> <outdir>                     dump the loaded scene + KD trees as .npy
//   vectors <gltf> <outdir> <seed> <n>          function-level known-answer vectors as .npy
//   materials <gltf> <outdir> <seed> <n>        per surface: material::get_* (texture lookups) at n random uvs
//   envmap <gltf> <png> <srgb> <outdir> <seed> <n>   environment-map lookups: equirectangular_proj + image_texture::sample + trace() on misses
//   image   <file> <outdir> <seed> <n>               decoded pixels of image::image::load's stb_image call (JPEG / PNG) + n bilinear
//                                                     image_texture::sample lookups on it (linear and sRGB)
//   trace   <gltf> <outdir> <seed> <n> <bounces>      trace() of n rays in sequence, ONE thread, ONE seeded mt19937 stream: pins the
//                                                     integrator's composition bit for bit (oracle side: ora_trace_mt)
//   mean    <gltf> <out.npy> W H spp bounces threads   float32 mean image by calling trace()
//   render  <gltf> W H spp bounces threads [out.png]   time renderer::render(), print JSON

#include <path_tracer/pch.hpp>
#include <path_tracer/core/material.hpp>
#include <path_tracer/core/mesh.hpp>
#include <path_tracer/core/pbr.hpp>
#include <path_tracer/core/utils.hpp>
#include <path_tracer/image/image.hpp>
#include <path_tracer/image/texture.hpp>
#include <path_tracer/image/image_texture.hpp>
#include <path_tracer/scene/camera.hpp>
#include <path_tracer/scene/entity.hpp>
#include <path_tracer/scene/model.hpp>
#include <path_tracer/scene/sun_light.hpp>
#include <path_tracer/util/rand_cone_vec.hpp>
#define private public
#include <path_tracer/core/renderer.hpp>
#undef private

#include <chrono>
#include <fstream>
#include <thread>

using namespace math;

// ---- deterministic seeding of the reference's thread_local mt19937 (core/utils.hpp:9) ----
// libstdc++'s std::random_device::operator() inlines to the out-of-line _M_getval();
// the executable's definition wins over libstdc++.so's. Each call returns seed+k so that
// every pool thread gets its own stream.
static std::atomic<unsigned> g_seed_calls{0};
unsigned int std::random_device::_M_getval() {
	const char* s = getenv("ORACLE_SEED");
	unsigned base = s ? (unsigned)strtoul(s, nullptr, 10) : 12345u;
	return base + 7919u * g_seed_calls.fetch_add(1);
}

// ---- tiny .npy writer ----
static void save_npy(const std::string& path, const char* descr, const std::vector<size_t>& shape,
                     const void* data, size_t bytes) {
	std::string shp = "(";
	for (size_t i = 0; i < shape.size(); i++) shp += std::to_string(shape[i]) + ",";
	shp += ")";
	std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + shp + ", }";
	size_t total = 10 + hdr.size() + 1;
	size_t pad = (64 - total % 64) % 64;
	hdr += std::string(pad, ' ') + "\n";
	std::ofstream f(path, std::ios::binary);
	const char magic[] = "\x93NUMPY\x01\x00";
	f.write(magic, 8);
	uint16_t hl = (uint16_t)hdr.size();
	f.write((const char*)&hl, 2);
	f.write(hdr.data(), hdr.size());
	f.write((const char*)data, bytes);
}
template <typename T> static const char* descr_of();
template <> const char* descr_of<float>() { return "<f4"; }
template <> const char* descr_of<uint32_t>() { return "<u4"; }
template <> const char* descr_of<int32_t>() { return "<i4"; }
template <> const char* descr_of<uint8_t>() { return "|u1"; }
template <typename T>
static void save(const std::string& dir, const std::string& name, const std::vector<T>& v,
                 std::vector<size_t> shape = {}) {
	if (shape.empty()) shape = {v.size()};
	save_npy(dir + "/" + name + ".npy", descr_of<T>(), shape, v.data(), v.size() * sizeof(T));
}

// ---- input generator (harness-owned PCG32; inputs are stored with the outputs) ----
struct pcg32 {
	uint64_t state, inc;
	explicit pcg32(uint64_t seed) : state(0), inc((seed << 1) | 1) { next(); state += 0x853c49e6748fea9bULL; next(); }
	uint32_t next() {
		uint64_t old = state;
		state = old * 6364136223846793005ULL + inc;
		uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((32 - rot) & 31));
	}
	float uni() { return (next() >> 8) * (1.0f / 16777216.0f); }
	float range(float a, float b) { return a + (b - a) * uni(); }
	fvec3 vec(float a, float b) { float x = range(a, b), y = range(a, b), z = range(a, b); return fvec3(x, y, z); }
	fvec3 dir() {
		for (;;) {
			fvec3 v = vec(-1, 1);
			float l = dot(v, v);
			if (l > 1e-4f && l <= 1) return normalize(v);
		}
	}
};
static void push3(std::vector<float>& o, const fvec3& v) { o.push_back(v.x); o.push_back(v.y); o.push_back(v.z); }

// ---- scene walk in the exact order renderer::intersect visits models (renderer.cpp:646-671) ----
struct model_ref { scene::entity* ent; std::shared_ptr<scene::model> model; };
static std::vector<model_ref> visit_order(const core::renderer& r) {
	std::vector<model_ref> out;
	std::stack<scene::entity*> stack;
	for (const auto& [_, e] : r.entities) stack.push(e.get());
	while (!stack.empty()) {
		scene::entity* e = stack.top();
		stack.pop();
		for (const auto& c : e->get_children()) stack.push(c.get());
		if (auto m = e->get_component<scene::model>()) out.push_back({e, m});
	}
	return out;
}

struct kd_flat {
	std::vector<uint8_t> type, axis;  // type 0 = branch, 1 = leaf
	std::vector<float> split;
	std::vector<int32_t> left, right, first, count;
	std::vector<uint32_t> refs;
	int max_depth = 0;
};
static int32_t flatten_kd(const core::kd_tree_node* n, kd_flat& o, int depth) {
	if (!n) return -1;
	int32_t id = (int32_t)o.type.size();
	o.max_depth = std::max(o.max_depth, depth);
	if (auto b = dynamic_cast<const core::kd_tree_branch*>(n)) {
		o.type.push_back(0); o.axis.push_back(b->axis); o.split.push_back(b->split);
		o.left.push_back(-1); o.right.push_back(-1); o.first.push_back(0); o.count.push_back(0);
		int32_t l = flatten_kd(b->left.get(), o, depth + 1);
		int32_t rr = flatten_kd(b->right.get(), o, depth + 1);
		o.left[id] = l; o.right[id] = rr;
	} else {
		auto leaf = static_cast<const core::kd_tree_leaf*>(n);
		o.type.push_back(1); o.axis.push_back(0); o.split.push_back(0);
		o.left.push_back(-1); o.right.push_back(-1);
		o.first.push_back((int32_t)o.refs.size()); o.count.push_back((int32_t)leaf->indices.size());
		for (uint32_t i : leaf->indices) o.refs.push_back(i);
	}
	return id;
}

static void load(core::renderer& r, const char* gltf) {
	std::cout.setstate(std::ios_base::failbit);  // silence the reference's progress chatter
	r.load_gltf(gltf);
	std::cout.clear();
}

static int cmd_scene(const char* gltf, const std::string& dir) {
	core::renderer r;
	load(r, gltf);
	std::filesystem::create_directories(dir);
	auto models = visit_order(r);

	std::vector<float> xf;            // per model: origin(3) + basis x,y,z (9)
	std::vector<float> model_aabb;    // per model: min(3) max(3)
	std::vector<int32_t> model_surf;  // per model: first surface, count
	std::vector<float> mesh_aabb;     // per surface
	std::vector<int32_t> surf_rng;    // per surface: vert0, nvert, tri0, ntri, kd0, nkd, ref0, nref
	std::vector<float> verts;         // 11 floats per vertex (pos3 uv2 n3 t3)
	std::vector<uint32_t> tris;       // 3 per triangle (mesh-local vertex ids)
	std::vector<float> mats;          // per surface: albedo3 opacity rough metal emissive3 ior shadow_catcher
	std::vector<uint8_t> mat_tex;     // per surface: 7 has-texture flags
	kd_flat kd;
	std::vector<int32_t> kd_depth;
	std::string names;
	int32_t nsurf = 0;
	for (auto& m : models) {
		const scene::transform& t = m.ent->get_global_transform();
		push3(xf, t.origin); push3(xf, t.basis.x); push3(xf, t.basis.y); push3(xf, t.basis.z);
		push3(model_aabb, m.model->aabb.min); push3(model_aabb, m.model->aabb.max);
		model_surf.push_back(nsurf); model_surf.push_back((int32_t)m.model->surfaces.size());
		names += m.ent->get_name() + "\n";
		for (auto& s : m.model->surfaces) {
			nsurf++;
			push3(mesh_aabb, s.mesh->aabb.min); push3(mesh_aabb, s.mesh->aabb.max);
			int32_t v0 = (int32_t)(verts.size() / 11), t0 = (int32_t)(tris.size() / 3);
			for (auto& v : s.mesh->vertices) {
				push3(verts, v.position); verts.push_back(v.tex_coord.x); verts.push_back(v.tex_coord.y);
				push3(verts, v.normal); push3(verts, v.tangent);
			}
			for (auto& tr : s.mesh->triangles) { tris.push_back(tr.x); tris.push_back(tr.y); tris.push_back(tr.z); }
			int32_t k0 = (int32_t)kd.type.size(), r0 = (int32_t)kd.refs.size();
			kd_flat one;
			flatten_kd(s.mesh->kd_tree.get(), one, 1);
			kd_depth.push_back(one.max_depth);
			// append with node ids rebased to the global array
			for (size_t i = 0; i < one.type.size(); i++) {
				kd.type.push_back(one.type[i]); kd.axis.push_back(one.axis[i]); kd.split.push_back(one.split[i]);
				kd.left.push_back(one.left[i] < 0 ? -1 : one.left[i] + k0);
				kd.right.push_back(one.right[i] < 0 ? -1 : one.right[i] + k0);
				kd.first.push_back(one.first[i] + r0); kd.count.push_back(one.count[i]);
			}
			for (uint32_t x : one.refs) kd.refs.push_back(x);
			surf_rng.insert(surf_rng.end(), {v0, (int32_t)s.mesh->vertices.size(), t0, (int32_t)s.mesh->triangles.size(),
			                                 k0, (int32_t)one.type.size(), r0, (int32_t)one.refs.size()});
			auto& mt = *s.material;
			push3(mats, mt.albedo_fac); mats.push_back(mt.opacity_fac); mats.push_back(mt.roughness_fac);
			mats.push_back(mt.metallic_fac); push3(mats, mt.emissive_fac); mats.push_back(mt.ior);
			mats.push_back(mt.shadow_catcher ? 1.f : 0.f);
			mat_tex.insert(mat_tex.end(), {(uint8_t)!!mt.normal_tex, (uint8_t)!!mt.albedo_tex, (uint8_t)!!mt.opacity_tex,
			                               (uint8_t)!!mt.occlusion_tex, (uint8_t)!!mt.roughness_tex,
			                               (uint8_t)!!mt.metallic_tex, (uint8_t)!!mt.emissive_tex});
		}
	}
	save(dir, "model_xform", xf, {models.size(), 12});
	save(dir, "model_aabb", model_aabb, {models.size(), 6});
	save(dir, "model_surf", model_surf, {models.size(), 2});
	save(dir, "mesh_aabb", mesh_aabb, {(size_t)nsurf, 6});
	save(dir, "surf_range", surf_rng, {(size_t)nsurf, 8});
	save(dir, "vertices", verts, {verts.size() / 11, 11});
	save(dir, "triangles", tris, {tris.size() / 3, 3});
	save(dir, "materials", mats, {(size_t)nsurf, 11});
	save(dir, "material_tex", mat_tex, {(size_t)nsurf, 7});
	save(dir, "kd_type", kd.type); save(dir, "kd_axis", kd.axis); save(dir, "kd_split", kd.split);
	save(dir, "kd_left", kd.left); save(dir, "kd_right", kd.right);
	save(dir, "kd_first", kd.first); save(dir, "kd_count", kd.count); save(dir, "kd_refs", kd.refs);
	save(dir, "kd_depth", kd_depth);
	std::vector<uint8_t> nm(names.begin(), names.end());
	save(dir, "model_names", nm);

	// camera (scene/camera.cpp:10-30) and sun
	auto cam = r.camera->get_component<scene::camera>();
	const scene::transform& ct = r.camera->get_global_transform();
	std::vector<float> camv;
	push3(camv, ct.origin); push3(camv, ct.basis.x); push3(camv, ct.basis.y); push3(camv, ct.basis.z);
	camv.push_back(cam->get_fov()); camv.push_back(math::tan(cam->get_fov() * 0.5F));  // same expression as camera.cpp:29
	save(dir, "camera", camv);
	std::vector<float> sun;
	if (r.sun_light) {
		const scene::transform& st = r.sun_light->get_global_transform();
		auto sl = r.sun_light->get_component<scene::sun_light>();
		push3(sun, st.basis.x); push3(sun, st.basis.y); push3(sun, st.basis.z);
		push3(sun, sl->energy); sun.push_back(sl->angular_radius);
	}
	save(dir, "sun", sun);
	std::cerr << "scene: " << models.size() << " models, " << nsurf << " surfaces, " << tris.size() / 3
	          << " tris, " << kd.type.size() << " kd nodes, " << kd.refs.size() << " leaf refs\n";
	return 0;
}

static int cmd_vectors(const char* gltf, const std::string& dir, uint64_t seed, size_t n) {
	core::renderer r;
	load(r, gltf);
	std::filesystem::create_directories(dir);
	auto models = visit_order(r);
	pcg32 g(seed);

	{  // geometry::triangle::intersect — triangle.cpp:120-190
		std::vector<float> in, out;
		for (size_t i = 0; i < n; i++) {
			fvec3 a = g.vec(-2, 2), b = g.vec(-2, 2), c = g.vec(-2, 2);
			fvec3 o = g.vec(-3, 3);
			fvec3 d;
			uint32_t kind = g.next() % 8;
			if (kind < 5) {  // aim at a point near the triangle so that many rays hit
				float u = g.uni(), v = g.uni() * (1 - u);
				fvec3 p = a * (1 - u - v) + b * u + c * v + g.vec(-0.05f, 0.05f);
				d = normalize(p - o);
			} else if (kind == 5) {  // aim exactly at an edge / vertex: exercises the +-epsilon slack
				float u = g.uni();
				fvec3 p = (g.next() & 1) ? a * (1 - u) + b * u : a;
				d = normalize(p - o);
			} else if (kind == 6) {  // axis-aligned direction (zeros in dir => inf in aabb, not here)
				d = fvec3(0, 0, 0); d[g.next() % 3] = (g.next() & 1) ? 1.f : -1.f;
			} else d = g.dir();
			if (i % 97 == 0) c = a + (b - a) * 2.0f;  // degenerate (collinear) triangle: det == 0 path
			geometry::ray ray(o, d);
			geometry::triangle tri(a, b, c);
			auto h = tri.intersect(ray);
			push3(in, a); push3(in, b); push3(in, c); push3(in, ray.origin); push3(in, ray.get_dir());
			out.push_back(h.distance); push3(out, h.barycentric);
		}
		save(dir, "tri_in", in, {n, 15}); save(dir, "tri_out", out, {n, 4});
	}
	{  // geometry::aabb::intersect — aabb.cpp:41-67
		std::vector<float> in, out;
		for (size_t i = 0; i < n; i++) {
			fvec3 p = g.vec(-2, 2), q = g.vec(-2, 2);
			geometry::aabb box(math::min(p, q), math::max(p, q));
			if (i % 53 == 0) std::swap(box.min.x, box.max.x);  // degenerate: any(min > max)
			if (i % 59 == 0) box.max.y = box.min.y;             // flat box
			fvec3 o = g.vec(-3, 3), d = g.dir();
			if (i % 7 == 0) { d = fvec3(0, 0, 0); d[g.next() % 3] = (g.next() & 1) ? 1.f : -1.f; }
			if (i % 11 == 0) o = box.min + (box.max - box.min) * g.vec(0, 1);  // origin inside
			geometry::ray ray(o, d);
			auto h = box.intersect(ray);
			push3(in, box.min); push3(in, box.max); push3(in, ray.origin); push3(in, ray.get_dir());
			out.push_back(h.has_hit() ? 1.f : 0.f);
			out.push_back(h.has_hit() ? h.near : 0.f); out.push_back(h.has_hit() ? h.far : -1.f);
		}
		save(dir, "aabb_in", in, {n, 12}); save(dir, "aabb_out", out, {n, 3});
	}

	// scene bounds in world space (for ray generation)
	fvec3 wmin(1e30f), wmax(-1e30f);
	for (auto& m : models) {
		const scene::transform& t = m.ent->get_global_transform();
		for (int k = 0; k < 8; k++) {
			fvec3 c((k & 1) ? m.model->aabb.max.x : m.model->aabb.min.x, (k & 2) ? m.model->aabb.max.y : m.model->aabb.min.y,
			        (k & 4) ? m.model->aabb.max.z : m.model->aabb.min.z);
			fvec3 w = t * c;
			wmin = math::min(wmin, w); wmax = math::max(wmax, w);
		}
	}
	auto world_ray = [&](size_t i) {
		fvec3 o = wmin + (wmax - wmin) * g.vec(0.02f, 0.98f);
		fvec3 d = g.dir();
		if (i % 5 == 0) {  // camera-like rays
			auto cam = r.camera->get_component<scene::camera>();
			return cam->get_ray(fvec2(g.range(-1, 1), g.range(-1, 1)), 16.f / 9.f);
		}
		if (i % 13 == 0) { d = fvec3(0, 0, 0); d[g.next() % 3] = (g.next() & 1) ? 1.f : -1.f; }
		return geometry::ray(o, d);
	};

	{  // core::mesh::intersect — mesh.cpp:300-405 — local-space rays against every surface's mesh
		std::vector<float> in, out;
		std::vector<int32_t> idx;
		size_t per = n;
		int32_t sid = 0;
		for (auto& m : models)
			for (auto& s : m.model->surfaces) {
				auto& bb = s.mesh->aabb;
				fvec3 ext = bb.max - bb.min;
				for (size_t i = 0; i < per; i++) {
					fvec3 o = bb.min - ext * 0.5f + (ext * 2.0f) * g.vec(0, 1);
					fvec3 d;
					if (i % 3) {  // aim at a random vertex-ish point so that hits are frequent
						auto& v = s.mesh->vertices[g.next() % s.mesh->vertices.size()].position;
						d = normalize(v + g.vec(-0.2f, 0.2f) * ext - o);
					} else d = g.dir();
					geometry::ray ray(o, d);
					auto h = s.mesh->intersect(ray);
					push3(in, ray.origin); push3(in, ray.get_dir());
					out.push_back(h.distance);
					if (h.has_hit()) { push3(out, h.barycentric); idx.push_back((int32_t)h.index); }
					else { push3(out, fvec3(0)); idx.push_back(-1); }
					idx.push_back(sid);
				}
				sid++;
			}
		size_t tot = in.size() / 6;
		save(dir, "mesh_in", in, {tot, 6}); save(dir, "mesh_out", out, {tot, 4}); save(dir, "mesh_idx", idx, {tot, 2});
	}
	{  // scene::model::intersect (model.cpp:20-72) and renderer::intersect (renderer.cpp:645-725)
		std::map<const scene::model::surface*, int32_t> surf_id;
		std::map<const core::material*, int32_t> mat_id;
		int32_t sid = 0;
		for (auto& m : models) for (auto& s : m.model->surfaces) { surf_id[&s] = sid; mat_id[s.material.get()] = sid; sid++; }
		std::vector<float> in, mout, rout;
		std::vector<int32_t> midx, ridx;
		size_t nr = n * 4;
		for (size_t i = 0; i < nr; i++) {
			geometry::ray ray = world_ray(i);
			push3(in, ray.origin); push3(in, ray.get_dir());
			for (auto& m : models) {
				auto h = m.model->intersect(ray);
				mout.push_back(h.distance);
				if (h.has_hit()) { push3(mout, h.barycentric); midx.push_back(surf_id[h.surface]); midx.push_back((int32_t)h.triangle_index); }
				else { push3(mout, fvec3(0)); midx.push_back(-1); midx.push_back(-1); }
			}
			auto res = r.intersect(ray);
			ridx.push_back(res.hit ? mat_id[res.material.get()] : -1);
			if (res.hit) {
				push3(rout, res.position); rout.push_back(res.tex_coord.x); rout.push_back(res.tex_coord.y);
				push3(rout, res.normal); push3(rout, res.tangent); push3(rout, res.get_normal());
			} else for (int k = 0; k < 14; k++) rout.push_back(0);
		}
		save(dir, "world_rays", in, {nr, 6});
		save(dir, "model_out", mout, {nr, models.size(), 4}); save(dir, "model_idx", midx, {nr, models.size(), 2});
		save(dir, "scene_out", rout, {nr, 14}); save(dir, "scene_idx", ridx, {nr});
	}
	{  // core::pbr::*, util::rand_cone_vec, core::reflect — pbr.cpp, rand_cone_vec.cpp:8-35
		std::vector<float> in, out;
		for (size_t i = 0; i < n; i++) {
			fvec3 nrm = g.dir();
			if (i % 17 == 0) { nrm = fvec3(0, 0, 0); nrm[g.next() % 3] = (g.next() & 1) ? 1.f : -1.f; }
			fvec3 o;
			do { o = g.dir(); } while (dot(nrm, o) <= 0.001f);
			fvec3 inc;
			do { inc = g.dir(); } while (dot(nrm, inc) <= 0.001f);
			float u1 = g.uni(), u2 = g.uni();
			float rough = math::max(g.uni(), 0.05F);
			if (i % 19 == 0) rough = 0.05F;
			if (i % 23 == 0) u1 = 0.f;
			float cos_theta = g.range(-1, 1);
			float ior = g.range(1.0f, 2.5f);
			push3(in, nrm); push3(in, o); push3(in, inc);
			in.push_back(u1); in.push_back(u2); in.push_back(rough); in.push_back(cos_theta); in.push_back(ior);
			push3(out, util::rand_cone_vec(u2, cos_theta, nrm));
			push3(out, core::pbr::importance_diffuse(fvec2(u1, u2), nrm, o));
			push3(out, core::pbr::importance_specular(fvec2(u1, u2), nrm, o, rough));
			out.push_back(core::pbr::pdf_diffuse(nrm, inc));
			out.push_back(core::pbr::pdf_specular(nrm, o, inc, rough));
			out.push_back(core::pbr::fresnel(o, core::reflect(-o, nrm), ior));
			push3(out, core::reflect(-o, nrm));
		}
		save(dir, "pbr_in", in, {n, 14}); save(dir, "pbr_out", out, {n, 15});
	}
	{  // scene::camera::get_ray — camera.cpp:10-21 — NDC grid
		std::vector<float> in, out;
		auto cam = r.camera->get_component<scene::camera>();
		for (int j = 0; j <= 16; j++)
			for (int i = 0; i <= 16; i++) {
				fvec2 ndc(-1 + i / 8.0f, -1 + j / 8.0f);
				float ratio = (j & 1) ? 16.f / 9.f : 1.f;
				auto ray = cam->get_ray(ndc, ratio);
				in.push_back(ndc.x); in.push_back(ndc.y); in.push_back(ratio);
				push3(out, ray.origin); push3(out, ray.get_dir());
			}
		save(dir, "cam_in", in, {in.size() / 3, 3}); save(dir, "cam_out", out, {out.size() / 6, 6});
	}
	{  // core::tonemap_approx_aces (utils.hpp:29-36) + image::image::write (image.cpp:143-154)
		std::vector<float> in;
		std::vector<uint8_t> out;
		uint32_t w = 64, h = (uint32_t)((n + 63) / 64);
		image::image img(uvec2(w, h), 4, false, true);
		for (uint32_t y = 0; y < h; y++)
			for (uint32_t x = 0; x < w; x++) {
				float s = math::pow(10.0f, g.range(-4, 2));
				fvec3 c = fvec3(g.uni(), g.uni(), g.uni()) * s;
				if ((x + y) % 29 == 0) c = fvec3(0);
				float a = (x % 5 == 0) ? g.uni() : 1.f;
				fvec3 t = core::tonemap_approx_aces(c);
				img.write(uvec2(x, y), 0, t.x); img.write(uvec2(x, y), 1, t.y);
				img.write(uvec2(x, y), 2, t.z); img.write(uvec2(x, y), 3, a);
				push3(in, c); in.push_back(a);
			}
		// recover the stored bytes by decoding the PNG the reference itself writes (image.cpp:111-122)
		auto png = img.save_to_memory_png();
		int W, H, C;
		unsigned char* px = stbi_load_from_memory(png.data(), (int)png.size(), &W, &H, &C, 4);
		out.assign(px, px + (size_t)W * H * 4);
		stbi_image_free(px);
		save(dir, "tone_in", in, {(size_t)h, (size_t)w, 4}); save(dir, "tone_out", out, {(size_t)h, (size_t)w, 4});
	}
	return 0;
}

// core::material::get_normal/albedo/opacity/roughness/metallic/emissive (core/material.cpp:6-53) over
// image::image_texture::sample (image/image_texture.cpp:21-62) and image::image::read (image/image.cpp:124-141)
static int cmd_materials(const char* gltf, const std::string& dir, uint64_t seed, size_t n) {
	core::renderer r;
	load(r, gltf);
	std::filesystem::create_directories(dir);
	auto models = visit_order(r);
	pcg32 g(seed);
	std::vector<float> in, out;
	size_t ns = 0;
	for (auto& m : models)
		for (auto& s : m.model->surfaces) {
			ns++;
			for (size_t i = 0; i < n; i++) {
				fvec2 uv(g.range(-1.5f, 2.5f), g.range(-1.5f, 2.5f));
				if (i % 4 == 0) uv = fvec2(g.uni(), g.uni());
				if (i % 31 == 0) uv = fvec2((float)(g.next() % 5) - 2.0f, (float)(g.next() % 5) - 2.0f);   // exact texel-grid edges
				auto& mt = *s.material;
				fvec3 nrm = mt.get_normal(uv), alb = mt.get_albedo(uv), em = mt.get_emissive(uv);
				in.push_back(uv.x); in.push_back(uv.y);
				push3(out, nrm); push3(out, alb); out.push_back(mt.get_opacity(uv)); out.push_back(mt.get_roughness(uv));
				out.push_back(mt.get_metallic(uv)); push3(out, em);
			}
		}
	save(dir, "mat_in", in, {ns, n, 2});
	save(dir, "mat_out", out, {ns, n, 12});
	return 0;
}

// renderer::trace's miss branch with an environment map (renderer.cpp:443-449): core::equirectangular_proj
// (core/utils.hpp:22-27) and image::image_texture::sample on a PNG loaded the way a caller of the library would
// (image_texture::load(path, srgb)). Also one trace() per direction from far outside the scene, with the map set.
static int cmd_envmap(const char* gltf, const char* png, int srgb, const std::string& dir, uint64_t seed, size_t n) {
	core::renderer r;
	load(r, gltf);
	std::filesystem::create_directories(dir);
	auto tex = image::image_texture::load(png, srgb != 0);
	r.environment = tex;
	r.environment_factor = fvec3(0.5F, 1.25F, 2.0F);
	pcg32 g(seed);
	std::vector<float> in, uvs, out, tr;
	for (size_t i = 0; i < n; i++) {
		fvec3 d(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1));
		if (i < 6) { d = fvec3(0); d[i / 2] = (i % 2) ? -1.0F : 1.0F; }      // the six axis directions (poles, seam)
		d = normalize(d);
		fvec2 uv = core::equirectangular_proj(d);
		fvec4 c = tex->sample(uv);
		push3(in, d); uvs.push_back(uv.x); uvs.push_back(uv.y);
		out.push_back(c.x); out.push_back(c.y); out.push_back(c.z); out.push_back(c.w);
		fvec4 t = r.trace(4, geometry::ray(fvec3(1000, 1000, 1000) + d, d));   // starts outside every box, points away: a miss
		tr.push_back(t.x); tr.push_back(t.y); tr.push_back(t.z); tr.push_back(t.w);
	}
	save(dir, "env_in", in, {n, 3}); save(dir, "env_uv", uvs, {n, 2}); save(dir, "env_out", out, {n, 4}); save(dir, "env_trace", tr, {n, 4});
	return 0;
}

// image::image::load (image/image.cpp:23-54): stbi_load(path, &w, &h, &channels, 0) — the decoded bytes — and image_texture::sample
// (image/image_texture.cpp:21-62) at n random uvs, loaded once as linear and once as sRGB.
static int cmd_image(const char* file, const std::string& dir, uint64_t seed, size_t n) {
	std::filesystem::create_directories(dir);
	int w = 0, h = 0, c = 0;
	if (stbi_is_hdr(file)) {   // image::load's HDR branch: stbi_loadf, the floats are kept
		float* pf = stbi_loadf(file, &w, &h, &c, 0);
		if (!pf) { fprintf(stderr, "image: stbi_loadf failed: %s\n", stbi_failure_reason()); return 2; }
		std::vector<float> data(pf, pf + (size_t)w * h * c);
		stbi_image_free(pf);
		save(dir, "pixels", data, {(size_t)h, (size_t)w, (size_t)c});
	} else {
		unsigned char* px = stbi_load(file, &w, &h, &c, 0);
		if (!px) { fprintf(stderr, "image: stbi_load failed: %s\n", stbi_failure_reason()); return 2; }
		std::vector<uint8_t> data(px, px + (size_t)w * h * c);
		stbi_image_free(px);
		save(dir, "pixels", data, {(size_t)h, (size_t)w, (size_t)c});
	}
	pcg32 g(seed);
	std::vector<float> uv, lin, srgb;
	auto tl = image::image_texture::load(file, false), ts = image::image_texture::load(file, true);
	for (size_t i = 0; i < n; i++) {
		fvec2 p(g.range(-1.5f, 2.5f), g.range(-1.5f, 2.5f));
		if (i % 4 == 0) p = fvec2(g.uni(), g.uni());
		uv.push_back(p.x); uv.push_back(p.y);
		fvec4 a = tl->sample(p), b = ts->sample(p);
		lin.insert(lin.end(), {a.x, a.y, a.z, a.w}); srgb.insert(srgb.end(), {b.x, b.y, b.z, b.w});
	}
	save(dir, "uv", uv, {n, 2}); save(dir, "sample_linear", lin, {n, 4}); save(dir, "sample_srgb", srgb, {n, 4});
	return 0;
}

// renderer::trace (renderer.cpp:437-643) on n rays, one after the other, on the calling thread only. core::rand() is a `static`
// function of core/utils.hpp, so renderer.cpp's copy owns its own thread_local mt19937, seeded at its first call from
// std::random_device — the shim above, which has not been called before (this harness draws its inputs from pcg32): the stream is
// std::mt19937(ORACLE_SEED). Half of the rays are camera rays (random NDC), half start at the camera and aim at a random point of
// the scene's bounds (so that assets most camera rays miss are still exercised); every ray is traced with the full bounce budget.
static int cmd_trace(const char* gltf, const std::string& dir, uint64_t seed, size_t n, uint32_t bounces) {
	core::renderer r;
	load(r, gltf);
	std::filesystem::create_directories(dir);
	r.bounce_count = (uint8_t)bounces;
	auto models = visit_order(r);
	pcg32 g(seed);
	fvec3 wmin(1e30f), wmax(-1e30f);
	for (auto& m : models) {
		const scene::transform& t = m.ent->get_global_transform();
		for (int k = 0; k < 8; k++) {
			fvec3 c((k & 1) ? m.model->aabb.max.x : m.model->aabb.min.x, (k & 2) ? m.model->aabb.max.y : m.model->aabb.min.y,
			        (k & 4) ? m.model->aabb.max.z : m.model->aabb.min.z);
			fvec3 w = t * c;
			wmin = math::min(wmin, w); wmax = math::max(wmax, w);
		}
	}
	auto cam = r.camera->get_component<scene::camera>();
	const fvec3 eye = r.camera->get_global_transform().origin;
	std::vector<float> in, out;
	if (g_seed_calls.load() != 0) { fprintf(stderr, "trace: random_device was already used\n"); return 3; }
	const size_t skip = getenv("ORACLE_TRACE_SKIP") ? strtoull(getenv("ORACLE_TRACE_SKIP"), nullptr, 10) : 0;
	for (size_t i = 0; i < n; i++) {
		geometry::ray ray = (i & 1) ? geometry::ray(eye, normalize(wmin + (wmax - wmin) * g.vec(0.05f, 0.95f) - eye))
		                            : cam->get_ray(fvec2(g.range(-1, 1), g.range(-1, 1)), 16.f / 9.f);
		if (i < skip) continue;   // ORACLE_TRACE_SKIP: inputs are still drawn, so ray i is the same ray whatever is skipped
		fvec4 c = r.trace((uint8_t)bounces, ray);
		push3(in, ray.origin); push3(in, ray.get_dir());
		out.push_back(c.x); out.push_back(c.y); out.push_back(c.z); out.push_back(c.w);
	}
	if (g_seed_calls.load() > 1) { fprintf(stderr, "trace: more than one mt19937 stream was seeded (%u)\n", g_seed_calls.load()); return 3; }
	save(dir, "trace_rays", in, {in.size() / 6, 6}); save(dir, "trace_out", out, {out.size() / 4, 4});
	const char* sd = getenv("ORACLE_SEED");
	std::vector<uint32_t> meta = {(uint32_t)(sd ? strtoul(sd, nullptr, 10) : 12345u), bounces, (uint32_t)g_seed_calls.load()};
	save(dir, "trace_meta", meta);
	return 0;
}

// float32 mean image: same pixel loop as renderer::render (renderer.cpp:354-402) but keeping the
// float running mean instead of the 8-bit image; rows are distributed statically over threads.
static int cmd_mean(const char* gltf, const std::string& out, uint32_t W, uint32_t H, uint32_t spp, uint32_t bounces,
                    uint32_t threads) {
	core::renderer r;
	load(r, gltf);
	r.resolution = uvec2(W, H);
	r.bounce_count = (uint8_t)bounces;
	std::vector<float> img((size_t)W * H * 3, 0.f);
	std::atomic<uint64_t> rays{0};
	auto cam = r.camera->get_component<scene::camera>();
	auto work = [&](uint32_t tid) {
		for (uint32_t y = tid; y < H; y += threads)
			for (uint32_t x = 0; x < W; x++) {
				fvec3 c(0);
				for (uint32_t s = 0; s < spp; s++) {
					fvec2 aa = fvec2(core::rand(), core::rand());
					fvec2 ndc = ((fvec2(uvec2(x, y)) + aa) / uvec2(W, H)) * 2 - fvec2::one;
					ndc.y = -ndc.y;
					float ratio = static_cast<float>(W) / H;
					fvec4 d = r.trace((uint8_t)bounces, cam->get_ray(ndc, ratio));
					c = c * s + fvec3(d);
					c /= s + 1;
				}
				size_t i = ((size_t)y * W + x) * 3;
				img[i] = c.x; img[i + 1] = c.y; img[i + 2] = c.z;
			}
	};
	std::vector<std::thread> th;
	for (uint32_t t = 0; t < threads; t++) th.emplace_back(work, t);
	for (auto& t : th) t.join();
	save_npy(out, "<f4", {H, W, 3}, img.data(), img.size() * 4);
	return 0;
}

static int cmd_render(const char* gltf, uint32_t W, uint32_t H, uint32_t spp, uint32_t bounces, uint32_t threads,
                      const char* out) {
	core::renderer r;
	load(r, gltf);
	r.resolution = uvec2(W, H);
	r.sample_count = spp;
	r.bounce_count = (uint8_t)bounces;
	r.thread_count = threads;
	std::cout.setstate(std::ios_base::failbit);
	auto t0 = std::chrono::steady_clock::now();
	auto png = r.render();
	auto t1 = std::chrono::steady_clock::now();
	std::cout.clear();
	double sec = std::chrono::duration<double>(t1 - t0).count();
	if (out) { std::ofstream f(out, std::ios::binary); f.write((const char*)png.data(), png.size()); }
	uint32_t used = threads ? threads : std::thread::hardware_concurrency();
	printf("{\"seconds\": %.6f, \"samples\": %llu, \"msamples_per_s\": %.6f, \"threads\": %u, \"W\": %u, \"H\": %u, "
	       "\"spp\": %u, \"bounces\": %u}\n",
	       sec, (unsigned long long)W * H * spp, (double)W * H * spp / sec / 1e6, used, W, H, spp, bounces);
	return 0;
}

int main(int argc, char** argv) {
	std::string cmd = argc > 1 ? argv[1] : "";
	try {
		if (cmd == "scene" && argc == 4) return cmd_scene(argv[2], argv[3]);
		if (cmd == "vectors" && argc == 6) return cmd_vectors(argv[2], argv[3], strtoull(argv[4], 0, 10), strtoull(argv[5], 0, 10));
		if (cmd == "materials" && argc == 6) return cmd_materials(argv[2], argv[3], strtoull(argv[4], 0, 10), strtoull(argv[5], 0, 10));
		if (cmd == "envmap" && argc == 8) return cmd_envmap(argv[2], argv[3], atoi(argv[4]), argv[5], strtoull(argv[6], 0, 10), strtoull(argv[7], 0, 10));
		if (cmd == "image" && argc == 6) return cmd_image(argv[2], argv[3], strtoull(argv[4], 0, 10), strtoull(argv[5], 0, 10));
		if (cmd == "trace" && argc == 7) return cmd_trace(argv[2], argv[3], strtoull(argv[4], 0, 10), strtoull(argv[5], 0, 10), atoi(argv[6]));
		if (cmd == "mean" && argc == 9)
			return cmd_mean(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]));
		if (cmd == "render" && argc >= 8)
			return cmd_render(argv[2], atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), argc > 8 ? argv[8] : nullptr);
	} catch (const std::exception& e) {
		fprintf(stderr, "ref_harness: %s\n", e.what());
		return 2;
	}
	fprintf(stderr, "usage: ref_harness scene|vectors|materials|envmap|image|trace|mean|render ... (see header comment)\n");
	return 1;
}
