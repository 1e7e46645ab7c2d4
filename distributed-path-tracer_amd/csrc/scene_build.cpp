// Host-side scene finalisation: AABBs, SAH KD-trees with the reference's topology, flattening.
//
// What is reproduced (so that traversal order, tie-breaks and therefore pixels match the reference):
//   mesh::recalculate_aabb      LIB/core/mesh.cpp:254-261   (+epsilon pad; aabb::clear quirk aabb.cpp:29-32)
//   model::recalculate_aabb     LIB/scene/model.cpp:13-18
//   kd_tree_builder::init_node_sah / split_triangles / split_aabb   LIB/core/mesh.cpp:21-80,131-247
//   mesh::build_kd_tree         LIB/core/mesh.cpp:263-298   (SAH, max depth 25)
//   transform::inverse          LIB/scene/transform.cpp:33-36, mat3 inverse LIB/math/mat3.inl:245-263
// How it differs: the tree is built over index lists (no triangle copies), emitted breadth-first into
// 8-byte nodes whose children are adjacent, and leaf references hold GLOBAL triangle ids.
// All arithmetic is float32 in the reference's operation order; build with -ffp-contract=off.
#include "flat_scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <atomic>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>

namespace ptx {
namespace {

constexpr float kEps = 0.0001f;  // math::epsilon, LIB/math/math.hpp:16

struct F3 { float v[3]; };
struct Box { float lo[3], hi[3]; };

inline float pick_max(float a, float b) { return b > a ? b : a; }  // math::max, math.inl:169
inline float pick_min(float a, float b) { return b < a ? b : a; }  // math::min, math.inl:179

inline void box_reset(Box& b) {
	// aabb::clear(): min = FLT_MAX, max = FLT_MIN (the smallest POSITIVE float, not -FLT_MAX) — kept on purpose (Q2)
	for (int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<float>::max(); b.hi[k] = std::numeric_limits<float>::min(); }
}
inline void box_grow(Box& b, const float* p) {
	for (int k = 0; k < 3; k++) { b.lo[k] = pick_min(b.lo[k], p[k]); b.hi[k] = pick_max(b.hi[k], p[k]); }
}
inline float box_area(const Box& b) {
	float wx = b.hi[0] - b.lo[0], wy = b.hi[1] - b.lo[1], wz = b.hi[2] - b.lo[2];
	return (wx * wy + wy * wz + wx * wz) * 2;
}

// ---- 3x3 helpers on column-major float[9] = {x.x,x.y,x.z, y.x,...} ----
inline void mat_inverse(const float* m, float* out) {
	const float xx = m[0], xy = m[1], xz = m[2], yx = m[3], yy = m[4], yz = m[5], zx = m[6], zy = m[7], zz = m[8];
	float det1 = +(yy * zz - zy * yz);
	float det2 = -(xy * zz - zy * xz);
	float det3 = +(xy * yz - yy * xz);
	float det = xx * det1 + yx * det2 + zx * det3;
	float s = 1 / det;
	float r[9] = {det1, det2, det3,
	              -(yx * zz - zx * yz), +(xx * zz - zx * xz), -(xx * yz - yx * xz),
	              +(yx * zy - zx * yy), -(xx * zy - zx * xy), +(xx * yy - yx * xy)};
	for (int k = 0; k < 9; k++) out[k] = r[k] * s;
}
inline void mat_mul_vec(const float* m, const float* v, float* out) {  // rows dotted with v (mat3.inl:219-224)
	for (int r = 0; r < 3; r++) out[r] = m[r] * v[0] + m[3 + r] * v[1] + m[6 + r] * v[2];
}

// ---- SAH builder -------------------------------------------------------------------------------
struct BuildNode {
	bool leaf = false;
	uint8_t axis = 0;
	float split = 0;
	int32_t left = -1, right = -1;
	std::vector<uint32_t> ids;  // leaf: mesh-local triangle ids, in ascending (= inherited) order
};

struct MeshBuilder {
	const std::vector<F3>& pa; const std::vector<F3>& pb; const std::vector<F3>& pc;  // corner positions per triangle
	std::vector<BuildNode> nodes;
	uint32_t max_depth_seen = 0;

	// par_levels > 0: the left subtree of a large node is built by a second thread into a builder of its own and grafted in afterwards
	// (child links are indices into `nodes`; the emitted tree does not depend on where a node sits in that vector)
	int32_t build(const Box& box, std::vector<uint32_t>&& ids, int levels_left, uint32_t depth, int par_levels = 0) {
		int32_t me = (int32_t)nodes.size();
		nodes.emplace_back();
		max_depth_seen = std::max(max_depth_seen, depth);
		if (levels_left == 0) {
			nodes[me].leaf = true;
			nodes[me].ids = std::move(ids);
			return me;
		}
		const size_t n = ids.size();
		const float base_cost = n * box_area(box);
		float best_cost = base_cost, best_split = 0;
		int best_axis = 0;
		// Same element type, comparator and std::sort as mesh.cpp:148-163 so that runs of equal keys
		// come out in the same (implementation-defined) order as in the reference on this libstdc++.
		std::vector<std::tuple<float, bool>> ev;
		ev.reserve(n * 2);
		for (int axis = 0; axis < 3; axis++) {
			ev.clear();
			for (uint32_t t : ids) {
				float a = pa[t].v[axis], b = pb[t].v[axis], c = pc[t].v[axis];
				ev.emplace_back(pick_min(pick_min(a, b), c), true);
				ev.emplace_back(pick_max(pick_max(a, b), c), false);
			}
			std::sort(ev.begin(), ev.end(), [](auto& x, auto& y) { return std::get<0>(x) < std::get<0>(y); });
			float split = 0;
			uint32_t nl = 0, nr = (uint32_t)n;
			for (size_t i = 0; i <= ev.size(); i++) {
				if (i == 0) split = std::get<0>(ev.front()) - kEps;
				else if (i == ev.size()) { nr--; split = std::get<0>(ev.back()) + kEps; }
				else {
					if (std::get<1>(ev[i - 1])) nl++;
					else nr--;
					if (std::get<0>(ev[i - 1]) == std::get<0>(ev[i])) continue;
					split = (std::get<0>(ev[i - 1]) + std::get<0>(ev[i])) * 0.5F;
				}
				if (split <= box.lo[axis]) continue;
				if (split >= box.hi[axis]) break;
				Box l = box, r = box;
				l.hi[axis] = split;
				r.lo[axis] = split;
				float cost = nl * box_area(l) + nr * box_area(r);
				if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = split; }
			}
		}
		if (!(best_cost < base_cost)) {
			nodes[me].leaf = true;
			nodes[me].ids = std::move(ids);
			return me;
		}
		nodes[me].axis = (uint8_t)best_axis;
		nodes[me].split = best_split;
		Box l = box, r = box;
		l.hi[best_axis] = best_split;
		r.lo[best_axis] = best_split;
		std::vector<uint32_t> lids, rids;
		lids.reserve(n);
		rids.reserve(n);
		for (uint32_t t : ids) {  // a triangle goes left if ANY corner is < split, right if ANY corner is >= split
			bool any_l = false, any_r = false;
			const float c3[3] = {pa[t].v[best_axis], pb[t].v[best_axis], pc[t].v[best_axis]};
			for (float c : c3) (c < best_split ? any_l : any_r) = true;
			if (any_l) lids.push_back(t);
			if (any_r) rids.push_back(t);
		}
		std::vector<uint32_t>().swap(ids);
		if (par_levels > 0 && !lids.empty() && !rids.empty() && lids.size() + rids.size() >= 16384) {
			MeshBuilder sub{pa, pb, pc, {}, 0};
			// an exception on either side (std::bad_alloc on a huge mesh) must not unwind past a joinable thread: the helper's is
			// carried over in `sub_err`, the thread is always joined, and whichever came first is rethrown on this thread
			std::exception_ptr sub_err;
			std::thread th([&] {
				try { sub.build(l, std::move(lids), levels_left - 1, depth + 1, par_levels - 1); } catch (...) { sub_err = std::current_exception(); }
			});
			struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_guard{th};
			const int32_t c = build(r, std::move(rids), levels_left - 1, depth + 1, par_levels - 1);
			nodes[me].right = c;
			th.join();
			if (sub_err) std::rethrow_exception(sub_err);
			const int32_t off = (int32_t)nodes.size();
			nodes.reserve(nodes.size() + sub.nodes.size());
			for (BuildNode& bn : sub.nodes) {
				if (bn.left >= 0) bn.left += off;
				if (bn.right >= 0) bn.right += off;
				nodes.push_back(std::move(bn));
			}
			nodes[me].left = off;
			max_depth_seen = std::max(max_depth_seen, sub.max_depth_seen);
			return me;
		}
		if (!lids.empty()) { int32_t c = build(l, std::move(lids), levels_left - 1, depth + 1, par_levels); nodes[me].left = c; }
		if (!rids.empty()) { int32_t c = build(r, std::move(rids), levels_left - 1, depth + 1, par_levels); nodes[me].right = c; }
		return me;
	}
};

}  // namespace

void finalize_scene(FlatScene& s, const float* cam, const float* sun) {
	const size_t n_models = s.model_surf.size() / 2;
	const size_t n_surf = s.surf_range.size() / 8;
	s.models.assign(n_models, ModelRec{});
	s.surfaces.assign(n_surf, SurfaceRec{});
	s.materials.assign(n_surf, MaterialRec{});
	s.kd_nodes.clear(); s.kd_refs.clear(); s.tris.clear(); s.tri_isect.clear(); s.hitrec.clear();
	s.kd_max_depth = 0;
	s.any_texture = false;
	s.any_alpha = false;

	// Per surface: records (in surface order), then the SAH builds — the expensive part, independent of each other: in parallel, one
	// surface per thread, and the top levels of a surface's own recursion on further threads when there are fewer surfaces than cores
	// (PTX_BUILD_THREADS overrides the thread count; the emitted arrays do not depend on it) — then the emission, in surface order again.
	struct SurfaceBuild { std::vector<F3> pa, pb, pc; Box box; std::unique_ptr<MeshBuilder> tree; };
	std::vector<SurfaceBuild> sb(n_surf);
	for (size_t si = 0; si < n_surf; si++) {
		int32_t* rg = &s.surf_range[8 * si];
		const int32_t v0 = rg[0], nvs = rg[1], t0 = rg[2], nt = rg[3];
		SurfaceRec& sr = s.surfaces[si];
		Box mb;
		box_reset(mb);
		for (int32_t k = 0; k < nvs; k++) box_grow(mb, &s.vertices[11 * (size_t)(v0 + k)]);
		for (int k = 0; k < 3; k++) { mb.lo[k] -= kEps; mb.hi[k] += kEps; }
		memcpy(sr.bmin, mb.lo, 12);
		memcpy(sr.bmax, mb.hi, 12);
		sr.tri_base = (uint32_t)t0;

		std::vector<F3>& pa = sb[si].pa; std::vector<F3>& pb = sb[si].pb; std::vector<F3>& pc = sb[si].pc;
		pa.resize(nt); pb.resize(nt); pc.resize(nt);
		sb[si].box = mb;
		for (int32_t t = 0; t < nt; t++) {
			const uint32_t* ix = &s.triangles[3 * (size_t)(t0 + t)];
			const float* a = &s.vertices[11 * (size_t)(v0 + ix[0])];
			const float* b = &s.vertices[11 * (size_t)(v0 + ix[1])];
			const float* c = &s.vertices[11 * (size_t)(v0 + ix[2])];
			memcpy(pa[t].v, a, 12); memcpy(pb[t].v, b, 12); memcpy(pc[t].v, c, 12);
			s.tris.push_back({a[0], a[1], a[2], (uint32_t)(v0 + ix[0]), b[0], b[1], b[2], (uint32_t)(v0 + ix[1]),
			                  c[0], c[1], c[2], (uint32_t)(v0 + ix[2])});
			// core::vertex = position(3) tex_coord(2) normal(3) tangent(3)
			s.hitrec.push_back({{a[0], a[1], a[2]}, a[3], {b[0], b[1], b[2]}, b[3], {c[0], c[1], c[2]}, c[3],
			                    {a[5], a[6], a[7]}, a[4], {b[5], b[6], b[7]}, b[4], {c[5], c[6], c[7]}, c[4],
			                    {a[8], a[9], a[10]}, 0.f, {b[8], b[9], b[10]}, 0.f, {c[8], c[9], c[10]}, 0.f});
			const float e1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, e2[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
			TriIsect rec{e2[1], e1[2], e2[2], e1[1], e1[0], e2[0], a[1], a[2], a[0], e1[1] * e2[2] - e2[1] * e1[2], 0.f, 0.f};
			const uint32_t gid = (uint32_t)(t0 + t);   // every record carries its global triangle id (what a hit reports)
			memcpy(&rec.p0, &gid, 4);
			s.tri_isect.push_back(rec);
		}
		{   // a test accepts beta, gamma in [-eps, 1 + eps] (triangle.cpp:160-183): the accepted point (1-b-g) a + b B + g C lies within
			// 2 eps max_edge of the triangle; 4 eps max_edge (and the eps the box is already padded with) bounds it with room to spare
			float max_edge = 0;
			for (int32_t t = 0; t < nt; t++)
				for (int k = 0; k < 3; k++) {
					const float* p = k == 0 ? pa[t].v : (k == 1 ? pb[t].v : pc[t].v);
					const float* q = k == 0 ? pb[t].v : (k == 1 ? pc[t].v : pa[t].v);
					max_edge = std::max(max_edge, std::sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2])));
				}
			const float pad = 4 * kEps * max_edge + kEps;
			for (int k = 0; k < 3; k++) { sr.pbmin[k] = sr.bmin[k] - pad; sr.pbmax[k] = sr.bmax[k] + pad; }
		}
	}
	{
		unsigned n_threads = std::thread::hardware_concurrency();
		if (const char* e = getenv("PTX_BUILD_THREADS")) n_threads = (unsigned)std::max(1, atoi(e));
		n_threads = std::max(1u, std::min(n_threads, 32u));
		int par_levels = 0;   // 2^par_levels subtree tasks per surface when surfaces alone do not fill the threads
		while (n_surf && (n_surf << par_levels) < n_threads && par_levels < 5) par_levels++;
		std::atomic<size_t> next_surface{0};
		std::mutex err_mu;
		std::exception_ptr first_err;   // an exception inside a pool thread would terminate the process: keep the first, rethrow after the joins
		auto worker = [&] {
			try {
				for (size_t si = next_surface++; si < n_surf; si = next_surface++) {
					const int32_t nt = s.surf_range[8 * si + 3];
					sb[si].tree.reset(new MeshBuilder{sb[si].pa, sb[si].pb, sb[si].pc, {}, 0});
					std::vector<uint32_t> all(nt);
					for (int32_t t = 0; t < nt; t++) all[t] = (uint32_t)t;
					sb[si].tree->build(sb[si].box, std::move(all), 25, 1, par_levels);  // mesh.hpp:34: max_depth = 25
				}
			} catch (...) {
				std::lock_guard<std::mutex> lk(err_mu);
				if (!first_err) first_err = std::current_exception();
				next_surface = n_surf;   // the other workers stop at their next fetch
			}
		};
		std::vector<std::thread> pool;
		try {
			for (unsigned k = 1; k < std::min<size_t>(n_threads, n_surf); k++) pool.emplace_back(worker);
		} catch (...) {   // thread creation failed: the ones that exist still run to the end
			std::lock_guard<std::mutex> lk(err_mu);
			if (!first_err) first_err = std::current_exception();
		}
		worker();
		for (std::thread& t : pool) t.join();
		if (first_err) std::rethrow_exception(first_err);
	}
	for (size_t si = 0; si < n_surf; si++) {
		int32_t* rg = &s.surf_range[8 * si];
		const int32_t t0 = rg[2];
		SurfaceRec& sr = s.surfaces[si];
		MeshBuilder& mbuild = *sb[si].tree;
		s.kd_max_depth = std::max(s.kd_max_depth, mbuild.max_depth_seen);

		// Emission order. Children of a branch are always adjacent (the node stores the index of the first one).
		// Chunked order: a node's children, grandchildren and great-grandchildren (<= 14 nodes = 112 bytes) are
		// written together, then each great-grandchild's own chunk follows depth-first — three tree levels per
		// 128-byte cache line instead of one, and neighbouring subtrees (and their leaf lists) stay neighbours in
		// memory. PTX_KD_LAYOUT=bfs selects plain breadth-first order (measurement only).
		const uint32_t node0 = (uint32_t)s.kd_nodes.size(), ref0 = (uint32_t)s.kd_refs.size();
		static const bool bfs_layout = [] { const char* e = getenv("PTX_KD_LAYOUT"); return e && !strcmp(e, "bfs"); }();
		const int levels_per_chunk = bfs_layout ? (1 << 30) : 3;
		s.kd_nodes.push_back({0, 0});
		std::vector<std::pair<int32_t, uint32_t>> pending{{0, node0}};   // chunk roots (build node, flat index), depth-first
		std::vector<std::pair<int32_t, uint32_t>> frontier, next;
		while (!pending.empty()) {
			frontier.assign(1, pending.back());
			pending.pop_back();
			for (int level = 0; !frontier.empty(); level++) {
				if (level == levels_per_chunk) {
					// later chunks are taken from the back: push in reverse so that the leftmost subtree comes next
					for (size_t k = frontier.size(); k-- > 0;) pending.push_back(frontier[k]);
					break;
				}
				next.clear();
				for (auto [bi, fi] : frontier) {
					const BuildNode& bn = mbuild.nodes[bi];
					if (bn.leaf) {
						s.kd_nodes[fi] = kd_make_leaf((uint32_t)s.kd_refs.size(), (uint32_t)bn.ids.size());
						for (uint32_t t : bn.ids) s.kd_refs.push_back((uint32_t)t0 + t);
					} else {
						bool hl = bn.left >= 0, hr = bn.right >= 0;
						uint32_t first = (uint32_t)s.kd_nodes.size();
						s.kd_nodes[fi] = kd_make_branch(bn.split, bn.axis, hl, hr, first);
						if (hl) { s.kd_nodes.push_back({0, 0}); next.push_back({bn.left, first}); }
						if (hr) { s.kd_nodes.push_back({0, 0}); next.push_back({bn.right, first + (hl ? 1u : 0u)}); }
					}
				}
				frontier.swap(next);
			}
		}
		sr.kd_root = node0;
		rg[4] = (int32_t)node0; rg[5] = (int32_t)(s.kd_nodes.size() - node0);
		rg[6] = (int32_t)ref0;  rg[7] = (int32_t)(s.kd_refs.size() - ref0);
		sb[si] = SurfaceBuild{};   // the build tree of this surface is no longer needed

		const float* m = &s.materials_raw[11 * si];
		MaterialRec& mr = s.materials[si];
		mr.albedo[0] = m[0]; mr.albedo[1] = m[1]; mr.albedo[2] = m[2];
		mr.opacity = m[3]; mr.roughness = m[4]; mr.metallic = m[5];
		for (int k = 0; k < 3; k++) mr.emissive[k] = m[6 + k];
		mr.ior = m[9];
		mr.shadow_catcher = m[10] != 0 ? 1u : 0u;
		mr.tex_mask = 0;
		for (int k = 0; k < 7; k++) {
			mr.tex[k] = s.surf_tex.size() >= 7 * (si + 1) ? s.surf_tex[7 * si + k] : -1;
			if (mr.tex[k] >= 0) mr.tex_mask |= 1u << k;
		}
		mr.pad = 0;
		if (mr.tex_mask) s.any_texture = true;
		if (mr.tex[2] >= 0) s.any_alpha = true;   // opacity comes from a texture: the pass-through branch is reachable
		// math::is_approx(opacity, 1) (renderer.cpp:466) false, or a shadow catcher => pass-through code is needed
		if (!(mr.opacity == 1.0f || std::fabs(mr.opacity - 1.0f) < kEps) || mr.shadow_catcher) s.any_alpha = true;
	}

	for (size_t mi = 0; mi < n_models; mi++) {
		ModelRec& mr = s.models[mi];
		const float* x = &s.model_xform[12 * mi];
		memcpy(mr.origin, x, 12);
		memcpy(mr.basis, x + 3, 36);
		mat_inverse(mr.basis, mr.inv_basis);
		const float neg[3] = {-x[0], -x[1], -x[2]};
		mat_mul_vec(mr.inv_basis, neg, mr.inv_origin);
		// normal matrix = transpose(inverse(basis)): column k of the transpose is row k of the inverse
		for (int c = 0; c < 3; c++)
			for (int r = 0; r < 3; r++) mr.nmat[3 * c + r] = mr.inv_basis[3 * r + c];
		mr.first_surface = s.model_surf[2 * mi];
		mr.n_surfaces = s.model_surf[2 * mi + 1];
		for (int32_t k = 0; k < mr.n_surfaces; k++) s.surfaces[mr.first_surface + k].model = (uint32_t)mi;
		Box b;
		box_reset(b);
		for (int32_t k = 0; k < mr.n_surfaces; k++) {
			box_grow(b, s.surfaces[mr.first_surface + k].bmin);
			box_grow(b, s.surfaces[mr.first_surface + k].bmax);
		}
		memcpy(mr.bmin, b.lo, 12);
		memcpy(mr.bmax, b.hi, 12);
		Box pbx;   // union of the surfaces' slack-grown boxes
		box_reset(pbx);
		for (int32_t k = 0; k < mr.n_surfaces; k++) {
			box_grow(pbx, s.surfaces[mr.first_surface + k].pbmin);
			box_grow(pbx, s.surfaces[mr.first_surface + k].pbmax);
		}
		memcpy(mr.pbmin, pbx.lo, 12);
		memcpy(mr.pbmax, pbx.hi, 12);
	}
	// distinct ray spaces (bitwise-equal inverse transforms)
	s.spaces.clear();
	s.model_space.assign(n_models, 0);
	for (size_t mi = 0; mi < n_models; mi++) {
		SpaceRec sp;
		memcpy(sp.inv_basis, s.models[mi].inv_basis, 36);
		memcpy(sp.inv_origin, s.models[mi].inv_origin, 12);
		size_t k = 0;
		for (; k < s.spaces.size(); k++) if (!memcmp(&s.spaces[k], &sp, sizeof sp)) break;
		if (k == s.spaces.size()) s.spaces.push_back(sp);
		s.model_space[mi] = (uint32_t)k;
	}
	s.shade.assign(n_surf, ShadeRec{});
	for (size_t mi = 0; mi < n_models; mi++)
		for (int32_t k = 0; k < s.models[mi].n_surfaces; k++) {
			ShadeRec& r = s.shade[s.models[mi].first_surface + k];
			memcpy(r.basis, s.models[mi].basis, 36);
			memcpy(r.origin, s.models[mi].origin, 12);
			memcpy(r.nmat, s.models[mi].nmat, 36);
			r.mat = s.materials[s.models[mi].first_surface + k];
		}

	memcpy(s.camera.origin, cam, 12);
	memcpy(s.camera.basis, cam + 3, 36);
	s.camera.fov = cam[12];
	s.camera.tan_half_fov = std::tan(cam[12] * 0.5F);  // camera::set_fov, LIB/scene/camera.cpp:27-30
	s.sun = SunRec{};
	if (sun) {
		memcpy(s.sun.basis, sun, 36);
		memcpy(s.sun.energy, sun + 9, 12);
		s.sun.angular_radius = sun[12];
		s.sun.present = 1;
	}
}

void plan_residency(FlatScene& s, size_t lds_budget) {
	const size_t n_surf = s.surfaces.size();
	s.res_nodes.clear(); s.res_refs.clear(); s.res_tris.clear();
	s.n_resident = 0;
	for (auto& sr : s.surfaces) sr.lds_root = 0xFFFFFFFFu;
	const size_t shade_bytes = s.shade.size() * sizeof(ShadeRec);
	auto bytes_of = [&](size_t si) {
		const int32_t* rg = &s.surf_range[8 * si];
		return (size_t)rg[5] * 8 + (size_t)rg[7] * 4 + (size_t)rg[3] * 48;
	};
	std::vector<size_t> order(n_surf);
	for (size_t i = 0; i < n_surf; i++) order[i] = i;
	std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bytes_of(a) < bytes_of(b); });
	std::vector<char> take(n_surf, 0);
	size_t used = shade_bytes + 48;   // 3 x 16 bytes: each region is padded to a multiple of 16
	for (size_t si : order) {
		if (used + bytes_of(si) > lds_budget) break;
		used += bytes_of(si);
		take[si] = 1;
	}
	for (size_t si = 0; si < n_surf; si++) {   // original surface order: with everything resident the copy is the identity
		if (!take[si]) continue;
		const int32_t* rg = &s.surf_range[8 * si];
		const uint32_t t0 = (uint32_t)rg[2], nt = (uint32_t)rg[3], node0 = (uint32_t)rg[4], nn = (uint32_t)rg[5], ref0 = (uint32_t)rg[6], nr = (uint32_t)rg[7];
		const uint32_t nb = (uint32_t)s.res_nodes.size(), rb = (uint32_t)s.res_refs.size(), tb = (uint32_t)s.res_tris.size();
		for (uint32_t k = 0; k < nn; k++) {
			KdNode nd = s.kd_nodes[node0 + k];
			if ((nd.w1 & 3u) == KD_LEAF) nd.w0 = nd.w0 - ref0 + rb;                         // first ref
			else nd.w1 = (nd.w1 & 15u) | ((((nd.w1 >> 4) - node0) + nb) << 4);              // first child
			s.res_nodes.push_back(nd);
		}
		for (uint32_t k = 0; k < nr; k++) s.res_refs.push_back(s.kd_refs[ref0 + k] - t0 + tb);
		for (uint32_t k = 0; k < nt; k++) s.res_tris.push_back(s.tri_isect[t0 + k]);
		s.surfaces[si].lds_root = (s.surfaces[si].kd_root - node0) + nb;
		s.n_resident++;
	}
	s.res_bytes = s.res_tris.size() * 48 + shade_bytes + ((s.res_nodes.size() * 8 + 15) & ~(size_t)15) + ((s.res_refs.size() * 4 + 15) & ~(size_t)15);
}

}  // namespace ptx
