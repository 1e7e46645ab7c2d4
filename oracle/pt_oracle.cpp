// =====================================================================================
// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU restatement ("oracle") of the reference's ray-intersection + Monte-Carlo shading
// path (path-tracer-core/path_tracer_lib). Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product (libptx_hip.so) never does.
//
// Parity status: PINNED. Every function below is checked bit-for-bit (or to the stated
// ulp bound for libm calls) against vectors produced by the unmodified reference compiled
// in oracle/_ref (tests/golden/*.npz, generator: oracle/make_golden.py + ref_harness.cpp).
//
// Each function cites the reference file:line it restates (paths relative to
// /root/reference/path-tracer-core/path_tracer_lib/path_tracer/). Arithmetic is IEEE
// binary32 with the reference's exact operation order; the double-precision islands of
// the reference (math::pi / math::sqrt3 are double, math::pow(float,int) promotes) are
// kept as double. Build with -ffp-contract=off (the reference is baseline x86-64: no FMA).
//
// The one deliberate departure: core::rand() (core/utils.hpp:8-13, thread_local mt19937
// seeded from random_device => not reproducible, not parallelisable) is replaced by a
// counter-based Philox4x32-10 stream keyed by (pixel, sample, depth, pass-through, block).
// The draw ORDER per path vertex follows renderer.cpp:363,466,492,500,572.
// =====================================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <unordered_map>
#include <thread>
#include <tuple>
#include <vector>

namespace {

// ---------------------------------------------------------------- math/ (vec3.inl, mat3.inl, math.inl)
constexpr float EPS = 0.0001f;                       // math.hpp:16
constexpr double PI = 3.141592653589793238462643;    // math.hpp:18 (std::numbers::pi, double)
constexpr double SQRT3 = 1.732050807568877293527446; // math.hpp:20

struct v3 { float x, y, z; };
inline v3 V(float a, float b, float c) { return {a, b, c}; }
inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline v3 operator/(v3 a, v3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }   // vec3.inl "Vector + Scalar"
inline v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline v3 operator*(float s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }   // "Scalar + Vector"
inline v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
inline float get(const v3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
inline void set(v3& v, int i, float f) { (i == 0 ? v.x : (i == 1 ? v.y : v.z)) = f; }

inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              // vec3.inl:236
inline v3 cross(v3 l, v3 r) {                                                             // vec3.inl:222
	return {(l.y * r.z) - (l.z * r.y), (l.z * r.x) - (l.x * r.z), (l.x * r.y) - (l.y * r.x)};
}
inline float length(v3 a) { return std::sqrt(dot(a, a)); }                               // vec3.inl:246
inline v3 normalize(v3 a) { return a * (1 / length(a)); }                                 // vec3.inl:251
inline float fmax2(float a, float b) { return b > a ? b : a; }                            // math.inl:169 (NaN-asymmetric)
inline float fmin2(float a, float b) { return b < a ? b : a; }                            // math.inl:179
inline float lerp(float a, float b, float w) { return a + (b - a) * w; }                  // math.inl:164
inline float clampf(float x, float lo, float hi) { return fmin2(fmax2(x, lo), hi); }      // math.inl:154
inline v3 vmin(v3 a, v3 b) { return {fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)}; }
inline v3 vmax(v3 a, v3 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)}; }
inline v3 vlerp(v3 a, v3 b, float w) { return {lerp(a.x, b.x, w), lerp(a.y, b.y, w), lerp(a.z, b.z, w)}; }
inline v3 vlerp(v3 a, v3 b, v3 w) { return {lerp(a.x, b.x, w.x), lerp(a.y, b.y, w.y), lerp(a.z, b.z, w.z)}; }
inline v3 reflect(v3 incident, v3 normal) { return incident - 2 * dot(normal, incident) * normal; }  // utils.hpp:38, vec3.inl:261
// math::pow(float, int): std::pow promotes both to double (math.inl:10, <cmath> overload rules)
inline float pow_fi(float x, int p) { return (float)std::pow((double)x, (double)p); }
inline bool is_approx(float a, float b) { return a == b || std::fabs(a - b) < EPS; }      // math.inl:49

struct m3 { v3 x, y, z; };  // column vectors (mat3.hpp)
inline m3 transpose(const m3& m) { return {{m.x.x, m.y.x, m.z.x}, {m.x.y, m.y.y, m.z.y}, {m.x.z, m.y.z, m.z.z}}; }
inline v3 operator*(const m3& m, v3 v) {   // mat3.inl:219-224: dot of each ROW with v
	m3 t = transpose(m);
	return {dot(t.x, v), dot(t.y, v), dot(t.z, v)};
}
inline m3 operator*(const m3& a, const m3& b) {  // mat3.inl:144-152
	return {a.x * b.x.x + a.y * b.x.y + a.z * b.x.z, a.x * b.y.x + a.y * b.y.y + a.z * b.y.z,
	        a.x * b.z.x + a.y * b.z.y + a.z * b.z.z};
}
inline m3 inverse(const m3& mat) {  // mat3.inl:245-263
	float det1 = +(mat.y.y * mat.z.z - mat.z.y * mat.y.z);
	float det2 = -(mat.x.y * mat.z.z - mat.z.y * mat.x.z);
	float det3 = +(mat.x.y * mat.y.z - mat.y.y * mat.x.z);
	float det = mat.x.x * det1 + mat.y.x * det2 + mat.z.x * det3;
	float s = 1 / det;
	m3 r = {{det1, det2, det3},
	        {-(mat.y.x * mat.z.z - mat.z.x * mat.y.z), +(mat.x.x * mat.z.z - mat.z.x * mat.x.z),
	         -(mat.x.x * mat.y.z - mat.y.x * mat.x.z)},
	        {+(mat.y.x * mat.z.y - mat.z.x * mat.y.y), -(mat.x.x * mat.z.y - mat.z.x * mat.x.y),
	         +(mat.x.x * mat.y.y - mat.y.x * mat.x.y)}};
	return {r.x * s, r.y * s, r.z * s};
}

struct xform { v3 origin; m3 basis; };                          // scene/transform.hpp
inline v3 apply(const xform& t, v3 v) { return t.basis * v + t.origin; }       // transform.cpp:116-118
inline xform xinverse(const xform& t) {                                        // transform.cpp:33-36
	m3 b = inverse(t.basis);
	return {b * -t.origin, b};
}

struct ray { v3 o, d; };
inline ray make_ray(v3 o, v3 d) { return {o, normalize(d)}; }                  // geometry/ray.cpp:6-8
inline ray xray(const ray& r, const xform& t) { return make_ray(apply(t, r.o), t.basis * r.d); }  // ray.cpp:10-15

// ---------------------------------------------------------------- geometry/aabb.cpp
struct aabb { v3 mn, mx; };
inline void aabb_clear(aabb& b) {  // aabb.cpp:29-32 — NOTE max starts at FLT_MIN (smallest POSITIVE float): quirk Q2
	float hi = std::numeric_limits<float>::max(), lo = std::numeric_limits<float>::min();
	b.mn = {hi, hi, hi};
	b.mx = {lo, lo, lo};
}
inline void aabb_add(aabb& b, v3 p) { b.mn = vmin(b.mn, p); b.mx = vmax(b.mx, p); }  // aabb.cpp:19-22
inline float aabb_area(const aabb& b) {                                               // aabb.cpp:34-39
	v3 w = b.mx - b.mn;
	return (w.x * w.y + w.y * w.z + w.x * w.z) * 2;
}
struct aabb_hit { bool hit; float nr, fr; };
inline aabb_hit aabb_intersect(const aabb& b, const ray& r) {  // aabb.cpp:41-67
	if (b.mn.x > b.mx.x || b.mn.y > b.mx.y || b.mn.z > b.mx.z) return {false, 0, -1};
	v3 inv = V(1, 1, 1) / r.d;
	v3 t0 = (b.mn - r.o) * inv;
	v3 t1 = (b.mx - r.o) * inv;
	v3 n = vmin(t0, t1), f = vmax(t0, t1);
	float nr = fmax2(fmax2(n.x, n.y), n.z);   // math::max(a,b,c) = max(max(a,b),c)
	float fr = fmin2(fmin2(f.x, f.y), f.z);
	if (nr > fr) return {false, 0, -1};
	return {fr >= 0, nr, fr};                  // has_hit(): far >= 0 (aabb.cpp:8-10)
}

// ---------------------------------------------------------------- geometry/triangle.cpp:120-190
struct tri_hit { float t; v3 bary; };  // t < 0 => miss (has_hit: distance >= 0)
inline tri_hit tri_intersect(v3 a, v3 b, v3 c, const ray& r) {
	v3 mx = a - b, my = a - c, mz = r.d;
	v3 v = a - r.o;
	float c1 = my.y * mz.z - mz.y * my.z;
	float c2 = mx.y * mz.z - mz.y * mx.z;
	float c3 = mx.y * my.z - my.y * mx.z;
	float c4 = v.y * mz.z - mz.y * v.z;
	float c5 = mx.y * v.z - v.y * mx.z;
	float c6 = my.y * v.z - v.y * my.z;
	float inv_det = 1 / (mx.x * c1 - my.x * c2 + mz.x * c3);
	float beta = inv_det * (v.x * c1 - my.x * c4 - mz.x * c6);
	if (beta < 0 - EPS || beta > 1 + EPS) return {-1, {0, 0, 0}};
	float gamma = inv_det * (mx.x * c4 - v.x * c2 + mz.x * c5);
	if (gamma < 0 - EPS || gamma + beta > 1 + EPS) return {-1, {0, 0, 0}};
	float dist = inv_det * (mx.x * c6 - my.x * c5 + v.x * c3);
	float alpha = 1 - beta - gamma;
	return {dist, {alpha, beta, gamma}};
}

// ---------------------------------------------------------------- core/mesh.cpp, kd_tree.hpp
struct vertex { v3 pos; float u, v; v3 nrm, tan; };  // core/vertex.hpp:7-12

struct kd_node {
	bool leaf;
	uint8_t axis;
	float split;
	int left, right;   // child index or -1 (nullptr in the reference)
	int first, count;  // leaf: range in refs
};

struct mesh {
	std::vector<vertex> verts;
	std::vector<uint32_t> tris;  // 3 per triangle
	aabb box;
	std::vector<kd_node> nodes;  // pre-order; nodes[0] is the root
	std::vector<uint32_t> refs;
};

struct tri3 { v3 a, b, c; };

// kd_tree_builder::split_triangles — mesh.cpp:35-80
static void split_triangles(const std::vector<tri3>& tris, const std::vector<uint32_t>& idx, int axis, float split,
                            std::vector<tri3>& lt, std::vector<tri3>& rt, std::vector<uint32_t>& li,
                            std::vector<uint32_t>& ri) {
	for (size_t i = 0; i < tris.size(); i++) {
		const v3 p[3] = {tris[i].a, tris[i].b, tris[i].c};
		bool l = false, r = false;
		for (int k = 0; k < 3; k++) {
			if (get(p[k], axis) < split) l = true;
			else r = true;
		}
		if (l) { lt.push_back(tris[i]); li.push_back(idx[i]); }
		if (r) { rt.push_back(tris[i]); ri.push_back(idx[i]); }
	}
}

// kd_tree_builder::init_node_sah — mesh.cpp:131-247. Returns node index (pre-order numbering).
static int build_sah(mesh& m, const aabb& box, std::vector<tri3>&& tris, std::vector<uint32_t>&& idx, int depth) {
	int id = (int)m.nodes.size();
	auto make_leaf = [&]() {
		m.nodes.push_back({true, 0, 0.f, -1, -1, (int)m.refs.size(), (int)idx.size()});
		for (uint32_t i : idx) m.refs.push_back(i);
		return id;
	};
	if (depth == 0) return make_leaf();

	float base_cost = tris.size() * aabb_area(box);   // size_t * float -> float
	float best_cost = base_cost;
	int best_axis = 0;
	float best_split = 0;
	std::vector<std::tuple<float, bool>> bounds;
	bounds.reserve(tris.size() * 2);
	for (int axis = 0; axis < 3; axis++) {
		bounds.clear();
		for (const tri3& t : tris) {
			float a = get(t.a, axis), b = get(t.b, axis), c = get(t.c, axis);
			float start = fmin2(fmin2(a, b), c);
			float end = fmax2(fmax2(a, b), c);
			bounds.emplace_back(start, true);
			bounds.emplace_back(end, false);
		}
		// Same container type, comparator and std::sort as mesh.cpp:162-163: the (unstable)
		// permutation of equal keys is then identical to the reference's on the same libstdc++.
		std::sort(bounds.begin(), bounds.end(), [](auto& x, auto& y) { return std::get<0>(x) < std::get<0>(y); });
		float split = 0;
		uint32_t lcount = 0;
		uint32_t rcount = (uint32_t)tris.size();
		for (size_t i = 0; i <= bounds.size(); i++) {
			if (i == 0) split = std::get<0>(bounds.front()) - EPS;
			else if (i == bounds.size()) {
				rcount--;
				split = std::get<0>(bounds.back()) + EPS;
			} else {
				auto& prev = bounds[i - 1];
				auto& next = bounds[i];
				if (std::get<1>(prev)) lcount++;
				else rcount--;
				if (std::get<0>(prev) == std::get<0>(next)) continue;
				split = (std::get<0>(prev) + std::get<0>(next)) * 0.5F;
			}
			if (split <= get(box.mn, axis)) continue;
			if (split >= get(box.mx, axis)) break;
			aabb l = box, r = box;                 // split_aabb mesh.cpp:21-33
			set(l.mx, axis, split);
			set(r.mn, axis, split);
			float cost = lcount * aabb_area(l) + rcount * aabb_area(r);  // uint32 * float -> float
			if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = split; }
		}
	}
	if (!(best_cost < base_cost)) return make_leaf();

	m.nodes.push_back({false, (uint8_t)best_axis, best_split, -1, -1, 0, 0});
	aabb l = box, r = box;
	set(l.mx, best_axis, best_split);
	set(r.mn, best_axis, best_split);
	std::vector<tri3> lt, rt;
	std::vector<uint32_t> li, ri;
	split_triangles(tris, idx, best_axis, best_split, lt, rt, li, ri);
	std::vector<tri3>().swap(tris);
	std::vector<uint32_t>().swap(idx);
	if (!lt.empty()) { int c = build_sah(m, l, std::move(lt), std::move(li), depth - 1); m.nodes[id].left = c; }
	if (!rt.empty()) { int c = build_sah(m, r, std::move(rt), std::move(ri), depth - 1); m.nodes[id].right = c; }
	return id;
}

static void mesh_finish(mesh& m) {
	// mesh::recalculate_aabb — mesh.cpp:254-261
	aabb_clear(m.box);
	for (auto& v : m.verts) aabb_add(m.box, v.pos);
	m.box.mn = m.box.mn - V(EPS, EPS, EPS);
	m.box.mx = m.box.mx + V(EPS, EPS, EPS);
	// mesh::build_kd_tree — mesh.cpp:263-298 (use_sah = true, max_depth = 25: mesh.hpp:34)
	size_t nt = m.tris.size() / 3;
	std::vector<tri3> t(nt);
	std::vector<uint32_t> idx(nt);
	for (size_t i = 0; i < nt; i++) {
		t[i] = {m.verts[m.tris[3 * i]].pos, m.verts[m.tris[3 * i + 1]].pos, m.verts[m.tris[3 * i + 2]].pos};
		idx[i] = (uint32_t)i;
	}
	m.nodes.clear();
	m.refs.clear();
	build_sah(m, m.box, std::move(t), std::move(idx), 25);
}

struct mesh_hit { float t = -1; v3 bary{0, 0, 0}; uint32_t index = 0; };

struct trav_stats { uint64_t branches = 0, leaves = 0, tris = 0, pushes = 0, mesh_tests = 0, model_tests = 0, hits = 0; uint64_t depth_hist[32] = {0}; };

// mesh::intersect — mesh.cpp:300-405
static mesh_hit mesh_intersect(const mesh& m, const ray& r, trav_stats* st) {
	if (st) st->mesh_tests++;
	aabb_hit bh = aabb_intersect(m.box, r);
	if (!bh.hit) return {};
	struct ent { int node; float mn, mx; };
	ent stack[64];
	int sp = 0;
	int max_sp = 0;   // diagnostics only: deepest simultaneous number of PENDING entries (root excluded)
	struct depth_rec { trav_stats* st; int* m; ~depth_rec() { if (st) st->depth_hist[*m < 31 ? *m : 31]++; } } rec{st, &max_sp};
	stack[sp++] = {0, bh.nr, bh.fr};
	while (sp > 0) {
		ent e = stack[--sp];
		int node = e.node;
		float min_dist = e.mn, max_dist = e.mx;
		while (node >= 0 && !m.nodes[node].leaf) {
			const kd_node& b = m.nodes[node];
			if (st) st->branches++;
			float o = get(r.o, b.axis), d = get(r.d, b.axis);
			float split_dist = (b.split - o) / d;
			int first, second;
			if (o < b.split) { first = b.left; second = b.right; }
			else { first = b.right; second = b.left; }
			if (split_dist < 0 || split_dist > max_dist) node = first;
			else if (split_dist < min_dist) node = second;
			else {
				if (second >= 0) { stack[sp++] = {second, split_dist, max_dist}; if (st) st->pushes++; if (sp > max_sp) max_sp = sp; }
				node = first;
				max_dist = split_dist;
			}
		}
		if (node < 0) continue;
		const kd_node& leaf = m.nodes[node];
		if (st) st->leaves++;
		tri_hit nearest{-1, {0, 0, 0}};
		uint32_t index = 0;
		for (int i = 0; i < leaf.count; i++) {
			uint32_t ti = m.refs[leaf.first + i];
			if (st) st->tris++;
			tri_hit h = tri_intersect(m.verts[m.tris[3 * ti]].pos, m.verts[m.tris[3 * ti + 1]].pos,
			                          m.verts[m.tris[3 * ti + 2]].pos, r);
			if (h.t >= 0 && h.t <= max_dist && (h.t < nearest.t || !(nearest.t >= 0))) {
				nearest = h;
				index = (uint32_t)i;
			}
		}
		if (!(nearest.t >= 0)) continue;
		return {nearest.t, nearest.bary, m.refs[leaf.first + index]};
	}
	return {};
}

// ---------------------------------------------------------------- image/image.cpp, image/image_texture.cpp
struct texture {   // 8-bit image as stb_image returns it (1-4 channels), + the sRGB flag it was first loaded with
	int w = 0, h = 0, c = 0;
	bool srgb = false;
	std::vector<uint8_t> data;
	std::vector<float> fdata;   // image::hdr (a Radiance .hdr, stbi_loadf): the floats as decoded; `data` is then empty
};
struct f4v { float x, y, z, w; };
// image::image::read — image.cpp:124-141: LDR byte / 255, HDR the stored float; sRGB decode pow(v, 2.2) on colour channels
static inline float tex_read(const texture& t, uint32_t px, uint32_t py, uint32_t ch) {
	uint32_t index = py * (uint32_t)t.w + px;
	index = index * (uint32_t)t.c + ch;
	float value = t.fdata.empty() ? t.data[index] / 255.0F : t.fdata[index];
	if (t.srgb && ch < 3) value = std::pow(value, 2.2F);
	return value;
}
// image_texture::read_pixel — image_texture.cpp:47-62 (missing channels stay 1)
static inline f4v tex_pixel(const texture& t, uint32_t px, uint32_t py) {
	f4v c = {1, 1, 1, 1};
	switch (t.c) {
	case 4: c.w = tex_read(t, px, py, 3); [[fallthrough]];
	case 3: c.z = tex_read(t, px, py, 2); [[fallthrough]];
	case 2: c.y = tex_read(t, px, py, 1); [[fallthrough]];
	case 1: c.x = tex_read(t, px, py, 0);
	}
	return c;
}
// float -> uint32 as the reference's build does it: `uvec2(floor(x), ..)` is an implicit float->unsigned conversion,
// which g++ on x86-64 compiles to a 64-bit cvttss2si and a truncation, so negative values wrap modulo 2^32 (quirk Q3)
static inline uint32_t f2u_wrap(float f) { return (uint32_t)(int64_t)f; }
static inline uint32_t umod(uint32_t x, uint32_t y) { return (y + (x % y)) % y; }   // math::mod, integer branch (math.inl:189-194)
// image_texture::sample — image_texture.cpp:21-45: bilinear, wrap by unsigned modulo
static f4v tex_sample(const texture& t, float u, float v) {
	const uint32_t sx = (uint32_t)t.w, sy = (uint32_t)t.h;
	const float cx = u * sx - 0.5F, cy = (1 - v) * sy - 0.5F;
	const uint32_t fx = f2u_wrap(std::floor(cx)), fy = f2u_wrap(std::floor(cy));
	const uint32_t gx = f2u_wrap(std::ceil(cx)), gy = f2u_wrap(std::ceil(cy));
	const f4v tl = tex_pixel(t, umod(fx, sx), umod(fy, sy)), tr = tex_pixel(t, umod(gx, sx), umod(fy, sy));
	const f4v bl = tex_pixel(t, umod(fx, sx), umod(gy, sy)), br = tex_pixel(t, umod(gx, sx), umod(gy, sy));
	const float dx = cx - std::floor(cx), dy = cy - std::floor(cy);   // math::fract
	const f4v tt = {lerp(tl.x, tr.x, dx), lerp(tl.y, tr.y, dx), lerp(tl.z, tr.z, dx), lerp(tl.w, tr.w, dx)};
	const f4v bb = {lerp(bl.x, br.x, dx), lerp(bl.y, br.y, dx), lerp(bl.z, br.z, dx), lerp(bl.w, br.w, dx)};
	return {lerp(tt.x, bb.x, dy), lerp(tt.y, bb.y, dy), lerp(tt.z, bb.z, dy), lerp(tt.w, bb.w, dy)};
}

// ---------------------------------------------------------------- core/material.{hpp,cpp}
enum { TEX_NORMAL = 0, TEX_ALBEDO, TEX_OPACITY, TEX_OCCLUSION, TEX_ROUGHNESS, TEX_METALLIC, TEX_EMISSIVE };
struct material {
	v3 albedo; float opacity, roughness, metallic; v3 emissive; float ior; bool shadow_catcher;
	int tex[7] = {-1, -1, -1, -1, -1, -1, -1};   // image index per slot (material.hpp:19-26), -1 = none
};
struct mat_sample { v3 normal_ts, albedo, emissive; float opacity, roughness, metallic; };
// material::get_normal / albedo / opacity / roughness / metallic / emissive — material.cpp:6-53
static mat_sample material_eval(const material& m, const std::vector<texture>& tx, float u, float v) {
	mat_sample o;
	o.normal_ts = {0, 0, 1};
	if (m.tex[TEX_NORMAL] >= 0) { f4v s = tex_sample(tx[m.tex[TEX_NORMAL]], u, v); o.normal_ts = V(s.x, s.y, s.z) * 2 - V(1, 1, 1); }
	o.albedo = m.albedo;
	if (m.tex[TEX_ALBEDO] >= 0) { f4v s = tex_sample(tx[m.tex[TEX_ALBEDO]], u, v); o.albedo = o.albedo * V(s.x, s.y, s.z); }
	o.opacity = m.opacity;
	if (m.tex[TEX_OPACITY] >= 0) o.opacity *= tex_sample(tx[m.tex[TEX_OPACITY]], u, v).w;
	o.roughness = m.roughness;
	if (m.tex[TEX_ROUGHNESS] >= 0) o.roughness *= tex_sample(tx[m.tex[TEX_ROUGHNESS]], u, v).y;
	o.metallic = m.metallic;
	if (m.tex[TEX_METALLIC] >= 0) o.metallic *= tex_sample(tx[m.tex[TEX_METALLIC]], u, v).z;
	o.emissive = m.emissive;
	if (m.tex[TEX_EMISSIVE] >= 0) { f4v s = tex_sample(tx[m.tex[TEX_EMISSIVE]], u, v); o.emissive = o.emissive * V(s.x, s.y, s.z); }
	return o;
}

// ---------------------------------------------------------------- scene/model.cpp
struct surface { mesh m; material mat; };
struct model {
	xform t, inv;            // global transform, and its inverse (recomputed per ray in model.cpp:22-25; same values)
	m3 normal_matrix;        // transpose(inverse(basis)) — renderer.cpp:698
	aabb box;
	int first_surface, n_surfaces;
};

struct scene_t {
	std::vector<model> models;      // in the order renderer::intersect visits them
	std::vector<surface> surfaces;
	std::vector<texture> textures;
	texture env;               // renderer::environment (renderer.hpp:28): set by ora_scene_set_environment
	bool has_env = false;
	xform camera; float fov, tan_half_fov;
	bool has_sun = false; m3 sun_basis; v3 sun_energy; float sun_radius;
};

struct model_hit { float dist = -1; int surface = -1; uint32_t tri = 0; v3 bary{0, 0, 0}; };

// model::intersect — model.cpp:20-72
static model_hit model_intersect(const scene_t& s, const model& md, const ray& r, trav_stats* st) {
	if (st) st->model_tests++;
	ray view = xray(r, md.inv);
	if (!aabb_intersect(md.box, view).hit) return {};
	mesh_hit nearest;
	int hit_surface = -1;
	for (int i = 0; i < md.n_surfaces; i++) {
		mesh_hit h = mesh_intersect(s.surfaces[md.first_surface + i].m, view, st);
		if (!(h.t >= 0)) continue;
		if (h.t < nearest.t || !(nearest.t >= 0)) { nearest = h; hit_surface = md.first_surface + i; }
	}
	if (!(nearest.t >= 0)) return {};
	v3 hit_vec = view.d * nearest.t;
	float dist = length(md.t.basis * hit_vec);
	return {dist, hit_surface, nearest.index, nearest.bary};
}

struct scene_hit { bool hit = false; int surface = -1; v3 pos, nrm, tan; float u = 0, v = 0; };

// renderer::intersect — renderer.cpp:645-725
static scene_hit scene_intersect(const scene_t& s, const ray& r, trav_stats* st) {
	model_hit nearest;
	const model* nm = nullptr;
	for (const model& md : s.models) {
		model_hit h = model_intersect(s, md, r, st);
		if (!(h.dist >= 0)) continue;
		if (h.dist < nearest.dist || !(nearest.dist >= 0)) { nearest = h; nm = &md; }
	}
	if (!(nearest.dist >= 0)) return {};
	if (st) st->hits++;
	const mesh& m = s.surfaces[nearest.surface].m;
	const vertex& v1 = m.verts[m.tris[3 * nearest.tri]];
	const vertex& v2 = m.verts[m.tris[3 * nearest.tri + 1]];
	const vertex& v3_ = m.verts[m.tris[3 * nearest.tri + 2]];
	v3 b = nearest.bary;
	scene_hit out;
	out.hit = true;
	out.surface = nearest.surface;
	out.pos = apply(nm->t, v1.pos * b.x + v2.pos * b.y + v3_.pos * b.z);
	out.u = v1.u * b.x + v2.u * b.y + v3_.u * b.z;
	out.v = v1.v * b.x + v2.v * b.y + v3_.v * b.z;
	out.nrm = normalize(nm->normal_matrix * (v1.nrm * b.x + v2.nrm * b.y + v3_.nrm * b.z));
	out.tan = normalize(nm->normal_matrix * (v1.tan * b.x + v2.tan * b.y + v3_.tan * b.z));
	return out;
}

// intersect_result::get_normal — renderer.cpp:430-435: TBN * material::get_normal(uv)
// (= (0,0,1) without a normal texture, 2*texel-1 with one: material.cpp:6-11)
static v3 shading_normal(const scene_hit& h, v3 normal_ts) {
	v3 binormal = cross(h.nrm, h.tan);
	m3 tbn = {h.tan, binormal, h.nrm};
	return tbn * normal_ts;
}

// ---------------------------------------------------------------- util/rand_cone_vec.cpp:8-35
static v3 rand_cone_vec(float rnd, float cos_theta, v3 normal) {
	float phi = (float)((double)(rnd * 2) * PI);
	float sin_theta = std::sqrt(1 - cos_theta * cos_theta);
	v3 cone = {std::cos(phi) * sin_theta, std::sin(phi) * sin_theta, cos_theta};
	v3 np = {0, 0, 0};
	if ((double)std::fabs(normal.x) < (1 / SQRT3)) np.x = 1;
	else if ((double)std::fabs(normal.y) < (1 / SQRT3)) np.y = 1;
	else np.z = 1;
	v3 tangent = normalize(cross(normal, np));
	v3 binormal = cross(normal, tangent);
	m3 tbn = {tangent, binormal, normal};
	return tbn * cone;
}

// ---------------------------------------------------------------- core/pbr.cpp
static float fresnel(v3 outcoming, v3 incoming, float ior) {   // pbr.cpp:13-25
	v3 halfway = normalize(outcoming + incoming);
	float cos_theta = dot(outcoming, halfway);
	float f0 = (ior - 1) / (ior + 1);
	f0 *= f0;
	return lerp(f0, 1, pow_fi(1 - cos_theta, 5));
}
static v3 importance_diffuse(float u1, float u2, v3 normal) {  // pbr.cpp:71-77
	float theta = std::acos(2 * u1 - 1) * 0.5F;
	return rand_cone_vec(u2, std::cos(theta), normal);
}
static v3 importance_specular(float u1, float u2, v3 normal, v3 outcoming, float roughness) {  // pbr.cpp:79-91
	roughness *= roughness;
	roughness *= roughness;
	float cos_theta = std::sqrt((1 - u1) / (1 + (roughness - 1) * u1));
	v3 halfway = rand_cone_vec(u2, cos_theta, normal);
	return reflect(-outcoming, halfway);
}
static float smith_g1(v3 normal, v3 dir, float k) {           // pbr.cpp:95-102
	float c = dot(normal, dir);
	return c / fmax2(lerp(k, 1, c), EPS);
}
static float geometry_smith(v3 n, v3 o, v3 i, float roughness) {  // pbr.cpp:104-114
	float r = roughness + 1;
	float k = (r * r) / 8;
	return smith_g1(n, o, k) * smith_g1(n, i, k);
}
static float pdf_diffuse(v3 normal, v3 incoming) {             // pbr.cpp:118-123
	float c = dot(normal, incoming);
	return (float)((double)c / PI);
}
static float distribution_ggx(v3 n, v3 o, v3 i, float roughness) {  // pbr.cpp:125-140
	roughness *= roughness;
	roughness *= roughness;
	v3 halfway = normalize(o + i);
	float cos_phi = dot(n, halfway);
	float denom = 1 + (roughness - 1) * (cos_phi * cos_phi);     // lerp(1, roughness, cos_phi*cos_phi)
	float cos_theta = dot(n, i);
	double d = PI * (double)denom * (double)denom;               // (pi*denom)*denom in double
	double mx = (double)EPS > d ? (double)EPS : d;               // math::max<double,float>
	return (float)((double)(cos_theta * roughness) / mx);
}
static float pdf_specular(v3 n, v3 o, v3 i, float roughness) {  // pbr.cpp:172-184
	float dist = distribution_ggx(n, o, i, roughness);
	float geo = geometry_smith(n, o, i, roughness);
	float ndo = dot(n, o), ndi = dot(n, i);
	return (dist * geo) / fmax2(4 * ndo * ndi, EPS);
}

// ---------------------------------------------------------------- scene/camera.cpp:10-21
static ray camera_ray(const scene_t& s, float ndc_x, float ndc_y, float ratio) {
	float dx = s.tan_half_fov * ndc_x, dy = s.tan_half_fov * ndc_y;
	dx *= ratio;
	ray r = make_ray(V(0, 0, 0), V(dx, dy, -1));
	return xray(r, s.camera);
}

// ---------------------------------------------------------------- RNG: Philox4x32-10 (Salmon et al., SC'11)
struct u4 { uint32_t x, y, z, w; };
static u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1) {
	for (int i = 0; i < 10; i++) {
		uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
		uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
		u4 n = {(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
		c = n;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	return c;
}
inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1), 24 bits
enum { BLOCK_SURFACE = 0, BLOCK_SUN = 1, BLOCK_JITTER = 2 };
struct path_key { uint32_t pixel, sample, seed_lo, seed_hi; };
struct f4 { float x, y, z, w; };
static f4 draws(const path_key& k, uint32_t depth, uint32_t pass, uint32_t block) {
	if (pass > 0xFFFF) pass = 0xFFFF;
	u4 r = philox4x32_10({k.pixel, k.sample, (depth << 16) | pass, block}, k.seed_lo, k.seed_hi);
	return {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
}

// Miss colour: environment_factor, times the environment map when one is set (renderer.cpp:443-449, shading_worker.cpp:28-35):
// core::equirectangular_proj (core/utils.hpp:22-27) + image_texture::sample.
static v3 miss_colour(const scene_t& s, const float* env_factor, v3 dir, float* uv_out = nullptr, f4v* tex_out = nullptr) {
	const v3 f = V(env_factor[0], env_factor[1], env_factor[2]);
	if (!s.has_env) return f;
	const float u = std::atan2(dir.z, dir.x) * 0.1591F + 0.5F;
	const float v = std::asin(dir.y) * 0.3183F + 0.5F;
	const f4v c = tex_sample(s.env, u, v);
	if (uv_out) { uv_out[0] = u; uv_out[1] = v; }
	if (tex_out) *tex_out = c;
	return V(c.x, c.y, c.z) * f;
}

// ---------------------------------------------------------------- renderer::trace — renderer.cpp:437-643
struct render_cfg {
	uint32_t W, H, spp, bounces;
	float env[3];
	uint32_t seed_lo, seed_hi;
	uint32_t x0, y0, w, h;      // tile
	uint32_t sample0;           // first sample index
	uint32_t integrator;        // 0 = renderer::trace (LIB), 1 = the HOST worker's stage pipeline (trace_worker below)
};
// RNG of the reference, for pinning trace() bit-exactly against the compiled reference run single-threaded with a fixed seed
// (oracle/ref_harness.cpp `trace`): core::rand() (core/utils.hpp:8-13) = one thread_local std::mt19937 seeded once from
// std::random_device, drawn through std::uniform_real_distribution<float>(0, 1) — the same libstdc++ classes here, consumed in
// the order trace() consumes them (renderer.cpp:466, 492, 500, 572). Where two rand() calls are arguments of ONE call
// (renderer.cpp:500: rand_cone_vec(rand(), cos(rand() * r), ..); :572: fvec2(rand(), rand())) C++ leaves the order to the
// compiler; `args_rtl` says which one the reference's g++ build took (settled empirically by tests/test_oracle_vs_reference.py:
// only one setting reproduces the reference bit for bit).
struct mt_stream {
	std::mt19937 rng;
	std::uniform_real_distribution<float> dist{0, 1};
	bool args_rtl = true;
	uint64_t n_draws = 0;
	explicit mt_stream(uint32_t seed, bool rtl) : rng(seed), args_rtl(rtl) {}
	float next() { n_draws++; return dist(rng); }
	// the two draws of one call's argument list: first = the value of the FIRST argument, second = of the second
	void pair(float& first, float& second) {
		if (args_rtl) { second = next(); first = next(); }
		else { first = next(); second = next(); }
	}
};
struct trace_ctx { const scene_t* s; const render_cfg* cfg; uint64_t rays = 0; trav_stats* st = nullptr; mt_stream* mt = nullptr; };

static v3 trace(trace_ctx& c, const path_key& key, uint32_t bounce, const ray& r, uint32_t pass) {
	if (bounce == 0) return V(0, 0, 0);                      // fvec4::future = (0,0,0,1)
	c.rays++;
	scene_hit res = scene_intersect(*c.s, r, c.st);
	if (!res.hit) return miss_colour(*c.s, c.cfg->env, r.d);                 // renderer.cpp:443-451
	const material& mt = c.s->surfaces[res.surface].mat;
	const mat_sample ms = material_eval(mt, c.s->textures, res.u, res.v);   // renderer.cpp:458-462
	v3 albedo = ms.albedo;
	float opacity = ms.opacity, roughness = ms.roughness, metallic = ms.metallic;
	v3 emissive = ms.emissive * 10;                          // renderer.cpp:462
	float ior = mt.ior;
	uint32_t depth = c.cfg->bounces - bounce;
	// counter-based stream (product and oracle share it): x: opacity, y: lobe, z,w: BSDF sample. With c.mt set the draws are
	// taken from the reference's sequential stream instead, at the points where trace() calls rand().
	f4 rnd = c.mt ? f4{0, 0, 0, 0} : draws(key, depth, pass, BLOCK_SURFACE);

	if (!is_approx(opacity, 1) && (c.mt ? c.mt->next() : rnd.x) > opacity)   // renderer.cpp:466-472 (rand() only when opacity != 1)
		return trace(c, key, bounce, make_ray(res.pos + r.d * EPS, r.d), pass + 1);

	v3 normal = shading_normal(res, ms.normal_ts);
	v3 outcoming = -r.d;
	if (dot(normal, outcoming) <= 0) return V(0, 0, 0);      // renderer.cpp:478-479

	roughness = fmax2(roughness, 0.05F);
	float specular_probability = fresnel(outcoming, reflect(-outcoming, normal), ior);
	specular_probability = fmax2(specular_probability, metallic);
	bool specular_sample = (c.mt ? c.mt->next() : rnd.y) < specular_probability;   // renderer.cpp:492

	v3 direct_out = V(0, 0, 0);
	if (c.s->has_sun) {                                      // renderer.cpp:498-564
		f4 sr = c.mt ? f4{0, 0, 0, 0} : draws(key, depth, pass, BLOCK_SUN);   // x: azimuth draw, y: cone-angle draw
		if (c.mt) c.mt->pair(sr.x, sr.y);                      // renderer.cpp:500: rand_cone_vec(rand(), cos(rand() * radius), ..)
		v3 direct_incoming = c.s->sun_basis * V(0, 0, 1);
		direct_incoming = rand_cone_vec(sr.x, std::cos(sr.y * c.s->sun_radius), direct_incoming);
		if (dot(normal, direct_incoming) > 0) {
			ray direct_ray = make_ray(res.pos + direct_incoming * EPS, direct_incoming);
			c.rays++;
			scene_hit dres = scene_intersect(*c.s, direct_ray, c.st);
			if (!dres.hit) {
				if (mt.shadow_catcher && bounce == c.cfg->bounces)
					return trace(c, key, bounce, make_ray(res.pos + r.d * EPS, r.d), pass + 1);
				float diffuse_pdf = pdf_diffuse(normal, direct_incoming);
				v3 diffuse_brdf = diffuse_pdf * albedo;
				float specular_pdf = pdf_specular(normal, outcoming, direct_incoming, roughness);
				v3 specular_brdf = V(specular_pdf, specular_pdf, specular_pdf);
				v3 fr = vlerp(V(0.04F, 0.04F, 0.04F), albedo, metallic);
				{
					v3 halfway = normalize(outcoming + direct_incoming);
					float cos_theta = dot(outcoming, halfway);
					fr = vlerp(fr, V(1, 1, 1), pow_fi(1 - cos_theta, 5));
				}
				diffuse_brdf = vlerp(diffuse_brdf, V(0, 0, 0), metallic);
				v3 brdf = vlerp(diffuse_brdf, specular_brdf, fr);
				float pdf = lerp(1, 1, specular_probability);
				v3 direct_in = c.s->sun_energy;
				direct_out = brdf * direct_in / fmax2(pdf, EPS);
				direct_out = {clampf(direct_out.x, 0, direct_in.x), clampf(direct_out.y, 0, direct_in.y),
				              clampf(direct_out.z, 0, direct_in.z)};
			} else if (mt.shadow_catcher && bounce == c.cfg->bounces) return V(0, 0, 0);
		}
	}

	v3 indirect_out = V(0, 0, 0);
	if (c.mt) c.mt->pair(rnd.z, rnd.w);                      // renderer.cpp:572: fvec2 rand(core::rand(), core::rand())
	v3 indirect_incoming = specular_sample ? importance_specular(rnd.z, rnd.w, normal, outcoming, roughness)
	                                       : importance_diffuse(rnd.z, rnd.w, normal);
	if (dot(normal, indirect_incoming) > 0) {                // renderer.cpp:578-621
		float diffuse_pdf = pdf_diffuse(normal, indirect_incoming);
		v3 diffuse_brdf = diffuse_pdf * albedo;
		float specular_pdf = pdf_specular(normal, outcoming, indirect_incoming, roughness);
		v3 specular_brdf = V(specular_pdf, specular_pdf, specular_pdf);
		v3 fr = vlerp(V(0.04F, 0.04F, 0.04F), albedo, metallic);
		{
			v3 halfway = normalize(outcoming + indirect_incoming);
			float cos_theta = dot(outcoming, halfway);
			fr = vlerp(fr, V(1, 1, 1), pow_fi(1 - cos_theta, 5));
		}
		diffuse_brdf = vlerp(diffuse_brdf, V(0, 0, 0), metallic);
		v3 brdf = vlerp(diffuse_brdf, specular_brdf, fr);
		float pdf = lerp(diffuse_pdf, specular_pdf, specular_probability);
		ray indirect_ray = make_ray(res.pos + indirect_incoming * EPS, indirect_incoming);
		v3 indirect_in = trace(c, key, bounce - 1, indirect_ray, 0);
		indirect_out = brdf * indirect_in / fmax2(pdf, EPS);
		indirect_out = {clampf(indirect_out.x, 0, indirect_in.x), clampf(indirect_out.y, 0, indirect_in.y),
		                clampf(indirect_out.z, 0, indirect_in.z)};
	}
	return direct_out + indirect_out + emissive;
}

// ---------------------------------------------------------------- the HOST worker's integrator ("parity unpinned")
// src/processors/worker/{worker,intersection_worker,shading_worker,accumulation_worker}.cpp run one camera sample as a
// message passed between stage queues: INTERSECT (closest hit + sun sample, intersection_worker.cpp:10-47) ->
// DIRECT_LIGHTING (shadow query, :49-67) -> SHADING (shading_worker.cpp:10-201) -> ... -> ACCUMULATE. With
// num_workers = 1 the two merge stages (intersection_worker.cpp:69-147) forward their input unchanged, so a sample is
// the loop below. HOST links the AWS SDK and cannot be built here: this restatement is checked against the text only.
// Differences from renderer::trace: emissive is added before the opacity test; throughput is clamped to [0,10], not
// [0,incoming]; Russian roulette below bounce_count-2; a shadow catcher without a lit sun sample turns the sample black.
static v3 trace_worker(trace_ctx& c, const path_key& key, ray r) {
	const uint32_t B = c.cfg->bounces;
	v3 color = V(0, 0, 0), scale = V(1, 1, 1);               // cloud_ray::color / ::scale (worker.cpp:140-141)
	uint32_t bounce = B, pass = 0;                           // cloud_ray::bounce (worker.cpp:142)
	while (bounce > 0) {                                     // shading_worker.cpp:193 / worker.cpp:143
		const uint32_t depth = B - bounce;
		c.rays++;
		scene_hit res = scene_intersect(*c.s, r, c.st);      // intersect_min_result == intersect on one worker
		if (!res.hit) {                                      // shading_worker.cpp:28-41
			color = color + scale * miss_colour(*c.s, c.cfg->env, r.d);
			break;
		}
		const material& mt = c.s->surfaces[res.surface].mat;
		const mat_sample ms = material_eval(mt, c.s->textures, res.u, res.v);
		v3 albedo = ms.albedo;
		float opacity = ms.opacity, roughness = ms.roughness, metallic = ms.metallic;
		v3 emissive = ms.emissive * 10;
		float ior = mt.ior;
		f4 rnd = draws(key, depth, pass, BLOCK_SURFACE);     // x: opacity, y: lobe, z,w: BSDF sample
		f4 sr = draws(key, depth, pass, BLOCK_SUN);          // x: azimuth, y: cone angle, z: Russian roulette
		v3 normal = shading_normal(res, ms.normal_ts);       // intersect_min_result's normal == result.get_normal()

		// INTERSECT stage, intersection_worker.cpp:22-39: the sun sample of this vertex
		bool have_direct = false, direct_hit = false;
		v3 direct_incoming = V(0, 0, 0);
		if (c.s->has_sun) {
			v3 din = c.s->sun_basis * V(0, 0, 1);
			din = rand_cone_vec(sr.x, std::cos(sr.y * c.s->sun_radius), din);
			ray direct_ray = make_ray(res.pos + din * EPS, din);
			if (dot(normal, din) > 0) {
				have_direct = true;
				direct_incoming = direct_ray.d;                  // ray::get_dir(): normalised
				// the DIRECT_LIGHTING stage (:58-64) runs for every such ray; the result is only read further down
			}
		}

		color = color + scale * emissive;                    // shading_worker.cpp:52

		if (!is_approx(opacity, 1) && rnd.x > opacity) {     // :54-63 — back to INTERSECT, bounce unchanged
			r = make_ray(res.pos + r.d * EPS, r.d);
			pass++;
			continue;
		}
		v3 outcoming = -r.d;
		if (dot(normal, outcoming) <= 0) break;              // :68-72
		if (have_direct) {
			c.rays++;
			direct_hit = scene_intersect(*c.s, make_ray(res.pos + direct_incoming * EPS, direct_incoming), c.st).hit;
		}
		const bool lit = have_direct && dot(normal, direct_incoming) > 0 && !direct_hit;   // :77-87, :112-118
		if (mt.shadow_catcher && bounce == B) {              // :74-105
			if (!lit) { color = V(0, 0, 0); break; }
			r = make_ray(res.pos + r.d * EPS, r.d);
			pass++;
			continue;
		}
		roughness = fmax2(roughness, 0.05F);
		float specular_probability = fresnel(outcoming, reflect(-outcoming, normal), ior);
		specular_probability = fmax2(specular_probability, metallic);
		bool specular_sample = rnd.y < specular_probability;
		if (lit) {                                           // :119-143
			float diffuse_pdf = pdf_diffuse(normal, direct_incoming);
			v3 diffuse_brdf = diffuse_pdf * albedo;
			float specular_pdf = pdf_specular(normal, outcoming, direct_incoming, roughness);
			v3 specular_brdf = V(specular_pdf, specular_pdf, specular_pdf);
			v3 fr = vlerp(V(0.04F, 0.04F, 0.04F), albedo, metallic);
			{
				v3 halfway = normalize(outcoming + direct_incoming);
				float cos_theta = dot(outcoming, halfway);
				fr = vlerp(fr, V(1, 1, 1), pow_fi(1 - cos_theta, 5));
			}
			diffuse_brdf = vlerp(diffuse_brdf, V(0, 0, 0), metallic);
			v3 brdf = vlerp(diffuse_brdf, specular_brdf, fr);
			float pdf = lerp(1, 1, specular_probability);
			v3 direct_in = c.s->sun_energy;
			v3 direct_out = brdf * direct_in / fmax2(pdf, EPS);
			direct_out = {clampf(direct_out.x, 0, direct_in.x), clampf(direct_out.y, 0, direct_in.y),
			              clampf(direct_out.z, 0, direct_in.z)};
			color = color + scale * direct_out;
		}
		v3 indirect_incoming = specular_sample ? importance_specular(rnd.z, rnd.w, normal, outcoming, roughness)
		                                       : importance_diffuse(rnd.z, rnd.w, normal);
		if (!(dot(normal, indirect_incoming) > 0)) break;    // :154, :196-199
		float diffuse_pdf = pdf_diffuse(normal, indirect_incoming);
		v3 diffuse_brdf = diffuse_pdf * albedo;
		float specular_pdf = pdf_specular(normal, outcoming, indirect_incoming, roughness);
		v3 specular_brdf = V(specular_pdf, specular_pdf, specular_pdf);
		v3 fr = vlerp(V(0.04F, 0.04F, 0.04F), albedo, metallic);
		{
			v3 halfway = normalize(outcoming + indirect_incoming);
			float cos_theta = dot(outcoming, halfway);
			fr = vlerp(fr, V(1, 1, 1), pow_fi(1 - cos_theta, 5));
		}
		diffuse_brdf = vlerp(diffuse_brdf, V(0, 0, 0), metallic);
		v3 brdf = vlerp(diffuse_brdf, specular_brdf, fr);
		float pdf = lerp(diffuse_pdf, specular_pdf, specular_probability);
		scale = scale * (brdf / fmax2(pdf, EPS));            // :173
		scale = {clampf(scale.x, 0, 10.0f), clampf(scale.y, 0, 10.0f), clampf(scale.z, 0, 10.0f)};   // :175
		r = make_ray(res.pos + indirect_incoming * EPS, indirect_incoming);
		if ((int)bounce < (int)B - 2) {                      // :182-190 (uint8_t operands promote to int)
			float p = fmax2(scale.x, fmax2(scale.y, scale.z));
			if (sr.z > p) break;
			scale = scale / p;
		}
		bounce -= 1;
		pass = 0;
	}
	return color;
}

// One camera sample: jitter -> NDC -> camera ray -> trace. renderer.cpp:359-371
static v3 sample_pixel(trace_ctx& c, uint32_t x, uint32_t y, uint32_t s) {
	const render_cfg& cfg = *c.cfg;
	path_key key = {y * cfg.W + x, s, cfg.seed_lo, cfg.seed_hi};
	f4 j = draws(key, 0, 0, BLOCK_JITTER);
	if (cfg.integrator == 1 && s == 0) j.x = j.y = 0;       // worker.cpp:125-126: the first sample is not offset
	float ndc_x = (((float)x + j.x) / (float)cfg.W) * 2 - 1;
	float ndc_y = (((float)y + j.y) / (float)cfg.H) * 2 - 1;
	ndc_y = -ndc_y;
	float ratio = (float)cfg.W / (float)cfg.H;
	if (cfg.integrator == 1) return trace_worker(c, key, camera_ray(*c.s, ndc_x, ndc_y, ratio));
	return trace(c, key, cfg.bounces, camera_ray(*c.s, ndc_x, ndc_y, ratio), 0);
}

// tonemap_approx_aces (utils.hpp:29-36) + image::write (image.cpp:143-154)
static inline float aces1(float x) {
	float v = (x * (2.51F * x + 0.03F)) / (x * (2.43F * x + 0.59F) + 0.14F);
	v = 0 > v ? 0 : v;   // math::max(x, 0)
	v = 1 < v ? 1 : v;   // math::min(., 1)
	return v;
}
static inline uint8_t to_srgb8(float v) { return (uint8_t)(std::pow(v, 1 / 2.2F) * 255 + 0.5F); }

}  // namespace

// =====================================================================================  C ABI (ctypes)
extern "C" {

// layout of the flat arrays is documented in oracle/pt_oracle.py
void* ora_scene_create(int n_models, const float* model_xform /*[n][12]*/, const int* model_surf /*[n][2]*/,
                       int n_surf, const int* surf_range /*[n_surf][4]: v0,nv,t0,nt*/, const float* verts /*[.][11]*/,
                       const uint32_t* tris /*[.][3]*/, const float* mats /*[n_surf][11]*/,
                       const float* camera /*[14]*/, const float* sun /*[13] or null*/) {
	scene_t* s = new scene_t;
	for (int i = 0; i < n_surf; i++) {
		surface sf;
		const int* rg = surf_range + 4 * i;
		for (int k = 0; k < rg[1]; k++) {
			const float* p = verts + 11 * (size_t)(rg[0] + k);
			sf.m.verts.push_back({{p[0], p[1], p[2]}, p[3], p[4], {p[5], p[6], p[7]}, {p[8], p[9], p[10]}});
		}
		sf.m.tris.assign(tris + 3 * (size_t)rg[2], tris + 3 * (size_t)(rg[2] + rg[3]));
		mesh_finish(sf.m);
		const float* m = mats + 11 * i;
		sf.mat = {{m[0], m[1], m[2]}, m[3], m[4], m[5], {m[6], m[7], m[8]}, m[9], m[10] != 0};
		s->surfaces.push_back(std::move(sf));
	}
	for (int i = 0; i < n_models; i++) {
		const float* x = model_xform + 12 * i;
		model md;
		md.t = {{x[0], x[1], x[2]}, {{x[3], x[4], x[5]}, {x[6], x[7], x[8]}, {x[9], x[10], x[11]}}};
		md.inv = xinverse(md.t);
		md.normal_matrix = transpose(inverse(md.t.basis));
		md.first_surface = model_surf[2 * i];
		md.n_surfaces = model_surf[2 * i + 1];
		aabb_clear(md.box);  // model::recalculate_aabb — model.cpp:13-18
		for (int k = 0; k < md.n_surfaces; k++) {
			aabb_add(md.box, s->surfaces[md.first_surface + k].m.box.mn);
			aabb_add(md.box, s->surfaces[md.first_surface + k].m.box.mx);
		}
		s->models.push_back(md);
	}
	s->camera = {{camera[0], camera[1], camera[2]},
	             {{camera[3], camera[4], camera[5]}, {camera[6], camera[7], camera[8]}, {camera[9], camera[10], camera[11]}}};
	s->fov = camera[12];
	s->tan_half_fov = std::tan(s->fov * 0.5F);  // camera.cpp:27-30
	if (sun) {
		s->has_sun = true;
		s->sun_basis = {{sun[0], sun[1], sun[2]}, {sun[3], sun[4], sun[5]}, {sun[6], sun[7], sun[8]}};
		s->sun_energy = {sun[9], sun[10], sun[11]};
		s->sun_radius = sun[12];
	}
	return s;
}
void ora_scene_destroy(void* p) { delete (scene_t*)p; }

// Textures: n images (8-bit, c channels, row-major as stb_image gives them) and, per surface, the image index of the
// seven material slots (normal, albedo, opacity, occlusion, roughness, metallic, emissive) or -1.
void ora_scene_set_textures(void* p, int n_images, const int* whc_srgb /*[n][4]*/, const uint8_t* const* data, const int* surf_tex /*[n_surf][7]*/) {
	scene_t* s = (scene_t*)p;
	s->textures.clear();
	for (int i = 0; i < n_images; i++) {
		texture t;
		t.w = whc_srgb[4 * i]; t.h = whc_srgb[4 * i + 1]; t.c = whc_srgb[4 * i + 2]; t.srgb = whc_srgb[4 * i + 3] != 0;
		t.data.assign(data[i], data[i] + (size_t)t.w * t.h * t.c);
		s->textures.push_back(std::move(t));
	}
	for (size_t k = 0; k < s->surfaces.size(); k++)
		for (int j = 0; j < 7; j++) s->surfaces[k].mat.tex[j] = surf_tex[7 * k + j];
}
// renderer::environment = image_texture (8-bit image, c channels, sRGB flag); data == nullptr removes it
void ora_scene_set_environment(void* p, int w, int h, int c, int srgb, const uint8_t* data) {
	scene_t* s = (scene_t*)p;
	s->has_env = data != nullptr;
	if (!data) return;
	s->env.w = w; s->env.h = h; s->env.c = c; s->env.srgb = srgb != 0;
	s->env.fdata.clear();
	s->env.data.assign(data, data + (size_t)w * h * c);
}
// ... from a Radiance .hdr: the decoded floats (image::hdr = true)
void ora_scene_set_environment_f32(void* p, int w, int h, int c, int srgb, const float* data) {
	scene_t* s = (scene_t*)p;
	s->has_env = data != nullptr;
	if (!data) return;
	s->env.w = w; s->env.h = h; s->env.c = c; s->env.srgb = srgb != 0;
	s->env.data.clear();
	s->env.fdata.assign(data, data + (size_t)w * h * c);
}
// dirs[n][3] (unit) -> uv[n][2], rgba[n][4] (texture sample), colour[n][3] (x environment_factor)
void ora_env_lookup(void* p, size_t n, const float* dirs, const float* env_factor, float* uv, float* rgba, float* colour) {
	const scene_t& s = *(scene_t*)p;
	for (size_t i = 0; i < n; i++) {
		f4v t = {0, 0, 0, 0};
		const v3 c = miss_colour(s, env_factor, V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), uv + 2 * i, &t);
		rgba[4 * i] = t.x; rgba[4 * i + 1] = t.y; rgba[4 * i + 2] = t.z; rgba[4 * i + 3] = t.w;
		colour[3 * i] = c.x; colour[3 * i + 1] = c.y; colour[3 * i + 2] = c.z;
	}
}
// in[n][2] uv -> out[n][12]: normal_ts(3) albedo(3) opacity roughness metallic emissive(3)  (material.cpp getters)
void ora_material_eval(void* p, int surf, size_t n, const float* uv, float* out) {
	const scene_t& s = *(scene_t*)p;
	for (size_t i = 0; i < n; i++) {
		mat_sample m = material_eval(s.surfaces[surf].mat, s.textures, uv[2 * i], uv[2 * i + 1]);
		float v[12] = {m.normal_ts.x, m.normal_ts.y, m.normal_ts.z, m.albedo.x, m.albedo.y, m.albedo.z, m.opacity, m.roughness, m.metallic,
		               m.emissive.x, m.emissive.y, m.emissive.z};
		memcpy(out + 12 * i, v, sizeof v);
	}
}

// Iteration order of the reference's root-entity container (core/renderer.hpp:25,
// std::unordered_map<std::string, shared_ptr<entity>>; filled by `entities[name] = entity`,
// renderer.cpp:171, in scene-node order; a repeated name REPLACES the earlier entity).
// Uses the same container type on the same libstdc++, so the hash order is the reference's.
float ora_tan_half_fov(float fov) { return std::tan(fov * 0.5F); }  // camera.cpp:27-30 (glibc tanf)

int ora_root_order(int n, const char* const* names, int* order_out) {
	std::unordered_map<std::string, int> m;
	for (int i = 0; i < n; i++) m[names[i]] = i;
	int k = 0;
	for (const auto& [_, v] : m) order_out[k++] = v;
	return k;
}

void ora_scene_boxes(void* p, float* model_aabb /*[n][6]*/, float* mesh_aabb /*[ns][6]*/) {
	scene_t* s = (scene_t*)p;
	for (size_t i = 0; i < s->models.size(); i++) {
		const aabb& b = s->models[i].box;
		float v[6] = {b.mn.x, b.mn.y, b.mn.z, b.mx.x, b.mx.y, b.mx.z};
		memcpy(model_aabb + 6 * i, v, sizeof v);
	}
	for (size_t i = 0; i < s->surfaces.size(); i++) {
		const aabb& b = s->surfaces[i].m.box;
		float v[6] = {b.mn.x, b.mn.y, b.mn.z, b.mx.x, b.mx.y, b.mx.z};
		memcpy(mesh_aabb + 6 * i, v, sizeof v);
	}
}
void ora_kd_counts(void* p, int surf, int* n_nodes, int* n_refs) {
	const mesh& m = ((scene_t*)p)->surfaces[surf].m;
	*n_nodes = (int)m.nodes.size();
	*n_refs = (int)m.refs.size();
}
// pre-order arrays, child / ref indices local to the surface
void ora_kd_get(void* p, int surf, uint8_t* type, uint8_t* axis, float* split, int* left, int* right, int* first,
                int* count, uint32_t* refs) {
	const mesh& m = ((scene_t*)p)->surfaces[surf].m;
	for (size_t i = 0; i < m.nodes.size(); i++) {
		const kd_node& n = m.nodes[i];
		type[i] = n.leaf; axis[i] = n.axis; split[i] = n.split; left[i] = n.left; right[i] = n.right;
		first[i] = n.leaf ? n.first : 0; count[i] = n.count;
	}
	memcpy(refs, m.refs.data(), m.refs.size() * 4);
}

void ora_tri_intersect(size_t n, const float* in /*[n][15]: a b c o d*/, float* out /*[n][4]: t bary*/) {
	for (size_t i = 0; i < n; i++) {
		const float* p = in + 15 * i;
		tri_hit h = tri_intersect({p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]},
		                          {{p[9], p[10], p[11]}, {p[12], p[13], p[14]}});
		float* o = out + 4 * i;
		o[0] = h.t; o[1] = h.bary.x; o[2] = h.bary.y; o[3] = h.bary.z;
	}
}
void ora_aabb_intersect(size_t n, const float* in /*[n][12]: min max o d*/, float* out /*[n][3]: hit near far*/) {
	for (size_t i = 0; i < n; i++) {
		const float* p = in + 12 * i;
		aabb_hit h = aabb_intersect({{p[0], p[1], p[2]}, {p[3], p[4], p[5]}}, {{p[6], p[7], p[8]}, {p[9], p[10], p[11]}});
		float* o = out + 3 * i;
		o[0] = h.hit ? 1.f : 0.f; o[1] = h.hit ? h.nr : 0.f; o[2] = h.hit ? h.fr : -1.f;
	}
}
void ora_mesh_intersect(void* p, int surf, size_t n, const float* rays /*[n][6]*/, float* out /*[n][4]*/, int* idx) {
	const mesh& m = ((scene_t*)p)->surfaces[surf].m;
	for (size_t i = 0; i < n; i++) {
		const float* q = rays + 6 * i;
		mesh_hit h = mesh_intersect(m, {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}}, nullptr);
		float* o = out + 4 * i;
		bool hit = h.t >= 0;
		o[0] = h.t; o[1] = hit ? h.bary.x : 0; o[2] = hit ? h.bary.y : 0; o[3] = hit ? h.bary.z : 0;
		idx[i] = hit ? (int)h.index : -1;
	}
}
void ora_model_intersect(void* p, int mdl, size_t n, const float* rays, float* out /*[n][4]*/, int* idx /*[n][2]*/) {
	const scene_t& s = *(scene_t*)p;
	for (size_t i = 0; i < n; i++) {
		const float* q = rays + 6 * i;
		model_hit h = model_intersect(s, s.models[mdl], {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}}, nullptr);
		float* o = out + 4 * i;
		bool hit = h.dist >= 0;
		o[0] = h.dist; o[1] = hit ? h.bary.x : 0; o[2] = hit ? h.bary.y : 0; o[3] = hit ? h.bary.z : 0;
		idx[2 * i] = hit ? h.surface : -1; idx[2 * i + 1] = hit ? (int)h.tri : -1;
	}
}
// out[n][14] = position(3) uv(2) normal(3) tangent(3) shading_normal(3); idx = surface id or -1
void ora_scene_intersect(void* p, size_t n, const float* rays, float* out, int* idx, uint64_t* stats /*[6+32] or null*/) {
	const scene_t& s = *(scene_t*)p;
	trav_stats st;
	for (size_t i = 0; i < n; i++) {
		const float* q = rays + 6 * i;
		scene_hit h = scene_intersect(s, {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}}, stats ? &st : nullptr);
		float* o = out + 14 * i;
		idx[i] = h.hit ? h.surface : -1;
		if (!h.hit) { for (int k = 0; k < 14; k++) o[k] = 0; continue; }
		v3 sn = shading_normal(h, material_eval(s.surfaces[h.surface].mat, s.textures, h.u, h.v).normal_ts);
		float v[14] = {h.pos.x, h.pos.y, h.pos.z, h.u, h.v, h.nrm.x, h.nrm.y, h.nrm.z, h.tan.x, h.tan.y, h.tan.z, sn.x, sn.y, sn.z};
		memcpy(o, v, sizeof v);
	}
	if (stats) {
		stats[0] = st.model_tests; stats[1] = st.mesh_tests; stats[2] = st.branches;
		stats[3] = st.leaves; stats[4] = st.tris; stats[5] = st.pushes;
		for (int k = 0; k < 32; k++) stats[6 + k] = st.depth_hist[k];
	}
}
// in[n][14] = n(3) o(3) i(3) u1 u2 rough cos_theta ior ; out[n][15] as in ref_harness.cpp "pbr_out"
void ora_pbr(size_t n, const float* in, float* out) {
	for (size_t k = 0; k < n; k++) {
		const float* p = in + 14 * k;
		v3 nrm = {p[0], p[1], p[2]}, o = {p[3], p[4], p[5]}, inc = {p[6], p[7], p[8]};
		float u1 = p[9], u2 = p[10], rough = p[11], ct = p[12], ior = p[13];
		v3 a = rand_cone_vec(u2, ct, nrm);
		v3 b = importance_diffuse(u1, u2, nrm);
		v3 c = importance_specular(u1, u2, nrm, o, rough);
		v3 rf = reflect(-o, nrm);
		float v[15] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, pdf_diffuse(nrm, inc), pdf_specular(nrm, o, inc, rough),
		               fresnel(o, rf, ior), rf.x, rf.y, rf.z};
		memcpy(out + 15 * k, v, sizeof v);
	}
}
void ora_camera_rays(void* p, size_t n, const float* in /*[n][3]: ndc.x ndc.y ratio*/, float* out /*[n][6]*/) {
	const scene_t& s = *(scene_t*)p;
	for (size_t i = 0; i < n; i++) {
		ray r = camera_ray(s, in[3 * i], in[3 * i + 1], in[3 * i + 2]);
		float v[6] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z};
		memcpy(out + 6 * i, v, sizeof v);
	}
}
// in[n][4] linear RGB + alpha -> out[n][4] bytes (tonemap, sRGB, quantise) — renderer.cpp:412-423
void ora_tonemap_write(size_t n, const float* in, uint8_t* out) {
	for (size_t i = 0; i < n; i++) {
		for (int c = 0; c < 3; c++) out[4 * i + c] = to_srgb8(aces1(in[4 * i + c]));
		out[4 * i + 3] = (uint8_t)(in[4 * i + 3] * 255 + 0.5F);
	}
}
// image::write's sRGB quantiser over EVERY float of [0, 1] (bit patterns 0 .. 0x3F800000): is the byte a non-decreasing function of
// the value? first[k] = bits of the smallest value whose byte is >= k (0 for k = 0); returns the number of adjacent pairs where the
// byte DEcreases (0 = monotone: the product may then evaluate the quantiser as a 255-step function, kernels.hip srgb8).
uint64_t ora_srgb8_scan(uint32_t* first /*[256]*/, int threads) {
	if (threads < 1) threads = (int)std::thread::hardware_concurrency();
	const uint64_t N = 0x3F800000ull + 1;
	std::vector<uint64_t> bad(threads, 0);
	std::vector<std::vector<uint32_t>> firsts(threads, std::vector<uint32_t>(256, 0xFFFFFFFFu));
	auto work = [&](int t) {
		const uint64_t a = N * t / threads, b = N * (t + 1) / threads;
		auto q = [](uint32_t bits) { float v; memcpy(&v, &bits, 4); return (uint32_t)to_srgb8(v); };
		uint32_t prev = a ? q((uint32_t)a - 1) : 0;
		for (uint64_t i = a; i < b; i++) {
			const uint32_t c = q((uint32_t)i);
			if (c < prev) bad[t]++;
			for (uint32_t k = prev + 1; k <= c; k++) if (firsts[t][k] == 0xFFFFFFFFu) firsts[t][k] = (uint32_t)i;
			prev = c;
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < threads; t++) th.emplace_back(work, t);
	for (auto& t : th) t.join();
	uint64_t nb = 0;
	for (int k = 0; k < 256; k++) first[k] = 0xFFFFFFFFu;
	first[0] = 0;
	for (int t = 0; t < threads; t++) {
		nb += bad[t];
		for (int k = 1; k < 256; k++) if (firsts[t][k] < first[k]) first[k] = firsts[t][k];
	}
	return nb;
}

void ora_philox(size_t n, const uint32_t* ctr /*[n][4]*/, const uint32_t* key /*[n][2]*/, uint32_t* out /*[n][4]*/) {
	for (size_t i = 0; i < n; i++) {
		u4 r = philox4x32_10({ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]}, key[2 * i], key[2 * i + 1]);
		out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
	}
}
void ora_draws(uint32_t pixel, uint32_t sample, uint32_t seed_lo, uint32_t seed_hi, uint32_t depth, uint32_t pass,
               uint32_t block, float* out4) {
	f4 d = draws({pixel, sample, seed_lo, seed_hi}, depth, pass, block);
	out4[0] = d.x; out4[1] = d.y; out4[2] = d.z; out4[3] = d.w;
}
// Camera rays for a tile of pixels at one sample index (jittered as in renderer.cpp:363-370)
void ora_primary_rays(void* p, const render_cfg* cfg, uint32_t sample, float* out /*[h][w][6]*/) {
	const scene_t& s = *(scene_t*)p;
	for (uint32_t yy = 0; yy < cfg->h; yy++)
		for (uint32_t xx = 0; xx < cfg->w; xx++) {
			uint32_t x = cfg->x0 + xx, y = cfg->y0 + yy;
			path_key key = {y * cfg->W + x, sample, cfg->seed_lo, cfg->seed_hi};
			f4 j = draws(key, 0, 0, BLOCK_JITTER);
			if (cfg->integrator == 1 && sample == 0) j.x = j.y = 0;
			float ndc_x = (((float)x + j.x) / (float)cfg->W) * 2 - 1;
			float ndc_y = -((((float)y + j.y) / (float)cfg->H) * 2 - 1);
			ray r = camera_ray(s, ndc_x, ndc_y, (float)cfg->W / (float)cfg->H);
			float v[6] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z};
			memcpy(out + 6 * ((size_t)yy * cfg->w + xx), v, sizeof v);
		}
}

// renderer::render — renderer.cpp:334-428 (default path: transparent_background = false).
// mean_rgba: [h][w][4] float running mean exactly as renderer.cpp:396-399 (alpha = 1);
// rows are distributed over `threads` std::threads (cf. the row jobs of renderer.cpp:357-402);
// the result does not depend on the thread count (counter-based RNG).
// stats: [0] = rays (renderer::intersect calls), [1..7] traversal counters when want_stats ([7] = rays that hit a surface).
void ora_render(void* p, const render_cfg* cfg, float* mean_rgba, int threads, uint64_t* stats, int want_stats) {
	const scene_t& s = *(scene_t*)p;
	if (threads < 1) threads = (int)std::thread::hardware_concurrency();
	std::vector<uint64_t> rays(threads, 0);
	std::vector<trav_stats> tst(threads);
	auto work = [&](int tid) {
		trace_ctx c{&s, cfg};
		if (want_stats) c.st = &tst[tid];
		for (uint32_t yy = tid; yy < cfg->h; yy += threads)
			for (uint32_t xx = 0; xx < cfg->w; xx++) {
				v3 color = {0, 0, 0};
				float alpha = 0;
				for (uint32_t k = 0; k < cfg->spp; k++) {
					v3 d = sample_pixel(c, cfg->x0 + xx, cfg->y0 + yy, cfg->sample0 + k);
					color = color * (float)k + d;          // pixels.color * sample + data  (uint32 -> float)
					color = color / (float)(k + 1);
					alpha = alpha * (float)k + 1.0f;
					alpha /= (float)(k + 1);
				}
				float* o = mean_rgba + 4 * ((size_t)yy * cfg->w + xx);
				o[0] = color.x; o[1] = color.y; o[2] = color.z; o[3] = alpha;
			}
		rays[tid] = c.rays;
	};
	std::vector<std::thread> th;
	for (int t = 0; t < threads; t++) th.emplace_back(work, t);
	for (auto& t : th) t.join();
	if (stats) {
		for (int k = 0; k < 8; k++) stats[k] = 0;
		for (int t = 0; t < threads; t++) {
			stats[0] += rays[t];
			stats[1] += tst[t].model_tests; stats[2] += tst[t].mesh_tests; stats[3] += tst[t].branches;
			stats[4] += tst[t].leaves; stats[5] += tst[t].tris; stats[6] += tst[t].pushes; stats[7] += tst[t].hits;
		}
	}
}

// renderer::trace(bounces, ray) for n rays IN SEQUENCE on ONE mt19937 stream seeded with `seed` — what the compiled reference
// computes single-threaded with the seed shim of oracle/ref_harness.cpp (`trace` sub-command). out[n][4] = rgb + alpha (alpha is 1
// on every return of trace() with transparent_background = false, renderer.cpp:438-451,643). n_draws (optional) = rand() calls.
void ora_trace_mt(void* p, size_t n, const float* rays /*[n][6]*/, uint32_t bounces, const float* env3, uint32_t seed, int args_rtl,
                  float* out /*[n][4]*/, uint64_t* n_draws) {
	const scene_t& s = *(scene_t*)p;
	render_cfg cfg{};
	cfg.bounces = bounces;
	cfg.env[0] = env3[0]; cfg.env[1] = env3[1]; cfg.env[2] = env3[2];
	mt_stream mt(seed, args_rtl != 0);
	trace_ctx c{&s, &cfg};
	c.mt = &mt;
	const path_key unused{0, 0, 0, 0};
	for (size_t i = 0; i < n; i++) {
		const float* r = rays + 6 * i;
		// the direction is used as given (the harness stores ray::get_dir(), already normalised by the ray's constructor; normalising
		// again is not idempotent in float and would move ~1 % of the rays by an ulp)
		const v3 d = trace(c, unused, bounces, ray{V(r[0], r[1], r[2]), V(r[3], r[4], r[5])}, 0);
		out[4 * i] = d.x; out[4 * i + 1] = d.y; out[4 * i + 2] = d.z; out[4 * i + 3] = 1.0f;
	}
	if (n_draws) *n_draws = mt.n_draws;
}

// Per-sample radiance (no averaging): out[h][w][spp][3]. Used to compare individual paths with the GPU.
void ora_render_samples(void* p, const render_cfg* cfg, float* out, int threads) {
	const scene_t& s = *(scene_t*)p;
	if (threads < 1) threads = (int)std::thread::hardware_concurrency();
	auto work = [&](int tid) {
		trace_ctx c{&s, cfg};
		for (uint32_t yy = tid; yy < cfg->h; yy += threads)
			for (uint32_t xx = 0; xx < cfg->w; xx++)
				for (uint32_t k = 0; k < cfg->spp; k++) {
					v3 d = sample_pixel(c, cfg->x0 + xx, cfg->y0 + yy, cfg->sample0 + k);
					float* o = out + 3 * (((size_t)yy * cfg->w + xx) * cfg->spp + k);
					o[0] = d.x; o[1] = d.y; o[2] = d.z;
				}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < threads; t++) th.emplace_back(work, t);
	for (auto& t : th) t.join();
}

}  // extern "C"
