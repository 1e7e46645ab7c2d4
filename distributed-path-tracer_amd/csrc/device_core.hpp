// Device functions shared by the kernels of kernels.hip (the fused persistent-wave integrator) and wavefront.hip (the queue-based
// pipeline for scenes whose geometry lives in global memory): vector math, the reference's intersection routines, BSDF, RNG,
// textures, one path vertex, LDS staging, stream accessors. Everything is __forceinline__: each .hip file is its own program.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>

#include "kernels.hpp"

namespace ptx {

#define DEV __device__ __forceinline__

// PTX_PROF builds (tools/build_variant.sh prof -DPTX_PROF): count wave-level trips and active lanes per code region, to
// weigh the static instruction counts of the ISA. Not compiled into the product.
#ifdef PTX_PROF
struct Prof { uint32_t t[kProfRegions], l[kProfRegions]; };
#define PROF_ARG , Prof& prof
#define PROF_PASS , prof
// every executing lane counts itself; the lowest executing lane also counts the trip (summed over lanes at the end)
#define PROF(k) do { const uint64_t m_ = __ballot(true); prof.l[k] += 1u; prof.t[k] += ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m_) - 1)) ? 1u : 0u; } while (0)
#else
#define PROF_ARG
#define PROF_PASS
#define PROF(k) do { } while (0)
#endif

constexpr float kEps = 0.0001f;                        // math::epsilon (math/math.hpp:16)
constexpr double kPi = 3.14159265358979323846;         // math::pi is double (math/math.hpp:18)
constexpr double kInvSqrt3 = 1.0 / 1.7320508075688772; // 1 / math::sqrt3 (util/rand_cone_vec.cpp:23)

// 3-vectors keep (x, y) in one register pair so that component-wise + - * compile to packed fp32 instructions (one
// v_pk_* for x and y, one scalar op for z): the same IEEE operation per component, two per issue slot.
typedef float f2 __attribute__((ext_vector_type(2)));
struct V3 {
	union { struct { float x, y; }; f2 xy; };
	float z;
	V3() = default;
	DEV V3(float x_, float y_, float z_) { x = x_; y = y_; z = z_; }
};
DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 mk2(f2 xy, float z) { V3 r; r.xy = xy; r.z = z; return r; }
DEV f2 bc2(float a) { return (f2){a, a}; }
DEV V3 operator+(V3 a, V3 b) { return mk2(a.xy + b.xy, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return mk2(a.xy - b.xy, a.z - b.z); }
DEV V3 operator*(V3 a, V3 b) { return mk2(a.xy * b.xy, a.z * b.z); }
DEV V3 operator*(V3 a, float s) { return mk2(a.xy * bc2(s), a.z * s); }
DEV V3 operator*(float s, V3 a) { return mk2(bc2(s) * a.xy, s * a.z); }
DEV V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
DEV V3 operator-(V3 a) { return mk2(-a.xy, -a.z); }
DEV float dot(V3 a, V3 b) { const f2 p = a.xy * b.xy; return p.x + p.y + a.z * b.z; }     // math/vec3.inl:236
DEV V3 cross(V3 l, V3 r) { return mk((l.y * r.z) - (l.z * r.y), (l.z * r.x) - (l.x * r.z), (l.x * r.y) - (l.y * r.x)); }
DEV float length(V3 a) { return sqrtf(dot(a, a)); }
DEV V3 normalize(V3 a) { return a * (1.0f / length(a)); }                                  // math/vec3.inl:251
DEV float pmax(float a, float b) { return b > a ? b : a; }                                 // math::max (NaN-asymmetric), math.inl:169
DEV float pmin(float a, float b) { return b < a ? b : a; }                                 // math::min, math.inl:179
DEV float lerpf(float a, float b, float w) { return a + (b - a) * w; }                     // math.inl:164
DEV float clampf(float x, float lo, float hi) { return pmin(pmax(x, lo), hi); }            // math.inl:154
DEV V3 lerp3(V3 a, V3 b, float w) { return a + (b - a) * w; }
DEV V3 lerp3(V3 a, V3 b, V3 w) { return a + (b - a) * w; }
DEV V3 reflect3(V3 incident, V3 normal) { return incident - 2 * dot(normal, incident) * normal; }  // core/utils.hpp:38
// math::pow(float, 5): std::pow promotes to double; x^5 by exact-ish double products, rounded once to float
DEV float pow5(float x) { double d = (double)x; double d2 = d * d; return (float)(d2 * d2 * d); }
DEV float sel3(V3 v, uint32_t axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
// column-major 3x3 (float[9] = x.xyz y.xyz z.xyz) times vector: each ROW dotted with v (math/mat3.inl:219-224)
DEV V3 mulmv(const float* m, V3 v) {
	const f2 c0 = {m[0], m[1]}, c1 = {m[3], m[4]}, c2 = {m[6], m[7]};
	return mk2(c0 * bc2(v.x) + c1 * bc2(v.y) + c2 * bc2(v.z), m[2] * v.x + m[5] * v.y + m[8] * v.z);
}

// ------------------------------------------------------------------------------------ geometry access
// The three traversal arrays of one copy of the geometry: the LDS-resident one or the global-memory one.
struct Geom {
	const uint2* nodes;
	const uint32_t* refs;
	const float4* tris;   // TriIsect records, three float4 each (flat_scene.hpp)
	// true (global-memory kernels): `tris` holds one record per LEAF REFERENCE, in leaf order, with the triangle id in its
	// spare word — a leaf's records are contiguous and the refs -> record indirection (a second dependent fetch from
	// L2/HBM per triangle) disappears. false (LDS kernels): one record per triangle, reached through `refs`.
	bool leaf_ordered;
	// true (the global-memory copy): a branch node's two children (adjacent, 16 bytes) are requested as soon as the parent arrives
	bool pair;
};
// Both copies a kernel may traverse (kernels.hpp: MODE_GLOBAL / MODE_LDS / MODE_HYBRID); in MODE_HYBRID the branch between
// them is wave-uniform (the surface index is).
struct Geoms { Geom lds, glb; };

// The small per-model / per-surface tables are read with a wave-uniform index. They are passed to the kernels as
// separate `const T* __restrict__` arguments (not inside DevScene): only then can the compiler prove that the
// kernel's own stores do not clobber them and fetch them with scalar loads (s_load -> SGPRs) instead of one
// vector load per field per lane.
struct Tables {
	const ModelRec* models;
	const SurfaceRec* surfaces;
	const SpaceRec* spaces;
	const uint32_t* model_space;
};

// geometry::triangle::intersect — geometry/triangle.cpp:120-190 (Cramer's rule, no culling, +-epsilon slack).
// The same solve from a TriIsect record (e1 = a-b, e2 = a-c and c3 precomputed with identical float operations), written
// on 2-wide vectors so that it compiles to packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, operands swizzled with
// op_sel): every product and every sum is the reference's IEEE operation on the reference's operands, two at a time.
// The record pairs the edge components so that the cofactors come out as (c1, -c2), (c4, -c4), (c6, -c5): a term the
// reference subtracts is then added with its sign already flipped, and x + (-y) == x - y, -(x*y) == (-x)*y exactly.
// Returns the distance, or -1 when the barycentric tests fail (a NaN from a zero determinant fails `t >= 0` later).
DEV f2 swp(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }
DEV f2 bc(float a) { return (f2){a, a}; }
struct PRay { f2 oyz, dyz; float ox, dx; };   // the local ray, arranged for tri_test_pk
DEV PRay pack_ray(V3 o, V3 d) { return {{o.y, o.z}, {d.y, d.z}, o.x, d.x}; }
DEV float tri_test_pk(float4 r0, float4 r1, float2 r2, const PRay& r, float& beta, float& gamma) {
	const f2 Pa = {r0.x, r0.y}, Pb = {r0.z, r0.w};               // (e2.y, e1.z), (e2.z, e1.y)
	const f2 Ex = {r1.x, r1.y}, Ayz = {r1.z, r1.w};              // (e1.x, e2.x), (a.y, a.z)
	const float c3 = r2.y;
	const f2 c1n2 = Pa * swp(r.dyz) - Pb * r.dyz;                // (e2.y*d.z - d.y*e2.z, d.y*e1.z - e1.y*d.z) = (c1, -c2)
	const f2 vyz = Ayz - r.oyz;
	const float vx = r2.x - r.ox;
	const f2 m4 = vyz * swp(r.dyz);                              // (v.y*d.z, d.y*v.z)
	const f2 c4s = m4 - swp(m4);                                 // (c4, -c4)
	const f2 c6n5 = Pa * swp(vyz) - Pb * vyz;                    // (e2.y*v.z - v.y*e2.z, v.y*e1.z - e1.y*v.z) = (c6, -c5)
	const f2 t12 = Ex * c1n2;                                    // (e1.x*c1, -(e2.x*c2))
	const float inv_det = 1.0f / ((t12.x + t12.y) + r.dx * c3);
	const f2 X = bc(vx) * c1n2 - swp(Ex) * c4s;                  // (v.x*c1 - e2.x*c4, e1.x*c4 - v.x*c2)
	const f2 N = X - bc(r.dx) * c6n5;                            // (.. - d.x*c6, .. + d.x*c5): numerators of beta, gamma
	const f2 bg = bc(inv_det) * N;
	const f2 Z = Ex * c6n5;                                      // (e1.x*c6, -(e2.x*c5))
	const float t = inv_det * ((Z.x + Z.y) + vx * c3);
	beta = bg.x; gamma = bg.y;
	const bool out = (bg.x < 0 - kEps) | (bg.x > 1 + kEps) | (bg.y < 0 - kEps) | (bg.y + bg.x > 1 + kEps);
	return out ? -1.0f : t;
}

struct MeshHit { float t; float b1, b2; uint32_t tri; };

constexpr int kRegStack = 3;    // pending KD subtrees kept in registers (covers > 99 % of traversals)
constexpr int kSpillStack = 24; // deeper entries go to a per-lane overflow area in global memory (touched by < 1 % of
                                // traversals): the reference pushes at most one entry per level and its trees are at
                                // most 26 levels deep (mesh.hpp:34, max_depth = 25)

// Overflow stack of one lane: entry k lives at base[k * 64] (uint2 = node, min_dist bits), so that the 64 lanes of a wave
// touch one 512-byte row per level. An explicit global array, not a private one: a private array would be turned into
// registers + compare/select chains, or into scratch whose address arithmetic sits in every push and pop.
struct Spill { uint2* base; };
DEV void spill_put(const Spill& sp, int k, uint32_t node, float m) { sp.base[k * 64] = make_uint2(node, __float_as_uint(m)); }
DEV void spill_get(const Spill& sp, int k, uint32_t& node, float& m) { uint2 v = sp.base[k * 64]; node = v.x; m = __uint_as_float(v.y); }

// geometry::aabb::intersect with the reciprocal direction hoisted: the same local ray is tested against the
// model box and every surface box, and 1/dir has one value per ray whatever box it meets.
// The reference's min/max are compare-selects whose result depends on operand order when a NaN is involved
// (math.inl:169-182); NaNs appear here only as 0 * inf (origin exactly on a box plane, direction exactly parallel
// to it). Without a NaN, hardware v_min/v_max give the same values (up to the sign of a zero, which no later
// comparison distinguishes), so: fast path on v_min_f32 / v_max3_f32, exact compare-select path when the sum of
// the six slab distances is NaN (which also catches inf - inf; taking the exact path then is merely slower).
// `b` = (min.x, max.x, min.y, max.y, min.z, max.z) (ModelRec / SurfaceRec::box): the six slab distances come out of three packed
// subtractions and three packed multiplications on (min, max) pairs — the reference's IEEE operations on its operands, two at a time,
// each pair one 64-bit scalar operand. The NaN test may sum them in any order (a NaN survives every order).
// the hardware minimum / maximum as they are: fminf / fmaxf put a canonicalising v_max x, x in front of every operand (six per box), which
// changes nothing on these operands (products, no NaN on this path)
DEV float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEV float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// The unpacked form (fused kernels: they sit at their register cap, and the aligned pairs of the packed form cost them more in spills than
// the halved multiplications give back: Cornell 1523 -> 1517 Msamples/s, profiles/round3_surface_order.txt)
DEV bool aabb_test_inv(const float* mn, const float* mx, V3 o, V3 inv, float& nr, float& fr) {
	const float m0 = mn[0], m1 = mn[1], m2 = mn[2], x0 = mx[0], x1 = mx[1], x2 = mx[2];
	if ((m0 > x0) | (m1 > x1) | (m2 > x2)) return false;
	const float ax = (m0 - o.x) * inv.x, ay = (m1 - o.y) * inv.y, az = (m2 - o.z) * inv.z;
	const float bx = (x0 - o.x) * inv.x, by = (x1 - o.y) * inv.y, bz = (x2 - o.z) * inv.z;
	const float s = ((ax + bx) + (ay + by)) + (az + bz);
	if (s == s) {
		nr = hw_max3(hw_min(ax, bx), hw_min(ay, by), hw_min(az, bz));   // (fminf / fmaxf here: Cornell 1531 -> 1545 Msamples/s without their canonicalising)
		fr = hw_min3(hw_max(ax, bx), hw_max(ay, by), hw_max(az, bz));
	} else {
		nr = pmax(pmax(pmin(ax, bx), pmin(ay, by)), pmin(az, bz));
		fr = pmin(pmin(pmax(ax, bx), pmax(ay, by)), pmax(az, bz));
	}
	if (nr > fr) return false;
	return fr >= 0;
}
DEV bool aabb_test_box(const float* b, V3 o, V3 inv, float& nr, float& fr) {
	// all six bounds are read before the first decision: the boxes sit in scalar-loaded tables, and a short-circuit `||` made three
	// dependent scalar-load round trips out of one
	const f2 X = {b[0], b[1]}, Y = {b[2], b[3]}, Z = {b[4], b[5]};
	if ((X.x > X.y) | (Y.x > Y.y) | (Z.x > Z.y)) return false;
	const f2 tx = (X - bc(o.x)) * bc(inv.x), ty = (Y - bc(o.y)) * bc(inv.y), tz = (Z - bc(o.z)) * bc(inv.z);   // (a, b) of each axis
	const f2 sm = (tx + ty) + tz;
	const float s = sm.x + sm.y;
	if (s == s) {
		nr = hw_max3(hw_min(tx.x, tx.y), hw_min(ty.x, ty.y), hw_min(tz.x, tz.y));
		fr = hw_min3(hw_max(tx.x, tx.y), hw_max(ty.x, ty.y), hw_max(tz.x, tz.y));
	} else {
		nr = pmax(pmax(pmin(tx.x, tx.y), pmin(ty.x, ty.y)), pmin(tz.x, tz.y));
		fr = pmin(pmin(pmax(tx.x, tx.y), pmax(ty.x, ty.y)), pmax(tz.x, tz.y));
	}
	if (nr > fr) return false;
	return fr >= 0;
}

// core::mesh::intersect — core/mesh.cpp:300-405: front-to-back stack traversal, returns at the first
// leaf that yields a hit within [.., max_dist].
// Stack entries are (node, min_dist) only: the max_dist the reference stores with an entry is always the
// min_dist of the entry beneath it (each push hands its old max_dist to the pushed subtree and continues
// with max_dist = split_dist = the pushed min_dist), and the AABB exit distance for the bottom one.
// (nr, fr) = the surface box's entry / exit distances (mesh.cpp:308-315): tested by the caller, which may decide with the
// result whether the lane traverses now or is set aside for a full-wave sweep.
template <int PB>
DEV bool mesh_traverse(const Geom& g, uint32_t root, float nr, float fr, V3 o, V3 d, MeshHit& out, const Spill& spill PROF_ARG) {
	PROF(PB);
	const PRay pr = pack_ray(o, d);
	int sp = 0;
	uint32_t n0 = 0, n1 = 0, n2 = 0;  // register stack: entry 0 is the top
	float m0 = 0, m1 = 0, m2 = 0;
	uint32_t node = root;
	float min_dist = nr, max_dist = fr;
	bool have = true;
	for (;;) {
		PROF(PB + 1);
		if (!have) {
			if (sp == 0) return false;
			sp--;
			node = n0; min_dist = m0;
			n0 = n1; m0 = m1; n1 = n2; m1 = m2;
			if (sp >= kRegStack) spill_get(spill, sp - kRegStack, n2, m2);
			max_dist = sp > 0 ? m0 : fr;
		}
		have = false;
		bool valid = true;
		uint2 nd = g.nodes[node];
		while ((nd.y & 3u) != KD_LEAF) {
			PROF(PB + 2);
			uint32_t axis = nd.y & 3u;
			// global-memory trees: both children (adjacent, 16 bytes) are requested as soon as the parent arrives, so the
			// split arithmetic below (an IEEE division) runs under the fetch instead of before it
			uint2 kid0 = make_uint2(0, 0), kid1 = make_uint2(0, 0);
#ifndef PTX_NO_PAIR_FETCH
			constexpr bool pair_fetch = true;
#else
			constexpr bool pair_fetch = false;
#endif
#ifdef PTX_LDS_PAIR_FETCH
			if (pair_fetch) { kid0 = g.nodes[nd.y >> 4]; kid1 = g.nodes[(nd.y >> 4) + 1u]; }
#else
			if (pair_fetch && g.pair) { kid0 = g.nodes[nd.y >> 4]; kid1 = g.nodes[(nd.y >> 4) + 1u]; }
#endif
			float split = __uint_as_float(nd.x);
			float oa = sel3(o, axis), da = sel3(d, axis);
			float split_dist = (split - oa) / da;
			bool has_l = nd.y & 4u, has_r = nd.y & 8u;
			uint32_t li = nd.y >> 4, ri = li + (has_l ? 1u : 0u);
			bool left_first = oa < split;
			uint32_t first = left_first ? li : ri, second = left_first ? ri : li;
			bool has_first = left_first ? has_l : has_r, has_second = left_first ? has_r : has_l;
			uint32_t next;
			bool has_next;
			if (split_dist < 0 || split_dist > max_dist) { next = first; has_next = has_first; }
			else if (split_dist < min_dist) { next = second; has_next = has_second; }
			else {
				if (has_second && sp < kRegStack + kSpillStack) {
					if (sp >= kRegStack) spill_put(spill, sp - kRegStack, n2, m2);
					n2 = n1; m2 = m1; n1 = n0; m1 = m0; n0 = second; m0 = split_dist;
					sp++;
				}
				next = first; has_next = has_first;
				max_dist = split_dist;
			}
			if (!has_next) { valid = false; break; }
			node = next;
#ifdef PTX_LDS_PAIR_FETCH
			nd = pair_fetch ? (next == li ? kid0 : kid1) : g.nodes[node];
#else
			nd = (pair_fetch && g.pair) ? (next == li ? kid0 : kid1) : g.nodes[node];
#endif
		}
		if (!valid) continue;
		// leaf: nearest triangle with t <= max_dist; ties keep the first (mesh.cpp:381-389)
		uint32_t first_ref = nd.x, count = nd.y >> 2;
		float best_t = -1.0f, bb1 = 0, bb2 = 0;
		uint32_t best_tri = 0;
		// (a software-pipelined form of this loop — triangle i + 1's reference and record requested before triangle i's solve — was
		// measured: -2.7 % on Cornell, +2 % on jack-of-blades; the extra registers cost more than the latency 4 waves already hide)
		for (uint32_t i = 0; i < count; i++) {
			PROF(PB + 3);
			const uint32_t slot = g.leaf_ordered ? first_ref + i : g.refs[first_ref + i];
			const float4 r0 = g.tris[3 * slot], r1 = g.tris[3 * slot + 1], r2 = g.tris[3 * slot + 2];
			const uint32_t ti = __float_as_uint(r2.z);   // global triangle id, carried by every record
			float be, ga;
			const float t = tri_test_pk(r0, r1, make_float2(r2.x, r2.y), pr, be, ga);
			if (t >= 0 && t <= max_dist && (t < best_t || !(best_t >= 0))) { best_t = t; bb1 = be; bb2 = ga; best_tri = ti; }
		}
		if (!(best_t >= 0)) continue;
		out.t = best_t; out.b1 = bb1; out.b2 = bb2; out.tri = best_tri;
		return true;
	}
}

// core::mesh::intersect on whichever copy of the surface's tree the scene's MODE prescribes; (nr, fr) from the box test
template <int MODE, int PB>
DEV bool mesh_traverse_m(const Geoms& G, const SurfaceRec& sf, float nr, float fr, V3 o, V3 d, MeshHit& out, const Spill& spill PROF_ARG) {
	if constexpr (MODE == MODE_GLOBAL) return mesh_traverse<PB>(G.glb, sf.kd_root, nr, fr, o, d, out, spill PROF_PASS);
	else if constexpr (MODE == MODE_LDS) return mesh_traverse<PB>(G.lds, sf.lds_root, nr, fr, o, d, out, spill PROF_PASS);
	else {
		if (sf.lds_root != 0xFFFFFFFFu) return mesh_traverse<PB>(G.lds, sf.lds_root, nr, fr, o, d, out, spill PROF_PASS);
		return mesh_traverse<PB>(G.glb, sf.kd_root, nr, fr, o, d, out, spill PROF_PASS);
	}
}
// ... including the box test (mesh.cpp:308-315)
template <int MODE, int PB>
DEV bool mesh_intersect_m(const Geoms& G, const SurfaceRec& sf, V3 o, V3 d, V3 inv, MeshHit& out, const Spill& spill PROF_ARG) {
	float nr, fr;
	if (!aabb_test_inv(sf.bmin, sf.bmax, o, inv, nr, fr)) return false;
	return mesh_traverse_m<MODE, PB>(G, sf, nr, fr, o, d, out, spill PROF_PASS);
}

// Closest hit record: what the shading phase needs to rebuild everything else.
// alpha is not stored: it is 1 - beta - gamma (triangle.cpp:185), recomputed with the same two subtractions.
struct SceneHit { float dist; int surface; uint32_t tri; float b1, b2; };

// renderer::intersect (core/renderer.cpp:645-671) over scene::model::intersect (scene/model.cpp:20-72)
template <int MODE>
DEV bool scene_traverse(const DevScene& S, const Geoms& g, V3 o, V3 d, SceneHit& best, const Spill& spill) {
#ifdef PTX_PROF
	Prof prof{};   // not reported: only the EXTEND sweeps of k_render_pass are profiled
#endif
	best.dist = -1.0f;
	best.surface = -1;
	uint32_t cur_space = 0xFFFFFFFFu;
	V3 lo = o, ld = d, inv = d;
	for (int m = 0; m < S.n_models; m++) {
		const ModelRec& M = S.models[m];
		// ray::transform(inverse): origin' = inv*o, dir' = normalize(inv.basis*dir)  (geometry/ray.cpp:10-15); models whose
		// transforms are bitwise equal share a SpaceRec, so the local ray is recomputed only when the space changes
		const uint32_t spc = S.model_space[m];   // wave-uniform
		if (spc != cur_space) {
			const SpaceRec& SP = S.spaces[spc];
			lo = mulmv(SP.inv_basis, o) + mk(SP.inv_origin[0], SP.inv_origin[1], SP.inv_origin[2]);
			ld = normalize(mulmv(SP.inv_basis, d));
			inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
			cur_space = spc;
		}
		float nr, fr;
		if (!aabb_test_inv(M.bmin, M.bmax, lo, inv, nr, fr)) continue;
		MeshHit nearest;
		nearest.t = -1.0f;
		int hit_surface = -1;
		for (int s = 0; s < M.n_surfaces; s++) {
			MeshHit h;
			if (!mesh_intersect_m<MODE, 4>(g, S.surfaces[M.first_surface + s], lo, ld, inv, h, spill PROF_PASS)) continue;
			if (h.t < nearest.t || !(nearest.t >= 0)) { nearest = h; hit_surface = M.first_surface + s; }
		}
		if (!(nearest.t >= 0)) continue;
		// local -> world distance (model.cpp:62-63)
		float wd = length(mulmv(M.basis, ld * nearest.t));
		if (!(wd >= 0)) continue;
		if (wd < best.dist || !(best.dist >= 0)) {
			best.dist = wd; best.surface = hit_surface; best.tri = nearest.tri; best.b1 = nearest.b1; best.b2 = nearest.b2;
		}
	}
	return best.surface >= 0;
}

// ------------------------------------------------------------------------------------ deferred models
// scene_traverse makes the whole wave wait for every model that ANY of its 64 rays enters: on Cornell 9 % / 4 % / 3 % /
// 1.5 % of the rays enter the two boxes, the light and the sphere, so their triangle loops run with a handful of
// lanes in almost every wave-iteration. Here a model entered by fewer than kInlineMin lanes of a wave-iteration is
// not traversed on the spot: the lanes append their ray index to that model's wave-private list, and after the sweep
// each list is traversed with full waves. Per ray the arithmetic is unchanged; the closest hit is the minimum over
// models of the world distance, ties going to the model visited first (renderer.cpp:663-669), which is evaluated
// here as (distance, surface id) order because surface ids grow with the visit order.
// renderer::intersect(shadow ray).has_hit() (renderer.cpp:509-511, intersection_worker.cpp:58-61): the reference finds the closest
// hit and then only asks whether there is one, so the sweep may stop at the first model that reports a hit.
template <int MODE>
DEV bool scene_occluded(const DevScene& S, const Geoms& g, V3 o, V3 d, const Spill& spill) {
#ifdef PTX_PROF
	Prof prof{};
#endif
	uint32_t cur_space = 0xFFFFFFFFu;
	V3 lo = o, ld = d, inv = d;
	for (int m = 0; m < S.n_models; m++) {
		const ModelRec& M = S.models[m];
		const uint32_t spc = S.model_space[m];   // wave-uniform
		if (spc != cur_space) {
			const SpaceRec& SP = S.spaces[spc];
			lo = mulmv(SP.inv_basis, o) + mk(SP.inv_origin[0], SP.inv_origin[1], SP.inv_origin[2]);
			ld = normalize(mulmv(SP.inv_basis, d));
			inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
			cur_space = spc;
		}
		float nr, fr;
		if (!aabb_test_inv(M.bmin, M.bmax, lo, inv, nr, fr)) continue;
		for (int s = 0; s < M.n_surfaces; s++) {
			MeshHit h;
			if (!mesh_intersect_m<MODE, 4>(g, S.surfaces[M.first_surface + s], lo, ld, inv, h, spill PROF_PASS)) continue;
			if (length(mulmv(M.basis, ld * h.t)) >= 0) return true;   // model.cpp:62-63: a hit whose world distance is not NaN
		}
	}
	return false;
}


// Exact pruning of a set-aside traversal: every point a triangle test of the unit can accept lies inside its slack-grown box
// (pbmin / pbmax, scene_build.cpp), so no hit of the unit is nearer than the ray's entry into that box. If even that entry —
// as a world distance, shortened by 1e-4 relative to cover every rounding on the way — lies beyond the closest hit the ray
// already has, no candidate of the unit can win (candidates win on `<`, or on `==` with a lower id) and the traversal is
// skipped; the result is bitwise what it would have been.
DEV bool cannot_win(const float* pbox /* pbmin[3] then pbmax[3] */, const float* basis, V3 lo, V3 ld, V3 inv, float best_wd) {
	if (!(best_wd >= 0)) return false;
	float pn, pf;
	if (!aabb_test_inv(pbox, pbox + 3, lo, inv, pn, pf)) return true;    // misses even the grown box: nothing to find
	if (!(pn > 0)) return false;
	return length(mulmv(basis, ld * pn)) * 0.9999f > best_wd;
}

// Closest hit inside ONE model for a lane that is known to enter its box: scene::model::intersect (model.cpp:27-63)
template <int MODE, int PB>
DEV bool model_traverse(const DevScene& S, const Geoms& g, const ModelRec& M, V3 lo, V3 ld, V3 inv, float& wd, int& surf, uint32_t& tri,
                        float& b1, float& b2, const Spill& spill PROF_ARG) {
	MeshHit nearest;
	nearest.t = -1.0f;
	int hit_surface = -1;
	for (int k = 0; k < M.n_surfaces; k++) {
		MeshHit h;
		if (!mesh_intersect_m<MODE, PB>(g, S.surfaces[M.first_surface + k], lo, ld, inv, h, spill PROF_PASS)) continue;
		if (h.t < nearest.t || !(nearest.t >= 0)) { nearest = h; hit_surface = M.first_surface + k; }
	}
	if (!(nearest.t >= 0)) return false;
	wd = length(mulmv(M.basis, ld * nearest.t));
	if (!(wd >= 0)) return false;
	surf = hit_surface; tri = nearest.tri; b1 = nearest.b1; b2 = nearest.b2;
	return true;
}

// SURF kernels set aside single SURFACES instead of whole models (a Sponza-class model has two dozen of them, each entered by
// a handful of the wave's rays). The running closest hit of a ray then receives (model, surface) candidates in any order —
// inline ones during the sweep, set-aside ones later. scene::model::intersect keeps, per model, the smallest LOCAL t (first
// surface wins ties, model.cpp:47-55); renderer::intersect keeps, over models, the smallest WORLD distance (first model wins
// ties, renderer.cpp:666-669). Hence: against a candidate of the same model compare local t, otherwise world distance.
struct Best { float wd, tl; int surf, model; uint32_t tri; float b1, b2; };
DEV void best_reset(Best& b) { b.wd = -1.0f; b.tl = -1.0f; b.surf = -1; b.model = -1; b.tri = 0; b.b1 = 0; b.b2 = 0; }
DEV bool best_offer(Best& b, const ModelRec& M, int model, V3 ld, int surf, const MeshHit& h) {
	const float wd = length(mulmv(M.basis, ld * h.t));   // local -> world distance (model.cpp:62-63)
	if (!(wd >= 0)) return false;
	bool win;
	if (b.surf >= 0 && b.model == model) win = h.t < b.tl || (h.t == b.tl && surf < b.surf);
	else win = b.surf < 0 || wd < b.wd || (wd == b.wd && model < b.model);
	if (win) { b.wd = wd; b.tl = h.t; b.surf = surf; b.model = model; b.tri = h.tri; b.b1 = h.b1; b.b2 = h.b2; }
	return win;
}

struct Surf { V3 pos, nrm, tan; float u, v; };

// attribute interpolation of renderer::intersect — core/renderer.cpp:688-715. One round of nine 16-byte fetches from the triangle's
// hit record (flat_scene.hpp: HitRec); the sums are the reference's, term by term.
DEV void hit_attributes(const DevScene& S, const ShadeRec& R, uint32_t tri_word, float b1, float b2, Surf& out) {
	const float b0 = 1 - b1 - b2;
	// the record: from LDS when the triangle is one of the hot ones (DevScene::hot_lds: nine LDS reads instead of a round of nine global
	// gathers — on Cornell the 170 largest triangles take nearly all hits), else from global memory
	const uint32_t slot = S.tri_id_mask == 0x00FFFFFFu ? tri_word >> 24 : 0xFFu;
	float4 A, B, C, a0, e0, c0, a1, e1, c1;
	if (S.hot_lds && slot != 0xFFu) {
		// (an explicit LDS pointer: left generic, the two branches merge into flat loads, which go through the vector-memory path anyway)
		typedef float f4v __attribute__((ext_vector_type(4)));
		typedef const __attribute__((address_space(3))) f4v lds_f4;
		lds_f4* H = (lds_f4*)S.hot_lds + 9u * slot;
		auto ld = [&](int k) { const f4v v = H[k]; return make_float4(v.x, v.y, v.z, v.w); };
		A = ld(0); B = ld(1); C = ld(2); a0 = ld(3); e0 = ld(4); c0 = ld(5); a1 = ld(6); e1 = ld(7); c1 = ld(8);
	} else {
		const float4* H = S.tris + 9 * (size_t)(tri_word & S.tri_id_mask);
		A = H[0]; B = H[1]; C = H[2]; a0 = H[3]; e0 = H[4]; c0 = H[5]; a1 = H[6]; e1 = H[7]; c1 = H[8];
	}
	V3 lp = mk(A.x, A.y, A.z) * b0 + mk(B.x, B.y, B.z) * b1 + mk(C.x, C.y, C.z) * b2;
	out.pos = mulmv(R.basis, lp) + mk(R.origin[0], R.origin[1], R.origin[2]);
	out.u = A.w * b0 + B.w * b1 + C.w * b2;
	out.v = a0.w * b0 + e0.w * b1 + c0.w * b2;
	out.nrm = normalize(mulmv(R.nmat, mk(a0.x, a0.y, a0.z) * b0 + mk(e0.x, e0.y, e0.z) * b1 + mk(c0.x, c0.y, c0.z) * b2));
	out.tan = normalize(mulmv(R.nmat, mk(a1.x, a1.y, a1.z) * b0 + mk(e1.x, e1.y, e1.z) * b1 + mk(c1.x, c1.y, c1.z) * b2));
}

// intersect_result::get_normal (renderer.cpp:430-435): TBN * material::get_normal(uv); nts = (0,0,1) without a normal map
DEV V3 shading_normal(const Surf& s, V3 nts) {
	V3 bin = cross(s.nrm, s.tan);
	return {s.tan.x * nts.x + bin.x * nts.y + s.nrm.x * nts.z, s.tan.y * nts.x + bin.y * nts.y + s.nrm.y * nts.z,
	        s.tan.z * nts.x + bin.z * nts.y + s.nrm.z * nts.z};
}

// ------------------------------------------------------------------------------------ sampling / BSDF
// util::rand_cone_vec — util/rand_cone_vec.cpp:8-35
DEV V3 rand_cone_vec(float rnd, float cos_theta, V3 normal) {
	float phi = (float)((double)(rnd * 2) * kPi);
	float sin_theta = sqrtf(1 - cos_theta * cos_theta);
	float sp, cp;
#ifdef PTX_SEPARATE_SINCOS
	sp = sinf(phi); cp = cosf(phi);
#else
	sincosf(phi, &sp, &cp);   // one argument reduction for both (ocml evaluates sinf / cosf through the same reduced kernels:
	                          // images are bit-identical to the two separate calls — checked with a -DPTX_SEPARATE_SINCOS build)
#endif
	V3 cone = {cp * sin_theta, sp * sin_theta, cos_theta};
	V3 np = {0, 0, 0};
	if ((double)fabsf(normal.x) < kInvSqrt3) np.x = 1;
	else if ((double)fabsf(normal.y) < kInvSqrt3) np.y = 1;
	else np.z = 1;
	V3 tangent = normalize(cross(normal, np));
	V3 binormal = cross(normal, tangent);
	return {tangent.x * cone.x + binormal.x * cone.y + normal.x * cone.z, tangent.y * cone.x + binormal.y * cone.y + normal.y * cone.z,
	        tangent.z * cone.x + binormal.z * cone.y + normal.z * cone.z};
}
DEV float fresnel_schlick(V3 outcoming, V3 incoming, float ior) {  // core/pbr.cpp:13-25
	V3 halfway = normalize(outcoming + incoming);
	float cos_theta = dot(outcoming, halfway);
	float f0 = (ior - 1) / (ior + 1);
	f0 *= f0;
	return lerpf(f0, 1, pow5(1 - cos_theta));
}
// pbr::importance_diffuse (core/pbr.cpp:71-77) and pbr::importance_specular (:79-91) differ in the cone angle they hand to
// rand_cone_vec and in the final reflection; the cone construction itself (sin / cos of the azimuth, tangent frame) is the
// same code — run it once for the whole wave instead of once per lobe under complementary lane masks.
DEV V3 importance_sample(bool specular, float u1, float u2, V3 normal, V3 outcoming, float roughness) {
	float cos_theta;
	if (specular) {
		roughness *= roughness;
		roughness *= roughness;
		cos_theta = sqrtf((1 - u1) / (1 + (roughness - 1) * u1));
	} else {
		cos_theta = cosf(acosf(2 * u1 - 1) * 0.5F);
	}
	const V3 h = rand_cone_vec(u2, cos_theta, normal);
	return specular ? reflect3(-outcoming, h) : h;
}
DEV float smith_g1(V3 n, V3 l, float k) { float c = dot(n, l); return c / pmax(lerpf(k, 1, c), kEps); }  // pbr.cpp:95-102
DEV float pdf_diffuse(V3 n, V3 i) { return (float)((double)dot(n, i) / kPi); }                           // pbr.cpp:118-123
DEV float pdf_specular(V3 n, V3 o, V3 i, float roughness) {  // core/pbr.cpp:172-184 (+ distribution_ggx :125-140, geometry_smith :104-114)
	float r4 = roughness * roughness;
	r4 *= r4;
	V3 halfway = normalize(o + i);
	float cos_phi = dot(n, halfway);
	float denom = 1 + (r4 - 1) * (cos_phi * cos_phi);
	float cos_theta = dot(n, i);
	double dd = kPi * (double)denom * (double)denom;
	double mxd = (double)kEps > dd ? (double)kEps : dd;
	float dist = (float)((double)(cos_theta * r4) / mxd);
	float r = roughness + 1;
	float k = (r * r) / 8;
	float geo = smith_g1(n, o, k) * smith_g1(n, i, k);
	float ndo = dot(n, o), ndi = dot(n, i);
	return (dist * geo) / pmax(4 * ndo * ndi, kEps);
}
// BRDF / PDF combination, inline in renderer::trace (core/renderer.cpp:521-556 and :579-606).
// Returns brdf (rgb); pdf_mix = lerp(pdf_d, pdf_s, specular_probability).
DEV V3 eval_brdf(V3 n, V3 o, V3 i, V3 albedo, float roughness, float metallic, float spec_prob, float& pdf_mix) {
	float diffuse_pdf = pdf_diffuse(n, i);
	V3 diffuse_brdf = diffuse_pdf * albedo;
	float specular_pdf = pdf_specular(n, o, i, roughness);
	V3 fr = lerp3(mk(0.04F, 0.04F, 0.04F), albedo, metallic);
	V3 halfway = normalize(o + i);
	float cos_theta = dot(o, halfway);
	fr = lerp3(fr, mk(1, 1, 1), pow5(1 - cos_theta));
	diffuse_brdf = lerp3(diffuse_brdf, mk(0, 0, 0), metallic);
	pdf_mix = lerpf(diffuse_pdf, specular_pdf, spec_prob);
	return lerp3(diffuse_brdf, mk(specular_pdf, specular_pdf, specular_pdf), fr);
}

// ------------------------------------------------------------------------------------ RNG
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw — SC'11). Replaces core::rand() (core/utils.hpp:8-13).
DEV uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
	for (int i = 0; i < 10; i++) {
		// one 32x32->64 multiply per word (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: integer multiplies are quarter rate
		const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
		c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0);
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	return c;
}
DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
enum { BLOCK_SURFACE = 0, BLOCK_SUN = 1, BLOCK_JITTER = 2 };
DEV float4 draws(const RenderParams& P, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t pass, uint32_t block) {
	if (pass > 0xFFFFu) pass = 0xFFFFu;
	uint4 r = philox4x32_10(make_uint4(pixel, sample, (depth << 16) | pass, block), P.seed_lo, P.seed_hi);
	return make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w));
}

// scene::camera::get_ray(ndc, ratio) — scene/camera.cpp:10-21
DEV void camera_get_ray(const DevScene& S, float ndc_x, float ndc_y, float ratio, V3& o, V3& d) {
	float dx = S.cam.tan_half_fov * ndc_x, dy = S.cam.tan_half_fov * ndc_y;
	dx *= ratio;
	V3 dir = normalize(mk(dx, dy, -1));
	// ray.transform(global): origin = basis*0 + origin, dir = normalize(basis*dir)
	V3 z = {0, 0, 0};
	o = mulmv(S.cam.basis, z) + mk(S.cam.origin[0], S.cam.origin[1], S.cam.origin[2]);
	d = normalize(mulmv(S.cam.basis, dir));
}
// pixel loop of renderer::render — core/renderer.cpp:359-370: jitter -> NDC -> camera ray
DEV void camera_ray(const DevScene& S, const RenderParams& P, uint32_t x, uint32_t y, uint32_t sample, V3& o, V3& d) {
	float4 j = draws(P, y * P.W + x, sample, 0, 0, BLOCK_JITTER);
	if (P.integrator == 1u && sample == 0) j.x = j.y = 0;   // HOST worker.cpp:125-126: the first sample is not offset
	float ndc_x = (((float)x + j.x) / (float)P.W) * 2 - 1;
	float ndc_y = (((float)y + j.y) / (float)P.H) * 2 - 1;
	ndc_y = -ndc_y;
	float ratio = (float)P.W / (float)P.H;
	camera_get_ray(S, ndc_x, ndc_y, ratio, o, d);
}

// ------------------------------------------------------------------------------------ textures
// image::image::read (image/image.cpp:124-141). LDR: byte / 255; colour channels of an sRGB image go through pow(v, 2.2) — here a
// 256-entry table of exactly those values. HDR (Radiance .hdr): the stored float as it is; pow(v, 2.2F) when the image was loaded as
// sRGB (powf of ocml here, of glibc there: last-place differences, the one lookup that is not bit-exact).
DEV float tex_chan(const DevScene& S, const TexRec& t, uint32_t px, uint32_t py, uint32_t ch) {
	const uint32_t c = t.c_srgb & 255u;
	const bool srgb = (t.c_srgb & kTexSrgb) != 0 && ch < 3;
	if (t.c_srgb & kTexFloat) {
		const float v = S.texels_f[t.offset + (py * t.w + px) * c + ch];
		return srgb ? powf(v, 2.2F) : v;
	}
	const uint32_t b = S.texels[t.offset + (py * t.w + px) * c + ch];
	return srgb ? S.srgb_lut[b] : (float)b / 255.0F;
}
// image_texture::read_pixel (image/image_texture.cpp:47-62): channels the image does not have stay 1
DEV float4 tex_pixel(const DevScene& S, const TexRec& t, uint32_t px, uint32_t py) {
	const uint32_t c = t.c_srgb & 255u;
	float4 v = make_float4(1, 1, 1, 1);
	v.x = tex_chan(S, t, px, py, 0);
	if (c >= 2) v.y = tex_chan(S, t, px, py, 1);
	if (c >= 3) v.z = tex_chan(S, t, px, py, 2);
	if (c >= 4) v.w = tex_chan(S, t, px, py, 3);
	return v;
}
// `uvec2(floor(x), ..)` in the reference is an implicit float -> unsigned conversion, compiled by g++ / x86-64 as a 64-bit
// truncation whose low word is kept: negative coordinates wrap modulo 2^32 before the modulo by the size (quirk Q3).
DEV uint32_t f2u_wrap(float f) { return (uint32_t)(long long)f; }
DEV uint32_t umod(uint32_t x, uint32_t y) { return (y + (x % y)) % y; }   // math::mod, integer branch
DEV float4 lerp4(float4 a, float4 b, float w) { return make_float4(lerpf(a.x, b.x, w), lerpf(a.y, b.y, w), lerpf(a.z, b.z, w), lerpf(a.w, b.w, w)); }
// image_texture::sample (image/image_texture.cpp:21-45): bilinear, four taps, unsigned-modulo wrap
DEV float4 tex_sample(const DevScene& S, int id, float u, float v) {
	const TexRec t = S.tex[id];
	const float cx = u * (float)t.w - 0.5F, cy = (1 - v) * (float)t.h - 0.5F;
	const float flx = floorf(cx), fly = floorf(cy);
	const uint32_t fx = umod(f2u_wrap(flx), t.w), fy = umod(f2u_wrap(fly), t.h);
	const uint32_t gx = umod(f2u_wrap(ceilf(cx)), t.w), gy = umod(f2u_wrap(ceilf(cy)), t.h);
	const float dx = cx - flx, dy = cy - fly;   // math::fract
	const float4 top = lerp4(tex_pixel(S, t, fx, fy), tex_pixel(S, t, gx, fy), dx);
	const float4 bot = lerp4(tex_pixel(S, t, fx, gy), tex_pixel(S, t, gx, gy), dx);
	return lerp4(top, bot, dy);
}

struct MatEval { V3 normal_ts, albedo, emissive10; float opacity, roughness, metallic; };
// core::material::get_normal / albedo / opacity / roughness / metallic / emissive (core/material.cpp:6-53).
// The reference samples the base-colour texture twice (albedo, opacity) and the metallic-roughness texture twice
// (G, B): same texture, same uv, same result — sampled once here.
template <bool TEX>
DEV MatEval material_eval(const DevScene& S, const MaterialRec& m, float u, float v) {
	MatEval e;
	e.normal_ts = mk(0, 0, 1);
	e.albedo = mk(m.albedo[0], m.albedo[1], m.albedo[2]);
	e.emissive10 = mk(m.emissive[0], m.emissive[1], m.emissive[2]);
	e.opacity = m.opacity; e.roughness = m.roughness; e.metallic = m.metallic;
	if constexpr (TEX) {
		if (m.tex[0] >= 0) { const float4 s = tex_sample(S, m.tex[0], u, v); e.normal_ts = mk(s.x, s.y, s.z) * 2 - mk(1, 1, 1); }
		if (m.tex[1] >= 0 || m.tex[2] >= 0) {
			const float4 s = tex_sample(S, m.tex[1] >= 0 ? m.tex[1] : m.tex[2], u, v);
			if (m.tex[1] >= 0) e.albedo = e.albedo * mk(s.x, s.y, s.z);
			if (m.tex[2] >= 0) e.opacity *= (m.tex[2] == m.tex[1] || m.tex[1] < 0) ? s.w : tex_sample(S, m.tex[2], u, v).w;
		}
		if (m.tex[4] >= 0 || m.tex[5] >= 0) {
			const float4 s = tex_sample(S, m.tex[4] >= 0 ? m.tex[4] : m.tex[5], u, v);
			if (m.tex[4] >= 0) e.roughness *= s.y;
			if (m.tex[5] >= 0) e.metallic *= (m.tex[5] == m.tex[4] || m.tex[4] < 0) ? s.z : tex_sample(S, m.tex[5], u, v).z;
		}
		if (m.tex[6] >= 0) { const float4 s = tex_sample(S, m.tex[6], u, v); e.emissive10 = e.emissive10 * mk(s.x, s.y, s.z); }
	}
	e.emissive10 = e.emissive10 * 10;   // get_emissive(uv) * 10 (renderer.cpp:462)
	return e;
}

// ------------------------------------------------------------------------------------ one path vertex
// The shading phase never traverses. What the reference does with a second intersect() call from inside trace() becomes a
// record handed to the next sweep of the wave's stream:
//   * opacity / lit-shadow-catcher pass-through (renderer.cpp:466-472, 513-519): the same path, same depth, pass + 1,
//     continued from behind the surface — an ordinary entry of the outgoing ray stream;
//   * the sun sample (renderer.cpp:498-564): a SHADOW REQUEST (ray + what to do with the answer), resolved by the
//     any-hit sweep that follows the shading sweep.
// Vertex outcomes:
enum : int {
	V_DEAD = 0,      // path ends; L is final unless a request of kind REQ_ADD is pending for it
	V_ALIVE = 1,     // (o, d, T, L, depth, pass) updated: next stream entry
	V_PENDING = 2,   // shadow catcher: the request's answer decides between pass-through and the end of the path
};
enum : uint32_t { REQ_NONE = 0, REQ_ADD = 1, REQ_CATCHER = 2 };
struct ShadowReq {
	uint32_t kind;
	V3 o, d;     // the shadow ray
	V3 x;        // REQ_ADD: T * direct_out, added to the path's radiance when unoccluded; REQ_CATCHER: origin of the pass-through ray
};

// PTX_INTEGRATOR_LIB — renderer::trace (core/renderer.cpp:437-643) in iterative throughput form (DESIGN.md "Estimator"):
//   L += T * (direct + emissive);  T *= clamp(brdf / max(pdf, eps), 0, 1);  next ray.
// PTX_INTEGRATOR_WORKER — one vertex of the HOST worker's stage pipeline: INTERSECT's sun sample
// (src/processors/worker/intersection_worker.cpp:22-39), SHADING (shading_worker.cpp:27-199); `L` is cloud_ray::color,
// `T` cloud_ray::scale, `depth` = bounce_count - cloud_ray::bounce.
// `h` is the closest hit of (o, d) found by the extend sweep. SUN / ALPHA compile the request / pass-through code in.
template <bool SUN, bool ALPHA, bool TEX, bool WORKER>
DEV int shade_vertex(const DevScene& S, const ShadeRec* shade, const RenderParams& P, uint32_t pixel, uint32_t sample,
                     uint32_t& depth, uint32_t& pass, SceneHit h, V3& o, V3& d, V3& T, V3& L, ShadowReq& rq) {
	rq.kind = REQ_NONE;
	if (h.surface < 0) {   // miss: environment_factor, times the environment map when one is set (renderer.cpp:443-451, shading_worker.cpp:28-41)
		V3 env = mk(P.env[0], P.env[1], P.env[2]);
		if constexpr (TEX) {
			if (S.env_tex >= 0) {   // core::equirectangular_proj (core/utils.hpp:22-27) of ray::get_dir()
				const float4 e = tex_sample(S, S.env_tex, atan2f(d.z, d.x) * 0.1591F + 0.5F, asinf(d.y) * 0.3183F + 0.5F);
				env = mk(e.x, e.y, e.z) * env;
			}
		}
		L = L + T * env;
		return V_DEAD;
	}
	const ShadeRec& R = shade[h.surface];
	Surf sf;
	hit_attributes(S, R, h.tri, h.b1, h.b2, sf);
	const MaterialRec& mt = R.mat;
	const MatEval me = material_eval<TEX>(S, mt, sf.u, sf.v);   // renderer.cpp:458-462
	float roughness = me.roughness;
	const float4 rnd = draws(P, pixel, sample, depth, pass, BLOCK_SURFACE);  // x opacity, y lobe, z/w BSDF sample
	if constexpr (WORKER) L = L + T * me.emissive10;                         // shading_worker.cpp:52 — before the opacity test

	if constexpr (ALPHA) {
		const bool transparent = !(me.opacity == 1.0f || fabsf(me.opacity - 1.0f) < kEps) && rnd.x > me.opacity;   // renderer.cpp:466-472
		if (transparent) {
			o = sf.pos + d * kEps;
			d = normalize(d);
			pass++;
			return pass > 4096 ? V_DEAD : V_ALIVE;   // safety bound; the reference would recurse / re-queue without limit
		}
	}
	const V3 normal = shading_normal(sf, me.normal_ts), outcoming = -d;
	if (dot(normal, outcoming) <= 0) return V_DEAD;                          // renderer.cpp:478-479: black, path ends
	roughness = pmax(roughness, 0.05F);
	float spec_prob = fresnel_schlick(outcoming, reflect3(-outcoming, normal), mt.ior);
	spec_prob = pmax(spec_prob, me.metallic);

	V3 direct_out = mk(0, 0, 0);   // LIB without a request: stays 0
	if constexpr (SUN) {
		const bool catcher = ALPHA && mt.shadow_catcher && depth == 0;
		bool sampled = false;
		V3 din = mk(0, 0, 0);
		if (S.sun.present) {                                                  // renderer.cpp:498-509 / intersection_worker.cpp:22-39
			const float4 sr = draws(P, pixel, sample, depth, pass, BLOCK_SUN);
			V3 c = mulmv(S.sun.basis, mk(0, 0, 1));
			c = rand_cone_vec(sr.x, cosf(sr.y * S.sun.angular_radius), c);
			const V3 cn = normalize(c);                                       // ray::get_dir()
			din = WORKER ? cn : c;                                            // the worker shades with the ray's direction, trace() with the sample
			sampled = dot(normal, c) > 0 && (!WORKER || dot(normal, cn) > 0);
			if (sampled) { rq.o = sf.pos + c * kEps; rq.d = cn; }
		}
		if (WORKER && catcher && !sampled) { L = mk(0, 0, 0); return V_DEAD; }   // shading_worker.cpp:74-94: in_shadow stays true
		if (sampled) {
			if (catcher) {                                                    // lit: transparent; shadowed: black (renderer.cpp:513-519,560-561)
				rq.kind = REQ_CATCHER;
				rq.x = sf.pos + d * kEps;
				return V_PENDING;
			}
			float pdf_unused;
			const V3 brdf = eval_brdf(normal, outcoming, din, me.albedo, roughness, me.metallic, spec_prob, pdf_unused);
			const V3 e = mk(S.sun.energy[0], S.sun.energy[1], S.sun.energy[2]);
			const float pdf = lerpf(1.0f, 1.0f, spec_prob);
			const V3 v = brdf * e / pmax(pdf, kEps);
			direct_out = mk(clampf(v.x, 0, e.x), clampf(v.y, 0, e.y), clampf(v.z, 0, e.z));
			rq.kind = REQ_ADD;
			rq.x = T * direct_out;
		}
	}
	const V3 inc = importance_sample(rnd.y < spec_prob, rnd.z, rnd.w, normal, outcoming, roughness);
	if constexpr (!WORKER) L = L + T * me.emissive10;
	if (!(dot(normal, inc) > 0)) return V_DEAD;                              // renderer.cpp:578 / shading_worker.cpp:154,196-199
	float pdf;
	const V3 brdf = eval_brdf(normal, outcoming, inc, me.albedo, roughness, me.metallic, spec_prob, pdf);
	const float ip = pmax(pdf, kEps);
	if constexpr (!WORKER) {
		T = T * mk(clampf(brdf.x / ip, 0, 1), clampf(brdf.y / ip, 0, 1), clampf(brdf.z / ip, 0, 1));  // renderer.cpp:617-620
	} else {
		T = T * mk(brdf.x / ip, brdf.y / ip, brdf.z / ip);                                           // shading_worker.cpp:173
		T = mk(clampf(T.x, 0, 10.0f), clampf(T.y, 0, 10.0f), clampf(T.z, 0, 10.0f));                 // :175
	}
	o = sf.pos + inc * kEps;
	d = normalize(inc);
	if constexpr (WORKER) {
		if ((int)(P.bounces - depth) < (int)P.bounces - 2) {                 // :182-190, Russian roulette
			const float4 sr = draws(P, pixel, sample, depth, pass, BLOCK_SUN);   // lane z
			const float p = pmax(T.x, pmax(T.y, T.z));
			if (sr.z > p) return V_DEAD;
			T = mk(T.x / p, T.y / p, T.z / p);
		}
	}
	depth++;
	pass = 0;
	return depth == P.bounces ? V_DEAD : V_ALIVE;                            // trace(0, ..) is black: renderer.cpp:438-439; bounce > 0: shading_worker.cpp:193
}

// ------------------------------------------------------------------------------------ LDS staging
struct Staged { Geoms g; const ShadeRec* shade; const float4* hot; };

template <int MODE>
DEV Staged stage_geometry(const DevScene& S, unsigned char* smem) {
	const Geom glb = {S.nodes, S.refs, S.tri_isect, S.glb_leaf_ordered != 0, true};   // tri_isect: records of ALL surfaces, per leaf reference or per triangle (upload_scene)
	if constexpr (MODE == MODE_GLOBAL) return {{glb, glb}, S.shade, nullptr};
	else {
		// the resident arrays (all surfaces in MODE_LDS, the ones that fit in MODE_HYBRID):
		// [triangle records][shade records][KD nodes][leaf refs][hot hit records], each region a multiple of 16 B
		uint4* dst = reinterpret_cast<uint4*>(smem);
		const uint32_t n_tri16 = S.n_res_tris * 3, n_shade16 = S.n_surfaces * (uint32_t)(sizeof(ShadeRec) / 16), n_node16 = (S.n_res_nodes + 1) / 2, n_ref16 = (S.n_res_refs + 3) / 4;
		const uint4* src_t = reinterpret_cast<const uint4*>(S.res_tris);
		const uint4* src_s = reinterpret_cast<const uint4*>(S.shade);
		const uint4* src_n = reinterpret_cast<const uint4*>(S.res_nodes);
		const uint4* src_r = reinterpret_cast<const uint4*>(S.res_refs);
		uint4* d_s = dst + n_tri16;
		uint4* d_n = d_s + n_shade16;
		uint4* d_r = d_n + n_node16;
		uint4* d_h = d_r + n_ref16;
		const uint32_t n_hot16 = S.n_hot * 9u;
		const uint4* src_h = reinterpret_cast<const uint4*>(S.hot_hitrec);
		for (uint32_t i = threadIdx.x; i < n_tri16; i += blockDim.x) dst[i] = src_t[i];
		for (uint32_t i = threadIdx.x; i < n_shade16; i += blockDim.x) d_s[i] = src_s[i];
		for (uint32_t i = threadIdx.x; i < n_node16; i += blockDim.x) d_n[i] = src_n[i];
		for (uint32_t i = threadIdx.x; i < n_ref16; i += blockDim.x) d_r[i] = src_r[i];
		for (uint32_t i = threadIdx.x; i < n_hot16; i += blockDim.x) d_h[i] = src_h[i];
		__syncthreads();
		const Geom lds = {reinterpret_cast<const uint2*>(d_n), reinterpret_cast<const uint32_t*>(d_r), reinterpret_cast<const float4*>(dst), false, false};
		return {{lds, glb}, reinterpret_cast<const ShadeRec*>(d_s), S.n_hot ? reinterpret_cast<const float4*>(d_h) : nullptr};
	}
}

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

// Wave-private stream arrays. Each entry is written once and read once or twice a whole sweep later, by which time the chip's
// 16 K waves have pushed ~1.6 GB through the caches: the accesses carry the non-temporal hint, so that they do
// not displace the scene's nodes and triangle records from L2 / Infinity Cache. Same-box A/B: +1.5 % on Cornell 1080p, neutral to
// +2 % on the large-mesh scenes, images bit-identical (profiles/round2_ab_nt_streams.txt); -DPTX_PLAIN_STREAMS builds the plain form.
#ifndef PTX_PLAIN_STREAMS
typedef float f4n __attribute__((ext_vector_type(4)));
typedef float f2n __attribute__((ext_vector_type(2)));
struct Q4Ref {
	float4* p;
	DEV operator float4() const { const f4n v = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
	DEV void operator=(float4 v) const { const f4n w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4n*>(p)); }
	DEV void operator=(const Q4Ref& o) const { *this = (float4)o; }
};
struct Q4 {
	float4* p;
	DEV Q4Ref operator[](size_t i) const { return {p + i}; }
	DEV Q4 operator+(size_t k) const { return {p + k}; }
};
struct Q2Ref {
	float2* p;
	DEV operator float2() const { const f2n v = __builtin_nontemporal_load(reinterpret_cast<const f2n*>(p)); return make_float2(v.x, v.y); }
	DEV void operator=(float2 v) const { const f2n w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<f2n*>(p)); }
};
struct Q2 {
	float2* p;
	DEV Q2Ref operator[](size_t i) const { return {p + i}; }
};
DEV float4* raw(Q4 q) { return q.p; }
DEV float2* raw(Q2 q) { return q.p; }
#else
typedef float4* Q4;
typedef float2* Q2;
DEV float4* raw(Q4 q) { return q; }
DEV float2* raw(Q2 q) { return q; }
#endif

// ------------------------------------------------------------------------------------ batch intersect outputs
// What ptx_intersect_batch reports per ray: the fields of renderer::intersect's result (core/renderer.cpp:671-725)
DEV void write_hit_outputs(const DevScene& S, const ShadeRec* shade, const IntersectArgs& A, size_t i, bool hit, const SceneHit& h) {
	A.distance[i] = hit ? h.dist : -1.0f;
	A.surface[i] = hit ? h.surface : -1;
	A.triangle[i] = hit ? (int32_t)((h.tri & S.tri_id_mask) - S.surfaces[h.surface].tri_base) : -1;
	A.b0[i] = hit ? 1 - h.b1 - h.b2 : 0.f; A.b1[i] = hit ? h.b1 : 0.f; A.b2[i] = hit ? h.b2 : 0.f;
	if (A.px || A.nx || A.u) {
		Surf sf = {};
		V3 sn = {0, 0, 0};
		if (hit) {
			const ShadeRec& R = shade[h.surface];
			hit_attributes(S, R, h.tri, h.b1, h.b2, sf);
			sn = shading_normal(sf, material_eval<true>(S, R.mat, sf.u, sf.v).normal_ts);
		}
		if (A.px) { A.px[i] = sf.pos.x; A.py[i] = sf.pos.y; A.pz[i] = sf.pos.z; }
		if (A.nx) { A.nx[i] = sn.x; A.ny[i] = sn.y; A.nz[i] = sn.z; }
		if (A.u) { A.u[i] = sf.u; A.v[i] = sf.v; }
	}
}

}  // namespace ptx
