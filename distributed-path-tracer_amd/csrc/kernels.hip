// HIP kernels for gfx950 (MI355X / CDNA4): the ray-intersection + Monte-Carlo shading hot path.
//
// Design: one ray per lane, 64-lane waves. A launch is a grid of persistent
// workgroups (one per CU) that first stage the traversal data that fits — 8-byte KD nodes, 4-byte leaf
// references, 48-byte triangle records of the whole scene, or of its small surfaces while large meshes
// stay in L2/HBM — into the CU's LDS, then each WAVE pulls chunks of camera paths from a global counter
// and runs them to the end by itself: the paths of a chunk live in a wave-private SoA-of-float4 stream in
// global memory (coalesced 1-KiB loads per 64 rays); every step sweeps the stream three times — EXTEND
// (closest hits; rarely entered models set aside and traversed afterwards with full waves), SHADE
// (BSDF, next ray, shadow requests; survivors written back compacted with a wave ballot + lane prefix
// count, so waves stay full while paths die) and SHADOW (any-hit sweep over the requests). No workgroup
// barrier after staging, no inter-workgroup traffic, no atomics besides one counter fetch per chunk;
// per-sample radiance leaves through plain stores and a second tiny kernel adds the samples of each
// pixel in a fixed order (bitwise reproducible, no float atomics).
//
// Numerics: IEEE binary32 in the reference's operation order, no FMA contraction (-ffp-contract=off),
// correctly rounded divide / sqrt, the reference's double-precision islands kept in double. Each device
// function cites what it restates (paths relative to path-tracer-core/path_tracer_lib/path_tracer/).
#include "device_core.hpp"

namespace ptx {

// PTX_CLK builds (tools/build_variant.sh clk -DPTX_CLK): where a wave's time goes, by s_memtime around regions that are closed with a
// forced s_waitcnt — how long the wave itself sits on each class of memory access (SQ_WAIT_ANY says only that it waits). Lane 0 of
// every wave adds its clocks (kilocycles) into the counters block; ptx_render prints them. Not compiled into the product.
#ifdef PTX_CLK
#define CLK_T0() const uint64_t clk_t0_ = __builtin_amdgcn_s_memtime()
#define CLK_T1(k) do { clk[k] += __builtin_amdgcn_s_memtime() - clk_t0_; } while (0)
#define CLK_WAIT() __builtin_amdgcn_s_waitcnt(0)
#else
#define CLK_T0() do { } while (0)
#define CLK_T1(k) do { } while (0)
#define CLK_WAIT() do { } while (0)
#endif

// ------------------------------------------------------------------------------------ integrator kernel
// One launch = `P.n_paths` camera paths (P.pass_spp samples of every tile pixel), all bounces.
// Per wave and chunk of up to kChunk paths, every step is up to three sweeps over the wave's private stream:
//   EXTEND: (generate or) load ray -> closest hit -> 16-byte hit record        (traversal state only in registers)
//   SHADE : load ray + hit + path state -> BSDF, radiance, next ray, shadow request -> compacted writes (no traversal)
//   SHADOW: load request -> any hit? -> add the sun contribution / resolve a pending shadow catcher   (SUN variants)
template <int MODE, bool SUN, bool ALPHA, bool TEX, bool WORKER, bool SURF>
__global__ void __launch_bounds__(kBlock) k_render_pass(DevScene S0, RenderParams P, PassBuffers B, const ModelRec* __restrict__ t_models, const SurfaceRec* __restrict__ t_surfaces, const SpaceRec* __restrict__ t_spaces, const uint32_t* __restrict__ t_model_space) {
	DevScene S = S0;
	S.models = t_models; S.surfaces = t_surfaces; S.spaces = t_spaces; S.model_space = t_model_space;  // see struct Tables
	const Staged st = stage_geometry<MODE>(S, g_smem);
	S.hot_lds = st.hot;
	const Geoms g = st.g;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave_slot = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	// wave-private streams: 2 ray buffers x 4 float4 arrays x kChunk entries, then 1 hit array
	const Q4 qbase = Q4{B.queues + (size_t)wave_slot * B.queue_stride};
	const Q4 hbuf = qbase + 2u * 4u * kChunk;
	const Q2 hdist = Q2{reinterpret_cast<float2*>(raw(hbuf) + kChunk)};               // [kChunk] (world distance, local t) of the current best hit (deferral)
	const Q4 sreq = hbuf + (kChunk + kChunk / 2u);                              // [3][kChunk] shadow requests: (origin, target) (dir, kind) (x, path id)
	const Q4 lists = sreq + 3u * kChunk;                                      // [units][2][kListCap]: (local origin, ray index), (local dir, -); units = models, or surfaces (SURF)
	const Spill spill{B.spill + (size_t)wave_slot * (kSpillStack * 64) + lane};
	uint32_t rays = 0;
#ifdef PTX_PROF
	Prof prof{};
#endif
#ifdef PTX_CLK
	uint64_t clk[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 kernel, 1 chunk fetch, 2 EXTEND entry loads, 3 EXTEND sweep (rest), 4 set-aside lists, 5 SHADE entry + hit loads, 6 SHADE hit-record gathers, 7 SHADE rest
	const uint64_t clk_start = __builtin_amdgcn_s_memtime();
#endif

	for (;;) {
		// Guided self-scheduling: full chunks while plenty of paths are left, then chunks that shrink with what remains (a
		// multiple of 64, at least 128), so that the waves of the launch run dry together instead of one chunk-time apart.
		uint32_t first_lo = 0, first_hi = 0, take = 0;
		CLK_T0();
		if (lane == 0) {
			const unsigned long long seen = __hip_atomic_load(B.chunk_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned long long left = seen < P.n_paths ? P.n_paths - seen : 0ull;
			const unsigned long long n_waves = (unsigned long long)gridDim.x * (kBlock / 64);
			// shrinking starts in the last round only (short chunks defer less efficiently), and only in launches of few
			// rounds: over 16 rounds the uneven end is < 3 % of the launch and full chunks measured 1.5 % faster
			unsigned long long want = P.n_paths >= 16ull * n_waves * kChunk ? (unsigned long long)kChunk : left / n_waves;
			want = (want + 63ull) & ~63ull;
			take = (uint32_t)(want < 128ull ? 128ull : (want > (unsigned long long)kChunk ? (unsigned long long)kChunk : want));
			const unsigned long long f = atomicAdd(B.chunk_counter, (unsigned long long)take);
			first_lo = (uint32_t)f; first_hi = (uint32_t)(f >> 32);
		}
		first_lo = __builtin_amdgcn_readfirstlane(first_lo);
		first_hi = __builtin_amdgcn_readfirstlane(first_hi);
		take = __builtin_amdgcn_readfirstlane(take);
		CLK_T1(1);
		const uint64_t first = ((uint64_t)first_hi << 32) | first_lo;
		if (first >= P.n_paths) break;
		uint32_t n_in = (uint32_t)((P.n_paths - first) < (uint64_t)take ? (P.n_paths - first) : take);
		if (P.bounces == 0)   // trace(0, ..) is black with alpha 1 (renderer.cpp:438-439): no vertex ever stores the sample
			for (uint32_t i = lane; i < n_in; i += 64) B.sample_rad[(uint32_t)first + i] = make_float4(0.f, 0.f, 0.f, 1.0f);

		// One step = every live path of the chunk advances by one stream entry: a scatter (depth + 1) or, with ALPHA, a
		// pass-through (same depth, pass + 1). Without pass-through all paths of a step have depth == step.
		for (uint32_t step = 0; P.bounces > 0 && n_in > 0; step++) {
			const Q4 qin = qbase + (size_t)(step & 1u) * (4u * kChunk);
			const Q4 qout = qbase + (size_t)((step + 1u) & 1u) * (4u * kChunk);

			// ---------------- EXTEND
			// lists live in the unused part of the wave's stream area: hdist (kChunk words) + n_models lists
			const bool defer = SURF ? (S.n_surfaces >= 1 && S.n_surfaces <= (uint32_t)kMaxDeferModels)
			                        : (S.n_models > 1 && S.n_models <= kMaxDeferModels);   // the list lengths live in the 64 lanes of one VGPR
			uint32_t list_len = 0;   // lane u: entries in the deferred list of unit u (model, or surface in SURF kernels)
#ifdef PTX_CLK
			const uint64_t clk_sweep0 = __builtin_amdgcn_s_memtime();
			const uint64_t clk_loads0 = clk[2];
#endif
			for (uint32_t base = 0; base < n_in; base += 64) {
				const uint32_t i = base + lane;
				const bool active = i < n_in;
				V3 o = {0, 0, 0}, d = {0, 0, 1};
				if (active) PROF(0);
				if (active) {
					if (step == 0) {
						const uint32_t id = (uint32_t)first + i;  // id within the pass: sample-major, pixel-minor
						const uint32_t s_local = id / P.n_pixels;
						uint32_t p_local = id - s_local * P.n_pixels;
						if (P.pixels) p_local = P.pixels[p_local];   // interleaved tile sharding: the pass renders a subset of the tile's pixels
						const uint32_t px = P.x0 + p_local % P.w, py = P.y0 + p_local / P.w;
						camera_ray(S, P, px, py, P.sample0 + s_local, o, d);
						// a fresh path: T = 1, L = 0, depth = pass = 0; its RNG key (pixel, sample) travels with it (no divisions per vertex)
						qin[i] = make_float4(o.x, o.y, o.z, __uint_as_float(id));
						qin[kChunk + i] = make_float4(d.x, d.y, d.z, 1.0f);
						qin[2 * kChunk + i] = make_float4(1.0f, 1.0f, 0.f, 0.f);
						qin[3 * kChunk + i] = make_float4(0.f, __uint_as_float(0u), __uint_as_float(py * P.W + px), __uint_as_float(P.sample0 + s_local));
					} else {
						float4 q0 = qin[i], q1 = qin[kChunk + i];
						o = mk(q0.x, q0.y, q0.z);
						d = mk(q1.x, q1.y, q1.z);
					}
				}
#ifdef PTX_CLK
				if (step > 0) { CLK_T0(); CLK_WAIT(); CLK_T1(2); }   // the wave sits here until its 64 stream entries have arrived
#endif
				SceneHit h;
				if (!defer) { if (active) scene_traverse<MODE>(S, g, o, d, h, spill); }
				else if constexpr (SURF) {
					// renderer::intersect's model loop / model::intersect's surface loop with the rarely entered SURFACES set aside
					Best bh;
					best_reset(bh);
					uint32_t cur_space = 0xFFFFFFFFu;
					V3 lo = o, ld = d, inv = d;
					for (int m = 0; m < S.n_models; m++) {
						const ModelRec& M = S.models[m];
						const uint32_t spc = S.model_space[m];
						if (active) PROF(1);
						if (spc != cur_space) {
							if (active) PROF(2);
							const SpaceRec& SP = S.spaces[spc];
							lo = mulmv(SP.inv_basis, o) + mk(SP.inv_origin[0], SP.inv_origin[1], SP.inv_origin[2]);
							ld = normalize(mulmv(SP.inv_basis, d));
							inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
							cur_space = spc;
						}
						float nr, fr;
						const bool enters = active && aabb_test_inv(M.bmin, M.bmax, lo, inv, nr, fr);   // model box first (model.cpp:38-40)
						if (__ballot(enters) == 0) continue;
						for (int k = 0; k < M.n_surfaces; k++) {
							const int u = M.first_surface + k;
							const SurfaceRec& sf = S.surfaces[u];
							const bool ent = enters && aabb_test_inv(sf.bmin, sf.bmax, lo, inv, nr, fr);
							const uint64_t em = __ballot(ent);
							if (em == 0) continue;
							const uint32_t cnt = (uint32_t)__popcll(em);
							if (cnt >= kInlineMin) {
								if (ent) {
									PROF(3);
									MeshHit mh;
									if (mesh_traverse_m<MODE, 4>(g, sf, nr, fr, lo, ld, mh, spill PROF_PASS)) best_offer(bh, M, m, ld, u, mh);
								}
							} else {
								const uint32_t len = __builtin_amdgcn_readlane(list_len, u);
								if (ent) {
									PROF(14);
									const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
									const Q4 L0 = lists + (size_t)u * 2u * kListCap;
									L0[len + r] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(i));
									L0[kListCap + len + r] = make_float4(ld.x, ld.y, ld.z, 0.f);
								}
								list_len = (int)lane == u ? len + cnt : list_len;
							}
						}
					}
					h.dist = bh.wd; h.surface = bh.surf; h.tri = bh.tri; h.b1 = bh.b1; h.b2 = bh.b2;
					if (active) hdist[i] = make_float2(bh.wd, bh.tl);
				} else {
					// renderer::intersect's model loop with the rarely entered models set aside
					h.dist = -1.0f; h.surface = -1; h.tri = 0; h.b1 = 0; h.b2 = 0;
					uint32_t cur_space = 0xFFFFFFFFu;
					V3 lo = o, ld = d, inv = d;
					for (int m = 0; m < S.n_models; m++) {
						const ModelRec& M = S.models[m];
						const uint32_t spc = S.model_space[m];
						if (active) PROF(1);
						if (spc != cur_space) {
							if (active) PROF(2);
							const SpaceRec& SP = S.spaces[spc];
							lo = mulmv(SP.inv_basis, o) + mk(SP.inv_origin[0], SP.inv_origin[1], SP.inv_origin[2]);
							ld = normalize(mulmv(SP.inv_basis, d));
							inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
							cur_space = spc;
						}
						float nr, fr;
						const bool enters = active && aabb_test_inv(M.bmin, M.bmax, lo, inv, nr, fr);
						const uint64_t em = __ballot(enters);
						if (em == 0) continue;
						const uint32_t cnt = (uint32_t)__popcll(em);
						if (cnt >= kInlineMin) {
							if (enters) {
								PROF(3);
								float wd, b1, b2; int surf; uint32_t tri;
								if (model_traverse<MODE, 4>(S, g, M, lo, ld, inv, wd, surf, tri, b1, b2, spill PROF_PASS) &&
								    (wd < h.dist || !(h.dist >= 0) || (wd == h.dist && surf < h.surface))) { h.dist = wd; h.surface = surf; h.tri = tri; h.b1 = b1; h.b2 = b2; }
							}
						} else {
							const uint32_t len = __builtin_amdgcn_readlane(list_len, m);
							if (enters) {
								PROF(14);
								const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
								// the deferred sweep gets the LOCAL ray: no gather of the stream entry, no second transform
								const Q4 L0 = lists + (size_t)m * 2u * kListCap;
								L0[len + r] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(i));
								L0[kListCap + len + r] = make_float4(ld.x, ld.y, ld.z, 0.f);
							}
							list_len = (int)lane == m ? len + cnt : list_len;
						}
					}
				}
				if (active) { hbuf[i] = make_float4(__int_as_float(h.surface), __uint_as_float(h.tri), h.b1, h.b2); if (defer && !SURF) raw(hdist)[i].x = h.dist; }
			}
#ifdef PTX_CLK
			CLK_WAIT();
			clk[3] += (__builtin_amdgcn_s_memtime() - clk_sweep0) - (clk[2] - clk_loads0);
			const uint64_t clk_lists0 = __builtin_amdgcn_s_memtime();
#endif
			if (defer) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				if constexpr (SURF) {
					// the lists, surface by surface, with full waves
					for (int u = 0; u < (int)S.n_surfaces; u++) {
						const uint32_t len = __builtin_amdgcn_readlane(list_len, u);
						if (len == 0) continue;
						const SurfaceRec& sf = S.surfaces[u];
						const int m = (int)sf.model;
						const ModelRec& M = S.models[m];
						const Q4 L0 = lists + (size_t)u * 2u * kListCap;
						for (uint32_t base = 0; base < len; base += 64) {
							if (base + lane < len) {
								PROF(8);
								const float4 e0 = L0[base + lane], e1 = L0[kListCap + base + lane];
								const uint32_t i = __float_as_uint(e0.w);
								const float4 hr = hbuf[i];                       // current best of this ray: issued before the traversal
								const float2 hd = hdist[i];
								const V3 lo = mk(e0.x, e0.y, e0.z), ld = mk(e1.x, e1.y, e1.z);
								const V3 inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
								MeshHit mh;
								if (!cannot_win(sf.pbmin, M.basis, lo, ld, inv, hd.x) && mesh_intersect_m<MODE, 10>(g, sf, lo, ld, inv, mh, spill PROF_PASS)) {
									Best b;
									b.surf = __float_as_int(hr.x); b.tri = __float_as_uint(hr.y); b.b1 = hr.z; b.b2 = hr.w; b.wd = hd.x; b.tl = hd.y;
									b.model = b.surf >= 0 ? (int)S.surfaces[b.surf].model : -1;   // per-lane table read, only on a hit
									if (best_offer(b, M, m, ld, u, mh)) {
										hbuf[i] = make_float4(__int_as_float(b.surf), __uint_as_float(b.tri), b.b1, b.b2);
										hdist[i] = make_float2(b.wd, b.tl);
									}
								}
							}
						}
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					}
				} else
				// the lists, model by model, with full waves
				for (int m = 0; m < S.n_models; m++) {
					const uint32_t len = __builtin_amdgcn_readlane(list_len, m);
					if (len == 0) continue;
					const ModelRec& M = S.models[m];
					const Q4 L0 = lists + (size_t)m * 2u * kListCap;
					for (uint32_t base = 0; base < len; base += 64) {
						if (base + lane < len) {
							PROF(8);
							const float4 e0 = L0[base + lane], e1 = L0[kListCap + base + lane];
							const uint32_t i = __float_as_uint(e0.w);
							const float bd = raw(hdist)[i].x;                     // current best of this ray: issued before the traversal
							const int bs = __float_as_int(raw(hbuf)[i].x);
							const V3 lo = mk(e0.x, e0.y, e0.z), ld = mk(e1.x, e1.y, e1.z);
							const V3 inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
							float wd, b1, b2; int surf; uint32_t tri;
							if (!cannot_win(M.pbmin, M.basis, lo, ld, inv, bd) &&
							    model_traverse<MODE, 10>(S, g, M, lo, ld, inv, wd, surf, tri, b1, b2, spill PROF_PASS) &&
							    (wd < bd || !(bd >= 0) || (wd == bd && surf < bs))) {
								hbuf[i] = make_float4(__int_as_float(surf), __uint_as_float(tri), b1, b2);
								raw(hdist)[i].x = wd;
							}
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				}
			}
#ifdef PTX_CLK
			CLK_WAIT();
			clk[4] += __builtin_amdgcn_s_memtime() - clk_lists0;
			const uint64_t clk_shade0 = __builtin_amdgcn_s_memtime();
			const uint64_t clk_sl0 = clk[5] + clk[6];
#endif
			rays += n_in > lane ? (n_in - lane + 63u) / 64u : 0u;  // rays this lane traced in the sweep
			// the wave re-reads below what other lanes of this wave just wrote
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

			// ---------------- SHADE + wave-level stream compaction (rays and shadow requests)
			uint32_t n_out = 0, n_sh = 0;
			for (uint32_t base = 0; base < n_in; base += 64) {
				const uint32_t i = base + lane;
				const bool active = i < n_in;
				V3 o = {0, 0, 0}, d = {0, 0, 1}, T = {1, 1, 1}, L = {0, 0, 0};
				uint32_t id = 0, depth = ALPHA ? 0u : step, pass = 0;
				float key_px = 0.f, key_s = 0.f;   // RNG key words, carried as raw bits
				int state = V_DEAD;
				ShadowReq rq;
				rq.kind = REQ_NONE;
				if (active) {
					PROF(9);
					const float4 q0 = qin[i], q1 = qin[kChunk + i], q2 = qin[2 * kChunk + i], q3 = qin[3 * kChunk + i], hq = hbuf[i];
#ifdef PTX_CLK
					{ CLK_T0(); CLK_WAIT(); CLK_T1(5); }
					{   // the hit record's nine 16-byte pieces, fetched here so that the wait for them has a region of its own (shade_vertex then finds them in L1)
						CLK_T0();
						float acc = 0;
						if (__float_as_int(hq.x) >= 0 && !(S.hot_lds && S.tri_id_mask == 0x00FFFFFFu && (__float_as_uint(hq.y) >> 24) != 0xFFu)) { const float4* Hh = S.tris + 9 * (size_t)(__float_as_uint(hq.y) & S.tri_id_mask);   // (hot records come from LDS inside shade_vertex)
						 for (int k = 0; k < 9; k++) acc += Hh[k].w; }
						CLK_WAIT();
						if (acc == 1.2345e-30f) rays++;   // keeps the fetches alive
						CLK_T1(6);
					}
#endif
					o = mk(q0.x, q0.y, q0.z); id = __float_as_uint(q0.w);
					d = mk(q1.x, q1.y, q1.z);
					T = mk(q1.w, q2.x, q2.y);
					L = mk(q2.z, q2.w, q3.x);
					key_px = q3.z; key_s = q3.w;
					if constexpr (ALPHA) { const uint32_t dp = __float_as_uint(q3.y); depth = dp >> 16; pass = dp & 0xFFFFu; }
					SceneHit h;
					h.dist = 0; h.surface = __float_as_int(hq.x); h.tri = __float_as_uint(hq.y); h.b1 = hq.z; h.b2 = hq.w;
					state = shade_vertex<SUN, ALPHA, TEX, WORKER>(S, st.shade, P, __float_as_uint(key_px), __float_as_uint(key_s), depth, pass, h, o, d, T, L, rq);
					if (state == V_DEAD) B.sample_rad[id] = make_float4(L.x, L.y, L.z, 1.0f);
				}
				const bool alive = state == V_ALIVE;
				const uint64_t mask = __ballot(alive);
				const uint32_t pos = n_out + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
				if (alive) {
					qout[pos] = make_float4(o.x, o.y, o.z, __uint_as_float(id));
					qout[kChunk + pos] = make_float4(d.x, d.y, d.z, T.x);
					qout[2 * kChunk + pos] = make_float4(T.y, T.z, L.x, L.y);
					qout[3 * kChunk + pos] = make_float4(L.z, __uint_as_float((depth << 16) | pass), key_px, key_s);
				}
				n_out += (uint32_t)__popcll(mask);
				if constexpr (SUN) {
					const bool want = rq.kind != REQ_NONE;
					const uint64_t sm = __ballot(want);
					if (want) {
						const uint32_t sp = n_sh + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
						// where the answer goes: the path's new stream entry, its final sample (0xFFFFFFFF), or — shadow catcher — its current entry
						const uint32_t target = rq.kind == REQ_CATCHER ? i : (alive ? pos : 0xFFFFFFFFu);
						sreq[sp] = make_float4(rq.o.x, rq.o.y, rq.o.z, __uint_as_float(target));
						sreq[kChunk + sp] = make_float4(rq.d.x, rq.d.y, rq.d.z, __uint_as_float(rq.kind));
						sreq[2 * kChunk + sp] = make_float4(rq.x.x, rq.x.y, rq.x.z, __uint_as_float(id));
						if (rq.kind == REQ_CATCHER) {   // the pass-through entry is built from the current one: make its path state complete
							qin[2 * kChunk + i] = make_float4(T.y, T.z, L.x, L.y);
							qin[3 * kChunk + i] = make_float4(L.z, __uint_as_float((depth << 16) | pass), key_px, key_s);
						}
					}
					n_sh += (uint32_t)__popcll(sm);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef PTX_CLK
			CLK_WAIT();
			clk[7] += (__builtin_amdgcn_s_memtime() - clk_shade0) - (clk[5] + clk[6] - clk_sl0);
#endif

			// ---------------- SHADOW: any-hit sweep over the requests of this step
			if constexpr (SUN) {
				constexpr uint32_t kOccluded = 0x80000000u;
				const bool listed = SURF && defer && n_sh > 0;
				if constexpr (SURF) if (listed) {
					// the requests go through the same inline-or-set-aside classification as the extend sweep; a request is marked
					// occluded (kOccluded in its kind word) by whichever surface reports a hit first, and entries of an already
					// occluded request are passed over
					list_len = 0;
					for (uint32_t base = 0; base < n_sh; base += 64) {
						const uint32_t j = base + lane;
						const bool active = j < n_sh;
						V3 o = {0, 0, 0}, d = {0, 0, 1};
						if (active) { const float4 r0 = sreq[j], r1 = sreq[kChunk + j]; o = mk(r0.x, r0.y, r0.z); d = mk(r1.x, r1.y, r1.z); }
						bool occ = false;
						uint32_t cur_space = 0xFFFFFFFFu;
						V3 lo = o, ld = d, inv = d;
						for (int m = 0; m < S.n_models; m++) {
							const ModelRec& M = S.models[m];
							const uint32_t spc = S.model_space[m];
							if (spc != cur_space) {
								const SpaceRec& SP = S.spaces[spc];
								lo = mulmv(SP.inv_basis, o) + mk(SP.inv_origin[0], SP.inv_origin[1], SP.inv_origin[2]);
								ld = normalize(mulmv(SP.inv_basis, d));
								inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
								cur_space = spc;
							}
							float nr, fr;
							const bool enters = active && !occ && aabb_test_inv(M.bmin, M.bmax, lo, inv, nr, fr);
							if (__ballot(enters) == 0) continue;
							for (int k = 0; k < M.n_surfaces; k++) {
								const int u = M.first_surface + k;
								const SurfaceRec& sf = S.surfaces[u];
								const bool ent = enters && !occ && aabb_test_inv(sf.bmin, sf.bmax, lo, inv, nr, fr);
								const uint64_t em = __ballot(ent);
								if (em == 0) continue;
								const uint32_t cnt = (uint32_t)__popcll(em);
								if (cnt >= kInlineMin) {
									if (ent) {
										MeshHit mh;
										if (mesh_traverse_m<MODE, 4>(g, sf, nr, fr, lo, ld, mh, spill PROF_PASS) && length(mulmv(M.basis, ld * mh.t)) >= 0) occ = true;
									}
								} else {
									const uint32_t len = __builtin_amdgcn_readlane(list_len, u);
									if (ent) {
										const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
										const Q4 L0 = lists + (size_t)u * 2u * kListCap;
										L0[len + r] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(j));
										L0[kListCap + len + r] = make_float4(ld.x, ld.y, ld.z, 0.f);
									}
									list_len = (int)lane == u ? len + cnt : list_len;
								}
							}
						}
						if (active && occ) { uint32_t* kw = reinterpret_cast<uint32_t*>(raw(sreq) + kChunk + j) + 3; *kw = *kw | kOccluded; }
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					for (int u = 0; u < (int)S.n_surfaces; u++) {
						const uint32_t len = __builtin_amdgcn_readlane(list_len, u);
						if (len == 0) continue;
						const SurfaceRec& sf = S.surfaces[u];
						const ModelRec& M = S.models[sf.model];
						const Q4 L0 = lists + (size_t)u * 2u * kListCap;
						for (uint32_t base = 0; base < len; base += 64) {
							if (base + lane < len) {
								const float4 e0 = L0[base + lane], e1 = L0[kListCap + base + lane];
								const uint32_t j = __float_as_uint(e0.w);
								uint32_t* kw = reinterpret_cast<uint32_t*>(raw(sreq) + kChunk + j) + 3;
								if (!(*kw & kOccluded)) {
									const V3 lo = mk(e0.x, e0.y, e0.z), ld = mk(e1.x, e1.y, e1.z);
									const V3 inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
									MeshHit mh;
									if (mesh_intersect_m<MODE, 10>(g, sf, lo, ld, inv, mh, spill PROF_PASS) && length(mulmv(M.basis, ld * mh.t)) >= 0) *kw = *kw | kOccluded;
								}
							}
						}
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					}
				}
				for (uint32_t base = 0; base < n_sh; base += 64) {
					const uint32_t j = base + lane;
					bool through = false;
					uint32_t target = 0, id = 0;
					V3 x = {0, 0, 0};
					if (j < n_sh) {
						const float4 r0 = sreq[j], r1 = sreq[kChunk + j], r2 = sreq[2 * kChunk + j];
						target = __float_as_uint(r0.w); id = __float_as_uint(r2.w);
						x = mk(r2.x, r2.y, r2.z);
						const bool occluded = listed ? (__float_as_uint(r1.w) & kOccluded) != 0
						                             : scene_occluded<MODE>(S, g, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), spill);
						if ((__float_as_uint(r1.w) & ~kOccluded) == REQ_ADD) {
							if (!occluded) {
								if (target == 0xFFFFFFFFu) {
									float4 v = B.sample_rad[id];
									B.sample_rad[id] = make_float4(v.x + x.x, v.y + x.y, v.z + x.z, v.w);
								} else {
									float4 q2 = qout[2 * kChunk + target];
									float* lz = reinterpret_cast<float*>(raw(qout) + 3 * kChunk + target);
									qout[2 * kChunk + target] = make_float4(q2.x, q2.y, q2.z + x.x, q2.w + x.y);
									*lz = *lz + x.z;
								}
							}
						} else if (occluded) {   // shadowed catcher: the path ends — trace() returns what it has (black), the worker zeroes the colour
							const float4 q2 = qin[2 * kChunk + target], q3 = qin[3 * kChunk + target];
							B.sample_rad[id] = WORKER ? make_float4(0.f, 0.f, 0.f, 1.0f) : make_float4(q2.z, q2.w, q3.x, 1.0f);
						} else {
							through = true;
						}
					}
					if (through) {   // lit catcher = fully transparent: same depth, next pass (renderer.cpp:513-519, shading_worker.cpp:95-104)
						const float4 q3 = qin[3 * kChunk + target];
						if (((__float_as_uint(q3.y) & 0xFFFFu) + 1u) > 4096u) {
							const float4 q2 = qin[2 * kChunk + target];
							B.sample_rad[id] = make_float4(q2.z, q2.w, q3.x, 1.0f);
							through = false;
						}
					}
					const uint64_t tm = __ballot(through);
					if (through) {
						const uint32_t pos = n_out + __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
						const float4 q1 = qin[kChunk + target], q2 = qin[2 * kChunk + target], q3 = qin[3 * kChunk + target];
						const V3 dn = normalize(mk(q1.x, q1.y, q1.z));
						qout[pos] = make_float4(x.x, x.y, x.z, __uint_as_float(id));
						qout[kChunk + pos] = make_float4(dn.x, dn.y, dn.z, q1.w);
						qout[2 * kChunk + pos] = q2;
						qout[3 * kChunk + pos] = make_float4(q3.x, __uint_as_float(__float_as_uint(q3.y) + 1u), q3.z, q3.w);
					}
					n_out += (uint32_t)__popcll(tm);
				}
				rays += n_sh > lane ? (n_sh - lane + 63u) / 64u : 0u;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			}
			n_in = n_out;
		}
	}
	// ray counter: one atomic per wave
	for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
	if (lane == 0 && rays) atomicAdd(B.ray_counter, (unsigned long long)rays);
#ifdef PTX_CLK
	clk[0] = __builtin_amdgcn_s_memtime() - clk_start;
	if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(B.ray_counter + 6 + k, (unsigned long long)(clk[k] >> 10));
#endif
#ifdef PTX_PROF
	for (int k = 0; k < kProfRegions; k++) {
		if (prof.t[k]) atomicAdd(B.ray_counter + 6 + 2 * k, (unsigned long long)prof.t[k]);       // counters block + 64 bytes
		if (prof.l[k]) atomicAdd(B.ray_counter + 6 + 2 * k + 1, (unsigned long long)prof.l[k]);
	}
#endif
}

// Adds the pass's samples of each pixel, in sample order, into the accumulation buffer (sums).
__global__ void k_resolve(const float4* __restrict__ sample_rad, float4* __restrict__ accum, const uint32_t* __restrict__ pixels, uint32_t n_pixels, uint32_t pass_spp) {
	uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pixels) return;
	const uint32_t dst = pixels ? pixels[p] : p;   // sharded passes: sample_rad is compact over the pass's pixel list
	float4 a = accum[dst];
	for (uint32_t s = 0; s < pass_spp; s++) {
		float4 r = sample_rad[(size_t)s * n_pixels + p];
		a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
	}
	accum[dst] = a;
}

// ------------------------------------------------------------------------------------ batch intersect
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_intersect_batch(DevScene S0, IntersectArgs A, const ModelRec* __restrict__ t_models, const SurfaceRec* __restrict__ t_surfaces, const SpaceRec* __restrict__ t_spaces, const uint32_t* __restrict__ t_model_space) {
	DevScene S = S0;
	S.models = t_models; S.surfaces = t_surfaces; S.spaces = t_spaces; S.model_space = t_model_space;  // see struct Tables
	const Staged st = stage_geometry<MODE>(S, g_smem);
	S.hot_lds = st.hot;
	const Geoms g = st.g;
	const Spill spill{A.spill + (size_t)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * (kSpillStack * 64) + (threadIdx.x & 63u)};
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += stride) {
		const V3 o = mk(A.ox[i], A.oy[i], A.oz[i]), d = mk(A.dx[i], A.dy[i], A.dz[i]);
		SceneHit h;
		const bool hit = scene_traverse<MODE>(S, g, o, d, h, spill);
		write_hit_outputs(S, st.shade, A, i, hit, h);
	}
}

// ------------------------------------------------------------------------------------ batch BSDF functions
// The sampling / pdf functions exactly as k_render_pass inlines them, one record per lane (ptx_pbr_eval_batch).
__global__ void k_pbr_eval(const float* __restrict__ in, float* __restrict__ out, size_t n) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float* p = in + 14 * i;
	const V3 nrm = mk(p[0], p[1], p[2]), o = mk(p[3], p[4], p[5]), inc = mk(p[6], p[7], p[8]);
	const float u1 = p[9], u2 = p[10], rough = p[11], cos_theta = p[12], ior = p[13];
	const V3 cone = rand_cone_vec(u2, cos_theta, nrm);
	const V3 idf = importance_sample(false, u1, u2, nrm, o, rough);
	const V3 isp = importance_sample(true, u1, u2, nrm, o, rough);
	const V3 rf = reflect3(-o, nrm);
	float* q = out + 15 * i;
	q[0] = cone.x; q[1] = cone.y; q[2] = cone.z;
	q[3] = idf.x; q[4] = idf.y; q[5] = idf.z;
	q[6] = isp.x; q[7] = isp.y; q[8] = isp.z;
	q[9] = pdf_diffuse(nrm, inc);
	q[10] = pdf_specular(nrm, o, inc, rough);
	q[11] = fresnel_schlick(o, rf, ior);
	q[12] = rf.x; q[13] = rf.y; q[14] = rf.z;
}

// ------------------------------------------------------------------------------------ batch camera rays
// scene::camera::get_ray(ndc, ratio) exactly as the integrator kernels inline it, one record per lane (ptx_camera_rays_batch)
__global__ void k_camera_rays(DevScene S, const float* __restrict__ in, float* __restrict__ out, size_t n) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	V3 o, d;
	camera_get_ray(S, in[3 * i], in[3 * i + 1], in[3 * i + 2], o, d);
	float* q = out + 6 * i;
	q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = d.x; q[4] = d.y; q[5] = d.z;
}

// ------------------------------------------------------------------------------------ tonemap + encode
// core::tonemap_approx_aces (core/utils.hpp:29-36) + image::image::write (image/image.cpp:143-154)
DEV float aces1(float x) {
	float v = (x * (2.51F * x + 0.03F)) / (x * (2.43F * x + 0.59F) + 0.14F);
	v = 0 > v ? 0 : v;
	v = 1 < v ? 1 : v;
	return v;
}
DEV uint32_t quant8(float v) { return (uint32_t)(uint8_t)(int)(v * 255 + 0.5F); }
// image::write on a colour channel of an sRGB image: static_cast<uint8_t>(math::pow(value, 1 / 2.2F) * 255 + 0.5F) — the pow is glibc's
// powf, which no other implementation reproduces bit for bit, and a 1-ulp difference flips a byte whenever value * 255 + 0.5 lands on an
// integer. The byte is a non-decreasing step function of value on [0, 1] (checked over every float of the interval,
// tests/test_oracle_vs_reference.py), so it is evaluated here as such: thr[k] = the smallest value whose byte is >= k, found on the
// host with that same powf (ptx_api.cpp: srgb_thresholds), and an 8-step binary search per channel. A NaN compares false: byte 0, as
// the reference's float -> int conversion of NaN gives after truncation to 8 bits.
DEV uint32_t srgb8(const float* __restrict__ thr, float v) {
	uint32_t k = 0;
#pragma unroll
	for (uint32_t step = 128; step; step >>= 1)
		if (v >= thr[k + step]) k += step;
	return k;
}
__global__ void k_tonemap(const float4* __restrict__ accum, uint32_t n_pixels, float spp, const float* __restrict__ thr, uchar4* __restrict__ out) {
	uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pixels) return;
	float4 a = accum[p];
	uchar4 o;
	o.x = (unsigned char)srgb8(thr, aces1(a.x / spp));
	o.y = (unsigned char)srgb8(thr, aces1(a.y / spp));
	o.z = (unsigned char)srgb8(thr, aces1(a.z / spp));
	o.w = (unsigned char)quant8(a.w / spp);
	out[p] = o;
}

// ------------------------------------------------------------------------------------ launchers
static hipError_t set_lds(const void* fn, size_t bytes) {
	return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int MODE, bool SURF, bool SUN, bool ALPHA, bool TEX, bool WORKER = false>
static hipError_t launch_pass_variant(const DevScene& S, const RenderParams& P, const PassBuffers& B, size_t lds_bytes, int grid, hipStream_t stream) {
	if (MODE != MODE_GLOBAL) {
		hipError_t e = set_lds(reinterpret_cast<const void*>(&k_render_pass<MODE, SUN, ALPHA, TEX, WORKER, SURF>), lds_bytes);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL((k_render_pass<MODE, SUN, ALPHA, TEX, WORKER, SURF>), dim3(grid), dim3(kBlock), MODE != MODE_GLOBAL ? lds_bytes : 0, stream, S, P, B, S.models, S.surfaces, S.spaces, S.model_space);
	return hipGetLastError();
}

// Kernel variants: {where the geometry lives} x {units set aside: models | surfaces} x {sun shadow rays} x {opacity / shadow-
// catcher pass-through}; textured scenes and the worker estimator get one variant with everything compiled in (sun code is
// skipped at run time without a sun).
template <int MODE, bool SURF>
static hipError_t launch_pass_mode(const DevScene& S, const RenderParams& P, const PassBuffers& B, size_t lds_bytes, int grid, hipStream_t stream) {
	const bool sun = S.sun.present != 0, alpha = S.any_alpha != 0;
	if (P.integrator == 1u)
		return S.any_texture ? launch_pass_variant<MODE, SURF, true, true, true, true>(S, P, B, lds_bytes, grid, stream)
		                     : launch_pass_variant<MODE, SURF, true, true, false, true>(S, P, B, lds_bytes, grid, stream);
	if (S.any_texture) return launch_pass_variant<MODE, SURF, true, true, true>(S, P, B, lds_bytes, grid, stream);
	if (sun) return alpha ? launch_pass_variant<MODE, SURF, true, true, false>(S, P, B, lds_bytes, grid, stream)
	                      : launch_pass_variant<MODE, SURF, true, false, false>(S, P, B, lds_bytes, grid, stream);
	return alpha ? launch_pass_variant<MODE, SURF, false, true, false>(S, P, B, lds_bytes, grid, stream)
	             : launch_pass_variant<MODE, SURF, false, false, false>(S, P, B, lds_bytes, grid, stream);
}
hipError_t launch_render_pass(const DevScene& S, const RenderParams& P, const PassBuffers& B, int mode, size_t lds_bytes, int grid,
                              hipStream_t stream) {
	if (B.surface_units) {
		if (mode == MODE_LDS) return launch_pass_mode<MODE_LDS, true>(S, P, B, lds_bytes, grid, stream);
		if (mode == MODE_HYBRID) return launch_pass_mode<MODE_HYBRID, true>(S, P, B, lds_bytes, grid, stream);
		return launch_pass_mode<MODE_GLOBAL, true>(S, P, B, lds_bytes, grid, stream);
	}
	if (mode == MODE_LDS) return launch_pass_mode<MODE_LDS, false>(S, P, B, lds_bytes, grid, stream);
	if (mode == MODE_HYBRID) return launch_pass_mode<MODE_HYBRID, false>(S, P, B, lds_bytes, grid, stream);
	return launch_pass_mode<MODE_GLOBAL, false>(S, P, B, lds_bytes, grid, stream);
}
hipError_t launch_resolve(const float4* sample_rad, float4* accum, const uint32_t* pixels, uint32_t n_pixels, uint32_t pass_spp, hipStream_t stream) {
	hipLaunchKernelGGL(k_resolve, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, sample_rad, accum, pixels, n_pixels, pass_spp);
	return hipGetLastError();
}
template <int MODE>
static hipError_t launch_intersect_mode(const DevScene& S, const IntersectArgs& A, size_t lds_bytes, int grid, hipStream_t stream) {
	if (MODE != MODE_GLOBAL) {
		hipError_t e = set_lds(reinterpret_cast<const void*>(&k_intersect_batch<MODE>), lds_bytes);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(k_intersect_batch<MODE>, dim3(grid), dim3(kBlock), MODE != MODE_GLOBAL ? lds_bytes : 0, stream, S, A, S.models, S.surfaces, S.spaces, S.model_space);
	return hipGetLastError();
}
hipError_t launch_intersect(const DevScene& S, const IntersectArgs& A, int mode, size_t lds_bytes, int grid, hipStream_t stream) {
	if (mode == MODE_LDS) return launch_intersect_mode<MODE_LDS>(S, A, lds_bytes, grid, stream);
	if (mode == MODE_HYBRID) return launch_intersect_mode<MODE_HYBRID>(S, A, lds_bytes, grid, stream);
	return launch_intersect_mode<MODE_GLOBAL>(S, A, lds_bytes, grid, stream);
}
hipError_t launch_pbr_eval(const float* in, float* out, size_t n, hipStream_t stream) {
	hipLaunchKernelGGL(k_pbr_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
	return hipGetLastError();
}
hipError_t launch_camera_rays(const DevScene& S, const float* in, float* out, size_t n, hipStream_t stream) {
	hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, in, out, n);
	return hipGetLastError();
}
hipError_t launch_tonemap(const float4* accum, uint32_t n_pixels, float spp, const float* thresholds, uchar4* out, hipStream_t stream) {
	hipLaunchKernelGGL(k_tonemap, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, accum, n_pixels, spp, thresholds, out);
	return hipGetLastError();
}

}  // namespace ptx
