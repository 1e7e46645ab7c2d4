// Queue-based ("wavefront") intersection for scenes whose trees live in global memory and whose models have many surfaces
// (a Sponza-class mesh: one model, dozens of surfaces, hundreds of thousands of triangles).
//
// In the fused kernel (kernels.hip) a wave owns 64 rays from start to end; on such a scene a ray enters 3-5 of the surface
// boxes, the 64 rays of a wave enter different ones, every tree walk has its own length and every step of it is a dependent
// fetch from L2 / HBM: the measured result is 6 % of the lanes on in the tree loops and 56 % of the wave-cycles parked on
// s_waitcnt (profiles/round2_pmc_atrium*.txt). Here the unit of work is the PAIR (ray, surface it enters):
//   k_wf_classify  one ray per lane: local ray per model space, model box, surface boxes (scene::model::intersect, model.cpp:27-60;
//                  core::mesh::intersect's box test, mesh.cpp:308-315) -> the ray's pairs, written as queue entries (local ray + result
//                  slot) grouped by surface: pair space comes from a pool sized by demand, one reservation per tile of 1024 rays
//   k_wf_traverse  persistent waves that eat the queues: every lane walks ONE pair's tree (core::mesh::intersect, mesh.cpp:300-405)
//                  and takes the next pair as soon as its walk ends, so lanes stay busy whatever the walk lengths; 256-thread
//                  blocks and a traversal-only register footprint (59 VGPRs, 24 KB of LDS) give 6 waves per SIMD to cover the fetch latency; queues are
//                  dealt to XCDs surface by surface so that each L2 sees a part of the geometry, and started largest tree first (DevScene::wf_order)
//                  so that a launch ends on the queues of the short walks
//   k_wf_merge_* / k_wf_shade   one ray per lane again: the ray's pair results (fetched ahead, four at a time) in surface order = model::intersect's loop (first
//                  surface wins ties on the local distance), then renderer::intersect's loop over models (first model wins ties on the
//                  world distance, renderer.cpp:663-669)
// The entry count of a step lives on the device (flow words): the host enqueues all steps of a slab without reading anything back.
// Every arithmetic operation on a ray is the one the fused kernel performs (same device functions); only where and when it
// happens differs, so results are bitwise those of the fused kernel and of the reference.
#include "device_core.hpp"

namespace ptx {

// Tunables (the -D overrides are for tools/build_variant.sh A/B builds; measured values in profiles/round2_wf_ab.txt, round3_wf_ab.txt)
constexpr int kWfBlock = 256;              // threads per workgroup of the traverse / shade / merge kernels
constexpr int kWfClassifyBlock = (int)kWfTile;   // ... of the classify kernel: pair space is reserved per tile of 1024 rays
#ifndef PTX_WF_UNIT
#define PTX_WF_UNIT 64
#endif
constexpr uint32_t kWfUnit = PTX_WF_UNIT;  // queue entries a wave stages into its LDS slice at a time
static_assert(kWfUnit >= 1 && kWfUnit <= 64, "a unit is staged by one pass of the wave's 64 lanes");
#ifndef PTX_WF_LDS_STACK
#define PTX_WF_LDS_STACK 8
#endif
constexpr int kWfLdsStack = PTX_WF_LDS_STACK;         // traversal-stack levels kept in LDS between the register levels and the global-memory overflow
#ifndef PTX_WF_REFILL_MIN
#define PTX_WF_REFILL_MIN 8
#endif
#ifndef PTX_WF_UNROLL
#define PTX_WF_UNROLL 2
#endif
constexpr uint32_t kWfRefillMin = PTX_WF_REFILL_MIN;   // idle lanes that make a hand-out of new pairs worth its instructions

// PTX_WF_PROF builds: wave-level trips and active lanes per region of k_wf_traverse, added into ctl[kWfCtlProf ..] (measurement only)
#ifdef PTX_WF_PROF
#define WFPROF(k) do { const uint64_t m_ = __ballot(true); pl[k] += 1u; pt[k] += (lane == (uint32_t)(__ffsll((long long)m_) - 1)) ? 1u : 0u; if ((k) == 2 || (k) == 3) walk_steps++; } while (0)
// wave-level clock spent per region (kilocycles, lane 0 adds it up): T0 / T1 bracket a region executed under wave-uniform control flow
#define WFT0() const uint64_t t0_ = __builtin_amdgcn_s_memtime()
#define WFT1(k) do { tc[k] += __builtin_amdgcn_s_memtime() - t0_; } while (0)
#else
#define WFPROF(k) do { } while (0)
#define WFT0() do { } while (0)
#define WFT1(k) do { } while (0)
#endif

DEV uint32_t lane_id() { return threadIdx.x & 63u; }
DEV uint32_t rank_in(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// the local ray of model space `spc` (geometry/ray.cpp:10-15), as scene_traverse computes it
DEV void to_space(const SpaceRec& SP, V3 o, V3 d, V3& lo, V3& ld, V3& inv) {
	lo = mulmv(SP.inv_basis, o) + mk(SP.inv_origin[0], SP.inv_origin[1], SP.inv_origin[2]);
	ld = normalize(mulmv(SP.inv_basis, d));
	inv = mk(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
}

// ------------------------------------------------------------------------------------ classify
// Ray sources: ray `i` of a step with `m` entries is (o, d) when valid(i, m); the step has count(m) rays
struct SoaRays {
	const float *ox, *oy, *oz, *dx, *dy, *dz;
	DEV uint32_t count(uint32_t m) const { return m; }
	DEV bool valid(uint32_t, uint32_t) const { return true; }
	DEV void load(uint32_t i, uint32_t, V3& o, V3& d) const { o = mk(ox[i], oy[i], oz[i]); d = mk(dx[i], dy[i], dz[i]); }
};

// One tile = 1024 consecutive rays, one ray per lane: local ray per model space, model box, surface boxes (scene::model::intersect,
// model.cpp:27-60; core::mesh::intersect's box test, mesh.cpp:308-315) -> a 64-bit mask of entered surfaces per ray. The tile reserves
// its pairs with one atomic; inside its block of the pool the queue entries are grouped by surface (one segment per surface, published
// in that surface's segment list), the result slots by ray. Persistent workgroups loop over the tiles of the step: the entry count of
// a step is only known on the device.
template <class Src>
__global__ void __launch_bounds__(kWfClassifyBlock) k_wf_classify(DevScene S0, Src src, uint32_t n_host, WfBuffers W, const ModelRec* __restrict__ t_models,
                                                         const SurfaceRec* __restrict__ t_surfaces, const SpaceRec* __restrict__ t_spaces,
                                                         const uint32_t* __restrict__ t_model_space) {
	DevScene S = S0;
	S.models = t_models; S.surfaces = t_surfaces; S.spaces = t_spaces; S.model_space = t_model_space;   // scalar-loaded tables (device_core.hpp: Tables)
	__shared__ uint32_t s_cnt[kWfMaxSurfaces], s_off[kWfMaxSurfaces], s_wave_first[kWfClassifyBlock / 64];
	__shared__ uint32_t s_rays, s_base, s_over;
	const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
	const uint32_t m_in = W.n_in ? *W.n_in : n_host;
	const uint32_t n = src.count(m_in);
	for (uint32_t tile = blockIdx.x; (uint64_t)tile * kWfTile < (uint64_t)n; tile += gridDim.x) {
		if (threadIdx.x < (uint32_t)kWfMaxSurfaces) s_cnt[threadIdx.x] = 0;
		if (threadIdx.x == 0) s_rays = 0;
		__syncthreads();
		const uint32_t i = tile * kWfTile + threadIdx.x;
		const bool active = i < n && src.valid(i, m_in);
		V3 o = {0, 0, 0}, d = {0, 0, 1};
		if (active) src.load(i, m_in, o, d);

		// pass 1: the surfaces this ray enters (bit u of `mine`), and per surface the number of rays of this wave that enter it (lane u of `cnt`)
		unsigned long long mine = 0;
		uint32_t cnt = 0;
		uint32_t cur_space = 0xFFFFFFFFu;   // the local ray is kept for the second pass: a scene of one ray space (one model, or models that
		V3 lo = o, ld = d, inv = d;         // share a transform) computes it once per ray
		{
			for (int m = 0; m < S.n_models; m++) {
				const ModelRec& M = S.models[m];
				const uint32_t spc = S.model_space[m];
				if (spc != cur_space) { to_space(S.spaces[spc], o, d, lo, ld, inv); cur_space = spc; }
				float nr, fr;
				const bool enters = active && aabb_test_box(M.box, lo, inv, nr, fr);   // model box first (model.cpp:38-40)
				if (__ballot(enters) == 0) continue;
				for (int k = 0; k < M.n_surfaces; k++) {
					const int u = M.first_surface + k;
					const SurfaceRec& sf = S.surfaces[u];
					const bool ent = enters && aabb_test_box(sf.box, lo, inv, nr, fr);
					const uint64_t em = __ballot(ent);
					if (em == 0) continue;
					if (ent) mine |= 1ull << u;
					cnt = (int)lane == u ? (uint32_t)__popcll(em) : cnt;
				}
			}
		}
		// result slots: the pairs of a ray are consecutive, in surface order. Reservations are made per TILE (one global atomic for the
		// pair space, one per entered surface for its segment list): per-wave atomics on two dozen addresses serialise in L2 and cost
		// more than the box tests (measured: 0.87 ms per 4 M rays with per-wave atomics)
		const uint32_t mycnt = (uint32_t)__popcll(mine);
		uint32_t incl = mycnt;
		for (uint32_t off = 1; off < 64; off <<= 1) {
			const uint32_t t = __shfl_up(incl, off);
			if (lane >= off) incl += t;
		}
		const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
		uint32_t woff = 0;   // lane u: where this wave's entries start inside the tile's segment of surface u
		if (cnt) woff = atomicAdd(&s_cnt[lane], cnt);
		if (lane == 0) s_wave_first[wave] = total;
		if (W.ray_counter) {   // rays traced (stats): one atomic per tile
			const uint32_t nv = (uint32_t)__popcll(__ballot(active));
			if (lane == 0 && nv) atomicAdd(&s_rays, nv);
		}
		__syncthreads();
		if (W.ray_counter && threadIdx.x == 0 && s_rays) atomicAdd(W.ray_counter, (unsigned long long)s_rays);
		if (wave == 0) {
			// lane u: entries of surface u in this tile, and where its segment starts inside the tile's block (exclusive prefix)
			const uint32_t c = s_cnt[lane];
			uint32_t cx = c;
			for (uint32_t off = 1; off < 64; off <<= 1) {
				const uint32_t v = __shfl_up(cx, off);
				if (lane >= off) cx += v;
			}
			const uint32_t block_total = __builtin_amdgcn_readlane(cx, 63);
			s_off[lane] = cx - c;
			// lane w: result slots of wave w start at ... (exclusive prefix of the waves' totals)
			uint32_t t = lane < (uint32_t)(kWfClassifyBlock / 64) ? s_wave_first[lane] : 0u, ex = t;
			for (uint32_t off = 1; off < (uint32_t)(kWfClassifyBlock / 64); off <<= 1) {
				const uint32_t v = __shfl_up(ex, off);
				if (lane >= off) ex += v;
			}
			uint32_t base = 0;
			if (lane == 0 && block_total) base = atomicAdd(&W.ctl[0], block_total);
			base = __builtin_amdgcn_readfirstlane(base);
			// the pool is sized from demand: a tile whose pairs do not fit writes none, raises the overflow word, and the host repeats
			// the slab in smaller pieces (64-bit sum: `base` keeps growing after the pool is full)
			const bool over = (uint64_t)base + block_total > (uint64_t)W.pool_cap;
			if (lane < (uint32_t)(kWfClassifyBlock / 64)) s_wave_first[lane] = base + ex - t;
			if (lane == 0) {
				s_base = base; s_over = over ? 1u : 0u;
				if (over) { W.ctl[1] = 1u; *W.overflow = 1u; }
			}
			if (!over && c) {   // publish this tile's segment of surface `lane`
				const uint32_t k = atomicAdd(&W.ctl[kWfCtlSeg + lane], 1u);
				W.seg[(size_t)lane * W.seg_cap + k] = make_uint2(base + (cx - c), c);
			}
		}
		__syncthreads();
		const bool over = s_over != 0;
		const uint32_t first = s_wave_first[wave] + incl - mycnt;
		if (i < n) { W.first[i] = over ? 0u : first; W.mask[i] = over ? 0ull : mine; }
		const uint32_t qb = cnt ? s_base + s_off[lane] + woff : 0u;   // lane u: first queue entry of this wave in the tile's segment of surface u

		// pass 2: the queue entries (local ray + result slot), grouped by surface
		if (!over) {
			for (int m = 0; m < S.n_models; m++) {
				const ModelRec& M = S.models[m];
				const unsigned long long range = (M.n_surfaces >= 64 ? ~0ull : ((1ull << M.n_surfaces) - 1ull)) << M.first_surface;
				if (__ballot((mine & range) != 0) == 0) continue;
				const uint32_t spc = S.model_space[m];
				if (spc != cur_space) { to_space(S.spaces[spc], o, d, lo, ld, inv); cur_space = spc; }
				for (int k = 0; k < M.n_surfaces; k++) {
					const int u = M.first_surface + k;
					const bool bit = (mine >> u) & 1ull;
					const uint64_t em = __ballot(bit);
					if (em == 0) continue;
					const uint32_t b = __builtin_amdgcn_readlane(qb, u);
					if (bit) {
						const uint32_t kk = (uint32_t)__popcll(mine & ((1ull << u) - 1ull));
						const size_t e = (size_t)b + rank_in(em);
						// (one store instruction per SURFACE has an eighth of its lanes on; collecting a ray's entries through LDS and writing them
						// ray by ray — 14 instead of 48 store instructions per 64 rays — changed nothing: the stores are 15 % of this kernel)
						W.qent[2 * e] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(first + kk));
						W.qent[2 * e + 1] = make_float4(ld.x, ld.y, ld.z, 0.f);
					}
				}
			}
		}
		__syncthreads();   // the next tile resets what this one still reads
	}
}

// ------------------------------------------------------------------------------------ traverse
// Visiting order of the queues for a workgroup on XCD x (workgroups are dealt to the 8 XCDs round-robin by blockIdx): first the
// surfaces u with u % 8 == x, then those of XCD x + 1, ... — while work lasts, every L2 serves its own eighth of the trees.
DEV int wf_surface_at(const uint32_t* __restrict__ order, uint32_t xcd, uint32_t r, uint32_t n_surf) {
	// r-th surface in the order above; rows of 8: ranks {x, x+8, ..}, then {x+1, ..}, ... of `order` (DevScene::wf_order, upload_scene)
	const uint32_t per = (n_surf + 7u) / 8u;
	const uint32_t off = r / per, j = r % per;
	const uint32_t u = ((xcd + off) & 7u) + 8u * j;
	return u < n_surf ? (int)order[u] : -1;
}

// The nested-loop form (kept for measurement, PTX_WF_KERNEL=1, and for layouts the one-loop kernel does not read: per-triangle records,
// geometry beyond 4 GB). Persistent 256-thread workgroups (75 VGPRs, no scratch, 6 waves per SIMD). Every lane walks ONE pair's tree (mesh.cpp:300-405, the
// loop of mesh_traverse) and takes its next pair from the wave's LDS-staged unit as soon as the walk ends. A wave takes work one
// SEGMENT (the entries one classify tile queued for one surface: contiguous, <= 1024) at a time with one atomic on the surface's
// cursor, and stages it in units of 64 entries with coalesced loads — the entries carry the local ray and the result slot, nothing
// is gathered.
#ifdef PTX_WF_WAVES
__attribute__((amdgpu_waves_per_eu(PTX_WF_WAVES, PTX_WF_WAVES)))
#endif
__global__ void __launch_bounds__(kWfBlock) k_wf_traverse(DevScene S0, WfBuffers W, const SurfaceRec* __restrict__ t_surfaces) {
	DevScene S = S0;
	S.surfaces = t_surfaces;
	if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(W.peak, W.ctl[0]);   // demand of this step (counted on even when it overflowed the pool)
	if (W.ctl[1]) return;   // this step's pairs did not fit the pool: the host repeats the slab
	const Geom g = {S.nodes, S.refs, S.tri_isect, S.glb_leaf_ordered != 0, true};
	__shared__ float4 s_ray[kWfBlock / 64][kWfUnit][2];
	// Pending subtrees beyond the register levels. A store to global memory here would sit in the same in-order counter as the node
	// fetches (gfx9 has one vmcnt for loads and stores): 4 of 10 node steps on these trees push deeper than the registers hold, and each
	// would make the next fetch wait for a write acknowledgement. LDS has its own counter and a tenth of the latency.
	__shared__ uint2 s_stack[kWfLdsStack > 0 ? kWfLdsStack : 1][kWfBlock];
	const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
	const uint32_t xcd = blockIdx.x & 7u;
	const Spill spill{W.spill + (size_t)(blockIdx.x * (kWfBlock / 64) + wave) * (kSpillStack * 64) + lane};
	auto stack_put = [&](int k, uint32_t nn, float mm) {
		if (k < kWfLdsStack) s_stack[k][threadIdx.x] = make_uint2(nn, __float_as_uint(mm));
		else spill_put(spill, k - kWfLdsStack, nn, mm);
	};
	auto stack_get = [&](int k, uint32_t& nn, float& mm) {
		if (k < kWfLdsStack) { const uint2 v = s_stack[k][threadIdx.x]; nn = v.x; mm = __uint_as_float(v.y); }
		else spill_get(spill, k - kWfLdsStack, nn, mm);
	};
	const uint32_t n_surf = S.n_surfaces;
	const uint32_t n_order = ((n_surf + 7u) / 8u) * 8u;

	// wave-uniform: the segment being handed out, and the unit of it staged in LDS
	uint32_t seg_pos = 0, seg_end = 0, unit_pos = 0, unit_n = 0, order_pos = 0;
	int unit_surf = -1;
	bool more = true;
	// per lane: the walk in progress (core::mesh::intersect's locals, as in mesh_traverse)
	bool busy = false, have = false;
	uint32_t slot = 0, node = 0;
	int sp = 0;
	uint32_t n0 = 0, n1 = 0, n2 = 0;
	float m0 = 0, m1 = 0, m2 = 0;
	float min_dist = 0, max_dist = 0, fr0 = 0;
	V3 o = {0, 0, 0}, d = {0, 0, 1};
#ifdef PTX_WF_PROF
	uint32_t pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 outer rounds, 1 busy rounds, 2 node steps, 3 triangle tests, 4 hand-outs, 5 unit fetches, 6 pops
	uint64_t tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // clocks: 0 whole kernel, 1 unit fetch, 2 hand-out, 3 pop, 4 descend loops, 5 leaf loops, 6 result stores
	const uint64_t t_start = __builtin_amdgcn_s_memtime();
	uint32_t walk_steps = 0, walk_max = 0;   // node steps + triangle tests of the lane's current walk / of its longest one
#endif

	for (;;) {
		WFPROF(0);
		const uint64_t idle_m = __ballot(!busy);
		if (more && ((uint32_t)__popcll(idle_m) >= kWfRefillMin || ~idle_m == 0)) {
			if (unit_pos == unit_n) {
				WFT0();
				if (seg_pos == seg_end) {
					// next segment: this XCD's surfaces first; one atomic on the surface's cursor per segment. A wave that finds a cursor past
					// the surface's last segment never returns to that surface
					for (;;) {
						if (unit_surf < 0) {
							int u = -1;
							for (; order_pos < n_order; order_pos++) {
								const int c = wf_surface_at(S.wf_order, xcd, order_pos, n_surf);
								if (c >= 0 && W.ctl[kWfCtlSeg + c] != 0) { u = c; break; }
							}
							u = __builtin_amdgcn_readfirstlane(u);
							if (u < 0) { more = false; break; }
							unit_surf = u;
						}
						uint32_t k = 0;
						if (lane == 0) k = atomicAdd(&W.ctl[kWfCtlCur + 64u * (uint32_t)unit_surf], 1u);
						k = __builtin_amdgcn_readfirstlane(k);
						if (k < W.ctl[kWfCtlSeg + unit_surf]) {
							const uint2 sg = W.seg[(size_t)unit_surf * W.seg_cap + k];
							seg_pos = __builtin_amdgcn_readfirstlane(sg.x);
							seg_end = seg_pos + __builtin_amdgcn_readfirstlane(sg.y);
							break;
						}
						unit_surf = -1;
						order_pos++;
					}
				}
				if (more) {
					WFPROF(5);
					// stage the next <= 64 entries of the segment into this wave's LDS slice (coalesced: the entries are contiguous)
					unit_n = seg_end - seg_pos < kWfUnit ? seg_end - seg_pos : kWfUnit;
					unit_pos = 0;
					if (lane < unit_n) {
						const size_t e = (size_t)seg_pos + lane;
						s_ray[wave][lane][0] = W.qent[2 * e];
						s_ray[wave][lane][1] = W.qent[2 * e + 1];
					}
					seg_pos += unit_n;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				}
				WFT1(1);
			}
			if (more) {
				WFT0();
				const uint32_t avail = unit_n - unit_pos;
				const uint32_t r = rank_in(idle_m);
				if (!busy && r < avail) {
					WFPROF(4);
#ifdef PTX_WF_PROF
					walk_max = walk_steps > walk_max ? walk_steps : walk_max; walk_steps = 0;
#endif
					const uint32_t e = unit_pos + r;
					const float4 e0 = s_ray[wave][e][0], e1 = s_ray[wave][e][1];
					slot = __float_as_uint(e0.w);
					o = mk(e0.x, e0.y, e0.z);
					d = mk(e1.x, e1.y, e1.z);
					const SurfaceRec& sf = S.surfaces[unit_surf];
					const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
					float nr, fr;
					if (aabb_test_box(sf.box, o, inv, nr, fr)) {   // mesh.cpp:308-315 (the classification saw the same test pass)
						busy = true; have = true;
						node = sf.kd_root; min_dist = nr; max_dist = fr; fr0 = fr; sp = 0;
					} else {
						W.pair_hit[slot] = make_float4(-1.0f, 0.f, 0.f, 0.f);
					}
				}
				const uint32_t n_idle = (uint32_t)__popcll(idle_m);
				unit_pos += n_idle < avail ? n_idle : avail;
				WFT1(2);
			}
		}
		if (__ballot(busy) == 0) {
			if (!more) break;
			continue;
		}
		// one round of core::mesh::intersect's loop (mesh.cpp:317-403; see mesh_traverse): pop, descend to a leaf, test it
		{
			WFT0();
			if (busy) {
				WFPROF(1);
				if (!have) {
					if (sp == 0) { W.pair_hit[slot] = make_float4(-1.0f, 0.f, 0.f, 0.f); busy = false; }
					else {
						WFPROF(6);
						sp--;
						node = n0; min_dist = m0;
						n0 = n1; m0 = m1; n1 = n2; m1 = m2;
						if (sp >= kRegStack) stack_get(sp - kRegStack, n2, m2);
						max_dist = sp > 0 ? m0 : fr0;
					}
				}
			}
			WFT1(3);
		}
		bool valid = false;
		uint2 nd = make_uint2(0, 0);
		{
			WFT0();
			if (busy) {
				have = false;
				valid = true;
				nd = g.nodes[node];
				while ((nd.y & 3u) != KD_LEAF) {
					WFPROF(2);
					const uint32_t axis = nd.y & 3u;
					const uint2 kid0 = g.nodes[nd.y >> 4], kid1 = g.nodes[(nd.y >> 4) + 1u];   // both children requested with the parent in hand (fetching only the chosen one afterwards: -5 %)
					const float split = __uint_as_float(nd.x);
					const float oa = sel3(o, axis), da = sel3(d, axis);
					const float split_dist = (split - oa) / da;
					const bool has_l = nd.y & 4u, has_r = nd.y & 8u;
					const uint32_t li = nd.y >> 4, ri = li + (has_l ? 1u : 0u);
					const bool left_first = oa < split;
					const uint32_t first = left_first ? li : ri, second = left_first ? ri : li;
					const bool has_first = left_first ? has_l : has_r, has_second = left_first ? has_r : has_l;
					uint32_t next;
					bool has_next;
					if (split_dist < 0 || split_dist > max_dist) { next = first; has_next = has_first; }
					else if (split_dist < min_dist) { next = second; has_next = has_second; }
					else {
						if (has_second && sp < kRegStack + kSpillStack) {
							if (sp >= kRegStack) { WFPROF(7); stack_put(sp - kRegStack, n2, m2); }
							n2 = n1; m2 = m1; n1 = n0; m1 = m0; n0 = second; m0 = split_dist;
							sp++;
						}
						next = first; has_next = has_first;
						max_dist = split_dist;
					}
					if (!has_next) { valid = false; break; }
					node = next;
					nd = next == li ? kid0 : kid1;
				}
			}
			WFT1(4);
		}
		{
			WFT0();
			if (busy && valid) {
				// leaf: nearest triangle with t <= max_dist; ties keep the first (mesh.cpp:381-389)
				const PRay pr = pack_ray(o, d);
				const uint32_t first_ref = nd.x, count = nd.y >> 2;
				float best_t = -1.0f, bb1 = 0, bb2 = 0;
				uint32_t best_tri = 0;
				for (uint32_t k = 0; k < count; k++) {
					WFPROF(3);
					const uint32_t rslot = g.leaf_ordered ? first_ref + k : g.refs[first_ref + k];
					const float4 r0 = g.tris[3 * rslot], r1 = g.tris[3 * rslot + 1], r2 = g.tris[3 * rslot + 2];
					const uint32_t ti = __float_as_uint(r2.z);
					float be, ga;
					const float t = tri_test_pk(r0, r1, make_float2(r2.x, r2.y), pr, be, ga);
					if (t >= 0 && t <= max_dist && (t < best_t || !(best_t >= 0))) { best_t = t; bb1 = be; bb2 = ga; best_tri = ti; }
				}
				if (best_t >= 0) {
					W.pair_hit[slot] = make_float4(best_t, __uint_as_float(best_tri), bb1, bb2);
					busy = false;
				}
			}
			WFT1(5);
		}
	}
#ifdef PTX_WF_PROF
	for (int k = 0; k < 8; k++) {
		if (pt[k]) atomicAdd(&W.ctl[kWfCtlProf + 2 * k], pt[k]);
		if (pl[k]) atomicAdd(&W.ctl[kWfCtlProf + 2 * k + 1], pl[k]);
	}
	tc[0] = __builtin_amdgcn_s_memtime() - t_start;
	if (lane == 0) for (int k = 0; k < 6; k++) atomicAdd(&W.ctl[kWfCtlProf + 16 + k], (uint32_t)(tc[k] >> 10));
	// the slowest wave: its clock, its trips (node steps + triangle tests, wave level) and the most trips any single lane-walk took
	if (lane == 0) { atomicMax(&W.ctl[kWfCtlProf + 24], (uint32_t)(tc[0] >> 10)); atomicMax(&W.ctl[kWfCtlProf + 25], pt[2] + pt[3]); }
	atomicMax(&W.ctl[kWfCtlProf + 26], walk_steps > walk_max ? walk_steps : walk_max);
	// when the waves ended: their own clock binned by log2 of kilocycles (ctl[kWfCtlProf + 32 .. + 63])
	if (lane == 0) { const uint32_t kc = (uint32_t)(tc[0] >> 10); atomicAdd(&W.ctl[kWfCtlProf + 32 + (kc ? 31 - __builtin_clz(kc) : 0)], 1u); if (pt[1] == 0) atomicAdd(&W.ctl[kWfCtlProf + 32 + 31], 1u); }
#endif
}

// ------------------------------------------------------------------------------------ traverse, one loop
// The same walks as k_wf_traverse, organised as ONE loop without inner loops: per trip every busy lane advances by one step of
// core::mesh::intersect — a node step (mesh.cpp:327-370) when it stands on a branch, a triangle test (mesh.cpp:381-389) when it
// stands in a leaf — and ALL the trip's fetches (the branch lanes' child pair, the leaf lanes' triangle record) are issued together
// at the top, so that a trip waits for memory once. In the nested loops a wave made one dependent fetch per node trip and per
// triangle trip, each at 22-38 % of the lanes (a lane that reached its leaf waited for the slowest descent, and the other way round):
// 14-26 lane steps per pair took 17-33 M wave trips per 25 M pairs; here the lanes of a wave step together: the same steps in a
// third of the trips. The pending-subtree stack keeps what the reference's stack keeps (mesh.cpp:317-325: node, min_dist, max_dist)
// with the node's CONTENT (8 bytes) in place of its pointer — both children are in registers when one of them is set aside, and a
// pop needs no fetch — as 16-byte entries in LDS (one ds_write_b128 / ds_read_b128; a register-held top would be rotated with a
// dozen moves on every push and pop of any lane). The surface's root node is read once per staged unit (wave-uniform). Nodes and
// triangle records live in ONE allocation (upload_scene): every fetch is `base + 32-bit offset`.
#ifndef PTX_WF_LDS_STACK2
#define PTX_WF_LDS_STACK2 4
#endif
constexpr int kWfLdsStack2 = PTX_WF_LDS_STACK2;
constexpr int kWfMaxStack = kRegStack + kSpillStack;   // entries a walk may set aside (the nested kernels' bound: 27 > mesh.hpp:34's 25 levels)
struct alignas(8) NodePair { uint32_t x, y, z, w; };   // two adjacent 8-byte nodes: 8-byte aligned, fetched as one 16-byte load

// BLOCK2 (measurement: PTX_WF_BLOCK2=1): the node array in 2-LEVEL BLOCKS (DevScene::nodes2, upload_scene: a branch at even depth owns 48
// contiguous bytes = its child pair, then the child pairs of its two children) — a lane that stands on such a branch fetches the whole
// block with the trip's three loads and makes TWO node steps in the trip: half the dependent fetches per descent, three times the
// bytes per fetch. Node word 1 there: axis | has-left << 2 | has-right << 3 | block-root << 4 | child pair index << 5.
template <bool BLOCK2>
__global__ void __launch_bounds__(kWfBlock) k_wf_traverse2(DevScene S0, WfBuffers W, const SurfaceRec* __restrict__ t_surfaces) {
	DevScene S = S0;
	S.surfaces = t_surfaces;
	if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(W.peak, W.ctl[0]);   // demand of this step (counted on even when it overflowed the pool)
	if (W.ctl[1]) return;   // this step's pairs did not fit the pool: the host repeats the slab
	__shared__ float4 s_ray[kWfBlock / 64][kWfUnit][2];
	__shared__ uint4 s_stk[kWfLdsStack2 > 0 ? kWfLdsStack2 : 1][kWfBlock];
	const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
	const uint32_t xcd = blockIdx.x & 7u;
	// beyond the LDS levels: lane-interleaved rows in global memory
	uint4* const spill4 = reinterpret_cast<uint4*>(W.spill) + (size_t)(blockIdx.x * (kWfBlock / 64) + wave) * (kSpillStack * 64) + lane;
	const unsigned char* const geom = reinterpret_cast<const unsigned char*>(S.nodes);
	const uint32_t tri_off = (uint32_t)(reinterpret_cast<const unsigned char*>(S.tri_isect) - geom);   // same allocation (upload_scene)
	[[maybe_unused]] const uint32_t n2_off = BLOCK2 ? (uint32_t)(reinterpret_cast<const unsigned char*>(S.nodes2) - geom) : 0u;
	const uint32_t n_surf = S.n_surfaces;
	const uint32_t n_order = ((n_surf + 7u) / 8u) * 8u;

	// wave-uniform: the segment being handed out, the unit of it staged in LDS, the root node of the unit's surface
	uint32_t seg_pos = 0, seg_end = 0, unit_pos = 0, unit_n = 0, order_pos = 0;
	int unit_surf = -1;
	bool more = true;
	uint2 root_nd = make_uint2(0, 0);
	// per lane: the walk in progress (core::mesh::intersect's locals)
	bool busy = false;
	uint2 nd = make_uint2(0, KD_LEAF);          // the node the lane stands on
	uint32_t slot = 0, k = 0, best_tri = 0;
	int sp = 0;
	float min_dist = 0, max_dist = 0, best_t = -1.0f, bb1 = 0, bb2 = 0;
	V3 o = {0, 0, 0}, d = {0, 0, 1};
#ifdef PTX_WF_PROF
	uint32_t pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 trips, 1 busy lanes, 2 node steps, 3 triangle tests, 4 hand-outs, 5 unit fetches, 6 pops
	uint64_t tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const uint64_t t_start = __builtin_amdgcn_s_memtime();
	uint32_t walk_steps = 0, walk_max = 0;
#endif
	const uint64_t clock0 = W.wave_clock ? __builtin_amdgcn_s_memtime() : 0;
	bool worked = false;

	auto trip = [&]() __attribute__((always_inline)) -> bool {   // one trip of the loop; true = the wave is through
		WFPROF(0);
		const uint64_t idle_m = __ballot(!busy);
		if (more && ((uint32_t)__popcll(idle_m) >= kWfRefillMin || ~idle_m == 0)) {
			if (unit_pos == unit_n) {
				WFT0();
				if (seg_pos == seg_end) {
					for (;;) {   // next segment: this XCD's surfaces first; one atomic on the surface's cursor per segment
						if (unit_surf < 0) {
							int u = -1;
							for (; order_pos < n_order; order_pos++) {
								const int c = wf_surface_at(S.wf_order, xcd, order_pos, n_surf);
								if (c >= 0 && W.ctl[kWfCtlSeg + c] != 0) { u = c; break; }
							}
							u = __builtin_amdgcn_readfirstlane(u);
							if (u < 0) { more = false; break; }
							unit_surf = u;
							const uint2 r = BLOCK2 ? S.roots2[u] : S.nodes[S.surfaces[u].kd_root];
							root_nd = make_uint2(__builtin_amdgcn_readfirstlane(r.x), __builtin_amdgcn_readfirstlane(r.y));
						}
						uint32_t g = 0;
						if (lane == 0) g = atomicAdd(&W.ctl[kWfCtlCur + 64u * (uint32_t)unit_surf], 1u);
						g = __builtin_amdgcn_readfirstlane(g);
						if (g < W.ctl[kWfCtlSeg + unit_surf]) {
							const uint2 sg = W.seg[(size_t)unit_surf * W.seg_cap + g];
							seg_pos = __builtin_amdgcn_readfirstlane(sg.x);
							seg_end = seg_pos + __builtin_amdgcn_readfirstlane(sg.y);
							break;
						}
						unit_surf = -1;
						order_pos++;
					}
				}
				if (more) {
					WFPROF(5);
					unit_n = seg_end - seg_pos < kWfUnit ? seg_end - seg_pos : kWfUnit;
					unit_pos = 0;
					if (lane < unit_n) {
						const size_t e = (size_t)seg_pos + lane;
						s_ray[wave][lane][0] = W.qent[2 * e];
						s_ray[wave][lane][1] = W.qent[2 * e + 1];
					}
					seg_pos += unit_n;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				}
				WFT1(1);
			}
			if (more) {
				WFT0();
				const uint32_t avail = unit_n - unit_pos;
				const uint32_t r = rank_in(idle_m);
				if (!busy && r < avail) {
					WFPROF(4);
#ifdef PTX_WF_PROF
					walk_max = walk_steps > walk_max ? walk_steps : walk_max; walk_steps = 0;
#endif
					const uint32_t e = unit_pos + r;
					const float4 e0 = s_ray[wave][e][0], e1 = s_ray[wave][e][1];
					slot = __float_as_uint(e0.w);
					o = mk(e0.x, e0.y, e0.z);
					d = mk(e1.x, e1.y, e1.z);
					const SurfaceRec& sf = S.surfaces[unit_surf];
					const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
					float nr, fr;
					if (aabb_test_box(sf.box, o, inv, nr, fr)) {   // mesh.cpp:308-315 (the classification saw the same test pass)
						busy = true;
						nd = root_nd; min_dist = nr; max_dist = fr; sp = 0; k = 0; best_t = -1.0f;
					} else {
						W.pair_hit[slot] = make_float4(-1.0f, 0.f, 0.f, 0.f);
					}
				}
				const uint32_t n_idle = (uint32_t)__popcll(idle_m);
				unit_pos += n_idle < avail ? n_idle : avail;
				WFT1(2);
			}
		}
		if (__ballot(busy) == 0) return !more;
		worked = true;
		if (busy) WFPROF(1);
		// The trip itself is straight-line code under per-lane predicates: both kinds of step are computed by every lane and kept
		// where they apply (selects). With branches around the two kinds, every state variable a kind updates was copied at every join
		// (a third of the loop's vector instructions were moves) for nothing — a wave has lanes of both kinds in nearly every trip.
		// ---- the trip's fetches: child pair (16 bytes) of a branch, the next triangle record (48 bytes) of a leaf
		const bool leaf = (nd.y & 3u) == KD_LEAF;
		const bool branch = busy && !leaf;
		const uint32_t count = nd.y >> 2;   // of a leaf
		const bool tri = busy && leaf && k < count;
		// every lane fetches (no join, no copies): a lane with nothing to fetch reads the allocation's first 16 bytes
		const bool broot = BLOCK2 && branch && (nd.y & 16u);
		const uint32_t off = branch ? (BLOCK2 ? n2_off + (nd.y >> 5) * 16u : (nd.y >> 4) * 8u) : (tri ? tri_off + (nd.x + k) * 48u : 0u);
		const NodePair q0 = *reinterpret_cast<const NodePair*>(geom + off);
		float4 r1, r2;
		if (tri || broot) { r1 = *reinterpret_cast<const float4*>(geom + off + 16u); r2 = *reinterpret_cast<const float4*>(geom + off + 32u); }
		// ---- node step (mesh.cpp:327-370; see mesh_traverse): on `cur` with its child pair (kid0, kid1); `on` = the lane makes this step
		bool descend = false, dead_end = false;
		uint2 next_nd = nd;
		bool next_is_kid0 = true;
		auto node_step = [&](bool on, uint2 cur, uint2 kid0, uint2 kid1) {
			const uint32_t axis = cur.y & 3u;
			const float split = __uint_as_float(cur.x);
			const float oa = sel3(o, axis), da = sel3(d, axis);
			const float split_dist = (split - oa) / da;
			const bool has_l = cur.y & 4u, has_r = cur.y & 8u;
			const bool left_first = oa < split;
			// children sit at slot 0 (left, or right when there is no left) and slot 1 (right when both exist): kid0 / kid1
			const bool first_is_kid0 = left_first || !has_l, second_is_kid0 = !left_first || !has_l;
			const bool has_first = left_first ? has_l : has_r, has_second = left_first ? has_r : has_l;
			const bool outside = split_dist < 0 || split_dist > max_dist;           // only the near child (mesh.cpp:354-357)
			const bool far_only = !outside && split_dist < min_dist;                 // only the far child (:358-361)
			const bool both = on && !outside && !far_only;                           // near child now, far child set aside (:362-369)
			const bool push = both && has_second && sp < kWfMaxStack;
			if (push) {
				WFPROF(7);
				const uint2 c = second_is_kid0 ? kid0 : kid1;
				const uint4 ent = make_uint4(c.x, c.y, __float_as_uint(split_dist), __float_as_uint(max_dist));
				if (sp < kWfLdsStack2) s_stk[sp][threadIdx.x] = ent; else spill4[(sp - kWfLdsStack2) * 64] = ent;
			}
			sp += push ? 1 : 0;
			max_dist = both ? split_dist : max_dist;
			const bool has_next = far_only ? has_second : has_first;
			const bool k0 = far_only ? second_is_kid0 : first_is_kid0;
			descend = on ? has_next : descend;
			dead_end = on ? !has_next : dead_end;
			next_is_kid0 = on ? k0 : next_is_kid0;
			next_nd = (on && has_next) ? (k0 ? kid0 : kid1) : next_nd;
		};
		if (branch) WFPROF(2);
		node_step(branch, nd, make_uint2(q0.x, q0.y), make_uint2(q0.z, q0.w));
		if constexpr (BLOCK2) {
			// second level of the block: the child just chosen is a branch, and its child pair came with the block
			const bool two = broot && descend && (next_nd.y & 3u) != KD_LEAF;
			if (two) WFPROF(2);
			const float4 gk = next_is_kid0 ? r1 : r2;
			node_step(two, next_nd, make_uint2(__float_as_uint(gk.x), __float_as_uint(gk.y)), make_uint2(__float_as_uint(gk.z), __float_as_uint(gk.w)));
		}
		// ---- triangle test: nearest triangle of the leaf with t <= max_dist; ties keep the first (mesh.cpp:381-389)
		if (tri) WFPROF(3);
		const PRay pr = pack_ray(o, d);
		const float4 r0 = make_float4(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w));
		float be, ga;
		const float t = tri_test_pk(r0, r1, make_float2(r2.x, r2.y), pr, be, ga);
		const bool better = tri && t >= 0 && t <= max_dist && (t < best_t || !(best_t >= 0));
		best_t = better ? t : best_t; bb1 = better ? be : bb1; bb2 = better ? ga : bb2; best_tri = better ? __float_as_uint(r2.z) : best_tri;
		const uint32_t k1 = k + 1u;
		const bool leaf_done = tri && k1 == count;
		const bool hit = leaf_done && best_t >= 0;
		// ---- the walk ends with a hit, goes on to the next pending subtree (mesh.cpp:317-325), or ends with nothing
		const bool need_pop = busy && ((branch && dead_end) || (leaf && !tri) || (leaf_done && !hit));   // `leaf && !tri`: an empty leaf (the builder makes none)
		const bool miss = need_pop && sp == 0;
		if (hit || miss) W.pair_hit[slot] = hit ? make_float4(best_t, __uint_as_float(best_tri), bb1, bb2) : make_float4(-1.0f, 0.f, 0.f, 0.f);
		const bool pop = need_pop && sp > 0;
		if (pop) WFPROF(6);
		sp -= pop ? 1 : 0;
		uint4 ent = s_stk[sp < kWfLdsStack2 ? sp : 0][threadIdx.x];       // read by every lane; kept by the popping ones
		if (__ballot(pop && sp >= kWfLdsStack2) != 0) { if (pop && sp >= kWfLdsStack2) ent = spill4[(sp - kWfLdsStack2) * 64]; }
		nd = pop ? make_uint2(ent.x, ent.y) : ((branch && descend) ? next_nd : nd);
		min_dist = pop ? __uint_as_float(ent.z) : min_dist;
		max_dist = pop ? __uint_as_float(ent.w) : max_dist;
		k = (pop || (branch && descend)) ? 0u : (tri ? k1 : k);
		best_t = (pop || (branch && descend)) ? -1.0f : best_t;
		busy = busy && !hit && !miss;
		return false;
	};
	for (;;) {
		if (trip()) break;
#if PTX_WF_UNROLL >= 2
		if (trip()) break;   // (several trips per iteration: the copies of the walk state at the loop's back edge are made once per iteration)
#endif
#if PTX_WF_UNROLL >= 3
		if (trip()) break;
#endif
#if PTX_WF_UNROLL >= 4
		if (trip()) break;
#endif
	}
	if (W.wave_clock && worked && lane == 0) {   // how long this wave ran: the host compares the sum over waves with waves x the longest run
		const uint64_t run = __builtin_amdgcn_s_memtime() - clock0;
		atomicAdd(reinterpret_cast<unsigned long long*>(W.ctl + kWfCtlClock), (unsigned long long)run);
		atomicAdd(&W.ctl[kWfCtlClock + 2], 1u);
		atomicMax(&W.ctl[kWfCtlClock + 3], (uint32_t)(run > 0xFFFFFFFFull ? 0xFFFFFFFFull : run));
	}
#ifdef PTX_WF_PROF
	for (int q = 0; q < 8; q++) {
		if (pt[q]) atomicAdd(&W.ctl[kWfCtlProf + 2 * q], pt[q]);
		if (pl[q]) atomicAdd(&W.ctl[kWfCtlProf + 2 * q + 1], pl[q]);
	}
	tc[0] = __builtin_amdgcn_s_memtime() - t_start;
	if (lane == 0) for (int q = 0; q < 6; q++) atomicAdd(&W.ctl[kWfCtlProf + 16 + q], (uint32_t)(tc[q] >> 10));
	if (lane == 0) { atomicMax(&W.ctl[kWfCtlProf + 24], (uint32_t)(tc[0] >> 10)); atomicMax(&W.ctl[kWfCtlProf + 25], pt[0]); }
	atomicMax(&W.ctl[kWfCtlProf + 26], walk_steps > walk_max ? walk_steps : walk_max);
	if (lane == 0) { const uint32_t kc = (uint32_t)(tc[0] >> 10); atomicAdd(&W.ctl[kWfCtlProf + 32 + (kc ? 31 - __builtin_clz(kc) : 0)], 1u); if (pt[1] == 0) atomicAdd(&W.ctl[kWfCtlProf + 32 + 31], 1u); }
#endif
}

// ------------------------------------------------------------------------------------ merge
// renderer::intersect (core/renderer.cpp:645-671) over scene::model::intersect (scene/model.cpp:20-72) with core::mesh::intersect
// replaced by the lookup of the pair's result: the loops, comparisons and the local -> world distance are scene_traverse's.
// The pair results of one ray, fetched AHEAD of their use: where they start and which surfaces they belong to (head), then the first four
// of them in one go (a ray of the 24-surface atrium has 3.2). Read one by one inside the merge loops, every result was a memory round
// trip of its own — six to eight dependent ones per path and step in the shade kernel, at four waves per SIMD.
constexpr uint32_t kWfPre = 4;
struct PairPre { uint32_t first; unsigned long long mask; float4 r0, r1, r2, r3; };   // (separate fields: an indexed array would live in scratch)
DEV void wf_pre_head(const WfBuffers& W, uint32_t i, PairPre& q) { q.first = W.first[i]; q.mask = W.mask[i]; }
DEV void wf_pre_results(const WfBuffers& W, PairPre& q) {
	// four unconditional 16-byte loads (clamped to the pool): what lies beyond the ray's own results is never looked at (wf_pre_get is
	// asked for result k only when the ray has more than k)
	const uint32_t last = W.pool_cap - 1u;
	const uint32_t a = q.first < last ? q.first : last, b = q.first + 1u < last ? q.first + 1u : last;
	const uint32_t c = q.first + 2u < last ? q.first + 2u : last, d = q.first + 3u < last ? q.first + 3u : last;
	q.r0 = W.pair_hit[a]; q.r1 = W.pair_hit[b]; q.r2 = W.pair_hit[c]; q.r3 = W.pair_hit[d];
}
DEV float4 wf_pre_get(const WfBuffers& W, const PairPre& q, uint32_t k) {   // the k-th result of the ray
	if (k >= kWfPre) return W.pair_hit[q.first + k];
	const float4 a = k & 1u ? q.r1 : q.r0, b = k & 1u ? q.r3 : q.r2;
	return k & 2u ? b : a;
}

DEV bool wf_closest(const DevScene& S, const WfBuffers& W, const PairPre& q, V3 o, V3 d, SceneHit& best) {
	best.dist = -1.0f;
	best.surface = -1;
	best.tri = 0; best.b1 = 0; best.b2 = 0;
	const unsigned long long mine = q.mask;
	uint32_t p = 0;
	uint32_t cur_space = 0xFFFFFFFFu;
	V3 lo = o, ld = d, inv = d;
	for (int m = 0; m < S.n_models; m++) {
		const ModelRec& M = S.models[m];
		const unsigned long long range = (M.n_surfaces >= 64 ? ~0ull : ((1ull << M.n_surfaces) - 1ull)) << M.first_surface;
		unsigned long long bits = mine & range;
		if (__ballot(bits != 0) == 0) continue;
		const uint32_t spc = S.model_space[m];
		if (spc != cur_space) { to_space(S.spaces[spc], o, d, lo, ld, inv); cur_space = spc; }
		MeshHit nearest;
		nearest.t = -1.0f; nearest.b1 = 0; nearest.b2 = 0; nearest.tri = 0;
		int hit_surface = -1;
		while (bits) {
			const int u = __builtin_ctzll(bits);
			bits &= bits - 1ull;
			const float4 h = wf_pre_get(W, q, p++);
			if (!(h.x >= 0)) continue;   // this surface reported no hit
			if (h.x < nearest.t || !(nearest.t >= 0)) { nearest.t = h.x; nearest.tri = __float_as_uint(h.y); nearest.b1 = h.z; nearest.b2 = h.w; hit_surface = u; }
		}
		if (!(nearest.t >= 0)) continue;
		const float wd = length(mulmv(M.basis, ld * nearest.t));   // local -> world distance (model.cpp:62-63)
		if (!(wd >= 0)) continue;
		if (wd < best.dist || !(best.dist >= 0)) { best.dist = wd; best.surface = hit_surface; best.tri = nearest.tri; best.b1 = nearest.b1; best.b2 = nearest.b2; }
	}
	return best.surface >= 0;
}

__global__ void __launch_bounds__(kWfBlock) k_wf_merge_batch(DevScene S0, IntersectArgs A, size_t first_ray, uint32_t n, WfBuffers W, const ModelRec* __restrict__ t_models,
                                                            const SurfaceRec* __restrict__ t_surfaces, const SpaceRec* __restrict__ t_spaces,
                                                            const uint32_t* __restrict__ t_model_space) {
	DevScene S = S0;
	S.models = t_models; S.surfaces = t_surfaces; S.spaces = t_spaces; S.model_space = t_model_space;
	const uint32_t i = blockIdx.x * kWfBlock + threadIdx.x;
	if (i >= n) return;
	const size_t gi = first_ray + i;
	const V3 o = mk(A.ox[gi], A.oy[gi], A.oz[gi]), d = mk(A.dx[gi], A.dy[gi], A.dz[gi]);
	PairPre q;
	wf_pre_head(W, i, q);
	wf_pre_results(W, q);
	SceneHit h;
	const bool hit = wf_closest(S, W, q, o, d, h);
	write_hit_outputs(S, S.shade, A, gi, hit, h);
}


// ------------------------------------------------------------------------------------ integrator on queues
// The path stream of a render slab: SoA of float4, the fused kernel's stream entry plus flags in the id word, and the entry's shadow
// request beside it (kernels.hpp: WfStream). One STEP = classify (the extend ray and the shadow ray of every entry) -> traverse ->
// shade (this file's k_wf_shade: the answer to the entry's shadow request, closest hit, one path vertex, next entry).
DEV uint32_t wf_id(float4 q0) { return __float_as_uint(q0.w) & kWfIdMask; }
DEV uint32_t wf_flags(float4 q0) { return __float_as_uint(q0.w) & ~kWfIdMask; }

// ray t of a step with n_in entries: t < n_in: the extend ray of entry t (entries that wait for a shadow answer have none); else the
// shadow ray of entry t - n_in
struct StreamRays {
	const float4* q;   // [4][cap]
	const float4* r;   // [3][cap]
	uint32_t cap;
	DEV uint32_t count(uint32_t n_in) const { return 2u * n_in; }
	DEV bool valid(uint32_t t, uint32_t n_in) const {
		const uint32_t i = t < n_in ? t : t - n_in;
		const uint32_t f = wf_flags(q[i]);
		return t < n_in ? !(f & (kWfZombie | kWfPending)) : (f & kWfRequest) != 0;
	}
	DEV void load(uint32_t t, uint32_t n_in, V3& o, V3& d) const {
		if (t < n_in) { const float4 a = q[t], b = q[cap + t]; o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); }
		else { const uint32_t i = t - n_in; const float4 a = r[i], b = r[cap + i]; o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); }
	}
};

// camera paths [first, first + n) of the pass -> stream entries 0 .. n-1 (scene::camera::get_ray; renderer.cpp:359-370)
__global__ void __launch_bounds__(kWfBlock) k_wf_generate(DevScene S, RenderParams P, WfStream out, uint32_t cap, uint32_t first, uint32_t n, float4* __restrict__ sample_rad) {
	const uint32_t i = blockIdx.x * kWfBlock + threadIdx.x;
	if (i >= n) return;
	const uint32_t id = first + i;   // id within the pass: sample-major, pixel-minor
	if (P.bounces == 0) { sample_rad[id] = make_float4(0.f, 0.f, 0.f, 1.0f); return; }   // trace(0, ..) is black with alpha 1 (renderer.cpp:438-439)
	const uint32_t s_local = id / P.n_pixels;
	uint32_t p_local = id - s_local * P.n_pixels;
	if (P.pixels) p_local = P.pixels[p_local];
	const uint32_t px = P.x0 + p_local % P.w, py = P.y0 + p_local / P.w;
	V3 o, d;
	camera_ray(S, P, px, py, P.sample0 + s_local, o, d);
	out.q[i] = make_float4(o.x, o.y, o.z, __uint_as_float(i));
	out.q[cap + i] = make_float4(d.x, d.y, d.z, 1.0f);
	out.q[2 * (size_t)cap + i] = make_float4(1.0f, 1.0f, 0.f, 0.f);
	out.q[3 * (size_t)cap + i] = make_float4(0.f, __uint_as_float(0u), __uint_as_float(py * P.W + px), __uint_as_float(P.sample0 + s_local));
}

// renderer::intersect(shadow ray).has_hit() (renderer.cpp:509-511, intersection_worker.cpp:58-61) from the pair results: some surface
// reports a hit whose world distance is not NaN (scene_occluded's test)
DEV bool wf_any(const DevScene& S, const WfBuffers& W, const PairPre& q, V3 o, V3 d) {
	const unsigned long long mine = q.mask;
	uint32_t p = 0;
	uint32_t cur_space = 0xFFFFFFFFu;
	V3 lo = o, ld = d, inv = d;
	bool occ = false;
	for (int m = 0; m < S.n_models; m++) {
		const ModelRec& M = S.models[m];
		const unsigned long long range = (M.n_surfaces >= 64 ? ~0ull : ((1ull << M.n_surfaces) - 1ull)) << M.first_surface;
		unsigned long long bits = mine & range;
		if (__ballot(bits != 0) == 0) continue;
		const uint32_t spc = S.model_space[m];
		if (spc != cur_space) { to_space(S.spaces[spc], o, d, lo, ld, inv); cur_space = spc; }
		while (bits) {
			bits &= bits - 1ull;
			const float4 h = wf_pre_get(W, q, p++);
			if (h.x >= 0 && length(mulmv(M.basis, ld * h.x)) >= 0) occ = true;
		}
	}
	return occ;
}

// One step of every path of the stream: what the fused kernel does in its SHADOW sweep of the previous step (the answer to the entry's
// shadow request: the sun's contribution, or the fate of a shadow catcher) and in its SHADE sweep of this one (shade_vertex), with
// the survivors appended to the other stream buffer. 256-thread workgroups: no 128-register cap, nothing spills.
template <bool SUN, bool ALPHA, bool TEX, bool WORKER>
__global__ void __launch_bounds__(kWfBlock) k_wf_shade(DevScene S0, RenderParams P, WfBuffers W, WfStream in, WfStream out, uint32_t cap, uint32_t slab_first,
                                                      uint32_t* __restrict__ n_out, float4* __restrict__ sample_rad, const ModelRec* __restrict__ t_models,
                                                      const SurfaceRec* __restrict__ t_surfaces, const SpaceRec* __restrict__ t_spaces, const uint32_t* __restrict__ t_model_space) {
	DevScene S = S0;
	S.models = t_models; S.surfaces = t_surfaces; S.spaces = t_spaces; S.model_space = t_model_space;
	if (W.ctl[1]) return;   // this step's pairs did not fit the pool: nothing is emitted, the later steps of the slab find no entries
	__shared__ uint32_t s_wave_n[kWfBlock / 64];
	const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
	// the grid covers the most entries the step can have; the entry count itself is only known on the device, and the workgroups
	// beyond it leave at once (a loop over tiles in persistent workgroups keeps the scene tables live across iterations: 99-171 VGPRs
	// instead of 83-123)
	const uint32_t n_in = *W.n_in;
	if ((uint64_t)blockIdx.x * kWfBlock >= (uint64_t)n_in) return;
	const uint32_t i = blockIdx.x * kWfBlock + threadIdx.x;
	V3 o = {0, 0, 0}, d = {0, 0, 1}, T = {1, 1, 1}, L = {0, 0, 0};
	uint32_t id = 0, depth = 0, pass = 0, out_flags = 0;
	float key_px = 0.f, key_s = 0.f;
	bool emit = false;
	ShadowReq rq;
	rq.kind = REQ_NONE;
	if (i < n_in) {
		// everything this entry may need that does not depend on anything else is asked for at once: the entry, its shadow request, and
		// where the pair results of its two rays start; then the results themselves (second round trip), then the hit record (third)
		const float4 q0 = in.q[i], q1 = in.q[cap + i], q2 = in.q[2 * (size_t)cap + i], q3 = in.q[3 * (size_t)cap + i];
		const float4 r0 = in.r[i], r1 = in.r[cap + i], r2 = in.r[2 * (size_t)cap + i];
		PairPre pe, ps;
		wf_pre_head(W, i, pe);
		wf_pre_head(W, n_in + i, ps);
		wf_pre_results(W, ps);
		wf_pre_results(W, pe);
		o = mk(q0.x, q0.y, q0.z); id = wf_id(q0);
		const uint32_t flags = wf_flags(q0);
		d = mk(q1.x, q1.y, q1.z);
		T = mk(q1.w, q2.x, q2.y);
		L = mk(q2.z, q2.w, q3.x);
		key_px = q3.z; key_s = q3.w;
		{ const uint32_t dp = __float_as_uint(q3.y); depth = dp >> 16; pass = dp & 0xFFFFu; }
		float4* const result = sample_rad + slab_first + id;
		bool occluded = false;
		V3 x = {0, 0, 0};
		if (flags & kWfRequest) {
			x = mk(r2.x, r2.y, r2.z);
			occluded = wf_any(S, W, ps, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z));
		}
		if (flags & kWfPending) {
			// shadow catcher (renderer.cpp:513-519, 560-561; shading_worker.cpp:74-104): shadowed -> the path ends (trace() returns what it has,
			// the worker zeroes the colour); lit -> fully transparent: same depth, next pass, continued from behind the surface (x)
			if (occluded) *result = WORKER ? make_float4(0.f, 0.f, 0.f, 1.0f) : make_float4(L.x, L.y, L.z, 1.0f);
			else if (pass + 1u > 4096u) *result = make_float4(L.x, L.y, L.z, 1.0f);
			else { o = x; d = normalize(d); pass++; emit = true; }
		} else {
			if ((flags & kWfRequest) && !occluded) L = mk(L.x + x.x, L.y + x.y, L.z + x.z);   // the sun's contribution of the previous vertex
			if (flags & kWfZombie) *result = make_float4(L.x, L.y, L.z, 1.0f);
			else {
				SceneHit h;
				wf_closest(S, W, pe, o, d, h);
				const int state = shade_vertex<SUN, ALPHA, TEX, WORKER>(S, S.shade, P, __float_as_uint(key_px), __float_as_uint(key_s), depth, pass, h, o, d, T, L, rq);
				if (state == V_ALIVE) emit = true;
				else if (state == V_PENDING) { emit = true; out_flags = kWfPending; o = rq.x; }
				else if (rq.kind == REQ_ADD) { emit = true; out_flags = kWfZombie; }   // the path is over, its last sun sample is not
				else *result = make_float4(L.x, L.y, L.z, 1.0f);
				if (rq.kind != REQ_NONE) out_flags |= kWfRequest;
			}
		}
	}
	// survivors -> the other buffer: wave ballot + prefix, one counter fetch per tile
	const uint64_t em = __ballot(emit);
	if (lane == 0) s_wave_n[wave] = (uint32_t)__popcll(em);
	__syncthreads();
	uint32_t before = 0, total = 0;
	for (uint32_t w = 0; w < (uint32_t)(kWfBlock / 64); w++) { const uint32_t c = s_wave_n[w]; if (w < wave) before += c; total += c; }
	__syncthreads();
	if (threadIdx.x == 0) s_wave_n[0] = total ? atomicAdd(n_out, total) : 0u;
	__syncthreads();
	if (emit) {
		const uint32_t pos = s_wave_n[0] + before + rank_in(em);
		out.q[pos] = make_float4(o.x, o.y, o.z, __uint_as_float(id | out_flags));
		out.q[cap + pos] = make_float4(d.x, d.y, d.z, T.x);
		out.q[2 * (size_t)cap + pos] = make_float4(T.y, T.z, L.x, L.y);
		out.q[3 * (size_t)cap + pos] = make_float4(L.z, __uint_as_float((depth << 16) | pass), key_px, key_s);
		if (out_flags & kWfRequest) {
			out.r[pos] = make_float4(rq.o.x, rq.o.y, rq.o.z, 0.f);
			out.r[cap + pos] = make_float4(rq.d.x, rq.d.y, rq.d.z, 0.f);
			out.r[2 * (size_t)cap + pos] = make_float4(rq.x.x, rq.x.y, rq.x.z, 0.f);
		}
	}
}

// ------------------------------------------------------------------------------------ launchers
static int wf_classify_grid(int n_cu) { return n_cu * (int)(2048u / kWfTile); }   // persistent 1024-thread workgroups (51 VGPRs: two per CU)

// PTX_WF_KERNEL=1 (measurement): the nested-loop form of the traverse kernel instead of the one-loop form
static void launch_traverse(const DevScene& S, const WfBuffers& W, int n_cu, hipStream_t stream) {
	const char* const e = getenv("PTX_WF_KERNEL");   // read per launch: the tests switch it inside one process
	const bool nested = e && e[0] == '1';
	// the one-loop kernel reads a leaf's records in place (leaf-ordered copy) at 32-bit offsets from the nodes
	if (nested || !S.glb_leaf_ordered || S.geom_bytes > 0xFFFFFFFFull) hipLaunchKernelGGL(k_wf_traverse, dim3(wf_traverse_grid(n_cu)), dim3(kWfBlock), 0, stream, S, W, S.surfaces);
	else if (S.nodes2 && getenv("PTX_WF_BLOCK2")) hipLaunchKernelGGL(k_wf_traverse2<true>, dim3(wf_traverse_grid(n_cu)), dim3(kWfBlock), 0, stream, S, W, S.surfaces);
	else hipLaunchKernelGGL(k_wf_traverse2<false>, dim3(wf_traverse_grid(n_cu)), dim3(kWfBlock), 0, stream, S, W, S.surfaces);
}

// One slice of a batch: W.ctl must be zeroed, W.n_in == nullptr (the slice's ray count is known to the host)
hipError_t launch_wf_intersect(const DevScene& S, const IntersectArgs& A, size_t first_ray, uint32_t n, const WfBuffers& W, int n_cu, hipStream_t stream) {
	const SoaRays src{A.ox + first_ray, A.oy + first_ray, A.oz + first_ray, A.dx + first_ray, A.dy + first_ray, A.dz + first_ray};
	const int tiles = (int)((n + kWfTile - 1) / kWfTile);
	hipLaunchKernelGGL(k_wf_classify<SoaRays>, dim3(tiles < wf_classify_grid(n_cu) ? tiles : wf_classify_grid(n_cu)), dim3(kWfClassifyBlock), 0, stream, S, src, n, W, S.models, S.surfaces, S.spaces, S.model_space);
	launch_traverse(S, W, n_cu, stream);
	hipLaunchKernelGGL(k_wf_merge_batch, dim3((n + kWfBlock - 1) / kWfBlock), dim3(kWfBlock), 0, stream, S, A, first_ray, n, W, S.models, S.surfaces, S.spaces, S.model_space);
	return hipGetLastError();
}

hipError_t launch_wf_generate(const DevScene& S, const RenderParams& P, const WfStream& out, uint32_t cap, uint32_t first, uint32_t n, float4* sample_rad, hipStream_t stream) {
	hipLaunchKernelGGL(k_wf_generate, dim3((n + kWfBlock - 1) / kWfBlock), dim3(kWfBlock), 0, stream, S, P, out, cap, first, n, sample_rad);
	return hipGetLastError();
}

template <bool SUN, bool ALPHA, bool TEX, bool WORKER>
static void launch_shade_variant(const DevScene& S, const RenderParams& P, const WfBuffers& W, const WfStream& in, const WfStream& out, uint32_t cap, uint32_t max_in,
                                 uint32_t slab_first, uint32_t* n_out, float4* sample_rad, hipStream_t stream) {
	hipLaunchKernelGGL((k_wf_shade<SUN, ALPHA, TEX, WORKER>), dim3((max_in + kWfBlock - 1) / kWfBlock), dim3(kWfBlock), 0, stream, S, P, W, in, out, cap,
	                   slab_first, n_out, sample_rad, S.models, S.surfaces, S.spaces, S.model_space);
}

// One step of a slab: the rays of the *W.n_in entries of `in` (extend + shadow) through the queues, then one vertex per path into `out`;
// the number of entries written is ADDED to *n_out (device, zero before the step). Nothing here depends on the entry count, which
// only the device knows (`max_in` bounds it for the grids): the host enqueues the steps of a slab back to back.
// Kernel variants as in the fused integrator (launch_pass_mode).
hipError_t launch_wf_step(const DevScene& S, const RenderParams& P, const WfBuffers& W, const WfStream& in, const WfStream& out, uint32_t cap, uint32_t max_in,
                          uint32_t slab_first, uint32_t* n_out, float4* sample_rad, int n_cu, hipStream_t stream, hipEvent_t* ev) {
	const StreamRays src{in.q, in.r, cap};
	const int tiles = (int)((2ull * max_in + kWfTile - 1) / kWfTile);
	if (ev) (void)hipEventRecord(ev[0], stream);
	hipLaunchKernelGGL(k_wf_classify<StreamRays>, dim3(tiles < wf_classify_grid(n_cu) ? tiles : wf_classify_grid(n_cu)), dim3(kWfClassifyBlock), 0, stream, S, src, 0u, W, S.models, S.surfaces,
	                   S.spaces, S.model_space);
	if (ev) (void)hipEventRecord(ev[1], stream);
	launch_traverse(S, W, n_cu, stream);
	if (ev) (void)hipEventRecord(ev[2], stream);
	const bool sun = S.sun.present != 0, alpha = S.any_alpha != 0;
#define WF_SHADE(SUN_, ALPHA_, TEX_, WORKER_) launch_shade_variant<SUN_, ALPHA_, TEX_, WORKER_>(S, P, W, in, out, cap, max_in, slab_first, n_out, sample_rad, stream)
	if (P.integrator == 1u) { if (S.any_texture) WF_SHADE(true, true, true, true); else WF_SHADE(true, true, false, true); }
	else if (S.any_texture) WF_SHADE(true, true, true, false);
	else if (sun) { if (alpha) WF_SHADE(true, true, false, false); else WF_SHADE(true, false, false, false); }
	else { if (alpha) WF_SHADE(false, true, false, false); else WF_SHADE(false, false, false, false); }
#undef WF_SHADE
	if (ev) (void)hipEventRecord(ev[3], stream);
	return hipGetLastError();
}

}  // namespace ptx
