// C ABI of libptx_hip.so (declared in include/ptx.h). Host glue only: contexts, scene upload,
// pass scheduling, staging of host buffers. All arithmetic of the hot path lives in kernels.hip;
// there is no CPU fallback — GPU entry points fail with PTX_ERR_NO_DEVICE when no HIP device exists.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "kernels.hpp"

using namespace ptx;

namespace {

thread_local std::string g_err;
int set_err(int code, const std::string& m) { g_err = m; return code; }
#define HIP_TRY(expr)                                                                                           \
	do {                                                                                                        \
		hipError_t e_ = (expr);                                                                                 \
		if (e_ != hipSuccess) return set_err(PTX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
	} while (0)

bool is_device_ptr(const void* p) {
	hipPointerAttribute_t a;
	hipError_t e = hipPointerGetAttributes(&a, p);
	if (e != hipSuccess) { (void)hipGetLastError(); return false; }
	return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	hipError_t ensure(size_t bytes) {
		if (bytes <= cap) return hipSuccess;
		if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
		hipError_t e = hipMalloc(&p, bytes);
		if (e == hipSuccess) cap = bytes;
		return e;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// image::write's quantiser for sRGB colour channels as a step function (kernels.hip: srgb8): thr[k] = the smallest float v in [0, 1]
// with byte(v) >= k, byte(v) = static_cast<uint8_t>(powf(v, 1 / 2.2F) * 255 + 0.5F) evaluated with this process's libm — the
// reference's own call (image.cpp:143-154). Floats in [0, 1] order like their bit patterns, so each threshold is a bisection over bits.
void srgb_thresholds(float thr[256]) {
	auto byte_of = [](uint32_t bits) {
		float v;
		memcpy(&v, &bits, 4);
		return (uint32_t)(uint8_t)(std::pow(v, 1 / 2.2F) * 255 + 0.5F);
	};
	thr[0] = 0.0f;
	for (uint32_t k = 1; k < 256; k++) {
		uint32_t lo = 0, hi = 0x3F800000u;   // byte(lo) = 0 < k <= 255 = byte(hi)
		while (hi - lo > 1) {
			const uint32_t mid = lo + (hi - lo) / 2;
			if (byte_of(mid) >= k) hi = mid; else lo = mid;
		}
		memcpy(&thr[k], &hi, 4);
	}
}

constexpr size_t kLdsBudget = 160 * 1024;
constexpr size_t kLeafOrderMaxBytes = (size_t)1 << 40;   // never reached: see decide_mode  // per-CU LDS on gfx950; one workgroup may take all of it

}  // namespace

struct ptx_ctx {
	std::atomic<int> refs{1};   // the caller's handle + one per scene created on it: a scene may outlive ptx_ctx_destroy
	int device = 0;
	hipStream_t stream = nullptr;
	int n_cu = 0;
	std::mutex mu;
	DevBuf queues, sample_rad, counters, spill, stage_a, stage_b, pixel_list, srgb_thr;
	// workspace of the queue-based pipeline (wavefront.hip). A render advances two slabs of paths side by side, each on its own stream with
	// its own set: the end of one slab's traverse kernel (a few waves finishing walks of hundreds of dependent fetches) and the host's
	// wait for its entry count then run under the other slab's kernels. ptx_intersect_batch uses set 0 on the context's stream.
	struct WfSet {
		DevBuf qent, pair_hit, seg, first, mask, ctl, spill, stream_buf, flow;
		uint32_t* flow_host = nullptr;   // pinned copy of the flow words (kWfFlowWords)
		hipStream_t stream = nullptr;
		hipEvent_t done = nullptr;
	} wf[2];
	hipEvent_t wf_main_ev = nullptr;
	std::vector<hipEvent_t> events;
	// per-kernel timing of the last ptx_render that was given a stats pointer (ptx_ctx_set_timing / ptx_ctx_get_timing)
	bool timing_on = false;
	std::vector<hipEvent_t> step_events;
	ptx_kernel_timing timing{};
	// pixel list of the last sharded render (ptx_render_cfg::shard_*), kept on the device: a frame is usually rendered again
	// with the same sharding (sample ranges, benchmark steps)
	uint32_t list_key[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t list_len = 0;
};

struct ptx_scene {
	ptx_ctx* ctx = nullptr;
	FlatScene host;
	DevBuf d_models, d_surfaces, d_materials, d_nodes, d_refs, d_tris, d_shade, d_tex, d_texels, d_lut, d_spaces, d_model_space, d_wf_order;
	DevBuf d_res_nodes, d_res_refs, d_res_tris, d_texels_f, d_hot;
	DevScene dev{};
	double lds_area_share = 0;     // share of the surfaces' box area (sum over surfaces) that belongs to LDS-resident surfaces: how much of what a ray can enter is served from LDS
	double wf_pairs_per_ray = 0;   // queue-based pipeline: pairs (ray, entered surface) per ray seen so far on this scene, 0 = not yet measured
	bool leaf_ordered = true; // global-memory copy of the triangle records: per leaf reference (true) or per triangle (false)
	int mode = MODE_GLOBAL;   // where the traversal arrays live: MODE_GLOBAL / MODE_LDS / MODE_HYBRID (kernels.hip)
	size_t lds_bytes = 0;     // dynamic LDS of the kernels (resident arrays + shade records)
};

namespace {

size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }

// Residency plan -> kernel family. PTX_FORCE_GLOBAL / PTX_NO_HYBRID: measurement switches.
void decide_mode(ptx_scene* sc) {
	FlatScene& h = sc->host;
	plan_residency(h, kLdsBudget);
	if (getenv("PTX_FORCE_GLOBAL") || h.n_resident == 0) sc->mode = MODE_GLOBAL;
	else if (h.n_resident == h.surfaces.size()) sc->mode = MODE_LDS;
	else sc->mode = getenv("PTX_NO_HYBRID") ? MODE_GLOBAL : MODE_HYBRID;
	if (sc->mode == MODE_GLOBAL) for (auto& sr : h.surfaces) sr.lds_root = 0xFFFFFFFFu;
	sc->lds_bytes = sc->mode == MODE_GLOBAL ? 0 : h.res_bytes;
	{
		double all = 0, res = 0;
		for (const SurfaceRec& sr : h.surfaces) {
			const double ex = sr.bmax[0] - sr.bmin[0], ey = sr.bmax[1] - sr.bmin[1], ez = sr.bmax[2] - sr.bmin[2];
			const double a = (ex > 0 && ey > 0 && ez > 0) ? 2 * (ex * ey + ey * ez + ex * ez) : 0;
			all += a;
			if (sr.lds_root != 0xFFFFFFFFu) res += a;
		}
		sc->lds_area_share = all > 0 ? res / all : 0;
	}
	// Leaf-ordered records duplicate a triangle once per leaf that references it (12x on deep SAH trees) and save a dependent fetch per
	// test. Measured up to 144 MB of records (the 262 k-triangle atrium, against 25 MB per triangle + references): the leaf order still
	// wins by 8 % — the dependent fetch costs more than the cache footprint (profiles/round2_ab_layout_blocksize.txt), so the threshold
	// is out of reach of any scene that fits the other limits. PTX_LEAF_ORDER=0/1 overrides (measurement).
	sc->leaf_ordered = h.kd_refs.size() * 48 <= kLeafOrderMaxBytes;
	if (const char* e = getenv("PTX_LEAF_ORDER")) sc->leaf_ordered = e[0] != '0';
}

int upload_scene(ptx_scene* sc) {
	ptx_ctx* c = sc->ctx;
	HIP_TRY(hipSetDevice(c->device));
	FlatScene& h = sc->host;
	auto up = [&](DevBuf& b, const void* src, size_t bytes, size_t padded) -> hipError_t {
		hipError_t e = b.ensure(std::max<size_t>(padded, 16));
		if (e != hipSuccess) return e;
		e = hipMemsetAsync(b.p, 0, std::max<size_t>(padded, 16), c->stream);
		if (e != hipSuccess) return e;
		return bytes ? hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess;
	};
	for (ModelRec& mr : h.models) interleave_boxes(mr);
	for (SurfaceRec& sr : h.surfaces) interleave_boxes(sr);
	HIP_TRY(up(sc->d_models, h.models.data(), h.models.size() * sizeof(ModelRec), h.models.size() * sizeof(ModelRec)));
	HIP_TRY(up(sc->d_materials, h.materials.data(), h.materials.size() * sizeof(MaterialRec), h.materials.size() * sizeof(MaterialRec)));
	HIP_TRY(up(sc->d_refs, h.kd_refs.data(), h.kd_refs.size() * 4, pad16(h.kd_refs.size() * 4)));
	HIP_TRY(up(sc->d_tris, h.hitrec.data(), h.hitrec.size() * sizeof(HitRec), h.hitrec.size() * sizeof(HitRec)));
	decide_mode(sc);   // sets SurfaceRec::lds_root: before the surface table goes up
	{
		// Hot hit records: what LDS is left beside the resident geometry takes the hit records (144 B) of the LARGEST triangles — where most
		// hits land (Cornell: walls, boxes, light) — so that shading them costs nine LDS reads instead of a round of nine global gathers
		// (9.8 % of the headline kernel's time, profiles/round3_clk_fused.txt). A triangle's slot travels in the top byte of the
		// triangle word of its traversal records (id | slot << 24; 0xFF = not hot), which needs ids below 2^24. PTX_NO_HOT_HITREC: measurement.
		const size_t n_tri = h.hitrec.size();
		const bool packed = n_tri < ((size_t)1 << 24);
		std::vector<uint8_t> slot_of(n_tri, 0xFF);
		h.hot_hitrec.clear();
		if (packed && sc->mode != MODE_GLOBAL && !getenv("PTX_NO_HOT_HITREC")) {
			const size_t left = kLdsBudget > sc->lds_bytes ? kLdsBudget - sc->lds_bytes : 0;
			const size_t n_hot = std::min<size_t>({(size_t)255, left / sizeof(HitRec), n_tri});
			if (n_hot) {
				std::vector<float> area(n_tri, 0.f);
				for (size_t si = 0; si < h.surfaces.size(); si++) {
					const int32_t* rg = &h.surf_range[8 * si];
					const ModelRec& mr = h.models[h.surfaces[si].model];
					double sc2 = 0;   // mean squared length of the basis columns: local -> world area scale (ranking only)
					for (int k = 0; k < 9; k++) sc2 += (double)mr.basis[k] * mr.basis[k];
					sc2 /= 3.0;
					for (int32_t t = rg[2]; t < rg[2] + rg[3]; t++) {
						const HitRec& r = h.hitrec[(size_t)t];
						const double e1[3] = {(double)r.b[0] - r.a[0], (double)r.b[1] - r.a[1], (double)r.b[2] - r.a[2]}, e2[3] = {(double)r.c[0] - r.a[0], (double)r.c[1] - r.a[1], (double)r.c[2] - r.a[2]};
						const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
						area[(size_t)t] = (float)(0.5 * std::sqrt(cx * cx + cy * cy + cz * cz) * sc2);
					}
				}
				std::vector<uint32_t> idx(n_tri);
				for (size_t i = 0; i < n_tri; i++) idx[i] = (uint32_t)i;
				std::partial_sort(idx.begin(), idx.begin() + (std::ptrdiff_t)n_hot, idx.end(), [&](uint32_t a, uint32_t b) { return area[a] != area[b] ? area[a] > area[b] : a < b; });
				for (size_t k = 0; k < n_hot; k++) { slot_of[idx[k]] = (uint8_t)k; h.hot_hitrec.push_back(h.hitrec[idx[k]]); }
				sc->lds_bytes += n_hot * sizeof(HitRec);
			}
		}
		auto pack = [&](std::vector<TriIsect>& recs) {
			for (TriIsect& r : recs) {
				uint32_t w; memcpy(&w, &r.p0, 4);
				const uint32_t id = packed ? (w & 0x00FFFFFFu) : w;
				w = packed ? (id | ((uint32_t)slot_of[id] << 24)) : id;
				memcpy(&r.p0, &w, 4);
			}
		};
		pack(h.tri_isect); pack(h.res_tris);
		sc->dev.tri_id_mask = packed ? 0x00FFFFFFu : 0xFFFFFFFFu;
		sc->dev.n_hot = (uint32_t)h.hot_hitrec.size();
	}
	HIP_TRY(up(sc->d_surfaces, h.surfaces.data(), h.surfaces.size() * sizeof(SurfaceRec), h.surfaces.size() * sizeof(SurfaceRec)));
	// KD nodes and the global-memory triangle records share ONE allocation: the queue-based traverse kernel addresses both as
	// `base + 32-bit offset` (wavefront.hip). + 16: the child-pair fetch of the last branch may read one node past the end.
	const size_t nodes_bytes = (pad16(h.kd_nodes.size() * 8) + 16 + 255) & ~(size_t)255;
	size_t isect_bytes = 0;
	std::vector<TriIsect> leaf;
	if (sc->mode != MODE_LDS) {
		if (sc->leaf_ordered) {
			// surfaces that stay in L2/HBM: one record per leaf reference, in leaf order, so that a leaf's triangles are one
			// contiguous run and the reference -> record indirection is gone (Geom::leaf_ordered)
			leaf.resize(h.kd_refs.size());
			for (size_t r = 0; r < leaf.size(); r++) leaf[r] = h.tri_isect[h.kd_refs[r]];
			isect_bytes = leaf.size() * 48;
		} else {
			// one record per TRIANGLE, reached through the leaf references: an extra dependent fetch per test, but a working set
			// (nodes + refs + records) several times smaller when leaves share many triangles
			isect_bytes = h.tri_isect.size() * 48;
		}
	}
	// measurement (PTX_WF_BLOCK2 at scene creation): a second copy of the nodes in 2-level blocks (wavefront.hip: BLOCK2) — a branch at even
	// depth owns three 16-byte pairs: its children, then the children of each child
	std::vector<uint2> nodes2, roots2;
	if (getenv("PTX_WF_BLOCK2") && sc->mode != MODE_LDS && sc->leaf_ordered) {
		auto W1 = [](uint32_t old_w1, bool blockroot, uint32_t pair) { return (old_w1 & 15u) | (blockroot ? 16u : 0u) | (pair << 5); };
		auto alloc_block = [&]() { const uint32_t pair = (uint32_t)(nodes2.size() / 2); nodes2.resize(nodes2.size() + 6, make_uint2(0, 0)); return pair; };
		std::vector<std::pair<uint32_t, uint32_t>> work;   // (old index of an even-depth branch, first pair of its block)
		for (const SurfaceRec& sr : h.surfaces) {
			const KdNode r = h.kd_nodes[sr.kd_root];
			if ((r.w1 & 3u) == KD_LEAF) { roots2.push_back(make_uint2(r.w0, r.w1)); continue; }
			const uint32_t b = alloc_block();
			roots2.push_back(make_uint2(r.w0, W1(r.w1, true, b)));
			work.push_back({sr.kd_root, b});
			while (!work.empty()) {
				const auto [old, blk] = work.back();
				work.pop_back();
				const KdNode P = h.kd_nodes[old];
				const uint32_t nk = ((P.w1 >> 2) & 1u) + ((P.w1 >> 3) & 1u), li = P.w1 >> 4;
				for (uint32_t i = 0; i < nk; i++) {
					const KdNode K = h.kd_nodes[li + i];
					if ((K.w1 & 3u) == KD_LEAF) { nodes2[2 * (size_t)blk + i] = make_uint2(K.w0, K.w1); continue; }
					const uint32_t kp = blk + 1 + i;   // the pair of K's children inside P's block
					nodes2[2 * (size_t)blk + i] = make_uint2(K.w0, W1(K.w1, false, kp));
					const uint32_t ng = ((K.w1 >> 2) & 1u) + ((K.w1 >> 3) & 1u), gi = K.w1 >> 4;
					for (uint32_t j = 0; j < ng; j++) {
						const KdNode G = h.kd_nodes[gi + j];
						if ((G.w1 & 3u) == KD_LEAF) { nodes2[2 * (size_t)kp + j] = make_uint2(G.w0, G.w1); continue; }
						const uint32_t b2 = alloc_block();
						nodes2[2 * (size_t)kp + j] = make_uint2(G.w0, W1(G.w1, true, b2));
						work.push_back({gi + j, b2});
					}
				}
			}
		}
	}
	const size_t isect_pad = (std::max<size_t>(isect_bytes, 16) + 255) & ~(size_t)255;
	const size_t n2_bytes = nodes2.size() * 8 + (nodes2.empty() ? 0 : 64), r2_bytes = (roots2.size() * 8 + 255) & ~(size_t)255;
	const size_t geom_total = nodes_bytes + isect_pad + (nodes2.empty() ? 0 : n2_bytes + r2_bytes);
	HIP_TRY(sc->d_nodes.ensure(geom_total));
	HIP_TRY(hipMemsetAsync(sc->d_nodes.p, 0, geom_total, c->stream));
	if (!nodes2.empty()) {
		HIP_TRY(hipMemcpyAsync((char*)sc->d_nodes.p + nodes_bytes + isect_pad, nodes2.data(), nodes2.size() * 8, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync((char*)sc->d_nodes.p + nodes_bytes + isect_pad + n2_bytes, roots2.data(), roots2.size() * 8, hipMemcpyHostToDevice, c->stream));
	}
	if (!h.kd_nodes.empty()) HIP_TRY(hipMemcpyAsync(sc->d_nodes.p, h.kd_nodes.data(), h.kd_nodes.size() * 8, hipMemcpyHostToDevice, c->stream));
	if (isect_bytes)
		HIP_TRY(hipMemcpyAsync((char*)sc->d_nodes.p + nodes_bytes, sc->leaf_ordered ? (const void*)leaf.data() : (const void*)h.tri_isect.data(), isect_bytes, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));   // `leaf` is a local
	std::vector<TriIsect>().swap(leaf);
	if (sc->mode != MODE_GLOBAL) {
		HIP_TRY(up(sc->d_res_nodes, h.res_nodes.data(), h.res_nodes.size() * 8, pad16(h.res_nodes.size() * 8)));
		HIP_TRY(up(sc->d_res_refs, h.res_refs.data(), h.res_refs.size() * 4, pad16(h.res_refs.size() * 4)));
		HIP_TRY(up(sc->d_res_tris, h.res_tris.data(), h.res_tris.size() * 48, h.res_tris.size() * 48));
		HIP_TRY(up(sc->d_hot, h.hot_hitrec.data(), h.hot_hitrec.size() * sizeof(HitRec), h.hot_hitrec.size() * sizeof(HitRec)));
	}
	HIP_TRY(up(sc->d_shade, h.shade.data(), h.shade.size() * sizeof(ShadeRec), h.shade.size() * sizeof(ShadeRec)));
	HIP_TRY(up(sc->d_tex, h.textures.data(), h.textures.size() * sizeof(TexRec), h.textures.size() * sizeof(TexRec)));
	HIP_TRY(up(sc->d_texels, h.texels.data(), h.texels.size(), pad16(h.texels.size())));
	HIP_TRY(up(sc->d_texels_f, h.texels_f.data(), h.texels_f.size() * 4, pad16(h.texels_f.size() * 4)));
	{   // image::read: value = byte / 255.0F; sRGB colour channels: math::pow(value, 2.2F) (image.cpp:135-138) — same libm call, once per byte value
		float lut[256];
		for (int b = 0; b < 256; b++) lut[b] = std::pow(b / 255.0F, 2.2F);
		HIP_TRY(up(sc->d_lut, lut, sizeof lut, sizeof lut));
	}
	HIP_TRY(up(sc->d_spaces, h.spaces.data(), h.spaces.size() * sizeof(SpaceRec), h.spaces.size() * sizeof(SpaceRec)));
	HIP_TRY(up(sc->d_model_space, h.model_space.data(), h.model_space.size() * 4, h.model_space.size() * 4));
	{
		// The order in which the queue-based traverse kernel starts the surfaces' queues. A render's steps: largest tree first, so that
		// a launch ends on the queues of the short walks and the few long walks (hundreds of dependent fetches in the big trees) start
		// early — atrium 1080p 446 -> 471 Msamples/s, 4K / 16 bounces 394 -> 418, jack-of-blades 2229 -> 2326; smallest first: no change
		// (profiles/round3_surface_order.txt). ptx_intersect_batch's launches keep the surface order: on its 5-8 M-ray slices the sorted
		// order was 10 % slower on bounce rays (2 % faster on camera rays). Second half of the table: the batch order.
		// PTX_WF_ORDER / PTX_WF_ORDER_BATCH = 0 surface order, 1 largest tree first, 2 smallest first (measurement).
		const size_t ns = h.surfaces.size();
		std::vector<uint32_t> order(2 * ns);
		auto fill = [&](uint32_t* o, int om) {
			for (size_t i = 0; i < ns; i++) o[i] = (uint32_t)i;
			if (om != 0) std::stable_sort(o, o + ns, [&](uint32_t a, uint32_t b) {
				const int32_t na = h.surf_range[8 * a + 5], nb = h.surf_range[8 * b + 5];   // KD nodes of the surface
				return om == 2 ? na < nb : na > nb;
			});
		};
		const char *oe = getenv("PTX_WF_ORDER"), *ob = getenv("PTX_WF_ORDER_BATCH");
		fill(order.data(), oe ? atoi(oe) : 1);
		fill(order.data() + ns, ob ? atoi(ob) : 0);
		HIP_TRY(up(sc->d_wf_order, order.data(), order.size() * 4, order.size() * 4));
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	DevScene& d = sc->dev;
	d.models = (const ModelRec*)sc->d_models.p;
	d.surfaces = (const SurfaceRec*)sc->d_surfaces.p;
	d.materials = (const MaterialRec*)sc->d_materials.p;
	d.nodes = (const uint2*)sc->d_nodes.p;
	d.refs = (const uint32_t*)sc->d_refs.p;
	d.tris = (const float4*)sc->d_tris.p;
	d.tri_isect = isect_bytes ? (const float4*)((const char*)sc->d_nodes.p + nodes_bytes) : nullptr;
	d.geom_bytes = (uint64_t)geom_total;
	d.nodes2 = nodes2.empty() ? nullptr : (const uint2*)((const char*)sc->d_nodes.p + nodes_bytes + isect_pad);
	d.roots2 = nodes2.empty() ? nullptr : (const uint2*)((const char*)sc->d_nodes.p + nodes_bytes + isect_pad + n2_bytes);
	d.res_nodes = (const uint2*)sc->d_res_nodes.p;
	d.res_refs = (const uint32_t*)sc->d_res_refs.p;
	d.res_tris = (const float4*)sc->d_res_tris.p;
	d.hot_hitrec = (const float4*)sc->d_hot.p;
	d.hot_lds = nullptr;
	d.n_res_nodes = (uint32_t)h.res_nodes.size();
	d.n_res_refs = (uint32_t)h.res_refs.size();
	d.n_res_tris = (uint32_t)h.res_tris.size();
	d.shade = (const ShadeRec*)sc->d_shade.p;
	d.spaces = (const SpaceRec*)sc->d_spaces.p;
	d.tex = (const TexRec*)sc->d_tex.p;
	d.texels = (const uint8_t*)sc->d_texels.p;
	d.texels_f = (const float*)sc->d_texels_f.p;
	d.srgb_lut = (const float*)sc->d_lut.p;
	d.glb_leaf_ordered = sc->leaf_ordered ? 1u : 0u;
	d.any_texture = (h.any_texture || h.env_tex >= 0) ? 1u : 0u;   // the TEX kernels also carry the environment lookup
	d.env_tex = h.env_tex;
	d.model_space = (const uint32_t*)sc->d_model_space.p;
	d.wf_order = (const uint32_t*)sc->d_wf_order.p;
	d.n_spaces = (uint32_t)h.spaces.size();
	d.n_surfaces = (uint32_t)h.surfaces.size();
	d.any_alpha = h.any_alpha ? 1u : 0u;
	d.n_models = (int32_t)h.models.size();
	d.n_nodes = (uint32_t)h.kd_nodes.size();
	d.n_refs = (uint32_t)h.kd_refs.size();
	d.n_tris = (uint32_t)h.tris.size();
	d.cam = h.camera;
	d.sun = h.sun;
	return PTX_OK;
}

void release_scene_buffers(ptx_scene* sc) {
	for (DevBuf* b : {&sc->d_models, &sc->d_surfaces, &sc->d_materials, &sc->d_nodes, &sc->d_refs, &sc->d_tris, &sc->d_shade, &sc->d_tex,
	                  &sc->d_texels, &sc->d_texels_f, &sc->d_lut, &sc->d_spaces, &sc->d_model_space, &sc->d_wf_order, &sc->d_res_nodes, &sc->d_res_refs, &sc->d_res_tris, &sc->d_hot})
		b->release();
}

int finish_scene(ptx_ctx* ctx, ptx_scene* sc, ptx_scene** out) {
	sc->ctx = ctx;
	if (ctx) {
		std::lock_guard<std::mutex> lk(ctx->mu);
		int rc = upload_scene(sc);
		if (rc != PTX_OK) { release_scene_buffers(sc); delete sc; return rc; }
		ctx->refs.fetch_add(1);
	} else {
		decide_mode(sc);
	}
	*out = sc;
	return PTX_OK;
}

}  // namespace

extern "C" {

const char* ptx_last_error(void) { return g_err.c_str(); }
const char* ptx_version(void) { return "ptx_hip 0.1 (gfx950)"; }

int ptx_ctx_create(int device, ptx_ctx** out) {
	if (!out) return set_err(PTX_ERR_INVALID, "ptx_ctx_create: out is NULL");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		(void)hipGetLastError();
		return set_err(PTX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
	}
	if (device < 0 || device >= n) return set_err(PTX_ERR_INVALID, "device ordinal out of range");
	HIP_TRY(hipSetDevice(device));
	ptx_ctx* c = new ptx_ctx;
	c->device = device;
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess) { delete c; return set_err(PTX_ERR_HIP, hipGetErrorString(e)); }
	c->n_cu = prop.multiProcessorCount;
	e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete c; return set_err(PTX_ERR_HIP, hipGetErrorString(e)); }
	*out = c;
	return PTX_OK;
}

static void ctx_release(ptx_ctx* c) {
	if (c->refs.fetch_sub(1) != 1) return;   // scenes still alive: the last one frees the context
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
	for (hipEvent_t ev : c->step_events) (void)hipEventDestroy(ev);
	c->queues.release(); c->spill.release(); c->sample_rad.release(); c->counters.release(); c->stage_a.release(); c->stage_b.release(); c->pixel_list.release(); c->srgb_thr.release();
	for (auto& w : c->wf) {
		for (DevBuf* b : {&w.qent, &w.pair_hit, &w.seg, &w.first, &w.mask, &w.ctl, &w.spill, &w.stream_buf, &w.flow}) b->release();
		if (w.flow_host) { (void)hipHostFree(w.flow_host); w.flow_host = nullptr; }
		if (w.done) { (void)hipEventDestroy(w.done); w.done = nullptr; }
		if (w.stream) { (void)hipStreamDestroy(w.stream); w.stream = nullptr; }
	}
	if (c->wf_main_ev) { (void)hipEventDestroy(c->wf_main_ev); c->wf_main_ev = nullptr; }
	(void)hipStreamDestroy(c->stream);
	delete c;
}
void ptx_ctx_destroy(ptx_ctx* c) {
	if (c) ctx_release(c);
}

void* ptx_ctx_stream(ptx_ctx* c) { return c ? (void*)c->stream : nullptr; }

int ptx_ctx_synchronize(ptx_ctx* c) {
	if (!c) return set_err(PTX_ERR_INVALID, "ctx is NULL");
	HIP_TRY(hipStreamSynchronize(c->stream));
	return PTX_OK;
}

int ptx_ctx_set_timing(ptx_ctx* c, int on) {
	if (!c) return set_err(PTX_ERR_INVALID, "ctx is NULL");
	std::lock_guard<std::mutex> lk(c->mu);
	c->timing_on = on != 0;
	return PTX_OK;
}

int ptx_ctx_get_timing(ptx_ctx* c, ptx_kernel_timing* out) {
	if (!c || !out) return set_err(PTX_ERR_INVALID, "NULL argument");
	std::lock_guard<std::mutex> lk(c->mu);
	*out = c->timing;
	return PTX_OK;
}

int ptx_scene_load_gltf(ptx_ctx* ctx, const char* path, const ptx_load_opts* opts, ptx_scene** out) {
	if (!path || !out) return set_err(PTX_ERR_INVALID, "ptx_scene_load_gltf: NULL argument");
	ptx_scene* sc = new ptx_scene;
	try {
		WorkFilter wf;
		if (opts && opts->filter_primitives) {
			wf.filter = true;
			for (uint32_t k = 0; k < opts->n_work; k++) {
				const ptx_work_item& it = opts->work[k];
				wf.work.emplace_back(it.mesh_name ? it.mesh_name : "", std::vector<int32_t>(it.primitives, it.primitives + it.n_primitives));
			}
		}
		load_gltf(path, opts ? opts->camera_index : 0u, opts ? opts->sun_light_index : 0u, wf, sc->host);
	} catch (const Error& e) {
		delete sc;
		return set_err(e.code, e.msg);
	} catch (const std::exception& e) {
		delete sc;
		return set_err(PTX_ERR_PARSE, e.what());
	}
	return finish_scene(ctx, sc, out);
}

int ptx_worker_event_load(ptx_ctx* ctx, const char* event_json_path, const char* local_scene_root, ptx_scene** scene, ptx_render_cfg* cfg,
                          ptx_worker_event* info) {
	if (!event_json_path || !local_scene_root || !scene || !cfg) return set_err(PTX_ERR_INVALID, "ptx_worker_event_load: NULL argument");
	ptx_scene* sc = new ptx_scene;
	try {
		WorkerEvent ev;
		parse_worker_event(event_json_path, ev);
		if (ev.samples < 0 || ev.bounces <= 0 || !(ev.X >= 1) || !(ev.Y >= 1)) throw Error{PTX_ERR_INVALID, "worker event: samples / bounces / X / Y out of range"};
		std::string root = local_scene_root;
		if (!root.empty() && root.back() != '/') root += '/';
		load_gltf(root + "scene.gltf", 0u, 0u, ev.work, sc->host);   // worker::download_gltf_file: scene_root + "scene.gltf" (worker.cpp:108-112)
		*cfg = ptx_render_cfg{};
		cfg->W = (uint32_t)ev.X; cfg->H = (uint32_t)ev.Y;            // worker.cpp:36-38
		cfg->spp = (uint32_t)ev.samples; cfg->bounces = (uint32_t)ev.bounces;
		cfg->env[0] = cfg->env[1] = cfg->env[2] = 1.0f;
		cfg->seed_lo = 0x5EEDu;
		cfg->integrator = PTX_INTEGRATOR_WORKER;                      // what processors::worker::run renders with
		if (info) {
			*info = ptx_worker_event{};
			info->num_workers = ev.num_workers;
			info->n_work_meshes = (uint32_t)ev.work.work.size();
			snprintf(info->worker_id, sizeof info->worker_id, "%s", ev.worker_id.c_str());
			snprintf(info->scene_root, sizeof info->scene_root, "%s", ev.scene_root.c_str());
			snprintf(info->scene_bucket, sizeof info->scene_bucket, "%s", ev.scene_bucket.c_str());
		}
	} catch (const Error& e) {
		delete sc;
		return set_err(e.code, e.msg);
	} catch (const std::exception& e) {
		delete sc;
		return set_err(PTX_ERR_PARSE, e.what());
	}
	return finish_scene(ctx, sc, scene);
}

int ptx_scene_set_environment(ptx_scene* sc, const char* png_path, int srgb) {
	if (!sc) return set_err(PTX_ERR_INVALID, "ptx_scene_set_environment: scene is NULL");
	// the file is read before the lock is taken; the scene's host arrays are only touched under it (a render may be in flight on
	// another thread: ptx_render holds the same mutex for its whole call)
	FlatScene img;   // the decoded file: one TexRec + its texels (8-bit or, for a Radiance .hdr, float)
	TexRec rec{};
	try {
		if (png_path) rec = load_texture(img, png_path, srgb != 0);   // PNG, JPEG or .hdr, by content
	} catch (const Error& e) {
		return set_err(e.code, e.msg);
	} catch (const std::exception& e) {
		return set_err(PTX_ERR_PARSE, e.what());
	}
	std::unique_lock<std::mutex> lk;
	if (sc->ctx) lk = std::unique_lock<std::mutex>(sc->ctx->mu);
	FlatScene& h = sc->host;
	if (h.env_tex >= 0) {   // the previous map is always the last texture (appended below): drop it instead of letting them pile up
		const TexRec old = h.textures[(size_t)h.env_tex];
		if (old.c_srgb & kTexFloat) h.texels_f.resize(old.offset); else h.texels.resize(old.offset);
		h.textures.pop_back();
		h.texture_paths.pop_back();
		h.env_tex = -1;
	}
	if (png_path) {
		if (rec.c_srgb & kTexFloat) { rec.offset = (uint32_t)h.texels_f.size(); h.texels_f.insert(h.texels_f.end(), img.texels_f.begin(), img.texels_f.end()); }
		else { rec.offset = (uint32_t)h.texels.size(); h.texels.insert(h.texels.end(), img.texels.begin(), img.texels.end()); }
		h.textures.push_back(rec);
		h.texture_paths.push_back(png_path);
		h.env_tex = (int32_t)h.textures.size() - 1;
	}
	if (sc->ctx) return upload_scene(sc);   // textures changed: the whole (small) scene goes up again
	return PTX_OK;
}

int ptx_scene_from_arrays(ptx_ctx* ctx, const ptx_scene_desc* d, ptx_scene** out) {
	if (!d || !out) return set_err(PTX_ERR_INVALID, "ptx_scene_from_arrays: NULL argument");
	if (!d->camera || (d->n_models && (!d->model_xform || !d->model_surf)) ||
	    (d->n_surfaces && (!d->surf_range || !d->vertices || !d->triangles || !d->materials)))
		return set_err(PTX_ERR_INVALID, "ptx_scene_from_arrays: missing array");
	ptx_scene* sc = new ptx_scene;
	FlatScene& h = sc->host;
	try {
		h.model_xform.assign(d->model_xform, d->model_xform + 12 * (size_t)d->n_models);
		h.model_surf.assign(d->model_surf, d->model_surf + 2 * (size_t)d->n_models);
		size_t nv = 0, nt = 0;
		for (uint32_t s = 0; s < d->n_surfaces; s++) {
			const int32_t* r = d->surf_range + 4 * (size_t)s;
			if (r[0] < 0 || r[1] < 0 || r[2] < 0 || r[3] < 0) throw Error{PTX_ERR_INVALID, "negative surface range"};
			int32_t rg[8] = {r[0], r[1], r[2], r[3], 0, 0, 0, 0};
			h.surf_range.insert(h.surf_range.end(), rg, rg + 8);
			nv = std::max(nv, (size_t)r[0] + r[1]);
			nt = std::max(nt, (size_t)r[2] + r[3]);
		}
		h.vertices.assign(d->vertices, d->vertices + 11 * nv);
		h.triangles.assign(d->triangles, d->triangles + 3 * nt);
		for (uint32_t s = 0; s < d->n_surfaces; s++) {
			const int32_t* r = &h.surf_range[8 * (size_t)s];
			for (int32_t t = 0; t < 3 * r[3]; t++)
				if (h.triangles[3 * (size_t)r[2] + t] >= (uint32_t)r[1]) throw Error{PTX_ERR_INVALID, "vertex index out of range"};
		}
		for (uint32_t m = 0; m < d->n_models; m++) {
			int32_t f = h.model_surf[2 * m], n = h.model_surf[2 * m + 1];
			if (f < 0 || n < 0 || (uint32_t)(f + n) > d->n_surfaces) throw Error{PTX_ERR_INVALID, "model surface range out of bounds"};
			if (f != (m ? h.model_surf[2 * m - 2] + h.model_surf[2 * m - 1] : 0)) throw Error{PTX_ERR_INVALID, "model surface ranges must be consecutive, in model order"};
			h.model_names.push_back("model" + std::to_string(m));
		}
		h.materials_raw.assign(d->materials, d->materials + 11 * (size_t)d->n_surfaces);
		h.material_tex.assign(7 * (size_t)d->n_surfaces, 0);
		finalize_scene(h, d->camera, d->sun);
	} catch (const Error& e) {
		delete sc;
		return set_err(e.code, e.msg);
	} catch (const std::exception& e) {
		delete sc;
		return set_err(PTX_ERR_INVALID, e.what());
	}
	return finish_scene(ctx, sc, out);
}

void ptx_scene_destroy(ptx_scene* sc) {
	if (!sc) return;
	if (sc->ctx) {
		std::lock_guard<std::mutex> lk(sc->ctx->mu);
		(void)hipSetDevice(sc->ctx->device);
		(void)hipStreamSynchronize(sc->ctx->stream);
		release_scene_buffers(sc);
	}
	ptx_ctx* c = sc->ctx;
	delete sc;
	if (c) ctx_release(c);
}

int ptx_scene_get_info(const ptx_scene* sc, ptx_scene_info* info) {
	if (!sc || !info) return set_err(PTX_ERR_INVALID, "NULL argument");
	const FlatScene& h = sc->host;
	info->n_models = (uint32_t)h.models.size();
	info->n_surfaces = (uint32_t)h.surfaces.size();
	info->n_vertices = (uint32_t)(h.vertices.size() / 11);
	info->n_triangles = (uint32_t)h.tris.size();
	info->n_kd_nodes = (uint32_t)h.kd_nodes.size();
	info->n_kd_refs = (uint32_t)h.kd_refs.size();
	info->kd_max_depth = h.kd_max_depth;
	info->has_sun = h.sun.present;
	info->geometry_bytes = (uint32_t)sc->host.geometry_bytes();
	info->lds_resident = (uint32_t)sc->mode;
	info->n_textures = (uint32_t)h.textures.size();
	return PTX_OK;
}

int64_t ptx_scene_get_array(const ptx_scene* sc, ptx_array which, void* dst, size_t dst_bytes) {
	if (!sc) { set_err(PTX_ERR_INVALID, "scene is NULL"); return -1; }
	const FlatScene& h = sc->host;
	const void* src = nullptr;
	size_t bytes = 0, elem = 4;
	std::vector<float> tmp;
	std::string names;
	switch (which) {
	case PTX_ARR_MODEL_XFORM: src = h.model_xform.data(); bytes = h.model_xform.size() * 4; break;
	case PTX_ARR_MODEL_AABB:
		for (auto& m : h.models) { tmp.insert(tmp.end(), m.bmin, m.bmin + 3); tmp.insert(tmp.end(), m.bmax, m.bmax + 3); }
		src = tmp.data(); bytes = tmp.size() * 4; break;
	case PTX_ARR_MODEL_SURF: src = h.model_surf.data(); bytes = h.model_surf.size() * 4; break;
	case PTX_ARR_SURF_RANGE: src = h.surf_range.data(); bytes = h.surf_range.size() * 4; break;
	case PTX_ARR_MESH_AABB:
		for (auto& s : h.surfaces) { tmp.insert(tmp.end(), s.bmin, s.bmin + 3); tmp.insert(tmp.end(), s.bmax, s.bmax + 3); }
		src = tmp.data(); bytes = tmp.size() * 4; break;
	case PTX_ARR_VERTICES: src = h.vertices.data(); bytes = h.vertices.size() * 4; break;
	case PTX_ARR_TRIANGLES: src = h.triangles.data(); bytes = h.triangles.size() * 4; break;
	case PTX_ARR_MATERIALS: src = h.materials_raw.data(); bytes = h.materials_raw.size() * 4; break;
	case PTX_ARR_KD_NODES: src = h.kd_nodes.data(); bytes = h.kd_nodes.size() * 8; break;
	case PTX_ARR_KD_REFS: src = h.kd_refs.data(); bytes = h.kd_refs.size() * 4; break;
	case PTX_ARR_CAMERA:
		tmp.assign(h.camera.origin, h.camera.origin + 3); tmp.insert(tmp.end(), h.camera.basis, h.camera.basis + 9);
		tmp.push_back(h.camera.fov); tmp.push_back(h.camera.tan_half_fov);
		src = tmp.data(); bytes = tmp.size() * 4; break;
	case PTX_ARR_SUN:
		if (h.sun.present) { tmp.assign(h.sun.basis, h.sun.basis + 9); tmp.insert(tmp.end(), h.sun.energy, h.sun.energy + 3); tmp.push_back(h.sun.angular_radius); }
		src = tmp.data(); bytes = tmp.size() * 4; break;
	case PTX_ARR_MODEL_NAMES:
		for (auto& n : h.model_names) names += n + "\n";
		src = names.data(); bytes = names.size(); elem = 1; break;
	case PTX_ARR_TEXTURES: src = h.textures.data(); bytes = h.textures.size() * sizeof(TexRec); break;
	case PTX_ARR_TEXELS: src = h.texels.data(); bytes = h.texels.size(); elem = 1; break;
	case PTX_ARR_SURF_TEX: src = h.surf_tex.data(); bytes = h.surf_tex.size() * 4; break;
	case PTX_ARR_TEXELS_F32: src = h.texels_f.data(); bytes = h.texels_f.size() * 4; break;
	default: set_err(PTX_ERR_INVALID, "unknown array id"); return -1;
	}
	if (dst) {
		if (dst_bytes < bytes) { set_err(PTX_ERR_INVALID, "destination too small"); return -1; }
		if (bytes) memcpy(dst, src, bytes);
	}
	return (int64_t)(bytes / elem);
}

namespace {

// Scenes the queue-based pipeline (wavefront.hip) takes: trees in global memory, a model of many surfaces (where the fused kernel's
// waves run nearly empty), at most 64 surfaces (one mask word per ray). PTX_WAVEFRONT=0/1 overrides the choice (measurement).
// Pair space is a pool sized from DEMAND: `wf_pairs_per_ray` of the scene (what its rays were seen to need; before the first
// measurement min(surfaces, 4)) plus a margin decides how many rays a pool of `pool_pairs` serves; a step that needs more raises the
// overflow word and the slab / slice is repeated in smaller pieces with the ratio it reported.
// pairs of a render's pool, 48 B each. Measured on the 24-surface atrium (3.7 pairs per ray, two rays per path and step; 1080p, 64 spp):
// 128 Mi pairs (6 GB: 17 M-path slabs, 10 GB of workspace in all) 409 Msamples/s, 256 Mi 435, 384 Mi 447, 512 Mi 452
// (profiles/round3_wf_ab.txt), and with the queues started largest tree first 384 / 512 Mi 473 / 477, 1024 Mi — the whole 133 M-path
// pass as ONE slab — 490-495 (profiles/round3_surface_order.txt): every step of a slab ends with a few waves finishing walks of
// hundreds of dependent fetches, and a larger slab has fewer such ends per path (jack-of-blades, whose steps after the first are small:
// 2300 -> 2850 Msamples/s from 66 M- to 133 M-path slabs). The default takes 1 Gi pairs (48 GB of a 288 GB device) unless that is more
// than a sixth of the free memory; what is ALLOCATED follows the scene's demand (ptx_render: alloc_pairs).
constexpr uint64_t kWfPoolPairs = 1024ull << 20;
constexpr uint64_t kWfBatchPairs = 256ull << 20;   // ... of a batch-intersect slice at most (12 GB); sized by the batch
constexpr uint32_t kWfFlowWords = 64, kWfFlowRays = 58 /* 64-bit */, kWfFlowPeak = 60, kWfFlowOverflow = 63, kWfMaxRound = 56;   // flow words: [s] entries of step s of the round, then the pool's peak demand and the overflow word
bool wf_eligible(const ptx_scene* sc) {
	const size_t n_surf = sc->host.surfaces.size();
	if (n_surf == 0 || n_surf > (size_t)kWfMaxSurfaces) return false;
	return sc->mode != MODE_LDS && sc->dev.tri_isect;   // LDS-resident scenes keep no global-memory copy of the traversal records
}
// Which pipeline renders a scene. The fused kernel is at its best when the geometry rays meet is in LDS; the queues, when it is in
// global memory: lanes are compacted per (ray, surface) pair and more waves cover the fetch latency. Two static signs of the latter:
// a model of eight or more surfaces (1.9 x on the 24-surface atrium), or little of the surfaces' box area being LDS-resident — the
// share is 0.98 for Cornell + 82 k-triangle mesh and 0.86 for the plaza (walls / ground resident: the queues run them 0.57 x / 0.73 x),
// 0.16 for the reference's jack-of-blades (1.08 x through the queues) and 0.08 for the atrium. PTX_WAVEFRONT=0/1 overrides
// (measurement, tests); the fused kernel's own measurement switches keep it selected.
int pipeline_choice(const ptx_scene* sc) {
	if (!wf_eligible(sc)) return 0;
	if (const char* e = getenv("PTX_WAVEFRONT")) return e[0] == '1' ? 1 : 0;
	if (getenv("PTX_FORCE_GLOBAL") || getenv("PTX_NO_HYBRID")) return 0;
	int32_t max_per_model = 0;
	for (const ModelRec& mr : sc->host.models) max_per_model = std::max(max_per_model, mr.n_surfaces);
	return (max_per_model >= 8 || sc->lds_area_share < 0.35) ? 1 : 0;
}
bool use_wavefront(const ptx_scene* sc) { return pipeline_choice(sc) == 1; }
double wf_ratio_guess(const ptx_scene* sc) {
	const double n_surf = (double)sc->host.surfaces.size();
	if (sc->wf_pairs_per_ray > 0) return std::min(n_surf, sc->wf_pairs_per_ray * 1.15 + 0.05);
	if (const char* e = getenv("PTX_WF_RATIO_GUESS")) return std::max(0.01, atof(e));   // tests: a guess that is too low exercises the overflow path
	return std::min(n_surf, 4.0);
}
// buffers of one workspace set for `rays` rays per step, a pool of `pool` pairs and `steps` control blocks
hipError_t wf_workspace(ptx_ctx* c, int set, size_t rays, size_t pool, size_t n_surf, size_t steps, WfBuffers& W) {
	ptx_ctx::WfSet& w = c->wf[set];
	const size_t tiles = (rays + kWfTile - 1) / kWfTile;
	hipError_t e;
	if ((e = w.qent.ensure(pool * 32)) != hipSuccess) return e;
	if ((e = w.pair_hit.ensure(pool * 16)) != hipSuccess) return e;
	if ((e = w.seg.ensure(n_surf * tiles * sizeof(uint2))) != hipSuccess) return e;
	if ((e = w.first.ensure(rays * 4)) != hipSuccess) return e;
	if ((e = w.mask.ensure(rays * 8)) != hipSuccess) return e;
	if ((e = w.ctl.ensure(steps * kWfCtlWords * 4)) != hipSuccess) return e;
	if ((e = w.flow.ensure(kWfFlowWords * 4)) != hipSuccess) return e;
	if (!w.flow_host && (e = hipHostMalloc((void**)&w.flow_host, kWfFlowWords * 4)) != hipSuccess) return e;
	if ((e = w.spill.ensure((size_t)wf_traverse_grid(c->n_cu) * 4 * (size_t)kSpillWords * sizeof(uint4))) != hipSuccess) return e;   // 16-byte entries: node content + entry distance
	W.qent = (float4*)w.qent.p; W.pair_hit = (float4*)w.pair_hit.p;
	W.pool_cap = (uint32_t)std::min<size_t>(pool, 0xFFFFFFFFu);
	W.seg = (uint2*)w.seg.p; W.seg_cap = (uint32_t)tiles;
	W.first = (uint32_t*)w.first.p; W.mask = (unsigned long long*)w.mask.p; W.ctl = (uint32_t*)w.ctl.p; W.spill = (uint2*)w.spill.p;
	W.n_in = nullptr;
	W.overflow = (uint32_t*)w.flow.p + kWfFlowOverflow;
	W.peak = (uint32_t*)w.flow.p + kWfFlowPeak;
	W.ray_counter = nullptr;
	W.wave_clock = 0;
	return hipSuccess;
}
size_t wf_workspace_bytes(const ptx_ctx* c) {
	size_t b = 0;
	for (const auto& w : c->wf)
		for (const DevBuf* d : {&w.qent, &w.pair_hit, &w.seg, &w.first, &w.mask, &w.ctl, &w.spill, &w.stream_buf, &w.flow}) b += d->cap;
	return b;
}

}  // namespace

int ptx_render(ptx_scene* sc, const ptx_render_cfg* cfg, float* accum, ptx_render_stats* stats) {
	if (!sc || !cfg || !accum) return set_err(PTX_ERR_INVALID, "ptx_render: NULL argument");
	if (!sc->ctx) return set_err(PTX_ERR_NO_DEVICE, "ptx_render: scene was created without a GPU context (no CPU path exists)");
	if (!cfg->W || !cfg->H) return set_err(PTX_ERR_INVALID, "ptx_render: W and H must be > 0");   // bounces = 0 is legal: a black frame (renderer.cpp:438-439)
	uint32_t x0 = cfg->x0, y0 = cfg->y0, w = cfg->w, h = cfg->h;
	if (w == 0 && h == 0) { x0 = 0; y0 = 0; w = cfg->W; h = cfg->H; }
	if (!w || !h || (uint64_t)x0 + w > cfg->W || (uint64_t)y0 + h > cfg->H) return set_err(PTX_ERR_INVALID, "ptx_render: tile outside the image");
	if (cfg->bounces > 0xFFFFu) return set_err(PTX_ERR_INVALID, "ptx_render: bounces > 65535");
	if (cfg->integrator > PTX_INTEGRATOR_WORKER) return set_err(PTX_ERR_INVALID, "ptx_render: unknown integrator");
	const bool sharded = cfg->shard_count > 1;
	if (sharded && cfg->shard_index >= cfg->shard_count) return set_err(PTX_ERR_INVALID, "ptx_render: shard_index >= shard_count");
	ptx_ctx* c = sc->ctx;
	std::lock_guard<std::mutex> lk(c->mu);
	HIP_TRY(hipSetDevice(c->device));
	const uint64_t rect_pixels = (uint64_t)w * h;
	if (rect_pixels > 0x7FFFFFFFull) return set_err(PTX_ERR_INVALID, "ptx_render: tile too large");
	if (stats) *stats = ptx_render_stats{};
	if (cfg->spp == 0) return PTX_OK;

	// The order in which a pass enumerates its pixels (path id -> pixel). Per-sample radiance is keyed by (pixel, sample), so the order
	// changes no result — only which rays sit next to each other in a wave and in a classify tile. "tiled": 8 x 8 pixel blocks (one
	// wave of camera rays) inside 32 x 32 blocks (one classify tile) — coherent rays enter the same surfaces and walk the same nodes;
	// PTX_PIXEL_ORDER=linear|tiled overrides (measurement). Interleaved tile sharding: only the pixels of this shard's image tiles.
	uint64_t n_pixels = rect_pixels;
	const uint32_t* d_pixels = nullptr;
	bool tiled = use_wavefront(sc);   // measured: profiles/round3_pixel_order.txt
	if (const char* e = getenv("PTX_PIXEL_ORDER")) tiled = e[0] == 't';
	if (sharded || tiled) {
		const uint32_t ts = sharded ? (cfg->shard_tile ? cfg->shard_tile : 64u) : 0u;
		const uint32_t key[9] = {cfg->W, cfg->H, x0, y0, w, h, sharded ? cfg->shard_index : 0u, (sharded ? cfg->shard_count : 1u) | (tiled ? 0x80000000u : 0u), ts};
		if (memcmp(key, c->list_key, sizeof key) != 0 || !c->pixel_list.p) {
			std::vector<uint32_t> list;
			// the pixels of the image rectangle [xa, xb) x [ya, yb): rows, or 8 x 8 blocks inside 32 x 32 blocks anchored at the image origin
			auto add_rect = [&](uint32_t xa, uint32_t ya, uint32_t xb, uint32_t yb) {
				if (!tiled) {
					for (uint32_t y = ya; y < yb; y++)
						for (uint32_t x = xa; x < xb; x++) list.push_back((y - y0) * w + (x - x0));
					return;
				}
				for (uint32_t by = ya / 32; by <= (yb - 1) / 32; by++)
					for (uint32_t bx = xa / 32; bx <= (xb - 1) / 32; bx++)
						for (uint32_t sy = 0; sy < 4; sy++)
							for (uint32_t sx = 0; sx < 4; sx++)
								for (uint32_t y = by * 32 + sy * 8; y < by * 32 + sy * 8 + 8; y++)
									for (uint32_t x = bx * 32 + sx * 8; x < bx * 32 + sx * 8 + 8; x++)
										if (x >= xa && x < xb && y >= ya && y < yb) list.push_back((y - y0) * w + (x - x0));
			};
			if (!sharded) add_rect(x0, y0, x0 + w, y0 + h);
			else {
				const uint32_t tiles_x = (cfg->W + ts - 1) / ts;
				for (uint32_t ty = y0 / ts; ty <= (y0 + h - 1) / ts; ty++)
					for (uint32_t tx = x0 / ts; tx <= (x0 + w - 1) / ts; tx++) {
						if ((uint64_t)(ty * (uint64_t)tiles_x + tx) % cfg->shard_count != cfg->shard_index) continue;
						add_rect(std::max(tx * ts, x0), std::max(ty * ts, y0), std::min((tx + 1) * ts, x0 + w), std::min((ty + 1) * ts, y0 + h));
					}
			}
			// a previous render with stats == NULL and a device buffer returns without a sync (ptx.h): its generate / resolve kernels may
			// still be reading the list this call is about to replace, and the context's stream is non-blocking (not ordered with the
			// NULL stream a plain hipMemcpy would use) — drain it first, then upload on the same stream
			HIP_TRY(hipStreamSynchronize(c->stream));
			HIP_TRY(c->pixel_list.ensure(std::max<size_t>(list.size() * 4, 16)));
			if (!list.empty()) {
				HIP_TRY(hipMemcpyAsync(c->pixel_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, c->stream));
				HIP_TRY(hipStreamSynchronize(c->stream));   // `list` is a local
			}
			memcpy(c->list_key, key, sizeof key);
			c->list_len = (uint32_t)list.size();
		}
		n_pixels = c->list_len;
		d_pixels = (const uint32_t*)c->pixel_list.p;
		if (n_pixels == 0) return PTX_OK;   // no tile of this shard meets the rectangle
	}

	// samples of every pixel per launch: enough paths to fill the chip many times over, bounded workspace
	uint32_t pass_spp = cfg->spp_per_pass;
	if (pass_spp == 0) {
		const uint64_t target_paths = 128ull << 20;   // 64 spp of a 1080p frame: 2 GB of per-sample radiance; fewer, longer launches (measured: 8 -> 64 spp per launch = +11 %)
		pass_spp = (uint32_t)std::max<uint64_t>(1, target_paths / n_pixels);
	}
	pass_spp = std::min(pass_spp, cfg->spp);
	while ((uint64_t)pass_spp * n_pixels > 0xFFFFFFFFull) pass_spp--;  // path ids are 32-bit
	if (pass_spp == 0) return set_err(PTX_ERR_INVALID, "ptx_render: tile too large for one pass");

	const int grid = c->n_cu;
	const size_t n_slots = (size_t)grid * (kBlock / 64);
	// Units the kernels may set aside: whole models, or — when some model has many surfaces (a Sponza-class mesh) — single
	// surfaces. PTX_SURFACE_UNITS=0/1 overrides the choice (measurement).
	const uint32_t n_surf = (uint32_t)sc->host.surfaces.size(), n_mod = (uint32_t)sc->host.models.size();
	int32_t max_per_model = 0;
	for (const ModelRec& mr : sc->host.models) max_per_model = std::max(max_per_model, mr.n_surfaces);
	bool surface_units = max_per_model >= 8 && n_surf <= (uint32_t)kMaxDeferModels;   // measured: +49 % on a 24-surface model, -2..-7 % on scenes of 1-3 surfaces per model
	if (const char* e = getenv("PTX_SURFACE_UNITS")) surface_units = e[0] == '1' && n_surf <= (uint32_t)kMaxDeferModels;
	const uint32_t queue_stride = queue_float4_per_wave(surface_units ? n_surf : n_mod);
	bool wavefront = use_wavefront(sc);
	HIP_TRY(c->sample_rad.ensure((size_t)pass_spp * n_pixels * sizeof(float4)));
	HIP_TRY(c->counters.ensure(1024));   // [0] chunk counter, [16] ray counter, [64..] PTX_PROF region counters
	HIP_TRY(c->spill.ensure(n_slots * (size_t)kSpillWords * sizeof(uint2)));
	unsigned long long* chunk_counter = (unsigned long long*)c->counters.p;
	unsigned long long* ray_counter = (unsigned long long*)((char*)c->counters.p + 16);
	HIP_TRY(hipMemsetAsync(c->counters.p, 0, 1024, c->stream));

	const bool dev_accum = is_device_ptr(accum);
	float4* d_accum = (float4*)accum;
	if (!dev_accum) {
		HIP_TRY(c->stage_a.ensure(rect_pixels * sizeof(float4)));
		d_accum = (float4*)c->stage_a.p;
		HIP_TRY(hipMemcpyAsync(d_accum, accum, rect_pixels * sizeof(float4), hipMemcpyHostToDevice, c->stream));
	}

	const uint32_t n_pass = (cfg->spp + pass_spp - 1) / pass_spp;
	if (stats)
		while (c->events.size() < 2 * (size_t)n_pass) {
			hipEvent_t ev;
			HIP_TRY(hipEventCreate(&ev));
			c->events.push_back(ev);
		}
	PassBuffers B{nullptr /* the fused kernel's streams: set below, once it is known which pipeline runs */, queue_stride, surface_units ? 1u : 0u, (float4*)c->sample_rad.p, (uint2*)c->spill.p, chunk_counter, ray_counter};
	// queue-based pipeline (wavefront.hip): a pass runs in slabs; slab size from the pair pool and the pairs a path of this scene was
	// seen to need — a step classifies two rays per path (extend + shadow)
	WfBuffers WF[2]{};
	WfStream wf_st[2][2]{};
	uint32_t wf_cap = 0;
	int wf_sets = 1;
	uint64_t pool_pairs = kWfPoolPairs;
	const uint64_t pass_paths = (uint64_t)pass_spp * n_pixels;
	const double ratio_at_entry = sc->wf_pairs_per_ray;
	// steps enqueued back to back before the host looks at the flow words again. The grids of a round are sized for the entries the
	// slab had when the round began (entries only ever get fewer): short rounds keep the later steps' grids close to what is alive —
	// the shade kernel's workgroups beyond the entry count only read it and leave, but a 66 M-path slab has 259 K of them per launch —
	// at the price of one host round trip (tens of microseconds) per round. PTX_WF_ROUND overrides (measurement).
	uint32_t wf_round = 3;
	if (const char* e = getenv("PTX_WF_ROUND")) wf_round = (uint32_t)std::max(1, atoi(e));
	wf_round = (uint32_t)std::min<uint64_t>({(uint64_t)wf_round, (uint64_t)cfg->bounces + 1u, (uint64_t)kWfMaxRound});
	const bool timing = stats && c->timing_on;
	double clock_busy = 0, clock_all = 0;   // queue-based pipeline with timing on: traverse waves' run times against waves x longest run, summed over launches
	if (stats) c->timing = ptx_kernel_timing{};
	auto slab_cap = [&]() -> uint32_t {   // paths of a slab: the pool must hold the pairs of its busiest step
		const double per_path = 2.0 * wf_ratio_guess(sc);
		const uint64_t by_pool = (uint64_t)std::max(65536.0, (double)pool_pairs / per_path);
		static const uint64_t max_slab = [] { const char* e = getenv("PTX_WF_MAX_SLAB_M"); return e ? std::min<uint64_t>((uint64_t)kWfIdMask, strtoull(e, nullptr, 10) << 20) : (uint64_t)kWfMaxSlab - 1; }();   // measurement
		const uint64_t cap = std::min<uint64_t>({(pass_paths + wf_sets - 1) / wf_sets, max_slab, by_pool});
		const uint64_t n_slabs = (pass_paths + cap - 1) / cap;   // slabs of equal size rather than full ones and a remainder
		return (uint32_t)((pass_paths + n_slabs - 1) / n_slabs);
	};
	auto allocate = [&]() -> hipError_t {
		hipError_t e;
		// the pool that is allocated: what the slab needs at the pairs per path this scene is expected to ask for (+ 10 %), not the
		// whole budget — a scene whose rays enter few boxes (jack-of-blades: 0.3 pairs per ray) holds 2 GB of pairs, not 18
		// (in steps of 64 Mi pairs: the ratio learnt from one frame must not move the allocation by a few per cent in the next)
		const uint64_t want_pairs = (uint64_t)((double)wf_cap * 2.0 * wf_ratio_guess(sc) * 1.1), step = want_pairs > (128ull << 20) ? (64ull << 20) : (16ull << 20);
		const uint64_t alloc_pairs = std::min<uint64_t>(pool_pairs, std::max<uint64_t>(16ull << 20, (want_pairs + step - 1) / step * step));
		for (int k = 0; k < wf_sets; k++) {
			ptx_ctx::WfSet& w = c->wf[k];
			if (w.qent.cap > 4 * alloc_pairs * 32) { w.qent.release(); w.pair_hit.release(); }   // held from a much hungrier scene: give it back
			if ((e = wf_workspace(c, k, 2 * (size_t)wf_cap, alloc_pairs, n_surf, wf_round, WF[k])) != hipSuccess) return e;
			WF[k].ray_counter = (unsigned long long*)((uint32_t*)w.flow.p + kWfFlowRays);   // rays of the slab: added to the total once the slab is through (an overflowing attempt is not counted)
			if ((e = w.stream_buf.ensure((size_t)wf_cap * 14 * sizeof(float4))) != hipSuccess) return e;
			if (!w.stream && (e = hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking)) != hipSuccess) return e;
			if (!w.done && (e = hipEventCreateWithFlags(&w.done, hipEventDisableTiming)) != hipSuccess) return e;
		}
		return hipSuccess;
	};
	if (wavefront) {
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
				const uint64_t held = wf_workspace_bytes(c);   // what the context already holds counts as available: the same answer frame after frame
				pool_pairs = std::min<uint64_t>(pool_pairs, std::max<uint64_t>(16ull << 20, (free_b + held) / 4 / 48));
			} else (void)hipGetLastError();
		}
		if (const char* e = getenv("PTX_WF_PAIRS_M")) pool_pairs = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;   // measurement: pool size in Mi pairs
		pool_pairs = std::min<uint64_t>(pool_pairs, 0xFFFFFFFFull);
		wf_sets = getenv("PTX_WF_TWO_STREAMS") ? 2 : 1;   // measurement: two slabs side by side on two streams
		if (!c->wf_main_ev) HIP_TRY(hipEventCreateWithFlags(&c->wf_main_ev, hipEventDisableTiming));
		// when the device cannot spare the pool: a smaller one (smaller slabs), and below 8 Mi pairs the fused kernel
		for (;;) {
			wf_cap = slab_cap();
			const hipError_t e = allocate();
			if (e == hipSuccess) break;
			if (e != hipErrorOutOfMemory) return set_err(PTX_ERR_HIP, std::string("queue-based pipeline workspace: ") + hipGetErrorString(e));
			(void)hipGetLastError();
			for (auto& w : c->wf)
				for (DevBuf* b : {&w.qent, &w.pair_hit, &w.seg, &w.first, &w.mask, &w.stream_buf}) b->release();
			pool_pairs /= 2;
			if (pool_pairs < (8ull << 20)) { wavefront = false; break; }
		}
	}
	auto set_streams = [&](uint32_t cap) {   // the two stream buffers of each set, `cap` entries per array
		for (int k = 0; k < wf_sets; k++) {
			float4* base = (float4*)c->wf[k].stream_buf.p;
			wf_st[k][0] = WfStream{base, base + 8 * (size_t)cap};
			wf_st[k][1] = WfStream{base + 4 * (size_t)cap, base + 11 * (size_t)cap};
		}
	};
	if (!wavefront) {
		HIP_TRY(c->queues.ensure(n_slots * (size_t)queue_stride * sizeof(float4)));
		B.queues = (float4*)c->queues.p;
	}
	size_t n_step_ev = 0;   // step events used so far (timing)
	unsigned long long wf_rays = 0;   // rays the queue-based pipeline traced (slabs that went through)
	auto step_events = [&]() -> hipEvent_t* {
		if (!timing) return nullptr;
		while (c->step_events.size() < n_step_ev + 4) {
			hipEvent_t ev;
			if (hipEventCreate(&ev) != hipSuccess) return nullptr;
			c->step_events.push_back(ev);
		}
		n_step_ev += 4;
		return &c->step_events[n_step_ev - 4];
	};
	for (uint32_t p = 0; p < n_pass; p++) {
		RenderParams P{};
		P.W = cfg->W; P.H = cfg->H; P.x0 = x0; P.y0 = y0; P.w = w; P.h = h;
		P.n_pixels = (uint32_t)n_pixels;
		P.sample0 = cfg->sample0 + p * pass_spp;
		P.pass_spp = std::min(pass_spp, cfg->spp - p * pass_spp);
		P.bounces = cfg->bounces;
		P.n_paths = (uint64_t)P.pass_spp * n_pixels;
		P.seed_lo = cfg->seed_lo; P.seed_hi = cfg->seed_hi;
		memcpy(P.env, cfg->env, sizeof P.env);
		P.integrator = cfg->integrator;
		P.pixels = d_pixels;
		HIP_TRY(hipMemsetAsync(chunk_counter, 0, 8, c->stream));
		if (stats) HIP_TRY(hipEventRecord(c->events[2 * p], c->stream));
		if (wavefront) {
			// queue-based pipeline: the pass in slabs of at most `wf_cap` paths (one per workspace set and stream at a time). The steps of a
			// slab are enqueued back to back, `wf_round` at a time: every kernel takes its entry count from the device (flow words), and the
			// host reads them back once per round — whether paths are left (pass-through materials can outlive bounces + 1 steps), whether
			// some step's pairs overflowed the pool, and the peak demand that sizes the next slab.
			HIP_TRY(hipEventRecord(c->wf_main_ev, c->stream));
			for (int k = 0; k < wf_sets; k++) HIP_TRY(hipStreamWaitEvent(c->wf[k].stream, c->wf_main_ev, 0));
			uint64_t first = 0;
			while (first < P.n_paths) {
				wf_cap = std::min(wf_cap, slab_cap());   // never above what the buffers were sized for
				set_streams(wf_cap);
				uint32_t n_slab[2] = {0, 0}, slab_first[2] = {0, 0}, n_round[2] = {0, 0};   // paths of the slab, entries when the current round began
				bool live[2] = {false, false};
				int cur[2] = {0, 0};
				for (int k = 0; k < wf_sets; k++) {
					const uint64_t f = first + (uint64_t)k * wf_cap;
					slab_first[k] = (uint32_t)f;
					n_slab[k] = f < P.n_paths ? (uint32_t)std::min<uint64_t>(wf_cap, P.n_paths - f) : 0u;
					if (!n_slab[k]) continue;
					ptx_ctx::WfSet& ws = c->wf[k];
					HIP_TRY(launch_wf_generate(sc->dev, P, wf_st[k][0], wf_cap, slab_first[k], n_slab[k], B.sample_rad, ws.stream));
					HIP_TRY(hipMemsetAsync(ws.flow.p, 0, kWfFlowWords * 4, ws.stream));
					HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ws.flow.p, (int)n_slab[k], 1, ws.stream));   // flow[0] = entries of step 0
					live[k] = P.bounces > 0;
					n_round[k] = n_slab[k];
				}
				bool overflow = false;
				uint64_t peak = 0;
				while (live[0] || live[1]) {
					for (int k = 0; k < wf_sets; k++) {
						if (!live[k]) continue;
						ptx_ctx::WfSet& ws = c->wf[k];
						uint32_t* flow = (uint32_t*)ws.flow.p;
						HIP_TRY(hipMemsetAsync(ws.ctl.p, 0, (size_t)wf_round * kWfCtlWords * 4, ws.stream));
						HIP_TRY(hipMemsetAsync(flow + 1, 0, (size_t)wf_round * 4, ws.stream));
						for (uint32_t st = 0; st < wf_round; st++) {
							WfBuffers W = WF[k];
							W.ctl = (uint32_t*)ws.ctl.p + (size_t)st * kWfCtlWords;
							W.n_in = flow + st;
							W.wave_clock = timing ? 1u : 0u;
							HIP_TRY(launch_wf_step(sc->dev, P, W, wf_st[k][cur[k]], wf_st[k][cur[k] ^ 1], wf_cap, n_round[k], slab_first[k], flow + st + 1, B.sample_rad, c->n_cu, ws.stream,
							                       step_events()));
							cur[k] ^= 1;
						}
						HIP_TRY(hipMemcpyAsync(ws.flow_host, flow, kWfFlowWords * 4, hipMemcpyDeviceToHost, ws.stream));
					}
					for (int k = 0; k < wf_sets; k++) {
						if (!live[k]) continue;
						ptx_ctx::WfSet& ws = c->wf[k];
						HIP_TRY(hipStreamSynchronize(ws.stream));
						peak = std::max<uint64_t>(peak, ws.flow_host[kWfFlowPeak]);
						if (timing) {   // the traverse waves' own clocks of this round's steps (wavefront.hip: kWfCtlClock)
							for (uint32_t st = 0; st < wf_round; st++) {
								uint32_t ck[4];
								HIP_TRY(hipMemcpy(ck, (uint32_t*)ws.ctl.p + (size_t)st * kWfCtlWords + kWfCtlClock, sizeof ck, hipMemcpyDeviceToHost));
								const double sum = (double)(((uint64_t)ck[1] << 32) | ck[0]), all = (double)ck[2] * (double)ck[3];
								clock_busy += sum; clock_all += all;
							}
						}
						if (ws.flow_host[kWfFlowOverflow]) { overflow = true; live[k] = false; continue; }
						const uint32_t left = ws.flow_host[wf_round];
						n_round[k] = left;
						if (left == 0) live[k] = false;
						else HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ws.flow.p, (int)left, 1, ws.stream));   // next round: flow[0] = what this one left
					}
					if (overflow) break;
				}
				// what a ray of this scene needs, from the busiest step of these slabs (two rays per path and step)
				const uint32_t n_big = std::max(n_slab[0], n_slab[1]);
				if (peak && n_big) sc->wf_pairs_per_ray = std::max(sc->wf_pairs_per_ray, (double)peak / (2.0 * n_big));
				if (stats) { c->timing.peak_pairs = std::max<uint64_t>(c->timing.peak_pairs, peak); c->timing.slab_paths = std::max<uint64_t>(c->timing.slab_paths, n_big); }
				if (overflow) {
					if (stats) c->timing.pool_overflows++;
					// some step needed more pairs than the pool holds: the same slabs again, smaller (the ratio just learnt says how much). The
					// samples the aborted attempt already stored are stored again with the same values.
					for (int k = 0; k < wf_sets; k++) HIP_TRY(hipStreamSynchronize(c->wf[k].stream));
					const uint32_t smaller = std::min<uint32_t>(slab_cap(), wf_cap - wf_cap / 4);
					if (wf_cap <= 4096) return set_err(PTX_ERR_HIP, "queue-based pipeline: the pair pool cannot hold one step of a 4096-path slab");
					wf_cap = std::max<uint32_t>(4096, smaller);
					continue;
				}
				for (int k = 0; k < wf_sets; k++)
					if (n_slab[k] && P.bounces > 0) { unsigned long long r; memcpy(&r, c->wf[k].flow_host + kWfFlowRays, 8); wf_rays += r; }
				first += (uint64_t)wf_sets * wf_cap;
			}
			for (int k = 0; k < wf_sets; k++) {
				HIP_TRY(hipEventRecord(c->wf[k].done, c->wf[k].stream));
				HIP_TRY(hipStreamWaitEvent(c->stream, c->wf[k].done, 0));
			}
			if (ratio_at_entry == 0 && sc->wf_pairs_per_ray > 0 && p + 1 == n_pass) {
				// the scene's first frame has just told what its rays need: bring the workspace to the size the NEXT frame of this kind
				// will ask for now (a larger slab, a smaller or larger pool), inside the frame that pays for allocations anyway
				for (int k = 0; k < wf_sets; k++) HIP_TRY(hipStreamSynchronize(c->wf[k].stream));
				wf_cap = slab_cap();
				const hipError_t se = allocate();
				if (se != hipSuccess) (void)hipGetLastError();   // not fatal: the next frame sizes its workspace itself
			}
		} else {
			if (!B.queues || !B.sample_rad || !B.spill) return set_err(PTX_ERR_HIP, "ptx_render: workspace of the fused kernel is not allocated");
			HIP_TRY(launch_render_pass(sc->dev, P, B, sc->mode, sc->lds_bytes, grid, c->stream));
		}
		if (stats) HIP_TRY(hipEventRecord(c->events[2 * p + 1], c->stream));
		HIP_TRY(launch_resolve(B.sample_rad, d_accum, d_pixels, P.n_pixels, P.pass_spp, c->stream));
	}
	if (!dev_accum) HIP_TRY(hipMemcpyAsync(accum, d_accum, rect_pixels * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
	if (stats || !dev_accum) HIP_TRY(hipStreamSynchronize(c->stream));
	if (stats) {
		unsigned long long rays = 0;
		HIP_TRY(hipMemcpy(&rays, ray_counter, 8, hipMemcpyDeviceToHost));
		stats->rays = rays + wf_rays;
		stats->samples = (uint64_t)cfg->spp * n_pixels;
		stats->passes = n_pass;
		double ms = 0;
		for (uint32_t p = 0; p < n_pass; p++) {
			float t = 0;
			HIP_TRY(hipEventElapsedTime(&t, c->events[2 * p], c->events[2 * p + 1]));
			ms += t;
		}
		stats->kernel_ms = ms;
		ptx_kernel_timing& tm = c->timing;
		tm.pipeline = wavefront ? 1u : 0u;
		tm.pool_pairs = wavefront ? WF[0].pool_cap : 0;
		tm.workspace_bytes = wavefront ? wf_workspace_bytes(c) : c->queues.cap + c->spill.cap;
		if (!wavefront) { tm.fused_ms = ms; tm.fused_launches = n_pass; }
		for (size_t k = 0; k + 4 <= n_step_ev; k += 4) {
			float t0 = 0, t1 = 0, t2 = 0;
			HIP_TRY(hipEventElapsedTime(&t0, c->step_events[k], c->step_events[k + 1]));
			HIP_TRY(hipEventElapsedTime(&t1, c->step_events[k + 1], c->step_events[k + 2]));
			HIP_TRY(hipEventElapsedTime(&t2, c->step_events[k + 2], c->step_events[k + 3]));
			tm.classify_ms += t0; tm.traverse_ms += t1; tm.shade_ms += t2;
			tm.steps++;
		}
		tm.traverse_drain_frac = clock_all > 0 ? 1.0 - clock_busy / clock_all : 0.0;
#ifdef PTX_CLK
		if (!wavefront) {
			unsigned long long clk[8];
			HIP_TRY(hipMemcpy(clk, (char*)c->counters.p + 64, sizeof clk, hipMemcpyDeviceToHost));
			static const char* names[8] = {"kernel", "chunk_fetch", "extend_entry_loads", "extend_sweep_rest", "set_aside_lists", "shade_entry_hit_loads", "shade_hitrec_gathers", "shade_rest"};
			for (int k = 0; k < 8; k++) fprintf(stderr, "CLK %-22s %12llu kcycles summed over waves (%.2f %% of the waves' time)\n", names[k], clk[k], 100.0 * clk[k] / (double)clk[0]);
		}
#endif
#ifdef PTX_PROF
		unsigned long long prof[2 * kProfRegions];
		HIP_TRY(hipMemcpy(prof, (char*)c->counters.p + 64, sizeof prof, hipMemcpyDeviceToHost));
		static const char* names[kProfRegions] = {"extend_iter", "model_iter", "space_xform", "inline_model", "mesh_call", "mesh_pop", "node_step",
		                                           "tri_test", "defer_iter", "shade_iter", "defer_mesh_call", "defer_mesh_pop", "defer_node_step", "defer_tri_test",
		                                           "list_append", "shade_hit"};
		for (int k = 0; k < kProfRegions; k++)
			fprintf(stderr, "PROF %-16s trips %12llu lanes %14llu  util %.3f  trips/64rays %.3f\n", names[k], prof[2 * k], prof[2 * k + 1],
			        prof[2 * k] ? (double)prof[2 * k + 1] / (64.0 * prof[2 * k]) : 0.0, (double)prof[2 * k] / ((double)rays / 64.0));
#endif
	}
	return PTX_OK;
}

int ptx_intersect_batch(ptx_scene* sc, const ptx_rays* r, size_t n, const ptx_hits* hh) {
	if (!sc || !r || !hh) return set_err(PTX_ERR_INVALID, "ptx_intersect_batch: NULL argument");
	if (!sc->ctx) return set_err(PTX_ERR_NO_DEVICE, "ptx_intersect_batch: scene was created without a GPU context (no CPU path exists)");
	if (n == 0) return PTX_OK;
	if (!r->ox || !r->oy || !r->oz || !r->dx || !r->dy || !r->dz || !hh->distance || !hh->surface || !hh->triangle || !hh->b0 || !hh->b1 || !hh->b2)
		return set_err(PTX_ERR_INVALID, "ptx_intersect_batch: required array is NULL");
	auto group_ok = [](const void* a, const void* b, const void* c) { return (!a && !b && !c) || (a && b && c); };
	if (!group_ok(hh->px, hh->py, hh->pz) || !group_ok(hh->nx, hh->ny, hh->nz) || ((hh->u != nullptr) != (hh->v != nullptr)))
		return set_err(PTX_ERR_INVALID, "ptx_intersect_batch: optional outputs must be given as whole groups");
	ptx_ctx* c = sc->ctx;
	std::lock_guard<std::mutex> lk(c->mu);
	HIP_TRY(hipSetDevice(c->device));
	const bool dev = is_device_ptr(r->ox);
	IntersectArgs A{};
	A.n = n;
	const int n_out = 6 + (hh->px ? 3 : 0) + (hh->nx ? 3 : 0) + (hh->u ? 2 : 0);
	if (dev) {
		A.ox = r->ox; A.oy = r->oy; A.oz = r->oz; A.dx = r->dx; A.dy = r->dy; A.dz = r->dz;
		A.distance = hh->distance; A.surface = hh->surface; A.triangle = hh->triangle;
		A.b0 = hh->b0; A.b1 = hh->b1; A.b2 = hh->b2;
		A.px = hh->px; A.py = hh->py; A.pz = hh->pz; A.nx = hh->nx; A.ny = hh->ny; A.nz = hh->nz; A.u = hh->u; A.v = hh->v;
	} else {
		HIP_TRY(c->stage_a.ensure(6 * n * 4));
		HIP_TRY(c->stage_b.ensure((size_t)n_out * n * 4));
		float* in = (float*)c->stage_a.p;
		const float* src[6] = {r->ox, r->oy, r->oz, r->dx, r->dy, r->dz};
		for (int k = 0; k < 6; k++) HIP_TRY(hipMemcpyAsync(in + k * n, src[k], n * 4, hipMemcpyHostToDevice, c->stream));
		A.ox = in; A.oy = in + n; A.oz = in + 2 * n; A.dx = in + 3 * n; A.dy = in + 4 * n; A.dz = in + 5 * n;
		float* o = (float*)c->stage_b.p;
		A.distance = o; A.surface = (int32_t*)(o + n); A.triangle = (int32_t*)(o + 2 * n);
		A.b0 = o + 3 * n; A.b1 = o + 4 * n; A.b2 = o + 5 * n;
		size_t k = 6;
		if (hh->px) { A.px = o + k * n; A.py = o + (k + 1) * n; A.pz = o + (k + 2) * n; k += 3; }
		if (hh->nx) { A.nx = o + k * n; A.ny = o + (k + 1) * n; A.nz = o + (k + 2) * n; k += 3; }
		if (hh->u) { A.u = o + k * n; A.v = o + (k + 1) * n; }
	}
	if (use_wavefront(sc)) {
		// queue-based pipeline, a slice of the batch at a time: as many rays as the pool serves at the pairs per ray this scene was seen
		// to need; a slice whose pairs do not fit is repeated smaller (the ratio it reported is remembered on the scene)
		const size_t n_surf = sc->host.surfaces.size();
		// the pool: what the whole batch is expected to need (in steps of 16 Mi pairs), at most kWfBatchPairs — one launch for a batch of
		// up to ~60 M rays of a 24-surface scene; larger batches go in slices
		uint64_t pool_pairs = std::min<uint64_t>(kWfBatchPairs, (((uint64_t)((double)n * wf_ratio_guess(sc) * 1.1) >> 24) + 1) << 24);
		if (const char* e = getenv("PTX_WF_PAIRS_M")) pool_pairs = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;
		auto slice_cap = [&]() { return (size_t)std::max(16384.0, (double)pool_pairs / wf_ratio_guess(sc)); };
		size_t slice = std::min<size_t>(n, slice_cap());
		WfBuffers W{};
		HIP_TRY(wf_workspace(c, 0, slice, pool_pairs, n_surf, 1, W));
		ptx_ctx::WfSet& ws = c->wf[0];
		for (size_t first = 0; first < n;) {
			const uint32_t m = (uint32_t)std::min(slice, n - first);
			HIP_TRY(hipMemsetAsync(ws.ctl.p, 0, kWfCtlWords * 4, c->stream));
			HIP_TRY(hipMemsetAsync(ws.flow.p, 0, kWfFlowWords * 4, c->stream));
			DevScene batch_dev = sc->dev;
			batch_dev.wf_order += sc->dev.n_surfaces;   // the batch order of the queues (upload_scene)
			HIP_TRY(launch_wf_intersect(batch_dev, A, first, m, W, c->n_cu, c->stream));
			HIP_TRY(hipMemcpyAsync(ws.flow_host, ws.flow.p, kWfFlowWords * 4, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (ws.flow_host[kWfFlowPeak]) sc->wf_pairs_per_ray = std::max(sc->wf_pairs_per_ray, (double)ws.flow_host[kWfFlowPeak] / (double)m);
#ifdef PTX_WF_PROF
			uint32_t ctl[kWfCtlCur];
			HIP_TRY(hipMemcpy(ctl, W.ctl, sizeof ctl, hipMemcpyDeviceToHost));
			const uint32_t* pr = ctl + kWfCtlProf;
			static const char* names[8] = {"outer_round", "busy_round", "node_step", "tri_test", "hand_out", "unit_fetch", "pop", "stack_spill"};
			fprintf(stderr, "WFPROF rays %u pairs %u (%.2f per ray)\n", m, ctl[0], (double)ctl[0] / (double)m);
			for (int k = 0; k < 8; k++)
				fprintf(stderr, "WFPROF %-12s trips %10u lanes %11u  util %.3f  lanes/pair %.2f\n", names[k], pr[2 * k], pr[2 * k + 1],
				        pr[2 * k] ? (double)pr[2 * k + 1] / (64.0 * pr[2 * k]) : 0.0, (double)pr[2 * k + 1] / (double)ctl[0]);
			static const char* tn[6] = {"kernel", "unit_fetch", "hand_out", "pop", "descend", "leaf"};
			fprintf(stderr, "WFHIST waves by log2(kilocycles of their run):");
			for (int k = 0; k < 31; k++) if (pr[32 + k]) fprintf(stderr, " [2^%d]=%u", k, pr[32 + k]);
			fprintf(stderr, "  waves that never had a busy round: %u\n", pr[32 + 31]);
			fprintf(stderr, "WFMAX slowest wave %u kcycles, most trips of a wave %u, longest walk of a lane %u steps\n", pr[24], pr[25], pr[26]);
			for (int k = 0; k < 6; k++) fprintf(stderr, "WFCLK %-10s %10u kcycles summed over waves (%.1f %%)\n", tn[k], pr[16 + k], 100.0 * pr[16 + k] / (double)pr[16]);
#endif
			if (ws.flow_host[kWfFlowOverflow]) {
				if (slice <= 16384) return set_err(PTX_ERR_HIP, "queue-based pipeline: the pair pool cannot hold a 16384-ray slice");
				slice = std::max<size_t>(16384, std::min(slice_cap(), slice - slice / 4));
				continue;   // the same rays again, fewer at a time
			}
			first += m;
		}
	} else {
		const int grid = (int)std::min<size_t>((size_t)c->n_cu, (n + kBlock - 1) / kBlock);
		HIP_TRY(c->spill.ensure((size_t)c->n_cu * (kBlock / 64) * (size_t)kSpillWords * sizeof(uint2)));
		A.spill = (uint2*)c->spill.p;
		HIP_TRY(launch_intersect(sc->dev, A, sc->mode, sc->lds_bytes, grid, c->stream));
	}
	if (!dev) {
		void* dst[14] = {hh->distance, hh->surface, hh->triangle, hh->b0, hh->b1, hh->b2, hh->px, hh->py, hh->pz, hh->nx, hh->ny, hh->nz, hh->u, hh->v};
		const void* srcs[14] = {A.distance, A.surface, A.triangle, A.b0, A.b1, A.b2, A.px, A.py, A.pz, A.nx, A.ny, A.nz, A.u, A.v};
		for (int k = 0; k < 14; k++)
			if (dst[k]) HIP_TRY(hipMemcpyAsync(dst[k], srcs[k], n * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	return PTX_OK;
}

int ptx_pbr_eval_batch(ptx_ctx* c, const float* in, size_t n, float* out) {
	if (!c) return set_err(PTX_ERR_NO_DEVICE, "ptx_pbr_eval_batch: no GPU context (no CPU path exists)");
	if (n == 0) return PTX_OK;
	if (!in || !out) return set_err(PTX_ERR_INVALID, "ptx_pbr_eval_batch: NULL argument");
	if (n > (size_t)0x7FFFFFFF) return set_err(PTX_ERR_INVALID, "ptx_pbr_eval_batch: batch too large");
	std::lock_guard<std::mutex> lk(c->mu);
	HIP_TRY(hipSetDevice(c->device));
	const bool dev = is_device_ptr(in);
	if (dev != is_device_ptr(out)) return set_err(PTX_ERR_INVALID, "ptx_pbr_eval_batch: in and out must both be device or both be host memory");
	const float* d_in = in;
	float* d_out = out;
	if (!dev) {
		HIP_TRY(c->stage_a.ensure(n * 14 * 4));
		HIP_TRY(c->stage_b.ensure(n * 15 * 4));
		HIP_TRY(hipMemcpyAsync(c->stage_a.p, in, n * 14 * 4, hipMemcpyHostToDevice, c->stream));
		d_in = (const float*)c->stage_a.p;
		d_out = (float*)c->stage_b.p;
	}
	HIP_TRY(launch_pbr_eval(d_in, d_out, n, c->stream));
	if (!dev) {
		HIP_TRY(hipMemcpyAsync(out, d_out, n * 15 * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	return PTX_OK;
}

int ptx_camera_rays_batch(ptx_scene* sc, const float* ndc_ratio, size_t n, float* rays) {
	if (!sc) return set_err(PTX_ERR_INVALID, "ptx_camera_rays_batch: scene is NULL");
	if (!sc->ctx) return set_err(PTX_ERR_NO_DEVICE, "ptx_camera_rays_batch: scene was created without a GPU context (no CPU path exists)");
	if (n == 0) return PTX_OK;
	if (!ndc_ratio || !rays) return set_err(PTX_ERR_INVALID, "ptx_camera_rays_batch: NULL argument");
	ptx_ctx* c = sc->ctx;
	std::lock_guard<std::mutex> lk(c->mu);
	HIP_TRY(hipSetDevice(c->device));
	const bool dev = is_device_ptr(ndc_ratio);
	if (dev != is_device_ptr(rays)) return set_err(PTX_ERR_INVALID, "ptx_camera_rays_batch: in and out must both be device or both be host memory");
	const float* d_in = ndc_ratio;
	float* d_out = rays;
	if (!dev) {
		HIP_TRY(c->stage_a.ensure(n * 3 * 4));
		HIP_TRY(c->stage_b.ensure(n * 6 * 4));
		HIP_TRY(hipMemcpyAsync(c->stage_a.p, ndc_ratio, n * 3 * 4, hipMemcpyHostToDevice, c->stream));
		d_in = (const float*)c->stage_a.p;
		d_out = (float*)c->stage_b.p;
	}
	HIP_TRY(launch_camera_rays(sc->dev, d_in, d_out, n, c->stream));
	if (!dev) {
		HIP_TRY(hipMemcpyAsync(rays, d_out, n * 6 * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	return PTX_OK;
}

int ptx_reduce_framebuffer(ptx_ctx* c, void* nccl_comm, float* accum, size_t n_floats, int root) {
	if (!c || !nccl_comm || !accum) return set_err(PTX_ERR_INVALID, "ptx_reduce_framebuffer: NULL argument");
	if (!is_device_ptr(accum)) return set_err(PTX_ERR_INVALID, "ptx_reduce_framebuffer: accum must be device memory");
	// ncclResult_t ncclReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) — rccl.h
	using reduce_fn = int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
	static reduce_fn fn = [] {
		void* sym = dlsym(RTLD_DEFAULT, "ncclReduce");            // the RCCL that created the communicator, if already loaded
		if (!sym)
			if (void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL)) sym = dlsym(h, "ncclReduce");
		return reinterpret_cast<reduce_fn>(sym);
	}();
	if (!fn) return set_err(PTX_ERR_UNSUPPORTED, "ptx_reduce_framebuffer: no RCCL (ncclReduce) in this process and librccl.so cannot be loaded");
	std::lock_guard<std::mutex> lk(c->mu);
	HIP_TRY(hipSetDevice(c->device));
	constexpr int kNcclFloat32 = 7, kNcclSum = 0;                 // rccl.h: ncclFloat32 = 7, ncclSum = 0
	const int rc = fn(accum, accum, n_floats, kNcclFloat32, kNcclSum, root, nccl_comm, c->stream);
	if (rc != 0) return set_err(PTX_ERR_HIP, "ncclReduce failed with ncclResult_t " + std::to_string(rc));
	return PTX_OK;
}

int ptx_tonemap_encode(ptx_ctx* c, const float* accum, uint32_t W, uint32_t H, uint32_t spp, uint8_t* rgba8) {
	if (!c) return set_err(PTX_ERR_NO_DEVICE, "ptx_tonemap_encode: no GPU context (no CPU path exists)");
	if (!accum || !rgba8 || !W || !H || !spp) return set_err(PTX_ERR_INVALID, "ptx_tonemap_encode: bad argument");
	std::lock_guard<std::mutex> lk(c->mu);
	HIP_TRY(hipSetDevice(c->device));
	const size_t n = (size_t)W * H;
	const bool dev_in = is_device_ptr(accum), dev_out = is_device_ptr(rgba8);
	const float4* d_in = (const float4*)accum;
	uchar4* d_out = (uchar4*)rgba8;
	if (!dev_in) {
		HIP_TRY(c->stage_a.ensure(n * 16));
		HIP_TRY(hipMemcpyAsync(c->stage_a.p, accum, n * 16, hipMemcpyHostToDevice, c->stream));
		d_in = (const float4*)c->stage_a.p;
	}
	if (!dev_out) {
		HIP_TRY(c->stage_b.ensure(n * 4));
		d_out = (uchar4*)c->stage_b.p;
	}
	if (!c->srgb_thr.p) {
		float thr[256];
		srgb_thresholds(thr);
		HIP_TRY(c->srgb_thr.ensure(sizeof thr));
		HIP_TRY(hipMemcpy(c->srgb_thr.p, thr, sizeof thr, hipMemcpyHostToDevice));
	}
	HIP_TRY(launch_tonemap(d_in, (uint32_t)n, (float)spp, (const float*)c->srgb_thr.p, d_out, c->stream));
	if (!dev_out) {
		HIP_TRY(hipMemcpyAsync(rgba8, d_out, n * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	return PTX_OK;
}

int ptx_encode_png(const uint8_t* rgba8, uint32_t W, uint32_t H, uint8_t** png, size_t* png_bytes) {
	if (!rgba8 || !png || !png_bytes || !W || !H) return set_err(PTX_ERR_INVALID, "ptx_encode_png: bad argument");
	const size_t row = (size_t)W * 4;
	std::vector<uint8_t> raw((row + 1) * H);
	for (uint32_t y = 0; y < H; y++) {
		raw[(row + 1) * y] = 0;  // filter type None
		memcpy(&raw[(row + 1) * y + 1], rgba8 + row * y, row);
	}
	uLongf zcap = compressBound((uLong)raw.size());
	std::vector<uint8_t> z(zcap);
	if (compress2(z.data(), &zcap, raw.data(), (uLong)raw.size(), 6) != Z_OK) return set_err(PTX_ERR_INVALID, "zlib compress2 failed");
	const size_t total = 8 + (12 + 13) + (12 + zcap) + 12;
	uint8_t* out = (uint8_t*)malloc(total);
	if (!out) return set_err(PTX_ERR_INVALID, "out of memory");
	uint8_t* p = out;
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	memcpy(p, sig, 8); p += 8;
	auto be32 = [](uint8_t* q, uint32_t v) { q[0] = v >> 24; q[1] = v >> 16; q[2] = v >> 8; q[3] = v; };
	auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
		be32(p, len); memcpy(p + 4, type, 4);
		if (len) memcpy(p + 8, data, len);
		be32(p + 8 + len, (uint32_t)crc32(0, p + 4, len + 4));
		p += 12 + len;
	};
	uint8_t ihdr[13];
	be32(ihdr, W); be32(ihdr + 4, H);
	ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8-bit RGBA
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", z.data(), (uint32_t)zcap);
	chunk("IEND", nullptr, 0);
	*png = out;
	*png_bytes = total;
	return PTX_OK;
}

void ptx_free(void* p) { free(p); }

}  // extern "C"
