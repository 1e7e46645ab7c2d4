// Kernel argument blocks and launchers shared by kernels.hip and ptx_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

#include "flat_scene.hpp"

namespace ptx {

#ifndef PTX_BLOCK
#define PTX_BLOCK 1024
#endif
constexpr int kBlock = PTX_BLOCK;  // threads per workgroup (one workgroup per CU): 1024 = 16 waves = 4 per SIMD
#ifndef PTX_CHUNK
#define PTX_CHUNK 2048
#endif
constexpr uint32_t kChunk = PTX_CHUNK; // camera paths a wave takes per counter fetch (32 wave-iterations). Measured with 64 spp per launch: 512 -7 %, 1024 0, 2048 +5.6 %, 4096 +6.3 % on Cornell; 4096 -12 % on the open plaza scene
// wave-private stream space, in float4
// Per-wave stream area in float4: 2 x 4 ray arrays + hit records (9 kChunk), hit distances (1/2), shadow requests (3), then one
// deferred list (2 x kListCap float4) per unit that can be set aside: per model, or per surface in the SURF kernels (<= 64 units:
// the list lengths live in the lanes of one VGPR).
#ifndef PTX_INLINE_MIN
#define PTX_INLINE_MIN 32
#endif
constexpr uint32_t kInlineMin = PTX_INLINE_MIN;        // lanes of a wave-iteration that make a unit worth traversing on the spot
constexpr uint32_t kListCap = (kChunk / 64u) * (kInlineMin - 1u) + 16u;   // entries a list can receive per chunk (kChunk / 64 iterations)
constexpr int kMaxDeferModels = 64;
constexpr uint32_t kQueueFixedFloat4 = 12u * kChunk + kChunk / 2u;
inline uint32_t queue_float4_per_wave(uint32_t n_units) { return kQueueFixedFloat4 + (n_units <= (uint32_t)kMaxDeferModels ? n_units : 0u) * 2u * kListCap; }
constexpr uint32_t kSpillWords = 24u * 64u;  // uint2 per wave: kSpillStack levels x 64 lanes

// Device view of a FlatScene (all pointers are device pointers).
// Where a scene's KD nodes and triangle records live while a kernel runs (kernel template parameter MODE):
// everything in global memory (L2/HBM) | everything staged in LDS | hybrid: the surfaces that fit one CU's LDS
// (SurfaceRec::lds_root valid) staged, the large ones in global memory.
enum : int { MODE_GLOBAL = 0, MODE_LDS = 1, MODE_HYBRID = 2 };

struct DevScene {
	const ModelRec* models;
	const SurfaceRec* surfaces;
	const MaterialRec* materials;
	const uint2* nodes;      // KD nodes of all surfaces (global-memory traversal)
	const uint32_t* refs;    // unused by the kernels since the global path reads leaf-ordered records; kept for ptx_scene_get_array parity
	const float4* tris;   // 9 per triangle: the hit record (HitRec: corners + u, normals + v, tangents)
	// The triangle word that traversal records carry and hits report is `id | slot << 24` when tri_id_mask == 0x00FFFFFF (scenes of fewer
	// than 2^24 triangles): slot < n_hot = the triangle's hit record is also among the n_hot records of `hot_hitrec`, which the LDS kernels
	// stage into LDS (hot_lds, set inside the kernel) — the largest triangles, which take most hits; 0xFF = only in `tris`.
	const float4* hot_hitrec;
	const float4* hot_lds;
	uint32_t n_hot, tri_id_mask;
	const float4* tri_isect; // global-memory traversal: one TriIsect per leaf reference, in leaf order, triangle id in word 10; same allocation as `nodes`, behind them
	uint64_t geom_bytes;     // bytes of that allocation (nodes + records [+ nodes2])
	const uint2* nodes2;     // nullptr, or the KD nodes in 2-level blocks (wavefront.hip: BLOCK2; built when PTX_WF_BLOCK2 is set at scene creation), same allocation
	const uint2* roots2;     // [n_surfaces] root node contents for nodes2
	// resident copy (staged into LDS by MODE_LDS / MODE_HYBRID kernels): nodes, refs and one TriIsect per triangle of the
	// surfaces that fit, indices rewritten to be local to these arrays (SurfaceRec::lds_root)
	const uint2* res_nodes;
	const uint32_t* res_refs;
	const float4* res_tris;
	uint32_t n_res_nodes, n_res_refs, n_res_tris;
	const ShadeRec* shade; // 1 per surface
	const SpaceRec* spaces; // distinct world->local transforms
	const TexRec* tex;       // textures
	const uint8_t* texels;   // 8-bit texel bytes of all textures
	const float* texels_f;   // float texels of Radiance .hdr images (TexRec::c_srgb & kTexFloat)
	const float* srgb_lut;   // [256] pow(b / 255, 2.2)
	uint32_t glb_leaf_ordered; // layout of tri_isect: 1 = one record per leaf reference (leaf order), 0 = one per triangle (reached through refs)
	uint32_t any_texture;
	int32_t env_tex;         // environment map (renderer::environment): texture index or -1
	const uint32_t* model_space; // per model
	const uint32_t* wf_order;    // [n_surfaces] queue-based pipeline: the surfaces in the order the traverse kernel starts their queues (upload_scene)
	int32_t n_models;
	uint32_t n_surfaces, n_nodes, n_refs, n_tris;
	uint32_t any_alpha;
	uint32_t n_spaces;
	CameraRec cam;
	SunRec sun;
};

struct RenderParams {
	uint32_t W, H;              // full image
	uint32_t x0, y0, w, h;      // tile
	uint32_t n_pixels;          // w*h
	uint32_t sample0;           // first sample index of this pass
	uint32_t pass_spp;          // samples per pixel in this pass
	uint32_t bounces;
	uint64_t n_paths;           // n_pixels * pass_spp
	uint32_t seed_lo, seed_hi;
	float env[3];
	uint32_t integrator;        // ptx_integrator
	const uint32_t* pixels;     // nullptr: the pass covers every pixel of the tile; else [n_pixels] tile-local pixel indices (ly * w + lx)
	                            // of the pixels this pass renders (interleaved tile sharding), and n_pixels is the list's length
};

constexpr int kProfRegions = 16;   // PTX_PROF builds only: (wave-level trips, active lanes) per code region

struct PassBuffers {
	float4* queues;                   // [n_wave_slots][queue_stride]
	uint32_t queue_stride;            // float4 per wave: queue_float4_per_wave(units)
	uint32_t surface_units;           // 1: the SURF kernels (single surfaces are set aside), 0: whole models
	float4* sample_rad;               // [pass_spp][n_pixels]
	uint2* spill;                     // [n_wave_slots][kSpillWords]: traversal-stack overflow, lane-interleaved
	unsigned long long* chunk_counter; // paths handed out so far; zeroed before each pass
	unsigned long long* ray_counter;  // accumulates
};

struct IntersectArgs {
	const float *ox, *oy, *oz, *dx, *dy, *dz;
	size_t n;
	float* distance; int32_t* surface; int32_t* triangle;
	float *b0, *b1, *b2;
	float *px, *py, *pz, *nx, *ny, *nz, *u, *v;  // optional groups (nullptr = skip)
	uint2* spill;                                // [grid * waves per block][kSpillWords]
};

// Workspace of the queue-based pipeline (wavefront.hip). A PAIR is (ray, surface whose box it enters). Pair space is a POOL sized from
// demand (a ray enters 3-5 of an atrium's 24 surface boxes, not all of them): every classify tile (1024 rays) reserves its pairs with
// ONE atomic and uses its block of the pool twice — as queue entries grouped by surface (what the traversal waves read, contiguous
// per surface: one SEGMENT per tile and surface) and as result slots grouped by ray (what the shading / merge kernels read: the
// pairs of a ray are consecutive, in surface order). A step whose pairs do not fit the pool raises the overflow word; the host then
// repeats the slab in smaller pieces.
struct WfBuffers {
	float4* qent;                  // [pool_cap][2] queue entries: (local origin, result slot bits) (local direction, -)
	float4* pair_hit;              // [pool_cap] results: (local t, global triangle id bits, beta, gamma); t = -1: the walk found nothing
	uint32_t pool_cap;             // pairs the pool holds
	uint2* seg;                    // [n_surfaces][seg_cap]: (first entry, entries) of the segments of each surface's queue
	uint32_t seg_cap;              // classify tiles of a step at most
	uint32_t* first;               // [rays] first result slot of the ray
	unsigned long long* mask;      // [rays] bit u: the ray enters surface u
	uint32_t* ctl;                 // control block of THIS step (kWfCtlWords words, zeroed before the step; layout below)
	const uint32_t* n_in;          // device word: stream entries of this step (render) / rays of this slice (batch intersect)
	uint32_t* overflow;            // device word, sticky over the steps of a slab: some step's pairs did not fit the pool
	uint32_t* peak;                // device word: the most pairs any step of the slab asked for (what the host sizes the next slab by)
	uint2* spill;                  // [wf_traverse_grid * 4 waves][kSpillWords]
	unsigned long long* ray_counter;   // nullptr, or where classify adds the number of rays it was given (render statistics)
	uint32_t wave_clock;           // 1: every traverse wave that found work adds its run time to ctl[kWfCtlClock ..] (ptx_ctx_set_timing)
};
// control block of a step (uint32 words): [0] pairs reserved so far, [1] this step overflowed the pool, [kWfCtlSeg + u] segments of
// surface u's queue, [kWfCtlCur + 64 u] hand-out cursor of surface u's segments (256 bytes apart: two dozen hot counters in one
// cache line serialised every hand-out of the chip in one L2 channel), [kWfCtlProf ..] PTX_WF_PROF region counters
constexpr uint32_t kWfCtlSeg = 16, kWfCtlProf = 80, kWfCtlCur = 192, kWfCtlWords = kWfCtlCur + 64u * 64u;
constexpr uint32_t kWfCtlClock = kWfCtlProf + 72;   // [+0, +1] 64-bit sum of the working waves' run times (s_memtime ticks), [+2] such waves, [+3] the longest run
#ifndef PTX_WF_TILE
#define PTX_WF_TILE 1024
#endif
constexpr uint32_t kWfTile = PTX_WF_TILE;   // rays per classify tile = threads of a classify workgroup (measured: profiles/round3_wf_ab.txt)
// One of the two path-stream buffers of a render slab (SoA of float4, `cap` entries per array)
struct WfStream {
	float4* q;   // [4][cap]: (origin, id | flags) (direction, T.x) (T.y, T.z, L.x, L.y) (L.z, depth << 16 | pass, RNG key pixel, RNG key sample) — the fused kernel's entry
	float4* r;   // [3][cap]: the entry's shadow request: (origin, -) (direction, -) (x: radiance to add when unoccluded | origin of a shadow catcher's pass-through ray, -)
};
// flags in the id word of a stream entry (ids are slab-local, < 2^28)
constexpr uint32_t kWfIdMask = 0x0FFFFFFFu, kWfZombie = 1u << 31, kWfPending = 1u << 30, kWfRequest = 1u << 29;
constexpr uint32_t kWfMaxSlab = 1u << 27;   // paths of a slab at most: the whole 128 Mi-path pass (the stream buffers of a slab take 224 bytes per path; ids have 28 bits)
constexpr int kWfMaxSurfaces = 64;   // surface masks are one 64-bit word
// persistent 256-thread workgroups of the traverse kernel: as many as can be resident (59 VGPRs and 24 KB of LDS allow 6 per CU; 8 are launched).
// PTX_WF_GRID=<workgroups per CU> (measurement): fewer leave room for another stream's kernels
inline int wf_traverse_grid(int n_cu) {
	static const int per_cu = [] { const char* e = getenv("PTX_WF_GRID"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 8 ? v : 8; }();
	return n_cu * per_cu;
}
// a slab's steps run back to back on the device: step s reads its entry count from flow[s] and adds what it emits to flow[s + 1]
hipError_t launch_wf_generate(const DevScene& S, const RenderParams& P, const WfStream& out, uint32_t cap, uint32_t first, uint32_t n, float4* sample_rad, hipStream_t stream);
hipError_t launch_wf_step(const DevScene& S, const RenderParams& P, const WfBuffers& W, const WfStream& in, const WfStream& out, uint32_t cap, uint32_t max_in,
                          uint32_t slab_first, uint32_t* n_out, float4* sample_rad, int n_cu, hipStream_t stream, hipEvent_t* ev /* nullptr, or 4 events: before classify / traverse / shade, after */);
hipError_t launch_wf_intersect(const DevScene& S, const IntersectArgs& A, size_t first_ray, uint32_t n, const WfBuffers& W, int n_cu, hipStream_t stream);

hipError_t launch_render_pass(const DevScene& S, const RenderParams& P, const PassBuffers& B, int mode, size_t lds_bytes, int grid,
                              hipStream_t stream);
hipError_t launch_resolve(const float4* sample_rad, float4* accum, const uint32_t* pixels, uint32_t n_pixels, uint32_t pass_spp, hipStream_t stream);
hipError_t launch_intersect(const DevScene& S, const IntersectArgs& A, int mode, size_t lds_bytes, int grid, hipStream_t stream);
hipError_t launch_pbr_eval(const float* in, float* out, size_t n, hipStream_t stream);
hipError_t launch_camera_rays(const DevScene& S, const float* in /* [n][3]: ndc.x ndc.y ratio */, float* out /* [n][6] */, size_t n, hipStream_t stream);
hipError_t launch_tonemap(const float4* accum, uint32_t n_pixels, float spp, const float* thresholds /* [256], device */, uchar4* out, hipStream_t stream);

}  // namespace ptx
