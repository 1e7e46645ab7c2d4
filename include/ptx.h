/* ptx.h — C ABI of the MI355X-native path-tracing integrator (libptx_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of vmanam0451/distributed-path-tracer: the
 * ray-intersection + Monte-Carlo shading integrator of path-tracer-core/path_tracer_lib.
 * The reference has no FFI layer (the host links the library statically and calls C++ classes,
 * path-tracer-core/CMakeLists.txt:44); each entry point below names the reference interface it
 * replaces. Paths are relative to path-tracer-core/; LIB = path_tracer_lib/path_tracer.
 *
 * Conventions: plain pointers and sizes only; every function returns a ptx_status (0 = ok) and never
 * throws; ptx_last_error() gives the thread-local message of the last failure. Buffers that the
 * documentation marks "device or host" may be either: the library inspects the pointer
 * (hipPointerGetAttributes) and stages host buffers through its own device workspace.
 * A scene is immutable after creation; one context per GPU; calls on one context are serialised
 * on that context's HIP stream.
 */
#ifndef PTX_H
#define PTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ptx_status {
	PTX_OK = 0,
	PTX_ERR_INVALID = 1,    /* bad argument */
	PTX_ERR_IO = 2,         /* file could not be read (LIB/core/renderer.cpp:64-69 only logs this and then crashes) */
	PTX_ERR_PARSE = 3,      /* malformed glTF */
	PTX_ERR_NO_CAMERA = 4,  /* "Scene does not contain camera #i." / "Scene is missing a camera." renderer.cpp:73-74,97-98 */
	PTX_ERR_NO_DEVICE = 5,  /* no HIP device / host-only scene used for GPU work: the product never falls back to a CPU path */
	PTX_ERR_HIP = 6,        /* HIP runtime error (message carries hipGetErrorString) */
	PTX_ERR_UNSUPPORTED = 7 /* feature outside the built scope (e.g. non-PNG or interlaced textures), refused rather than rendered wrong */
} ptx_status;

typedef struct ptx_ctx ptx_ctx;     /* one per GPU: device, stream, workspace */
typedef struct ptx_scene ptx_scene; /* flattened, immutable scene (+ its device copy) */

const char* ptx_last_error(void);
const char* ptx_version(void);

/* ---- context ------------------------------------------------------------------------------------
 * device >= 0: HIP device ordinal. Fails with PTX_ERR_NO_DEVICE when there is none. */
int ptx_ctx_create(int device, ptx_ctx** out);
/* Scenes created on the context keep it alive: it is freed together with the last of them, in whatever order the handles
 * are destroyed. */
void ptx_ctx_destroy(ptx_ctx* ctx);
/* Stream the context launches on (hipStream_t as void*), for callers that time or order work. */
void* ptx_ctx_stream(ptx_ctx* ctx);
int ptx_ctx_synchronize(ptx_ctx* ctx);

/* ---- scene load ---------------------------------------------------------------------------------
 * Replaces core::renderer::load_gltf(path) (LIB/core/renderer.hpp:35, renderer.cpp:61-331) and the
 * host's cloud::distributed_scene::load_scene (src/scene/scene.hpp:19): parses the glTF, builds the
 * entity transforms, unpacks every primitive (with the reference's loader quirks), builds one SAH
 * KD-tree per primitive with the reference's topology (LIB/core/mesh.cpp:131-298) and flattens all of
 * it into pointer-free arrays. ctx may be NULL: the scene is then host-only (inspection, no GPU work). */
typedef struct ptx_work_item {   /* one entry of models::work_info::work (src/models/work_info.hpp:11-15) */
	const char* mesh_name;
	const int32_t* primitives;
	uint32_t n_primitives;
} ptx_work_item;
typedef struct ptx_load_opts {
	uint32_t camera_index;    /* renderer.hpp:31, default 0 */
	uint32_t sun_light_index; /* renderer.hpp:32, default 0; 0xFFFFFFFF = renderer::no_sun_light */
	/* The host's per-worker primitive filter (distributed_scene::load_scene's scene_work, src/scene/load_gltf.cpp:93-99):
	 * when filter_primitives != 0 only the listed primitive indices of each named mesh are loaded, and a mesh that is not
	 * listed loads none (its model stays, empty). 0 = load everything, as core::renderer::load_gltf does. */
	uint32_t filter_primitives;
	uint32_t n_work;
	const ptx_work_item* work;
} ptx_load_opts;
int ptx_scene_load_gltf(ptx_ctx* ctx, const char* gltf_path, const ptx_load_opts* opts /* NULL = defaults */,
                        ptx_scene** out);

/* The reference worker's entry: a Lambda event (models::worker_info, src/models/work_info.hpp:17-32; sample in
 * path-tracer-core/events/event.json) names the scene, this worker's primitives, samples, bounces and X x Y
 * (src/main.cpp:9-25 -> processors::worker::run, worker.cpp:25-38). S3 is out of scope: `local_scene_root` is a local
 * directory holding the event's scene_root files; `<local_scene_root>/scene.gltf` is loaded with the event's filter.
 * cfg receives W, H, spp, bounces (+ defaults for the rest), ready for ptx_render. info may be NULL. */
typedef struct ptx_worker_event {
	int32_t num_workers;
	uint32_t n_work_meshes;
	char worker_id[64];
	char scene_root[256];
	char scene_bucket[128];
} ptx_worker_event;
struct ptx_render_cfg;
int ptx_worker_event_load(ptx_ctx* ctx, const char* event_json_path, const char* local_scene_root, ptx_scene** scene,
                          struct ptx_render_cfg* cfg, ptx_worker_event* info /* NULL ok */);

/* Same, from caller-provided arrays (procedural scenes, or a host that did its own parsing).
 * Models are given in the order the reference's renderer::intersect would visit them. */
typedef struct ptx_scene_desc {
	uint32_t n_models;
	const float* model_xform;    /* [n_models][12]: origin(3), basis.x(3), basis.y(3), basis.z(3) — scene::transform */
	const int32_t* model_surf;   /* [n_models][2]: first surface, surface count — scene::model::surfaces */
	uint32_t n_surfaces;
	const int32_t* surf_range;   /* [n_surfaces][4]: first vertex, vertex count, first triangle, triangle count */
	const float* vertices;       /* [.][11]: position(3) tex_coord(2) normal(3) tangent(3) — core::vertex */
	const uint32_t* triangles;   /* [.][3]: mesh-local vertex ids — core::mesh::triangles */
	const float* materials;      /* [n_surfaces][11]: albedo(3) opacity roughness metallic emissive(3) ior shadow_catcher */
	const float* camera;         /* [13]: origin(3) basis(9) vertical fov (radians) */
	const float* sun;            /* NULL, or [13]: basis(9) energy(3) angular_radius — scene::sun_light */
} ptx_scene_desc;
int ptx_scene_from_arrays(ptx_ctx* ctx, const ptx_scene_desc* desc, ptx_scene** out);
void ptx_scene_destroy(ptx_scene* scene);

/* renderer::environment (LIB/core/renderer.hpp:28: std::shared_ptr<image::texture>, sampled on a miss through
 * core::equirectangular_proj, renderer.cpp:443-449 / shading_worker.cpp:28-35) = image_texture::load(png_path, srgb).
 * The miss colour becomes texture(dir) * environment_factor. png_path == NULL removes the map. The file may be a PNG, a JPEG or a
 * Radiance .hdr (by content, as stb_image decides); an .hdr keeps its float texels (image::hdr). */
int ptx_scene_set_environment(ptx_scene* scene, const char* png_path, int srgb);

typedef struct ptx_scene_info {
	uint32_t n_models, n_surfaces, n_vertices, n_triangles;
	uint32_t n_kd_nodes;      /* flattened 8-byte nodes (branches + leaves) */
	uint32_t n_kd_refs;       /* leaf triangle references */
	uint32_t kd_max_depth;
	uint32_t has_sun;
	uint32_t geometry_bytes;  /* nodes + refs + triangle records: what the kernels stage through LDS */
	uint32_t lds_resident;    /* where the kernels read KD nodes / triangle records from: 0 = L2/HBM, 1 = all of it staged in
	                           * each CU's LDS, 2 = hybrid (the surfaces that fit in LDS, the large ones in L2/HBM) */
	uint32_t n_textures;
} ptx_scene_info;
int ptx_scene_get_info(const ptx_scene* scene, ptx_scene_info* info);

/* Host copies of the flattened arrays (tests, tooling). Returns the element count; dst may be NULL
 * to query it. Element layouts are documented in DESIGN.md §"Data layout". */
typedef enum ptx_array {
	PTX_ARR_MODEL_XFORM = 0,  /* float[n_models][12] */
	PTX_ARR_MODEL_AABB = 1,   /* float[n_models][6]  */
	PTX_ARR_MODEL_SURF = 2,   /* int32[n_models][2]  */
	PTX_ARR_SURF_RANGE = 3,   /* int32[n_surfaces][8]: v0,nv,t0,nt,kd_root,n_nodes,ref0,n_refs */
	PTX_ARR_MESH_AABB = 4,    /* float[n_surfaces][6] */
	PTX_ARR_VERTICES = 5,     /* float[n_vertices][11] */
	PTX_ARR_TRIANGLES = 6,    /* uint32[n_triangles][3] */
	PTX_ARR_MATERIALS = 7,    /* float[n_surfaces][11] */
	PTX_ARR_KD_NODES = 8,     /* uint32[n_kd_nodes][2]  (packed device nodes) */
	PTX_ARR_KD_REFS = 9,      /* uint32[n_kd_refs]      (global triangle ids) */
	PTX_ARR_CAMERA = 10,      /* float[14]: origin basis fov tan_half_fov */
	PTX_ARR_SUN = 11,         /* float[13] or empty */
	PTX_ARR_MODEL_NAMES = 12, /* char[]: '\n'-separated entity names in visit order */
	PTX_ARR_TEXTURES = 13,    /* uint32[n_textures][4]: width, height, channels | srgb << 8 | float << 16, byte offset into TEXELS (float offset into TEXELS_F32) */
	PTX_ARR_TEXELS = 14,      /* uint8[]: 8-bit texels of all textures (rows top to bottom, as decoded) */
	PTX_ARR_SURF_TEX = 15,    /* int32[n_surfaces][7]: texture id per material slot (normal, albedo, opacity, occlusion, roughness, metallic, emissive), -1 = none */
	PTX_ARR_TEXELS_F32 = 16   /* float[]: texels of Radiance .hdr images (TEXTURES entries with bit 16 of the third word; their offset counts floats here) */
} ptx_array;
int64_t ptx_scene_get_array(const ptx_scene* scene, ptx_array which, void* dst, size_t dst_bytes);

/* ---- tile worker --------------------------------------------------------------------------------
 * Replaces core::renderer::render() (LIB/core/renderer.hpp:36, renderer.cpp:334-428) and the worker's
 * staged pipeline (src/processors/worker/worker.hpp:27-41): renders samples [sample0, sample0+spp) of
 * the pixel rectangle [x0,x0+w) x [y0,y0+h) of a W x H image and ADDS the per-pixel radiance SUMS
 * (not means) into accum_rgba[h][w][4] (float32; alpha accumulates 1 per sample, as renderer.cpp:398).
 * The fields mirror core::renderer's public fields (renderer.hpp:21-33) with the same defaults.
 * Random numbers are a counter-based Philox4x32-10 stream keyed by (seed, pixel y*W+x, sample index,
 * depth, draw), so any tiling / sample split / GPU count gives the same per-sample radiance. */
/* Which of the reference's two estimators one sample runs.
 * PTX_INTEGRATOR_LIB:    core::renderer::trace (LIB/core/renderer.cpp:437-643) — what `path_tracer_lib` and its example
 *                        program render with; pinned against the compiled reference (oracle/_ref).
 * PTX_INTEGRATOR_WORKER: the HOST worker's stage pipeline for one worker — INTERSECT -> DIRECT_LIGHTING -> SHADING ->
 *                        ACCUMULATE (src/processors/worker/intersection_worker.cpp:10-67, shading_worker.cpp:10-201,
 *                        worker.cpp:114-149): emissive added before the opacity test, throughput clamped to [0,10],
 *                        Russian roulette once bounce < bounce_count-2, un-jittered sample 0, and a shadow catcher that
 *                        is black unless its sun sample is unoccluded. HOST cannot be built here: parity unpinned. */
typedef enum ptx_integrator { PTX_INTEGRATOR_LIB = 0, PTX_INTEGRATOR_WORKER = 1 } ptx_integrator;
typedef struct ptx_render_cfg {
	uint32_t W, H;          /* renderer::resolution (1920 x 1080) */
	uint32_t spp;           /* renderer::sample_count */
	uint32_t bounces;       /* renderer::bounce_count (4) */
	float env[3];           /* renderer::environment_factor (1,1,1) */
	uint32_t seed_lo, seed_hi;
	uint32_t x0, y0, w, h;  /* tile; w = h = 0 means the whole image */
	uint32_t sample0;       /* first sample index */
	uint32_t spp_per_pass;  /* 0 = library default; samples of every pixel traced per kernel launch */
	uint32_t integrator;    /* ptx_integrator */
	/* Interleaved tile sharding (one frame split over several GPUs / workers): when shard_count > 1, only
	 * the pixels of the rectangle that lie in image tiles t with t % shard_count == shard_index are rendered, where t is the
	 * row-major index of the shard_tile x shard_tile tile of the FULL W x H image that holds the pixel (shard_tile 0 = 64).
	 * accum keeps the [h][w] layout of the rectangle; pixels of other shards are left untouched, so the sum of all shards'
	 * buffers (x + 0 = x) is bitwise the unsharded frame. shard_count 0 or 1 = no sharding. */
	uint32_t shard_index, shard_count, shard_tile;
} ptx_render_cfg;
typedef struct ptx_render_stats {
	uint64_t rays;          /* closest-hit + shadow queries = renderer::intersect calls (renderer.cpp:441,509) */
	uint64_t samples;       /* camera paths = trace() root calls */
	uint64_t passes;        /* kernel launches of the integrator */
	double kernel_ms;       /* HIP-event time of the integrator kernels on the context stream */
} ptx_render_stats;
/* accum_rgba: device or host pointer. stats may be NULL (no device->host sync is then forced). */
int ptx_render(ptx_scene* scene, const ptx_render_cfg* cfg, float* accum_rgba, ptx_render_stats* stats);

/* Measurement aid (no counterpart in the reference): where the time of the last ptx_render that was given a stats pointer went.
 * Scenes whose geometry fits the LDS or whose models have few surfaces run ONE fused kernel per pass (pipeline 0: fused_ms);
 * many-surface scenes in global memory run the queue-based pipeline (pipeline 1) — per step of a slab of paths a classify, a
 * traverse and a shade kernel. Their HIP-event times are collected only after ptx_ctx_set_timing(ctx, 1) (four event records per
 * step); the workspace figures are always filled. */
typedef struct ptx_kernel_timing {
	uint32_t pipeline;          /* 0 = fused kernel, 1 = queue-based pipeline */
	uint32_t steps;             /* queue-based pipeline: steps timed (classify + traverse + shade each) */
	double classify_ms, traverse_ms, shade_ms;   /* sums over those steps */
	double fused_ms;            /* fused kernel: sum over its launches */
	uint32_t fused_launches;
	uint32_t pool_overflows;    /* queue-based pipeline: slabs that were repeated smaller because a step's pairs did not fit the pool */
	uint64_t pool_pairs;        /* queue-based pipeline: pairs (ray, entered surface) the pool holds, 48 bytes each */
	uint64_t peak_pairs;        /* ... the most pairs one step of one slab asked for */
	uint64_t slab_paths;        /* ... camera paths per slab */
	uint64_t workspace_bytes;   /* device memory the pipeline that ran holds on the context (streams, pool, queues) */
	double traverse_drain_frac; /* queue-based pipeline, with timing on: share of the traverse launches' wave-time between a wave running out of work
	                             * and the launch's last wave ending (the persistent waves' own clocks; 0 when not measured) */
} ptx_kernel_timing;
int ptx_ctx_set_timing(ptx_ctx* ctx, int on);
int ptx_ctx_get_timing(ptx_ctx* ctx, ptx_kernel_timing* out);

/* Batch form of renderer::intersect (renderer.cpp:645-725) / distributed_scene::intersect
 * (src/scene/scene.hpp:20-21): the unit the host's INTERSECT stage queue would call.
 * Rays are SoA; directions are used as given (the reference normalises on construction, ray.cpp:6-8,
 * so pass unit vectors). All pointers device or host (all of one kind). */
typedef struct ptx_rays {
	const float *ox, *oy, *oz, *dx, *dy, *dz;
} ptx_rays;
typedef struct ptx_hits {
	float* distance;    /* world-space hit distance; -1 = miss (model::intersection::distance, model.hpp:21) */
	int32_t* surface;   /* global surface (primitive) id, -1 = miss */
	int32_t* triangle;  /* triangle index within the surface's mesh */
	float *b0, *b1, *b2;            /* barycentrics (alpha, beta, gamma) — triangle.cpp:185-189 */
	float *px, *py, *pz;            /* world position        (may be NULL as a group of three) */
	float *nx, *ny, *nz;            /* shading normal = intersect_result::get_normal(), renderer.cpp:430-435 (may be NULL) */
	float *u, *v;                   /* interpolated tex_coord (may be NULL) */
} ptx_hits;
int ptx_intersect_batch(ptx_scene* scene, const ptx_rays* rays, size_t n, const ptx_hits* hits);

/* Batch form of scene::camera::get_ray(ndc, ratio) (LIB/scene/camera.cpp:10-21): what the integrator kernels compute for every camera
 * sample after the pixel jitter (renderer.cpp:359-370), evaluated by the same device function.
 * in [n][3]: ndc.x, ndc.y, aspect ratio;  out [n][6]: ray origin(3), direction(3). Pointers device or host (both of one kind). */
int ptx_camera_rays_batch(ptx_scene* scene, const float* ndc_ratio, size_t n, float* rays);

/* Batch form of the SHADING stage's sampling functions — core::pbr::importance_diffuse / importance_specular / pdf_diffuse /
 * pdf_specular / fresnel (LIB/core/pbr.cpp:71-184), util::rand_cone_vec (LIB/util/rand_cone_vec.cpp:8-35) and core::reflect
 * (LIB/core/utils.hpp:38-40) — evaluated by the same device functions the integrator kernel inlines. Function-level check of the
 * GPU's libm (ocml sin / cos / acos) against the reference's (glibc), and the unit a host SHADING stage queue would call.
 * in [n][14]: normal(3) outcoming(3) incoming(3) u1 u2 roughness cos_theta ior;
 * out[n][15]: rand_cone_vec(u2, cos_theta, normal)(3), importance_diffuse((u1,u2), normal)(3), importance_specular((u1,u2), normal,
 * outcoming, roughness)(3), pdf_diffuse(normal, incoming), pdf_specular(normal, outcoming, incoming, roughness),
 * fresnel(outcoming, reflect(-outcoming, normal), ior), reflect(-outcoming, normal)(3). Pointers device or host (both of one kind). */
int ptx_pbr_eval_batch(ptx_ctx* ctx, const float* in, size_t n, float* out);

/* ---- multi-GPU fan-in ------------------------------------------------------------------------------
 * The one exchange step of the path: the sum of the per-rank accumulation buffers on rank `root`. Replaces the
 * reference's planned (never implemented) SNS/SQS result fan-in (src/models/work_info.hpp:22-23,
 * src/processors/worker/intersection_worker.cpp:69-147). `nccl_comm` is an RCCL communicator the host created
 * (ncclCommInitRank: one rank per GPU); the library does not link RCCL, it resolves ncclReduce from the RCCL the
 * process has already loaded (or from librccl.so) at the first call, and enqueues
 * ncclReduce(accum, accum, n_floats, ncclFloat32, ncclSum, root, comm, ptx_ctx_stream(ctx)) — in place, device memory.
 * Call ptx_ctx_synchronize (or chain work on that stream) before reading the result. PTX_ERR_UNSUPPORTED when no RCCL
 * can be found, PTX_ERR_HIP when RCCL reports an error. */
int ptx_reduce_framebuffer(ptx_ctx* ctx, void* nccl_comm, float* accum_rgba, size_t n_floats, int root);

/* ---- image write --------------------------------------------------------------------------------
 * Replaces the tonemap + image::write loop of renderer.cpp:409-424 (core::tonemap_approx_aces,
 * LIB/core/utils.hpp:29-36; image::image::write, LIB/image/image.cpp:143-154): divides the sums by
 * `spp`, applies ACES, sRGB (pow 1/2.2) and quantises to RGBA8 row-major. accum/rgba8 device or host. */
int ptx_tonemap_encode(ptx_ctx* ctx, const float* accum_rgba, uint32_t W, uint32_t H, uint32_t spp, uint8_t* rgba8);
/* Replaces image::image::save_to_memory_png (LIB/image/image.cpp:111-122). Host memory only.
 * *png is malloc'ed; release with ptx_free. Decoded pixels are exact; the byte stream is zlib's, not stb's. */
int ptx_encode_png(const uint8_t* rgba8, uint32_t W, uint32_t H, uint8_t** png, size_t* png_bytes);
void ptx_free(void* p);

/* ---- multi-GPU ----------------------------------------------------------------------------------
 * The reference's planned fan-in (SNS/SQS, never implemented: src/models/work_info.hpp:22-23) is replaced
 * by ONE sum-reduce of the float accumulation buffer. The library does not own a communicator:
 * torch.distributed (backend "nccl" = RCCL over xGMI) reduces the device buffer ptx_render filled; see
 * INTEGRATION.md. */

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
