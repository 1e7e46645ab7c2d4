// Pointer-free scene layout shared by the host builder and the HIP kernels.
//
// The reference keeps the scene as a shared_ptr graph (entity -> model -> surfaces -> mesh ->
// kd_tree_node, LIB/scene/entity.hpp, LIB/core/kd_tree.hpp:20-31). Here everything is a flat array:
// fixed-size records for models / surfaces / materials (read with scalar loads: the loop index is
// wave-uniform), 8-byte KD nodes, 4-byte leaf references and 48-byte triangle records (the three
// arrays the kernels stage through LDS), and 144-byte per-triangle hit records that stay in HBM/L2.
#pragma once
#include <cstdint>
#include <vector>
#include <string>

namespace ptx {

// ---- KD node: 2 dwords -------------------------------------------------------------------------
// word1[1:0] = kind: 0,1,2 = branch on that axis, 3 = leaf
// branch: word0 = split plane (float bits); word1 bit2 = has_left, bit3 = has_right,
//         word1[31:4] = index of the first existing child; the right child sits at +has_left.
//         (a missing child is the reference's nullptr child, mesh.cpp:227-241)
// leaf:   word0 = index of the first leaf reference; word1[31:2] = reference count
struct KdNode { uint32_t w0, w1; };
constexpr uint32_t KD_LEAF = 3u;
inline KdNode kd_make_branch(float split, uint32_t axis, bool has_l, bool has_r, uint32_t first_child) {
	uint32_t bits;
	__builtin_memcpy(&bits, &split, 4);
	return {bits, axis | (has_l ? 4u : 0u) | (has_r ? 8u : 0u) | (first_child << 4)};
}
inline KdNode kd_make_leaf(uint32_t first_ref, uint32_t count) { return {first_ref, KD_LEAF | (count << 2)}; }

// ---- triangle corner record (host side only: counts and inspection; the device reads HitRec) ----
// xyz = corner position (mesh-local space), w = bit pattern of the GLOBAL vertex id of that corner
struct TriRec { float ax, ay, az; uint32_t ia; float bx, by, bz; uint32_t ib; float cx, cy, cz; uint32_t ic; };

// ---- triangle intersection record: 3 x float4 (48 B), the form the traversal kernels stage through LDS:
// a.xyz, c3 | e1 = a-b | e2 = a-c, with c3 = e1.y*e2.z - e2.y*e1.z — the ray-independent sub-terms of the
// reference's Cramer solve (triangle.cpp:136-147: m.x = a-b, m.y = a-c, c3), computed once on the host with
// the same float operations (so the same bits) instead of once per ray-triangle test.
// Ordered for packed-fp32 math (v_pk_mul_f32 / v_pk_add_f32 take even-aligned register pairs): every pair the solve
// multiplies lane-wise sits in one 8-byte slot, so the three LDS / global reads deliver it ready to use.
struct TriIsect { float e2y, e1z, e2z, e1y; float e1x, e2x, ay, az; float ax, c3, p0, p1; };

// ---- hit record: 9 x float4 (144 B) per triangle — everything renderer::intersect interpolates on a hit (renderer.cpp:688-715),
// gathered per triangle so that a hit costs ONE round of fetches: corners + u | normals + v | tangents. (Fetching the corner
// record first and the three vertices' attributes through its ids afterwards was two dependent rounds per shaded vertex.)
struct HitRec { float a[3], ua, b[3], ub, c[3], uc, na[3], va, nb[3], vb, nc[3], vc, ta[3], p0, tb[3], p1, tc[3], p2; };
static_assert(sizeof(HitRec) == 144, "HitRec layout");

// ---- per-model record (scene::model + its entity's global transform) ----
struct ModelRec {
	float inv_basis[9];   // columns x,y,z of inverse(basis)            — transform::inverse, transform.cpp:33-36
	float inv_origin[3];  // inverse(basis) * -origin
	float basis[9];       // columns of the global basis (local -> world; distance conversion model.cpp:62-63)
	float origin[3];
	float nmat[9];        // columns of transpose(inverse(basis))        — renderer.cpp:698
	float bmin[3], bmax[3];  // model AABB in local space                 — model.cpp:13-18
	int32_t first_surface, n_surfaces;
	float pbmin[3], pbmax[3];  // the box grown by the reach of the +-epsilon barycentric slack: every point the triangle tests of this
	                           // model can accept lies inside it (pruning of set-aside traversals, kernels.hip)
	float pad0;
	float box[6], pbox[6];     // the two boxes again as (min.x, max.x, min.y, max.y, min.z, max.z): the kernels' slab test works on (min, max)
	                           // pairs with packed fp32 instructions and takes each pair as one 64-bit scalar operand (interleave_boxes)
};
static_assert(sizeof(ModelRec) == 60 * 4, "ModelRec layout");

// ---- per-surface record (model::surface = mesh + material) ----
struct SurfaceRec {
	float bmin[3], bmax[3];  // mesh AABB (mesh.cpp:254-261)
	uint32_t kd_root;        // index of the root KD node in the full node array
	uint32_t tri_base;       // global id of the mesh's triangle 0
	uint32_t lds_root;       // index of the root in the LDS-resident node array, 0xFFFFFFFF when the surface is not resident
	uint32_t model;          // the model this surface belongs to
	float pbmin[3], pbmax[3];  // mesh AABB grown by the reach of the barycentric slack (see ModelRec)
	float box[6], pbox[6];     // both boxes as (min, max) pairs per axis (see ModelRec)
};
static_assert(sizeof(SurfaceRec) == 28 * 4, "SurfaceRec layout");
template <class Rec> inline void interleave_boxes(Rec& r) {
	for (int k = 0; k < 3; k++) { r.box[2 * k] = r.bmin[k]; r.box[2 * k + 1] = r.bmax[k]; r.pbox[2 * k] = r.pbmin[k]; r.pbox[2 * k + 1] = r.pbmax[k]; }
}

// ---- material factors (core/material.hpp:11-17) and texture slots ----
struct MaterialRec {
	float albedo[3]; float opacity;
	float emissive[3]; float roughness;   // emissive factor; the shader multiplies the looked-up value by 10 (renderer.cpp:462)
	float metallic; float ior; uint32_t shadow_catcher; uint32_t tex_mask;
	int32_t tex[7];   // texture id per slot: normal, albedo, opacity, occlusion, roughness, metallic, emissive; -1 = none
	int32_t pad;
};
static_assert(sizeof(MaterialRec) == 80, "MaterialRec layout");

// ---- texture: 8-bit texels exactly as the reference keeps them (image::image::data, image.cpp:124-141); the sRGB
// decode pow(v/255, 2.2) of colour channels is a 256-entry table computed on the host with the same libm call.
// A Radiance .hdr image keeps its floats (image::read returns them unscaled): c_srgb bit 16, offset counts floats in `texels_f`.
struct TexRec { uint32_t w, h, c_srgb /* channels | srgb << 8 | float texels << 16 */, offset /* first byte in the texel array (first float for float texels) */; };
constexpr uint32_t kTexSrgb = 1u << 8, kTexFloat = 1u << 16;

// ---- everything the SHADING phase needs about the surface that was hit, gathered per surface so that a
// divergent lookup is 9 aligned 16-byte reads from one place (LDS when it fits): the owning model's
// local->world transform and normal matrix, and the material. 176 B.
struct ShadeRec {
	float basis[9];   // model global basis, columns
	float origin[3];
	float nmat[9];    // transpose(inverse(basis)), columns
	float pad[3];
	MaterialRec mat;
};
static_assert(sizeof(ShadeRec) == 176 && sizeof(ShadeRec) % 16 == 0, "ShadeRec layout: staged into LDS in 16-byte units");

// ---- ray space: a distinct world->local transform. Models whose entity transforms are bitwise equal
// share one, so the local ray (origin, normalised direction, reciprocal direction) is computed once per
// space instead of once per model (model.cpp:22-29 recomputes it per model; same operations, same bits).
struct SpaceRec {
	float inv_basis[9];
	float inv_origin[3];
};

struct CameraRec { float origin[3]; float basis[9]; float fov; float tan_half_fov; };
struct SunRec { float basis[9]; float energy[3]; float angular_radius; uint32_t present; };

// ---- host-side container ----
struct FlatScene {
	// description as loaded (kept for inspection / tests)
	std::vector<std::string> model_names;
	std::vector<float> model_xform;      // [n][12]
	std::vector<int32_t> model_surf;     // [n][2]
	std::vector<int32_t> surf_range;     // [ns][8]: v0,nv,t0,nt,kd_root,n_nodes,ref0,n_refs
	std::vector<float> vertices;         // [nv][11]
	std::vector<uint32_t> triangles;     // [nt][3] mesh-local ids
	std::vector<float> materials_raw;    // [ns][11]
	std::vector<uint8_t> material_tex;   // [ns][7] texture present per slot
	std::vector<int32_t> surf_tex;       // [ns][7] texture id per slot or -1
	std::vector<TexRec> textures;
	std::vector<uint8_t> texels;
	std::vector<float> texels_f;         // texels of .hdr images
	std::vector<std::string> texture_paths;
	// derived, device-ready
	std::vector<ModelRec> models;
	std::vector<SurfaceRec> surfaces;
	std::vector<MaterialRec> materials;
	std::vector<ShadeRec> shade;         // per surface
	std::vector<SpaceRec> spaces;        // distinct world->local transforms
	std::vector<uint32_t> model_space;   // per model
	std::vector<KdNode> kd_nodes;
	std::vector<uint32_t> kd_refs;       // global triangle ids
	std::vector<TriRec> tris;            // corners + vertex ids (host side: counts, inspection)
	std::vector<TriIsect> tri_isect;     // intersection form (traversal)
	std::vector<HitRec> hitrec;          // per triangle: what a hit interpolates (shading)
	std::vector<HitRec> hot_hitrec;      // the hit records kept in LDS beside the resident geometry (largest triangles first; upload_scene)
	CameraRec camera{};
	SunRec sun{};
	uint32_t kd_max_depth = 0;
	bool any_texture = false;
	bool any_alpha = false;              // some material can take the opacity / shadow-catcher pass-through branch
	int32_t env_tex = -1;                // renderer::environment: index into `textures`, -1 = none

	// LDS residency plan (plan_residency): the traversal arrays of the surfaces that fit one CU's LDS, indices local to them
	std::vector<KdNode> res_nodes;
	std::vector<uint32_t> res_refs;      // indices into res_tris
	std::vector<TriIsect> res_tris;      // p0 = global triangle id
	uint32_t n_resident = 0;             // surfaces with SurfaceRec::lds_root valid
	size_t res_bytes = 0;                // LDS bytes the resident arrays + shade records take

	size_t geometry_bytes() const { return kd_nodes.size() * 8 + kd_refs.size() * 4 + tris.size() * 48; }
};

// Chooses which surfaces are staged into LDS (smallest first, while they fit `lds_budget` together with the shade records)
// and builds their compact arrays; sets SurfaceRec::lds_root. All surfaces resident => the compact arrays equal the full ones.
void plan_residency(FlatScene& s, size_t lds_budget);

// Builds everything derived (AABBs, KD-trees, records) from the "as loaded" arrays + camera/sun floats.
// camera13: origin(3) basis(9) fov ; sun13: basis(9) energy(3) angular_radius or nullptr.
void finalize_scene(FlatScene& s, const float* camera13, const float* sun13);

// glTF loader (throws ptx::Error)
struct Error {
	int code;
	std::string msg;
};
void read_png(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<uint8_t>& out);
void read_jpeg(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<uint8_t>& out);    // jpeg_read.cpp
bool is_hdr_file(const std::string& path);                                                                      // hdr_read.cpp
void read_hdr(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<float>& out);
void read_image(const std::string& path, uint32_t& W, uint32_t& H, uint32_t& C, std::vector<uint8_t>& out);   // PNG or JPEG, by content
// image::image::load: appends the image's texels to the scene's arrays (8-bit ones to `texels`, .hdr floats to `texels_f`) and returns its record
TexRec load_texture(FlatScene& s, const std::string& path, bool srgb);
// work: the host's `scene_info.work` primitive filter (src/models/work_info.hpp:11-15, src/scene/load_gltf.cpp:95-99):
// when `filter` is set, only the listed primitive indices of each named mesh are loaded (a mesh that is not listed
// loads nothing but keeps its model); when clear, every primitive is loaded (core::renderer::load_gltf).
struct WorkFilter {
	bool filter = false;
	std::vector<std::pair<std::string, std::vector<int32_t>>> work;
};
void load_gltf(const std::string& path, uint32_t camera_index, uint32_t sun_light_index, const WorkFilter& work, FlatScene& out);

// The Lambda event of the reference's worker (models::worker_info, src/models/work_info.hpp:17-32; sample:
// path-tracer-core/events/event.json)
struct WorkerEvent {
	WorkFilter work;
	std::string scene_bucket, scene_root, worker_id;
	int32_t num_workers = 1, samples = 0, bounces = 0;
	float X = 0, Y = 0;
};
void parse_worker_event(const std::string& json_path, WorkerEvent& out);

}  // namespace ptx
