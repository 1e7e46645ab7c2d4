// glTF 2.0 scene loader producing the flat "as loaded" arrays of ptx::FlatScene.
//
// Behavioural contract = the reference loader, core::renderer::load_gltf and friends
// (LIB/core/renderer.cpp:61-331), including the parts that change pixels:
//   * node transforms come from TRS only, a node `matrix` is ignored (renderer.cpp:113-128);
//   * an entity is named after its camera / light when it has one (renderer.cpp:106-111) and the
//     camera / sun light are found by NAME match against cameras[camera_index] / lights[sun_index];
//   * root entities live in a std::unordered_map<string, entity> (renderer.hpp:25): a repeated root
//     name replaces the earlier entity and the visit order of renderer::intersect (renderer.cpp:646-658)
//     is the container's hash order pushed on a stack — reproduced with the same container type;
//   * TANGENT (VEC4) is unpacked as count*3 floats of the packed xyzw stream and read back with
//     stride 3 (renderer.cpp:215-218,248-250; cgltf writes whole elements only) — quirk Q1;
//   * only scenes[0] is read; materials: factors + which textures are present (renderer.cpp:265-331).
// The JSON reader below is a small recursive-descent parser written for this file (the reference
// vendors cgltf; nothing of it is used here). Numbers are converted like cgltf does: (float)strtod().
#include "flat_scene.hpp"
#include "json_min.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <unordered_map>

namespace ptx {
namespace {

std::string read_file(const std::string& path, bool binary) {
	std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
	if (!f) fail(E_IO, "cannot open '" + path + "'");
	return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::string dir_of(const std::string& path) {
	size_t k = path.find_last_of('/');
	return k == std::string::npos ? std::string(".") : path.substr(0, k);
}
std::string uri_decode_spaces(std::string s) {  // renderer.cpp:40 does this for textures; buffers go through cgltf, which decodes too
	size_t k;
	while ((k = s.find("%20")) != std::string::npos) s.replace(k, 3, " ");
	return s;
}

// ------------------------------------------------------------------------------------------- accessors
struct Gltf {
	JVal root;
	std::string dir;
	std::vector<std::string> buffers;
	const std::string& buffer(size_t i) {
		if (i >= buffers.size()) fail(E_PARSE, "glTF: buffer index out of range");
		if (buffers[i].empty()) {
			const JVal& b = root.at("buffers").el(i);
			if (!b.has("uri")) fail(E_PARSE, "glTF: buffer without uri (GLB / embedded buffers are not loaded by the reference either)");
			buffers[i] = read_file(dir + "/" + uri_decode_spaces(b.at("uri").s()), true);
		}
		return buffers[i];
	}
};

int n_components(const std::string& t) {
	if (t == "SCALAR") return 1; if (t == "VEC2") return 2; if (t == "VEC3") return 3; if (t == "VEC4") return 4;
	if (t == "MAT2") return 4; if (t == "MAT3") return 9; if (t == "MAT4") return 16;
	fail(E_PARSE, "glTF: unknown accessor type " + t);
}
int component_size(int64_t ct) {
	switch (ct) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; }
	fail(E_PARSE, "glTF: unknown componentType");
}

struct AccessorView { const uint8_t* base; size_t stride; size_t count; int64_t ctype; int ncomp; bool normalized; };

AccessorView view(Gltf& g, size_t idx) {
	const JVal& a = g.root.at("accessors").el(idx);
	if (!a.has("bufferView")) fail(E_PARSE, "glTF: accessor without bufferView (sparse accessors unsupported)");
	const JVal& bv = g.root.at("bufferViews").el((size_t)a.at("bufferView").i());
	AccessorView v;
	v.ctype = a.at("componentType").i();
	v.ncomp = n_components(a.at("type").s());
	// every number below comes from the file as a signed integer: negative or absurd values must not wrap through size_t
	const int64_t count_i = a.at("count").i();
	const int64_t bv_off = bv.has("byteOffset") ? bv.at("byteOffset").i() : 0, a_off = a.has("byteOffset") ? a.at("byteOffset").i() : 0;
	const int64_t stride_i = bv.has("byteStride") ? bv.at("byteStride").i() : 0, buf_i = bv.at("buffer").i();
	if (count_i < 0 || bv_off < 0 || a_off < 0 || stride_i < 0 || buf_i < 0) fail(E_PARSE, "glTF: negative count / byteOffset / byteStride / buffer index");
	v.count = (size_t)count_i;
	v.normalized = a.has("normalized") && a.at("normalized").b;
	const size_t elem = (size_t)component_size(v.ctype) * v.ncomp;
	v.stride = stride_i > 0 ? (size_t)stride_i : elem;
	const std::string& buf = g.buffer((size_t)buf_i);
	// off + (count - 1) * stride + elem <= size, evaluated without overflow
	const size_t size = buf.size();
	if ((uint64_t)bv_off > size || (uint64_t)a_off > size - (size_t)bv_off) fail(E_PARSE, "glTF: accessor exceeds its buffer");
	const size_t off = (size_t)bv_off + (size_t)a_off;
	if (v.count && (size - off < elem || (v.count - 1) > (size - off - elem) / v.stride)) fail(E_PARSE, "glTF: accessor exceeds its buffer");
	v.base = (const uint8_t*)buf.data() + off;
	return v;
}

float component_as_float(const uint8_t* p, int64_t ctype, bool normalized) {
	switch (ctype) {
	case 5126: { float f; memcpy(&f, p, 4); return f; }
	case 5125: { uint32_t u; memcpy(&u, p, 4); return (float)u; }
	case 5123: { uint16_t u; memcpy(&u, p, 2); return normalized ? u / 65535.0f : (float)u; }
	case 5122: { int16_t i; memcpy(&i, p, 2); float f = normalized ? i / 32767.0f : (float)i; return normalized && f < -1 ? -1.0f : f; }
	case 5121: { uint8_t u = *p; return normalized ? u / 255.0f : (float)u; }
	default: { int8_t i = (int8_t)*p; float f = normalized ? i / 127.0f : (float)i; return normalized && f < -1 ? -1.0f : f; }
	}
}

// Semantics of cgltf_accessor_unpack_floats(accessor, out, float_count): whole elements only.
std::vector<float> unpack_floats(Gltf& g, size_t idx, size_t float_count) {
	AccessorView v = view(g, idx);
	size_t avail = v.count * v.ncomp;
	float_count = std::min(avail, float_count);
	size_t n_el = float_count / v.ncomp;
	std::vector<float> out(n_el * v.ncomp);
	int cs = component_size(v.ctype);
	for (size_t e = 0; e < n_el; e++)
		for (int c = 0; c < v.ncomp; c++) out[e * v.ncomp + c] = component_as_float(v.base + e * v.stride + (size_t)c * cs, v.ctype, v.normalized);
	return out;
}
std::vector<uint32_t> unpack_indices(Gltf& g, size_t idx) {
	AccessorView v = view(g, idx);
	std::vector<uint32_t> out(v.count);
	for (size_t e = 0; e < v.count; e++) {
		const uint8_t* p = v.base + e * v.stride;
		switch (v.ctype) {
		case 5121: out[e] = *p; break;
		case 5123: { uint16_t u; memcpy(&u, p, 2); out[e] = u; break; }
		case 5125: { uint32_t u; memcpy(&u, p, 4); out[e] = u; break; }
		default: fail(E_PARSE, "glTF: index accessor must be unsigned");
		}
	}
	return out;
}

// ------------------------------------------------------------------------------------------- entities
struct Xf { float o[3]; float b[9]; };  // origin + basis columns

Xf compose(const Xf& p, const Xf& c) {  // transform::operator*, LIB/scene/transform.cpp:110-115
	Xf r;
	for (int k = 0; k < 3; k++) r.o[k] = (p.b[k] * c.o[0] + p.b[3 + k] * c.o[1] + p.b[6 + k] * c.o[2]) + p.o[k];
	for (int col = 0; col < 3; col++)   // mat3*mat3: x*rhs.col.x + y*rhs.col.y + z*rhs.col.z (mat3.inl:144-152)
		for (int k = 0; k < 3; k++)
			r.b[3 * col + k] = p.b[k] * c.b[3 * col] + p.b[3 + k] * c.b[3 * col + 1] + p.b[6 + k] * c.b[3 * col + 2];
	return r;
}

struct Prim { std::vector<float> verts; std::vector<uint32_t> tris; float mat[11]; int32_t tex[7]; };
struct Entity {
	std::string name;
	Xf local;
	Entity* parent = nullptr;
	std::vector<Entity*> children;
	bool is_model = false;
	std::vector<Prim> prims;
};

struct Loader {
	Gltf g;
	std::vector<std::unique_ptr<Entity>> pool;
	std::string camera_name, sun_name;
	bool want_sun = false;
	Entity* camera = nullptr;
	Entity* sun = nullptr;
	const JVal* lights = nullptr;
	const WorkFilter* work = nullptr;
	// get_cached_texture (renderer.cpp:33-51): one texture object per file path; the sRGB flag of the FIRST request sticks
	std::unordered_map<std::string, int32_t> tex_by_path;
	FlatScene tex_store;   // textures / texels / texels_f / texture_paths as load_texture fills them

	int32_t texture(const JVal* ref, bool srgb) {
		if (!ref) return -1;
		const JVal& tex = g.root.at("textures").el((size_t)ref->at("index").i());
		const JVal& img = g.root.at("images").el((size_t)tex.at("source").i());
		if (!img.has("uri")) fail(E_PARSE, "glTF: image without uri (buffer-view images are not loaded by the reference either)");
		std::string path = g.dir + "/" + uri_decode_spaces(img.at("uri").s());
		auto it = tex_by_path.find(path);
		if (it != tex_by_path.end()) return it->second;
		const TexRec t = load_texture(tex_store, path, srgb);   // PNG, JPEG or Radiance HDR, by content (image::image::load -> stb_image)
		const int32_t id = (int32_t)tex_store.textures.size();
		tex_store.textures.push_back(t);
		tex_store.texture_paths.push_back(path);
		tex_by_path[path] = id;
		return id;
	}

	Prim load_prim(const JVal& p) {
		Prim out{};
		std::vector<float> pos, uv, nrm, tan;
		const JVal& attrs = p.at("attributes");
		for (auto& kv : attrs.obj) {  // JSON order: a later TEXCOORD_n overwrites an earlier one (renderer.cpp:205-208)
			size_t acc = (size_t)kv.second.i();
			size_t cnt = (size_t)g.root.at("accessors").el(acc).at("count").i();
			const std::string& nm = kv.first;
			if (nm == "POSITION") pos = unpack_floats(g, acc, cnt * 3);
			else if (nm.rfind("TEXCOORD", 0) == 0) uv = unpack_floats(g, acc, cnt * 2);
			else if (nm == "NORMAL") nrm = unpack_floats(g, acc, cnt * 3);
			else if (nm == "TANGENT") { tan = unpack_floats(g, acc, cnt * 3); tan.resize(cnt * 3, 0.0f); }  // Q1
		}
		if (!p.has("indices")) fail(E_PARSE, "glTF: non-indexed primitive (the reference dereferences a null accessor here)");
		std::vector<uint32_t> idx = unpack_indices(g, (size_t)p.at("indices").i());
		size_t nv = pos.size() / 3, nt = idx.size() / 3;
		out.verts.assign(nv * 11, 0.0f);
		for (size_t k = 0; k < nv; k++) {
			float* v = &out.verts[11 * k];
			memcpy(v, &pos[3 * k], 12);
			if (uv.size() >= 2 * (k + 1)) memcpy(v + 3, &uv[2 * k], 8);
			if (nrm.size() >= 3 * (k + 1)) memcpy(v + 5, &nrm[3 * k], 12);
			if (tan.size() >= 3 * (k + 1)) memcpy(v + 8, &tan[3 * k], 12);
		}
		out.tris.assign(idx.begin(), idx.begin() + nt * 3);
		for (uint32_t i : out.tris) if (i >= nv) fail(E_PARSE, "glTF: vertex index out of range");

		// material — renderer.cpp:265-331; defaults of core::material (material.hpp:11-17) when absent
		float m[11] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1.33F, 0};
		int32_t tx[7] = {-1, -1, -1, -1, -1, -1, -1};
		if (p.has("material")) {
			const JVal& mat = g.root.at("materials").el((size_t)p.at("material").i());
			float bc[4] = {1, 1, 1, 1}, em[3] = {0, 0, 0}, rough = 1, metal = 1;
			const JVal* pbr = mat.find("pbrMetallicRoughness");
			if (pbr) {
				if (const JVal* f = pbr->find("baseColorFactor")) for (int k = 0; k < 4; k++) bc[k] = f->el(k).f();
				if (const JVal* f = pbr->find("roughnessFactor")) rough = f->f();
				if (const JVal* f = pbr->find("metallicFactor")) metal = f->f();
			}
			if (const JVal* f = mat.find("emissiveFactor")) for (int k = 0; k < 3; k++) em[k] = f->el(k).f();
			std::string name = mat.has("name") ? mat.at("name").s() : "";
			bool opaque = !mat.has("alphaMode") || mat.at("alphaMode").s() == "OPAQUE";
			float mm[11] = {bc[0], bc[1], bc[2], bc[3], rough, metal, em[0], em[1], em[2], 1.33F,
			                (name.find("shadow") != std::string::npos && name.find("catcher") != std::string::npos) ? 1.0f : 0.0f};
			memcpy(m, mm, sizeof m);
			// texture loads in the order of renderer.cpp:297-324: normal, base colour (also opacity unless OPAQUE), occlusion,
			// metallic-roughness (G = roughness, B = metallic), emissive; base colour and emissive are sRGB
			const int32_t t_n = texture(mat.find("normalTexture"), false);
			const int32_t t_a = texture(pbr ? pbr->find("baseColorTexture") : nullptr, true);
			const int32_t t_oc = texture(mat.find("occlusionTexture"), false);
			const int32_t t_mr = texture(pbr ? pbr->find("metallicRoughnessTexture") : nullptr, false);
			const int32_t t_e = texture(mat.find("emissiveTexture"), true);
			const int32_t tt[7] = {t_n, t_a, (t_a >= 0 && !opaque) ? t_a : -1, t_oc, t_mr, t_mr, t_e};
			memcpy(tx, tt, sizeof tx);
		}
		memcpy(out.mat, m, sizeof m);
		memcpy(out.tex, tx, sizeof tx);
		return out;
	}

	Entity* process_node(size_t ni, Entity* parent) {  // renderer.cpp:101-174
		// a node hierarchy with a cycle would recurse forever in the reference; here it is a parse error
		int depth = 0;
		for (const Entity* a = parent; a; a = a->parent) depth++;
		if (depth > 512) fail(E_PARSE, "glTF: node hierarchy deeper than 512 levels (cycle?)");
		const JVal& n = g.root.at("nodes").el(ni);
		pool.emplace_back(new Entity);
		Entity* e = pool.back().get();
		e->parent = parent;
		const JVal* light_ref = nullptr;
		if (const JVal* ext = n.find("extensions"))
			if (const JVal* kl = ext->find("KHR_lights_punctual")) light_ref = kl->find("light");
		auto name_of = [](const JVal& o) { return o.has("name") ? o.at("name").s() : std::string(); };
		if (n.has("camera")) e->name = name_of(g.root.at("cameras").el((size_t)n.at("camera").i()));
		else if (light_ref && lights) e->name = name_of(lights->el((size_t)light_ref->i()));
		else e->name = name_of(n);

		float q[4] = {0, 0, 0, 0};  // w, x, y, z — math::quat() is all-zero, which to_basis() maps to identity
		if (const JVal* r = n.find("rotation")) { q[0] = r->el(3).f(); q[1] = r->el(0).f(); q[2] = r->el(1).f(); q[3] = r->el(2).f(); }
		float sc[3] = {1, 1, 1}, tr[3] = {0, 0, 0};
		if (const JVal* s = n.find("scale")) for (int k = 0; k < 3; k++) sc[k] = s->el(k).f();
		if (const JVal* t = n.find("translation")) for (int k = 0; k < 3; k++) tr[k] = t->el(k).f();
		const float w = q[0], x = q[1], y = q[2], z = q[3];
		float b[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),      // quat::to_basis, quat.cpp:95-113
		              2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),
		              2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};
		for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) e->local.b[3 * c + k] = b[3 * c + k] * sc[c];  // make_basis
		memcpy(e->local.o, tr, 12);

		if (n.has("mesh")) {
			e->is_model = true;
			const JVal& mesh = g.root.at("meshes").el((size_t)n.at("mesh").i());
			const JVal& prims = mesh.at("primitives");
			const std::vector<int32_t>* listed = nullptr;   // scene_work[mesh->name] (src/scene/load_gltf.cpp:93-99)
			static const std::vector<int32_t> none;
			if (work && work->filter) {
				listed = &none;
				const std::string mname = mesh.has("name") ? mesh.at("name").s() : std::string();
				for (auto& kv : work->work) if (kv.first == mname) listed = &kv.second;
			}
			for (size_t k = 0; k < prims.size(); k++) {
				if (listed && std::find(listed->begin(), listed->end(), (int32_t)k) == listed->end()) continue;
				e->prims.push_back(load_prim(prims.el(k)));
			}
		}
		if (e->name == camera_name) camera = e;            // pre-order; the last match wins (renderer.cpp:145-152)
		if (want_sun && e->name == sun_name) sun = e;
		if (const JVal* ch = n.find("children"))
			for (size_t k = 0; k < ch->size(); k++) {
				Entity* c = process_node((size_t)ch->el(k).i(), e);
				c->parent = e;
				e->children.push_back(c);
			}
		(void)parent;
		return e;
	}

	static Xf global_of(const Entity* e) {  // entity::get_global_transform, LIB/scene/entity.cpp:72-85
		return e->parent ? compose(global_of(e->parent), e->local) : e->local;
	}
};

}  // namespace

void load_gltf(const std::string& path, uint32_t camera_index, uint32_t sun_light_index, const WorkFilter& work, FlatScene& out) {
	Loader L;
	L.work = &work;
	L.g.root = JsonReader(read_file(path, false)).parse();
	L.g.dir = dir_of(path);
	L.g.buffers.resize(L.g.root.has("buffers") ? L.g.root.at("buffers").size() : 0);

	const JVal* cams = L.g.root.find("cameras");
	if (!cams || cams->size() < (size_t)camera_index + 1)
		fail(E_NO_CAMERA, "Scene does not contain camera #" + std::to_string(camera_index) + ".");
	const JVal& cam = cams->el(camera_index);
	L.camera_name = cam.has("name") ? cam.at("name").s() : std::string();
	const JVal* sun_def = nullptr;
	if (const JVal* ext = L.g.root.find("extensions"))
		if (const JVal* kl = ext->find("KHR_lights_punctual")) L.lights = kl->find("lights");
	if (sun_light_index != 0xFFFFFFFFu && L.lights && L.lights->size() >= (size_t)sun_light_index + 1) {
		const JVal& l = L.lights->el(sun_light_index);
		if (l.has("type") && l.at("type").s() == "directional") {  // renderer.cpp:84-90
			sun_def = &l;
			L.want_sun = true;
			L.sun_name = l.has("name") ? l.at("name").s() : std::string();
		}
	}

	const JVal& scene_nodes = L.g.root.at("scenes").el(0).at("nodes");  // data->scenes[0], renderer.cpp:71
	std::vector<Entity*> roots;
	for (size_t k = 0; k < scene_nodes.size(); k++) roots.push_back(L.process_node((size_t)scene_nodes.el(k).i(), nullptr));
	if (!L.camera) fail(E_NO_CAMERA, "Scene is missing a camera.");

	// entities[name] = entity (renderer.cpp:171), then the DFS of renderer::intersect (renderer.cpp:646-671)
	std::unordered_map<std::string, Entity*> by_name;
	for (Entity* e : roots) by_name[e->name] = e;
	std::vector<Entity*> stack, visit;
	for (auto& kv : by_name) stack.push_back(kv.second);
	while (!stack.empty()) {
		Entity* e = stack.back();
		stack.pop_back();
		for (Entity* c : e->children) stack.push_back(c);
		if (e->is_model) visit.push_back(e);
	}

	out = FlatScene{};
	int32_t ns = 0, nv = 0, nt = 0;
	for (Entity* e : visit) {
		Xf gx = Loader::global_of(e);
		out.model_names.push_back(e->name);
		out.model_xform.insert(out.model_xform.end(), gx.o, gx.o + 3);
		out.model_xform.insert(out.model_xform.end(), gx.b, gx.b + 9);
		out.model_surf.push_back(ns);
		out.model_surf.push_back((int32_t)e->prims.size());
		for (Prim& p : e->prims) {
			int32_t pv = (int32_t)(p.verts.size() / 11), pt = (int32_t)(p.tris.size() / 3);
			int32_t rg[8] = {nv, pv, nt, pt, 0, 0, 0, 0};
			out.surf_range.insert(out.surf_range.end(), rg, rg + 8);
			out.vertices.insert(out.vertices.end(), p.verts.begin(), p.verts.end());
			out.triangles.insert(out.triangles.end(), p.tris.begin(), p.tris.end());
			out.materials_raw.insert(out.materials_raw.end(), p.mat, p.mat + 11);
			for (int k = 0; k < 7; k++) { out.surf_tex.push_back(p.tex[k]); out.material_tex.push_back(p.tex[k] >= 0 ? 1 : 0); }
			nv += pv; nt += pt; ns++;
		}
	}
	Xf cx = Loader::global_of(L.camera);
	float cam13[13];
	memcpy(cam13, cx.o, 12);
	memcpy(cam13 + 3, cx.b, 36);
	cam13[12] = cam.at("perspective").at("yfov").f();
	float sun13[13];
	bool have_sun = L.sun != nullptr && sun_def != nullptr;
	if (have_sun) {
		Xf sx = Loader::global_of(L.sun);
		memcpy(sun13, sx.b, 36);
		float col[3] = {1, 1, 1}, inten = 1;
		if (const JVal* c = sun_def->find("color")) for (int k = 0; k < 3; k++) col[k] = c->el(k).f();
		if (const JVal* i = sun_def->find("intensity")) inten = i->f();
		for (int k = 0; k < 3; k++) sun13[9 + k] = col[k] * inten;  // renderer.cpp:159
		sun13[12] = 0.004732f;                                      // sun_light::angular_radius, sun_light.hpp:10
	}
	out.textures = std::move(L.tex_store.textures);
	out.texels = std::move(L.tex_store.texels);
	out.texels_f = std::move(L.tex_store.texels_f);
	out.texture_paths = std::move(L.tex_store.texture_paths);
	finalize_scene(out, cam13, have_sun ? sun13 : nullptr);
}

}  // namespace ptx

namespace ptx {
void parse_worker_event(const std::string& json_path, WorkerEvent& out) {
	JVal root = JsonReader(read_file(json_path, false)).parse();
	out = WorkerEvent{};
	out.work.filter = true;
	if (const JVal* si = root.find("scene_info"))
		if (const JVal* w = si->find("work"))
			for (auto& kv : w->obj) {
				std::vector<int32_t> prims;
				for (size_t k = 0; k < kv.second.size(); k++) prims.push_back((int32_t)kv.second.el(k).i());
				out.work.work.emplace_back(kv.first, std::move(prims));
			}
	auto str = [&](const char* k) { const JVal* v = root.find(k); return v && v->kind == JVal::Str ? v->str : std::string(); };
	out.scene_bucket = str("scene_bucket");
	out.scene_root = str("scene_root");
	out.worker_id = str("worker_id");
	auto num = [&](const char* k) -> double {
		const JVal* v = root.find(k);
		if (!v || v->kind != JVal::Num) fail(E_PARSE, std::string("worker event: missing number '") + k + "'");
		return v->num;
	};
	out.num_workers = (int32_t)num("num_workers");
	out.samples = (int32_t)num("samples");
	out.bounces = (int32_t)num("bounces");
	out.X = (float)num("X");
	out.Y = (float)num("Y");
}
}  // namespace ptx
