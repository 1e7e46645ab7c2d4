// Measurement (CPU): how many node steps and triangle tests a ray's walks would make if every walk of a (ray, model) group knew a bound
// on the nearest hit — (a) no bound (what the reference does), (b) the FINAL nearest hit of the group (ideal: two-phase), (c) surfaces
// walked one after another in surface order, each knowing what the earlier ones found, (d) as (c) but in order of box entry.
//   g++ -O2 -shared -fPIC -o tools/bin/libprune_potential.so tools/prune_potential.cpp ; driven by tools/prune_potential.py
#include "../oracle/pt_oracle.cpp"
#include <algorithm>
namespace {
struct work { uint64_t nodes = 0, tris = 0, walks = 0; };
// mesh_intersect (oracle) with a cut: subtrees whose range starts beyond `cut` are not set aside, a box entered beyond it is not walked
static float walk(const mesh& m, const ray& r, float cut, work& w) {
	aabb_hit bh = aabb_intersect(m.box, r);
	if (!bh.hit || bh.nr > cut) return -1;
	w.walks++;
	struct ent { int node; float mn, mx; };
	ent stack[64]; int sp = 0;
	stack[sp++] = {0, bh.nr, bh.fr};
	while (sp > 0) {
		ent e = stack[--sp];
		int node = e.node; float min_dist = e.mn, max_dist = e.mx;
		while (node >= 0 && !m.nodes[node].leaf) {
			const kd_node& b = m.nodes[node];
			w.nodes++;
			float o = get(r.o, b.axis), d = get(r.d, b.axis);
			float split_dist = (b.split - o) / d;
			int first, second;
			if (o < b.split) { first = b.left; second = b.right; } else { first = b.right; second = b.left; }
			if (split_dist < 0 || split_dist > max_dist) node = first;
			else if (split_dist < min_dist) node = second;
			else { if (second >= 0 && !(split_dist > cut) && sp < 64) stack[sp++] = {second, split_dist, max_dist}; node = first; max_dist = split_dist; }
		}
		if (node < 0) continue;
		const kd_node& leaf = m.nodes[node];
		float nearest = -1;
		for (int i = 0; i < leaf.count; i++) {
			uint32_t ti = m.refs[leaf.first + i];
			w.tris++;
			tri_hit h = tri_intersect(m.verts[m.tris[3 * ti]].pos, m.verts[m.tris[3 * ti + 1]].pos, m.verts[m.tris[3 * ti + 2]].pos, r);
			if (h.t >= 0 && h.t <= max_dist && (h.t < nearest || !(nearest >= 0))) nearest = h.t;
		}
		if (nearest >= 0) return nearest;
	}
	return -1;
}
}
// out[4][3]: nodes, tris, walks for the four policies; returns rays whose nearest hit differs between (a) and any other policy
extern "C" uint64_t prune_potential(void* p, size_t n, const float* rays, uint64_t* out) {
	const scene_t& s = *(scene_t*)p;
	work W[4]; uint64_t diff = 0;
	const float inf = std::numeric_limits<float>::infinity();
	for (size_t i = 0; i < n; i++) {
		const float* q = rays + 6 * i;
		ray r{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
		for (const model& md : s.models) {
			ray view = xray(r, md.inv);
			if (!aabb_intersect(md.box, view).hit) continue;
			float best = -1;
			for (int k = 0; k < md.n_surfaces; k++) { float t = walk(s.surfaces[md.first_surface + k].m, view, inf, W[0]); if (t >= 0 && (t < best || !(best >= 0))) best = t; }
			const float cut = best >= 0 ? best : inf;
			float b1 = -1;
			for (int k = 0; k < md.n_surfaces; k++) { float t = walk(s.surfaces[md.first_surface + k].m, view, cut, W[1]); if (t >= 0 && (t < b1 || !(b1 >= 0))) b1 = t; }
			float b2 = -1;
			for (int k = 0; k < md.n_surfaces; k++) { float t = walk(s.surfaces[md.first_surface + k].m, view, b2 >= 0 ? b2 : inf, W[2]); if (t >= 0 && (t < b2 || !(b2 >= 0))) b2 = t; }
			std::vector<std::pair<float, int>> ord;
			for (int k = 0; k < md.n_surfaces; k++) { aabb_hit bh = aabb_intersect(s.surfaces[md.first_surface + k].m.box, view); if (bh.hit) ord.push_back({bh.nr, k}); }
			std::sort(ord.begin(), ord.end());
			float b3 = -1;
			for (auto& e : ord) { float t = walk(s.surfaces[md.first_surface + e.second].m, view, b3 >= 0 ? b3 : inf, W[3]); if (t >= 0 && (t < b3 || !(b3 >= 0))) b3 = t; }
			if (!(b1 == best && b2 == best && b3 == best)) diff++;
		}
	}
	for (int k = 0; k < 4; k++) { out[3 * k] = W[k].nodes; out[3 * k + 1] = W[k].tris; out[3 * k + 2] = W[k].walks; }
	return diff;
}
// per surface: out[u][5] = walks, node steps, triangle tests, steps of the longest walk, KD nodes of the tree
extern "C" void surface_work(void* p, size_t n, const float* rays, uint64_t* out) {
	const scene_t& s = *(scene_t*)p;
	const float inf = std::numeric_limits<float>::infinity();
	for (size_t i = 0; i < n; i++) {
		const float* q = rays + 6 * i;
		ray r{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
		for (const model& md : s.models) {
			ray view = xray(r, md.inv);
			if (!aabb_intersect(md.box, view).hit) continue;
			for (int k = 0; k < md.n_surfaces; k++) {
				const int u = md.first_surface + k;
				work w;
				walk(s.surfaces[u].m, view, inf, w);
				out[5 * u] += w.walks; out[5 * u + 1] += w.nodes; out[5 * u + 2] += w.tris;
				out[5 * u + 3] = std::max<uint64_t>(out[5 * u + 3], w.nodes + w.tris);
				out[5 * u + 4] = s.surfaces[u].m.nodes.size();
			}
		}
	}
}
