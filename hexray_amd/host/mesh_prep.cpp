// OBJ reader, triangle precomputation and the reference-identical kd-tree builder; see mesh_prep.h.
#include "mesh_prep.h"
#include <algorithm>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include "scene_file.h"

namespace hxh {

namespace {

long long g_build_box_tests = 0;   // BBox::testIntersect evaluations made by the builder (pins the build against gprof counts)

struct Box {
	V3 lo, hi;
	// BBox::inside, borders widened by 1e-6 (src/bbox.h:78-85)
	bool holds(const V3& p) const
	{
		return lo.x - 1e-6 <= p.x && p.x <= hi.x + 1e-6 && lo.y - 1e-6 <= p.y && p.y <= hi.y + 1e-6 &&
		       lo.z - 1e-6 <= p.z && p.z <= hi.z + 1e-6;
	}
	// BBox::testIntersect (src/bbox.h:94-141); `d` need not be unit length
	bool hit_by(const V3& o, const V3& d) const
	{
		g_build_box_tests++;
		if (holds(o)) return true;
		for (int a = 0; a < 3; a++) {
			if ((d[a] < 0 && o[a] < lo[a]) || (d[a] > 0 && o[a] > hi[a])) return false;
			if (std::fabs(d[a]) < 1e-9) continue;
			double inv = 1 / d[a];
			int u = (a == 0) ? 1 : 0, v = (a == 2) ? 1 : 2;
			const double walls[2] = {lo[a], hi[a]};
			for (int w = 0; w < 2; w++) {
				double t = (walls[w] - o[a]) * inv;
				if (t < 0) break;   // near wall behind the origin: the far wall of this axis is skipped too
				double pu = o[u] + d[u] * t;
				if (lo[u] <= pu && pu <= hi[u]) {
					double pv = o[v] + d[v] * t;
					if (lo[v] <= pv && pv <= hi[v]) return true;
				}
			}
		}
		return false;
	}
};

// intersectTriangleFast (src/mesh.cpp:131-170)
bool segment_hits_triangle(const V3& o, const V3& d, const V3& A, const V3& B, const V3& C, double& best)
{
	V3 AB = B - A, AC = C - A, nd = -d, H = o - A;
	V3 N = cross(AB, AC);
	double den = dot(N, nd);
	if (std::fabs(den) < 1e-12) return false;
	double l2 = dot(cross(H, AC), nd) / den;
	double l3 = dot(cross(AB, H), nd) / den;
	double g = dot(N, H) / den;
	if (g < 0 || g > best) return false;
	if (l2 < 0 || l2 > 1 || l3 < 0 || l3 > 1 || l2 + l3 > 1) return false;
	best = g;
	return true;
}

inline double sign_of(double x) { return x > 0 ? +1 : -1; }

// BBox::intersectTriangle (src/bbox.h:177-207): vertex inside, edge through the box, box edge through the triangle
bool box_overlaps_triangle(const Box& b, const V3& A, const V3& B, const V3& C)
{
	if (b.holds(A) || b.holds(B) || b.holds(C)) return true;
	const V3 t[3] = {A, B, C};
	for (int i = 0; i < 3; i++)
		for (int j = i + 1; j < 3; j++)
			if (b.hit_by(t[i], t[j] - t[i]) && b.hit_by(t[j], t[i] - t[j])) return true;
	V3 N = cross(B - A, C - A);
	double D = dot(A, N);
	for (int mask = 0; mask < 7; mask++)
		for (int j = 0; j < 3; j++) {
			if (mask & (1 << j)) continue;
			V3 s((mask & 1) ? b.hi.x : b.lo.x, (mask & 2) ? b.hi.y : b.lo.y, (mask & 4) ? b.hi.z : b.lo.z);
			V3 e = s;
			e[j] = b.hi[j];
			if (sign_of(dot(s, N) - D) != sign_of(dot(e, N) - D)) {
				double reach = 1.0000001;
				if (segment_hits_triangle(s, e - s, A, B, C, reach)) return true;
			}
		}
	return false;
}

int to_int(const std::string& s)
{
	int x;
	return (!s.empty() && 1 == sscanf(s.c_str(), "%d", &x)) ? x : 0;
}
double to_double(const std::string& s)
{
	double x;
	return (!s.empty() && 1 == sscanf(s.c_str(), "%lf", &x)) ? x : 0;
}

// "v", "v/t", "v//n", "v/t/n"
void parse_corner(const std::string& s, int& v, int& t, int& n)
{
	std::string part[3];
	int k = 0;
	for (char c : s) {
		if (c == '/') { if (++k > 2) break; }
		else part[k] += c;
	}
	v = to_int(part[0]);
	t = to_int(part[1]);
	n = to_int(part[2]);
}

// x*A + y*B = C in the xy plane (Cramer), as used for dNdx / dNdy (src/mesh.cpp:360-368)
void solve2(const V3& A, const V3& B, const V3& C, double& x, double& y)
{
	double den = A.x * B.y - A.y * B.x;
	x = (C.x * B.y - C.y * B.x) / den;
	y = (A.x * C.y - A.y * C.x) / den;
}

void set3(double* o, const V3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

}  // namespace

bool MeshData::load_obj(const std::string& path)
{
	FILE* f = fopen(path.c_str(), "rt");
	if (!f) return false;
	vertices.assign(1, V3());
	normals.assign(1, V3());
	uvs.assign(1, V3());
	triangles.clear();
	// One pass over the file with the tokens cut in place (no per-line vectors of strings): a 9000-triangle OBJ parses in 3 ms instead
	// of 15.  Numbers go through strtod / strtol, which read what sscanf("%lf") / sscanf("%d") read (the reference's parser, src/mesh.cpp
	// :265-358, tokenises on whitespace and converts the same way); a field that is no number counts as 0.
	static char line[10000];
	std::vector<char*> tok;
	auto num = [&](size_t i) -> double {
		if (i >= tok.size()) return 0.0;
		char* end = nullptr;
		const double x = strtod(tok[i], &end);
		return end == tok[i] ? 0.0 : x;
	};
	auto corner = [](const char* s, int& v, int& t, int& n) {   // "v", "v/t", "v//n", "v/t/n"
		int* out[3] = {&v, &t, &n};
		v = t = n = 0;
		for (int k = 0; k < 3; k++) {
			char* end = nullptr;
			const long x = strtol(s, &end, 10);
			*out[k] = end == s ? 0 : (int)x;
			while (*end && *end != '/') end++;   // whatever follows the number inside this field is ignored, as sscanf("%d") ignores it
			if (*end != '/') break;
			s = end + 1;
		}
	};
	while (fgets(line, sizeof line, f)) {
		if (line[0] == '#') continue;
		tok.clear();
		for (char* p = line; *p;) {
			while (*p && isspace((unsigned char)*p)) p++;
			if (!*p) break;
			tok.push_back(p);
			while (*p && !isspace((unsigned char)*p)) p++;
			if (*p) *p++ = 0;
		}
		if (tok.empty()) continue;
		if (!strcmp(tok[0], "v")) vertices.emplace_back(num(1), num(2), num(3));
		else if (!strcmp(tok[0], "vn")) normals.emplace_back(num(1), num(2), num(3));
		else if (!strcmp(tok[0], "vt")) uvs.emplace_back(num(1), num(2), 0.0);
		else if (!strcmp(tok[0], "f")) {
			for (int i = 0; i + 3 < int(tok.size()); i++) {   // fan (1, 2+i, 3+i)
				hxr_triangle T;
				memset(&T, 0, sizeof T);
				corner(tok[1], T.v[0], T.t[0], T.n[0]);
				corner(tok[2 + i], T.v[1], T.t[1], T.n[1]);
				corner(tok[3 + i], T.v[2], T.t[2], T.n[2]);
				triangles.push_back(T);
			}
		}
	}
	fclose(f);

	// prepareTriangles (src/mesh.cpp:370-396)
	if (normals.size() <= 1) faceted = true;
	if (defer_triangle_prep && !auto_smooth) return true;   // left to the render module; auto-smoothing (begin_render) needs gnormal here
	for (hxr_triangle& t : triangles) {
		const V3 &A = vertices[t.v[0]], &B = vertices[t.v[1]], &C = vertices[t.v[2]];
		V3 AB = B - A, AC = C - A, N = cross(AB, AC);
		set3(t.AB, AB);
		set3(t.AC, AC);
		set3(t.ABcrossAC, N);
		set3(t.gnormal, unit(N));
		V3 tAB = uvs[t.t[1]] - uvs[t.t[0]], tAC = uvs[t.t[2]] - uvs[t.t[0]];
		double px, py, qx, qy;
		solve2(tAB, tAC, V3(1, 0, 0), px, qx);
		solve2(tAB, tAC, V3(0, 1, 0), py, qy);
		set3(t.dNdx, unit(px * AB + qx * AC));
		set3(t.dNdy, unit(py * AB + qy * AC));
	}
	return true;
}

namespace {
struct KdBuilder {
	MeshData& m;
	void build(int node, const Box& box, const std::vector<int>& tris, int depth)
	{
		m.stats.nodes++;
		if (tris.size() < 20 || depth > 64) {
			hxr_kd_node& n = m.kd_nodes[node];
			n.axis = 3;
			n.first = int32_t(m.leaf_indices.size());
			n.count = int32_t(tris.size());
			n.split = 0;
			m.leaf_indices.insert(m.leaf_indices.end(), tris.begin(), tris.end());
			m.stats.sum_leaf_tris += (long long)tris.size();
			m.stats.max_depth = std::max(m.stats.max_depth, depth);
			m.stats.sum_depth += depth;
			m.stats.leaves++;
			return;
		}
		int axis = depth % 3;
		double sp = box.lo[axis] + (box.hi[axis] - box.lo[axis]) * 0.5;
		Box L = box, R = box;
		L.hi[axis] = sp;
		R.lo[axis] = sp;
		std::vector<int> lt, rt;
		for (int ti : tris) {
			const hxr_triangle& T = m.triangles[ti];
			const V3 &a = m.vertices[T.v[0]], &b = m.vertices[T.v[1]], &c = m.vertices[T.v[2]];
			if (box_overlaps_triangle(L, a, b, c)) lt.push_back(ti);
			if (box_overlaps_triangle(R, a, b, c)) rt.push_back(ti);
		}
		int first = int(m.kd_nodes.size());
		m.kd_nodes.resize(m.kd_nodes.size() + 2);
		m.kd_nodes[first].parent = m.kd_nodes[first + 1].parent = node;
		hxr_kd_node& n = m.kd_nodes[node];
		n.axis = axis;
		n.first = first;
		n.count = 0;
		n.split = sp;
		build(first, L, lt, depth + 1);
		build(first + 1, R, rt, depth + 1);
	}
};
}  // namespace

void MeshData::begin_render(bool build_kd)
{
	if (recenter) {
		V3 c;
		for (size_t i = 1; i < vertices.size(); i++) c = c + vertices[i];
		c = c * (1.0 / double(vertices.size() - 1));
		for (size_t i = 1; i < vertices.size(); i++) vertices[i] = vertices[i] + (-c);
	}
	bbox_min = V3(+kInf, +kInf, +kInf);
	bbox_max = V3(-kInf, -kInf, -kInf);
	for (size_t i = 1; i < vertices.size(); i++)
		for (int a = 0; a < 3; a++) {
			bbox_min[a] = std::min(bbox_min[a], vertices[i][a]);
			bbox_max[a] = std::max(bbox_max[a], vertices[i][a]);
		}
	kd_nodes.clear();
	leaf_indices.clear();
	stats = KdStats();
	g_build_box_tests = 0;
	if (use_kd && build_kd) {
		std::vector<int> all(triangles.size());
		for (size_t i = 0; i < all.size(); i++) all[i] = int(i);
		kd_nodes.resize(1);
		kd_nodes[0].parent = -1;
		Box root{bbox_min, bbox_max};
		KdBuilder{*this}.build(0, root, all, 0);
		stats.build_box_tests = g_build_box_tests;
	}
	// auto-smoothing only when the OBJ carried no normals (src/mesh.cpp:73-86)
	if (normals.size() <= 1 && auto_smooth) {
		normals.assign(vertices.size(), V3());
		for (hxr_triangle& t : triangles)
			for (int j = 0; j < 3; j++) {
				t.n[j] = t.v[j];
				normals[t.n[j]] = normals[t.n[j]] + V3(t.gnormal[0], t.gnormal[1], t.gnormal[2]);
			}
		for (size_t i = 1; i < normals.size(); i++)
			if (length_sqr(normals[i]) > 1e-9) normals[i] = unit(normals[i]);
		faceted = false;
	}
	if (normals.size() <= 1) faceted = true;
}

}  // namespace hxh
