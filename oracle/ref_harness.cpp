// TEST INFRASTRUCTURE -- generator of golden vectors from the REAL reference code.
//
// This program is linked against the reference's own translation units, compiled where they lie
// (/root/reference/src/{matrix,util,geometry,lights}.cpp -- see oracle/Makefile) and includes the
// reference's own headers; nothing of the reference is copied into this repository.  Only translation
// units whose include closure exists in this image are used: main.cpp, mesh.cpp, heightfield.cpp,
// camera.cpp, scene.cpp, bitmap.cpp and sdl.cpp need SDL2 / OpenEXR headers the image lacks and are
// therefore NOT built ("Oracle").
//
// The harness drives the reference classes through their public interface:
//   * elements are configured through fillProperties(ParsedBlock&) with a canned ParsedBlock (the
//     reference's own plugin interface, src/scene.h:137-185);
//   * SceneElement's five trivial members live in scene.cpp (src/scene.cpp:53-59), which cannot be
//     compiled here, so they are defined below with the same (empty) bodies -- they take no part in any
//     computed value.
// Output: JSON on stdout, every double printed with 17 significant digits (round-trip exact).
// tests/golden/make_golden.sh runs it and commits the result as tests/golden/reference_vectors.json.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "bbox.h"
#include "color.h"
#include "geometry.h"
#include "lights.h"
#include "matrix.h"
#include "node.h"
#include "util.h"

// ---- SceneElement members normally provided by scene.cpp (src/scene.cpp:53-59)
SceneElement::SceneElement() { name[0] = 0; }
void SceneElement::beginRender() {}
void SceneElement::beginFrame() {}
void SceneElement::fillProperties(ParsedBlock&) {}

// ---- a ParsedBlock that serves canned values
struct CannedBlock : ParsedBlock {
	std::map<std::string, double> numbers;
	std::map<std::string, Vector> vectors;
	std::map<std::string, Geometry*> geometries;
	std::vector<std::pair<std::string, Vector>> transform_ops;   // scale / rotate / translate in order

	bool getIntProp(const char* name, int* value, int, int) override
	{
		auto it = numbers.find(name);
		if (it == numbers.end()) return false;
		*value = int(it->second);
		return true;
	}
	bool getBoolProp(const char*, bool*) override { return false; }
	bool getFloatProp(const char* name, float* value, float, float) override
	{
		auto it = numbers.find(name);
		if (it == numbers.end()) return false;
		*value = float(it->second);
		return true;
	}
	bool getDoubleProp(const char* name, double* value, double, double) override
	{
		auto it = numbers.find(name);
		if (it == numbers.end()) return false;
		*value = it->second;
		return true;
	}
	bool getColorProp(const char* name, Color* value, float, float) override
	{
		auto it = vectors.find(name);
		if (it == vectors.end()) return false;
		*value = Color(float(it->second.x), float(it->second.y), float(it->second.z));
		return true;
	}
	bool getVectorProp(const char* name, Vector* value) override
	{
		auto it = vectors.find(name);
		if (it == vectors.end()) return false;
		*value = it->second;
		return true;
	}
	bool getGeometryProp(const char* name, Geometry** value) override
	{
		auto it = geometries.find(name);
		if (it == geometries.end()) return false;
		*value = it->second;
		return true;
	}
	bool getIntersectableProp(const char*, Intersectable**) override { return false; }
	bool getShaderProp(const char*, Shader**) override { return false; }
	bool getTextureProp(const char*, Texture**) override { return false; }
	bool getNodeProp(const char*, Node**) override { return false; }
	bool getStringProp(const char*, char*) override { return false; }
	bool getFilenameProp(const char*, char*) override { return false; }
	bool getBitmapFileProp(const char*, Bitmap&) override { return false; }
	void getTransformProp(Transform& T) override
	{
		for (auto& op : transform_ops) {
			if (op.first == "scale") T.scale(op.second.x, op.second.y, op.second.z);
			else if (op.first == "rotate") T.rotate(op.second.x, op.second.y, op.second.z);
			else T.translate(op.second);
		}
	}
	void requiredProp(const char*) override {}
	void signalError(const char*) override {}
	void signalWarning(const char*) override {}
	int getBlockLines() override { return 0; }
	void getBlockLine(int, int&, char[], char[]) override {}
	SceneParser& getParser() override { return *(SceneParser*)nullptr; }
};

// ---- deterministic test inputs (harness-local generator; not the reference's)
static std::mt19937_64 input_rng(20261004);
static double uni(double lo, double hi) { return lo + (hi - lo) * ((input_rng() >> 11) * (1.0 / 9007199254740992.0)); }
static Vector rand_vec(double r) { return Vector(uni(-r, r), uni(-r, r), uni(-r, r)); }
static Vector rand_dir()
{
	Vector v;
	do v = rand_vec(1); while (v.lengthSqr() < 1e-3 || v.lengthSqr() > 1);
	v.normalize();
	return v;
}

// ---- JSON helpers
static bool first_in_scope = true;
static void sep() { if (!first_in_scope) printf(",\n"); first_in_scope = false; }
static void open_list(const char* key) { sep(); printf("\"%s\": [\n", key); first_in_scope = true; }
static void close_list() { printf("\n]"); first_in_scope = false; }
static std::string num(double v)
{
	char b[64];
	snprintf(b, sizeof b, "%.17g", v);
	return b;
}
static std::string vec(const Vector& v) { return "[" + num(v.x) + "," + num(v.y) + "," + num(v.z) + "]"; }
static std::string mat(const Matrix& m)
{
	std::string s = "[";
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) s += num(m.m[i][j]) + ((i == 2 && j == 2) ? "" : ",");
	return s + "]";
}
static std::string ops_json(const std::vector<std::pair<std::string, Vector>>& ops)
{
	std::string s = "[";
	for (size_t i = 0; i < ops.size(); i++) s += std::string(i ? "," : "") + "[\"" + ops[i].first + "\"," + vec(ops[i].second) + "]";
	return s + "]";
}
static std::vector<std::pair<std::string, Vector>> random_ops()
{
	std::vector<std::pair<std::string, Vector>> ops;
	int n = 1 + int(uni(0, 4));
	for (int i = 0; i < n; i++) {
		int kind = int(uni(0, 3));
		if (kind == 0) ops.push_back({"scale", Vector(uni(0.3, 4), uni(0.3, 4), uni(0.3, 4))});
		else if (kind == 1) ops.push_back({"rotate", Vector(uni(-180, 180), uni(-90, 90), uni(-180, 180))});
		else ops.push_back({"translate", rand_vec(50)});
	}
	return ops;
}

static void emit_hit(bool ok, const IntersectionInfo& info)
{
	printf("\"hit\":%d", ok ? 1 : 0);
	if (ok) printf(",\"dist\":%s,\"ip\":%s,\"norm\":%s,\"u\":%s,\"v\":%s", num(info.dist).c_str(), vec(info.ip).c_str(), vec(info.norm).c_str(), num(info.u).c_str(), num(info.v).c_str());
}

int main()
{
	printf("{\n");
	printf("\"_comment\": \"golden vectors produced by oracle/ref_harness.cpp from the reference's own translation units\"");
	first_in_scope = false;

	// 1. Transform composition (src/matrix.cpp:119-152) and use (:156-187)
	open_list("transforms");
	for (int i = 0; i < 64; i++) {
		auto ops = random_ops();
		Transform T;
		CannedBlock pb;
		pb.transform_ops = ops;
		pb.getTransformProp(T);
		Vector p = rand_vec(20), d = rand_dir();
		sep();
		printf("{\"ops\":%s,\"offset\":%s,\"m\":%s,\"inv\":%s,\"tinv\":%s,\"p\":%s,\"d\":%s,\"point\":%s,\"unpoint\":%s,\"dir\":%s,\"undir\":%s,\"normal\":%s}",
		       ops_json(ops).c_str(), vec(T.offset).c_str(), mat(T.m).c_str(), mat(T.invM).c_str(), mat(T.transposedInverse).c_str(), vec(p).c_str(),
		       vec(d).c_str(), vec(T.transformPoint(p)).c_str(), vec(T.untransformPoint(p)).c_str(), vec(T.transformDir(d)).c_str(),
		       vec(T.untransformDir(d)).c_str(), vec(T.normal(d)).c_str());
	}
	close_list();

	// 2. BBox::testIntersect / inside / closestIntersection (src/bbox.h:78-174)
	open_list("boxes");
	for (int i = 0; i < 4096; i++) {
		BBox b;
		Vector c = rand_vec(5), e(uni(0.05, 3), uni(0.05, 3), uni(0.05, 3));
		b.vmin = c - e;
		b.vmax = c + e;
		Ray r;
		r.start = (i % 7 == 0) ? c + Vector(e.x * uni(-1, 1), e.y * uni(-1, 1), e.z * uni(-1, 1)) : rand_vec(9);
		r.dir = rand_dir();
		if (i % 11 == 0) r.dir = Vector(r.dir.x, 0, r.dir.z), r.dir.normalize();       // axis-parallel components
		if (i % 13 == 0) r.dir = Vector(0, r.dir.y < 0 ? -1 : 1, 0);
		if (i % 5 == 0) { Vector t = c + Vector(e.x * uni(-1, 1), e.y * uni(-1, 1), e.z * uni(-1, 1)) - r.start; if (t.lengthSqr() > 1e-6) { t.normalize(); r.dir = t; } }
		sep();
		printf("{\"lo\":%s,\"hi\":%s,\"o\":%s,\"d\":%s,\"test\":%d,\"inside\":%d,\"nearest\":%s}", vec(b.vmin).c_str(), vec(b.vmax).c_str(), vec(r.start).c_str(),
		       vec(r.dir).c_str(), b.testIntersect(r) ? 1 : 0, b.inside(r.start) ? 1 : 0, num(b.closestIntersection(r)).c_str());
	}
	close_list();

	// 3. analytic primitives in object space (src/geometry.cpp:31-147)
	open_list("planes");
	for (int i = 0; i < 1024; i++) {
		Plane pl;
		pl.y = uni(-5, 5);
		pl.limit = (i % 3 == 0) ? 1e99 : uni(1, 30);
		Ray r;
		r.start = rand_vec(20);
		r.dir = rand_dir();
		IntersectionInfo info;
		bool ok = pl.intersect(r, info);
		sep();
		printf("{\"y\":%s,\"limit\":%s,\"o\":%s,\"d\":%s,", num(pl.y).c_str(), num(pl.limit).c_str(), vec(r.start).c_str(), vec(r.dir).c_str());
		emit_hit(ok, info);
		printf("}");
	}
	close_list();
	open_list("spheres");
	for (int i = 0; i < 1024; i++) {
		Sphere s;
		s.O = rand_vec(4);
		s.R = uni(0.2, 6);
		s.uvscaling = (i % 2) ? 1 : uni(0.5, 8);
		Ray r;
		r.start = (i % 4 == 0) ? s.O + rand_dir() * (s.R * uni(0, 0.9)) : rand_vec(15);
		r.dir = rand_dir();
		if (i % 3 == 0) { Vector t = s.O + rand_dir() * (s.R * uni(0, 1.2)) - r.start; t.normalize(); r.dir = t; }
		IntersectionInfo info;
		bool ok = s.intersect(r, info);
		sep();
		printf("{\"O\":%s,\"R\":%s,\"uvscaling\":%s,\"o\":%s,\"d\":%s,", vec(s.O).c_str(), num(s.R).c_str(), num(s.uvscaling).c_str(), vec(r.start).c_str(), vec(r.dir).c_str());
		emit_hit(ok, info);
		printf("}");
	}
	close_list();
	open_list("cubes");
	for (int i = 0; i < 1024; i++) {
		Cube c;
		c.O = rand_vec(4);
		c.side = uni(0.3, 7);
		c.beginFrame();
		Ray r;
		r.start = (i % 4 == 0) ? c.O + rand_vec(c.side * 0.45) : rand_vec(15);
		r.dir = rand_dir();
		if (i % 3 == 0) { Vector t = c.O + rand_vec(c.side * 0.6) - r.start; t.normalize(); r.dir = t; }
		IntersectionInfo info;
		bool ok = c.intersect(r, info);
		sep();
		printf("{\"O\":%s,\"side\":%s,\"o\":%s,\"d\":%s,", vec(c.O).c_str(), num(c.side).c_str(), vec(r.start).c_str(), vec(r.dir).c_str());
		emit_hit(ok, info);
		printf("}");
	}
	close_list();

	// 4. CSG of a sphere and a cube (src/geometry.cpp:149-194)
	open_list("csg");
	for (int i = 0; i < 768; i++) {
		Sphere s;
		s.O = rand_vec(1.5);
		s.R = uni(0.8, 3);
		Cube c;
		c.O = rand_vec(1.5);
		c.side = uni(1, 5);
		c.beginFrame();
		CSGBase* op;
		const char* kind;
		if (i % 3 == 0) { op = new CSGUnion; kind = "CSGUnion"; }
		else if (i % 3 == 1) { op = new CSGInter; kind = "CSGInter"; }
		else { op = new CSGDiff; kind = "CSGDiff"; }
		CannedBlock pb;
		pb.geometries["left"] = &s;
		pb.geometries["right"] = &c;
		op->fillProperties(pb);
		Ray r;
		r.start = rand_vec(10);
		Vector t = rand_vec(2.5) - r.start;
		t.normalize();
		r.dir = t;
		IntersectionInfo info;
		bool ok = op->intersect(r, info);
		sep();
		printf("{\"kind\":\"%s\",\"sphere_O\":%s,\"sphere_R\":%s,\"cube_O\":%s,\"cube_side\":%s,\"o\":%s,\"d\":%s,", kind, vec(s.O).c_str(), num(s.R).c_str(),
		       vec(c.O).c_str(), num(c.side).c_str(), vec(r.start).c_str(), vec(r.dir).c_str());
		emit_hit(ok, info);
		printf("}");
		delete op;
	}
	close_list();

	// 5. Node::intersect (src/geometry.cpp:196-208) over transformed primitives
	open_list("nodes");
	for (int i = 0; i < 768; i++) {
		auto ops = random_ops();
		Node n;
		CannedBlock pb;
		pb.transform_ops = ops;
		pb.getTransformProp(n.T);
		Plane pl;
		Sphere s;
		Cube c;
		const char* kind;
		double a = 0, b = 0;
		if (i % 3 == 0) { kind = "Plane"; pl.y = a = uni(-2, 2); pl.limit = b = uni(2, 40); n.geom = &pl; }
		else if (i % 3 == 1) { kind = "Sphere"; s.R = a = uni(0.5, 4); s.uvscaling = b = 1; n.geom = &s; }
		else { kind = "Cube"; c.side = a = uni(0.5, 5); c.beginFrame(); n.geom = &c; }
		Ray r;
		r.start = rand_vec(60);
		Vector t = n.T.transformPoint(rand_vec(2)) - r.start;
		t.normalize();
		r.dir = t;
		IntersectionInfo info;
		bool ok = n.intersect(r, info);
		sep();
		printf("{\"kind\":\"%s\",\"a\":%s,\"b\":%s,\"ops\":%s,\"o\":%s,\"d\":%s,", kind, num(a).c_str(), num(b).c_str(), ops_json(ops).c_str(), vec(r.start).c_str(), vec(r.dir).c_str());
		emit_hit(ok, info);
		printf("}");
	}
	close_list();

	// 6. RectLight (src/lights.cpp:32-102).  getNthSample draws two randDouble(); a default-seeded twin of
	//    util.cpp's generator is advanced in lockstep so that the two draws can be recorded.
	open_list("rect_lights");
	{
		std::mt19937 twin;   // same engine + default seed as `thread_local static std::mt19937 generator` (src/util.cpp:81)
		std::uniform_real_distribution<double> twin_dist;
		for (int i = 0; i < 256; i++) {
			std::vector<std::pair<std::string, Vector>> ops = {{"translate", rand_vec(30)}, {"rotate", Vector(uni(-180, 180), uni(-80, 80), uni(-180, 180))},
			                                                   {"scale", Vector(uni(2, 40), uni(2, 40), uni(2, 40))}};
			RectLight L;
			CannedBlock pb;
			pb.transform_ops = ops;
			pb.numbers["xSubd"] = 1 + int(uni(0, 6));
			pb.numbers["ySubd"] = 1 + int(uni(0, 6));
			pb.numbers["power"] = uni(1, 20000);
			pb.vectors["color"] = Vector(uni(0, 1), uni(0, 1), uni(0, 1));
			L.fillProperties(pb);
			L.beginFrame();
			Transform T;
			pb.getTransformProp(T);
			Ray r;
			r.start = rand_vec(60);
			Vector t = T.transformPoint(Vector(uni(-0.8, 0.8), 0, uni(-0.8, 0.8))) - r.start;
			t.normalize();
			r.dir = t;
			double dist_in = (i % 4 == 0) ? uni(1, 40) : 1e99, dist = dist_in;
			int side = L.intersect(r, dist);
			Vector shade = rand_vec(60);
			int idx = int(uni(0, L.getNumSamples()));
			double j1 = twin_dist(twin), j2 = twin_dist(twin);
			Vector sp;
			Color sc;
			L.getNthSample(idx, shade, sp, sc);
			Color lc = L.getColor();
			sep();
			printf("{\"ops\":%s,\"xSubd\":%d,\"ySubd\":%d,\"power\":%s,\"color\":%s,\"scale_factor\":%s,\"light_color\":[%s,%s,%s],\"o\":%s,\"d\":%s,\"dist_in\":%s,"
			       "\"side\":%d,\"dist_out\":%s,\"shade\":%s,\"solid_angle\":%s,\"idx\":%d,\"j1\":%s,\"j2\":%s,\"sample_pos\":%s,\"sample_color\":[%s,%s,%s]}",
			       ops_json(ops).c_str(), int(pb.numbers["xSubd"]), int(pb.numbers["ySubd"]), num(float(pb.numbers["power"])).c_str(), vec(pb.vectors["color"]).c_str(),
			       num(L.getScaleFactor()).c_str(), num(lc.r).c_str(), num(lc.g).c_str(), num(lc.b).c_str(), vec(r.start).c_str(), vec(r.dir).c_str(), num(dist_in).c_str(),
			       side, num(dist).c_str(), vec(shade).c_str(), num(L.getSolidAngle(shade)).c_str(), idx, num(j1).c_str(), num(j2).c_str(), vec(sp).c_str(),
			       num(sc.r).c_str(), num(sc.g).c_str(), num(sc.b).c_str());
		}
	}
	close_list();

	// 7. sRGB: the 4097-entry table Color::init_sRGB_cache builds (src/sdl.cpp:406-412) from convertTo8bit_sRGB (src/color.h:36-47),
	//    and decompress_sRGB (src/color.h:49-57)
	sep();
	printf("\"srgb_table\": [");
	for (int i = 0; i <= 4096; i++) printf("%s%u", i ? "," : "", convertTo8bit_sRGB(i / 4096.0f));
	printf("]");
	open_list("srgb_decode");
	for (int i = 0; i <= 255; i++) {
		sep();
		printf("[%s,%s]", num(i / 255.0f).c_str(), num(decompress_sRGB(i / 255.0f)).c_str());
	}
	close_list();

	// 8. vector helpers (src/vector.h:160-207)
	open_list("vector_ops");
	for (int i = 0; i < 256; i++) {
		Vector a = rand_dir(), n = rand_dir(), w = rand_vec(3);
		auto basis = othonormedBasis(n);
		sep();
		printf("{\"a\":%s,\"n\":%s,\"w\":%s,\"faceforward\":%s,\"reflect\":%s,\"normalize\":%s,\"basis_u\":%s,\"basis_v\":%s,\"max_dim\":%d}", vec(a).c_str(), vec(n).c_str(),
		       vec(w).c_str(), vec(faceforward(a, n)).c_str(), vec(reflect(a, n)).c_str(), vec(normalize(w)).c_str(), vec(basis.first).c_str(), vec(basis.second).c_str(),
		       w.maxDimension());
	}
	close_list();

	// 9. NESTED CSG (src/geometry.cpp:149-194 is generic over Geometry*): an inner operation over a sphere and a cube is itself the left or
	//    the right operand of an outer operation whose other operand is a second sphere.  Appended last, so that the random sequence of the
	//    lists above -- and with it every earlier vector -- stays what it was.
	open_list("csg_nested");
	for (int i = 0; i < 576; i++) {
		Sphere s;
		s.O = rand_vec(1.5);
		s.R = uni(0.8, 3);
		Cube c;
		c.O = rand_vec(1.5);
		c.side = uni(1, 5);
		c.beginFrame();
		Sphere s2;
		s2.O = rand_vec(2);
		s2.R = uni(0.8, 3);
		const char* kinds[3] = {"CSGUnion", "CSGInter", "CSGDiff"};
		auto make = [](int k) -> CSGBase* { return k == 0 ? (CSGBase*)new CSGUnion : k == 1 ? (CSGBase*)new CSGInter : (CSGBase*)new CSGDiff; };
		const int inner_kind = i % 3, outer_kind = (i / 3) % 3, inner_is_left = (i / 9) % 2;
		CSGBase* inner = make(inner_kind);
		CSGBase* outer = make(outer_kind);
		CannedBlock pi;
		pi.geometries["left"] = &s;
		pi.geometries["right"] = &c;
		inner->fillProperties(pi);
		CannedBlock po;
		po.geometries["left"] = inner_is_left ? (Geometry*)inner : (Geometry*)&s2;
		po.geometries["right"] = inner_is_left ? (Geometry*)&s2 : (Geometry*)inner;
		outer->fillProperties(po);
		Ray r;
		r.start = rand_vec(10);
		Vector t = rand_vec(2.5) - r.start;
		t.normalize();
		r.dir = t;
		IntersectionInfo info;
		bool ok = outer->intersect(r, info);
		sep();
		printf("{\"inner\":\"%s\",\"outer\":\"%s\",\"inner_is_left\":%d,\"sphere_O\":%s,\"sphere_R\":%s,\"cube_O\":%s,\"cube_side\":%s,\"sphere2_O\":%s,\"sphere2_R\":%s,\"o\":%s,\"d\":%s,",
		       kinds[inner_kind], kinds[outer_kind], inner_is_left, vec(s.O).c_str(), num(s.R).c_str(), vec(c.O).c_str(), num(c.side).c_str(), vec(s2.O).c_str(),
		       num(s2.R).c_str(), vec(r.start).c_str(), vec(r.dir).c_str());
		emit_hit(ok, info);
		printf("}");
		delete outer;
		delete inner;
	}
	close_list();

	printf("\n}\n");
	return 0;
}
