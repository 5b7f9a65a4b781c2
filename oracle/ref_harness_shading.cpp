// TEST INFRASTRUCTURE -- second generator of golden vectors from the REAL reference code: shaders, textures, samplers.
//
// Linked against the reference's own shading.cpp, matrix.cpp, util.cpp and lights.cpp, compiled where they lie
// (/root/reference/src, see oracle/Makefile), with the reference's own headers; nothing of the reference is copied here.
//
// shading.cpp as a whole cannot be linked in this image: Lambert / Phong::computeColor read the global `Scene scene`
// (src/shading.cpp:74,89,119,143; defined in scene.cpp, which needs SDL.h), BitmapTexture / BumpTexture call into bitmap.cpp
// (OpenEXR headers).  No stand-in is written for either.  Instead the objects are compiled with -ffunction-sections and linked
// with --gc-sections: members this program never reaches (those above) are dropped by the linker together with their undefined
// references (Layered::fillProperties -- the .hexray parser's door, kept alive by Layered's vtable and never called here -- refers to
// get3Doubles / getFrontToken / getLastToken / stripPunctuation / SyntaxError of scene.cpp: the linker is told to leave unresolved
// symbols of object files alone, `nm -u` shows exactly these five), and every member
// this program DOES reach is the reference's own machine code:
//     CheckerTexture::sample, Fresnel::sample (+ fresnelSchlickApprox), Bumps::modifyNormal, ConstantShader / Const,
//     Reflection::computeColor (mirror and glossy) / eval / spawnRay, Refraction::computeColor (+ refract) / eval / spawnRay,
//     Layered::addLayer / computeColor, hemisphereSample, unitDiskSample, BBox::split;
//     round 3: Lambert::computeColor / eval / spawnRay, Phong::computeColor (+ getLambertTerm), over PointLight / RectLight ::getNthSample.
// What this program itself defines, because the reference declares it elsewhere:
//   * `Color raytrace(const Ray&)` and `bool visible(...)` (src/main.h; defined in main.cpp, which needs SDL.h): shaders call them
//     for their secondary rays.  Here raytrace() RECORDS the ray and answers with a colour that is a fixed function of its
//     direction (the oracle's probe mode answers with the same function), so the test sees both the rays a shader asks for and
//     what it does with the answer; visible() is never reached by the members above.
//   * SceneElement's five trivial members (src/scene.cpp:53-59), as in ref_harness.cpp.
//   * round 3, for Lambert / Phong: the global `Scene scene` (src/scene.cpp:811) with Scene's two-line constructor (src/scene.cpp:724-728:
//     two null pointers) and an EMPTY destructor (the reference's deletes the scene's elements, src/scene.cpp:730-733; the lights
//     listed here live on the stack).  scene.h itself compiles (shading.h includes it).  GlobalSettings' vtable lives in scene.cpp
//     (its key function fillProperties is defined there): it is left unresolved like the parser helpers above -- no virtual call is
//     ever made on `scene.settings`, only its ambientLight field is read.  The lights are the reference's own PointLight / RectLight
//     (lights.cpp is linked), configured through their fillProperties; visible() (src/main.h) RECORDS its two points and answers
//     with bit k of a per-case pattern for the k-th call of the case.  `ref_harness_shading diffuse` prints these sections (a
//     process of its own: the samplers' generator starts from its default seed, so the draws of RectLight::getNthSample and
//     hemisphereSample can be recorded from a twin engine advanced in lockstep, two per call).
// NOT pinned by any reference vector, and why:
// BitmapTexture::sample, BumpTexture::modifyNormal, CubemapEnvironment::getEnvironment, Heightfield (Bitmap -> OpenEXR);
// Mesh::intersectTriangle / intersectKD / buildKD and BBox::intersectTriangle (it calls intersectTriangleFast, defined in mesh.cpp,
// which includes SDL.h); Camera (sdl.h); the render loops of main.cpp.
// Output: JSON on stdout (doubles with 17 significant digits); tests/golden/make_golden.sh commits it as
// tests/golden/reference_shading_vectors.json.
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <map>

#include "bbox.h"
#include "color.h"
#include "lights.h"
#include "main.h"
#include "scene.h"
#include "shading.h"
#include "util.h"

SceneElement::SceneElement() { name[0] = 0; }
void SceneElement::beginRender() {}
void SceneElement::beginFrame() {}
void SceneElement::fillProperties(ParsedBlock&) {}

// ---- the recording raytrace()
static std::vector<Ray> recorded;
Color raytrace(const Ray& ray)
{
	recorded.push_back(ray);
	return Color(float(0.2 + 0.6 * fabs(ray.dir.x)), float(0.2 + 0.6 * fabs(ray.dir.y)), float(0.2 + 0.6 * fabs(ray.dir.z)));
}
// ---- the recording visible(): the k-th call of a case is answered with bit k of the case's pattern
struct VisCall { Vector a, b; };
static std::vector<VisCall> vis_calls;
static unsigned long long vis_pattern = ~0ull;
bool visible(const Vector& a, const Vector& b)
{
	bool answer = (vis_pattern >> (vis_calls.size() & 63)) & 1;
	vis_calls.push_back(VisCall{a, b});
	return answer;
}

// ---- the scene Lambert / Phong read their lights and the ambient term from (see the header)
Scene::Scene() { environment = NULL; camera = NULL; }
Scene::~Scene() {}
Scene scene;

// ---- deterministic inputs (harness-local generator; not the reference's)
static std::mt19937_64 input_rng(20261005);
static double uni(double lo, double hi) { return lo + (hi - lo) * ((input_rng() >> 11) * (1.0 / 9007199254740992.0)); }
static Vector rand_vec(double r) { return Vector(uni(-r, r), uni(-r, r), uni(-r, r)); }
static Vector rand_dir()
{
	Vector v;
	do v = rand_vec(1); while (v.lengthSqr() < 1e-3 || v.lengthSqr() > 1);
	v.normalize();
	return v;
}
static Color rand_color() { return Color(float(uni(0, 1)), float(uni(0, 1)), float(uni(0, 1))); }

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
static std::string col(const Color& c) { return "[" + num(c.r) + "," + num(c.g) + "," + num(c.b) + "]"; }
static std::string rays_json(size_t from = 0)
{
	std::string s = "[";
	for (size_t i = from; i < recorded.size(); i++)
		s += std::string(i > from ? "," : "") + "{\"o\":" + vec(recorded[i].start) + ",\"d\":" + vec(recorded[i].dir) + ",\"depth\":" + std::to_string(recorded[i].depth) +
		     ",\"flags\":" + std::to_string(recorded[i].flags) + "}";
	return s + "]";
}

struct Case {
	Ray ray;
	IntersectionInfo info;
};
static Case random_case(int i)
{
	Case c;
	c.ray.start = rand_vec(20);
	c.ray.dir = rand_dir();
	c.ray.depth = i % 5;
	c.ray.flags = (i % 7 == 0) ? RF_GI_DIFFUSE : 0;
	c.info.ip = rand_vec(20);
	c.info.norm = rand_dir();
	if (i % 3 == 0 && dot(c.ray.dir, c.info.norm) > 0) c.info.norm = -c.info.norm;   // mostly front-facing hits, some back-facing
	c.info.dNdx = rand_dir();
	c.info.dNdy = rand_dir();
	c.info.u = uni(-50, 50);
	c.info.v = uni(-50, 50);
	c.info.dist = uni(0.1, 100);
	return c;
}
static std::string case_json(const Case& c)
{
	return "\"o\":" + vec(c.ray.start) + ",\"d\":" + vec(c.ray.dir) + ",\"depth\":" + std::to_string(c.ray.depth) + ",\"flags\":" + std::to_string(c.ray.flags) +
	       ",\"ip\":" + vec(c.info.ip) + ",\"norm\":" + vec(c.info.norm) + ",\"dNdx\":" + vec(c.info.dNdx) + ",\"dNdy\":" + vec(c.info.dNdy) + ",\"u\":" + num(c.info.u) +
	       ",\"v\":" + num(c.info.v);
}

struct OneNumber : ParsedBlock {   // serves one float property (Bumps keeps its strength private; fillProperties is its only door)
	const char* key;
	double value;
	bool getIntProp(const char*, int*, int, int) override { return false; }
	bool getBoolProp(const char*, bool*) override { return false; }
	bool getFloatProp(const char* name, float* v, float, float) override { if (strcmp(name, key)) return false; *v = float(value); return true; }
	bool getDoubleProp(const char* name, double* v, double, double) override { if (strcmp(name, key)) return false; *v = value; return true; }
	bool getColorProp(const char*, Color*, float, float) override { return false; }
	bool getVectorProp(const char*, Vector*) override { return false; }
	bool getGeometryProp(const char*, Geometry**) override { return false; }
	bool getIntersectableProp(const char*, Intersectable**) override { return false; }
	bool getShaderProp(const char*, Shader**) override { return false; }
	bool getTextureProp(const char*, Texture**) override { return false; }
	bool getNodeProp(const char*, Node**) override { return false; }
	bool getStringProp(const char*, char*) override { return false; }
	bool getFilenameProp(const char*, char*) override { return false; }
	bool getBitmapFileProp(const char*, Bitmap&) override { return false; }
	void getTransformProp(Transform&) override {}
	void requiredProp(const char*) override {}
	void signalError(const char*) override {}
	void signalWarning(const char*) override {}
	int getBlockLines() override { return 0; }
	void getBlockLine(int, int&, char[], char[]) override {}
	SceneParser& getParser() override { return *(SceneParser*)nullptr; }
};

struct LightBlock : OneNumber {   // serves a light's properties: colour / position vectors, numbers, a transform as a list of operations
	std::map<std::string, Vector> vectors;
	std::map<std::string, double> numbers;
	std::vector<std::pair<std::string, Vector>> ops;
	LightBlock() { key = ""; value = 0; }
	bool getIntProp(const char* name, int* v, int, int) override { auto it = numbers.find(name); if (it == numbers.end()) return false; *v = int(it->second); return true; }
	bool getFloatProp(const char* name, float* v, float, float) override { auto it = numbers.find(name); if (it == numbers.end()) return false; *v = float(it->second); return true; }
	bool getDoubleProp(const char* name, double* v, double, double) override { auto it = numbers.find(name); if (it == numbers.end()) return false; *v = it->second; return true; }
	bool getColorProp(const char* name, Color* v, float, float) override
	{
		auto it = vectors.find(name);
		if (it == vectors.end()) return false;
		*v = Color(float(it->second.x), float(it->second.y), float(it->second.z));
		return true;
	}
	bool getVectorProp(const char* name, Vector* v) override { auto it = vectors.find(name); if (it == vectors.end()) return false; *v = it->second; return true; }
	void getTransformProp(Transform& T) override
	{
		for (auto& op : ops) {
			if (op.first == "scale") T.scale(op.second.x, op.second.y, op.second.z);
			else if (op.first == "rotate") T.rotate(op.second.x, op.second.y, op.second.z);
			else T.translate(op.second);
		}
	}
};

// Lambert / Phong (src/shading.cpp:57-145) over the reference's own lights.  Everything a case needs to be replayed is printed: the
// lights as .hexray properties, the pattern visible() answered with, the two points of every visible() call, the generator's draws.
static int diffuse_sections()
{
	printf("{\n\"_comment\": \"golden vectors produced by `oracle/ref_harness_shading diffuse` from the reference's own shading.cpp / lights.cpp / util.cpp\"");
	first_in_scope = false;
	std::mt19937 twin;   // same engine + default seed as `thread_local static std::mt19937 generator` (src/util.cpp:81), advanced in lockstep
	std::uniform_real_distribution<double> twin_dist;
	open_list("diffuse");
	for (int i = 0; i < 256; i++) {
		Case c = random_case(i);
		const bool phong = i % 2;
		Lambert lam;
		Phong ph;
		Lambert& sh = phong ? (Lambert&)ph : lam;
		sh.diffuse = rand_color();
		ph.specular = rand_color();
		ph.exponent = float(uni(1, 120));
		CheckerTexture ck;
		ck.color1 = rand_color();
		ck.color2 = rand_color();
		ck.scaling = uni(0.5, 20);
		const bool textured = i % 5 == 3;
		if (textured) sh.diffuseTex = &ck;
		scene.settings.ambientLight = rand_color() * 0.4f;
		const int nl = 1 + i % 3;
		std::vector<PointLight> points(nl);
		RectLight rect;
		const int rect_at = (i % 4 == 2) ? int(uni(0, nl)) : -1;   // which of the lights is the rectangle light, if any
		scene.lights.clear();
		std::string lights = "[", draws = "[";
		int ndraws = 0;
		for (int k = 0; k < nl; k++) {
			LightBlock pb;
			pb.vectors["color"] = Vector(uni(0, 1), uni(0, 1), uni(0, 1));
			pb.numbers["power"] = uni(20, 4000);
			if (i % 16 == 9 && k == 0) pb.vectors["color"] = Vector(0, 0, 0);   // a light whose samples are skipped (lightColor.isZero())
			if (k == rect_at) {
				pb.ops = {{"translate", c.info.ip + rand_dir() * uni(4, 30)}, {"rotate", Vector(uni(-180, 180), uni(-80, 80), uni(-180, 180))},
				          {"scale", Vector(uni(1, 8), uni(1, 8), uni(1, 8))}};
				pb.numbers["xSubd"] = 1 + int(uni(0, 3));
				pb.numbers["ySubd"] = 1 + int(uni(0, 3));
				rect.fillProperties(pb);
				rect.beginFrame();
				scene.lights.push_back(&rect);
				std::string ops = "[";
				for (size_t q = 0; q < pb.ops.size(); q++) ops += std::string(q ? "," : "") + "{\"op\":\"" + pb.ops[q].first + "\",\"v\":" + vec(pb.ops[q].second) + "}";
				lights += std::string(k ? "," : "") + "{\"rect\":1,\"color\":" + vec(pb.vectors["color"]) + ",\"power\":" + num(float(pb.numbers["power"])) + ",\"ops\":" + ops +
				          "],\"xSubd\":" + std::to_string(int(pb.numbers["xSubd"])) + ",\"ySubd\":" + std::to_string(int(pb.numbers["ySubd"])) + "}";
				for (int q = 0; q < 2 * rect.getNumSamples(); q++) draws += std::string(ndraws++ ? "," : "") + num(twin_dist(twin));   // two randDouble() per sample
			} else {
				pb.vectors["pos"] = c.info.ip + rand_dir() * uni(3, 40);
				points[k].fillProperties(pb);
				scene.lights.push_back(&points[k]);
				lights += std::string(k ? "," : "") + "{\"rect\":0,\"color\":" + vec(pb.vectors["color"]) + ",\"power\":" + num(float(pb.numbers["power"])) + ",\"pos\":" +
				          vec(pb.vectors["pos"]) + "}";
			}
		}
		lights += "]";
		draws += "]";
		vis_pattern = input_rng();
		if (i % 8 == 0) vis_pattern = ~0ull;   // everything visible
		vis_calls.clear();
		Color out = sh.computeColor(c.ray, c.info);
		std::string vis = "[";
		for (size_t q = 0; q < vis_calls.size(); q++) vis += std::string(q ? "," : "") + "{\"a\":" + vec(vis_calls[q].a) + ",\"b\":" + vec(vis_calls[q].b) + "}";
		vis += "]";
		sep();
		printf("{%s,\"phong\":%d,\"color\":%s,\"specular\":%s,\"exponent\":%s,\"textured\":%d,\"checker1\":%s,\"checker2\":%s,\"checker_scaling\":%s,\"ambient\":%s,"
		       "\"lights\":%s,\"pattern\":\"%llu\",\"visible_calls\":%s,\"draws\":%s,\"out\":%s}",
		       case_json(c).c_str(), phong ? 1 : 0, col(sh.diffuse).c_str(), col(ph.specular).c_str(), num(ph.exponent).c_str(), textured ? 1 : 0, col(ck.color1).c_str(),
		       col(ck.color2).c_str(), num(ck.scaling).c_str(), col(scene.settings.ambientLight).c_str(), lights.c_str(), vis_pattern, vis.c_str(), draws.c_str(), col(out).c_str());
		scene.lights.clear();
	}
	close_list();
	// Lambert::eval / spawnRay (src/shading.cpp:93-109) as pathtrace() calls them; spawnRay draws two numbers (hemisphereSample)
	open_list("diffuse_brdf");
	for (int i = 0; i < 128; i++) {
		Case c = random_case(i);
		Lambert lam;
		lam.diffuse = rand_color();
		CheckerTexture ck;
		ck.color1 = rand_color();
		ck.color2 = rand_color();
		ck.scaling = uni(0.5, 20);
		const bool textured = i % 4 == 1;
		if (textured) lam.diffuseTex = &ck;
		Ray w_out = c.ray;
		w_out.depth++;
		Color bc;
		float pdf = 0;
		const double u = twin_dist(twin), v = twin_dist(twin);
		lam.spawnRay(c.info, c.ray.dir, w_out, bc, pdf);
		Vector other = rand_dir();
		Color e = lam.eval(c.info, c.ray.dir, other);
		sep();
		printf("{%s,\"color\":%s,\"textured\":%d,\"checker1\":%s,\"checker2\":%s,\"checker_scaling\":%s,\"draw_u\":%s,\"draw_v\":%s,\"spawn_o\":%s,\"spawn_d\":%s,\"spawn_flags\":%u,"
		       "\"spawn_color\":%s,\"pdf\":%s,\"other\":%s,\"eval_other\":%s}",
		       case_json(c).c_str(), col(lam.diffuse).c_str(), textured ? 1 : 0, col(ck.color1).c_str(), col(ck.color2).c_str(), num(ck.scaling).c_str(), num(u).c_str(),
		       num(v).c_str(), vec(w_out.start).c_str(), vec(w_out.dir).c_str(), w_out.flags, col(bc).c_str(), num(pdf).c_str(), vec(other).c_str(), col(e).c_str());
	}
	close_list();
	printf("\n}\n");
	return 0;
}

int main(int argc, char** argv)
{
	if (argc > 1 && !strcmp(argv[1], "diffuse")) return diffuse_sections();
	printf("{\n\"_comment\": \"golden vectors produced by oracle/ref_harness_shading.cpp from the reference's own shading.cpp / util.cpp / bbox.h\"");
	first_in_scope = false;

	// 1. BBox::split (src/bbox.h:210-216)
	open_list("box_split");
	for (int i = 0; i < 128; i++) {
		BBox b, l, r;
		Vector c = rand_vec(5), e(uni(0.05, 3), uni(0.05, 3), uni(0.05, 3));
		b.vmin = c - e;
		b.vmax = c + e;
		int axis = i % 3;
		double where = b.vmin[axis] + (b.vmax[axis] - b.vmin[axis]) * uni(0, 1);
		b.split((Axis)axis, where, l, r);
		sep();
		printf("{\"lo\":%s,\"hi\":%s,\"axis\":%d,\"where\":%s,\"left_lo\":%s,\"left_hi\":%s,\"right_lo\":%s,\"right_hi\":%s}", vec(b.vmin).c_str(), vec(b.vmax).c_str(), axis,
		       num(where).c_str(), vec(l.vmin).c_str(), vec(l.vmax).c_str(), vec(r.vmin).c_str(), vec(r.vmax).c_str());
	}
	close_list();

	// 2. the samplers of util.cpp (src/util.cpp:81-129).  Their generator is a default-seeded thread_local std::mt19937 read through
	//    std::uniform_real_distribution<double>; `twin` is the same engine with the same seed read the same way, so rng_stream[k] is
	//    the k-th number randDouble() hands out in this process.  The sections "unit_disk", "hemisphere" and "glossy" consume that
	//    stream in this order and nothing else does (the test walks it with the oracle's own disk / hemisphere / glossy maps).
	{
		std::mt19937 twin;
		std::uniform_real_distribution<double> twin_dist;
		sep();
		printf("\"rng_stream\": [");
		for (int i = 0; i < 8192; i++) printf("%s%s", i ? "," : "", num(twin_dist(twin)).c_str());
		printf("]");
	}
	open_list("unit_disk");
	for (int i = 0; i < 256; i++) {
		double x, y;
		unitDiskSample(x, y);
		sep();
		printf("[%s,%s]", num(x).c_str(), num(y).c_str());
	}
	close_list();
	open_list("hemisphere");
	for (int i = 0; i < 256; i++) {
		Vector n = rand_dir();
		Vector d = hemisphereSample(n);
		sep();
		printf("{\"n\":%s,\"dir\":%s}", vec(n).c_str(), vec(d).c_str());
	}
	close_list();
	// glossy Reflection::computeColor (src/shading.cpp:165-184): every sample runs unitDiskSample inside a rejection loop
	open_list("glossy");
	for (int i = 0; i < 64; i++) {
		Case c = random_case(i);
		Reflection refl;
		refl.glossiness = float(uni(0.3, 0.95));
		refl.numSamples = 1 + i % 5;
		refl.reflColor = rand_color();
		recorded.clear();
		Color out = refl.computeColor(c.ray, c.info);
		sep();
		printf("{%s,\"glossiness\":%s,\"num_samples\":%d,\"color\":%s,\"out\":%s,\"rays\":%s}", case_json(c).c_str(), num(refl.glossiness).c_str(), refl.numSamples,
		       col(refl.reflColor).c_str(), col(out).c_str(), rays_json().c_str());
	}
	close_list();

	// 3. textures: CheckerTexture::sample (src/shading.cpp:50-55), Fresnel::sample (:326-343), Bumps::modifyNormal (:363-377)
	open_list("checker");
	for (int i = 0; i < 256; i++) {
		Case c = random_case(i);
		CheckerTexture t;
		t.color1 = rand_color();
		t.color2 = rand_color();
		t.scaling = uni(0.3, 30);
		if (i % 8 == 0) { c.info.u = t.scaling * int(uni(-6, 6)); c.info.v = -t.scaling * int(uni(0, 6)); }   // cell borders, negative cells
		Color out = t.sample(c.ray.dir, c.info);
		sep();
		printf("{%s,\"color1\":%s,\"color2\":%s,\"scaling\":%s,\"out\":%s}", case_json(c).c_str(), col(t.color1).c_str(), col(t.color2).c_str(), num(t.scaling).c_str(),
		       col(out).c_str());
	}
	close_list();
	open_list("fresnel");
	for (int i = 0; i < 256; i++) {
		Case c = random_case(i);
		Fresnel t;
		t.ior = uni(1.01, 2.5);
		Color out = t.sample(c.ray.dir, c.info);
		sep();
		printf("{%s,\"ior\":%s,\"out\":%s}", case_json(c).c_str(), num(t.ior).c_str(), col(out).c_str());
	}
	close_list();
	open_list("bumps");
	for (int i = 0; i < 256; i++) {
		Case c = random_case(i);
		Bumps t;
		OneNumber pb;
		pb.key = "strength";
		pb.value = (i % 16 == 0) ? 0.0 : uni(0.1, 12);
		t.fillProperties(pb);
		IntersectionInfo after = c.info;
		t.modifyNormal(after);
		sep();
		printf("{%s,\"strength\":%s,\"norm_out\":%s}", case_json(c).c_str(), num(float(pb.value)).c_str(), vec(after.norm).c_str());
	}
	close_list();

	// 4. shaders: Reflection (mirror) / Refraction / Layered ::computeColor, with the rays they send to raytrace()
	open_list("reflection");
	for (int i = 0; i < 256; i++) {
		Case c = random_case(i);
		Reflection s;
		s.reflColor = rand_color();
		recorded.clear();
		Color out = s.computeColor(c.ray, c.info);
		sep();
		printf("{%s,\"color\":%s,\"out\":%s,\"rays\":%s}", case_json(c).c_str(), col(s.reflColor).c_str(), col(out).c_str(), rays_json().c_str());
	}
	close_list();
	open_list("refraction");
	for (int i = 0; i < 384; i++) {
		Case c = random_case(i);
		if (i % 4 == 1) {   // grazing incidence: total internal reflection when leaving the denser medium
			Vector t = c.info.norm ^ rand_dir();
			t.normalize();
			c.ray.dir = t * 0.97 + c.info.norm * ((i % 8 == 1) ? 0.2 : -0.2);
			c.ray.dir.normalize();
		}
		Refraction s;
		s.refrColor = rand_color();
		s.ior = uni(1.05, 2.4);
		recorded.clear();
		Color out = s.computeColor(c.ray, c.info);
		sep();
		printf("{%s,\"color\":%s,\"ior\":%s,\"out\":%s,\"rays\":%s}", case_json(c).c_str(), col(s.refrColor).c_str(), num(s.ior).c_str(), col(out).c_str(), rays_json().c_str());
	}
	close_list();
	// Layered::computeColor (src/shading.cpp:315-324): col = col * (1 - blend) + layer * blend over Const / Reflection / Refraction
	// layers, blend = a colour, a CheckerTexture or a Fresnel texture (the glass of data/meshes.hexray is layer 2 + layer 1 below)
	open_list("layered");
	for (int i = 0; i < 192; i++) {
		Case c = random_case(i);
		ConstantShader k0(rand_color());
		Reflection r1;
		r1.reflColor = rand_color();
		Refraction r2;
		r2.refrColor = rand_color();
		r2.ior = uni(1.1, 2.0);
		Fresnel fr;
		fr.ior = r2.ior;
		CheckerTexture ck;
		ck.color1 = rand_color();
		ck.color2 = rand_color();
		ck.scaling = uni(1, 20);
		Layered L;
		int n = 1 + i % 4;
		std::string layers = "[";
		for (int j = 0; j < n; j++) {
			int which = int(uni(0, 3)), tex = int(uni(0, 3));   // shader: 0 const 1 reflection 2 refraction; blend: 0 colour 1 checker 2 fresnel
			Color blend = rand_color();
			Shader* sh = which == 0 ? (Shader*)&k0 : which == 1 ? (Shader*)&r1 : (Shader*)&r2;
			L.addLayer(sh, blend, tex == 0 ? nullptr : tex == 1 ? (Texture*)&ck : (Texture*)&fr);
			layers += std::string(j ? "," : "") + "{\"shader\":" + std::to_string(which) + ",\"tex\":" + std::to_string(tex) + ",\"blend\":" + col(blend) + "}";
		}
		layers += "]";
		recorded.clear();
		Color out = L.computeColor(c.ray, c.info);
		sep();
		printf("{%s,\"const\":%s,\"refl\":%s,\"refr\":%s,\"ior\":%s,\"checker1\":%s,\"checker2\":%s,\"checker_scaling\":%s,\"layers\":%s,\"out\":%s,\"rays\":%s}",
		       case_json(c).c_str(), col(k0.color).c_str(), col(r1.reflColor).c_str(), col(r2.refrColor).c_str(), num(r2.ior).c_str(), col(ck.color1).c_str(),
		       col(ck.color2).c_str(), num(ck.scaling).c_str(), layers.c_str(), col(out).c_str(), rays_json().c_str());
	}
	close_list();

	// 5. BRDF::eval / spawnRay of the specular shaders (src/shading.cpp:189-205,234-263), as pathtrace() calls them (src/main.cpp:122-168)
	open_list("brdf");
	for (int i = 0; i < 256; i++) {
		Case c = random_case(i);
		Reflection rl;
		rl.reflColor = rand_color();
		Refraction rr;
		rr.refrColor = rand_color();
		rr.ior = uni(1.05, 2.4);
		bool refraction = i % 2;
		BRDF* b = refraction ? (BRDF*)&rr : (BRDF*)&rl;
		Ray w_out = c.ray;
		w_out.depth++;
		Color bc;
		float pdf = 0;
		b->spawnRay(c.info, c.ray.dir, w_out, bc, pdf);
		// eval: once along the mirror direction (the only place it is not black), once along a random one
		Vector mirror = reflect(c.ray.dir, faceforward(c.ray.dir, c.info.norm)), other = rand_dir();
		Color e1 = b->eval(c.info, c.ray.dir, mirror), e2 = b->eval(c.info, c.ray.dir, other);
		sep();
		printf("{%s,\"kind\":%d,\"color\":%s,\"ior\":%s,\"spawn_o\":%s,\"spawn_d\":%s,\"spawn_flags\":%u,\"spawn_color\":%s,\"pdf\":%s,\"mirror\":%s,\"other\":%s,"
		       "\"eval_mirror\":%s,\"eval_other\":%s}",
		       case_json(c).c_str(), refraction ? 1 : 0, col(refraction ? rr.refrColor : rl.reflColor).c_str(), num(rr.ior).c_str(), vec(w_out.start).c_str(),
		       vec(w_out.dir).c_str(), w_out.flags, col(bc).c_str(), num(pdf).c_str(), vec(mirror).c_str(), vec(other).c_str(), col(e1).c_str(), col(e2).c_str());
	}
	close_list();

	printf("\n}\n");
	return 0;
}
