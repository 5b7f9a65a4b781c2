// .hexray blocks -> flat element tables.  Each builder cites the reference fillProperties /
// beginRender / beginFrame it reproduces.
#include "flatten.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <sys/stat.h>

namespace hxh {

namespace {

enum Kind { K_SETTINGS, K_CAMERA, K_ENVIRONMENT, K_GEOMETRY, K_TEXTURE, K_SHADER, K_LIGHT, K_NODE, K_UNKNOWN };

// class factory, reference src/scene.cpp:780-809
Kind kind_of(const std::string& cls)
{
	static const struct { const char* name; Kind kind; } table[] = {
	    {"GlobalSettings", K_SETTINGS}, {"Camera", K_CAMERA}, {"CubemapEnvironment", K_ENVIRONMENT},
	    {"Plane", K_GEOMETRY}, {"Sphere", K_GEOMETRY}, {"Cube", K_GEOMETRY}, {"CSGUnion", K_GEOMETRY},
	    {"CSGInter", K_GEOMETRY}, {"CSGDiff", K_GEOMETRY}, {"Mesh", K_GEOMETRY}, {"Heightfield", K_GEOMETRY},
	    {"CheckerTexture", K_TEXTURE}, {"BitmapTexture", K_TEXTURE}, {"Fresnel", K_TEXTURE},
	    {"BumpTexture", K_TEXTURE}, {"Bumps", K_TEXTURE},
	    {"Lambert", K_SHADER}, {"Phong", K_SHADER}, {"Reflection", K_SHADER}, {"Refraction", K_SHADER},
	    {"Layered", K_SHADER}, {"Const", K_SHADER},
	    {"PointLight", K_LIGHT}, {"RectLight", K_LIGHT}, {"Node", K_NODE}};
	for (const auto& e : table)
		if (cls == e.name) return e.kind;
	return K_UNKNOWN;
}

bool file_exists(const std::string& p)
{
	struct stat st;
	return stat(p.c_str(), &st) == 0;
}

int index_of(const std::vector<std::string>& names, const std::string& name)
{
	for (size_t i = 0; i < names.size(); i++)
		if (names[i] == name) return int(i);
	return -1;
}

void set_color(float* dst, float r, float g, float b) { dst[0] = r; dst[1] = g; dst[2] = b; }

struct Loader {
	FlatScene& fs;
	std::string root_dir;
	std::vector<Block>& blocks;
	std::vector<int> slot;   // per block: index into its kind's table

	std::string resolve(const Block& b, const char* key, bool& present)
	{
		const Property* p = b.find(key);
		present = p != nullptr;
		if (!p) return "";
		std::string full = root_dir + p->value;
		if (!file_exists(full)) throw ParseError(p->line, "Required file not found (" + p->value + ")");
		return full;
	}

	int add_bitmap(const Image& img, int prep = 0, float gamma = 1.0f)
	{
		hxr_bitmap bm;
		bm.width = img.ok() ? img.w : 0;
		bm.height = img.ok() ? img.h : 0;
		bm.first_texel = int64_t(fs.f32_pool.size());
		bm.prep = img.ok() ? prep : 0;
		bm.gamma = gamma;
		bm.first_gamma_table = -1;
		if (img.ok()) fs.f32_pool.insert(fs.f32_pool.end(), img.rgb.begin(), img.rgb.end());
		if (bm.prep & HXR_BITMAP_GAMMA) {
			float table[256];
			if (gamma_table(img, gamma, table)) {
				bm.first_gamma_table = int64_t(fs.f32_pool.size());
				fs.f32_pool.insert(fs.f32_pool.end(), table, table + 256);
			}
		}
		fs.bitmaps.push_back(bm);
		return int(fs.bitmaps.size()) - 1;
	}

	// A bitmap whose loader passes (Bitmap::decompressGamma / Bitmap::differentiate) run here, or -- FlatScene::gpu_bitmap_prep -- are
	// left to the render module (hxr_bitmap::prep)
	int add_prepared_bitmap(Image& img, float gamma, bool differentiated)
	{
		int prep = 0;
		if (gamma != 1.0f) { if (fs.gpu_bitmap_prep) prep |= HXR_BITMAP_GAMMA; else decode_gamma(img, gamma); }
		if (differentiated) { if (fs.gpu_bitmap_prep) prep |= HXR_BITMAP_DIFFERENTIATE; else differentiate(img); }
		return add_bitmap(img, prep, gamma);
	}

	int64_t add_vectors(const std::vector<V3>& v)
	{
		int64_t first = int64_t(fs.f64_pool.size());
		for (const V3& p : v) { fs.f64_pool.push_back(p.x); fs.f64_pool.push_back(p.y); fs.f64_pool.push_back(p.z); }
		return first;
	}

	int texture_ref(const Block& b, const char* key)
	{
		const Property* p = b.find(key);
		if (!p) return -1;
		int i = index_of(fs.texture_names, p->value);
		if (i < 0) throw ParseError(p->line, "Texture not defined");
		return i;
	}

	// ---------------------------------------------------------------- settings / camera / environment
	void fill_settings(const Block& b)   // GlobalSettings::fillProperties, src/scene.cpp:764-778
	{
		hxr_settings& s = fs.settings;
		b.get_int("frameWidth", s.frame_width);
		b.get_int("frameHeight", s.frame_height);
		b.get_color("ambientLight", s.ambient);
		b.get_int("maxTraceDepth", s.max_trace_depth);
		b.get_bool("dbg", fs.dbg);
		bool flag;
		if (b.get_bool("wantAA", flag)) s.want_aa = flag;
		b.get_int("prepassSamples", fs.prepass_samples, 0);
		if (b.get_bool("gi", flag)) s.gi = flag;
		b.get_int("numPaths", s.num_paths, 1);
		b.get_int("numThreads", fs.num_threads, 0, 1024);
		b.get_bool("interactive", fs.interactive);
		b.get_int("foveatedRadius", fs.foveated_radius, 0, 1000);
	}

	void fill_camera(const Block& b)   // Camera::fillProperties, src/camera.h:57-72
	{
		CameraSpec c;
		c.present = true;
		if (!b.get_vec("pos", c.pos)) b.require("pos");
		b.get_double("aspectRatio", c.aspect, 1e-6);
		b.get_double("fov", c.fov, 0.0001, 179);
		b.get_double("yaw", c.yaw);
		b.get_double("pitch", c.pitch, -90, 90);
		b.get_double("roll", c.roll);
		b.get_double("fNumber", c.f_number, 0.5, 128.0);
		b.get_int("numSamples", c.num_samples, 1);
		b.get_double("focalPlaneDist", c.focal_plane_dist, 1e-3, 1e+6);
		b.get_bool("dof", c.dof);
		b.get_bool("autoFocus", c.auto_focus);
		b.get_double("stereoSeparation", c.stereo_separation, 0.0);
		fs.camera_spec = c;
	}

	void fill_environment(const Block& b)   // CubemapEnvironment, src/environment.h:63-73, src/environment.cpp:32-52
	{
		hxr_environment env;
		env.present = 1;
		env.loaded = 0;
		for (int& s : env.side_bitmap) s = -1;
		float gamma = 1.0f;
		b.get_float("assumedGamma", gamma, 0.1f, 10.0f);
		bool present;
		std::string folder = resolve(b, "folder", present);
		if (!present) b.require("folder");
		static const char* prefix[2] = {"neg", "pos"};
		static const char* axis[3] = {"x", "y", "z"};
		static const char* suffix[2] = {".bmp", ".exr"};
		bool all_ok = true;
		int n = 0;
		for (int pi = 0; pi < 2 && all_ok; pi++)
			for (int ax = 0; ax < 3 && all_ok; ax++, n++) {
				Image img;
				for (int si = 0; si < 2; si++) {
					std::string fn = folder + "/" + prefix[pi] + axis[ax] + suffix[si];
					if (file_exists(fn) && load_image(fn, img)) break;
				}
				if (!img.ok()) { all_ok = false; break; }
				env.side_bitmap[n] = add_prepared_bitmap(img, gamma, false);
			}
		env.loaded = all_ok ? 1 : 0;
		if (!all_ok) {
			fprintf(stderr, "CubemapEnvironment: Could not load maps from `%s'\n", folder.c_str());
			fs.warnings.push_back("CubemapEnvironment: could not load maps from " + folder + " (environment renders black)");
		}
		fs.environment = env;
	}

	// ---------------------------------------------------------------- geometry
	void fill_geometry(const Block& b, int gi)
	{
		hxr_geometry& g = fs.geometries[gi];
		if (b.cls == "Plane") {          // src/geometry.h:57-67
			g.type = HXR_GEOM_PLANE;
			g.p[0] = 0;
			g.p[1] = 1e99;
			b.get_double("y", g.p[0]);
			b.get_double("limit", g.p[1]);
		} else if (b.cls == "Sphere") {  // src/geometry.h:69-81
			g.type = HXR_GEOM_SPHERE;
			V3 O(0, 0, 0);
			g.p[3] = 1;
			g.p[4] = 1;
			b.get_vec("O", O);
			b.get_double("R", g.p[3], 0.0);
			b.get_double("uvscaling", g.p[4], 1e-6);
			store3(g.p, O);
		} else if (b.cls == "Cube") {    // src/geometry.h:83-107; beginFrame halves the side
			g.type = HXR_GEOM_CUBE;
			V3 O(0, 0, 0);
			double side = 1;
			b.get_vec("O", O);
			b.get_double("side", side, 0.0);
			store3(g.p, O);
			g.p[3] = side * 0.5;
		} else if (b.cls == "CSGUnion" || b.cls == "CSGInter" || b.cls == "CSGDiff") {   // src/geometry.h:109-136
			g.type = b.cls == "CSGUnion" ? HXR_GEOM_CSG_UNION : b.cls == "CSGInter" ? HXR_GEOM_CSG_INTER : HXR_GEOM_CSG_DIFF;
			b.require("left");
			b.require("right");
			const Property* l = b.find("left");
			const Property* r = b.find("right");
			g.left = index_of(fs.geometry_names, l->value);
			g.right = index_of(fs.geometry_names, r->value);
			if (g.left < 0) throw ParseError(l->line, "Geometry not defined");
			if (g.right < 0) throw ParseError(r->line, "Geometry not defined");
		} else if (b.cls == "Mesh") {    // src/mesh.h:89-119 + beginRender
			g.type = HXR_GEOM_MESH;
			MeshData md;
			b.get_bool("faceted", md.faceted);
			b.get_bool("backfaceCulling", md.backface_culling);
			b.get_bool("useKDTree", md.use_kd);
			b.get_bool("autoSmooth", md.auto_smooth);
			b.get_bool("recenter", md.recenter);
			bool present;
			std::string fn = resolve(b, "file", present);
			if (!present) b.require("file");
			md.defer_triangle_prep = fs.gpu_prepare_triangles;
			const auto t0 = std::chrono::steady_clock::now();
			if (!md.load_obj(fn)) throw ParseError(b.line_end, "Could not parse OBJ file!");
			const auto t1 = std::chrono::steady_clock::now();
			md.begin_render(!fs.gpu_kd_build);
			if (getenv("HXH_TIMING"))   // developer aid: where the host preparation of a mesh goes
				fprintf(stderr, "[hxh] %s: OBJ parse + prepareTriangles %.1f ms, kd build %.1f ms (%d triangles, %lld nodes, %zu leaf entries)\n", fn.c_str(),
				        std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(),
				        (int)md.triangles.size(), (long long)md.kd_nodes.size(), md.leaf_indices.size());
			hxr_mesh m;
			memset(&m, 0, sizeof m);
			m.faceted = md.faceted;
			m.backface_culling = md.backface_culling;
			m.has_kd = md.use_kd;
			store3(m.bbox_min, md.bbox_min);
			store3(m.bbox_max, md.bbox_max);
			m.first_vertex = add_vectors(md.vertices) / 3;
			m.num_vertices = int64_t(md.vertices.size());
			m.first_normal = add_vectors(md.normals) / 3;
			m.num_normals = int64_t(md.normals.size());
			m.first_uv = add_vectors(md.uvs) / 3;
			m.num_uvs = int64_t(md.uvs.size());
			m.first_triangle = int64_t(fs.triangles.size());
			m.num_triangles = int64_t(md.triangles.size());
			fs.triangles.insert(fs.triangles.end(), md.triangles.begin(), md.triangles.end());
			m.first_kd_node = int64_t(fs.kd_nodes.size());
			m.num_kd_nodes = int64_t(md.kd_nodes.size());
			fs.kd_nodes.insert(fs.kd_nodes.end(), md.kd_nodes.begin(), md.kd_nodes.end());
			m.first_leaf_index = int64_t(fs.leaf_indices.size());
			m.num_leaf_indices = int64_t(md.leaf_indices.size());
			fs.leaf_indices.insert(fs.leaf_indices.end(), md.leaf_indices.begin(), md.leaf_indices.end());
			g.ref = int(fs.meshes.size());
			fs.meshes.push_back(m);
			fs.mesh_stats.push_back(md.stats);
		} else if (b.cls == "Heightfield") {   // src/heightfield.cpp:173-261
			g.type = HXR_GEOM_HEIGHTFIELD;
			HeightfieldData hd;
			b.get_bool("useOptimization", hd.use_optimization);
			bool present;
			std::string fn = resolve(b, "file", present);
			if (!present) b.require("file");
			Image img;
			load_image(fn, img);
			double blur = 0;
			b.get_double("blur", blur, 0, 1000);
			if (!img.ok()) throw ParseError(b.line_end, "Heightfield: could not load the height bitmap");
			const auto t0 = std::chrono::steady_clock::now();
			hd.build(img, blur, !fs.gpu_heightfield_tables, fs.gpu_heightfield_blur);
			if (getenv("HXH_TIMING"))
				fprintf(stderr, "[hxh] heightfield %dx%d: blur %g + normals%s %.1f ms\n", hd.W, hd.H, blur, fs.gpu_heightfield_tables ? "" : " + maxima tables",
				        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
			hxr_heightfield h;
			memset(&h, 0, sizeof h);
			h.W = hd.W;
			h.H = hd.H;
			h.max_k = hd.max_k;
			h.use_optimization = hd.use_optimization;
			store3(h.bbox_min, hd.bbox_min);
			store3(h.bbox_max, hd.bbox_max);
			h.first_height = int64_t(fs.f32_pool.size());
			fs.f32_pool.insert(fs.f32_pool.end(), hd.heights.begin(), hd.heights.end());
			if (fs.gpu_heightfield_tables) {
				h.first_maxh = h.first_highmap = -1;
				h.max_k = 0;
			} else {
				h.first_maxh = int64_t(fs.f32_pool.size());
				fs.f32_pool.insert(fs.f32_pool.end(), hd.max_h.begin(), hd.max_h.end());
				h.first_highmap = int64_t(fs.f32_pool.size());
				fs.f32_pool.insert(fs.f32_pool.end(), hd.high_map.begin(), hd.high_map.end());
			}
			if (fs.gpu_heightfield_blur) {
				h.raw_heights = 1;
				h.first_normal = -1;
				h.blur_radius = hd.blur_radius;
				h.first_blur_kernel = int64_t(fs.f32_pool.size());
				fs.f32_pool.insert(fs.f32_pool.end(), hd.blur_kernel.begin(), hd.blur_kernel.end());
			} else {
				h.first_normal = add_vectors(hd.normals) / 3;
			}
			g.ref = int(fs.heightfields.size());
			fs.heightfields.push_back(h);
		}
	}

	// ---------------------------------------------------------------- textures
	void fill_texture(const Block& b, int ti)
	{
		hxr_texture& t = fs.textures[ti];
		if (b.cls == "CheckerTexture") {        // src/shading.h:56-67
			t.type = HXR_TEX_CHECKER;
			set_color(t.color1, 1, 1, 1);
			set_color(t.color2, 0, 0, 0);
			t.scaling = 20.0;
			b.get_color("color1", t.color1);
			b.get_color("color2", t.color2);
			b.get_double("scaling", t.scaling);
		} else if (b.cls == "BitmapTexture") {  // src/shading.h:69-85
			t.type = HXR_TEX_BITMAP;
			t.scaling = 100.0;
			b.get_double("scaling", t.scaling);
			bool present;
			std::string fn = resolve(b, "file", present);
			if (!present) b.require("file");
			Image img;
			load_image(fn, img);
			float gamma = 1.0f;
			b.get_float("assumedGamma", gamma, 0.1f, 100.0f);
			t.bitmap = add_prepared_bitmap(img, gamma, false);
		} else if (b.cls == "Fresnel") {        // src/shading.h:179-187
			t.type = HXR_TEX_FRESNEL;
			t.ior = 1.33;
			b.get_double("ior", t.ior, 1e-6, 10);
		} else if (b.cls == "BumpTexture") {    // src/shading.h:189-204; beginRender differentiates
			t.type = HXR_TEX_BUMPTEX;
			t.strength = 1;
			t.scaling = 1;
			b.get_double("strength", t.strength);
			b.get_double("scaling", t.scaling);
			bool present;
			std::string fn = resolve(b, "file", present);
			if (!present) b.require("file");
			Image img;
			load_image(fn, img);
			t.bitmap = add_prepared_bitmap(img, 1.0f, true);
		} else if (b.cls == "Bumps") {          // src/shading.h:217-227 (strength is a float there)
			t.type = HXR_TEX_BUMPS;
			float s = 1;
			b.get_float("strength", s);
			t.strength = s;
		}
	}

	// ---------------------------------------------------------------- shaders
	void fill_layers(const Block& b, hxr_shader& s)   // Layered::fillProperties, src/shading.cpp:270-312
	{
		s.first_layer = int(fs.layers.size());
		s.num_layers = 0;
		for (const Property& p : b.props) {
			p.used = true;
			if (p.key != "layer") continue;
			const char* bad = "Expected a line like `layer <shader>, <color>[, <texture>]'";
			std::vector<std::string> tok = split_ws(p.value);
			if (tok.size() < 2) throw ParseError(p.line, bad);
			auto depunct = [](std::string t) {
				t.erase(std::remove(t.begin(), t.end(), ','), t.end());
				return t;
			};
			std::string shader_name = depunct(tok[0]);
			std::string rest = p.value.substr(p.value.find(tok[0]) + tok[0].size());
			std::string tex_name;
			if (!rest.empty() && rest.back() != ')') {   // a trailing token that is not the colour's ')' names the blend texture
				size_t cut = rest.find_last_of(" \t");
				if (cut == std::string::npos) throw ParseError(p.line, bad);
				tex_name = depunct(rest.substr(cut + 1));
				rest = rest.substr(0, cut + 1);
			}
			if (tex_name == "NULL") tex_name.clear();
			hxr_layer L;
			L.shader = index_of(fs.shader_names, shader_name);
			L.texture = -1;
			if (L.shader < 0) throw ParseError(p.line, bad);
			if (!tex_name.empty()) {
				L.texture = index_of(fs.texture_names, tex_name);
				if (L.texture < 0) throw ParseError(p.line, bad);
			}
			double x, y, z;
			parse_triple(p.line, rest, x, y, z);
			set_color(L.blend, float(x), float(y), float(z));
			fs.layers.push_back(L);
			s.num_layers++;
		}
	}

	void fill_shader(const Block& b, int si)
	{
		hxr_shader s = fs.shaders[si];
		double mult;
		if (b.cls == "Lambert" || b.cls == "Phong") {   // src/shading.h:95-123
			s.type = b.cls == "Phong" ? HXR_SHADER_PHONG : HXR_SHADER_LAMBERT;
			set_color(s.color, 0.5f, 0.5f, 0.5f);
			set_color(s.specular, 1, 1, 1);
			s.exponent = 10.0f;
			b.get_color("color", s.color);
			if (s.type == HXR_SHADER_PHONG) b.get_color("specular", s.specular);
			s.texture = texture_ref(b, "texture");
			if (s.type == HXR_SHADER_PHONG) b.get_float("exponent", s.exponent);
		} else if (b.cls == "Reflection") {             // src/shading.h:125-145
			s.type = HXR_SHADER_REFLECTION;
			set_color(s.color, 0.95f, 0.95f, 0.95f);
			s.glossiness = 1.0f;
			s.num_samples = 50;
			if (b.get_double("multiplier", mult)) set_color(s.color, float(mult), float(mult), float(mult));
			else b.get_color("reflColor", s.color);
			b.get_float("glossiness", s.glossiness, 0, 1);
			b.get_int("numSamples", s.num_samples, 1);
		} else if (b.cls == "Refraction") {             // src/shading.h:147-164
			s.type = HXR_SHADER_REFRACTION;
			set_color(s.color, 0.95f, 0.95f, 0.95f);
			s.ior = 1.33;
			if (b.get_double("multiplier", mult)) set_color(s.color, float(mult), float(mult), float(mult));
			else b.get_color("refrColor", s.color);
			b.get_double("ior", s.ior, 1e-6, 10);
		} else if (b.cls == "Layered") {
			s.type = HXR_SHADER_LAYERED;
			fill_layers(b, s);
		} else if (b.cls == "Const") {                  // src/shading.h:206-214
			s.type = HXR_SHADER_CONST;
			set_color(s.color, 0.5f, 0.5f, 0.5f);
			b.get_color("color", s.color);
		}
		fs.shaders[si] = s;
	}

	// ---------------------------------------------------------------- lights / nodes
	void fill_light(const Block& b, int li)   // src/lights.h:64-110, RectLight::beginFrame src/lights.cpp:77-89
	{
		hxr_light& L = fs.lights[li];
		set_color(L.color, 1, 1, 1);
		L.power = 1.0f;
		L.scale_factor = 1.0f;
		L.x_subd = L.y_subd = 3;
		b.get_color("color", L.color);
		b.get_float("power", L.power);
		Xform T;
		if (b.cls == "PointLight") {
			L.type = HXR_LIGHT_POINT;
			V3 pos;
			b.get_vec("pos", pos);
			store3(L.pos, pos);
		} else {
			L.type = HXR_LIGHT_RECT;
			b.apply_transform(T);
			b.get_int("xSubd", L.x_subd, 1, 1000);
			b.get_int("ySubd", L.y_subd, 1, 1000);
			V3 pts[3];
			for (int i = 0; i < 3; i++)
				pts[i] = T.point(V3((i % 2) ? -0.5 : 0.5, 0, (i < 2) ? -0.5 : 0.5));
			L.area = dist(pts[0], pts[1]) * dist(pts[0], pts[2]);
			L.scale_factor = float(1 / L.area);
		}
		T.store(L.T);
	}

	void fill_node(const Block& b, int ni, std::vector<hxr_node>& all_nodes)   // src/node.h:40-46
	{
		hxr_node& n = all_nodes[ni];
		n.geometry = n.shader = n.bump = -1;
		if (const Property* p = b.find("geometry")) {
			n.geometry = index_of(fs.geometry_names, p->value);
			if (n.geometry < 0) throw ParseError(p->line, "Geometry not defined");
		}
		if (const Property* p = b.find("shader")) {
			n.shader = index_of(fs.shader_names, p->value);
			if (n.shader < 0) throw ParseError(p->line, "Shader not defined");
		}
		Xform T;
		b.apply_transform(T);
		T.store(n.T);
		n.bump = texture_ref(b, "bump");
	}
};

}  // namespace

void CameraSpec::basis(hxr_camera& c) const   // Camera::beginFrame, src/camera.cpp:30-63
{
	V3 tl(-aspect, +1, +1);
	double corner = dist(V3(0, 0, 1), tl);
	double wanted = std::tan(radians(fov / 2));
	double k = wanted / corner;
	tl = V3(-aspect * k, +k, +1);
	V3 tr(+aspect * k, +k, +1), bl(-aspect * k, -k, +1);
	M3 R = rot_z(radians(roll)) * rot_x(radians(pitch)) * rot_y(radians(yaw));
	tl = tl * R + pos;
	tr = tr * R + pos;
	bl = bl * R + pos;
	V3 up = unit_lazy(tl - bl), right = unit_lazy(tr - tl);
	store3(c.pos, pos);
	store3(c.top_left, tl);
	store3(c.top_right, tr);
	store3(c.bottom_left, bl);
	store3(c.up, up);
	store3(c.right, right);
	store3(c.front, cross(right, up));
	c.aperture_size = 2.5 / f_number;
	c.focal_plane_dist = focal_plane_dist;
	c.stereo_separation = stereo_separation;
	c.dof = dof;
	c.num_samples = num_samples;
	c.auto_focus = auto_focus;
	c.pad_ = 0;
}

void CameraSpec::move(double sideways, double front_back)
{
	hxr_camera c;
	basis(c);
	pos = pos + V3(c.right[0], c.right[1], c.right[2]) * sideways + V3(c.front[0], c.front[1], c.front[2]) * front_back;
}

void CameraSpec::rotate(double yaw_diff, double pitch_diff)
{
	yaw += yaw_diff;
	pitch += pitch_diff;
	pitch = std::min(90.0, pitch);
	pitch = std::max(-90.0, pitch);
}

void FlatScene::refresh_camera() { camera_spec.basis(camera); }

void FlatScene::refresh_desc()
{
	memset(&desc, 0, sizeof desc);
	desc.abi_version = HXR_ABI_VERSION;
	desc.settings = settings;
	desc.camera = camera;
	desc.environment = environment;
#define HXH_TABLE(field, vec, count) desc.field = vec.empty() ? nullptr : vec.data(); desc.count = int64_t(vec.size());
	HXH_TABLE(geometries, geometries, num_geometries)
	HXH_TABLE(meshes, meshes, num_meshes)
	HXH_TABLE(heightfields, heightfields, num_heightfields)
	HXH_TABLE(triangles, triangles, num_triangles)
	HXH_TABLE(kd_nodes, kd_nodes, num_kd_nodes)
	HXH_TABLE(leaf_indices, leaf_indices, num_leaf_indices)
	HXH_TABLE(bitmaps, bitmaps, num_bitmaps)
	HXH_TABLE(textures, textures, num_textures)
	HXH_TABLE(shaders, shaders, num_shaders)
	HXH_TABLE(layers, layers, num_layers)
	HXH_TABLE(lights, lights, num_lights)
	HXH_TABLE(nodes, nodes, num_nodes)
	HXH_TABLE(f64_pool, f64_pool, f64_pool_size)
	HXH_TABLE(f32_pool, f32_pool, f32_pool_size)
#undef HXH_TABLE
}

std::unique_ptr<FlatScene> load_scene(const std::string& scene_path, const std::string& overlay_text, bool gpu_heightfield_tables, bool gpu_kd_build, bool gpu_prepare_triangles, bool gpu_heightfield_blur, bool gpu_bitmap_prep)
{
	std::ifstream in(scene_path);
	if (!in) throw std::runtime_error("Cannot open scene file `" + scene_path + "'!");
	std::stringstream buf;
	buf << in.rdbuf();
	std::vector<Block> blocks = parse_scene_text(buf.str(), scene_path);
	if (!overlay_text.empty()) merge_overlay(blocks, parse_scene_text(overlay_text, "<overlay>"));

	auto fs = std::make_unique<FlatScene>();
	fs->gpu_heightfield_tables = gpu_heightfield_tables || gpu_heightfield_blur;
	fs->gpu_heightfield_blur = gpu_heightfield_blur;
	fs->gpu_bitmap_prep = gpu_bitmap_prep;
	fs->gpu_kd_build = gpu_kd_build;
	fs->gpu_prepare_triangles = gpu_prepare_triangles;
	// defaults, src/scene.h:244-269
	set_color(fs->settings.ambient, 0.15f, 0.15f, 0.15f);
	set_color(fs->settings.background, 0, 0, 0);
	fs->settings.max_trace_depth = 8;
	fs->settings.want_aa = 1;
	fs->settings.gi = 0;
	fs->settings.num_paths = 32;
	fs->settings.frame_width = 800;
	fs->settings.frame_height = 600;
	memset(&fs->environment, 0, sizeof fs->environment);
	for (int& s : fs->environment.side_bitmap) s = -1;

	size_t slash = scene_path.find_last_of("/\\");
	std::string root = slash == std::string::npos ? "" : scene_path.substr(0, slash + 1);
	Loader L{*fs, root, blocks, {}};
	L.slot.assign(blocks.size(), -1);
	std::vector<hxr_node> all_nodes;
	std::vector<std::string> all_node_names;

	// pass 1: create elements in file order so that names resolve regardless of definition order
	for (size_t i = 0; i < blocks.size(); i++) {
		const Block& b = blocks[i];
		switch (kind_of(b.cls)) {
			case K_GEOMETRY: {
				hxr_geometry g;
				memset(&g, 0, sizeof g);
				g.left = g.right = g.ref = -1;
				L.slot[i] = int(fs->geometries.size());
				fs->geometries.push_back(g);
				fs->geometry_names.push_back(b.name);
				break;
			}
			case K_TEXTURE: {
				hxr_texture t;
				memset(&t, 0, sizeof t);
				t.bitmap = -1;
				L.slot[i] = int(fs->textures.size());
				fs->textures.push_back(t);
				fs->texture_names.push_back(b.name);
				break;
			}
			case K_SHADER: {
				hxr_shader s;
				memset(&s, 0, sizeof s);
				s.texture = -1;
				L.slot[i] = int(fs->shaders.size());
				fs->shaders.push_back(s);
				fs->shader_names.push_back(b.name);
				break;
			}
			case K_LIGHT: {
				hxr_light l;
				memset(&l, 0, sizeof l);
				L.slot[i] = int(fs->lights.size());
				fs->lights.push_back(l);
				break;
			}
			case K_NODE: {
				hxr_node n;
				memset(&n, 0, sizeof n);
				L.slot[i] = int(all_nodes.size());
				all_nodes.push_back(n);
				all_node_names.push_back(b.name);
				break;
			}
			case K_UNKNOWN:
				throw ParseError(b.line_begin, "Unknown object class `" + b.cls + "'");
			default: break;
		}
	}
	// pass 2: fill by element type in the reference's order (src/scene.cpp:533-557)
	static const Kind order[] = {K_SETTINGS, K_CAMERA, K_ENVIRONMENT, K_GEOMETRY, K_TEXTURE, K_SHADER, K_LIGHT, K_NODE};
	for (Kind kind : order)
		for (size_t i = 0; i < blocks.size(); i++) {
			const Block& b = blocks[i];
			if (kind_of(b.cls) != kind) continue;
			switch (kind) {
				case K_SETTINGS: L.fill_settings(b); break;
				case K_CAMERA: L.fill_camera(b); break;
				case K_ENVIRONMENT: L.fill_environment(b); break;
				case K_GEOMETRY: L.fill_geometry(b, L.slot[i]); break;
				case K_TEXTURE: L.fill_texture(b, L.slot[i]); break;
				case K_SHADER: L.fill_shader(b, L.slot[i]); break;
				case K_LIGHT: L.fill_light(b, L.slot[i]); break;
				case K_NODE: L.fill_node(b, L.slot[i], all_nodes); break;
				default: break;
			}
			for (const Property& p : b.props)
				if (!p.used) {
					char msg[256];
					snprintf(msg, sizeof msg, "%s:%d: Warning: the property `%s' isn't recognized!", scene_path.c_str(), p.line, p.key.c_str());
					fs->warnings.push_back(msg);
				}
		}
	if (!fs->camera_spec.present) throw std::runtime_error("scene has no Camera");
	// shader-less nodes are not rendered (src/scene.cpp:558-565)
	for (size_t i = 0; i < all_nodes.size(); i++) {
		if (all_nodes[i].shader < 0) continue;
		if (all_nodes[i].geometry < 0) throw std::runtime_error("Node `" + all_node_names[i] + "' has no geometry");
		fs->nodes.push_back(all_nodes[i]);
		fs->node_names.push_back(all_node_names[i]);
	}
	fs->refresh_camera();
	fs->refresh_desc();
	return fs;
}

std::vector<int> bucket_list(int width, int height, int bucket_size)
{
	std::vector<int> r;
	int BW = (width - 1) / bucket_size + 1, BH = (height - 1) / bucket_size + 1;
	for (int y = 0; y < BH; y++)
		for (int i = 0; i < BW; i++) {
			int x = (y % 2) ? BW - 1 - i : i;   // odd rows run right-to-left
			r.push_back(x * bucket_size);
			r.push_back(y * bucket_size);
			r.push_back(std::min((x + 1) * bucket_size, width));
			r.push_back(std::min((y + 1) * bucket_size, height));
		}
	return r;
}

}  // namespace hxh
