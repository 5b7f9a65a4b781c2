// libhexray_hip.so -- host side of the C ABI declared in include/hexray_hip.h.
// Uploads the flattened scene into HBM in the kernels' layout (hxr_device_scene.h), launches the
// render / AA-detect / assemble kernels on the caller's stream and times the render kernel with HIP
// events.  One process drives one GPU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>
#include "hexray_hip.h"
#include "hxr_kernels.hip.h"
#include "hxr_accel.h"
#include "hxr_kd_build.h"

#include "hxr_variants.h"
// compiled in hxr_kernels_mesh.hip / hxr_kernels_mesh2.hip / hxr_kernels_heightfield.hip with their own flags
HXR_VARIANTS_MESH_DEFER(HXR_DECLARE_EXTERN)
HXR_VARIANTS_MESH_INPLACE(HXR_DECLARE_EXTERN_LAUNCH_MODE)
HXR_VARIANTS_MESH_INPLACE_MC_THROUGHPUT(HXR_DECLARE_EXTERN_MONTE_CARLO)
HXR_VARIANTS_MESH_INPLACE_MC(HXR_DECLARE_EXTERN_MONTE_CARLO)
HXR_VARIANTS_MESH_PARTS(HXR_DECLARE_EXTERN)
HXR_VARIANTS_MESH_STACK1(HXR_DECLARE_EXTERN)
HXR_VARIANTS_MESH_INPLACE(HXR_DECLARE_EXTERN_CONTINUE)
HXR_VARIANTS_MESH_LANEWALK(HXR_DECLARE_EXTERN_CONTINUE)
HXR_VARIANTS_MESH_OTHER(HXR_DECLARE_EXTERN)
HXR_VARIANTS_MESH_GI(HXR_DECLARE_EXTERN)
HXR_VARIANTS_HEIGHTFIELD(HXR_DECLARE_EXTERN)
HXR_VARIANTS_NESTED_CSG(HXR_DECLARE_EXTERN)

namespace {

thread_local char g_err[512] = "";
int g_device = -1;
int g_num_cu = 256;

void set_err(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
			return HXR_E_HIP;                                                                      \
		}                                                                                          \
	} while (0)

struct Arena {   // every device allocation of a scene, freed together
	std::vector<void*> blocks;
	template <class T>
	int upload(const T* host, size_t n, const T** out)
	{
		*out = nullptr;
		if (!n) return 0;
		void* p = nullptr;
		HIP_TRY(hipMalloc(&p, n * sizeof(T)));
		blocks.push_back(p);
		HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
		*out = (const T*)p;
		return 0;
	}
	int alloc(size_t bytes, void** out)
	{
		HIP_TRY(hipMalloc(out, bytes));
		blocks.push_back(*out);
		return 0;
	}
	void release()
	{
		for (void* p : blocks) (void)hipFree(p);
		blocks.clear();
	}
};

}  // namespace

struct hxr_scene {
	Arena arena;
	HxrSceneDev host_view;          // device pointers inside, host copy of the header
	HxrSceneDev* dev = nullptr;     // the same struct in HBM
	hxr_settings settings;
	hxr_camera camera;
	unsigned features = 0;          // HXR_F_* geometry kinds present among the nodes
	int num_cu = 256;               // compute units of THIS scene's device (grids are sized per scene: a group may mix devices)
	struct TreeStats { int64_t nodes, leaf_entries; int depth; };
	std::vector<TreeStats> tree_stats;   // per mesh: the tree actually walked
	double kd_build_ms = 0;              // HXR_ACCEL_REFERENCE_DEVICE_BUILD: time of the kd builds on the GPU, all meshes
	const hxr_triangle* d_triangles = nullptr;   // the triangle records in HBM (hxr_scene_verify_tables reads them back)
	std::vector<int> bitmap_prep;        // per bitmap: hxr_bitmap::prep it was created from (| 4: gamma curve evaluated on the device)
	std::vector<char> hf_raw;            // per heightfield: created from hxr_heightfield::raw_heights (blur / box / normals computed here)
	// per-frame scratch, grown on demand
	float* d_frame = nullptr;       size_t frame_bytes = 0;    // hxr_render staging / AA scratch frame
	float* d_scratch = nullptr;     size_t scratch_bytes = 0;  // sharded-AA full frame
	int32_t* d_aa_list = nullptr;   size_t aa_capacity = 0;
	int32_t* d_extra = nullptr;     size_t extra_capacity = 0;
	uint32_t* d_work = nullptr;     // [0..255] dispenser shards pass 1 / MC, [256..511] dispenser shards pass 3, [512] aa count
	unsigned long long* d_counters = nullptr;
	// adaptive schedule (see HxrLaunch::sched_*): cost history of the last frame and what it was a frame of
	bool adaptive = true;
	uint32_t* d_sched_cost = nullptr;   size_t sched_cost_cap = 0;     // chunks
	uint8_t* d_sched_flags = nullptr;   size_t sched_flags_cap = 0;    // chunks
	uint32_t* d_sched_items = nullptr;  size_t sched_items_cap = 0;    // pixels + room for the sample-split parts
	float* d_sched_side = nullptr;      size_t sched_side_cap = 0;     // HXR_PART_SLOTS x lanes per pixel x 3
	uint32_t* d_sched_part = nullptr;                                  // [0..SLOTS) parts done, [SLOTS..2*SLOTS) pixel of the slot
	unsigned long long* d_sched_ctl = nullptr;                         // [0] total cost, then as uint32: [2] listed from the front, [3] from the back, [4] part slots, [5] ticket
	struct SchedKey { int W, H, mode, spp, tile, rank, count; uint32_t chunks; } sched_key = {0, 0, 0, 0, 0, 0, 0, 0};
	// the replicating blocks (HXR_F_REPL instance) that take the schedule's sample parts and single pixels beside the kernel that renders the rest
	uint32_t* d_sched_items_r = nullptr; size_t sched_items_r_cap = 0;
	hipStream_t parts_stream = nullptr;
	hipEvent_t ev_parts_go = nullptr, ev_parts_done = nullptr;
	HxrLaunch parts_launch;             // its launch record, prepared with the schedule
	bool parts_pending = false;         // ... and to be launched with the next render kernel
	bool frame_counts = false;          // the frame being rendered collects work counters (the counting kernels take every packet themselves)
	bool sched_history = false;
	bool sched_frame = false;           // the last frame went through prepare_schedule (hxr_frame_timing::sched_*)
	bool sched_sum_valid = false;       // ... and d_sched_ctl[0] holds the cost total it was scheduled with (there was a usable cost map)
	bool aa_cost_aware = false;         // the last frame was an adaptive-AA frame whose flagged list has a back part
	uint8_t* d_lut = nullptr;
	// deferred secondary rays (HXR_F_DEFER variants; plan_defer)
	bool defer_allowed = false;         // scene option hxr_scene_options::deferred_rays / HXR_DEFER=1
	int fanout0 = 0;                    // most Reflection / Refraction rays one hit can spawn (over all nodes)
	HxrRayRec* d_queue = nullptr;       size_t queue_bytes = 0;
	uint32_t* d_qctl = nullptr;         // two control blocks (pass 1 / Monte-Carlo, pass 3) of QCTL_WORDS words
	long long* d_accum = nullptr;       size_t accum_bytes = 0;
	bool accum_dirty = false;           // a frame stopped between the continuation pass and the resolve
	bool no_defer_this_frame = false;   // the prepass fills blocks, not pixels: its rays are traced in place
	struct HotSphere { double c[3], r; };
	std::vector<HotSphere> hot_spheres; // world bounding spheres of the bounded nodes whose shaders spawn secondary rays (first-frame schedule)
	bool hot_everywhere = false;        // an unbounded node (a plane) spawns secondary rays: no region stands out
	// hxr_render() with a cancel flag or a progress callback: the frame runs on `main_stream` while the calling thread watches
	// the caller's flag and the per-tile pixel counts through `aux_stream`
	hipStream_t main_stream = nullptr, aux_stream = nullptr;
	uint32_t* d_watch = nullptr;        size_t watch_words = 0;   // [0] cancel word, [16 ...] pixels written per shard tile
	uint32_t* h_watch = nullptr;                                  // pinned host copy of the tile counts
	uint32_t* h_cancel = nullptr;                                 // the cancel word: pinned host memory mapped into the device ...
	uint32_t* d_cancel = nullptr;                                 // ... and its device-side address
	hipEvent_t ev_done = nullptr;
	const uint32_t* watch_cancel = nullptr;                       // what the next render_on_stream puts into HxrLaunch (null: nothing watched)
	uint32_t* watch_tiles = nullptr;
	int device = 0;
	uint32_t* d_surface = nullptr;      size_t surface_bytes = 0;   // hxr_interactive_frame: the 32-bit display pixels before they leave
	hipEvent_t ev0 = nullptr, ev1 = nullptr;   // the frame's first and last event: ALIASES of ev_pass[..][0] / [..][3] (an event record is a barrier packet: ~5 us of the stream each)
	hipEvent_t ev_pass[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // per pass: before render, after render, after continuation, after resolve
	int passes_timed = 0;
	bool pass_deferred[2] = {false, false};
	bool timed = false;
	bool counters_valid = false;
	uint64_t aa_pixels_host = 0;
};

namespace {

// How many secondary rays one hit of `shader` can leave pending, and how many (shader, weight) pairs
// its expansion can hold at once -- checked against the kernels' fixed per-lane stacks.
void shader_fanout(const hxr_scene_desc* d, int si, int depth, int& rays, int& pending, bool& glossy, bool& cyclic)
{
	if (depth > 16) { cyclic = true; return; }
	if (si < 0 || si >= d->num_shaders) return;   // validate() has range-checked every table reference before it walks the trees
	const hxr_shader& s = d->shaders[si];
	if (s.type == HXR_SHADER_REFLECTION) { rays += 1; if (s.glossiness < 1.0f) glossy = true; }
	else if (s.type == HXR_SHADER_REFRACTION) rays += 1;
	else if (s.type == HXR_SHADER_LAYERED) {
		pending += s.num_layers;
		for (int i = 0; i < s.num_layers; i++) shader_fanout(d, d->layers[s.first_layer + i].shader, depth + 1, rays, pending, glossy, cyclic);
	}
}

// depth of the operation tree below CSG geometry `gi` (an operation over primitives = 1); -1: an operand index outside the table,
// -2: an operand that is neither an analytic primitive nor an operation; anything deeper than 16 (a cycle) counts as 17
int csg_depth(const hxr_scene_desc* d, int gi, int level)
{
	if (level > 16) return 17;
	const hxr_geometry& g = d->geometries[gi];
	int deepest = level;
	for (int side = 0; side < 2; side++) {
		const int op = side ? g.right : g.left;
		if (op < 0 || op >= d->num_geometries) return -1;
		const int t = d->geometries[op].type;
		if (t == HXR_GEOM_CSG_UNION || t == HXR_GEOM_CSG_INTER || t == HXR_GEOM_CSG_DIFF) {
			const int below = csg_depth(d, op, level + 1);
			if (below < 0) return below;
			deepest = std::max(deepest, below);
		} else if (t != HXR_GEOM_PLANE && t != HXR_GEOM_SPHERE && t != HXR_GEOM_CUBE) {
			return -2;
		}
	}
	return deepest;
}

int validate(const hxr_scene_desc* d)
{
	if (!d || d->abi_version != HXR_ABI_VERSION) { set_err("bad descriptor / ABI version"); return HXR_E_INVALID; }
	if (d->num_nodes < 0 || d->num_nodes > 4096) { set_err("bad node count"); return HXR_E_INVALID; }
	if (d->num_shaders < 0 || d->num_layers < 0 || d->num_textures < 0 || d->num_bitmaps < 0 || d->num_geometries < 0 || d->num_lights < 0) {
		set_err("negative table size");
		return HXR_E_INVALID;
	}
	// every table-to-table reference is range-checked BEFORE anything follows one (the node loop below walks shader trees)
	for (int64_t i = 0; i < d->num_textures; i++) {
		const hxr_texture& t = d->textures[i];
		if ((t.type == HXR_TEX_BITMAP || t.type == HXR_TEX_BUMPTEX) && (t.bitmap < 0 || t.bitmap >= d->num_bitmaps)) { set_err("bad bitmap reference"); return HXR_E_INVALID; }
	}
	for (int64_t i = 0; i < d->num_shaders; i++) {
		const hxr_shader& s = d->shaders[i];
		if (s.texture >= d->num_textures) { set_err("bad texture reference"); return HXR_E_INVALID; }
		if (s.type == HXR_SHADER_LAYERED) {
			if (s.num_layers < 0 || s.first_layer < 0 || (int64_t)s.first_layer + s.num_layers > d->num_layers) {
				set_err("Layered shader %lld: layers %d..%d outside the layer table", (long long)i, s.first_layer, s.first_layer + s.num_layers);
				return HXR_E_INVALID;
			}
			for (int k = 0; k < s.num_layers; k++) {
				const hxr_layer& L = d->layers[s.first_layer + k];
				if (L.shader < 0 || L.shader >= d->num_shaders || L.texture >= d->num_textures) { set_err("bad layer reference"); return HXR_E_INVALID; }
			}
		}
	}
	// meshes: every index a kernel will follow (vertex / normal / uv numbers of the triangles, kd children, leaf lists) stays inside its table
	if (d->num_meshes < 0 || d->num_triangles < 0 || d->num_kd_nodes < 0 || d->num_leaf_indices < 0 || d->f64_pool_size < 0 || d->f32_pool_size < 0) { set_err("negative table size"); return HXR_E_INVALID; }
	for (int64_t i = 0; i < d->num_meshes; i++) {
		const hxr_mesh& m = d->meshes[i];
		auto vectors_ok = [&](int64_t first, int64_t n) { return first >= 0 && n >= 0 && first <= d->f64_pool_size / 3 && n <= d->f64_pool_size / 3 - first; };
		auto range_ok = [](int64_t first, int64_t n, int64_t table) { return first >= 0 && n >= 0 && first <= table && n <= table - first; };
		if (!vectors_ok(m.first_vertex, m.num_vertices) || !vectors_ok(m.first_normal, m.num_normals) || !vectors_ok(m.first_uv, m.num_uvs) ||
		    !range_ok(m.first_triangle, m.num_triangles, d->num_triangles) || !range_ok(m.first_kd_node, m.num_kd_nodes, d->num_kd_nodes) ||
		    !range_ok(m.first_leaf_index, m.num_leaf_indices, d->num_leaf_indices) || m.num_triangles > INT32_MAX || m.num_kd_nodes > INT32_MAX) {
			set_err("mesh %lld: a table range lies outside the descriptor", (long long)i);
			return HXR_E_INVALID;
		}
		for (int64_t k = 0; k < m.num_triangles; k++) {
			const hxr_triangle& t = d->triangles[m.first_triangle + k];
			for (int c = 0; c < 3; c++)
				if (t.v[c] < 0 || t.v[c] >= m.num_vertices || t.n[c] < 0 || (!m.faceted && t.n[c] >= m.num_normals) || t.t[c] < 0 || t.t[c] >= m.num_uvs) {
					set_err("mesh %lld: triangle %lld refers outside the vertex / normal / uv arrays", (long long)i, (long long)k);
					return HXR_E_INVALID;
				}
		}
		for (int64_t k = 0; k < m.num_kd_nodes; k++) {   // children are stored after their parent (every builder here numbers them so): no cycles
			const hxr_kd_node& n = d->kd_nodes[m.first_kd_node + k];
			const bool ok = n.axis == 3 ? (n.first >= 0 && n.count >= 0 && (int64_t)n.first + n.count <= m.num_leaf_indices)
			                            : (n.axis >= 0 && n.axis <= 2 && n.first > k && (int64_t)n.first + 1 < m.num_kd_nodes);
			if (!ok) { set_err("mesh %lld: kd node %lld is malformed", (long long)i, (long long)k); return HXR_E_INVALID; }
		}
		for (int64_t k = 0; k < m.num_leaf_indices; k++) {
			const int32_t t = d->leaf_indices[m.first_leaf_index + k];
			if (t < 0 || t >= m.num_triangles) { set_err("mesh %lld: leaf entry %lld names triangle %d of %lld", (long long)i, (long long)k, t, (long long)m.num_triangles); return HXR_E_INVALID; }
		}
	}
	if (d->num_heightfields < 0) { set_err("negative table size"); return HXR_E_INVALID; }
	for (int64_t i = 0; i < d->num_heightfields; i++) {
		const hxr_heightfield& h = d->heightfields[i];
		const int64_t cells = (int64_t)h.W * h.H;
		auto floats_ok = [&](int64_t first, int64_t n) { return first >= 0 && first <= d->f32_pool_size && n <= d->f32_pool_size - first; };
		const bool tables = h.first_maxh >= 0 && h.first_highmap >= 0;
		const bool normals_ok = h.raw_heights ? true : (h.first_normal >= 0 && h.first_normal <= d->f64_pool_size / 3 && cells <= d->f64_pool_size / 3 - h.first_normal);
		const bool raw_ok = !h.raw_heights || (!tables && h.blur_radius >= 0 && h.blur_radius <= 128 && (h.blur_radius == 0 || floats_ok(h.first_blur_kernel, (int64_t)h.blur_radius * h.blur_radius)));
		if (h.W <= 0 || h.H <= 0 || cells > (1ll << 30) || !floats_ok(h.first_height, cells) || (tables && (!floats_ok(h.first_maxh, cells) || (h.use_optimization && !floats_ok(h.first_highmap, cells * 16)))) ||
		    h.max_k < 0 || h.max_k > 16 || !normals_ok || !raw_ok) {
			set_err("heightfield %lld: a table range lies outside the descriptor", (long long)i);
			return HXR_E_INVALID;
		}
	}
	for (int64_t i = 0; i < d->num_bitmaps; i++) {
		const hxr_bitmap& b = d->bitmaps[i];
		const int64_t floats = (int64_t)b.width * b.height * 3;
		if (b.width < 0 || b.height < 0 || b.first_texel < 0 || b.first_texel > d->f32_pool_size || floats > d->f32_pool_size - b.first_texel) {
			set_err("bitmap %lld: texels outside the descriptor", (long long)i);
			return HXR_E_INVALID;
		}
		if ((b.prep & ~(HXR_BITMAP_GAMMA | HXR_BITMAP_DIFFERENTIATE)) != 0 || ((b.prep & HXR_BITMAP_GAMMA) && !(b.gamma > 0 && b.gamma <= 100.0f)) ||
		    ((b.prep & HXR_BITMAP_GAMMA) && b.first_gamma_table != -1 && (b.first_gamma_table < 0 || b.first_gamma_table > d->f32_pool_size - 256))) {
			set_err("bitmap %lld: bad prep / gamma / gamma table", (long long)i);
			return HXR_E_INVALID;
		}
	}
	if (d->environment.present && d->environment.loaded)
		for (int side = 0; side < 6; side++)
			if (d->environment.side_bitmap[side] < 0 || d->environment.side_bitmap[side] >= d->num_bitmaps) { set_err("environment side %d: bad bitmap reference", side); return HXR_E_INVALID; }
	for (int64_t i = 0; i < d->num_nodes; i++) {
		const hxr_node& n = d->nodes[i];
		if (n.geometry < 0 || n.geometry >= d->num_geometries || n.shader < 0 || n.shader >= d->num_shaders || n.bump >= d->num_textures) {
			set_err("node %lld refers outside the tables", (long long)i);
			return HXR_E_INVALID;
		}
		const hxr_geometry& g = d->geometries[n.geometry];
		if (g.type == HXR_GEOM_CSG_UNION || g.type == HXR_GEOM_CSG_INTER || g.type == HXR_GEOM_CSG_DIFF) {
			// operands: analytic primitives or further operations, HXR_CSG_DEPTH operations deep at most (the kernels unroll the recursion
			// of CSGBase::intersect at compile time).  CSG over a kd-tree mesh is undefined in the reference itself: CSGBase::intersect tells
			// the operands apart by `infoCandidate.geom == left` (src/geometry.cpp:186) and Mesh::intersectKD never sets IntersectionInfo::geom
			// (src/mesh.cpp:247-263: only the brute-force branch does), so the comparison reads an uninitialised pointer; it is refused.
			const int depth = csg_depth(d, n.geometry, 1);
			if (depth == -1) { set_err("CSG operand outside the geometry table"); return HXR_E_INVALID; }
			if (depth == -2) { set_err("CSG (node %lld) over a mesh or a heightfield is not rendered by this build", (long long)i); return HXR_E_UNSUPPORTED; }
			if (depth > HXR_CSG_DEPTH) { set_err("CSG of node %lld is nested %d operations deep (limit %d)", (long long)i, depth, HXR_CSG_DEPTH); return HXR_E_UNSUPPORTED; }
		}
		if (g.type == HXR_GEOM_MESH && (g.ref < 0 || g.ref >= d->num_meshes)) { set_err("bad mesh reference"); return HXR_E_INVALID; }
		if (g.type == HXR_GEOM_HEIGHTFIELD && (g.ref < 0 || g.ref >= d->num_heightfields)) { set_err("bad heightfield reference"); return HXR_E_INVALID; }
		int rays = 0, pending = 1;
		bool glossy = false, cyclic = false;
		shader_fanout(d, n.shader, 0, rays, pending, glossy, cyclic);
		if (cyclic) { set_err("Layered shaders of node %lld refer to themselves", (long long)i); return HXR_E_UNSUPPORTED; }
		if (pending > HXR_SHADER_STACK) { set_err("Layered shader of node %lld is wider than %d entries", (long long)i, HXR_SHADER_STACK); return HXR_E_UNSUPPORTED; }
		long need = rays <= 1 ? 2 : (long)(d->settings.max_trace_depth + 1) * (rays - 1) + 2;
		if (need > HXR_RAY_STACK) {
			set_err("node %lld can leave %ld secondary rays pending (limit %d)", (long long)i, need, HXR_RAY_STACK);
			return HXR_E_UNSUPPORTED;
		}
	}
	return 0;
}

// does anything reachable from this shader read a texture (and with it IntersectionInfo::u / v)?
bool shader_reads_textures(const hxr_scene_desc* d, int shader, int depth)
{
	if (shader < 0 || shader >= d->num_shaders || depth > 16) return true;   // be conservative about what validate() did not follow
	const hxr_shader& sh = d->shaders[shader];
	if (sh.texture >= 0) return true;
	if (sh.type == HXR_SHADER_LAYERED)
		for (int i = 0; i < sh.num_layers; i++) {
			const hxr_layer& L = d->layers[sh.first_layer + i];
			if (L.texture >= 0 || shader_reads_textures(d, L.shader, depth + 1)) return true;
		}
	return false;
}

// Slim kd records + node boxes + leaf-ordered hot triangles of one mesh, appended to the scene-wide arrays
// (see hxr_device_scene.h)
// `nodes` / `num_nodes` / `leaf_idx`: the tree to lay out (the descriptor's, or an HXR_ACCEL_SAH rebuild)
int append_mesh_tables(const hxr_scene_desc* d, const hxr_mesh& m, const hxr_kd_node* nodes, int64_t num_nodes, const int32_t* leaf_idx, HxrMeshDev& o,
                       std::vector<HxrKdSlim>& kd, std::vector<HxrKdBox>& boxes, std::vector<HxrTriHot>& hot, std::vector<int32_t>& ids)
{
	int max_depth = 0;
	const hxr_triangle* tris = d->triangles + m.first_triangle;
	const double* verts = d->f64_pool + m.first_vertex * 3;
	if (kd.size() % 2 == 0) {   // roots sit at odd global indices: every child pair is then 32-byte aligned
		kd.push_back(HxrKdSlim{0.0, 0, HXR_KD_LEAF});
		boxes.push_back(HxrKdBox{{0, 0, 0}, {0, 0, 0}});
	}
	o.kd_first = (int32_t)kd.size();
	o.hot_first = (int32_t)ids.size();
	const size_t hot0 = ids.size();
	(void)tris; (void)verts;
	// only the ORDER of the hot records is decided here (triangle numbers in leaf order); the 96-byte records themselves are
	// gathered on the device from the cold tables (hxr_expand_hot_kernel)
	auto push_hot = [&](int ti) { ids.push_back(ti); };
	if (!m.has_kd || num_nodes == 0) {
		for (int64_t i = 0; i < m.num_triangles; i++) push_hot((int)i);
		return 0;
	}
	const size_t kd0 = kd.size();
	kd.resize(kd0 + (size_t)num_nodes);
	boxes.resize(kd0 + (size_t)num_nodes);
	o.boxes_nested = 1;
	struct Item { int node; int depth; double lo[3], hi[3]; };
	std::vector<Item> todo;
	Item root;
	root.node = 0;
	root.depth = 0;
	for (int c = 0; c < 3; c++) { root.lo[c] = m.bbox_min[c]; root.hi[c] = m.bbox_max[c]; }
	todo.push_back(root);
	while (!todo.empty()) {
		Item it = todo.back();
		todo.pop_back();
		const hxr_kd_node& n = nodes[it.node];
		HxrKdSlim& r = kd[kd0 + (size_t)it.node];
		HxrKdBox& rb = boxes[kd0 + (size_t)it.node];
		for (int c = 0; c < 3; c++) { rb.lo[c] = it.lo[c]; rb.hi[c] = it.hi[c]; }
		r.split = n.split;
		if (n.axis == HXR_KD_LEAF) {
			r.first = (int32_t)(ids.size() - hot0);
			r.axis_count = HXR_KD_LEAF | (n.count << 2);
			for (int i = 0; i < n.count; i++) push_hot(leaf_idx[n.first + i]);
			continue;
		}
		r.first = n.first;
		r.axis_count = n.axis;
		// every split position inside its node's box <=> every node's box inside its parent's: mesh_query's shortcut for the
		// inflated inside() test relies on it (true for the reference's midpoints and for the SAH builder's planes)
		if (!(n.split >= it.lo[n.axis] && n.split <= it.hi[n.axis])) o.boxes_nested = 0;
		Item L = it, R = it;   // BBox::split: the children copy the box and replace one bound by the split position
		L.node = n.first;
		R.node = n.first + 1;
		L.depth = R.depth = it.depth + 1;
		if (L.depth > max_depth) max_depth = L.depth;
		L.hi[n.axis] = n.split;
		R.lo[n.axis] = n.split;
		todo.push_back(R);
		todo.push_back(L);   // left first, so hot records follow the builder's leaf order
	}
	return max_depth;
}

int ensure(void** p, size_t* have, size_t need)
{
	if (*have >= need && *p) return 0;
	if (*p) (void)hipFree(*p);
	*p = nullptr;
	*have = 0;
	HIP_TRY(hipMalloc(p, need));
	*have = need;
	return 0;
}

struct Plan {   // resolved frame parameters
	int W, H, mode, spp, tile, tiles_x, tiles_y, rank, count, ntiles, layout;
	int fovea, vip_x, vip_y;   // foveated rendering (modes SINGLE / ADAPTIVE_AA): radius (0 = off) and point of interest
	uint64_t seed;
};

int resolve(const hxr_scene* s, const hxr_frame_params* fp, Plan& p)
{
	if (!fp) { set_err("null params"); return HXR_E_INVALID; }
	if (!s && (fp->width <= 0 || fp->height <= 0 || fp->mode == HXR_MODE_AUTO)) {   // tile bookkeeping may be asked without a scene
		set_err("without a scene the frame size and mode must be explicit");
		return HXR_E_INVALID;
	}
	p.W = fp->width > 0 ? fp->width : s->settings.frame_width;
	p.H = fp->height > 0 ? fp->height : s->settings.frame_height;
	if (p.W <= 0 || p.H <= 0 || p.W > 1920 || p.H > 1920) { set_err("frame %dx%d outside 1..1920 (VFB_MAX_SIZE)", p.W, p.H); return HXR_E_INVALID; }
	p.mode = fp->mode;
	p.spp = fp->spp;
	int rpp = 0;   // render(): raysPerPixel = max(dof ? numSamples : 0, gi ? numPaths : 0)  (src/main.cpp:421-424)
	if (s && s->camera.dof) rpp = s->camera.num_samples;
	if (s && s->settings.gi) rpp = std::max(rpp, s->settings.num_paths);
	if (p.mode == HXR_MODE_AUTO) {
		if (rpp > 0) { p.mode = HXR_MODE_MONTE_CARLO; if (p.spp <= 0) p.spp = rpp; }
		else p.mode = (s && s->settings.want_aa) ? HXR_MODE_ADAPTIVE_AA : HXR_MODE_SINGLE;
	}
	if (p.mode == HXR_MODE_MONTE_CARLO) { if (p.spp <= 0) p.spp = rpp > 0 ? rpp : 1; }
	else if (p.mode == HXR_MODE_SINGLE || p.mode == HXR_MODE_ADAPTIVE_AA) p.spp = 1;
	else if (p.mode == HXR_MODE_PREPASS) { if (p.spp <= 0) { set_err("HXR_MODE_PREPASS needs spp = prepassSamples > 0"); return HXR_E_INVALID; } }
	else { set_err("unknown render mode %d", p.mode); return HXR_E_INVALID; }
	if (p.spp > 65536) { set_err("spp too large"); return HXR_E_INVALID; }
	p.tile = fp->tile_size > 0 ? fp->tile_size : 64;
	if (p.tile % 8 || p.tile > 1024) { set_err("tile_size must be a multiple of 8"); return HXR_E_INVALID; }
	p.tiles_x = (p.W - 1) / p.tile + 1;
	p.tiles_y = (p.H - 1) / p.tile + 1;
	p.count = fp->shard_count > 1 ? fp->shard_count : 1;
	p.rank = fp->shard_count > 1 ? fp->shard_rank : 0;
	if (p.rank < 0 || p.rank >= p.count) { set_err("shard_rank outside shard_count"); return HXR_E_INVALID; }
	int total = p.tiles_x * p.tiles_y;
	p.ntiles = total > p.rank ? (total - p.rank + p.count - 1) / p.count : 0;
	p.layout = fp->layout;
	if (p.layout != HXR_LAYOUT_FRAME && p.layout != HXR_LAYOUT_TILES) { set_err("unknown layout"); return HXR_E_INVALID; }
	p.seed = fp->rng_seed;
	// renderWithMonteCarlo has no foveated branch (src/main.cpp:347-384): the fields only matter to the other two modes
	p.fovea = (p.mode == HXR_MODE_SINGLE || p.mode == HXR_MODE_ADAPTIVE_AA) ? fp->foveated_radius : 0;
	p.vip_x = fp->vip_x;
	p.vip_y = fp->vip_y;
	if (p.fovea < 0 || p.fovea > 1000) { set_err("foveated_radius outside 0..1000"); return HXR_E_INVALID; }   // GlobalSettings::fillProperties, src/scene.cpp:764-778
	if (p.mode == HXR_MODE_PREPASS && (p.count > 1 || p.layout != HXR_LAYOUT_FRAME)) { set_err("the prepass is an unsharded FRAME-layout frame"); return HXR_E_INVALID; }
	if (p.fovea > 0 && p.count > 1) { set_err("foveated frames are not sharded (an interactive, single-GPU mode)"); return HXR_E_UNSUPPORTED; }
	return 0;
}

void fill_launch(const hxr_scene* s, const Plan& p, HxrLaunch& L)
{
	memset(&L, 0, sizeof L);
	L.W = p.W; L.H = p.H;
	L.seed = p.seed;
	L.tile_size = p.tile; L.tiles_x = p.tiles_x; L.tiles_y = p.tiles_y;
	L.shard_rank = p.rank; L.shard_count = p.count; L.num_shard_tiles = p.ntiles;
	L.aa_count = (const int32_t*)(s->d_work + 512);
	L.cancel = s->watch_cancel;
	L.tile_done = s->watch_tiles;
	L.fov_radius = p.fovea; L.vip_x = p.vip_x; L.vip_y = p.vip_y;
	L.no_replication = getenv("HXR_NO_REPLICATION") ? 1 : 0;
}

int parts_grid(const hxr_scene* s)   // blocks of the replicating instance launched beside the main kernel: three sixteenths of the machine at three waves per SIMD
{
	if (const char* e = getenv("HXR_PARTS_GRID")) return std::max(1, atoi(e));   // developer aid
	return std::max(1, s->num_cu * HXR_MESH_WAVES * 3 / 16 * (256 / HXR_BLOCK));   // (counted in blocks of HXR_BLOCK threads: 144 of four waves)  slowest 8-way rank: 1.84 ms at 48 blocks, 1.24-1.34 at 96, 1.22 at 144, 1.20-1.21 at 192 (sum of ranks 9.5 / 9.5 / 9.6)
}

int resident_waves_per_cu(const hxr_scene* s, int mode, bool in_place, uint32_t num_chunks);
int grid_for(const hxr_scene* s, uint32_t chunks, int blocks_per_cu)
{
	// persistent waves: exactly the blocks a CU holds at the variant's occupancy (`blocks_per_cu`; end of round 4), never more than the work.  It used to be eight
	// per CU whatever the variant: the blocks beyond the resident ones start as the first ones leave, find the dispensers empty and leave -- at the very time the
	// frame waits for its last packets (headline 6.35 -> 6.30 ms at four per CU, 6.37 at sixteen; heightfield16 7.81 -> 7.76 at three; one block too few costs 14 %)
	uint32_t blocks = (uint32_t)s->num_cu * (uint32_t)std::max(1, blocks_per_cu);
	if (const char* e = getenv("HXR_GRID_PER_CU")) blocks = (uint32_t)s->num_cu * (uint32_t)std::max(1, atoi(e));   // developer aid
	uint32_t need = (chunks + (HXR_BLOCK / 64) - 1) / (HXR_BLOCK / 64);
	return (int)std::max(1u, std::min(blocks, need));
}

// `timed`: TIME_BEGIN records the start event before the launch, TIME_END the stop event after it -- hxr_last_kernel_ms() spans every
// kernel of the frame (all three passes of an adaptive-AA frame), not just the first
enum { TIME_BEGIN = 1, TIME_END = 2 };
enum { QCTL_WORDS = (2 * HXR_QUEUE_SHARDS + 1) * HXR_SHARD_STRIDE };
static_assert(HXR_SHARDS == HXR_QUEUE_SHARDS, "one sub-queue per dispenser range");

// Smallest compiled variant that covers the scene's geometry kinds and integrator (see HXR_F_* in hxr_kernels.hip.h); occupancy
// target per variant: measured (see hxr_variants.h and the note above hxr_render_kernel).
enum Variant { V_MESH, V_MESH_REFRACT, V_HEIGHTFIELD, V_SPHERE_GI, V_MESH_GI, V_ALL_GI_3, V_ALL_GI_2, V_ALL_3, V_ALL_2, V_NESTED_GI, V_NESTED };
Variant pick_variant(const hxr_scene* s)
{
	unsigned geom = s->features & (HXR_F_MESH | HXR_F_HEIGHTFIELD | HXR_F_SPHERE | HXR_F_CUBE | HXR_F_EXTRA);
	if (s->camera.stereo_separation != 0.0) geom |= HXR_F_EXTRA;   // the anaglyph pair lives in the catch-all variants
	const bool gi = s->settings.gi != 0, meshes = (geom & HXR_F_MESH) != 0;
	const bool rect = (s->features & HXR_F_RECT) != 0;   // the specialised Whitted variants know PointLights only
	const bool refr = (s->features & HXR_F_REFRACT) != 0;
	if ((s->features & HXR_F_CSGNEST) != 0) return gi ? V_NESTED_GI : V_NESTED;   // CSG over CSG: the variants that unroll CSGBase::intersect's recursion
	if (!gi && !rect && !refr && (geom | HXR_F_MESH) == HXR_F_MESH) return V_MESH;
	if (!gi && !rect && (geom | HXR_F_MESH) == HXR_F_MESH) return V_MESH_REFRACT;
	if (!gi && !rect && geom == HXR_F_HEIGHTFIELD) return V_HEIGHTFIELD;
	if (gi && (geom | HXR_F_SPHERE) == HXR_F_SPHERE) return V_SPHERE_GI;
	if (gi && (geom | HXR_F_MESH) == HXR_F_MESH) return V_MESH_GI;
	// the catch-alls: three waves per SIMD for Whitted, two for path tracing, whatever the geometry (tools/all_waves.py, end of round 2:
	// boxed.hexray 27 -> 18 ms and hw10/bokeh 15.7 -> 12.9 ms at 3; smallpt + a cube 46 ms at 2 against 77 at 3, cornell box + a cube
	// 38.2 against 39.1; round 1 chose by "has meshes")
	bool three = !gi;
	(void)meshes;
	if (const char* e = getenv("HXR_ALL_WAVES")) three = atoi(e) == 3;   // developer aid: force the 2- or 3-wave catch-all
	if (gi) return three ? V_ALL_GI_3 : V_ALL_GI_2;
	return three ? V_ALL_3 : V_ALL_2;
}

// Waves of the render kernel a CU holds at once (4 SIMDs x the occupancy the variant is compiled for): what the schedule takes for "a wave's even
// share of the frame" (hxr_sched_build_kernel).  A wrong figure here makes the split threshold swing from frame to frame: the 4-waves build of
// the headline kernel ran 6.76 ... 7.77 ms with the 12 that used to be written here, and 6.84 with its own 16.
// The in-place Monte-Carlo mesh kernel exists at 4 and at 3 waves per SIMD (hxr_variants.h): four when every resident wave has a long queue of
// packets (throughput: the whole frame), three when the frame is shared or small (one packet's latency).
// the control block of a scene (hxr_scene::d_work): [0..1023] dispenser counters and the AA count, [1024..1055] the schedule's control words (d_sched_ctl),
// then HXR_PART_SLOTS part-done counters and HXR_PART_SLOTS part pixels (d_sched_part); all but the pixels are cleared at the start of every frame
#define HXR_CONTROL_CLEARED_BYTES ((1024 + 32 + (size_t)HXR_PART_SLOTS) * sizeof(uint32_t))
#define HXR_DEEP_LAUNCH 96u   // packets per resident wave from which a Monte-Carlo mesh launch counts as throughput-bound
int mesh_mc_waves(const hxr_scene* s, uint32_t num_chunks)
{
	if (const char* e = getenv("HXR_MESH_MC_FOUR")) return atoi(e) ? HXR_MESH_MC_WAVES : HXR_MESH_WAVES;   // developer aid
	// Round 4: with its touch prefetch, priorities and replicating blocks the latency instance wins wherever a frame is shared -- per rank of the headline frame
	// (steady state, ms, four waves against three): 2-way 3.69 / 3.55, 3-way 2.85 / 2.45, 4-way 2.39 / 1.96, 6-way 1.75 / 1.43, 8-way 1.62 / 1.13 -- and only the whole frame
	// (126 packets per resident wave) keeps the fourth wave: 6.62 against 6.72.  The limit was 24 packets per resident wave, set before those measures existed.
	return num_chunks >= (uint32_t)s->num_cu * 4u * HXR_MESH_MC_WAVES * HXR_DEEP_LAUNCH ? HXR_MESH_MC_WAVES : HXR_MESH_WAVES;
}

int resident_waves_per_cu(const hxr_scene* s, int mode, bool in_place, uint32_t num_chunks)
{
	if (const char* e = getenv("HXR_SCHED_RESIDENT")) return std::max(1, atoi(e));   // developer aid
	switch (pick_variant(s)) {
		case V_MESH: return 4 * (!in_place ? 3 : (mode == 1 ? mesh_mc_waves(s, num_chunks) : HXR_MESH_WAVES));
		case V_MESH_REFRACT: return 4 * (!in_place ? 3 : HXR_MESH_WAVES);
		case V_HEIGHTFIELD: return 4 * HXR_HEIGHTFIELD_WAVES;
		case V_SPHERE_GI: return 4 * HXR_SPHERE_GI_WAVES;
		case V_MESH_GI: return 4 * HXR_MESH_GI_WAVES;
		case V_NESTED: case V_NESTED_GI: case V_ALL_2: case V_ALL_GI_2: return 8;
		case V_ALL_3: return 4 * HXR_ALL_WHITTED_WAVES;
		default: return 12;
	}
}

int ensure_parts_stream(hxr_scene* s)   // the stream of the replicating blocks and the two events that tie it to the caller's stream
{
	if (s->parts_stream) return 0;
	int lo = 0, hi = 0;
	HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
	HIP_TRY(hipStreamCreateWithPriority(&s->parts_stream, hipStreamNonBlocking, hi));
	HIP_TRY(hipEventCreateWithFlags(&s->ev_parts_go, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&s->ev_parts_done, hipEventDisableTiming));
	return 0;
}

// Is there a replicated-packet kernel for this launch?  (Whitted over meshes without refraction, in place, uninstrumented; HXR_NO_PARTS_KERNEL=1: A/B aid)
uint64_t defer_capacity(const hxr_scene* s, uint32_t num_chunks, int spp);
bool parts_kernel_for(const hxr_scene* s, const HxrLaunch& L)
{
	if (getenv("HXR_NO_PARTS_KERNEL") || s->frame_counts || pick_variant(s) != V_MESH) return false;
	if (defer_capacity(s, L.num_chunks, L.spp) != 0 || (s->fanout0 <= 1 && getenv("HXR_STACK1") != nullptr)) return false;
	// launches that run the three-waves (latency) instances: 1-spp and AA passes, Monte-Carlo launches a few packets deep -- the replicating instance is one of them
	// and shares the SIMDs with the others on equal terms; a whole Monte-Carlo frame (four waves of 128 registers) does not take it
	if (L.mode == 1 && mesh_mc_waves(s, L.num_chunks) != HXR_MESH_WAVES) return false;   // (tried: with the 168-register instance 6.45 -> 8.2 ... 9.5 ms per whole frame, with a 128-register one 6.68 -> 6.75 ... 6.97)
	return L.mode == 0 || L.mode == 1 || L.mode == 2;
}

// Deferred secondary rays: does this launch use them, and with what queue?  Only the Whitted mesh variants have a deferring
// build; the queue is sized for the worst case (every lane of every packet of a dispenser range emits `fanout0` rays), so an
// append can never fail, and a frame whose worst case does not fit HXR_QUEUE_MAX_RECORDS is traced in place as before.
#define HXR_QUEUE_MAX_RECORDS (64ull << 20)   // 5 GB of 80-byte records
#define HXR_SAH_DEVICE_MAX_TRIANGLES 65536   // largest mesh whose SAH tree the quadratic device builder takes (run: 9120 in 2.7 ms; ~50x the comparisons here)
uint64_t defer_capacity(const hxr_scene* s, uint32_t num_chunks, int spp)   // records per sub-queue, 0 = this launch traces in place
{
	const Variant v = pick_variant(s);
	if (!s->defer_allowed || s->no_defer_this_frame || (v != V_MESH && v != V_MESH_REFRACT) || s->fanout0 <= 0 || s->settings.max_trace_depth > 254 || spp > 65535 || num_chunks == 0) return 0;
	const uint64_t per_shard = ((uint64_t)num_chunks + HXR_SHARDS * HXR_BATCH - 1) / (HXR_SHARDS * HXR_BATCH) * HXR_BATCH;   // as in the kernel
	const uint64_t rounds = ((uint64_t)spp + 63) / 64;
	const uint64_t capacity = per_shard * 64 * rounds * (uint64_t)s->fanout0;
	if (capacity * HXR_QUEUE_SHARDS > HXR_QUEUE_MAX_RECORDS || capacity > 0xffffffffull) return 0;
	return capacity;
}

int plan_defer(hxr_scene* s, const Plan& p, HxrLaunch& L, int pass, hipStream_t st, bool& defer)
{
	defer = false;
	L.q_records = nullptr; L.q_ctl = nullptr; L.q_capacity = 0; L.q_accum = nullptr;
	const uint64_t capacity = defer_capacity(s, L.num_chunks, L.spp);
	if (!capacity) return 0;
	int rc = ensure((void**)&s->d_queue, &s->queue_bytes, (size_t)capacity * HXR_QUEUE_SHARDS * sizeof(HxrRayRec));
	if (rc) return rc;
	const size_t need = (size_t)p.W * p.H * 3 * sizeof(long long);
	if (s->accum_bytes < need || !s->d_accum) {
		rc = ensure((void**)&s->d_accum, &s->accum_bytes, need);
		if (rc) return rc;
		s->accum_dirty = true;
	}
	if (s->accum_dirty) {
		HIP_TRY(hipMemsetAsync(s->d_accum, 0, s->accum_bytes, st));
		s->accum_dirty = false;
	}
	if (!s->d_qctl) HIP_TRY(hipMalloc((void**)&s->d_qctl, 2 * QCTL_WORDS * sizeof(uint32_t)));
	L.q_ctl = s->d_qctl + (size_t)pass * QCTL_WORDS;
	HIP_TRY(hipMemsetAsync(L.q_ctl, 0, QCTL_WORDS * sizeof(uint32_t), st));
	L.q_records = s->d_queue;
	L.q_capacity = (uint32_t)capacity;
	L.q_accum = s->d_accum;
	defer = true;
	return 0;
}

// One pass over the frame: the render kernel and -- when its secondary rays are deferred -- the continuation and resolve kernels.
// `pass`: 0 = pass 1 / the Monte-Carlo pass, 1 = the AA taps; `scale`: what the render kernel multiplies a pixel's sample sum by.
int launch_render(hxr_scene* s, const Plan& p, HxrLaunch& L, bool counters, hipStream_t st, int timed, int pass, float scale)
{
	bool defer = false;
	int rc = plan_defer(s, p, L, pass, st, defer);
	if (rc) return rc;
	if (timed & TIME_BEGIN) { s->ev0 = s->ev_pass[pass][0]; s->passes_timed = 0; }
	s->pass_deferred[pass] = defer;
	HIP_TRY(hipEventRecord(s->ev_pass[pass][0], st));
	const HxrSceneDev& V = s->host_view;
#define HXR_KERNEL_ARGS_OF(LAUNCH) s->dev, V.nodes, V.meshes, V.heightfields, V.bitmaps, V.textures, V.shaders, V.layers, V.lights, V.kd, V.kd_box, V.hot, V.hot_ids, V.geometries, LAUNCH
#define HXR_KERNEL_ARGS HXR_KERNEL_ARGS_OF(L)
	bool parts_launched = false;
	if (s->parts_pending && !counters && !defer && L.num_chunks > 0) {
		// the replicated-packet kernel first, on its own stream behind the schedule's kernels, so that its few hundred waves find room before the persistent
		// waves of the main kernel fill the machine; the caller's stream waits for it after the main kernel
		int rcs = ensure_parts_stream(s);
		if (rcs) return rcs;
		HIP_TRY(hipStreamWaitEvent(s->parts_stream, s->ev_pass[pass][0], 0));   // (the pass's first event, recorded behind the schedule's kernels just above: no event of its own)
		const HxrLaunch& K = s->parts_launch;
		const int pgrid = parts_grid(s);
		if (K.mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<false, (HXR_F_MESH | HXR_F_TOUCH | HXR_F_REPL), HXR_MESH_WAVES, 1>), dim3(pgrid), dim3(HXR_BLOCK), 0, s->parts_stream, HXR_KERNEL_ARGS_OF(K));
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<false, (HXR_F_MESH | HXR_F_TOUCH | HXR_F_REPL), HXR_MESH_WAVES, -1>), dim3(pgrid), dim3(HXR_BLOCK), 0, s->parts_stream, HXR_KERNEL_ARGS_OF(K));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(s->ev_parts_done, s->parts_stream));
		parts_launched = true;
	}
	s->parts_pending = false;
	if (L.num_chunks > 0) {
		// (instrumented launches run variants of other occupancies than the table knows: eight per CU as before)
		// ... and so do launches a few packets deep per wave (1-spp passes: 0.593 ms with eight, 0.600 with the exact count)
		int grid = grid_for(s, L.num_chunks, counters || L.num_chunks < (uint32_t)s->num_cu * 16u * 24u ? 8 * (256 / HXR_BLOCK) : resident_waves_per_cu(s, L.mode, !defer && !(s->fanout0 <= 1 && getenv("HXR_STACK1") != nullptr), L.num_chunks) / (HXR_BLOCK / 64));
		// beside the replicating blocks the main kernel is launched with exactly the blocks that fit the machine with them (three blocks of four waves per CU at
		// three waves per SIMD): whichever kernel the dispatcher starts first, the other finds its room
		if (parts_launched) grid = std::max(1, std::min(grid, s->num_cu * HXR_MESH_WAVES * (256 / HXR_BLOCK) - parts_grid(s)));
#define HXR_LAUNCH(CNT, FEAT, WAVES)                                                                                                                      \
	do {                                                                                                                                                      \
		if (L.mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<CNT, FEAT, WAVES, 1>), dim3(grid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);     \
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<CNT, FEAT, WAVES, -1>), dim3(grid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);                \
	} while (0)
// ... for a variant whose Monte-Carlo instance is compiled for another occupancy than the instance that takes its mode from the launch
#define HXR_LAUNCH_W2(CNT, FEAT, WAVES_MC, WAVES)                                                                                                         \
	do {                                                                                                                                                      \
		if (L.mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<CNT, FEAT, WAVES_MC, 1>), dim3(grid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);  \
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<CNT, FEAT, WAVES, -1>), dim3(grid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);                \
	} while (0)
#define HXR_LAUNCH_F(FEAT, WAVES)                                                         \
	do {                                                                                  \
		if (counters) HXR_LAUNCH(true, FEAT, WAVES); else HXR_LAUNCH(false, FEAT, WAVES); \
	} while (0)
// a variant that exists only uninstrumented: the counted frames (tests, bench counters) run its superset COUNTED_FEAT
#define HXR_LAUNCH_F2(FEAT, COUNTED_FEAT, WAVES)                                                  \
	do {                                                                                          \
		if (counters) HXR_LAUNCH(true, COUNTED_FEAT, WAVES); else HXR_LAUNCH(false, FEAT, WAVES); \
	} while (0)
		// HXR_STACK1=1: the one-slot LDS ray stack for scenes where no hit leaves more than one secondary ray waiting.  It takes the
		// 12-entry per-lane stack out of scratch memory (private segment 1232 -> 352 B per lane) and is 2 % SLOWER on the headline frame
		// (9.13 vs 8.91 ms; 9.02 ms with the slot in registers): the stack entries are touched once per reflected ray, what costs is the
		// extra live state around the traversal.  Off by default; kept for the measurement.
		const bool one_pending = s->fanout0 <= 1 && getenv("HXR_STACK1") != nullptr;
		const int variant = pick_variant(s);
		switch (variant) {
			case V_MESH:
				if (defer) HXR_LAUNCH_F2(HXR_F_MESH | HXR_F_DEFER, HXR_F_MESH | HXR_F_REFRACT | HXR_F_DEFER, 3);
				else if (one_pending) HXR_LAUNCH_F2(HXR_F_MESH | HXR_F_STACK1, HXR_F_MESH | HXR_F_REFRACT | HXR_F_STACK1, 3);
				else if (counters) HXR_LAUNCH(true, HXR_F_MESH | HXR_F_REFRACT, HXR_MESH_WAVES);
				else if (mesh_mc_waves(s, L.num_chunks) == HXR_MESH_MC_WAVES) HXR_LAUNCH_W2(false, HXR_F_MESH, HXR_MESH_MC_WAVES, HXR_MESH_WAVES);
				else if (L.mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_render_kernel<false, (HXR_F_MESH | HXR_F_TOUCH), HXR_MESH_WAVES, 1>), dim3(grid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);   // the latency instance
				else HXR_LAUNCH(false, HXR_F_MESH, HXR_MESH_WAVES);
				break;
			case V_MESH_REFRACT:
				if (defer) HXR_LAUNCH_F(HXR_F_MESH | HXR_F_REFRACT | HXR_F_DEFER, 3);
				else if (one_pending) HXR_LAUNCH_F(HXR_F_MESH | HXR_F_REFRACT | HXR_F_STACK1, 3);
				else HXR_LAUNCH_F(HXR_F_MESH | HXR_F_REFRACT, HXR_MESH_WAVES);
				break;
			case V_HEIGHTFIELD: HXR_LAUNCH_F(HXR_HEIGHTFIELD_F, HXR_HEIGHTFIELD_WAVES); break;
			case V_SPHERE_GI: HXR_LAUNCH_F(HXR_F_SPHERE | HXR_F_GI | HXR_F_RECT | HXR_F_REFRACT, HXR_SPHERE_GI_WAVES); break;
			case V_MESH_GI: HXR_LAUNCH_F(HXR_F_MESH | HXR_F_GI | HXR_F_RECT | HXR_F_REFRACT, HXR_MESH_GI_WAVES); break;
			case V_ALL_GI_3: HXR_LAUNCH_F(HXR_F_ALL, 3); break;
			case V_ALL_GI_2: HXR_LAUNCH_F(HXR_F_ALL, 2); break;
			case V_ALL_3: HXR_LAUNCH_F(HXR_F_ALL & ~HXR_F_GI, HXR_ALL_WHITTED_WAVES); break;
			case V_ALL_2: HXR_LAUNCH_F(HXR_F_ALL & ~HXR_F_GI, 2); break;
			case V_NESTED_GI: HXR_LAUNCH_F(HXR_F_ALL | HXR_F_CSGNEST, 2); break;
			case V_NESTED: HXR_LAUNCH_F((HXR_F_ALL & ~HXR_F_GI) | HXR_F_CSGNEST, 2); break;
		}
#undef HXR_LAUNCH_F2
#undef HXR_LAUNCH_W2
#undef HXR_LAUNCH_F
#undef HXR_LAUNCH
		HIP_TRY(hipGetLastError());
	}
	if (parts_launched) HIP_TRY(hipStreamWaitEvent(st, s->ev_parts_done, 0));
	if (defer) HIP_TRY(hipEventRecord(s->ev_pass[pass][1], st));   // (a pass without a continuation has two events only: before and after)
	if (defer) {
		// the number of queued rays stays on the device: a fixed grid of persistent waves drains the sub-queues
		s->accum_dirty = true;
		const bool refract = pick_variant(s) == V_MESH_REFRACT;
		int deepest = 0;
		for (const auto& t : s->tree_stats) deepest = std::max(deepest, t.depth);
		L.lane_stack_entries = deepest + 2;
		L.cont_ticket = getenv("HXR_CONT_TICKET") ? atoi(getenv("HXR_CONT_TICKET")) : 0;   // developer aid
		const size_t lane_lds = (size_t)(HXR_LANE_BLOCK / 64) * (size_t)L.lane_stack_entries * 64 * sizeof(uint32_t);
		// per-lane walk unless its stacks would not fit the 64 KB a block may take (trees deeper than 126 levels), or HXR_CONTINUE_PACKET=1 (A/B aid)
		const bool lanewalk = lane_lds <= 64 * 1024 - 4096 && !getenv("HXR_CONTINUE_PACKET");
		if (lanewalk) {
			const int cgrid = std::max(1, s->num_cu * 16);
			if (counters) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_continue_kernel<true, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_LANEWALK), HXR_LANEWALK_WAVES>), dim3(cgrid), dim3(HXR_LANE_BLOCK), lane_lds, st, HXR_KERNEL_ARGS);
			else if (refract) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_continue_kernel<false, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_LANEWALK), HXR_LANEWALK_WAVES>), dim3(cgrid), dim3(HXR_LANE_BLOCK), lane_lds, st, HXR_KERNEL_ARGS);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_continue_kernel<false, (HXR_F_MESH | HXR_F_LANEWALK), HXR_LANEWALK_WAVES>), dim3(cgrid), dim3(HXR_LANE_BLOCK), lane_lds, st, HXR_KERNEL_ARGS);
		} else {
			const int cgrid = std::max(1, s->num_cu * 4);
			if (counters) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_continue_kernel<true, (HXR_F_MESH | HXR_F_REFRACT), 3>), dim3(cgrid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);
			else if (refract) hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_continue_kernel<false, (HXR_F_MESH | HXR_F_REFRACT), 3>), dim3(cgrid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(hxr_continue_kernel<false, (HXR_F_MESH), 3>), dim3(cgrid), dim3(HXR_BLOCK), 0, st, HXR_KERNEL_ARGS);
		}
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(s->ev_pass[pass][2], st));
		const uint32_t pixels = (uint32_t)p.ntiles * (uint32_t)(p.tile * p.tile) + (uint32_t)L.num_extra_pixels;
		const int rgrid = (int)std::max(1u, std::min<uint32_t>((pixels + HXR_BLOCK - 1) / HXR_BLOCK, (uint32_t)s->num_cu * 8u));
		hipLaunchKernelGGL(hxr_resolve_kernel, dim3(rgrid), dim3(HXR_BLOCK), 0, st, L, scale);
		HIP_TRY(hipGetLastError());
		s->accum_dirty = false;   // enqueued: the accumulator is zero again when this stream gets there
	}
	HIP_TRY(hipEventRecord(s->ev_pass[pass][3], st));
	s->passes_timed = pass + 1;
#undef HXR_KERNEL_ARGS
#undef HXR_KERNEL_ARGS_OF
	if (timed & TIME_END) { s->ev1 = s->ev_pass[pass][3]; s->timed = true; }
	return 0;
}

// pixels just outside this shard's tiles whose pass-1 colour detectAApixels needs
void halo_pixels(const Plan& p, std::vector<int32_t>& out)
{
	out.clear();
	if (p.count <= 1) return;
	std::vector<uint8_t> mark((size_t)p.W * p.H, 0);
	auto own = [&](int x, int y) { return hxr_tile_number(x / p.tile, y / p.tile, p.tiles_x, p.count) % p.count == p.rank; };
	for (int i = 0; i < p.ntiles; i++) {
		int t = i * p.count + p.rank;
		int tx_, ty_;
		hxr_tile_position(t, p.tiles_x, p.count, &tx_, &ty_);
		int x0 = tx_ * p.tile, y0 = ty_ * p.tile;
		int x1 = std::min(x0 + p.tile, p.W), y1 = std::min(y0 + p.tile, p.H);
		for (int y = y0 - 1; y <= y1; y++)
			for (int x = x0 - 1; x <= x1; x++) {
				if (x < 0 || y < 0 || x >= p.W || y >= p.H) continue;
				if (x >= x0 && x < x1 && y >= y0 && y < y1) continue;
				if (own(x, y) || mark[(size_t)y * p.W + x]) continue;
				mark[(size_t)y * p.W + x] = 1;
				out.push_back(y * p.W + x);
			}
	}
}

// Adaptive schedule: classify the chunks by what they cost in the previous frame of the same shape (if there was one) and
// point the launch at the lists; the render kernel records this frame's costs for the next one.
int prepare_schedule(hxr_scene* s, const Plan& p, HxrLaunch& L, uint32_t num_pixels, hipStream_t st)
{
	const uint32_t n = L.num_chunks;
	int rc = ensure((void**)&s->d_sched_cost, &s->sched_cost_cap, (size_t)n * sizeof(uint32_t));
	if (!rc) rc = ensure((void**)&s->d_sched_flags, &s->sched_flags_cap, (size_t)n);
	// sample-split level: Monte-Carlo frames whose pixel fills a quarter wave or more with one round of samples
	const int lanes_per_pixel = L.spp < 64 ? L.spp : 64;
	const uint32_t part_groups = (L.mode == 1 && L.spp <= 64 && lanes_per_pixel >= 2 * HXR_PART_LANES && lanes_per_pixel % HXR_PART_LANES == 0 && !getenv("HXR_SCHED_NO_PARTS"))
	                                 ? (uint32_t)(lanes_per_pixel / HXR_PART_LANES) : 0u;
	const uint32_t capacity = num_pixels + HXR_PART_SLOTS * 16u;
	if (!rc) rc = ensure((void**)&s->d_sched_items, &s->sched_items_cap, (size_t)capacity * sizeof(uint32_t));
	if (!rc) rc = ensure((void**)&s->d_sched_side, &s->sched_side_cap, (size_t)HXR_PART_SLOTS * 64 * 3 * sizeof(float));
	if (rc) return rc;
	// (d_sched_part and d_sched_ctl lie in the scene's control block, cleared at the start of the frame: render_on_stream)
	const hxr_scene::SchedKey key = {p.W, p.H, p.mode, p.spp, p.tile, p.rank, p.count, n};
	const bool usable = s->sched_history && memcmp(&key, &s->sched_key, sizeof key) == 0;
	// sample parts and single pixels go to the replicated-packet kernel where the scene's variant has one (words [10] count, [11] = 0, [12] ticket)
	const bool use_parts = parts_kernel_for(s, L);
	if (use_parts) {
		rc = ensure((void**)&s->d_sched_items_r, &s->sched_items_r_cap, (size_t)capacity * sizeof(uint32_t));
		if (rc) return rc;
	}
	if (usable) {
		const int grid = (int)std::min<uint32_t>((n + 255u) / 256u, 1024u);
		const int sum_grid = (int)std::min<uint32_t>((n + 255u) / 256u, 64u);   // one 64-bit atomic per wave on ONE word: 4096 of them cost 50 us, 256 cost 3
		// thresholds in multiples of the mean chunk cost: split into one-pixel packets / start first as a whole
		uint32_t split_x = 16, first_x = 6;   // first_x re-measured in round 3 at four waves per SIMD: 6.74 ms at 1, 6.69 at 2, 6.56 at 3, 6.51 at 4 and 6, 6.56 at 8, 6.73 at 16
		// ... and in round 4 for launches less than HXR_DEEP_LAUNCH packets deep per resident wave, where the tail is the late start of a packet of a few times the mean: 4
		// (slowest 8-way rank of the headline frame 1.21 ms at 6, 1.17 at 5, 1.13 at 4, 1.15 at 3, 1.21 at 2; 2-way 3.86 / 3.62 at 6 / 4; the whole frame 6.62 / 6.66)
		if (L.mode == 1 && n < (uint32_t)s->num_cu * 4u * 4u * HXR_DEEP_LAUNCH) first_x = 4;
		// ... and 2 for 1-spp passes (64 pixels per packet, ten packets per resident wave: every packet above the mean that starts late shows): simple 1 spp 0.72 ms at 6,
		// 0.65 at 3, 0.62 at 2 and 1; meshes 1 spp 1.00 / 0.97 / 0.90 / 0.95; the adaptive-AA frame (its pass 1) 1.82 / 1.76 / 1.79 / 1.72
		if (L.mode == 0) first_x = 2;
		if (const char* e = getenv("HXR_SCHED_SPLIT_X")) split_x = (uint32_t)atoi(e);   // developer aids
		if (const char* e = getenv("HXR_SCHED_FIRST_X")) first_x = (uint32_t)atoi(e);
		uint32_t frame_div = 1;   // "long against the frame": more than 1/frame_div of a wave's even share of the work (1 beat 2 and 4 at every N)
		if (const char* e = getenv("HXR_SCHED_FRAME_DIV")) frame_div = (uint32_t)atoi(e);
		// launches a few packets deep per resident wave list their costliest chunks first, in a phase of their own (hxr_sched_build_kernel; HXR_SCHED_ONE_PHASE=1: A/B aid)
		const bool two_phases = !(L.mode == 1 && n >= (uint32_t)s->num_cu * 4u * 4u * HXR_DEEP_LAUNCH) && !getenv("HXR_SCHED_ONE_PHASE");
		uint32_t* const largest = (uint32_t*)s->d_sched_ctl + 14;
		hipLaunchKernelGGL(hxr_sched_sum_kernel, dim3(sum_grid), dim3(256), 0, st, (const uint32_t*)s->d_sched_cost, n, s->d_sched_ctl, two_phases ? largest : (uint32_t*)nullptr);
		for (uint32_t phase = two_phases ? 1u : 0u; phase <= (two_phases ? 2u : 0u); phase++)
		hipLaunchKernelGGL(hxr_sched_build_kernel, dim3(grid), dim3(256), 0, st, s->d_sched_cost, n, (uint32_t)L.pixels_per_chunk, num_pixels,
		                   (const unsigned long long*)s->d_sched_ctl, split_x, first_x, (uint32_t)s->num_cu * (uint32_t)resident_waves_per_cu(s, L.mode, defer_capacity(s, L.num_chunks, L.spp) == 0 && !(s->fanout0 <= 1 && getenv("HXR_STACK1") != nullptr), n), frame_div, part_groups, s->d_sched_flags, s->d_sched_items, capacity,
		                   s->d_sched_part + HXR_PART_SLOTS, (uint32_t*)s->d_sched_ctl + 2, use_parts ? s->d_sched_items_r : (uint32_t*)nullptr, (uint32_t*)s->d_sched_ctl + 10,
		                   lanes_per_pixel <= 16 ? 1u : 0u, phase, (const uint32_t*)largest);
		HIP_TRY(hipGetLastError());
	} else {
		// No cost map yet.  Where can long chains of secondary rays be?  Only on the screen footprint of nodes whose shaders spawn them:
		// their chunks go first (whole), so that an 8 ms packet in the crease of a mirror starts at the beginning of the frame instead of
		// wherever frame order reaches it (first frame of the headline: 10.5 -> 9.x ms).
		HxrRects R;
		memset(&R, 0, sizeof R);
		if (!s->hot_everywhere && !getenv("HXR_SCHED_NO_RECTS")) {
			const hxr_camera& c = s->camera;
			double a[3], b[3];
			for (int k = 0; k < 3; k++) { a[k] = c.top_right[k] - c.top_left[k]; b[k] = c.bottom_left[k] - c.top_left[k]; }
			for (const auto& hsph : s->hot_spheres) {
				if (R.n >= 8) { R.n = 0; break; }   // too many to be worth telling apart
				double lo[2] = {1e30, 1e30}, hi[2] = {-1e30, -1e30};
				bool behind = false;
				for (int corner = 0; corner < 8 && !behind; corner++) {
					double d3[3], rhs[3];
					for (int k = 0; k < 3; k++) {
						d3[k] = hsph.c[k] + ((corner >> k) & 1 ? hsph.r : -hsph.r) - c.pos[k];   // P - pos
						rhs[k] = c.pos[k] - c.top_left[k];
					}
					// u a + v b - w (P - pos) = pos - top_left  (Cramer); w > 0: the corner is in front of the camera
					auto det = [](const double* x, const double* y, const double* z) {
						return x[0] * (y[1] * z[2] - y[2] * z[1]) - y[0] * (x[1] * z[2] - x[2] * z[1]) + z[0] * (x[1] * y[2] - x[2] * y[1]);
					};
					const double nd[3] = {-d3[0], -d3[1], -d3[2]};
					const double D = det(a, b, nd);
					if (std::fabs(D) < 1e-300) { behind = true; break; }
					const double u = det(rhs, b, nd) / D, v = det(a, rhs, nd) / D, w = det(a, b, rhs) / D;
					if (!(w > 1e-9)) { behind = true; break; }
					lo[0] = std::min(lo[0], u); hi[0] = std::max(hi[0], u);
					lo[1] = std::min(lo[1], v); hi[1] = std::max(hi[1], v);
				}
				if (behind) { R.n = 0; break; }
				const double margin = 0.03;   // depth-of-field blur, bump-mapped silhouettes
				const int x0 = (int)std::floor((lo[0] - margin) * p.W), x1 = (int)std::ceil((hi[0] + margin) * p.W);
				const int y0 = (int)std::floor((lo[1] - margin) * p.H), y1 = (int)std::ceil((hi[1] + margin) * p.H);
				if (x1 <= 0 || y1 <= 0 || x0 >= p.W || y0 >= p.H) continue;   // off screen
				if ((double)(std::min(x1, p.W) - std::max(x0, 0)) * (std::min(y1, p.H) - std::max(y0, 0)) > 0.4 * p.W * p.H) { R.n = 0; break; }   // most of the frame: nothing to put first
				R.r[R.n][0] = x0; R.r[R.n][1] = y0; R.r[R.n][2] = x1; R.r[R.n][3] = y1;
				R.n++;
			}
		}
		if (R.n > 0) {
			const int grid = (int)std::min<uint32_t>((n + 255u) / 256u, 1024u);
			hipLaunchKernelGGL(hxr_sched_rects_kernel, dim3(grid), dim3(256), 0, st, L, R, n, (uint32_t)L.pixels_per_chunk, num_pixels, s->d_sched_cost, s->d_sched_flags,
			                   s->d_sched_items, capacity, (uint32_t*)s->d_sched_ctl + 2);
			HIP_TRY(hipGetLastError());
		} else {
			HIP_TRY(hipMemsetAsync(s->d_sched_cost, 0, (size_t)n * sizeof(uint32_t), st));
			HIP_TRY(hipMemsetAsync(s->d_sched_flags, 0, (size_t)n, st));
		}
	}
	L.sched_cost = s->d_sched_cost;
	L.sched_flags = s->d_sched_flags;
	L.sched_items = s->d_sched_items;
	L.sched_counts = (const uint32_t*)s->d_sched_ctl + 2;
	L.sched_ticket = (uint32_t*)s->d_sched_ctl + 5;
	L.sched_capacity = capacity;
	L.sched_side = s->d_sched_side;
	L.sched_part_done = s->d_sched_part;
	L.sched_part_pixels = s->d_sched_part + HXR_PART_SLOTS;
	L.sched_part_groups = (int32_t)part_groups;
	// a launch a few packets deep per resident wave waits for its longest chain: the listed (long) packets get issue priority (HxrLaunch::front_priority)
	// ... and so do all launches of the Monte-Carlo mesh kernel's latency instance (2- to 8-way shards of the headline frame: -0.5 ... -3 % with the lower `first_x`)
	const bool latency_instance = pick_variant(s) == V_MESH && L.mode == 1 && mesh_mc_waves(s, n) == HXR_MESH_WAVES && HXR_MESH_WAVES != HXR_MESH_MC_WAVES;
	L.front_priority = ((n < (uint32_t)s->num_cu * (uint32_t)resident_waves_per_cu(s, L.mode, true, n) * 24u || latency_instance) && !getenv("HXR_NO_FRONT_PRIORITY")) || getenv("HXR_FRONT_PRIORITY") ? 1 : 0;
	// A frame without history: the chunks inside the screen rectangles -- where the long chains are, unsplit for want of measured costs -- run at the raised
	// priority as well; once the list is through, the stragglers among them keep it against the ordinary packets (first frame of the headline 7.7 -> 7.15 ms)
	if (!usable && !getenv("HXR_NO_FIRST_PRIORITY")) L.front_priority = 2;
	if (use_parts && usable) {
		s->parts_launch = L;                                   // same frame, same dispenser counters: once its list is through it takes ordinary packets
		s->parts_launch.sched_items = s->d_sched_items_r;
		s->parts_launch.sched_counts = (const uint32_t*)s->d_sched_ctl + 10;
		s->parts_launch.sched_ticket = (uint32_t*)s->d_sched_ctl + 12;
		s->parts_launch.front_priority = getenv("HXR_NO_FRONT_PRIORITY") ? 0 : 1;
		s->parts_pending = true;
		// ... and the main kernel takes from the same list (same ticket counter) once its own is through: a 1-spp pass lists thousands of single pixels
		L.sched2_items = s->d_sched_items_r;
		L.sched2_count = (const uint32_t*)s->d_sched_ctl + 10;
		L.sched2_ticket = (uint32_t*)s->d_sched_ctl + 12;
		L.sched2_early_above = (uint32_t)parts_grid(s) * (HXR_BLOCK / 64) * 4u;   // more than four items per replicating wave: everybody takes them from the start
	}
	s->sched_key = key;
	s->sched_history = true;
	s->sched_frame = true;
	s->sched_sum_valid = usable;
	return 0;
}

int ensure_cancel_word(hxr_scene* s)
{
	if (!s->h_cancel) {
		HIP_TRY(hipHostMalloc((void**)&s->h_cancel, 64, hipHostMallocMapped | hipHostMallocCoherent));
		HIP_TRY(hipHostGetDevicePointer((void**)&s->d_cancel, s->h_cancel, 0));
	}
	__atomic_store_n(s->h_cancel, 0u, __ATOMIC_SEQ_CST);
	return 0;
}

int render_on_stream(hxr_scene* s, const hxr_frame_params* fp, float* d_out, size_t stride, hipStream_t st)
{
	Plan p;
	int rc = resolve(s, fp, p);
	if (rc) return rc;
	if (p.layout == HXR_LAYOUT_FRAME && stride < (size_t)p.W * 12) { set_err("row stride smaller than a row"); return HXR_E_INVALID; }
	const bool counters = fp->collect_counters != 0;
#if defined(HXR_PHASE_TIMERS) || defined(HXR_LEAF_STATS) || defined(HXR_TURN_STATS) || defined(HXR_LEAF_HIST) || defined(HXR_WALK_STATS)
	const bool want_counter_buffer = true;    // profiling build: the phase clocks land in hxr_counters.reserved_
#else
	const bool want_counter_buffer = counters;
#endif
	if (getenv("HXR_SCHED_VERBOSE") && s->sched_history) {   // developer aid: what the LAST frame's schedule listed (synchronises)
		unsigned long long ctl[8] = {0};
		if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(ctl, s->d_sched_ctl, sizeof ctl, hipMemcpyDeviceToHost) == hipSuccess) {
			const uint32_t* w = (const uint32_t*)ctl;
			fprintf(stderr, "[hxr] last schedule: total cost %llu, listed pixels / parts %u, whole chunks first %u, part slots %u\n", ctl[0], w[2], w[3], w[4]);
		}
	}
	HIP_TRY(hipMemsetAsync(s->d_work, 0, HXR_CONTROL_CLEARED_BYTES, st));   // dispenser counters, schedule control words, part-done counters
	if (want_counter_buffer) HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(hxr_counters), st));
	s->counters_valid = want_counter_buffer;
	s->frame_counts = counters;
	s->parts_pending = false;
	s->timed = false;
	s->aa_cost_aware = false;
	s->sched_frame = false;
	s->sched_sum_valid = false;
	HxrLaunch L;
	fill_launch(s, p, L);
	s->no_defer_this_frame = p.mode == HXR_MODE_PREPASS;
	L.counters = want_counter_buffer ? s->d_counters : nullptr;
	const uint32_t tile_pixels = (uint32_t)p.ntiles * (uint32_t)(p.tile * p.tile);

	if (p.mode == HXR_MODE_MONTE_CARLO && s->camera.dof && s->camera.auto_focus) {
		const HxrSceneDev& V = s->host_view;
		double* focal = (double*)((char*)s->dev + offsetof(HxrSceneDev, camera) + offsetof(hxr_camera, focal_plane_dist));
		hipLaunchKernelGGL(hxr_autofocus_kernel, dim3(1), dim3(64), 0, st, s->dev, V.nodes, V.meshes, V.heightfields, V.geometries, V.kd, V.kd_box, V.hot, V.hot_ids, p.W,
		                   p.H, focal);
		HIP_TRY(hipGetLastError());
	}
	if (p.mode == HXR_MODE_PREPASS) {   // coarseRender: one "pixel" per 24x24 block, spp = prepassSamples rays each
		L.mode = 3;
		L.spp = p.spp;
		L.pixels_per_chunk = p.spp <= 64 ? 64 / p.spp : 1;
		const uint32_t blocks = (uint32_t)((p.W + 23) / 24) * (uint32_t)((p.H + 23) / 24);
		L.num_chunks = (blocks + (uint32_t)L.pixels_per_chunk - 1) / (uint32_t)L.pixels_per_chunk;
		L.out = d_out;
		L.out_stride = stride;
		L.work_counter = s->d_work;
		L.tile_done = nullptr;
		return launch_render(s, p, L, counters, st, TIME_BEGIN | TIME_END, 0, 1.0f / (float)L.spp);
	}
	if (p.mode == HXR_MODE_MONTE_CARLO || p.mode == HXR_MODE_SINGLE) {
		L.mode = p.mode == HXR_MODE_MONTE_CARLO ? 1 : 0;
		L.spp = p.spp;
		L.pixels_per_chunk = p.spp <= 64 ? 64 / p.spp : 1;
		if (getenv("HXR_DEBUG_LANES")) L.pixels_per_chunk = atoi(getenv("HXR_DEBUG_LANES"));   // developer aid: live lanes per wave
		L.num_chunks = (tile_pixels + (uint32_t)L.pixels_per_chunk - 1) / (uint32_t)L.pixels_per_chunk;
		L.out = d_out;
		L.out_stride = stride;
		L.out_tiles = p.layout == HXR_LAYOUT_TILES;
		L.work_counter = s->d_work;
		// (a deferring render kernel has packets of equal length: nothing to schedule)
		if (s->adaptive && L.num_chunks >= 4096 && !defer_capacity(s, L.num_chunks, L.spp) && p.fovea == 0) {   // small frames have no tail worth scheduling
			rc = prepare_schedule(s, p, L, tile_pixels, st);
			if (rc) return rc;
		}
		return launch_render(s, p, L, counters, st, TIME_BEGIN | TIME_END, 0, L.mode == 1 ? 1.0f / (float)L.spp : 1.0f);
	}

	// adaptive AA: pass 1 -> detect -> pass 3 (src/main.cpp:253-345)
	const bool direct = p.count == 1 && p.layout == HXR_LAYOUT_FRAME;   // render straight into the caller's frame
	float* frame = d_out;
	size_t fstride = stride;
	std::vector<int32_t> halo;
	if (!direct) {
		fstride = (size_t)p.W * 12;
		rc = ensure((void**)&s->d_scratch, &s->scratch_bytes, fstride * p.H);
		if (rc) return rc;
		frame = s->d_scratch;
		halo_pixels(p, halo);
		if (!halo.empty()) {
			size_t cap = s->extra_capacity * sizeof(int32_t);
			rc = ensure((void**)&s->d_extra, &cap, halo.size() * sizeof(int32_t));
			if (rc) return rc;
			s->extra_capacity = cap / sizeof(int32_t);
			HIP_TRY(hipMemcpyAsync(s->d_extra, halo.data(), halo.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
			HIP_TRY(hipStreamSynchronize(st));   // `halo` is a local
		}
	}
	{
		size_t cap = s->aa_capacity * sizeof(int32_t);
		rc = ensure((void**)&s->d_aa_list, &cap, ((size_t)tile_pixels + 64) * sizeof(int32_t));
		if (rc) return rc;
		s->aa_capacity = cap / sizeof(int32_t);
	}
	L.mode = 0;
	L.spp = 1;
	L.pixels_per_chunk = 64;
	L.num_extra_pixels = (int32_t)halo.size();
	L.extra_pixels = s->d_extra;
	L.num_chunks = (tile_pixels + (uint32_t)halo.size() + 63u) / 64u;
	L.out = frame;
	L.out_stride = fstride;
	L.out_tiles = 0;
	L.work_counter = s->d_work;
	// pass 1 records what its 64-pixel chunks cost (same machinery as the adaptive schedule, nothing listed): pass 3 of THIS
	// frame starts with the flagged pixels of the expensive chunks, one pixel per packet
	const uint32_t list_capacity = tile_pixels + 64u;
	const bool cost_aware = s->adaptive && L.num_chunks >= 1024 && !getenv("HXR_AA_NO_COST") && !defer_capacity(s, L.num_chunks, L.spp) && p.fovea == 0;
	s->aa_cost_aware = cost_aware;
	// Pass 1 itself is scheduled from the previous frame's pass-1 costs like a 1-spp frame (chunks that were expensive go first, one pixel
	// per packet): unscheduled, ONE 64-pixel packet in the crease of the heart holds the pass for 3.8 ms of a 1.3 ms pass.
	const bool schedule_pass1 = cost_aware && L.num_chunks >= 4096 && !getenv("HXR_AA_NO_PASS1_SCHEDULE");
	if (schedule_pass1) {
		rc = prepare_schedule(s, p, L, tile_pixels, st);   // also clears the cost array this pass writes and pass 3 reads
		if (rc) return rc;
	} else if (cost_aware) {
		rc = ensure((void**)&s->d_sched_cost, &s->sched_cost_cap, (size_t)L.num_chunks * sizeof(uint32_t));
		if (!rc) rc = ensure((void**)&s->d_sched_flags, &s->sched_flags_cap, (size_t)L.num_chunks);
		if (rc) return rc;
		HIP_TRY(hipMemsetAsync(s->d_sched_cost, 0, (size_t)L.num_chunks * sizeof(uint32_t), st));
		HIP_TRY(hipMemsetAsync(s->d_sched_flags, 0, (size_t)L.num_chunks, st));
		L.sched_cost = s->d_sched_cost;
		L.sched_flags = s->d_sched_flags;
		L.sched_counts = (const uint32_t*)s->d_sched_ctl + 6;   // two zeros: nothing is listed in pass 1
		L.sched_ticket = (uint32_t*)s->d_sched_ctl + 8;
		s->sched_history = false;                                // the Monte-Carlo cost map, if any, is gone
	}
	rc = launch_render(s, p, L, counters, st, TIME_BEGIN, 0, 1.0f);
	if (rc) return rc;
	if (tile_pixels) {
		if (cost_aware) {
			// Pass 3 has control words of its own ([24] [25] counts, [27] ticket: cleared with the block at the start of the frame), so nothing is cleared between the passes;
			// and the mean chunk cost its "heavy" threshold needs is the one pass 1 was scheduled with -- the previous frame's, [0] -- when there was one: no second sum
			// (a fill, a kernel and their launch gaps: 20 us of a 1.6 ms frame).  A first frame sums this frame's costs as before ([0] is still zero then).
			if (!(schedule_pass1 && s->sched_sum_valid)) {
				const int sgrid = (int)std::min<uint32_t>((L.num_chunks + 255u) / 256u, 64u);
				hipLaunchKernelGGL(hxr_sched_sum_kernel, dim3(sgrid), dim3(256), 0, st, (const uint32_t*)s->d_sched_cost, L.num_chunks, s->d_sched_ctl, (uint32_t*)nullptr);
			}
		}
		uint32_t heavy_x = 2;    // round 1, frame ms at 2 / 4 / 8 / 16 / 32: 5.13 / 5.07 / 4.96 / 4.82 / 4.99 (5.09 without the list); round 4, with the heavy pixels replicated: 4 / 8 / 16 / 32: 1.75 / 1.82 / 1.83 / 1.82;
		                         // end of round 4 (medians of 40 frames, twice): 1 / 2 / 3 / 4: 1.83 / 1.74 / 1.78 / 1.80
		if (const char* e = getenv("HXR_AA_HEAVY_X")) heavy_x = (uint32_t)atoi(e);   // developer aid
		uint32_t detect_per_cu = 4;   // 1920 x 1080: 78 us at one block per CU, 50 at two, 38 at four, 42 at eight
		if (const char* e = getenv("HXR_DETECT_PER_CU")) detect_per_cu = (uint32_t)std::max(1, atoi(e));   // developer aid
		int grid = (int)std::min<uint32_t>((tile_pixels + HXR_BLOCK - 1) / HXR_BLOCK, (uint32_t)s->num_cu * detect_per_cu);
		hipLaunchKernelGGL(hxr_detect_aa_kernel, dim3(grid), dim3(HXR_BLOCK), 0, st, L, (const float*)frame, fstride, s->d_aa_list, (int32_t*)(s->d_work + 512),
		                   cost_aware ? (const uint32_t*)s->d_sched_cost : (const uint32_t*)nullptr, (const unsigned long long*)s->d_sched_ctl, L.num_chunks,
		                   cost_aware ? (uint32_t*)s->d_sched_ctl + 25 : (uint32_t*)nullptr, list_capacity, heavy_x);
		HIP_TRY(hipGetLastError());
	}
	// pass 3: the flagged-pixel count stays on the device; launch enough waves for the worst case and let them drain
	HxrLaunch L3 = L;
	L3.mode = 2;
	L3.spp = 4;
	// Flagged pixels are the expensive ones (object edges) and there are few of them: half-full packets of 8 pixels x 4 taps
	// balance better over the waves than full ones (frame 5.40 -> 5.10 ms; 4 pixels the same, 2 and 16 slower)
	L3.pixels_per_chunk = 8;
	if (const char* e = getenv("HXR_AA_PPC")) L3.pixels_per_chunk = atoi(e);   // developer aid
	L3.num_extra_pixels = 0;
	L3.tile_done = nullptr;
	L3.aa_list = s->d_aa_list;
	L3.sched_cost = nullptr;
	L3.sched_flags = nullptr;
	L3.sched_counts = nullptr;
	L3.sched2_items = nullptr;
	L3.front_priority = 0;
	if (cost_aware) {   // the back of the flagged list, single pixels, first
		L3.front_priority = getenv("HXR_NO_FRONT_PRIORITY") ? 0 : 2;   // ... and at the raised priority: they are this pass's long chains
		L3.sched_items = (const uint32_t*)s->d_aa_list;
		L3.sched_capacity = list_capacity;
		L3.sched_counts = (const uint32_t*)s->d_sched_ctl + 24;   // [24] stays 0 (nothing from the front), [25] = pixels at the back
		L3.sched_ticket = (uint32_t*)s->d_sched_ctl + 27;
	}
	L3.num_chunks = (tile_pixels + (uint32_t)L3.pixels_per_chunk - 1u) / (uint32_t)L3.pixels_per_chunk;
	L3.work_counter = s->d_work + 256;
	if (cost_aware && parts_kernel_for(s, L3)) {   // the heavy flagged pixels (the back of the list, one pixel = four taps per packet) go to the replicated-packet kernel
		s->parts_launch = L3;   // same list, same ticket counter as the main kernel: whoever gets there first renders the pixel
		s->parts_pending = true;
	}
	rc = launch_render(s, p, L3, counters, st, TIME_END, 1, 1.0f / 5);
	if (rc) return rc;
	if (!direct && tile_pixels) {
		// Only this shard's own pixels leave the scratch frame (tile-major, or in place for FRAME layout): the rest of it holds
		// halo colours of pass 1 and whatever an earlier frame left there.  Pixels of other shards in the caller's frame stay
		// untouched, exactly like sharded SINGLE / MONTE_CARLO frames.
		int grid = (int)std::min<uint32_t>((tile_pixels + HXR_BLOCK - 1) / HXR_BLOCK, 2048u);
		hipLaunchKernelGGL(hxr_extract_tiles_kernel, dim3(grid), dim3(HXR_BLOCK), 0, st, L, (const float*)frame, fstride, d_out, stride,
		                   p.layout == HXR_LAYOUT_TILES ? 1 : 0);
		HIP_TRY(hipGetLastError());
	}
	return 0;
}

}  // namespace

extern "C" {

int hxr_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int hxr_init(int device_id)
{
	int n = hxr_device_count();
	if (n <= 0) { set_err("no HIP device visible"); return HXR_E_NO_DEVICE; }
	if (device_id < 0 || device_id >= n) { set_err("device %d outside 0..%d", device_id, n - 1); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(device_id));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device_id));
	g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	g_device = device_id;
	return 0;
}

hxr_scene* hxr_scene_create(const hxr_scene_desc* d) { return hxr_scene_create_ex(d, nullptr); }

hxr_scene* hxr_scene_create_ex(const hxr_scene_desc* d, const hxr_scene_options* opt)
{
	if (g_device < 0) { set_err("hxr_init has not been called"); return nullptr; }
	if (validate(d)) return nullptr;
	int accel = HXR_ACCEL_REFERENCE;
	hxr_scene_options own;   // the caller's record may be an older, shorter one: fields it does not have are 0
	memset(&own, 0, sizeof own);
	if (opt) {
		if (opt->struct_size < 2 * sizeof(uint32_t)) { set_err("hxr_scene_options.struct_size too small"); return nullptr; }
		memcpy(&own, opt, std::min<size_t>(opt->struct_size, sizeof own));
		opt = &own;
		accel = opt->accel;
		if (accel != HXR_ACCEL_REFERENCE && accel != HXR_ACCEL_SAH && accel != HXR_ACCEL_REFERENCE_DEVICE_BUILD) { set_err("unknown acceleration structure %d", accel); return nullptr; }
	}
	const bool adaptive = !(opt && opt->no_adaptive_schedule) && !getenv("HXR_NO_ADAPTIVE_SCHEDULE");
	// allocations and launches below go to the CURRENT device of the calling thread: make that the device hxr_init chose, whatever
	// the thread has been bound to since (hxr_group_create walks over several devices)
	if (hipSetDevice(g_device) != hipSuccess) { set_err("hipSetDevice(%d) failed", g_device); return nullptr; }
	hxr_scene* s = new hxr_scene;
	s->adaptive = adaptive;
	s->device = g_device;
	s->num_cu = g_num_cu;
	s->defer_allowed = ((opt && opt->deferred_rays) || getenv("HXR_DEFER")) && !getenv("HXR_NO_DEFER");
	for (int64_t i = 0; i < d->num_nodes; i++) {   // validate() has bounded this walk
		int rays = 0, pending = 1;
		bool glossy = false, cyclic = false;
		shader_fanout(d, d->nodes[i].shader, 0, rays, pending, glossy, cyclic);
		s->fanout0 = std::max(s->fanout0, rays);
	}
	auto fail = [&]() -> hxr_scene* { hxr_scene_destroy(s); return nullptr; };
	HxrSceneDev& V = s->host_view;
	memset(&V, 0, sizeof V);
	s->settings = V.settings = d->settings;
	s->camera = V.camera = d->camera;
	V.environment = d->environment;
	V.num_nodes = (int32_t)d->num_nodes;
	V.num_lights = (int32_t)d->num_lights;
	V.num_shaders = (int32_t)d->num_shaders;
	V.num_textures = (int32_t)d->num_textures;

	const double* d_f64 = nullptr;
	const float* d_f32 = nullptr;
	const hxr_triangle* d_tris = nullptr;
	if (s->arena.upload(d->f64_pool, (size_t)d->f64_pool_size, &d_f64)) return fail();
	if (s->arena.upload(d->f32_pool, (size_t)d->f32_pool_size, &d_f32)) return fail();
	if (s->arena.upload(d->triangles, (size_t)d->num_triangles, &d_tris)) return fail();

	if (own.device_prepare_triangles)   // Mesh::prepareTriangles as a kernel over the records just uploaded (before anything reads AB / AC / ABcrossAC)
		for (int64_t i = 0; i < d->num_meshes; i++) {
			const hxr_mesh& m = d->meshes[i];
			if (hxr_kd::prepare_triangles(const_cast<hxr_triangle*>(d_tris) + m.first_triangle, m.num_triangles, d_f64 + m.first_vertex * 3, d_f64 + m.first_uv * 3)) {
				set_err("prepare_triangles kernel failed to launch");
				return fail();
			}
		}
	s->d_triangles = d_tris;

	std::vector<HxrMeshDev> meshes((size_t)d->num_meshes);
	std::vector<HxrKdSlim> all_kd;
	std::vector<HxrKdBox> all_boxes;
	std::vector<HxrTriHot> all_hot;
	std::vector<int32_t> all_ids;
	for (int64_t i = 0; i < d->num_meshes; i++) {
		const hxr_mesh& m = d->meshes[i];
		HxrMeshDev& o = meshes[(size_t)i];
		memset(&o, 0, sizeof o);
		memcpy(o.bbox_min, m.bbox_min, sizeof o.bbox_min);
		memcpy(o.bbox_max, m.bbox_max, sizeof o.bbox_max);
		o.faceted = m.faceted;
		o.backface_culling = m.backface_culling;
		o.has_kd = m.has_kd && (m.num_kd_nodes > 0 || accel != HXR_ACCEL_REFERENCE);   // a descriptor without trees (HXH_GPU_KD_BUILD): the module builds them
		o.num_triangles = (int32_t)m.num_triangles;
		const hxr_kd_node* nodes = d->kd_nodes + m.first_kd_node;
		const int32_t* leaf_idx = d->leaf_indices + m.first_leaf_index;
		int64_t num_nodes = m.num_kd_nodes, num_leaf_entries = m.num_leaf_indices;
		hxr_sah::SahTree sah;
		hxr_kd::DeviceBuiltTree built;
		if (accel == HXR_ACCEL_SAH && o.has_kd) {   // meshes the caller wants brute-forced (useKDTree off) stay that way
			hxr_sah::SahParams sp;
			if (const char* e = getenv("HXR_SAH_TRAVERSAL_COST")) sp.traversal_cost = atof(e);   // developer aids for tuning the cost model
			if (const char* e = getenv("HXR_SAH_EMPTY_BONUS")) sp.empty_bonus = atof(e);
			if (const char* e = getenv("HXR_SAH_MIN_LEAF")) sp.min_leaf = atoi(e);
			// The device build counts a node's candidates against all its triangles (N x 6N comparisons at the root): right for the reference's meshes (<= 9120
			// triangles, 2.7 ms), tens of seconds at a million.  Above HXR_SAH_DEVICE_MAX_TRIANGLES, or when the device build cannot get its buffers, the
			// O(N log N) host builder makes the same tree (hxr_accel.h).
			bool on_host = getenv("HXR_SAH_HOST") != nullptr || m.num_triangles > HXR_SAH_DEVICE_MAX_TRIANGLES;
			if (!on_host) {                 // built by kernels over the tables uploaded above (hxr_kd_build.hip)
				char why[256] = "";
				if (hxr_kd::build_sah_tree(d_tris + m.first_triangle, m.num_triangles, d_f64 + m.first_vertex * 3, m.bbox_min, m.bbox_max, sp.traversal_cost, sp.intersect_cost,
				                           sp.empty_bonus, sp.max_depth, sp.min_leaf, built, why, sizeof why)) {
					(void)hipGetLastError();   // e.g. an allocation that did not fit: the host builder needs none of it
					on_host = true;
				} else {
					nodes = built.nodes.data();
					leaf_idx = built.leaf_indices.data();
					num_nodes = (int64_t)built.nodes.size();
					num_leaf_entries = (int64_t)built.leaf_indices.size();
					s->kd_build_ms += built.build_ms;
				}
			}
			if (on_host) {                  // the host builder (hxr_accel.h): what the device build is compared with
				hxr_sah::SahBuilder(d->triangles + m.first_triangle, m.num_triangles, d->f64_pool + m.first_vertex * 3, sp).build(m.bbox_min, m.bbox_max, sah);
				nodes = sah.nodes.data();
				leaf_idx = sah.leaf_indices.data();
				num_nodes = (int64_t)sah.nodes.size();
				num_leaf_entries = (int64_t)sah.leaf_indices.size();
			}
		}
		if (accel == HXR_ACCEL_REFERENCE_DEVICE_BUILD && o.has_kd) {   // Mesh::buildKD as kernels, over the tables uploaded above
			char why[256] = "";
			if (hxr_kd::build_reference_tree(d_tris + m.first_triangle, m.num_triangles, d_f64 + m.first_vertex * 3, m.bbox_min, m.bbox_max, built, why, sizeof why)) {
				set_err("kd build of mesh %lld: %s", (long long)i, why);
				return fail();
			}
			nodes = built.nodes.data();
			leaf_idx = built.leaf_indices.data();
			num_nodes = (int64_t)built.nodes.size();
			num_leaf_entries = (int64_t)built.leaf_indices.size();
			s->kd_build_ms += built.build_ms;
		}
		int depth = append_mesh_tables(d, m, nodes, num_nodes, leaf_idx, o, all_kd, all_boxes, all_hot, all_ids);
		s->tree_stats.push_back({num_nodes, num_leaf_entries, depth});
		if (3 * (depth + 1) > HXR_KD_STACK) {   // the packet walk queues at most three visits per level
			set_err("kd-tree of mesh %lld is %d levels deep (limit %d)", (long long)i, depth, HXR_KD_STACK / 3 - 1);
			return fail();
		}
		o.tris = d_tris + m.first_triangle;
		o.vertices = d_f64 + m.first_vertex * 3;
		o.normals = d_f64 + m.first_normal * 3;
		o.uvs = d_f64 + m.first_uv * 3;
	}
	if (s->arena.upload(meshes.data(), meshes.size(), &V.meshes)) return fail();
	if (s->arena.upload(all_kd.data(), all_kd.size(), &V.kd)) return fail();
	if (s->arena.upload(all_boxes.data(), all_boxes.size(), &V.kd_box)) return fail();
	if (s->arena.upload(all_ids.data(), all_ids.size(), &V.hot_ids)) return fail();
	if (!all_ids.empty()) {   // the hot records, gathered on the device mesh by mesh
		void* p = nullptr;
		if (s->arena.alloc(all_ids.size() * sizeof(HxrTriHot), &p)) return fail();
		V.hot = (const HxrTriHot*)p;
		for (int64_t i = 0; i < d->num_meshes; i++) {
			const HxrMeshDev& o = meshes[(size_t)i];
			const int64_t end = i + 1 < d->num_meshes ? meshes[(size_t)i + 1].hot_first : (int64_t)all_ids.size();
			const int64_t n = end - o.hot_first;
			if (n <= 0) continue;
			const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
			hipLaunchKernelGGL(hxr_expand_hot_kernel, dim3(grid), dim3(256), 0, nullptr, o.tris, o.vertices, V.hot_ids + o.hot_first, n, (HxrTriHot*)p + o.hot_first);
		}
		if (hipGetLastError() != hipSuccess) { set_err("hxr_expand_hot_kernel failed to launch"); return fail(); }
	}

	std::vector<HxrHeightfieldDev> hfs((size_t)d->num_heightfields);
	for (int64_t i = 0; i < d->num_heightfields; i++) {
		const hxr_heightfield& h = d->heightfields[i];
		HxrHeightfieldDev& o = hfs[(size_t)i];
		memset(&o, 0, sizeof o);
		o.W = h.W; o.H = h.H; o.max_k = h.max_k; o.use_optimization = h.use_optimization;
		memcpy(o.bbox_min, h.bbox_min, sizeof o.bbox_min);
		memcpy(o.bbox_max, h.bbox_max, sizeof o.bbox_max);
		o.heights = d_f32 + h.first_height;
		o.normals = d_f64 + h.first_normal * 3;
		if (h.W <= 0 || h.H <= 0 || h.first_height < 0 || h.first_height + (int64_t)h.W * h.H > d->f32_pool_size) { set_err("bad heightfield ranges"); return fail(); }
		s->hf_raw.push_back(h.raw_heights ? 1 : 0);
		if (h.raw_heights) {
			// the descriptor stops at the image: blur, y extent of the box and normals as kernels (the maxima tables follow below)
			const int cells = h.W * h.H, grid = std::min((cells + 255) / 256, 4096);
			void *ph = nullptr, *pn = nullptr, *pr = nullptr;
			if (s->arena.alloc((size_t)cells * sizeof(float), &ph) || s->arena.alloc((size_t)cells * 3 * sizeof(double), &pn) || s->arena.alloc(2 * sizeof(float), &pr)) return fail();
			if (h.blur_radius > 0) hipLaunchKernelGGL(hxr_hf_blur_kernel, dim3(grid), dim3(256), 0, nullptr, o.heights, h.W, h.H, d_f32 + h.first_blur_kernel, h.blur_radius, (float*)ph);
			else if (hipMemcpyAsync(ph, o.heights, (size_t)cells * sizeof(float), hipMemcpyDeviceToDevice, nullptr) != hipSuccess) { set_err("hipMemcpyAsync failed"); return fail(); }
			hipLaunchKernelGGL(hxr_hf_range_kernel, dim3(1), dim3(1024), 0, nullptr, (const float*)ph, cells, (float*)pr);
			hipLaunchKernelGGL(hxr_hf_normals_kernel, dim3(grid), dim3(256), 0, nullptr, (const float*)ph, h.W, h.H, (double*)pn);
			float range[2] = {0, 0};
			if (hipGetLastError() != hipSuccess || hipMemcpy(range, pr, sizeof range, hipMemcpyDeviceToHost) != hipSuccess) { set_err("heightfield preparation kernels failed"); return fail(); }
			o.bbox_min[0] = 0; o.bbox_min[1] = range[0]; o.bbox_min[2] = 0;   // src/heightfield.cpp:213-216
			o.bbox_max[0] = h.W; o.bbox_max[1] = range[1]; o.bbox_max[2] = h.H;
			o.heights = (const float*)ph;
			o.normals = (const double*)pn;
		}
		if (h.first_maxh >= 0 && h.first_highmap >= 0) {
			o.max_h = d_f32 + h.first_maxh;
			o.high_map = d_f32 + h.first_highmap;
		} else {
			// the host left the maxima tables to the module (first_maxh / first_highmap = -1): pure maxima of the heights, the same
			// values the host builds (src/heightfield.cpp:49-81,222-235), level by level on the device
			const int cells = h.W * h.H;
			void* pm = nullptr;
			void* ph = nullptr;
			if (s->arena.alloc((size_t)cells * sizeof(float), &pm) || s->arena.alloc((size_t)cells * 16 * sizeof(float), &ph)) return fail();
			if (hipMemsetAsync(ph, 0, (size_t)cells * 16 * sizeof(float), nullptr) != hipSuccess) { set_err("hipMemsetAsync failed"); return fail(); }
			const int grid = std::min((cells + 255) / 256, 2048);
			hipLaunchKernelGGL(hxr_hf_maxh_kernel, dim3(grid), dim3(256), 0, nullptr, o.heights, h.W, h.H, (float*)pm);
			o.max_k = 0;
			if (h.use_optimization) {
				o.max_k = (int)std::ceil(std::log((double)h.W) / std::log(2.0));   // Heightfield::buildHighMap's maxK
				if (o.max_k > 16) { set_err("heightfield wider than 65536 texels"); return fail(); }
				for (int k = 0; k < o.max_k; k++) hipLaunchKernelGGL(hxr_hf_highmap_kernel, dim3(grid), dim3(256), 0, nullptr, o.heights, h.W, h.H, k, (float*)ph);
			}
			if (hipGetLastError() != hipSuccess) { set_err("heightfield table kernels failed to launch"); return fail(); }
			o.max_h = (const float*)pm;
			o.high_map = (const float*)ph;
		}
		{   // the marcher's per-cell record {maxH, levels 1..3}, gathered from the two tables above (whoever made them)
			const int cells = h.W * h.H;
			void* pr = nullptr;
			if (s->arena.alloc((size_t)cells * 4 * sizeof(float), &pr)) return fail();
			const bool pyramid = o.use_optimization && o.max_k > 1;   // only then does validate() vouch for 16 floats per cell behind high_map
			hipLaunchKernelGGL(hxr_hf_pack_kernel, dim3(std::min((cells + 255) / 256, 2048)), dim3(256), 0, nullptr, o.max_h, pyramid ? o.high_map : (const float*)nullptr, cells,
			                   (float*)pr);
			if (hipGetLastError() != hipSuccess) { set_err("heightfield table kernels failed to launch"); return fail(); }
			o.step_rec = (const float*)pr;
		}
	}
	if (s->arena.upload(hfs.data(), hfs.size(), &V.heightfields)) return fail();

	std::vector<HxrBitmapDev> bms((size_t)d->num_bitmaps);
	s->bitmap_prep.assign((size_t)d->num_bitmaps, 0);
	for (int64_t i = 0; i < d->num_bitmaps; i++) {
		bms[(size_t)i].width = d->bitmaps[i].width;
		bms[(size_t)i].height = d->bitmaps[i].height;
		bms[(size_t)i].texels = d_f32 + d->bitmaps[i].first_texel;
		const hxr_bitmap& b = d->bitmaps[i];
		const size_t texels = (size_t)b.width * b.height;
		s->bitmap_prep[(size_t)i] = b.prep | (b.first_gamma_table < 0 && (b.prep & HXR_BITMAP_GAMMA) ? 4 : 0);
		if (!b.prep || texels == 0) continue;
		// hxr_bitmap::prep: the loader's passes over the file's texels, here as kernels on the uploaded pool (in place; the derivative
		// reads neighbours, so it goes to a second table)
		float* t = const_cast<float*>(d_f32) + b.first_texel;
		const int grid = (int)std::min<size_t>((texels * 3 + 255) / 256, 4096);
		if (b.prep & HXR_BITMAP_GAMMA)
			hipLaunchKernelGGL(hxr_bitmap_gamma_kernel, dim3(grid), dim3(256), 0, nullptr, t, texels * 3, b.first_gamma_table >= 0 ? d_f32 + b.first_gamma_table : (const float*)nullptr,
			                   fabsf(b.gamma - 2.2f) < 1e-6f ? 1 : 0, b.gamma);
		if (b.prep & HXR_BITMAP_DIFFERENTIATE) {
			void* pd = nullptr;
			if (s->arena.alloc(texels * 3 * sizeof(float), &pd)) return fail();
			hipLaunchKernelGGL(hxr_bitmap_differentiate_kernel, dim3(grid), dim3(256), 0, nullptr, (const float*)t, b.width, b.height, (float*)pd);
			bms[(size_t)i].texels = (const float*)pd;
		}
		if (hipGetLastError() != hipSuccess) { set_err("bitmap preparation kernels failed to launch"); return fail(); }
	}
	if (s->arena.upload(bms.data(), bms.size(), &V.bitmaps)) return fail();

	std::vector<HxrNodeDev> nodes((size_t)d->num_nodes);
	for (int64_t i = 0; i < d->num_nodes; i++) {
		const hxr_node& n = d->nodes[i];
		const hxr_geometry& g = d->geometries[n.geometry];
		HxrNodeDev& o = nodes[(size_t)i];
		memset(&o, 0, sizeof o);
		o.T = n.T;
		if (g.type == HXR_GEOM_MESH) s->features |= HXR_F_MESH;
		else if (g.type == HXR_GEOM_HEIGHTFIELD) s->features |= HXR_F_HEIGHTFIELD;
		else if (g.type == HXR_GEOM_SPHERE) s->features |= HXR_F_SPHERE;
		else if (g.type == HXR_GEOM_CUBE) s->features |= HXR_F_CUBE;
		else if (g.type >= HXR_GEOM_CSG_UNION && g.type <= HXR_GEOM_CSG_DIFF) s->features |= HXR_F_EXTRA | (csg_depth(d, n.geometry, 1) > 1 ? HXR_F_CSGNEST : 0u);
		o.geom_type = g.type;
		o.geom = n.geometry;
		o.ref = g.ref;
		o.shader = n.shader;
		o.bump = n.bump;
		o.needs_uv = (n.bump >= 0 || shader_reads_textures(d, n.shader, 0)) ? 1 : 0;
		o.identity = 1;
		for (int r = 0; r < 3; r++)
			for (int c = 0; c < 3; c++) {
				const double want = r == c ? 1.0 : 0.0;
				if (n.T.m[r * 3 + c] != want || n.T.inv_m[r * 3 + c] != want || n.T.tinv_m[r * 3 + c] != want) o.identity = 0;
			}
		if (getenv("HXR_NO_IDENTITY_NODES")) o.identity = 0;   // developer aid: A/B
		memcpy(o.p, g.p, sizeof o.p);
		// object-space bounds of the geometry, if it has finite ones
		double lo[3], hi[3];
		bool bounded = true;
		if (g.type == HXR_GEOM_MESH) { memcpy(lo, d->meshes[g.ref].bbox_min, sizeof lo); memcpy(hi, d->meshes[g.ref].bbox_max, sizeof hi); }
		else if (g.type == HXR_GEOM_HEIGHTFIELD) { memcpy(lo, hfs[(size_t)g.ref].bbox_min, sizeof lo); memcpy(hi, hfs[(size_t)g.ref].bbox_max, sizeof hi); }   // the module's own copy: a raw heightfield's box is computed here
		else if (g.type == HXR_GEOM_SPHERE) for (int c = 0; c < 3; c++) { lo[c] = g.p[c] - g.p[3]; hi[c] = g.p[c] + g.p[3]; }
		else if (g.type == HXR_GEOM_CUBE) for (int c = 0; c < 3; c++) { lo[c] = g.p[c] - g.p[3]; hi[c] = g.p[c] + g.p[3]; }
		else bounded = false;   // planes (possibly unlimited) and CSG are always scanned
		for (int c = 0; c < 3 && bounded; c++) bounded = std::isfinite(lo[c]) && std::isfinite(hi[c]) && lo[c] <= hi[c];
		o.cullable = 0;
		if (g.type == HXR_GEOM_PLANE && !getenv("HXR_NO_PLANE_SIDE_TEST")) {
			// world-space side test of a plane node (node_query): s(p) = (p - offset) . c - py with c = the y-column of inv_m
			const double c[3] = {n.T.inv_m[1], n.T.inv_m[4], n.T.inv_m[7]};
			const double k = n.T.offset[0] * c[0] + n.T.offset[1] * c[1] + n.T.offset[2] * c[2] + g.p[0];
			const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
			if (std::isfinite(k) && std::isfinite(cn) && cn > 0) {
				memcpy(o.cull_lo, c, sizeof c);
				o.cull_hi[0] = k;
				o.cull_hi[1] = cn;
				o.cull_hi[2] = 0;
				o.cullable = 2;
			}
		}
		if (bounded) {
			double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300}, scale = 0;
			for (int c = 0; c < 3; c++) {   // the reference's own slacks: BBox::inside +-1e-6 (src/bbox.h:80-85), Cube inBounds +-1e-6 (src/geometry.cpp:87-95)
				double pad = 2e-6 + 1e-9 * std::max(std::fabs(lo[c]), std::fabs(hi[c]));
				lo[c] -= pad;
				hi[c] += pad;
			}
			for (int corner = 0; corner < 8; corner++) {
				double pt[3] = {corner & 1 ? hi[0] : lo[0], corner & 2 ? hi[1] : lo[1], corner & 4 ? hi[2] : lo[2]};
				for (int c = 0; c < 3; c++) {   // p * M + offset
					double w = pt[0] * n.T.m[0 * 3 + c] + pt[1] * n.T.m[1 * 3 + c] + pt[2] * n.T.m[2 * 3 + c] + n.T.offset[c];
					wlo[c] = std::min(wlo[c], w);
					whi[c] = std::max(whi[c], w);
					scale = std::max(scale, std::fabs(w));
				}
			}
			// sphere around the world box of the 8 transformed corners (it contains the transformed box, which is their convex hull)
			double r2 = 0;
			bool ok = true;
			for (int c = 0; c < 3; c++) {
				o.cull_lo[c] = 0.5 * (wlo[c] + whi[c]);
				double h = 0.5 * (whi[c] - wlo[c]);
				r2 += h * h;
				ok = ok && std::isfinite(o.cull_lo[c]);
			}
			double r = std::sqrt(r2) * (1 + 1e-9) + 1e-9 * (scale + 1.0);   // ~1e7 ulps of margin for the transform and the test's own rounding
			o.cull_hi[0] = r * r;
			o.cull_hi[1] = o.cull_hi[2] = 0;
			o.cullable = (ok && std::isfinite(o.cull_hi[0])) ? 1 : 0;
			// an untransformed mesh node: nothing is saved by the sphere test -- no matrix products to skip, and Mesh::intersect's own root-box test is as cheap and
			// leaves the wave just the same when no lane reaches the box (cornell16 46.95 -> 46.28 ms; HXR_CULL_IDENTITY_MESH=1: A/B aid)
			if (o.identity && g.type == HXR_GEOM_MESH && !getenv("HXR_CULL_IDENTITY_MESH")) o.cullable = 0;
		}
		{   // does a hit of this node spawn secondary rays?  (first-frame schedule, prepare_schedule)
			int rays = 0, pending = 1;
			bool glossy = false, cyclic = false;
			shader_fanout(d, n.shader, 0, rays, pending, glossy, cyclic);
			if (rays > 0) {
				if (o.cullable == 1) s->hot_spheres.push_back({{o.cull_lo[0], o.cull_lo[1], o.cull_lo[2]}, std::sqrt(o.cull_hi[0])});
				else s->hot_everywhere = true;
			}
		}
	}
	if (s->arena.upload(nodes.data(), nodes.size(), &V.nodes)) return fail();
	if (s->arena.upload(d->geometries, (size_t)d->num_geometries, &V.geometries)) return fail();
	{   // Refraction, glossy Reflection, Fresnel and Bumps textures count only where a node reaches them through its shader tree
		// or bump map (scene files define materials they do not use)
		std::vector<char> seen((size_t)d->num_shaders, 0);
		std::vector<int> todo;
		auto texture_used = [&](int t) {
			if (t >= 0 && t < d->num_textures && (d->textures[t].type == HXR_TEX_FRESNEL || d->textures[t].type == HXR_TEX_BUMPS)) s->features |= HXR_F_REFRACT;
		};
		for (int64_t i = 0; i < d->num_nodes; i++) {
			todo.push_back(d->nodes[i].shader);
			texture_used(d->nodes[i].bump);
		}
		while (!todo.empty()) {
			int sh = todo.back();
			todo.pop_back();
			if (sh < 0 || sh >= d->num_shaders || seen[(size_t)sh]) continue;
			seen[(size_t)sh] = 1;
			const hxr_shader& S = d->shaders[sh];
			texture_used(S.texture);
			if (S.type == HXR_SHADER_REFRACTION) s->features |= HXR_F_REFRACT;
			if (S.type == HXR_SHADER_REFLECTION && S.glossiness < 1.0f) s->features |= HXR_F_EXTRA;
			if (S.type == HXR_SHADER_LAYERED)
				for (int k = 0; k < S.num_layers; k++) {
					todo.push_back(d->layers[S.first_layer + k].shader);
					texture_used(d->layers[S.first_layer + k].texture);
				}
		}
	}
	for (int64_t i = 0; i < d->num_lights; i++)
		if (d->lights[i].type == HXR_LIGHT_RECT) s->features |= HXR_F_RECT;
	if (s->arena.upload(d->textures, (size_t)d->num_textures, &V.textures)) return fail();
	if (s->arena.upload(d->shaders, (size_t)d->num_shaders, &V.shaders)) return fail();
	if (s->arena.upload(d->layers, (size_t)d->num_layers, &V.layers)) return fail();
	if (s->arena.upload(d->lights, (size_t)d->num_lights, &V.lights)) return fail();

	const int32_t* d_sides = nullptr;
	if (s->arena.upload(d->environment.side_bitmap, 6, &d_sides)) return fail();
	V.env_sides = d_sides;
	void* p = nullptr;
	// one block: the dispenser counters, the schedule's control words and its part-slot counters -- ONE clear per frame (three fills cost a 1 ms launch 1 % each)
	if (s->arena.alloc(HXR_CONTROL_CLEARED_BYTES + (size_t)HXR_PART_SLOTS * sizeof(uint32_t), &p)) return fail();
	s->d_work = (uint32_t*)p;
	s->d_sched_ctl = (unsigned long long*)(s->d_work + 1024);
	s->d_sched_part = s->d_work + 1024 + 32;
	if (s->arena.alloc(sizeof(hxr_counters), &p)) return fail();
	s->d_counters = (unsigned long long*)p;
#if defined(HXR_LEAF_STATS) || defined(HXR_TURN_STATS) || defined(HXR_LEAF_HIST) || defined(HXR_WALK_STATS)
	V.dev_stats = s->d_counters + HC_NUM;
#endif
	const HxrSceneDev* dv = nullptr;
	if (s->arena.upload(&V, 1, &dv)) return fail();
	s->dev = const_cast<HxrSceneDev*>(dv);
	// sRGB table, built with the same expression as Color::init_sRGB_cache (src/sdl.cpp:406-412, src/color.h:36-47)
	uint8_t lut[4097];
	for (int i = 0; i <= 4096; i++) {
		float x = i / 4096.0f;
		unsigned v;
		if (x <= 0) v = 0;
		else if (x >= 1) v = 255;
		else {
			x = (x <= 0.0031308f) ? x * 12.02f : (1.0f + 0.055f) * powf(x, 1.0f / 2.4f) - 0.055f;
			v = (unsigned)(int)floorf(x * 255.0f + 0.5f);
		}
		lut[i] = (uint8_t)v;
	}
	const uint8_t* dl = nullptr;
	if (s->arena.upload(lut, 4097, &dl)) return fail();
	s->d_lut = const_cast<uint8_t*>(dl);
	for (int pass = 0; pass < 2; pass++)
		for (int k = 0; k < 4; k++)
			if (hipEventCreate(&s->ev_pass[pass][k]) != hipSuccess) { set_err("hipEventCreate failed"); return fail(); }
	if (pick_variant(s) == V_MESH && ensure_parts_stream(s)) return fail();   // (here, not inside a frame: creating a stream costs milliseconds)
	// the preparation kernels above ran on the null stream; frames may be rendered on streams that do not wait for it
	if (hipDeviceSynchronize() != hipSuccess) { set_err("scene preparation kernels failed: %s", hipGetErrorString(hipGetLastError())); return fail(); }
	return s;
}

int hxr_scene_update_camera(hxr_scene* s, const hxr_camera* cam)
{
	if (!s || !cam) { set_err("null argument"); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(s->device));
	s->camera = *cam;
	s->host_view.camera = *cam;
	HIP_TRY(hipMemcpy((char*)s->dev + offsetof(HxrSceneDev, camera), cam, sizeof *cam, hipMemcpyHostToDevice));
	return 0;
}

size_t hxr_output_bytes(const hxr_scene* s, const hxr_frame_params* fp, size_t row_stride_bytes)
{
	Plan p;
	if (resolve(s, fp, p)) return 0;
	if (p.layout == HXR_LAYOUT_TILES) return (size_t)p.ntiles * p.tile * p.tile * 12;
	return (size_t)p.H * row_stride_bytes;
}

int hxr_shard_tile_count(const hxr_scene* s, const hxr_frame_params* fp)
{
	Plan p;
	int rc = resolve(s, fp, p);
	return rc ? rc : p.ntiles;
}

int hxr_shard_tile_rect(const hxr_scene* s, const hxr_frame_params* fp, int i, int* x0, int* y0, int* x1, int* y1)
{
	Plan p;
	int rc = resolve(s, fp, p);
	if (rc) return rc;
	if (i < 0 || i >= p.ntiles) { set_err("tile index outside the shard"); return HXR_E_INVALID; }
	int t = i * p.count + p.rank, tx_, ty_;
	hxr_tile_position(t, p.tiles_x, p.count, &tx_, &ty_);
	*x0 = tx_ * p.tile;
	*y0 = ty_ * p.tile;
	*x1 = std::min(*x0 + p.tile, p.W);
	*y1 = std::min(*y0 + p.tile, p.H);
	return 0;
}

int hxr_render_device(hxr_scene* s, const hxr_frame_params* fp, float* d_out, size_t row_stride_bytes, void* hip_stream)
{
	if (!s || !fp || !d_out) { set_err("null argument"); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(s->device));
	return render_on_stream(s, fp, d_out, row_stride_bytes, (hipStream_t)hip_stream);
}

int hxr_render(hxr_scene* s, const hxr_frame_params* fp, float* vfb, size_t row_stride_bytes, const volatile int* cancel,
               hxr_progress_cb cb, void* user)
{
	if (!s || !fp || !vfb) { set_err("null argument"); return HXR_E_INVALID; }
	Plan p;
	int rc = resolve(s, fp, p);
	if (rc) return rc;
	if (p.layout != HXR_LAYOUT_FRAME) { set_err("hxr_render writes FRAME layout only"); return HXR_E_INVALID; }
	if (cancel && *cancel) return HXR_CANCELLED;
	HIP_TRY(hipSetDevice(s->device));
	size_t dstride = (size_t)p.W * 12;
	rc = ensure((void**)&s->d_frame, &s->frame_bytes, dstride * p.H);
	if (rc) return rc;
	auto check_queue = [&]() -> int {
		for (int pass = 0; pass < s->passes_timed; pass++)
			if (s->pass_deferred[pass]) {   // cannot happen (the queue is sized for the worst case); a frame with dropped rays must not pass silently
				uint32_t overflow = 0;
				HIP_TRY(hipMemcpy(&overflow, s->d_qctl + (size_t)pass * QCTL_WORDS + 2 * HXR_QUEUE_SHARDS * HXR_SHARD_STRIDE, sizeof overflow, hipMemcpyDeviceToHost));
				if (overflow) { set_err("the deferred-ray queue overflowed"); return HXR_E_HIP; }
			}
		return 0;
	};
	if (!cancel && !cb) {   // nothing to watch: one stream, one copy
		if (p.count > 1) HIP_TRY(hipMemsetAsync(s->d_frame, 0, dstride * p.H, nullptr));
		rc = render_on_stream(s, fp, s->d_frame, dstride, nullptr);
		if (rc) return rc;
		HIP_TRY(hipMemcpy2D(vfb, row_stride_bytes, s->d_frame, dstride, dstride, p.H, hipMemcpyDeviceToHost));
		return check_queue();
	}

	// Watched frame.  The kernels run on a stream of their own; this thread polls the caller's cancel flag (the reference polls
	// checkForUserExit() per bucket row, src/main.cpp:280,333,370) and the per-tile pixel counts the render kernel keeps
	// (displayVFBRect per finished bucket, src/main.cpp:312,379), and talks to the GPU through a second stream: raising the device-side
	// cancel word makes every wavefront stop taking packets, finished tiles are copied out and reported while the rest still renders.
	if (!s->main_stream) {
		HIP_TRY(hipStreamCreateWithFlags(&s->main_stream, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
	}
	const size_t words = 16 + (size_t)std::max(p.ntiles, 1);
	if (s->watch_words < words) {
		if (s->d_watch) (void)hipFree(s->d_watch);
		if (s->h_watch) (void)hipHostFree(s->h_watch);
		s->d_watch = nullptr; s->h_watch = nullptr; s->watch_words = 0;
		HIP_TRY(hipMalloc((void**)&s->d_watch, words * sizeof(uint32_t)));
		HIP_TRY(hipHostMalloc((void**)&s->h_watch, words * sizeof(uint32_t), hipHostMallocDefault));
		s->watch_words = words;
	}
	HIP_TRY(hipMemsetAsync(s->d_watch, 0, words * sizeof(uint32_t), s->main_stream));
	// The poll below reads these words on aux_stream, which nothing else orders against main_stream: without this the first look could
	// run before the clear and see hipMalloc's garbage or the previous frame's "every tile complete" counts.
	HIP_TRY(hipEventRecord(s->ev_done, s->main_stream));
	HIP_TRY(hipStreamWaitEvent(s->aux_stream, s->ev_done, 0));
	rc = ensure_cancel_word(s);
	if (rc) return rc;
	if (p.count > 1) HIP_TRY(hipMemsetAsync(s->d_frame, 0, dstride * p.H, s->main_stream));
	// tiles can be handed over one by one when the render kernel writes final pixels straight into d_frame
	const bool final_pixels = (p.mode == HXR_MODE_SINGLE || p.mode == HXR_MODE_MONTE_CARLO);
	s->watch_cancel = cancel ? s->d_cancel : nullptr;
	s->watch_tiles = cb ? s->d_watch + 16 : nullptr;
	rc = render_on_stream(s, fp, s->d_frame, dstride, s->main_stream);
	const bool deferred = s->pass_deferred[0];   // then pixels are final only after the resolve pass
	s->watch_cancel = nullptr;
	s->watch_tiles = nullptr;
	if (rc) return rc;
	HIP_TRY(hipEventRecord(s->ev_done, s->main_stream));
	std::vector<char> reported((size_t)p.ntiles, 0);
	auto tile_rect = [&](int i, int& x0, int& y0, int& x1, int& y1) {
		int t = i * p.count + p.rank, tx_, ty_;
		hxr_tile_position(t, p.tiles_x, p.count, &tx_, &ty_);
		x0 = tx_ * p.tile;
		y0 = ty_ * p.tile;
		x1 = std::min(x0 + p.tile, p.W);
		y1 = std::min(y0 + p.tile, p.H);
	};
	auto hand_over = [&](int i) -> int {   // copy one finished tile into the caller's frame and report it
		int x0, y0, x1, y1;
		tile_rect(i, x0, y0, x1, y1);
		HIP_TRY(hipMemcpy2DAsync((char*)vfb + (size_t)y0 * row_stride_bytes + (size_t)x0 * 12, row_stride_bytes, (char*)s->d_frame + (size_t)y0 * dstride + (size_t)x0 * 12,
		                         dstride, (size_t)(x1 - x0) * 12, (size_t)(y1 - y0), hipMemcpyDeviceToHost, s->aux_stream));
		HIP_TRY(hipStreamSynchronize(s->aux_stream));
		reported[(size_t)i] = 1;
		cb(user, x0, y0, x1, y1);
		return 0;
	};
	bool signalled = false;
	for (;;) {
		hipError_t q = hipEventQuery(s->ev_done);
		if (q == hipSuccess) break;
		if (q != hipErrorNotReady) { set_err("hipEventQuery failed: %s", hipGetErrorString(q)); return HXR_E_HIP; }
		if (cancel && *cancel && !signalled) {
			__atomic_store_n(s->h_cancel, 1u, __ATOMIC_SEQ_CST);   // the waves read this word itself
			signalled = true;
		}
		if (cb && final_pixels && !deferred && !signalled) {
			HIP_TRY(hipMemcpyAsync(s->h_watch + 16, s->d_watch + 16, (size_t)p.ntiles * sizeof(uint32_t), hipMemcpyDeviceToHost, s->aux_stream));
			HIP_TRY(hipStreamSynchronize(s->aux_stream));
			for (int i = 0; i < p.ntiles; i++) {
				if (reported[(size_t)i]) continue;
				int x0, y0, x1, y1;
				tile_rect(i, x0, y0, x1, y1);
				if (s->h_watch[16 + i] >= (uint32_t)((x1 - x0) * (y1 - y0))) {
					rc = hand_over(i);
					if (rc) return rc;
				}
			}
		}
		struct timespec nap = {0, 100 * 1000};   // 0.1 ms between looks
		nanosleep(&nap, nullptr);
	}
	HIP_TRY(hipStreamSynchronize(s->main_stream));
	if (signalled || (cancel && *cancel)) return HXR_CANCELLED;   // like render() returning false: the frame is incomplete
	// whatever has not gone out tile by tile (adaptive-AA and deferred frames: everything)
	bool any_reported = false;
	for (char r : reported) any_reported = any_reported || r;
	if (!any_reported) {
		HIP_TRY(hipMemcpy2D(vfb, row_stride_bytes, s->d_frame, dstride, dstride, p.H, hipMemcpyDeviceToHost));
		if (cb)
			for (int i = 0; i < p.ntiles; i++) {
				int x0, y0, x1, y1;
				tile_rect(i, x0, y0, x1, y1);
				cb(user, x0, y0, x1, y1);
			}
	} else {
		for (int i = 0; i < p.ntiles; i++)
			if (!reported[(size_t)i]) {
				rc = hand_over(i);
				if (rc) return rc;
			}
	}
	return check_queue();
}

int hxr_interactive_frame(hxr_scene* s, const hxr_frame_params* fp, const hxr_camera* cam, uint32_t* surface, size_t pitch_bytes, int rshift, int gshift, int bshift,
                          double* latency_ms)
{
	if (!s || !fp || !surface) { set_err("null argument"); return HXR_E_INVALID; }
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	Plan p;
	int rc = resolve(s, fp, p);
	if (rc) return rc;
	if (p.layout != HXR_LAYOUT_FRAME || p.count > 1) { set_err("the interactive frame is an unsharded FRAME-layout frame"); return HXR_E_INVALID; }
	if (pitch_bytes < (size_t)p.W * 4 || rshift < 0 || gshift < 0 || bshift < 0 || rshift > 24 || gshift > 24 || bshift > 24) { set_err("bad surface format"); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(s->device));
	if (!s->main_stream) {
		HIP_TRY(hipStreamCreateWithFlags(&s->main_stream, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
	}
	const size_t dstride = (size_t)p.W * 12;
	rc = ensure((void**)&s->d_frame, &s->frame_bytes, dstride * p.H);
	if (!rc) rc = ensure((void**)&s->d_surface, &s->surface_bytes, (size_t)p.W * p.H * 4);
	if (rc) return rc;
	if (cam) {   // the camera record is the only part of the scene that changes between the frames of the loop
		s->camera = *cam;
		s->host_view.camera = *cam;
		HIP_TRY(hipMemcpyAsync((char*)s->dev + offsetof(HxrSceneDev, camera), &s->camera, sizeof s->camera, hipMemcpyHostToDevice, s->main_stream));
	}
	rc = render_on_stream(s, fp, s->d_frame, dstride, s->main_stream);
	if (rc) return rc;
	const size_t total = (size_t)p.W * p.H;
	const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
	hipLaunchKernelGGL(hxr_quantize_rgb32_kernel, dim3(grid), dim3(256), 0, s->main_stream, (const float*)s->d_frame, dstride, p.W, p.H, (const uint8_t*)s->d_lut, rshift, gshift,
	                   bshift, s->d_surface, (size_t)p.W * 4);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpy2DAsync(surface, pitch_bytes, s->d_surface, (size_t)p.W * 4, (size_t)p.W * 4, (size_t)p.H, hipMemcpyDeviceToHost, s->main_stream));
	HIP_TRY(hipStreamSynchronize(s->main_stream));
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (latency_ms) *latency_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
	return 0;
}

int hxr_interactive_float_frame(hxr_scene* s, const hxr_frame_params* fp, float* vfb, size_t row_stride_bytes)
{
	if (!s || !fp || !vfb) { set_err("null argument"); return HXR_E_INVALID; }
	Plan p;
	int rc = resolve(s, fp, p);
	if (rc) return rc;
	const size_t dstride = (size_t)p.W * 12;
	if (!s->d_frame || s->frame_bytes < dstride * p.H) { set_err("no interactive frame of this size has been rendered"); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipMemcpy2D(vfb, row_stride_bytes, s->d_frame, dstride, dstride, p.H, hipMemcpyDeviceToHost));
	return 0;
}

int hxr_assemble_frame(const hxr_scene* s, const hxr_frame_params* fp, const float* d_gathered, size_t per_rank_bytes, float* d_frame,
                       size_t row_stride_bytes, void* hip_stream)
{
	Plan p;
	int rc = resolve(s, fp, p);
	if (rc) return rc;
	if (!d_gathered || !d_frame) { set_err("null argument"); return HXR_E_INVALID; }
	size_t total = (size_t)p.tiles_x * p.tiles_y * p.tile * p.tile;
	int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
	hipLaunchKernelGGL(hxr_assemble_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, d_gathered, per_rank_bytes / sizeof(float), d_frame,
	                   row_stride_bytes, p.W, p.H, p.tile, p.tiles_x, p.tiles_y, p.count);
	HIP_TRY(hipGetLastError());
	return 0;
}

int hxr_quantize_srgb8(const float* d_frame, size_t row_stride_bytes, int width, int height, uint8_t* d_bgr, void* hip_stream)
{
	if (!d_frame || !d_bgr || width <= 0 || height <= 0) { set_err("bad argument"); return HXR_E_INVALID; }
	static uint8_t* lut = nullptr;   // process-wide copy of the table
	if (!lut) {
		uint8_t host[4097];
		for (int i = 0; i <= 4096; i++) {
			float x = i / 4096.0f;
			unsigned v;
			if (x <= 0) v = 0;
			else if (x >= 1) v = 255;
			else {
				x = (x <= 0.0031308f) ? x * 12.02f : (1.0f + 0.055f) * powf(x, 1.0f / 2.4f) - 0.055f;
				v = (unsigned)(int)floorf(x * 255.0f + 0.5f);
			}
			host[i] = (uint8_t)v;
		}
		HIP_TRY(hipMalloc((void**)&lut, sizeof host));
		HIP_TRY(hipMemcpy(lut, host, sizeof host, hipMemcpyHostToDevice));
	}
	size_t total = (size_t)width * height;
	int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
	hipLaunchKernelGGL(hxr_quantize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, d_frame, row_stride_bytes, width, height,
	                   (const uint8_t*)lut, d_bgr);
	HIP_TRY(hipGetLastError());
	return 0;
}

int hxr_sah_build(const hxr_scene_desc* d, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf, int64_t* num_nodes,
                  int64_t* num_leaf, int* max_depth)
{
	if (!d || mesh < 0 || mesh >= d->num_meshes) { set_err("no such mesh"); return HXR_E_INVALID; }
	const hxr_mesh& m = d->meshes[mesh];
	if (m.first_triangle < 0 || m.num_triangles < 0 || m.first_triangle + m.num_triangles > d->num_triangles || m.first_vertex < 0) { set_err("bad mesh ranges"); return HXR_E_INVALID; }
	hxr_sah::SahTree t;
	hxr_sah::SahBuilder(d->triangles + m.first_triangle, m.num_triangles, d->f64_pool + m.first_vertex * 3, hxr_sah::SahParams()).build(m.bbox_min, m.bbox_max, t);
	if (num_nodes) *num_nodes = (int64_t)t.nodes.size();
	if (num_leaf) *num_leaf = (int64_t)t.leaf_indices.size();
	if (max_depth) *max_depth = t.max_depth;
	if (!nodes && !leaf_indices) return 0;
	if (!nodes || !leaf_indices || max_nodes < (int64_t)t.nodes.size() || max_leaf < (int64_t)t.leaf_indices.size()) { set_err("output arrays too small"); return HXR_E_INVALID; }
	memcpy(nodes, t.nodes.data(), t.nodes.size() * sizeof(hxr_kd_node));
	if (!t.leaf_indices.empty()) memcpy(leaf_indices, t.leaf_indices.data(), t.leaf_indices.size() * sizeof(int32_t));
	return 0;
}

static int build_tree_on_device(bool sah, const hxr_scene_desc* d, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf,
                                int64_t* num_nodes, int64_t* num_leaf, int* max_depth, double* build_ms);

int hxr_kd_build_device(const hxr_scene_desc* d, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf, int64_t* num_nodes,
                        int64_t* num_leaf, int* max_depth, double* build_ms)
{
	return build_tree_on_device(false, d, mesh, nodes, max_nodes, leaf_indices, max_leaf, num_nodes, num_leaf, max_depth, build_ms);
}

int hxr_sah_build_device(const hxr_scene_desc* d, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf, int64_t* num_nodes,
                         int64_t* num_leaf, int* max_depth, double* build_ms)
{
	return build_tree_on_device(true, d, mesh, nodes, max_nodes, leaf_indices, max_leaf, num_nodes, num_leaf, max_depth, build_ms);
}

static int build_tree_on_device(bool sah, const hxr_scene_desc* d, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf,
                                int64_t* num_nodes, int64_t* num_leaf, int* max_depth, double* build_ms)
{
	if (g_device < 0) { set_err("hxr_init has not been called"); return HXR_E_INVALID; }
	if (!d || mesh < 0 || mesh >= d->num_meshes) { set_err("no such mesh"); return HXR_E_INVALID; }
	const hxr_mesh& m = d->meshes[mesh];
	if (m.first_triangle < 0 || m.num_triangles < 0 || m.first_triangle + m.num_triangles > d->num_triangles || m.first_vertex < 0 || m.num_vertices < 0 ||
	    (m.first_vertex + m.num_vertices) * 3 > d->f64_pool_size) { set_err("bad mesh ranges"); return HXR_E_INVALID; }
	for (int64_t i = 0; i < m.num_triangles; i++)
		for (int k = 0; k < 3; k++)
			if (d->triangles[m.first_triangle + i].v[k] < 0 || d->triangles[m.first_triangle + i].v[k] >= m.num_vertices) { set_err("triangle %lld refers to a vertex outside the mesh", (long long)i); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(g_device));
	hxr_triangle* d_tris = nullptr;
	double* d_verts = nullptr;
	HIP_TRY(hipMalloc((void**)&d_tris, std::max<size_t>(1, (size_t)m.num_triangles) * sizeof(hxr_triangle)));
	if (hipMalloc((void**)&d_verts, std::max<size_t>(1, (size_t)m.num_vertices * 3) * sizeof(double)) != hipSuccess) { (void)hipFree(d_tris); set_err("out of device memory"); return HXR_E_HIP; }
	int rc = 0;
	hxr_kd::DeviceBuiltTree t;
	char why[256] = "";
	if (hipMemcpy(d_tris, d->triangles + m.first_triangle, (size_t)m.num_triangles * sizeof(hxr_triangle), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(d_verts, d->f64_pool + m.first_vertex * 3, (size_t)m.num_vertices * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
		set_err("upload of the mesh failed");
		rc = HXR_E_HIP;
	} else if (sah) {
		const hxr_sah::SahParams sp;
		if ((rc = hxr_kd::build_sah_tree(d_tris, m.num_triangles, d_verts, m.bbox_min, m.bbox_max, sp.traversal_cost, sp.intersect_cost, sp.empty_bonus, sp.max_depth, sp.min_leaf,
		                                 t, why, sizeof why)) != 0)
			set_err("SAH build: %s", why);
	} else if ((rc = hxr_kd::build_reference_tree(d_tris, m.num_triangles, d_verts, m.bbox_min, m.bbox_max, t, why, sizeof why)) != 0) {
		set_err("kd build: %s", why);
	}
	(void)hipFree(d_tris);
	(void)hipFree(d_verts);
	if (rc) return rc;
	if (num_nodes) *num_nodes = (int64_t)t.nodes.size();
	if (num_leaf) *num_leaf = (int64_t)t.leaf_indices.size();
	if (max_depth) *max_depth = t.max_depth;
	if (build_ms) *build_ms = t.build_ms;
	if (!nodes && !leaf_indices) return 0;
	if (!nodes || !leaf_indices || max_nodes < (int64_t)t.nodes.size() || max_leaf < (int64_t)t.leaf_indices.size()) { set_err("output arrays too small"); return HXR_E_INVALID; }
	memcpy(nodes, t.nodes.data(), t.nodes.size() * sizeof(hxr_kd_node));
	if (!t.leaf_indices.empty()) memcpy(leaf_indices, t.leaf_indices.data(), t.leaf_indices.size() * sizeof(int32_t));
	return 0;
}

int hxr_scene_verify_tables(const hxr_scene* s, const hxr_scene_desc* d, int64_t* mismatches)
{
	if (!s || !d || !mismatches) { set_err("null argument"); return HXR_E_INVALID; }
	*mismatches = 0;
	HIP_TRY(hipSetDevice(s->device));
	const HxrSceneDev& V = s->host_view;
	// hot triangle records: re-derive every record on the host from the descriptor, in the order the device holds
	std::vector<HxrMeshDev> meshes((size_t)d->num_meshes);
	if (d->num_meshes) HIP_TRY(hipMemcpy(meshes.data(), V.meshes, meshes.size() * sizeof(HxrMeshDev), hipMemcpyDeviceToHost));
	for (int64_t i = 0; i < d->num_meshes; i++) {
		const hxr_mesh& m = d->meshes[i];
		const int64_t first = meshes[(size_t)i].hot_first;
		// the number of hot records of a mesh = leaf entries of the tree walked (or its triangles, without a tree)
		int64_t n = m.num_triangles;
		if ((size_t)i < s->tree_stats.size() && meshes[(size_t)i].has_kd) n = s->tree_stats[(size_t)i].leaf_entries;
		if (n <= 0) continue;
		std::vector<HxrTriHot> hot((size_t)n);
		std::vector<int32_t> ids((size_t)n);
		HIP_TRY(hipMemcpy(hot.data(), V.hot + first, hot.size() * sizeof(HxrTriHot), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(ids.data(), V.hot_ids + first, ids.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
		const hxr_triangle* tris = d->triangles + m.first_triangle;
		const double* verts = d->f64_pool + m.first_vertex * 3;
		for (int64_t k = 0; k < n; k++) {
			if (ids[(size_t)k] < 0 || ids[(size_t)k] >= m.num_triangles) { ++*mismatches; continue; }
			const hxr_triangle& t = tris[ids[(size_t)k]];
			HxrTriHot w;
			for (int c = 0; c < 3; c++) { w.A[c] = verts[(size_t)t.v[0] * 3 + c]; w.AB[c] = t.AB[c]; w.AC[c] = t.AC[c]; w.N[c] = t.ABcrossAC[c]; }
			if (memcmp(&w, &hot[(size_t)k], sizeof w) != 0) ++*mismatches;
		}
	}
	// the derived fields of the triangle records (computed on the GPU with device_prepare_triangles, copies of the descriptor's otherwise)
	if (d->num_triangles > 0 && s->d_triangles) {
		std::vector<hxr_triangle> got((size_t)d->num_triangles);
		HIP_TRY(hipMemcpy(got.data(), s->d_triangles, got.size() * sizeof(hxr_triangle), hipMemcpyDeviceToHost));
		auto same = [](const double* a, const double* b) {   // bit-equal, or both NaN (degenerate uv triangles: 0/0 on both sides)
			for (int c = 0; c < 3; c++)
				if (memcmp(a + c, b + c, sizeof(double)) != 0 && !(a[c] != a[c] && b[c] != b[c])) return false;
			return true;
		};
		int64_t per_field[6] = {0, 0, 0, 0, 0, 0};
		for (int64_t k = 0; k < d->num_triangles; k++) {
			const hxr_triangle &a = got[(size_t)k], &b = d->triangles[k];
			const bool ok[6] = {same(a.gnormal, b.gnormal), same(a.AB, b.AB), same(a.AC, b.AC), same(a.ABcrossAC, b.ABcrossAC), same(a.dNdx, b.dNdx), same(a.dNdy, b.dNdy)};
			bool all = true;
			for (int f = 0; f < 6; f++) { if (!ok[f]) per_field[f]++; all = all && ok[f]; }
			if (!all) ++*mismatches;
		}
		if (getenv("HXR_VERIFY_VERBOSE"))   // developer aid: which derived field disagrees
			fprintf(stderr, "[hxr] triangle records differing in gnormal %lld, AB %lld, AC %lld, ABcrossAC %lld, dNdx %lld, dNdy %lld (of %lld)\n", (long long)per_field[0],
			        (long long)per_field[1], (long long)per_field[2], (long long)per_field[3], (long long)per_field[4], (long long)per_field[5], (long long)d->num_triangles);
	}
	// heightfield maxima tables against the descriptor's own (when it carries them)
	std::vector<HxrHeightfieldDev> hfs((size_t)d->num_heightfields);
	if (d->num_heightfields) HIP_TRY(hipMemcpy(hfs.data(), V.heightfields, hfs.size() * sizeof(HxrHeightfieldDev), hipMemcpyDeviceToHost));
	for (int64_t i = 0; i < d->num_heightfields; i++) {
		const hxr_heightfield& h = d->heightfields[i];
		if (h.first_maxh < 0 || h.first_highmap < 0) continue;
		const size_t cells = (size_t)h.W * h.H;
		std::vector<float> mh(cells), hm(cells * 16);
		HIP_TRY(hipMemcpy(mh.data(), hfs[(size_t)i].max_h, mh.size() * sizeof(float), hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(hm.data(), hfs[(size_t)i].high_map, hm.size() * sizeof(float), hipMemcpyDeviceToHost));
		if (hfs[(size_t)i].max_k != h.max_k) ++*mismatches;
		// a scene created from raw heights blurred them itself, in another summation order than the host: heights, and the maxima of
		// heights, then agree to float rounding (a few 1e-7 of the height range) instead of bit for bit
		const bool raw = (size_t)i < s->hf_raw.size() && s->hf_raw[(size_t)i] && !h.raw_heights;
		const float tol = raw ? 1e-5f * (float)std::max(1.0, h.bbox_max[1] - h.bbox_min[1]) : 0.0f;
		auto differs = [&](float a, float b) { return raw ? !(std::fabs(a - b) <= tol) : a != b; };
		for (size_t c = 0; c < cells; c++) {
			if (differs(mh[c], d->f32_pool[h.first_maxh + (int64_t)c])) ++*mismatches;
			for (int k = 0; k < h.max_k; k++)
				if (differs(hm[c * 16 + k], d->f32_pool[h.first_highmap + (int64_t)(c * 16 + k)])) ++*mismatches;
		}
		if (raw) {
			std::vector<float> hts(cells);
			std::vector<double> nrm(cells * 3);
			HIP_TRY(hipMemcpy(hts.data(), hfs[(size_t)i].heights, hts.size() * sizeof(float), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(nrm.data(), hfs[(size_t)i].normals, nrm.size() * sizeof(double), hipMemcpyDeviceToHost));
			for (size_t c = 0; c < cells; c++) {
				if (differs(hts[c], d->f32_pool[h.first_height + (int64_t)c])) ++*mismatches;
				for (int k = 0; k < 3; k++)
					if (!(std::fabs(nrm[c * 3 + k] - d->f64_pool[(h.first_normal + (int64_t)c) * 3 + k]) <= 1e-5)) ++*mismatches;
			}
			for (int k = 0; k < 3; k++)
				if (!(std::fabs(hfs[(size_t)i].bbox_min[k] - h.bbox_min[k]) <= tol) || !(std::fabs(hfs[(size_t)i].bbox_max[k] - h.bbox_max[k]) <= tol)) ++*mismatches;
		}
	}
	// bitmaps the module decoded / differentiated itself (hxr_bitmap::prep) against the complete descriptor's: bit for bit, except where the
	// gamma curve had no host table (float texels: the device's powf, compared to 4 ulp-ish of the value)
	std::vector<HxrBitmapDev> bms((size_t)d->num_bitmaps);
	if (d->num_bitmaps) HIP_TRY(hipMemcpy(bms.data(), V.bitmaps, bms.size() * sizeof(HxrBitmapDev), hipMemcpyDeviceToHost));
	for (int64_t i = 0; i < d->num_bitmaps; i++) {
		if ((size_t)i >= s->bitmap_prep.size() || !s->bitmap_prep[(size_t)i]) continue;
		const hxr_bitmap& b = d->bitmaps[i];
		if (b.prep) continue;   // `desc` is not the complete one for this bitmap
		if (b.width != bms[(size_t)i].width || b.height != bms[(size_t)i].height) { ++*mismatches; continue; }
		std::vector<float> got((size_t)b.width * b.height * 3);
		if (got.empty()) continue;
		HIP_TRY(hipMemcpy(got.data(), bms[(size_t)i].texels, got.size() * sizeof(float), hipMemcpyDeviceToHost));
		const float* want = d->f32_pool + b.first_texel;
		const bool curve = (s->bitmap_prep[(size_t)i] & 4) != 0;
		for (size_t k = 0; k < got.size(); k++) {
			const bool same = curve ? std::fabs(got[k] - want[k]) <= 1e-6f * std::max(1.0f, std::fabs(want[k])) : memcmp(&got[k], &want[k], sizeof(float)) == 0;
			if (!same) ++*mismatches;
		}
	}
	return 0;
}

int hxr_scene_tree_stats(const hxr_scene* s, int mesh, int64_t* nodes, int64_t* leaf_entries, int* max_depth)
{
	if (!s || mesh < 0 || (size_t)mesh >= s->tree_stats.size()) { set_err("no such mesh"); return HXR_E_INVALID; }
	if (nodes) *nodes = s->tree_stats[(size_t)mesh].nodes;
	if (leaf_entries) *leaf_entries = s->tree_stats[(size_t)mesh].leaf_entries;
	if (max_depth) *max_depth = s->tree_stats[(size_t)mesh].depth;
	return 0;
}

int hxr_get_counters(const hxr_scene* s, hxr_counters* out)
{
	if (!s || !out) { set_err("null argument"); return HXR_E_INVALID; }
	memset(out, 0, sizeof *out);
	if (!s->counters_valid) { set_err("the last frame was rendered without collect_counters"); return HXR_E_INVALID; }
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipDeviceSynchronize());
	unsigned long long raw[HC_NUM + PH_NUM];
	HIP_TRY(hipMemcpy(raw, s->d_counters, sizeof raw, hipMemcpyDeviceToHost));
	uint64_t* o = (uint64_t*)out;
	for (int i = 0; i < HC_NUM; i++) o[i] = raw[i];
	static_assert(sizeof(hxr_counters) >= (HC_NUM + PH_NUM) * sizeof(uint64_t), "phase clocks must fit the reserved words");
	for (int i = 0; i < PH_NUM; i++) out->reserved_[i] = raw[HC_NUM + i];   // zero unless built with -DHXR_PHASE_TIMERS
	uint32_t flagged = 0, flagged_heavy = 0;
	HIP_TRY(hipMemcpy(&flagged, s->d_work + 512, sizeof flagged, hipMemcpyDeviceToHost));
	if (s->aa_cost_aware) HIP_TRY(hipMemcpy(&flagged_heavy, (uint32_t*)s->d_sched_ctl + 25, sizeof flagged_heavy, hipMemcpyDeviceToHost));   // the back of the list
	out->aa_pixels = (uint64_t)flagged + flagged_heavy;
	return 0;
}

double hxr_last_kernel_ms(const hxr_scene* s)
{
	if (!s || !s->timed) return -1.0;
	if (hipEventSynchronize(s->ev1) != hipSuccess) return -1.0;
	float ms = 0;
	if (hipEventElapsedTime(&ms, s->ev0, s->ev1) != hipSuccess) return -1.0;
	return ms;
}

#ifdef HXR_WAVE_TIMELINE
// developer aid (tools/wave_timeline.py): the four words per wave the render kernel of a -DHXR_WAVE_TIMELINE build leaves 8 MB into the side buffer
int hxr_debug_wave_timeline(hxr_scene* s, unsigned long long* dst, size_t waves)
{
	if (!s->d_sched_side) return HXR_E_INVALID;
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(dst, (char*)s->d_sched_side + (8u << 20), waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemset((char*)s->d_sched_side + (8u << 20), 0, waves * 4 * sizeof(unsigned long long)));
	return 0;
}
#endif

int hxr_get_frame_timing(const hxr_scene* s, hxr_frame_timing* out)
{
	if (!s || !out) { set_err("null argument"); return HXR_E_INVALID; }
	memset(out, 0, sizeof *out);
	if (!s->timed || s->passes_timed <= 0) { set_err("no frame has been rendered"); return HXR_E_INVALID; }
	HIP_TRY(hipEventSynchronize(s->ev1));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
	out->total_ms = ms;
	for (int pass = 0; pass < s->passes_timed; pass++) {
		float a = 0, b = 0, c = 0;
		if (s->pass_deferred[pass]) {
			HIP_TRY(hipEventElapsedTime(&a, s->ev_pass[pass][0], s->ev_pass[pass][1]));
			HIP_TRY(hipEventElapsedTime(&b, s->ev_pass[pass][1], s->ev_pass[pass][2]));
			HIP_TRY(hipEventElapsedTime(&c, s->ev_pass[pass][2], s->ev_pass[pass][3]));
		} else HIP_TRY(hipEventElapsedTime(&a, s->ev_pass[pass][0], s->ev_pass[pass][3]));
		out->render_ms += a;
		out->continue_ms += b;
		out->resolve_ms += c;
		if (s->pass_deferred[pass] && s->d_qctl) {
			uint32_t ctl[QCTL_WORDS];
			HIP_TRY(hipMemcpy(ctl, s->d_qctl + (size_t)pass * QCTL_WORDS, sizeof ctl, hipMemcpyDeviceToHost));
			for (int q = 0; q < HXR_QUEUE_SHARDS; q++) out->deferred_rays += ctl[q * HXR_SHARD_STRIDE];
#ifdef HXR_CONT_CLOCKS
			{
				const uint32_t* dbg = ctl + 2 * HXR_QUEUE_SHARDS * HXR_SHARD_STRIDE;
				fprintf(stderr, "continuation tickets: %u, mean %.0f kclk, longest %u kclk; longer than 131k/262k/524k/1M/2M/4M/8M/16M clocks: %u %u %u %u %u %u %u %u\n", dbg[2],
				        dbg[2] ? (double)dbg[3] / dbg[2] : 0.0, dbg[1], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10], dbg[11]);
			}
#endif
			if (ctl[2 * HXR_QUEUE_SHARDS * HXR_SHARD_STRIDE]) { set_err("the ray queue overflowed (internal error: it is sized for the worst case)"); return HXR_E_HIP; }
		}
	}
	if (s->sched_frame && s->d_sched_ctl) {   // what prepare_schedule listed for this frame (the words are rewritten by the next frame only)
		uint32_t w[16];
		HIP_TRY(hipMemcpy(w, s->d_sched_ctl, sizeof w, hipMemcpyDeviceToHost));
		out->sched_front_items = w[2];
		out->sched_back_items = w[3];
		out->sched_part_pixels = w[4];
	}
	if (s->passes_timed > 1) {   // the detect kernel between the passes of an adaptive-AA frame
		float gap = 0;
		HIP_TRY(hipEventElapsedTime(&gap, s->ev_pass[0][3], s->ev_pass[1][0]));
		out->resolve_ms += gap;
	}
	return 0;
}

void hxr_scene_destroy(hxr_scene* s)
{
	if (!s) return;
	(void)hipSetDevice(s->device);
	(void)hipDeviceSynchronize();
	s->arena.release();
	if (s->d_frame) (void)hipFree(s->d_frame);
	if (s->d_scratch) (void)hipFree(s->d_scratch);
	if (s->d_aa_list) (void)hipFree(s->d_aa_list);
	if (s->d_extra) (void)hipFree(s->d_extra);
	if (s->d_sched_cost) (void)hipFree(s->d_sched_cost);
	if (s->d_sched_flags) (void)hipFree(s->d_sched_flags);
	if (s->d_sched_items) (void)hipFree(s->d_sched_items);
	if (s->d_sched_items_r) (void)hipFree(s->d_sched_items_r);
	if (s->parts_stream) (void)hipStreamDestroy(s->parts_stream);
	if (s->ev_parts_go) (void)hipEventDestroy(s->ev_parts_go);
	if (s->ev_parts_done) (void)hipEventDestroy(s->ev_parts_done);
	if (s->d_sched_side) (void)hipFree(s->d_sched_side);
	if (s->d_queue) (void)hipFree(s->d_queue);
	if (s->d_qctl) (void)hipFree(s->d_qctl);
	if (s->d_accum) (void)hipFree(s->d_accum);
	if (s->d_surface) (void)hipFree(s->d_surface);
	if (s->d_watch) (void)hipFree(s->d_watch);
	if (s->h_watch) (void)hipHostFree(s->h_watch);
	if (s->h_cancel) (void)hipHostFree(s->h_cancel);
	if (s->main_stream) (void)hipStreamDestroy(s->main_stream);
	if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
	if (s->ev_done) (void)hipEventDestroy(s->ev_done);
	for (int pass = 0; pass < 2; pass++)
		for (int k = 0; k < 4; k++)
			if (s->ev_pass[pass][k]) (void)hipEventDestroy(s->ev_pass[pass][k]);
	delete s;
}

// ------------------------------------------------------------------ one process, several GPUs
struct hxr_group {
	std::vector<hxr_scene*> scenes;        // one replica per entry of device_ids
	std::vector<int> devices;
	std::vector<hipStream_t> streams;
	std::vector<hipEvent_t> arrived;       // device d's tiles have landed on the first device
	std::vector<float*> d_tiles;           // per device: its tile-major shard buffer (entry 0 renders straight into the landing buffer)
	std::vector<size_t> tiles_bytes;
	float* d_landing = nullptr;   size_t landing_bytes = 0;   // first device: all shard buffers, rank-major
	float* d_frame = nullptr;     size_t frame_bytes = 0;     // first device: the assembled frame
	hipEvent_t ev_begin = nullptr, ev_end = nullptr;
	double last_ms = -1;
};

int hxr_group_size(const hxr_group* g) { return g ? (int)g->scenes.size() : 0; }

void hxr_group_destroy(hxr_group* g)
{
	if (!g) return;
	for (size_t d = 0; d < g->scenes.size(); d++) {
		(void)hipSetDevice(g->devices[d]);
		(void)hipDeviceSynchronize();
		if (d < g->d_tiles.size() && g->d_tiles[d]) (void)hipFree(g->d_tiles[d]);
		if (d < g->streams.size() && g->streams[d]) (void)hipStreamDestroy(g->streams[d]);
		if (d < g->arrived.size() && g->arrived[d]) (void)hipEventDestroy(g->arrived[d]);
		hxr_scene_destroy(g->scenes[d]);
	}
	if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
	if (g->d_landing) (void)hipFree(g->d_landing);
	if (g->d_frame) (void)hipFree(g->d_frame);
	if (g->ev_begin) (void)hipEventDestroy(g->ev_begin);
	if (g->ev_end) (void)hipEventDestroy(g->ev_end);
	delete g;
}

hxr_group* hxr_group_create(const hxr_scene_desc* desc, const hxr_scene_options* options, const int* device_ids, int num_devices)
{
	if (!device_ids || num_devices <= 0 || num_devices > 64) { set_err("bad device list"); return nullptr; }
	hxr_group* g = new hxr_group;
	// hxr_init() moves the process-wide selection (g_device, g_num_cu) from replica to replica; what the caller had chosen before the group
	// is put back on every way out, so that a plain hxr_scene_create() afterwards lands on the caller's device, not on the group's last one
	struct Restore {
		int device, num_cu;
		~Restore() { g_device = device; g_num_cu = num_cu; if (device >= 0) (void)hipSetDevice(device); }
	} restore{g_device, g_num_cu};
	for (int d = 0; d < num_devices; d++) {
		if (hxr_init(device_ids[d])) { hxr_group_destroy(g); return nullptr; }   // binds the calling thread to the device and records its size
		hxr_scene* s = hxr_scene_create_ex(desc, options);
		if (!s) { hxr_group_destroy(g); return nullptr; }
		g->scenes.push_back(s);
		g->devices.push_back(device_ids[d]);
		hipStream_t st = nullptr;
		hipEvent_t ev = nullptr;
		if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
			set_err("cannot create the stream of device %d", device_ids[d]);
			if (st) (void)hipStreamDestroy(st);
			hxr_group_destroy(g);
			return nullptr;
		}
		g->streams.push_back(st);
		g->arrived.push_back(ev);
		g->d_tiles.push_back(nullptr);
		g->tiles_bytes.push_back(0);
		if (d > 0 && device_ids[d] != device_ids[0]) {
			// Device d pushes its tile buffer into device 0's landing buffer (hipMemcpyPeerAsync on d's stream).  The current device here is
			// device_ids[d] (hxr_init above), so this enables d -> 0 access, the direction of that copy.  Without peer access the runtime
			// would stage the copy through the host -- a path this module has never run (one-GPU test boxes list device 0 several times) --
			// so a platform that refuses it is reported, not silently used.
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, device_ids[d], device_ids[0]) != hipSuccess || !can) {
				(void)hipGetLastError();
				set_err("device %d cannot access device %d peer-to-peer: hxr_group needs direct (xGMI / PCIe P2P) copies into the first device", device_ids[d], device_ids[0]);
				hxr_group_destroy(g);
				return nullptr;
			}
			hipError_t pe = hipDeviceEnablePeerAccess(device_ids[0], 0);
			if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
				set_err("hipDeviceEnablePeerAccess(%d -> %d) failed: %s", device_ids[d], device_ids[0], hipGetErrorString(pe));
				(void)hipGetLastError();
				hxr_group_destroy(g);
				return nullptr;
			}
			(void)hipGetLastError();
		}
	}
	(void)hipSetDevice(device_ids[0]);
	if (hipEventCreate(&g->ev_begin) != hipSuccess || hipEventCreate(&g->ev_end) != hipSuccess) { set_err("hipEventCreate failed"); hxr_group_destroy(g); return nullptr; }
	return g;
}

int hxr_group_update_camera(hxr_group* g, const hxr_camera* cam)
{
	if (!g || !cam) { set_err("null argument"); return HXR_E_INVALID; }
	for (hxr_scene* s : g->scenes) {
		int rc = hxr_scene_update_camera(s, cam);
		if (rc) return rc;
	}
	return 0;
}

int hxr_group_render(hxr_group* g, const hxr_frame_params* fp, float* vfb, size_t row_stride_bytes, const volatile int* cancel, hxr_progress_cb cb, void* user)
{
	if (!g || !fp || !vfb || g->scenes.empty()) { set_err("null argument"); return HXR_E_INVALID; }
	const int n = (int)g->scenes.size();
	if (fp->shard_count > 1 || fp->layout != HXR_LAYOUT_FRAME) { set_err("hxr_group_render shards the frame itself: pass an unsharded FRAME-layout request"); return HXR_E_INVALID; }
	if (cancel && *cancel) return HXR_CANCELLED;
	hxr_frame_params base = *fp;
	if (base.tile_size <= 0) base.tile_size = 8;
	base.shard_count = n;
	base.layout = HXR_LAYOUT_TILES;
	Plan p0;
	base.shard_rank = 0;
	int rc = resolve(g->scenes[0], &base, p0);
	if (rc) return rc;
	const size_t per_rank = (size_t)p0.ntiles * p0.tile * p0.tile * 12;   // rank 0 owns the most tiles: every rank's slot has its size
	const size_t dstride = (size_t)p0.W * 12;
	HIP_TRY(hipSetDevice(g->devices[0]));
	rc = ensure((void**)&g->d_landing, &g->landing_bytes, per_rank * (size_t)n);
	if (!rc) rc = ensure((void**)&g->d_frame, &g->frame_bytes, dstride * p0.H);
	if (rc) return rc;
	// every device renders its shard; devices other than the first then push their buffer into their slot of the landing buffer
	for (int d = 0; d < n; d++) {
		hxr_scene* s = g->scenes[d];
		HIP_TRY(hipSetDevice(g->devices[d]));
		float* out = g->d_landing;
		if (d > 0) {
			rc = ensure((void**)&g->d_tiles[d], &g->tiles_bytes[d], per_rank);
			if (rc) return rc;
			out = g->d_tiles[d];
		}
		if (cancel) {   // the cancel word of this replica (see hxr_render)
			rc = ensure_cancel_word(s);
			if (rc) return rc;
			s->watch_cancel = s->d_cancel;
		}
		if (d == 0) HIP_TRY(hipEventRecord(g->ev_begin, g->streams[0]));
		hxr_frame_params mine = base;
		mine.shard_rank = d;
		rc = render_on_stream(s, &mine, out, 0, g->streams[d]);
		s->watch_cancel = nullptr;
		if (rc) return rc;
		if (d > 0) {
			HIP_TRY(hipMemcpyPeerAsync((char*)g->d_landing + per_rank * (size_t)d, g->devices[0], g->d_tiles[d], g->devices[d], per_rank, g->streams[d]));
			HIP_TRY(hipEventRecord(g->arrived[d], g->streams[d]));
		}
	}
	HIP_TRY(hipSetDevice(g->devices[0]));
	for (int d = 1; d < n; d++) HIP_TRY(hipStreamWaitEvent(g->streams[0], g->arrived[d], 0));
	rc = hxr_assemble_frame(g->scenes[0], &base, g->d_landing, per_rank, g->d_frame, dstride, g->streams[0]);
	if (rc) return rc;
	HIP_TRY(hipEventRecord(g->ev_end, g->streams[0]));
	bool signalled = false;
	for (;;) {
		hipError_t q = hipEventQuery(g->ev_end);
		if (q == hipSuccess) break;
		if (q != hipErrorNotReady) { set_err("hipEventQuery failed: %s", hipGetErrorString(q)); return HXR_E_HIP; }
		if (cancel && *cancel && !signalled) {
			for (int d = 0; d < n; d++) __atomic_store_n(g->scenes[d]->h_cancel, 1u, __ATOMIC_SEQ_CST);
			signalled = true;
		}
		struct timespec nap = {0, 100 * 1000};
		nanosleep(&nap, nullptr);
	}
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, g->ev_begin, g->ev_end));
	g->last_ms = ms;
	if (signalled || (cancel && *cancel)) return HXR_CANCELLED;
	HIP_TRY(hipMemcpy2D(vfb, row_stride_bytes, g->d_frame, dstride, dstride, p0.H, hipMemcpyDeviceToHost));
	if (cb)
		for (int t = 0; t < p0.tiles_x * p0.tiles_y; t++) {
			int tx_, ty_;
			hxr_tile_position(t, p0.tiles_x, n, &tx_, &ty_);
			int x0 = tx_ * p0.tile, y0 = ty_ * p0.tile;
			cb(user, x0, y0, std::min(x0 + p0.tile, p0.W), std::min(y0 + p0.tile, p0.H));
		}
	return 0;
}

double hxr_group_last_ms(const hxr_group* g, double* per_device_kernel_ms)
{
	if (!g) return -1.0;
	if (per_device_kernel_ms)
		for (size_t d = 0; d < g->scenes.size(); d++) {
			(void)hipSetDevice(g->devices[d]);
			per_device_kernel_ms[d] = hxr_last_kernel_ms(g->scenes[d]);
		}
	if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
	return g->last_ms;
}

const char* hxr_last_error(void) { return g_err; }

}  // extern "C"
