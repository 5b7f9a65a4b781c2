// Device-side scene layout of the MI355X render module.
//
// hxr_scene_create() re-lays the flattened descriptor (include/hexray_hip.h) out for the kernels:
//   * kd nodes become 16-byte records {split, first child / first triangle, axis, count}; the two children of
//     a node are adjacent and 32-byte aligned, so entering a node costs ONE s_load_dwordx8 for both.  A
//     node's box is an exact copy of root-box bounds and split positions (BBox::split only copies values,
//     src/bbox.h:210-216): on the way down the walk derives it from the parent's, and a parallel array of
//     48-byte boxes serves the visits that resume from the stack, which frees the traversal from carrying /
//     restoring the reference's per-level BBox copies (src/mesh.cpp:233-243);
//   * leaf triangle lists are expanded into leaf-ordered, 96-byte "hot" records (A, AB, AC, ABxAC --
//     exactly the operands of Mesh::intersectTriangle's rejection tests, src/mesh.cpp:178-195), so a
//     leaf is one contiguous stream instead of index -> 184-byte Triangle gathers;
//   * the node and hot-triangle arrays of ALL meshes are concatenated and handed to the kernel as
//     `const __restrict__` arguments: traversal control is wave-uniform, so these records are fetched with
//     scalar loads into SGPRs, once per wave instead of once per lane;
//   * per-hit attributes (uv / normal / dNdx / dNdy interpolation, src/mesh.cpp:196-218) are read once,
//     for the accepted hit of the closest node only, from the "cold" tables;
//   * small tables (nodes, shaders, layers, textures, lights) are copied verbatim: their indices are
//     wave-uniform in the hot loops, so they are fetched through the scalar cache as well.
#pragma once
#include <stdint.h>
#include "hexray_hip.h"

#define HXR_KD_LEAF 3

// Tile dealing (include/hexray_hip.h, "sharding").  Tiles are numbered row-major, number t belongs to shard t % N and is that shard's
// local tile t / N.  With N > 1 the numbers of tile row ty are rotated by HXR_TILE_SKEW * ty columns (number ty * tiles_x + v sits at
// column (v - HXR_TILE_SKEW * ty) mod tiles_x): 1920 / 8 = 240 tile columns are a multiple of 8, so without the rotation a shard would
// own the same columns in every row and a vertical image feature -- the crease of the heart in data/meshes.hexray -- would land on
// the same two or three GPUs all the way down (measured on the 8-way split of the headline frame: slowest rank 1.62 ms against a mean of 1.44).
#define HXR_TILE_SKEW 3
#if defined(__HIPCC__) || defined(__CUDACC__)
#define HXR_HD __host__ __device__
#else
#define HXR_HD
#endif
HXR_HD inline void hxr_tile_position(int t, int tiles_x, int shard_count, int* tx, int* ty)
{
	*ty = t / tiles_x;
	const int v = t - *ty * tiles_x;
	*tx = shard_count > 1 ? (v + tiles_x - (HXR_TILE_SKEW * *ty) % tiles_x) % tiles_x : v;
}
HXR_HD inline int hxr_tile_number(int tx, int ty, int tiles_x, int shard_count)
{
	return ty * tiles_x + (shard_count > 1 ? (tx + HXR_TILE_SKEW * ty) % tiles_x : tx);
}

struct HxrKdSlim {         // 16 B
	double split;          // binary: splitPos
	int32_t first;         // binary: index of the left child (right = first + 1), local to the mesh;
	                       // leaf  : index of its first hot triangle, local to the mesh
	int32_t axis_count;    // bits 0..1: split axis, or HXR_KD_LEAF; bits 2..: number of triangles of a leaf
};

struct HxrKdPair { HxrKdSlim kid[2]; };   // the two children of a binary node: 32 B, 32-B aligned

struct HxrKdBox {          // 48 B: the box of the node with the same index
	double lo[3], hi[3];
};

struct HxrTriHot {         // 96 B
	double A[3], AB[3], AC[3], N[3];
};

struct HxrMeshDev {
	double bbox_min[3], bbox_max[3];
	int32_t faceted, backface_culling, has_kd, num_triangles;
	int32_t kd_first;              // this mesh's nodes in the concatenated kd arrays (node 0 of the range = root); odd, so that
	                               // child pairs (local indices 1-2, 3-4, ...) start at even global indices
	int32_t hot_first;             // this mesh's leaf-ordered (has_kd) or triangle-ordered (no kd) hot records / triangle ids
	int32_t boxes_nested;          // every kd node's box lies inside its parent's (split positions inside their node's box): an origin
	                               // outside the inflated root box is then outside every node's inflated box (mesh_query)
	int32_t pad_;
	const hxr_triangle* tris;      // cold records of this mesh
	const double* vertices;        // 3 doubles each (element 0 = sentinel)
	const double* normals;
	const double* uvs;
};

struct HxrHeightfieldDev {
	int32_t W, H, max_k, use_optimization;
	double bbox_min[3], bbox_max[3];
	const float* heights;
	const float* max_h;
	const float* high_map;   // 16 floats per cell
	const double* normals;   // 3 doubles per texel
	const float* step_rec;   // 4 floats per cell: {maxH, highMap levels 1, 2, 3} -- what a marching step reads, in one 16-byte load from a
	                         // table a quarter of the pyramid's size (4.2 MB for 512 x 512: at home in an XCD's L2); levels >= 4 stay in high_map
};

struct HxrBitmapDev {
	int32_t width, height;
	const float* texels;     // 3 floats per texel
};

struct HxrNodeDev {
	hxr_transform T;
	int32_t geom_type;       // hxr_geom_type of its geometry
	int32_t geom;            // geometry table index (CSG operands refer to it)
	int32_t ref;             // mesh / heightfield index
	int32_t shader, bump;
	int32_t cullable;        // 1: cull_lo / cull_hi hold the culling sphere of a bounded geometry; 2: a plane node's world-space side test (c, k, |c|)
	int32_t needs_uv;        // the node's bump map or a texture in its shader tree reads IntersectionInfo::u / v
	int32_t identity;        // T.m, T.inv_m and T.tinv_m are exactly the identity (a translate-only node): the products are skipped
	double p[6];             // analytic parameters of the geometry
	// World-space sphere that encloses everything a ray must reach to get ANY answer other than "miss" from this
	// node: it contains the transformed, 1e-6-inflated (BBox::inside slack) bounding box of the geometry, with a
	// relative safety margin far above FP64 rounding.  A ray that misses it is a miss for the node without
	// transforming the ray into object space.  cull_lo = centre, cull_hi[0] = squared radius.
	double cull_lo[3], cull_hi[3];
};

struct HxrSceneDev {
	hxr_settings settings;
	hxr_camera camera;
	hxr_environment environment;
	int32_t num_nodes, num_lights, num_shaders, num_textures;
	const HxrNodeDev* nodes;
	const hxr_geometry* geometries;
	const HxrMeshDev* meshes;
	const HxrHeightfieldDev* heightfields;
	const HxrBitmapDev* bitmaps;
	const hxr_texture* textures;
	const hxr_shader* shaders;
	const hxr_layer* layers;
	const hxr_light* lights;
	const int32_t* env_sides;   // = &environment.side_bitmap[0] of the copy in HBM (indexed at run time; keeps this header in registers)
	const HxrKdSlim* kd;        // all meshes' kd nodes
	const HxrKdBox* kd_box;     // their boxes
	unsigned long long* dev_stats;   // developer builds (-DHXR_LEAF_STATS): six counters in hxr_counters.reserved_; null otherwise
	const HxrTriHot* hot;       // all meshes' hot triangle records
	const int32_t* hot_ids;     // triangle number (local to its mesh) of each hot record
};

// A secondary ray that the render kernel hands to the continuation pass instead of tracing it in place (HXR_F_DEFER variants):
// everything raytrace() needs to go on -- the ray, the product of blend / reflectance factors on the way to it, and the identity
// of the sample for the random numbers.  80 bytes, written by one lane with five 16-byte stores.
struct alignas(16) HxrRayRec {
	double o[3], d[3];
	float weight[3];
	uint32_t key;            // ray key of the counter-based generator (hxr_rng.hip.h)
	uint32_t pixel;          // y * W + x
	uint32_t sample_depth;   // bits 0..15 sample number, 16..23 Ray::depth, 24..31 Ray::flags
	uint32_t pad_[2];
};
#define HXR_QUEUE_SHARDS 8   // sub-queues = dispenser ranges (HXR_SHARDS in hxr_kernels.hip.h): one append counter each

// One frame's launch parameters (kernel argument, wave-uniform).
struct HxrLaunch {
	int32_t W, H;              // frame size
	int32_t spp;               // samples per pixel of this pass
	int32_t mode;              // 0: one sample at the pixel corner (pass 1); 1: Monte-Carlo; 2: AA taps (pass 3)
	uint64_t seed;
	int32_t tile_size, tiles_x, tiles_y;
	int32_t shard_rank, shard_count, num_shard_tiles;
	int32_t num_extra_pixels;  // pass 1 of a sharded AA frame: halo pixels listed in `extra_pixels`
	int32_t num_aa_pixels;     // pass 3: entries of `aa_list`
	const int32_t* extra_pixels;
	const int32_t* aa_list;
	const int32_t* aa_count;   // device counter written by the detect kernel (pass 3 reads it)
	float* out;                // destination
	size_t out_stride;         // bytes per row (FRAME layout) -- ignored for TILES
	int32_t out_tiles;         // 1: tile-major layout
	uint32_t* work_counter;    // dynamic chunk dispenser
	unsigned long long* counters;   // hxr_counters as 64-bit words, or nullptr
	uint32_t num_chunks;
	int32_t pixels_per_chunk;  // pixels one 64-lane chunk covers
	// Feedback-directed packet schedule (hxr_api.hip, "adaptive schedule"): the shader-clock duration of every chunk of THIS
	// frame goes to sched_cost; chunks that were expensive in the PREVIOUS frame of the same shape are listed pixel by
	// pixel in sched_items and handed out first, one pixel per packet.  All null / zero when the feature is off.
	uint32_t* sched_cost;            // [num_chunks]
	const uint8_t* sched_flags;      // [num_chunks] 1 = listed in sched_items: the ordinary dispenser skips it
	const uint32_t* sched_items;     // pixel numbers: the heaviest tier from the front, the second tier from the back
	const uint32_t* sched_counts;    // [0] items in the first tier, [1] items in the second
	uint32_t* sched_ticket;          // dispenser of the listed items
	uint32_t sched_capacity;         // entries of sched_items
	// third level of splitting: HXR_PART_LANES samples of one pixel per packet.  The parts leave their sample colours in
	// sched_side; the part that finishes last (sched_part_done) adds them up in sample order and writes the pixel.
	float* sched_side;               // [part slot][sample][3]
	uint32_t* sched_part_done;       // [part slot] parts finished
	const uint32_t* sched_part_pixels;   // [part slot] pixel number
	int32_t sched_part_groups;       // parts per pixel (0: level disabled)
	// Deferred secondary rays (HXR_F_DEFER variants; hxr_api.hip, plan_defer).  The render kernel appends the Reflection / Refraction
	// rays of its hits to q_records instead of tracing them; hxr_continue_kernel traces them to the end, 64 unrelated rays per wave,
	// and adds what they contribute to their pixels into q_accum as 64-bit fixed point (integer addition commutes: the frame does
	// not depend on the order in which waves finish); hxr_resolve_kernel folds q_accum into the frame and clears it.
	HxrRayRec* q_records;            // HXR_QUEUE_SHARDS sub-queues of q_capacity records each
	uint32_t* q_ctl;                 // [shard * 32] records appended, [(8 + shard) * 32] records handed out, [16 * 32] overflow flag
	uint32_t q_capacity;             // records per sub-queue: the packets of its dispenser range cannot emit more (0: nothing is deferred)
	long long* q_accum;              // [W * H][3] colour sums in units of 2^-32
	int32_t lane_stack_entries;      // per-lane-walk continuation kernels: rows of the per-lane stack (deepest tree + 2)
	int32_t cont_ticket;             // continuation kernels: rays per ticket (0 = 64, a full wave)
	// hxr_render() with a cancel flag / a progress callback (null otherwise: the asynchronous entry pays nothing for them)
	const uint32_t* cancel;          // device word the host raises while the frame runs: waves stop taking packets (checkForUserExit, src/sdl.cpp:253)
	uint32_t* tile_done;             // [shard tile] pixels written so far: a tile whose count reaches its pixel count is finished (displayVFBRect, src/main.cpp:312)
	// foveated rendering (src/main.cpp:284-311): mode 0 fills 8x8 blocks whose centre is farther than fov_radius from (vip_x, vip_y)
	// with the mean of five rays; the detect kernel only flags pixels inside the radius.  fov_radius 0 = off.
	int32_t fov_radius, vip_x, vip_y;
	// 1: packets handed out from the FRONT of the schedule (split pixels, sample parts, chunks long against the frame) raise their wave's issue priority
	// (s_setprio) for their duration.  Set for launches that are a few packets deep per resident wave (shards of a multi-GPU frame): there one long chain
	// is what the launch waits for, and the arbiter lets it run at nearly the speed of a wave alone on its SIMD (8-way slowest rank 1.65 -> 1.51 ms); a whole
	// frame loses 2.5 % of throughput by it and leaves it off.
	int32_t front_priority;
	int32_t no_replication;    // 1: HXR_F_REPL variants trace every packet one sample per lane (A/B aid: HXR_NO_REPLICATION)
	// A second list of schedule items, taken once the first (sched_items) is through: the sample parts and single pixels that the replicating blocks
	// (HXR_F_REPL instance, launched beside this kernel) take as THEIR first list -- whoever gets there first renders the item.  Null: none.
	const uint32_t* sched2_items;
	const uint32_t* sched2_count;    // device word: number of items
	uint32_t* sched2_ticket;
	// ... taken BEFORE the ordinary packets when it holds more items than this (a 1-spp pass lists thousands of single pixels: too many for the few
	// replicating blocks), otherwise only once the ordinary packets are through (a shard of a multi-GPU frame lists a few hundred parts: the replicating
	// blocks are through with them long before, and every one of them is traced replicated)
	uint32_t sched2_early_above;
};
