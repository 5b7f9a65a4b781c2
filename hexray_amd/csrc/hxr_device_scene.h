// Device-side scene layout of the MI355X render module.
//
// hxr_scene_create() re-lays the flattened descriptor (include/hexray_hip.h) out for the kernels:
//   * kd nodes become 32-byte records that carry their parent link and the box bound they override,
//     so traversal needs NO stack: the walk goes down by tightening one bound of the box it carries in
//     registers and comes back up by restoring it from the record (reference recursion:
//     src/mesh.cpp:221-245, which copies a whole BBox per level);
//   * leaf triangle lists are expanded into leaf-ordered, 96-byte "hot" records (A, AB, AC, ABxAC --
//     exactly the operands of Mesh::intersectTriangle's rejection tests, src/mesh.cpp:178-195), so a
//     leaf is one contiguous stream instead of index -> 184-byte Triangle gathers;
//   * per-hit attributes (uv / normal / dNdx / dNdy interpolation, src/mesh.cpp:196-218) are read once,
//     for the accepted hit of the closest node only, from the "cold" tables;
//   * small tables (nodes, shaders, layers, textures, lights) are copied verbatim: their indices are
//     wave-uniform in the hot loops, so they are fetched through the scalar cache.
#pragma once
#include <stdint.h>
#include "hexray_hip.h"

#define HXR_KD_LEAF 3

struct HxrKdRec {          // 32 B, 16-B aligned: two dwordx4 loads
	double split;          // binary: splitPos
	double saved;          // the parent's value of the bound THIS node's box overrides (restored on the way up)
	int32_t first;         // binary: index of the left child (right = first + 1); leaf: first hot-triangle index
	int32_t parent;        // -1 for the root
	int32_t axis;          // 0..2, or HXR_KD_LEAF
	int32_t count;         // leaf: number of triangles
};

struct HxrTriHot {         // 96 B
	double A[3], AB[3], AC[3], N[3];
};

struct HxrMeshDev {
	double bbox_min[3], bbox_max[3];
	int32_t faceted, backface_culling, has_kd, num_triangles;
	const HxrKdRec* kd;            // node 0 = root
	const HxrTriHot* leaf_tris;    // leaf-ordered (has_kd) or triangle-ordered (no kd) hot records
	const int32_t* leaf_tri_ids;   // triangle number of each hot record
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
	int32_t pad_[3];
	double p[6];             // analytic parameters of the geometry
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
};

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
};
