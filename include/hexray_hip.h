/*
 * hexray_hip.h -- C ABI of the MI355X render module (libhexray_hip.so).
 *
 * This is the drop-in boundary for heX-Ray's per-pixel / per-sample raytrace loop.  The reference
 * has no FFI: its seam is `bool render(bool displayProgress)` (reference src/main.cpp:416-426), which
 * reads the global `Scene scene` (src/scene.h:271-294) and writes the global
 * `Color vfb[VFB_MAX_SIZE][VFB_MAX_SIZE]` (src/main.cpp:50).  A host that keeps the reference's loader,
 * scene graph, camera and bitmap output flattens its Scene once into `hxr_scene_desc` (plain arrays,
 * no pointers inside records), hands it to `hxr_scene_create`, and calls `hxr_render` where the
 * reference calls `renderWithoutMonteCarlo` / `renderWithMonteCarlo` (src/main.cpp:253-384).
 * INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch / HIP types in any signature (a stream is `void*`).
 *   - every record mirrors a reference class; the citation next to it names the fields it carries.
 *   - vectors are 3 doubles, colours 3 floats (reference src/vector.h:30-34, src/color.h:62-68).
 *   - matrices are row-major 3x3 used with ROW vectors, v' = v * M (reference src/matrix.h:53-60).
 *   - indices are 0-based into the tables of the same descriptor; -1 means "none".
 *   - all functions return 0 on success, a negative HXR_E_* code on failure; nothing throws.
 */
#ifndef HEXRAY_HIP_H
#define HEXRAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HXR_ABI_VERSION 3u   /* 2: hxr_heightfield grew raw_heights / blur_radius / first_blur_kernel; 3: hxr_bitmap grew prep / gamma / first_gamma_table */

/* ------------------------------------------------------------------ error codes */
enum {
	HXR_OK = 0,
	HXR_CANCELLED = 1,        /* hxr_render only: the cancel flag was raised (reference render() returns false) */
	HXR_E_INVALID = -1,       /* bad argument / malformed descriptor */
	HXR_E_NO_DEVICE = -2,     /* no usable gfx950 device, or hxr_init not called */
	HXR_E_HIP = -3,           /* a HIP runtime call failed; see hxr_last_error() */
	HXR_E_UNSUPPORTED = -4,   /* scene uses a feature this build does not render */
	HXR_E_NOMEM = -5
};

/* ------------------------------------------------------------------ element tables */

/* Transform (reference src/matrix.h:65-99): offset, m, invM, transposedInverse. */
typedef struct hxr_transform {
	double offset[3];
	double m[9];
	double inv_m[9];
	double tinv_m[9];
} hxr_transform;

enum hxr_geom_type {
	HXR_GEOM_PLANE = 0,       /* src/geometry.h:57-67   p = {y, limit} */
	HXR_GEOM_SPHERE = 1,      /* src/geometry.h:69-81   p = {Ox, Oy, Oz, R, uvscaling} */
	HXR_GEOM_CUBE = 2,        /* src/geometry.h:83-107  p = {Ox, Oy, Oz, halfSide} */
	HXR_GEOM_CSG_UNION = 3,   /* src/geometry.h:109-136 left/right = geometry indices */
	HXR_GEOM_CSG_INTER = 4,
	HXR_GEOM_CSG_DIFF = 5,
	HXR_GEOM_MESH = 6,        /* src/mesh.h:75-119      ref = index into meshes[] */
	HXR_GEOM_HEIGHTFIELD = 7  /* src/heightfield.h:30-52 ref = index into heightfields[] */
};

typedef struct hxr_geometry {
	int32_t type;
	int32_t left, right;
	int32_t ref;
	double p[6];
} hxr_geometry;

/* Triangle (reference src/bbox.h:36-43), same fields, indices local to the owning mesh. */
typedef struct hxr_triangle {
	int32_t v[3], n[3], t[3];
	int32_t pad_;
	double gnormal[3];
	double AB[3], AC[3], ABcrossAC[3];
	double dNdx[3], dNdy[3];
} hxr_triangle;

/* KDTreeNode (reference src/mesh.h:32-60) flattened: children are stored as an adjacent pair, AFTER their parent (first > the node's own
 * index: what a depth-first builder that appends a pair when it splits a node produces; hxr_scene_create refuses anything else). */
typedef struct hxr_kd_node {
	int32_t axis;      /* 0,1,2 = split axis of a binary node; 3 = leaf (AXIS_NONE) */
	int32_t first;     /* binary: index of the LEFT child (right = first+1), local to the mesh;
	                      leaf  : index of its first entry in the mesh's leaf-index list */
	int32_t count;     /* leaf: number of triangle indices; binary: 0 */
	int32_t parent;    /* local index of the parent node, -1 for the root */
	double split;      /* splitPos (binary nodes) */
} hxr_kd_node;

/* Mesh (reference src/mesh.h:75-119).  Offsets are ELEMENT offsets into the descriptor's pools. */
typedef struct hxr_mesh {
	int32_t faceted, backface_culling, has_kd, pad_;
	double bbox_min[3], bbox_max[3];
	/* vertex / normal / uv arrays live in f64_pool as 3-double vectors; first_* counts VECTORS:
	   component c of element e is f64_pool[(first_vertex + e) * 3 + c]; element 0 is the OBJ sentinel */
	int64_t first_vertex, num_vertices;
	int64_t first_normal, num_normals;
	int64_t first_uv, num_uvs;               /* (u, v, 0) */
	int64_t first_triangle, num_triangles;   /* triangles[] */
	int64_t first_kd_node, num_kd_nodes;     /* kd_nodes[]; node 0 of the range is the root */
	int64_t first_leaf_index, num_leaf_indices; /* leaf_indices[] (triangle numbers local to the mesh) */
} hxr_mesh;

/* Heightfield (reference src/heightfield.h:30-52). */
typedef struct hxr_heightfield {
	int32_t W, H, max_k, use_optimization;
	double bbox_min[3], bbox_max[3];
	int64_t first_height;    /* f32_pool, W*H            heights[y*W+x] */
	int64_t first_maxh;      /* f32_pool, W*H            maxH[y*W+x];  -1 together with first_highmap = not provided: the module    */
	int64_t first_highmap;   /* f32_pool, W*H*16         highMap[y*W+x].h[k]   builds both on the GPU from the heights (max_k too) */
	int64_t first_normal;    /* f64_pool, counted in 3-double VECTORS like hxr_mesh: normals[y*W+x]; -1 with raw_heights */
	/* raw_heights = 1: the whole of Heightfield::fillProperties after the image load (src/heightfield.cpp:173-261) is left to the module,
	   which runs it on the GPU at hxr_scene_create: `first_height` holds the UNBLURRED intensities, the Gaussian table of the blur is
	   handed over (its R*R entries come from the host's exp()), and heights, bounding box (bbox_min[1] / bbox_max[1] are ignored),
	   normals and maxima tables are derived there (first_maxh = first_highmap = first_normal = -1).  For the HIP module only. */
	int32_t raw_heights;
	int32_t blur_radius;       /* R of the table: taps dx, dy = -(R-1) .. R-1; 0 = no blur */
	int64_t first_blur_kernel; /* f32_pool, R*R: g[|dy|*R + |dx|] */
} hxr_heightfield;

/* Bitmap (reference src/bitmap.h:32-63): float RGB texels, row-major, 3 floats per texel. */
enum hxr_bitmap_prep {
	HXR_BITMAP_GAMMA = 1,          /* Bitmap::decompressGamma (src/bitmap.cpp:321-338) is still to be applied */
	HXR_BITMAP_DIFFERENTIATE = 2   /* Bitmap::differentiate (src/bitmap.cpp:304-319) is still to be applied (after the gamma decode, if both) */
};

typedef struct hxr_bitmap {
	int32_t width, height;
	int64_t first_texel;     /* f32_pool index of texel (0,0).r */
	/* prep != 0: the texels are the FILE's, and the loader's passes over them named by `prep` are left to the module, which runs them
	   as kernels at hxr_scene_create.  For the HIP module only (the oracle refuses such a descriptor).
	   Gamma decode: a texel c > 0 becomes decompress_sRGB(c) when |gamma - 2.2| < 1e-6, pow(c, gamma) otherwise.  An 8-bit source has 256
	   distinct texel values q / 255.0f: the host decodes those once (its own powf) into `first_gamma_table` (f32_pool, 256 entries:
	   entry q = the decoded q / 255.0f) and the kernel looks every texel up -- bit-identical to the host's pass over the image.
	   first_gamma_table = -1 (a float source, e.g. EXR): the kernel evaluates the curve itself (device powf: equal to float rounding). */
	int32_t prep;
	float gamma;
	int64_t first_gamma_table;
} hxr_bitmap;

enum hxr_texture_type {
	HXR_TEX_CHECKER = 0,     /* src/shading.h:56-67   color1, color2, scaling */
	HXR_TEX_BITMAP = 1,      /* src/shading.h:69-85   bitmap, scaling */
	HXR_TEX_FRESNEL = 2,     /* src/shading.h:179-187 ior */
	HXR_TEX_BUMPTEX = 3,     /* src/shading.h:189-204 bitmap (already differentiated), strength, scaling */
	HXR_TEX_BUMPS = 4        /* src/shading.h:217-227 strength (float in the reference) */
};

typedef struct hxr_texture {
	int32_t type;
	int32_t bitmap;
	float color1[3], color2[3];
	double scaling;
	double strength;
	double ior;
} hxr_texture;

enum hxr_shader_type {
	HXR_SHADER_LAMBERT = 0,     /* src/shading.h:95-109  color, texture */
	HXR_SHADER_PHONG = 1,       /* src/shading.h:111-123 + specular, exponent */
	HXR_SHADER_REFLECTION = 2,  /* src/shading.h:125-145 color = reflColor, glossiness, num_samples */
	HXR_SHADER_REFRACTION = 3,  /* src/shading.h:147-164 color = refrColor, ior */
	HXR_SHADER_LAYERED = 4,     /* src/shading.h:166-177 first_layer, num_layers */
	HXR_SHADER_CONST = 5        /* src/shading.h:206-214 color */
};

typedef struct hxr_shader {
	int32_t type;
	int32_t texture;
	float color[3];
	float specular[3];
	float exponent;
	float glossiness;
	int32_t num_samples;
	int32_t first_layer, num_layers;
	int32_t pad_;
	double ior;
} hxr_shader;

/* Layered::Layer (reference src/shading.h:167-171). */
typedef struct hxr_layer {
	int32_t shader;
	int32_t texture;
	float blend[3];
} hxr_layer;

enum hxr_light_type {
	HXR_LIGHT_POINT = 0,     /* src/lights.h:79-96 */
	HXR_LIGHT_RECT = 1       /* src/lights.h:98-121; area / scale_factor as left by beginFrame (src/lights.cpp:77-89) */
};

typedef struct hxr_light {
	int32_t type;
	int32_t x_subd, y_subd;
	float color[3];
	float power;
	float scale_factor;
	double pos[3];
	double area;
	hxr_transform T;
} hxr_light;

/* Node (reference src/node.h:31-47). Only nodes WITH a shader are listed (src/scene.cpp:558-565). */
typedef struct hxr_node {
	int32_t geometry, shader, bump;
	int32_t pad_;
	hxr_transform T;
} hxr_node;

/* Camera after beginFrame (reference src/camera.cpp:30-63): the frustum corner basis. */
typedef struct hxr_camera {
	double pos[3];
	double top_left[3], top_right[3], bottom_left[3];
	double up[3], right[3], front[3];
	double aperture_size;       /* 2.5 / fNumber */
	double focal_plane_dist;
	double stereo_separation;
	int32_t dof, num_samples, auto_focus, pad_;
} hxr_camera;

/* GlobalSettings (reference src/scene.h:244-269), the fields the path reads. */
typedef struct hxr_settings {
	float ambient[3];
	float background[3];
	int32_t max_trace_depth;
	int32_t want_aa;
	int32_t gi;
	int32_t num_paths;
	int32_t frame_width, frame_height;
} hxr_settings;

/* CubemapEnvironment (reference src/environment.h:49-73): sides in CubeOrder NEGX..POSZ. */
typedef struct hxr_environment {
	int32_t present;         /* scene.environment != NULL */
	int32_t loaded;          /* Environment::loaded */
	int32_t side_bitmap[6];
} hxr_environment;

typedef struct hxr_scene_desc {
	uint32_t abi_version;    /* HXR_ABI_VERSION */
	uint32_t reserved_;
	hxr_settings settings;
	hxr_camera camera;
	hxr_environment environment;

	const hxr_geometry* geometries;        int64_t num_geometries;
	const hxr_mesh* meshes;                int64_t num_meshes;
	const hxr_heightfield* heightfields;   int64_t num_heightfields;
	const hxr_triangle* triangles;         int64_t num_triangles;
	const hxr_kd_node* kd_nodes;           int64_t num_kd_nodes;
	const int32_t* leaf_indices;           int64_t num_leaf_indices;
	const hxr_bitmap* bitmaps;             int64_t num_bitmaps;
	const hxr_texture* textures;           int64_t num_textures;
	const hxr_shader* shaders;             int64_t num_shaders;
	const hxr_layer* layers;               int64_t num_layers;
	const hxr_light* lights;               int64_t num_lights;
	const hxr_node* nodes;                 int64_t num_nodes;
	const double* f64_pool;                int64_t f64_pool_size;   /* in doubles */
	const float* f32_pool;                 int64_t f32_pool_size;   /* in floats */
} hxr_scene_desc;

/* ------------------------------------------------------------------ frame parameters */

enum hxr_render_mode {
	HXR_MODE_AUTO = 0,        /* what render() picks (src/main.cpp:416-426): MC if dof/gi, else 1 spp (+AA if wantAA) */
	HXR_MODE_SINGLE = 1,      /* pass 1 of renderWithoutMonteCarlo only (src/main.cpp:273-314) */
	HXR_MODE_ADAPTIVE_AA = 2, /* all three passes of renderWithoutMonteCarlo (src/main.cpp:253-345) */
	HXR_MODE_MONTE_CARLO = 3, /* renderWithMonteCarlo (src/main.cpp:347-384) with `spp` rays per pixel */
	HXR_MODE_PREPASS = 4      /* coarseRender (src/main.cpp:387-414), the preview render() shows first when displayProgress and
	                             prepassSamples > 0: every 24x24 block is filled with the mean of `spp` (= prepassSamples) rays through
	                             random points of the block.  The reference paints it on the window only; here it fills the frame. */
};

enum hxr_output_layout {
	HXR_LAYOUT_FRAME = 0,     /* out[y*row_stride + x*12]: the reference's vfb[y][x] (row stride 23040 B there) */
	HXR_LAYOUT_TILES = 1      /* this rank's tiles back to back, each tile_size*tile_size*3 floats, row-major inside */
};

typedef struct hxr_frame_params {
	int32_t width, height;    /* 0 = settings.frame_width / frame_height; at most 1920 (VFB_MAX_SIZE, src/constants.h:27) */
	int32_t mode;             /* hxr_render_mode */
	int32_t spp;              /* MONTE_CARLO: rays per pixel; 0 = max(dof ? camera.num_samples : 0, gi ? num_paths : 0) */
	uint64_t rng_seed;        /* counter-based generator key; 5489 = std::mt19937's default seed, used as OUR default */
	int32_t tile_size;        /* sharding granule in pixels (square); 0 = 64 like the reference's buckets (src/main.cpp:515) */
	int32_t shard_rank;       /* this process renders the tiles whose NUMBER t has t % shard_count == shard_rank.  Numbers run row-major; with
	                             shard_count > 1 every tile row is rotated by three more columns than the row above it, so that a shard does not own
	                             the same columns all the way down (hxr_shard_tile_rect gives the rectangles; hxr_assemble_frame undoes it) */
	int32_t shard_count;      /* 0 or 1 = whole frame */
	int32_t layout;           /* hxr_output_layout */
	int32_t collect_counters; /* nonzero: run the instrumented kernel build and fill hxr_get_counters() */
	/* Foveated rendering of renderWithoutMonteCarlo (GlobalSettings::foveatedRadius, src/main.cpp:284-311,203-205; modes SINGLE and
	 * ADAPTIVE_AA): 8x8 pixel blocks whose centre lies farther than foveated_radius pixels from the point of interest (vip_x, vip_y)
	 * -- the reference's vipX / vipY, the mouse position; (-100, -100) until the mouse has moved -- are filled with the mean of five
	 * rays, and only pixels inside the radius are candidates for anti-aliasing.  0 = off (every pixel is traced). */
	int32_t foveated_radius;
	int32_t vip_x, vip_y;
	int32_t reserved_[2];
} hxr_frame_params;

/* Work counters of the last frame (feeds the algorithmic-bytes formula). */
typedef struct hxr_counters {
	uint64_t camera_samples;
	uint64_t raycasts;        /* TraceContext::raycast calls (src/main.cpp:63) */
	uint64_t shadow_queries;  /* visible() calls (src/main.cpp:171) */
	uint64_t node_tests;      /* Node::intersect calls (src/geometry.cpp:196) */
	uint64_t bbox_tests;      /* BBox::testIntersect calls (src/bbox.h:94) */
	uint64_t kd_visits;       /* kd nodes entered (src/mesh.cpp:221) */
	uint64_t leaf_indices;    /* leaf list entries read */
	uint64_t tri_tests;       /* Mesh::intersectTriangle calls (src/mesh.cpp:178) */
	uint64_t mesh_hits;       /* accepted mesh hits whose attributes were interpolated */
	uint64_t tex_bilinear;    /* Bitmap::getFilteredPixel from textures / bump maps */
	uint64_t env_bilinear;    /* Bitmap::getFilteredPixel from the cubemap */
	uint64_t hf_calls;        /* Heightfield::intersect calls */
	uint64_t hf_highest;      /* getHighest reads */
	uint64_t hf_height;       /* getHeight + maxH reads */
	uint64_t hf_tri_tests;    /* intersectTriangleFast calls */
	uint64_t hf_normals;      /* getNormal calls */
	uint64_t aa_pixels;       /* pixels flagged by detectAApixels */
	uint64_t box_test_mismatch; /* instrumented GPU build: box tests whose slab form and literal BBox::testIntersect form disagreed */
	uint64_t reserved_[6];
} hxr_counters;

typedef struct hxr_scene hxr_scene;   /* opaque: the scene resident in HBM */

/* Rectangle finished callback: maps to displayVFBRect(r, vfb) (reference src/main.cpp:312,340,379). */
typedef void (*hxr_progress_cb)(void* user, int x0, int y0, int x1, int y1);

/* ------------------------------------------------------------------ entry points */

/* Number of gfx950 devices visible to this process (0 if none).  Never fails. */
int hxr_device_count(void);

/* Bind this process to one device (one process per GPU).  Replaces the reference's
 * `threadPool = std::make_unique<ThreadPool>(numThreads)` (src/main.cpp:543-545). */
int hxr_init(int device_id);

/* Deep-copy a flattened scene into HBM (re-laid out for the kernels).  Caller keeps the host arrays.
 * Replaces scene.beginRender()+beginFrame() as far as the render path is concerned (src/scene.cpp:754-762).
 * Returns NULL on failure; hxr_last_error() says why. */
hxr_scene* hxr_scene_create(const hxr_scene_desc* desc);

/* Acceleration structure walked by Mesh::intersect's replacement. */
enum hxr_accel {
	HXR_ACCEL_REFERENCE = 0,  /* the kd-tree in the descriptor, i.e. the reference's own (Mesh::buildKD, src/mesh.cpp:86-122) */
	HXR_ACCEL_SAH = 1         /* kd-trees rebuilt per mesh with the surface-area heuristic ("a better
	                             tree proven <= 1e-4 against the reference-identical one"); same walk, same leaf rule, same
	                             triangle test -- ties between triangles hit at bit-equal distance may resolve differently */
	,
	HXR_ACCEL_REFERENCE_DEVICE_BUILD = 2 /* the reference's tree again (midpoint splits, axis = depth % 3, leaves below 20 triangles or
	                             past depth 64, BBox::intersectTriangle decides membership), but BUILT BY THE MODULE ON THE GPU from
	                             the mesh's triangles at hxr_scene_create: the descriptor's kd_nodes / leaf_indices are not read
	                             and may be absent (num_kd_nodes = 0 with has_kd = 1).  See hxr_kd_build_device. */
};

typedef struct hxr_scene_options {
	uint32_t struct_size;     /* sizeof(hxr_scene_options): lets the record grow */
	int32_t accel;            /* enum hxr_accel */
	int32_t no_adaptive_schedule; /* 1: hand the ray packets out in frame order every frame.  Default (0): packets that were
	                             expensive in the previous frame of the same size / mode / shard are started first and split
	                             into one-pixel packets -- a schedule only, every frame still traces every sample and the
	                             image does not depend on it ("adaptive schedule") */
	int32_t deferred_rays;    /* 1: Whitted frames over meshes append the Reflection / Refraction rays of their hits to a queue and a
	                             continuation pass traces them re-packed, 64 unrelated rays per wavefront, each lane walking the
	                             kd-trees for itself.  Default (0): they are traced in place, on a per-lane
	                             stack, inside the render kernel.  Same rays, same work counters; the frames of the two settings
	                             differ at float rounding level (summation order). */
	int32_t device_prepare_triangles; /* 1: Mesh::prepareTriangles (src/mesh.cpp:370-396) runs on the GPU at hxr_scene_create: the derived fields of
	                             hxr_triangle (gnormal, AB, AC, ABcrossAC, dNdx, dNdy) are computed there from the v / t numbers and are not read
	                             from the descriptor (the host loader's HXH_GPU_PREPARE_TRIANGLES leaves them zero).  hxr_scene_verify_tables
	                             compares them with a complete descriptor's. */
} hxr_scene_options;

/* hxr_scene_create with options (NULL = defaults = hxr_scene_create). */
hxr_scene* hxr_scene_create_ex(const hxr_scene_desc* desc, const hxr_scene_options* options);

/* Host-only helper (touches no GPU): the kd-tree HXR_ACCEL_SAH builds for mesh `mesh` of `desc`, in the descriptor's own
 * node format, so that it can be inspected or walked by other code (the CPU tests hand it to the oracle).  Call with
 * nodes == NULL to learn the sizes; returns 0 or a negative HXR_E_*. */
int hxr_sah_build(const hxr_scene_desc* desc, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf_indices,
                  int64_t* num_nodes, int64_t* num_leaf_indices, int* max_depth);

/* Mesh::buildKD (reference src/mesh.cpp:95-122) on the GPU: the tree HXR_ACCEL_REFERENCE_DEVICE_BUILD builds for mesh `mesh` of
 * `desc`, returned in the descriptor's node format and in the numbering of a depth-first host build (a node's child pair is
 * appended when the node is split, left subtree first; leaf lists in depth-first order), so that it can be compared with a
 * host-built tree entry by entry or handed to other code.  Needs hxr_init; uploads the mesh's triangles and vertices itself.
 * Call with nodes == NULL to learn the sizes.  *build_ms (optional): the build without the upload, copy back and renumbering.
 * Arithmetic: IEEE FP64 with the device compiler's FMA contraction.  A host build compiled with other contraction / fast-math
 * choices can classify a triangle that only grazes a split plane differently (the reference's own tree depends on its compiler
 * flags in the same way); tests/test_gpu_kd_build.py compares the two builds for every mesh under data/geom. */
int hxr_kd_build_device(const hxr_scene_desc* desc, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf_indices,
                        int64_t* num_nodes, int64_t* num_leaf_indices, int* max_depth, double* build_ms);

/* The module's surface-area-heuristic tree (hxr_scene_options.accel = HXR_ACCEL_SAH) of one mesh, built by kernels: node for node the tree
 * hxr_sah_build makes on the host.  Same arguments and conventions as hxr_kd_build_device. */
int hxr_sah_build_device(const hxr_scene_desc* desc, int mesh, hxr_kd_node* nodes, int64_t max_nodes, int32_t* leaf_indices, int64_t max_leaf_indices,
                        int64_t* num_nodes, int64_t* num_leaf_indices, int* max_depth, double* build_ms);

/* Nodes / leaf entries / depth of the tree actually walked for mesh `mesh` (after an HXR_ACCEL_SAH rebuild these
 * differ from the descriptor's).  The reference prints the same statistics at load (src/mesh.cpp:39-47). */
int hxr_scene_tree_stats(const hxr_scene* scene, int mesh, int64_t* nodes, int64_t* leaf_entries, int* max_depth);

/* Checks the tables the module prepared on the GPU at hxr_scene_create -- the leaf-ordered hot triangle records and, when the
 * scene was created from a descriptor WITHOUT heightfield maxima tables, those tables -- against the host's: every hot record is
 * re-derived from `desc`, the maxima tables are compared with the ones `desc` carries (pass the complete descriptor here).
 * *mismatches receives the number of differing records / values: the preparation is pure copies and maxima, so it must be 0.
 * A scene created from hxr_heightfield::raw_heights is compared (heights, box, normals, maxima) with a tolerance of 1e-5 of the height range:
 * its blur adds the same taps in another order than the host's.  Bitmaps created from hxr_bitmap::prep are compared bit for bit with the
 * descriptor's decoded / differentiated texels (to 1e-6 where the gamma curve had no host table). */
int hxr_scene_verify_tables(const hxr_scene* scene, const hxr_scene_desc* desc, int64_t* mismatches);

/* Per-frame camera update (Camera::beginFrame, src/camera.cpp:30-63, after move/rotate). */
int hxr_scene_update_camera(hxr_scene* scene, const hxr_camera* camera);

/* Blocking frame render into a HOST buffer.  Replaces render() (src/main.cpp:416-426).
 * `vfb` receives float RGB at byte offset y*row_stride_bytes + x*12 (pass &vfb[0][0], 23040 for the
 * reference's global).
 * `cancel` may be NULL.  While the frame runs the calling thread watches *cancel (the reference polls checkForUserExit() once
 * per bucket row, src/main.cpp:280,333,370); when it becomes nonzero a device-side word is raised, every wavefront stops taking
 * ray packets (each finishes the 64 samples it holds) and the call returns HXR_CANCELLED with an incomplete frame, like render()
 * returning false.
 * `cb` may be NULL.  It is called once per tile (params->tile_size, default 64 like the reference's buckets) from the calling
 * thread, maps to displayVFBRect(r, vfb) (src/main.cpp:312,340,379): for 1-spp and Monte-Carlo frames AS SOON AS all pixels of
 * the tile have been written -- the tile is copied into `vfb` first, the rest of the frame is still rendering -- for adaptive-AA
 * and deferred-ray frames when the whole frame is final. */
int hxr_render(hxr_scene* scene, const hxr_frame_params* params, float* vfb, size_t row_stride_bytes,
               const volatile int* cancel, hxr_progress_cb cb, void* user);

/* Asynchronous frame render into a DEVICE buffer on `hip_stream` (a hipStream_t, NULL = default
 * stream).  Used by one-process-per-GPU sharded runs: each rank renders its tiles, the ranks' buffers
 * are gathered with one RCCL gather.  `d_out` must hold hxr_output_bytes() bytes. */
int hxr_render_device(hxr_scene* scene, const hxr_frame_params* params, float* d_out,
                      size_t row_stride_bytes, void* hip_stream);

/* Bytes hxr_render_device writes for these parameters (layout FRAME: height*row_stride_bytes). */
size_t hxr_output_bytes(const hxr_scene* scene, const hxr_frame_params* params, size_t row_stride_bytes);

/* Number of tiles this shard owns, and the pixel rectangle of its i-th tile.  Pure bookkeeping: `scene`
 * may be NULL when width, height and mode are given explicitly (no GPU is touched). */
int hxr_shard_tile_count(const hxr_scene* scene, const hxr_frame_params* params);
int hxr_shard_tile_rect(const hxr_scene* scene, const hxr_frame_params* params, int i,
                        int* x0, int* y0, int* x1, int* y1);

/* Scatter tile-major shard buffers (as gathered from all ranks, rank-major) into a row-major frame.
 * All pointers are DEVICE pointers; runs on `hip_stream`. */
int hxr_assemble_frame(const hxr_scene* scene, const hxr_frame_params* params, const float* d_gathered,
                       size_t per_rank_bytes, float* d_frame, size_t row_stride_bytes, void* hip_stream);

/* float RGB -> 8-bit sRGB through the reference's 4097-entry table (src/sdl.cpp:404-419,
 * src/color.h:36-47,84-90).  DEVICE pointers; `d_bgr` receives B,G,R bytes per pixel, rows top-down. */
int hxr_quantize_srgb8(const float* d_frame, size_t row_stride_bytes, int width, int height,
                       uint8_t* d_bgr, void* hip_stream);

/* One frame of the interactive loop (gameloop(), src/main.cpp:443-504): Camera::beginFrame's result for this frame (`camera`, NULL =
 * unchanged), render(false), displayVFB (src/sdl.cpp:79-90) -- the frame is converted on the GPU with Color::toRGB32(rshift, gshift,
 * bshift) and lands in the caller's 32-bit display surface (`surface` = SDL_Surface::pixels, `pitch_bytes` = SDL_Surface::pitch; any
 * host memory).  Blocking; *latency_ms (may be NULL) receives the wall-clock time from entry to the surface being complete.  The float
 * frame of the last call stays in HBM: hxr_interactive_float_frame copies it out (screenshots, tests). */
int hxr_interactive_frame(hxr_scene* scene, const hxr_frame_params* params, const hxr_camera* camera, uint32_t* surface, size_t pitch_bytes,
                          int rshift, int gshift, int bshift, double* latency_ms);
int hxr_interactive_float_frame(hxr_scene* scene, const hxr_frame_params* params, float* vfb, size_t row_stride_bytes);

int hxr_get_counters(const hxr_scene* scene, hxr_counters* out);

/* Milliseconds the dominant render kernel took in the last hxr_render / hxr_render_device call that
 * has completed (HIP events on the launch stream); negative if unknown. */
double hxr_last_kernel_ms(const hxr_scene* scene);

/* Where the last frame's GPU time went (HIP events on the launch stream; blocks until the frame has finished).  An adaptive-AA
 * frame is two rounds of render -> continuation -> resolve with the detect kernel between them; the fields are sums over both. */
typedef struct hxr_frame_timing {
	double total_ms;          /* first kernel of the frame to the last: what hxr_last_kernel_ms() returns */
	double render_ms;         /* hxr_render_kernel: camera rays, their shading and shadow rays (+ everything, when nothing is deferred) */
	double continue_ms;       /* hxr_continue_kernel: the deferred secondary rays; 0 when nothing is deferred */
	double resolve_ms;        /* hxr_resolve_kernel (+ the AA detect kernel) */
	uint64_t deferred_rays;   /* records the render kernel(s) appended to the ray queue */
	/* what the adaptive schedule listed for the frame's LAST scheduled pass (0 when the frame was not scheduled): items handed out from the front of the list
	 * (pixels / sample parts of chunks that were expensive, whole chunks long against the frame), items at the back (whole chunks / runs of sixteen pixels that
	 * only start first), pixels split by samples.  Diagnostics: the image never depends on them (they occupy what was reserved_[3]). */
	uint64_t sched_front_items, sched_back_items, sched_part_pixels;
} hxr_frame_timing;
int hxr_get_frame_timing(const hxr_scene* scene, hxr_frame_timing* out);

void hxr_scene_destroy(hxr_scene* scene);

/* ------------------------------------------------------------------ one process, several GPUs
 * The frame's tiles (params->tile_size, default 8: small tiles spread the expensive image regions over the devices) are dealt
 * round-robin to the devices, every device renders its tiles from its own replica of the scene, the tile buffers travel to the
 * first device with one peer-to-peer copy each (xGMI), which assembles the row-major frame and copies it to the host: the
 * C-level twin of bench.py's one-process-per-GPU RCCL gather for hosts that, like the reference's main() (src/main.cpp:506-526),
 * are a single process.  The frame is bit-identical to the single-GPU frame for every number of devices.  A device may be
 * listed more than once (each entry gets its own replica and its own share of the tiles). */
typedef struct hxr_group hxr_group;
hxr_group* hxr_group_create(const hxr_scene_desc* desc, const hxr_scene_options* options, const int* device_ids, int num_devices);
int hxr_group_update_camera(hxr_group* group, const hxr_camera* camera);
/* Same contract as hxr_render (cancel is honoured by every device; `cb` is called per tile once the frame has been assembled). */
int hxr_group_render(hxr_group* group, const hxr_frame_params* params, float* vfb, size_t row_stride_bytes,
                     const volatile int* cancel, hxr_progress_cb cb, void* user);
/* Wall-clock milliseconds of the last hxr_group_render call from its first launch to the assembled frame on the first device
 * (the copy to the host excluded); per_device_kernel_ms (may be NULL, num_devices entries) receives each device's render time. */
double hxr_group_last_ms(const hxr_group* group, double* per_device_kernel_ms);
int hxr_group_size(const hxr_group* group);
void hxr_group_destroy(hxr_group* group);

/* Thread-local, never NULL. */
const char* hxr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HEXRAY_HIP_H */
