// Mesh preparation on the GPU ("scene preparation on the fast path"): Mesh::buildKD and Mesh::prepareTriangles.
//
// Builds the REFERENCE's kd-tree of one mesh -- midpoint split on axis depth % 3, a node becomes a leaf below 20
// triangles or past depth 64, a triangle goes to every child whose box BBox::intersectTriangle says it touches
// (reference src/mesh.cpp:95-122, src/bbox.h:177-207) -- level by level on the device, from the triangle and vertex
// tables hxr_scene_create has already put in HBM.  The result comes back in the descriptor's own node format and in
// the host builder's numbering (children pairs in pre-order, leaf lists in depth-first order), so it can be laid out,
// inspected, compared and handed to the CPU oracle exactly like a tree the caller supplied.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "hexray_hip.h"

namespace hxr_kd {

struct DeviceBuiltTree {
	std::vector<hxr_kd_node> nodes;     // node 0 = root; children adjacent (first, first + 1)
	std::vector<int32_t> leaf_indices;  // triangle numbers, local to the mesh
	int max_depth = 0;
	int levels = 0;                     // kernel launches the build took
	double build_ms = 0;                // device work + per-level hand-shake, without the final copy back and renumbering
};

// d_tris / d_vertices: device pointers to the mesh's triangles and to its vertex array (3 doubles per vertex).
// Returns 0 or a negative HXR_E_*; `err` receives a message on failure.
int build_reference_tree(const hxr_triangle* d_tris, int64_t num_triangles, const double* d_vertices, const double bbox_min[3], const double bbox_max[3],
                         DeviceBuiltTree& out, char* err, size_t err_len);

// The surface-area-heuristic tree of hxr_accel.h (hxr_sah::SahBuilder, the module's own optional acceleration structure) built on the device:
// the same tree node for node (split planes, leaf contents and their order) in the same numbering -- tests/test_gpu_kd_build.py compares them.
// The five numbers are hxr_sah::SahParams.
int build_sah_tree(const hxr_triangle* d_tris, int64_t num_triangles, const double* d_vertices, const double bbox_min[3], const double bbox_max[3],
                   double traversal_cost, double intersect_cost, double empty_bonus, int max_depth, int min_leaf, DeviceBuiltTree& out, char* err, size_t err_len);

// Mesh::prepareTriangles (src/mesh.cpp:370-396) on the GPU: gnormal, AB, AC, ABcrossAC, dNdx, dNdy of `num_triangles` records are
// (re)computed in place from the vertex and uv numbers they carry.  Asynchronous on the null stream; returns 0 or HXR_E_HIP.
int prepare_triangles(hxr_triangle* d_tris, int64_t num_triangles, const double* d_vertices, const double* d_uvs);

}  // namespace hxr_kd
