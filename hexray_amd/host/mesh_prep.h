// Mesh preparation on the host: OBJ reader, per-triangle precomputation, recentring, auto-smoothing
// and the kd-tree builder.  The tree must be IDENTICAL to the reference's (midpoint split on axis
// depth%3, leaves below 20 triangles or beyond depth 64, a triangle goes to every child box it
// overlaps) because traversal order decides which hit a ray reports.
// Reference: src/mesh.cpp:49-122 (beginRender, buildKD), :265-396 (OBJ, prepareTriangles),
// src/bbox.h:177-221 (box/triangle overlap).
#pragma once
#include <string>
#include <vector>
#include "hexray_hip.h"
#include "linalg.h"

namespace hxh {

struct KdStats {
	int nodes = 0, leaves = 0, max_depth = 0;
	long long sum_depth = 0, sum_leaf_tris = 0;
	long long build_box_tests = 0;   // box/segment tests made while building (BBox::testIntersect calls)
};

struct MeshData {
	bool defer_triangle_prep = false;   // set before load_obj: leave the derived triangle fields zero (hxr_scene_options.device_prepare_triangles)
	bool faceted = false, backface_culling = false, use_kd = true, auto_smooth = false, recenter = false;
	std::vector<V3> vertices, normals, uvs;   // entry 0 is the sentinel
	std::vector<hxr_triangle> triangles;
	V3 bbox_min, bbox_max;
	std::vector<hxr_kd_node> kd_nodes;        // node 0 = root; empty when use_kd is false
	std::vector<int32_t> leaf_indices;
	KdStats stats;

	bool load_obj(const std::string& path);   // Mesh::loadFromOBJ + prepareTriangles
	void begin_render(bool build_kd = true);  // Mesh::beginRender; build_kd false: the tree is left to the render module (HXR_ACCEL_REFERENCE_DEVICE_BUILD)
};

}  // namespace hxh
