// Scene loading + flattening: .hexray text -> typed element tables -> hxr_scene_desc.
// Element semantics follow the reference's fillProperties / beginRender / beginFrame callbacks
// (reference src/scene.cpp:533-566 fill order; per-class citations in flatten.cpp).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "heightfield_prep.h"
#include "hexray_hip.h"
#include "image_io.h"
#include "mesh_prep.h"
#include "scene_file.h"

namespace hxh {

// Camera as parsed (reference src/camera.h:31-72); basis() is Camera::beginFrame.
struct CameraSpec {
	V3 pos;
	double yaw = 0, pitch = 0, roll = 0, aspect = 4.0 / 3.0, fov = 90, f_number = 2.0, focal_plane_dist = 100,
	       stereo_separation = 0;
	int num_samples = 32;
	bool dof = false, auto_focus = false, present = false;
	void basis(hxr_camera& out) const;
	void move(double sideways, double front_back);   // Camera::move, src/camera.cpp:88-91
	void rotate(double yaw_diff, double pitch_diff); // Camera::rotate, src/camera.cpp:93-99
};

struct FlatScene {
	// extra settings the render path does not read but the shell does
	int num_threads = 0, prepass_samples = 5, foveated_radius = 0;
	bool interactive = false, dbg = false;
	CameraSpec camera_spec;
	std::vector<std::string> warnings;

	hxr_settings settings;
	hxr_camera camera;
	hxr_environment environment;
	std::vector<hxr_geometry> geometries;
	std::vector<std::string> geometry_names;
	std::vector<hxr_mesh> meshes;
	std::vector<KdStats> mesh_stats;
	std::vector<hxr_heightfield> heightfields;
	std::vector<hxr_triangle> triangles;
	std::vector<hxr_kd_node> kd_nodes;
	std::vector<int32_t> leaf_indices;
	std::vector<hxr_bitmap> bitmaps;
	std::vector<hxr_texture> textures;
	std::vector<std::string> texture_names;
	std::vector<hxr_shader> shaders;
	std::vector<std::string> shader_names;
	std::vector<hxr_layer> layers;
	std::vector<hxr_light> lights;
	std::vector<hxr_node> nodes;
	std::vector<std::string> node_names;
	std::vector<double> f64_pool;
	std::vector<float> f32_pool;

	bool gpu_heightfield_blur = false;     // hxr_heightfield::raw_heights: blur, box, normals and maxima tables are all left to the render module
	bool gpu_heightfield_tables = false;   // leave maxH / highMap to the render module (hxr_heightfield::first_maxh = first_highmap = -1)
	bool gpu_prepare_triangles = false;    // leave Mesh::prepareTriangles to the render module (derived hxr_triangle fields zero)
	bool gpu_bitmap_prep = false;          // leave Bitmap::decompressGamma / Bitmap::differentiate to the render module (hxr_bitmap::prep)
	bool gpu_kd_build = false;             // leave Mesh::buildKD to the render module (hxr_mesh::num_kd_nodes = 0 with has_kd = 1)
	hxr_scene_desc desc;
	void refresh_desc();     // point desc at the vectors
	void refresh_camera();   // recompute camera basis from camera_spec (Camera::beginFrame)
};

// Throws ParseError / std::runtime_error.  `overlay_text` may be empty.
// `gpu_heightfield_tables`: do not build the heightfield maxima tables on the host (the render module builds them on the GPU).
// `gpu_kd_build`: do not build the meshes' kd-trees on the host (create the device scene with HXR_ACCEL_REFERENCE_DEVICE_BUILD).
std::unique_ptr<FlatScene> load_scene(const std::string& scene_path, const std::string& overlay_text, bool gpu_heightfield_tables = false, bool gpu_kd_build = false,
                                      bool gpu_prepare_triangles = false, bool gpu_heightfield_blur = false, bool gpu_bitmap_prep = false);

// The reference's zig-zag bucket list (src/sdl.cpp:280-294): rects as x0,y0,x1,y1.
std::vector<int> bucket_list(int width, int height, int bucket_size);

}  // namespace hxh
