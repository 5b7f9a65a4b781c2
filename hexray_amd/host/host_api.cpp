// C entry points of libhexray_host.so; see host_api.h.
#include "host_api.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "flatten.h"

struct hxh_scene {
	std::unique_ptr<hxh::FlatScene> fs;
};

static void put_err(char* err, size_t n, const std::string& msg)
{
	if (!err || !n) return;
	snprintf(err, n, "%s", msg.c_str());
}

extern "C" {

hxh_scene* hxh_scene_load(const char* scene_path, const char* overlay_text, char* err, size_t err_len)
{
	return hxh_scene_load_ex(scene_path, overlay_text, 0, err, err_len);
}

hxh_scene* hxh_scene_load_ex(const char* scene_path, const char* overlay_text, unsigned flags, char* err, size_t err_len)
{
	try {
		auto fs = hxh::load_scene(scene_path ? scene_path : "", overlay_text ? overlay_text : "", (flags & HXH_GPU_HEIGHTFIELD_TABLES) != 0, (flags & HXH_GPU_KD_BUILD) != 0,
		                          (flags & HXH_GPU_PREPARE_TRIANGLES) != 0, (flags & HXH_GPU_HEIGHTFIELD_BLUR) != 0, (flags & HXH_GPU_BITMAP_PREP) != 0);
		hxh_scene* s = new hxh_scene;
		s->fs = std::move(fs);
		return s;
	} catch (const hxh::ParseError& e) {
		put_err(err, err_len, std::string(scene_path ? scene_path : "") + ":" + std::to_string(e.line) + ": " + e.what());
	} catch (const std::exception& e) {
		put_err(err, err_len, e.what());
	}
	return nullptr;
}

const hxr_scene_desc* hxh_scene_desc(const hxh_scene* s) { return s ? &s->fs->desc : nullptr; }
void hxh_scene_free(hxh_scene* s) { delete s; }
int hxh_scene_num_threads(const hxh_scene* s) { return s->fs->num_threads; }
int hxh_scene_prepass_samples(const hxh_scene* s) { return s->fs->prepass_samples; }
int hxh_scene_interactive(const hxh_scene* s) { return s->fs->interactive; }
int hxh_scene_num_warnings(const hxh_scene* s) { return int(s->fs->warnings.size()); }
const char* hxh_scene_warning(const hxh_scene* s, int i)
{
	return (i >= 0 && i < int(s->fs->warnings.size())) ? s->fs->warnings[i].c_str() : "";
}

int hxh_scene_kd_stats(const hxh_scene* s, int mesh, int* nodes, int* leaf_nodes, int* max_depth,
                       double* avg_depth, double* avg_tris_per_leaf)
{
	if (!s || mesh < 0 || mesh >= int(s->fs->mesh_stats.size())) return -1;
	const hxh::KdStats& k = s->fs->mesh_stats[mesh];
	if (nodes) *nodes = k.nodes;
	if (leaf_nodes) *leaf_nodes = k.leaves;
	if (max_depth) *max_depth = k.max_depth;
	if (avg_depth) *avg_depth = k.leaves ? double(k.sum_depth) / k.leaves : 0;
	if (avg_tris_per_leaf) *avg_tris_per_leaf = k.leaves ? double(k.sum_leaf_tris) / k.leaves : 0;
	return 0;
}

long long hxh_scene_kd_build_box_tests(const hxh_scene* s, int mesh)
{
	if (!s || mesh < 0 || mesh >= int(s->fs->mesh_stats.size())) return -1;
	return s->fs->mesh_stats[mesh].build_box_tests;
}

int hxh_camera_move(hxh_scene* s, double sideways, double front_back)
{
	if (!s) return -1;
	s->fs->camera_spec.move(sideways, front_back);
	s->fs->refresh_camera();
	s->fs->desc.camera = s->fs->camera;
	return 0;
}

int hxh_camera_rotate(hxh_scene* s, double yaw_diff, double pitch_diff)
{
	if (!s) return -1;
	s->fs->camera_spec.rotate(yaw_diff, pitch_diff);
	s->fs->refresh_camera();
	s->fs->desc.camera = s->fs->camera;
	return 0;
}

int hxh_bucket_list(int width, int height, int bucket_size, int* rects, int capacity)
{
	if (width <= 0 || height <= 0 || bucket_size <= 0) return 0;
	std::vector<int> r = hxh::bucket_list(width, height, bucket_size);
	int n = int(r.size() / 4);
	for (int i = 0; i < n && i < capacity; i++) memcpy(rects + i * 4, &r[size_t(i) * 4], 4 * sizeof(int));
	return n;
}

int hxh_save_bmp(const char* path, const float* rgb, int width, int height, size_t row_stride_bytes)
{
	return hxh::save_bmp(path, rgb, width, height, row_stride_bytes) ? 0 : -1;
}

int hxh_save_exr(const char* path, const float* rgb, int width, int height, size_t row_stride_bytes)
{
	return hxh::save_exr(path, rgb, width, height, row_stride_bytes) ? 0 : -1;
}

unsigned hxh_float_to_half(float x) { return hxh::float_to_half(x); }

unsigned hxh_srgb8(float x) { return hxh::srgb_table().encode(x); }

float* hxh_load_image(const char* path, int* width, int* height)
{
	hxh::Image img;
	if (!hxh::load_image(path, img) || !img.ok()) return nullptr;
	float* p = (float*)malloc(img.rgb.size() * sizeof(float));
	if (!p) return nullptr;
	memcpy(p, img.rgb.data(), img.rgb.size() * sizeof(float));
	*width = img.w;
	*height = img.h;
	return p;
}

void hxh_free_pixels(float* p) { free(p); }

int hxh_save_raw(const char* path, const float* rgb, int width, int height, size_t row_stride_bytes)
{
	FILE* f = fopen(path, "wb");
	if (!f) return -1;
	int32_t dims[2] = {width, height};
	fwrite(dims, sizeof dims, 1, f);
	for (int y = 0; y < height; y++) fwrite((const char*)rgb + size_t(y) * row_stride_bytes, sizeof(float) * 3, size_t(width), f);
	fclose(f);
	return 0;
}

}  // extern "C"
