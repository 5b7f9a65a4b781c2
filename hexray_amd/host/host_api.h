/*
 * host_api.h -- C entry points of libhexray_host.so, the host side that stays on the CPU:
 * .hexray loader, scene flattening, camera, bitmap output.  It produces the hxr_scene_desc that
 * libhexray_hip.so (include/hexray_hip.h) consumes.  Used by the `hexray` command-line program, by
 * the Python binding (hexray_amd/__init__.py) and by the tests.
 */
#ifndef HEXRAY_HOST_API_H
#define HEXRAY_HOST_API_H

#include <stddef.h>
#include "hexray_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hxh_scene hxh_scene;

/* Parse `scene_path` (+ optional overlay text in .hexray syntax), load its assets, run the
 * beginRender/beginFrame preparation and flatten.  NULL on failure with a message in err. */
hxh_scene* hxh_scene_load(const char* scene_path, const char* overlay_text, char* err, size_t err_len);
/* flags: HXH_GPU_HEIGHTFIELD_TABLES = leave Heightfield::maxH / highMap (src/heightfield.cpp:49-81,222-235) to the render module, which
 * builds them on the GPU (hxr_heightfield::first_maxh = first_highmap = -1).  A descriptor loaded this way is for the HIP module only. */
/*        HXH_GPU_KD_BUILD = skip Mesh::buildKD (src/mesh.cpp:95-122) on the host: meshes that want a tree say has_kd = 1 with num_kd_nodes = 0, and the
 * device scene must be created with HXR_ACCEL_REFERENCE_DEVICE_BUILD (or HXR_ACCEL_SAH), which build it from the triangles. */
/*        HXH_GPU_PREPARE_TRIANGLES = skip Mesh::prepareTriangles (src/mesh.cpp:370-396): gnormal / AB / AC / ABcrossAC / dNdx / dNdy stay zero (except for
 * autoSmooth meshes, whose smoothing reads gnormal) and hxr_scene_options.device_prepare_triangles must be set when the device scene is created. */
/*        HXH_GPU_HEIGHTFIELD_BLUR = leave ALL of a heightfield's preparation after the image load to the render module (hxr_heightfield::raw_heights):
 * Gaussian blur, bounding box, normals and maxima tables (src/heightfield.cpp:173-261); implies HXH_GPU_HEIGHTFIELD_TABLES. */
/*        HXH_GPU_BITMAP_PREP = hand bitmaps over as their files hold them and leave Bitmap::decompressGamma (BitmapTexture / cubemap assumedGamma) and
 * Bitmap::differentiate (BumpTexture) to the render module (hxr_bitmap::prep; src/bitmap.cpp:304-338). */
enum { HXH_GPU_HEIGHTFIELD_TABLES = 1u, HXH_GPU_KD_BUILD = 2u, HXH_GPU_PREPARE_TRIANGLES = 4u, HXH_GPU_HEIGHTFIELD_BLUR = 8u, HXH_GPU_BITMAP_PREP = 16u };
hxh_scene* hxh_scene_load_ex(const char* scene_path, const char* overlay_text, unsigned flags, char* err, size_t err_len);
const hxr_scene_desc* hxh_scene_desc(const hxh_scene* s);
void hxh_scene_free(hxh_scene* s);

/* Shell-only settings (GlobalSettings fields the render path does not read). */
int hxh_scene_num_threads(const hxh_scene* s);
int hxh_scene_prepass_samples(const hxh_scene* s);
int hxh_scene_interactive(const hxh_scene* s);
int hxh_scene_num_warnings(const hxh_scene* s);
const char* hxh_scene_warning(const hxh_scene* s, int i);

/* The statistics Mesh::beginRender prints (reference src/mesh.cpp:39-47). */
int hxh_scene_kd_stats(const hxh_scene* s, int mesh, int* nodes, int* leaf_nodes, int* max_depth,
                       double* avg_depth, double* avg_tris_per_leaf);

/* BBox::testIntersect evaluations made by the kd builder for this mesh (pins the builder against the
 * reference's profile: its gprof count includes build-time calls). */
long long hxh_scene_kd_build_box_tests(const hxh_scene* s, int mesh);

/* Camera::move / Camera::rotate followed by Camera::beginFrame; refreshes desc->camera. */
int hxh_camera_move(hxh_scene* s, double sideways, double front_back);
int hxh_camera_rotate(hxh_scene* s, double yaw_diff, double pitch_diff);

/* getBucketsList (reference src/sdl.cpp:280-294): writes x0,y0,x1,y1 per bucket, returns the count. */
int hxh_bucket_list(int width, int height, int bucket_size, int* rects, int capacity);

/* Bitmap::saveBMP of a float frame through the sRGB table; 0 on success. */
int hxh_save_bmp(const char* path, const float* rgb, int width, int height, size_t row_stride_bytes);
/* Bitmap::saveEXR (src/bitmap.cpp:270-288): RGBA, 16-bit HALF, alpha 1 (written uncompressed; see exr_writer.cpp) */
int hxh_save_exr(const char* path, const float* rgb, int width, int height, size_t row_stride_bytes);
unsigned hxh_float_to_half(float x);
/* Color::toRGB32 per channel. */
unsigned hxh_srgb8(float x);
/* Load an image (BMP / EXR) as float RGB; caller frees with hxh_free_pixels. */
float* hxh_load_image(const char* path, int* width, int* height);
void hxh_free_pixels(float* p);
/* raw dump: int32 W, int32 H, float32[H][W][3] */
int hxh_save_raw(const char* path, const float* rgb, int width, int height, size_t row_stride_bytes);

#ifdef __cplusplus
}
#endif
#endif
