// hexray -- headless command-line front end of the MI355X render path.
// Stands where the reference's main() / renderThreadEntry() / renderStatic() stand (reference
// src/main.cpp:506-569): parse the scene, prepare it, render one frame, report "Elapsed time", save
// the frame.  The reference opens an SDL window and can only save on F12 (src/sdl.cpp:103-116,201);
// there is no window here, and W x H / spp / AA can be overridden from the command line (the
// reference reads them from the scene file only).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "hexray_hip.h"
#include "host_api.h"

static void usage()
{
	printf("usage: hexray [scene.hexray] [--width W] [--height H] [--aa on|off] [--dof on|off] [--spp N]\n"
	       "              [--gi on|off] [--paths N] [--mode auto|single|aa|mc|prepass] [--seed S] [--device D]\n"
	       "              [--frames N] [--out frame.bmp|frame.exr|frame.raw] [--counters] [--accel reference|sah|device-build]\n"
	       "              [--no-adaptive-schedule] [--deferred-rays] [--gpus N | --devices a,b,...] [--tile T] [--progress]\n"
	       "              [--interactive FRAMES] [--fovea RADIUS] [--gpu-heightfield-blur]\n");
}

int main(int argc, char** argv)
{
	std::string scene = "data/simple.hexray", out;
	std::string g, c;   // overlay lines for GlobalSettings / Camera
	hxr_frame_params fp;
	memset(&fp, 0, sizeof fp);
	fp.rng_seed = 5489;
	int device = 0, frames = 1;
	std::vector<int> devices;   // more than one entry: the frame is sharded over them (hxr_group_*)
	bool progress = false;
	bool gpu_blur = false;        // Heightfield blur / box / normals on the GPU as well (hxr_heightfield::raw_heights)
	int interactive_frames = 0;   // > 0: the interactive loop (gameloop(), src/main.cpp:443-504) over a scripted camera path
	hxr_scene_options opt;
	memset(&opt, 0, sizeof opt);
	opt.struct_size = sizeof opt;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
		if (a == "--width") g += std::string("frameWidth ") + next() + "\n";
		else if (a == "--height") g += std::string("frameHeight ") + next() + "\n";
		else if (a == "--aa") g += std::string("wantAA ") + next() + "\n";
		else if (a == "--gi") g += std::string("gi ") + next() + "\n";
		else if (a == "--paths") g += std::string("numPaths ") + next() + "\n";
		else if (a == "--dof") c += std::string("dof ") + next() + "\n";
		else if (a == "--spp") { const char* v = next(); c += std::string("numSamples ") + v + "\n"; fp.spp = atoi(v); }
		else if (a == "--mode") {
			std::string m = next();
			fp.mode = m == "single" ? HXR_MODE_SINGLE : m == "aa" ? HXR_MODE_ADAPTIVE_AA : m == "mc" ? HXR_MODE_MONTE_CARLO : m == "prepass" ? HXR_MODE_PREPASS : HXR_MODE_AUTO;
		} else if (a == "--seed") fp.rng_seed = strtoull(next(), nullptr, 10);
		else if (a == "--device") device = atoi(next());
		else if (a == "--frames") frames = atoi(next());
		else if (a == "--out") out = next();
		else if (a == "--counters") fp.collect_counters = 1;
		else if (a == "--accel") { std::string v = next(); opt.accel = v == "sah" ? HXR_ACCEL_SAH : v == "device-build" ? HXR_ACCEL_REFERENCE_DEVICE_BUILD : HXR_ACCEL_REFERENCE; }
		else if (a == "--no-adaptive-schedule") opt.no_adaptive_schedule = 1;
		else if (a == "--deferred-rays") opt.deferred_rays = 1;
		else if (a == "--gpus") { int n = atoi(next()); devices.clear(); for (int k = 0; k < n; k++) devices.push_back(k); }
		else if (a == "--devices") { std::string v = next(); devices.clear(); for (size_t at = 0; at < v.size();) { devices.push_back(atoi(v.c_str() + at)); size_t c = v.find(',', at); if (c == std::string::npos) break; at = c + 1; } }
		else if (a == "--tile") fp.tile_size = atoi(next());
		else if (a == "--progress") progress = true;
		else if (a == "--interactive") interactive_frames = atoi(next());
		else if (a == "--fovea") fp.foveated_radius = atoi(next());
		else if (a == "--gpu-heightfield-blur") gpu_blur = true;
		else if (a == "-h" || a == "--help") { usage(); return 0; }
		else if (!a.empty() && a[0] != '-') scene = a;
		else { usage(); return 2; }
	}
	std::string overlay = "GlobalSettings {\nprepassSamples 0\ninteractive off\n" + g + "}\n";
	if (!c.empty()) overlay += "Camera {\n" + c + "}\n";
	if (fp.mode != HXR_MODE_MONTE_CARLO && fp.mode != HXR_MODE_PREPASS) fp.spp = 0;

	char err[1024] = "";
	// the module builds the heightfield maxima tables on the GPU, and with --accel device-build the kd-trees (Mesh::buildKD) as well
	const unsigned load_flags = HXH_GPU_HEIGHTFIELD_TABLES | HXH_GPU_BITMAP_PREP | (opt.accel == HXR_ACCEL_REFERENCE_DEVICE_BUILD ? HXH_GPU_KD_BUILD : 0u) | (gpu_blur ? HXH_GPU_HEIGHTFIELD_BLUR : 0u);
	hxh_scene* hs = hxh_scene_load_ex(scene.c_str(), overlay.c_str(), load_flags, err, sizeof err);
	if (!hs) {
		printf("Could not parse the scene file (%s)!\n%s\n", scene.c_str(), err);
		return 1;
	}
	for (int i = 0; i < hxh_scene_num_warnings(hs); i++) fprintf(stderr, "%s\n", hxh_scene_warning(hs, i));
	const hxr_scene_desc* d = hxh_scene_desc(hs);
	for (int m = 0;; m++) {
		int nodes, leaves, depth;
		double avg_depth, avg_tris;
		if (hxh_scene_kd_stats(hs, m, &nodes, &leaves, &depth, &avg_depth, &avg_tris)) break;
		if (!nodes) continue;
		// the lines Mesh::beginRender prints (src/mesh.cpp:39-47)
		printf("K-d Tree statistics:\n   nodes            : %d\n   leaf nodes       : %d\n   max depth        : %d\n"
		       "   avg depth        : %.1f\n   avg tris per leaf: %.1f\n", nodes, leaves, depth, avg_depth, avg_tris);
	}
	int W = d->settings.frame_width, H = d->settings.frame_height;
	std::vector<float> vfb((size_t)W * H * 3);
	int rc = 0, tiles_reported = 0;
	// displayVFBRect's stand-in (src/main.cpp:312): count the tiles as they arrive
	auto on_tile = [](void* user, int, int, int, int) { ++*(int*)user; };
	auto t0 = std::chrono::steady_clock::now();
	hxr_scene* ds = nullptr;
	hxr_group* group = nullptr;
	if (interactive_frames > 0) {
		// The reference's interactive mode without its window: the scene stays in HBM, per frame only the camera record travels, the
		// frame comes back as the 32-bit pixels of a display surface (XRGB8888: the shifts SDL reports for a desktop window surface).
		// Keys and mouse are replaced by a fixed path: forward, slowly turning, the point of interest circling the frame centre.
		if (devices.size() == 1) device = devices[0];
		if (hxr_init(device)) { printf("GPU init failed: %s\n", hxr_last_error()); return 1; }
		ds = hxr_scene_create_ex(d, &opt);
		if (!ds) { printf("hxr_scene_create_ex failed: %s\n", hxr_last_error()); return 1; }
		if (fp.tile_size <= 0) fp.tile_size = 16;   // getBucketsList(scene.settings.interactive ? 16 : 64), src/main.cpp:515
		std::vector<uint32_t> surface((size_t)W * H);
		std::vector<double> lat;
		auto loop0 = std::chrono::steady_clock::now();
		for (int f = 0; f < interactive_frames && rc == 0; f++) {
			hxh_camera_move(hs, 0.0, 0.35);
			hxh_camera_rotate(hs, 0.4, 0.0);
			fp.vip_x = W / 2 + (int)(W * 0.25 * cos(f * 0.1));
			fp.vip_y = H / 2 + (int)(H * 0.25 * sin(f * 0.1));
			double ms = 0;
			rc = hxr_interactive_frame(ds, &fp, &hxh_scene_desc(hs)->camera, surface.data(), (size_t)W * 4, 16, 8, 0, &ms);
			lat.push_back(ms);
		}
		auto loop1 = std::chrono::steady_clock::now();
		if (rc < 0) { printf("hxr_interactive_frame failed: %s\n", hxr_last_error()); return 1; }
		double total_ms = std::chrono::duration<double, std::milli>(loop1 - loop0).count();
		printf("%d frames in %.0f ms: %.2f FPS\n", interactive_frames, total_ms, interactive_frames / (total_ms * 0.001));   // the line gameloop() prints
		std::vector<double> sorted = lat;
		std::sort(sorted.begin(), sorted.end());
		printf("frame latency (camera update -> display surface complete): min %.3f ms, median %.3f ms, max %.3f ms (first frame %.3f ms)\n", sorted.front(),
		       sorted[sorted.size() / 2], sorted.back(), lat.front());
		rc = hxr_interactive_float_frame(ds, &fp, vfb.data(), (size_t)W * 12);   // the last frame, for --out
	} else if (devices.size() > 1) {
		group = hxr_group_create(d, &opt, devices.data(), (int)devices.size());
		if (!group) { printf("hxr_group_create failed: %s\n", hxr_last_error()); return 1; }
		for (int f = 0; f < frames && rc == 0; f++) rc = hxr_group_render(group, &fp, vfb.data(), (size_t)W * 12, nullptr, progress ? +on_tile : nullptr, &tiles_reported);
	} else {
		if (devices.size() == 1) device = devices[0];
		if (hxr_init(device)) { printf("GPU init failed: %s\n", hxr_last_error()); return 1; }
		ds = hxr_scene_create_ex(d, &opt);
		if (!ds) { printf("hxr_scene_create_ex failed: %s\n", hxr_last_error()); return 1; }
		if (opt.accel != HXR_ACCEL_REFERENCE)   // the trees were (re)built by the module: its statistics instead of the loader's
			for (int m = 0; m < (int)d->num_meshes; m++) {
				int64_t nodes = 0, entries = 0;
				int depth = 0;
				if (!hxr_scene_tree_stats(ds, m, &nodes, &entries, &depth) && nodes)
					printf("K-d Tree statistics (%s):\n   nodes            : %lld\n   leaf entries     : %lld\n   max depth        : %d\n",
					       opt.accel == HXR_ACCEL_SAH ? "SAH rebuild" : "built on the GPU", (long long)nodes, (long long)entries, depth);
			}
		for (int f = 0; f < frames && rc == 0; f++) rc = hxr_render(ds, &fp, vfb.data(), (size_t)W * 12, nullptr, progress ? +on_tile : nullptr, &tiles_reported);
	}
	auto t1 = std::chrono::steady_clock::now();
	if (rc < 0) { printf("render failed: %s\n", hxr_last_error()); return 1; }
	printf("Elapsed time: %.2f seconds.\n", std::chrono::duration<double>(t1 - t0).count());
	if (group) {
		std::vector<double> each(devices.size());
		double ms = hxr_group_last_ms(group, each.data());
		printf("%d devices: %.3f ms to the assembled frame (%d x %d); render kernels:", (int)devices.size(), ms, W, H);
		for (double e : each) printf(" %.3f", e);
		printf(" ms\n");
	} else printf("render kernel: %.3f ms (%d x %d)\n", hxr_last_kernel_ms(ds), W, H);
	if (progress) printf("tiles reported: %d\n", tiles_reported);
	if (fp.collect_counters && ds) {
		hxr_counters k;
		if (!hxr_get_counters(ds, &k))
			printf("samples %llu raycasts %llu shadow %llu node tests %llu box tests %llu triangle tests %llu\n", (unsigned long long)k.camera_samples,
			       (unsigned long long)k.raycasts, (unsigned long long)k.shadow_queries, (unsigned long long)k.node_tests, (unsigned long long)k.bbox_tests,
			       (unsigned long long)k.tri_tests);
	}
	if (!out.empty()) {
		bool raw = out.size() > 4 && out.substr(out.size() - 4) == ".raw", exr = out.size() > 4 && out.substr(out.size() - 4) == ".exr";
		int w = raw ? hxh_save_raw(out.c_str(), vfb.data(), W, H, (size_t)W * 12) : exr ? hxh_save_exr(out.c_str(), vfb.data(), W, H, (size_t)W * 12)
		                                                                                 : hxh_save_bmp(out.c_str(), vfb.data(), W, H, (size_t)W * 12);
		if (w) { printf("cannot write %s\n", out.c_str()); return 1; }
	}
	if (ds) hxr_scene_destroy(ds);
	if (group) hxr_group_destroy(group);
	hxh_scene_free(hs);
	printf("Exited cleanly\n");
	return 0;
}
