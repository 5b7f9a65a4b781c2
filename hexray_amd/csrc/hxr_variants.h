// The compiled variants of the render kernel (hxr_kernels.hip.h: template <COUNT, F, W, M>), listed ONCE.
// Each X(COUNT, F, W) stands for the two render modes known at compile time (M = 1 Monte-Carlo, M = -1 taken from
// HxrLaunch::mode).  Translation units instantiate their list (HXR_INSTANTIATE), hxr_api.hip declares the lists of the
// other units `extern template` (HXR_DECLARE_EXTERN) and instantiates the rest implicitly at its launch sites.
// Which unit a variant lives in decides its compiler flags (build.sh): the kd-walking variants gain from -O2 and the
// max-ilp scheduling strategy, the heightfield marcher from -O2 with the default strategy, the sphere path tracer from -O3.
#pragma once

#define HXR_KERNEL_PARAMS                                                                                                                              \
	const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*,     \
	    const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch

// hxr_kernels_mesh.hip: Whitted over meshes, deferring secondary rays to the continuation pass (the default for mesh scenes)
#define HXR_VARIANTS_MESH_DEFER(X)                         \
	X(false, (HXR_F_MESH | HXR_F_DEFER), 3)                \
	X(false, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_DEFER), 3) \
	X(true, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_DEFER), 3)
// ... and the same integrator tracing secondary rays in place: the continuation pass runs these (hxr_continue_kernel), and so
// do frames whose ray queue would not fit (hxr_api.hip, plan_defer)
#ifndef HXR_MESH_WAVES
#define HXR_MESH_WAVES 3   // waves per SIMD the in-place Whitted mesh kernels are compiled for (re-measured at the end of round 2: 8.9 ms at 3, 10.45 ms at 2)
#endif
#define HXR_VARIANTS_MESH_INPLACE(X)          \
	X(false, (HXR_F_MESH), HXR_MESH_WAVES)                 \
	X(false, (HXR_F_MESH | HXR_F_REFRACT), HXR_MESH_WAVES) \
	X(true, (HXR_F_MESH | HXR_F_REFRACT), HXR_MESH_WAVES)
// ... whose Monte-Carlo instances (render mode known at compile time) live in hxr_kernels_mesh_mc.hip, the one without refraction -- the headline
// kernel -- compiled for FOUR waves per SIMD: 6.86 ms against 7.12 at three (the SAH frame 5.79 against 6.00) once the schedule is told the right number of
// resident waves (resident_waves_per_cu, hxr_api.hip).  The instances whose mode arrives at launch (adaptive anti-aliasing, prepass, interactive)
// lose 10 % at four waves and stay at HXR_MESH_WAVES.  So does a launch with less than a whole frame's work per resident wave -- a rank's share of a
// multi-GPU frame: there one packet's chain of dependent steps is what the frame waits for, and a fourth wave on the SIMD only stretches it (slowest 8-way rank
// 1.62 ms at four waves, 1.13 at three; 2-way 3.69 / 3.55): mesh_mc_waves() in hxr_api.hip picks the instance per launch.  Round 4: the three-waves instance -- the latency
// instance -- also prefetches the next hot record in its leaf loop (HXR_F_TOUCH: slowest 8-way rank 1.51 -> 1.36 ms; the throughput instance loses 6 % by it).
#ifndef HXR_MESH_MC_WAVES
#define HXR_MESH_MC_WAVES 4
#endif
// End of round 4: the throughput instance alone in hxr_kernels_mesh_mc.hip, compiled with LLVM's AMDGPU register-pressure trackers (build.sh: 6.60 -> 6.42 ms, the SAH
// frame 5.64 -> 5.47); the three-waves instances in hxr_kernels_mesh_lat.hip with the flags they had (with the trackers the slowest 8-way rank goes from 1.14 to 1.19 ms).
#define HXR_VARIANTS_MESH_INPLACE_MC_THROUGHPUT(X) X(false, (HXR_F_MESH), HXR_MESH_MC_WAVES)
#define HXR_VARIANTS_MESH_INPLACE_MC(X)          \
	X(false, (HXR_F_MESH | HXR_F_TOUCH), HXR_MESH_WAVES)   \
	X(false, (HXR_F_MESH | HXR_F_REFRACT), HXR_MESH_WAVES) \
	X(true, (HXR_F_MESH | HXR_F_REFRACT), HXR_MESH_WAVES)
// ... and the instances that trace the schedule's sample parts and single pixels as REPLICATED packets with the lane-parallel leaf (HXR_F_REPL; round 4).  A few
// blocks of them are launched beside the kernel that renders the rest of the frame (hxr_api.hip, launch_render): they drain the list of such packets first and
// then take ordinary packets like everybody else.  In EVERY wave the mere presence of that leaf costs what it gains (8-way slowest rank 1.40 -> 1.61 ms with the
// code compiled in and unused, 1.37 used), in an eighth of the waves it does not.
#define HXR_VARIANTS_MESH_PARTS(X) X(false, (HXR_F_MESH | HXR_F_TOUCH | HXR_F_REPL), HXR_MESH_WAVES)
// ... with at most one secondary ray pending per lane (no Layered shader of the scene has two Reflection / Refraction layers): the
// default for mesh scenes -- the per-lane ray stack leaves scratch memory
#define HXR_VARIANTS_MESH_STACK1(X)                              \
	X(false, (HXR_F_MESH | HXR_F_STACK1), 3)                     \
	X(false, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_STACK1), 3)     \
	X(true, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_STACK1), 3)
// ... and the continuation kernels that walk the trees lane by lane (hxr_kernels_mesh.hip as well)
#ifndef HXR_LANEWALK_WAVES
#define HXR_LANEWALK_WAVES 4   // waves per SIMD the per-lane-walk kernels are compiled for
#endif
#define HXR_VARIANTS_MESH_LANEWALK(X)                                               \
	X(false, (HXR_F_MESH | HXR_F_LANEWALK), HXR_LANEWALK_WAVES)                     \
	X(false, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_LANEWALK), HXR_LANEWALK_WAVES)     \
	X(true, (HXR_F_MESH | HXR_F_REFRACT | HXR_F_LANEWALK), HXR_LANEWALK_WAVES)
#ifndef HXR_MESH_GI_WAVES
#define HXR_MESH_GI_WAVES 4   // cornell16 (round 3, with the schedule told the real occupancy): 50.3 ms at 3, 48.3 at 4 (-O2; 48.5 at -O1)
#endif
#ifndef HXR_ALL_WHITTED_WAVES
#define HXR_ALL_WHITTED_WAVES 3   // the Whitted catch-all (kdtree_test, simple, boxed, bokeh ...)
#endif
// hxr_kernels_mesh_gi.hip: path tracing over meshes
#define HXR_VARIANTS_MESH_GI(X)                                      \
	X(false, (HXR_F_MESH | HXR_F_GI | HXR_F_RECT | HXR_F_REFRACT), HXR_MESH_GI_WAVES)   \
	X(true, (HXR_F_MESH | HXR_F_GI | HXR_F_RECT | HXR_F_REFRACT), HXR_MESH_GI_WAVES)
// hxr_kernels_mesh2.hip: the catch-alls for scenes with meshes
#define HXR_VARIANTS_MESH_OTHER(X)                                      \
	X(false, (HXR_F_ALL), 3)                                            \
	X(true, (HXR_F_ALL), 3)                                             \
	X(false, (HXR_F_ALL & ~HXR_F_GI), HXR_ALL_WHITTED_WAVES)            \
	X(true, (HXR_F_ALL & ~HXR_F_GI), HXR_ALL_WHITTED_WAVES)
// hxr_kernels_nested.hip: the catch-alls for scenes whose CSG operations have operations as operands (HXR_F_CSGNEST)
#define HXR_VARIANTS_NESTED_CSG(X)                              \
	X(false, (HXR_F_ALL | HXR_F_CSGNEST), 2)                    \
	X(true, (HXR_F_ALL | HXR_F_CSGNEST), 2)                     \
	X(false, ((HXR_F_ALL & ~HXR_F_GI) | HXR_F_CSGNEST), 2)      \
	X(true, ((HXR_F_ALL & ~HXR_F_GI) | HXR_F_CSGNEST), 2)
// hxr_kernels_heightfield.hip
#ifndef HXR_HEIGHTFIELD_WAVES
#define HXR_HEIGHTFIELD_WAVES 3   // waves per SIMD of the heightfield kernels.  Round 3: 3 waves with TWO ray-stack entries in LDS (HXR_LDS_RAY_SLOTS; four allow two waves only)
                                  // and the unit compiled -O1 (build.sh): heightfield16 8.65 -> 7.77 ms; -O2 spills the same 168 registers differently and takes 9.04;
                                  // 4 waves 8.50; round 2 had 9.14 ms at 2 waves, 11.2 at 3
#endif
#ifndef HXR_SPHERE_GI_WAVES
#define HXR_SPHERE_GI_WAVES 4   // smallpt 16 paths, round 3: 29.2 ms at 3, 27.3 at 4 (-O3; 28.0 at -O1) -- round 1 had 31.4 at 3 and 32.3 at 4 with the schedule assuming 12 resident waves per CU
#endif
#ifndef HXR_HEIGHTFIELD_F
#define HXR_HEIGHTFIELD_F (HXR_F_HEIGHTFIELD | HXR_F_REFRACT | HXR_F_LDSSTACK)   // the first entries of the per-lane ray stack in LDS (round 3)
#endif
#define HXR_VARIANTS_HEIGHTFIELD(X)                  \
	X(false, HXR_HEIGHTFIELD_F, HXR_HEIGHTFIELD_WAVES) \
	X(true, HXR_HEIGHTFIELD_F, HXR_HEIGHTFIELD_WAVES)

#define HXR_INSTANTIATE(C, F, W)                                                    \
	template __global__ void hxr_render_kernel<C, F, W, 1>(HXR_KERNEL_PARAMS);      \
	template __global__ void hxr_render_kernel<C, F, W, -1>(HXR_KERNEL_PARAMS);
// one render mode only: M = 1 (Monte-Carlo, known at compile time) / M = -1 (taken from HxrLaunch::mode)
#define HXR_INSTANTIATE_MONTE_CARLO(C, F, W) template __global__ void hxr_render_kernel<C, F, W, 1>(HXR_KERNEL_PARAMS);
#define HXR_INSTANTIATE_LAUNCH_MODE(C, F, W) template __global__ void hxr_render_kernel<C, F, W, -1>(HXR_KERNEL_PARAMS);
#define HXR_DECLARE_EXTERN(C, F, W)                                                        \
	extern template __global__ void hxr_render_kernel<C, F, W, 1>(HXR_KERNEL_PARAMS);      \
	extern template __global__ void hxr_render_kernel<C, F, W, -1>(HXR_KERNEL_PARAMS);
#define HXR_INSTANTIATE_CONTINUE(C, F, W) template __global__ void hxr_continue_kernel<C, F, W>(HXR_KERNEL_PARAMS);
#define HXR_DECLARE_EXTERN_MONTE_CARLO(C, F, W) extern template __global__ void hxr_render_kernel<C, F, W, 1>(HXR_KERNEL_PARAMS);
#define HXR_DECLARE_EXTERN_LAUNCH_MODE(C, F, W) extern template __global__ void hxr_render_kernel<C, F, W, -1>(HXR_KERNEL_PARAMS);
#define HXR_DECLARE_EXTERN_CONTINUE(C, F, W) extern template __global__ void hxr_continue_kernel<C, F, W>(HXR_KERNEL_PARAMS);
