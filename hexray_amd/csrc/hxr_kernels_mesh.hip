// The variants of the render kernel that walk kd-trees (3 waves per SIMD), in a translation unit of their own so that they can
// be compiled with `-mllvm -amdgpu-sched-strategy=max-ilp`: the packet kd walk gains from that scheduling strategy (headline
// frame 9.85 -> 9.60 ms, cornell box 66.5 -> 65.5 ms), the heightfield marcher loses 3 % and the sphere path tracer is
// indifferent, so those stay with the default in hxr_api.hip (which declares the ones below `extern template`).
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#include "hxr_kernels.hip.h"

template __global__ void hxr_render_kernel<false, HXR_F_MESH, 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*,
                                                                 const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*,
                                                                 const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<true, HXR_F_MESH, 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*,
                                                                const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*,
                                                                const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<false, (HXR_F_MESH | HXR_F_GI | HXR_F_RECT), 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<true, (HXR_F_MESH | HXR_F_GI | HXR_F_RECT), 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<false, (HXR_F_ALL), 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<true, (HXR_F_ALL), 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<false, (HXR_F_ALL & ~HXR_F_GI), 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
template __global__ void hxr_render_kernel<true, (HXR_F_ALL & ~HXR_F_GI), 3>(const HxrSceneDev*, const HxrNodeDev*, const HxrMeshDev*, const HxrHeightfieldDev*, const HxrBitmapDev*, const hxr_texture*, const hxr_shader*, const hxr_layer*, const hxr_light*, const HxrKdSlim*, const HxrKdBox*, const HxrTriHot*, const int32_t*, const hxr_geometry*, HxrLaunch);
