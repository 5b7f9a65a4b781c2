// The mesh-only Whitted variants of the render kernel, in a translation unit of their own so that they can be compiled with
// `-mllvm -amdgpu-sched-strategy=max-ilp`: the packet kd walk gains 2 % from that scheduling strategy (9.83 -> 9.62 ms on the
// headline frame), the heightfield marcher loses 3 % and the path tracers are indifferent, so the other variants stay with
// the default (hxr_api.hip declares these two `extern template`).
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
