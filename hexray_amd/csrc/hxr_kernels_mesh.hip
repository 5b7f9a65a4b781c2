// The Whitted variants of the render kernel that walk kd-trees (3 waves per SIMD) and their continuation kernels, in a translation
// unit of their own so that they can be compiled at -O2 (2 % faster than -O3 here: 9.03 -> 8.85 ms) with
// `-mllvm -amdgpu-sched-strategy=max-ilp`: the packet kd walk gains from that scheduling strategy (headline frame 9.85 -> 9.60 ms),
// the heightfield marcher loses 3 % and the sphere path tracer is indifferent (see hxr_variants.h, build.sh).
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#define HXR_LEAF_TOUCH   // round 4: every kernel of this unit (three waves per SIMD, launches a few packets deep: 1-spp passes, adaptive-AA passes) prefetches the next hot record in its leaf loop (HXR_F_TOUCH): meshes 1 spp 1.08 -> 1.01 ms, simple 0.75 -> 0.72
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_MESH_DEFER(HXR_INSTANTIATE)
HXR_VARIANTS_MESH_INPLACE(HXR_INSTANTIATE_LAUNCH_MODE)
HXR_VARIANTS_MESH_PARTS(HXR_INSTANTIATE_LAUNCH_MODE)   // the replicated-packet kernel of 1-spp and adaptive-AA passes   // their Monte-Carlo instances -- the headline kernel -- live in hxr_kernels_mesh_mc.hip
HXR_VARIANTS_MESH_STACK1(HXR_INSTANTIATE)
HXR_VARIANTS_MESH_INPLACE(HXR_INSTANTIATE_CONTINUE)
HXR_VARIANTS_MESH_LANEWALK(HXR_INSTANTIATE_CONTINUE)
