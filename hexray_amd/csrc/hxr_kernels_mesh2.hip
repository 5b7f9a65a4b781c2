// The catch-all variants for scenes with meshes (path tracing over meshes alone: hxr_kernels_mesh_gi.hip), a unit of their own so that the
// units compile side by side; -O2 with the max-ilp scheduling strategy (build.sh).
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#ifndef HXR_TRI_DECIDE_SHORT_CIRCUIT
#define HXR_TRI_DECIDE_BITWISE   // triangle_decide: the spelling these kernels are faster with (cornell16 52.2 ms against 54.2)
#endif
#define HXR_LEAF_TOUCH   // round 4: the Whitted catch-alls (three waves per SIMD) prefetch the next hot record in their leaf loop: kdtree_test 9.58 -> 9.43 ms
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_MESH_OTHER(HXR_INSTANTIATE)
