// Path tracing over meshes and the catch-all variants for scenes with meshes: same flags as hxr_kernels_mesh.hip (cornell box
// 66.5 -> 65.5 ms under max-ilp), a unit of their own so that the two compile side by side.
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#ifndef HXR_TRI_DECIDE_SHORT_CIRCUIT
#define HXR_TRI_DECIDE_BITWISE   // triangle_decide: the spelling these kernels are faster with (cornell16 52.2 ms against 54.2)
#endif
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_MESH_OTHER(HXR_INSTANTIATE)
