// Path tracing over meshes (the cornell box's kernel): a unit of its own since round 3, so that its flags can differ from the catch-alls' of
// hxr_kernels_mesh2.hip (build.sh; both -O2 with the max-ilp scheduling strategy at present).
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#define HXR_TRI_DECIDE_BITWISE   // triangle_decide: the spelling this kernel is faster with (cornell16 52.2 ms against 54.2 at the time)
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_MESH_GI(HXR_INSTANTIATE)
