// The catch-all variants for scenes with NESTED CSG (HXR_F_CSGNEST: the unrolled recursion of CSGBase::intersect behind real calls):
// a unit of their own, so that the call ABI and the extra code stay out of every other kernel and the unit compiles beside the others.
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_NESTED_CSG(HXR_INSTANTIATE)
