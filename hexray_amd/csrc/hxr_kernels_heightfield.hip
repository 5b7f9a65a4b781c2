// The heightfield Whitted variant of the render kernel in a translation unit of its own: it is 1.8 % faster at -O2 than at -O3
// (9.46 vs 9.64 ms on heightfield.hexray 1080p x 16 spp) and loses 3 % under the max-ilp scheduling strategy of the kd-walking
// variants, while the sphere path tracer in hxr_api.hip is 1 % faster at -O3.
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_HEIGHTFIELD(HXR_INSTANTIATE)
