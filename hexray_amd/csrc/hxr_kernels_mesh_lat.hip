// The Monte-Carlo instances of the in-place Whitted mesh kernels that are compiled for three waves per SIMD: the latency instance (a rank's share of a multi-GPU frame),
// the replicating instance launched beside it, and the instances with refraction.  A unit of their own since the end of round 4: hxr_kernels_mesh_mc.hip -- the
// throughput instance, the headline kernel -- is built with LLVM's AMDGPU register-pressure trackers, which these lose by (build.sh).  Same -O1 / max-memory-clause
// flags and the same constant-address-space table reads as before the split.
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#define HXR_CONST_TABLES
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_MESH_INPLACE_MC(HXR_INSTANTIATE_MONTE_CARLO)
HXR_VARIANTS_MESH_PARTS(HXR_INSTANTIATE_MONTE_CARLO)
