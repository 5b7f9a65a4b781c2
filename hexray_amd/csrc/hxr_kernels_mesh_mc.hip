// The Monte-Carlo instances of the in-place Whitted mesh kernels -- hxr_render_kernel<false, HXR_F_MESH, 3, 1> is the headline kernel of
// bench.py -- in a unit of their own, compiled at -O1 (build.sh).  At -O2 the same kernel keeps 51 of its values in scratch memory, at -O1 9:
// the frame takes the same 7.1 ms either way (the spills are not what it waits for), but the scratch traffic behind it drops from
// 3.56 GB to 0.61 GB per launch (WRITE_SIZE 2.15 -> 0.35 GB; profiles/).  The instances whose mode arrives at launch (adaptive
// anti-aliasing, the prepass) lose 7 % at -O1 and stay in hxr_kernels_mesh.hip.
#include <hip/hip_runtime.h>
#include "hexray_hip.h"
#define HXR_RENDER_KERNEL_ONLY
#define HXR_TRI_DECIDE_BITWISE   // triangle_decide: with the end-of-round-4 flags this unit is faster with the bitwise spelling (6.40 -> 6.345 ms; it had been the other way round)
#define HXR_CONST_TABLES   // round 4: the kd and hot tables of the packet walk are read through the constant address space in this unit (headline 6.71 -> 6.62 ms; the other units do not gain)
#include "hxr_kernels.hip.h"
#include "hxr_variants.h"

HXR_VARIANTS_MESH_INPLACE_MC_THROUGHPUT(HXR_INSTANTIATE_MONTE_CARLO)   // the headline kernel; the three-waves instances live in hxr_kernels_mesh_lat.hip
