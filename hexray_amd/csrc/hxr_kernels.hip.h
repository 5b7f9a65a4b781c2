// HIP kernels of the render module (gfx950, wave64).  FP64 geometry, FP32 colour, no MFMA.
//
// Parallel decomposition: one camera sample per lane; a wavefront is a "ray packet" of 64 samples
// taken from one 8x8 (Morton-ordered) pixel block, so the lanes of a wave walk the same nodes, the
// same kd subtrees and the same shader.  Waves are persistent: each one pulls packet numbers from an
// atomic dispenser until the frame is exhausted (the GPU analogue of the reference's atomic bucket
// cursor, src/main.cpp:273-275).
//
// The reference's mutually recursive raytrace() <-> Shader::computeColor() (src/main.cpp:105-113,
// src/shading.cpp:159-232,315-324) is flattened: because every shader is linear in the colours its
// secondary rays return, a ray carries the product of the blend / reflectance factors on its path (its
// "weight") and adds weight x leaf-colour into the sample's accumulator.  Pending secondary rays wait
// on a small per-lane stack; Layered shaders are expanded through a per-lane shader work list.  The
// float summation order therefore differs from the reference's (rounding-level, ~1e-7); every
// geometric decision (which node, which triangle, which texel, which cell) follows the reference's
// arithmetic expression by expression.
#pragma once
#include <hip/hip_runtime.h>
#include "hxr_device_scene.h"
#include "hxr_rng.hip.h"

#define HXR_INF 1e99                  // src/constants.h:29
#define HXR_PI 3.141592653589793238   // src/constants.h:28
#define HXR_RF_GI_DIFFUSE 2u          // src/vector.h:176
#define HXR_ACCUM_SCALE 4294967296.0f   // 2^32: one unit of the fixed-point colour sums = 2.3e-10, range +-2.1e9
#define HXR_RAY_STACK 12              // pending secondary rays per lane (checked against the scene at create time)
#define HXR_SHADER_STACK 8            // pending (shader, weight) pairs while a hit is being shaded
#define HXR_DEV __device__ __forceinline__
#define HXR_DEVX __device__ __forceinline__

// Kernel specialisation.  The register allocation of a kernel is set by its hungriest code path, whether or not
// a scene ever runs it (the heightfield marcher alone wants 96 VGPRs, the packet kd walk 60), so the render
// kernel is compiled for the feature sets scenes actually have and the launcher picks the smallest fit.
enum : unsigned {
	HXR_F_MESH = 1u, HXR_F_HEIGHTFIELD = 2u, HXR_F_SPHERE = 4u, HXR_F_CUBE = 8u,   // geometry kinds besides Plane
	HXR_F_GI = 16u,                                                                 // pathtrace() instead of raytrace()
	HXR_F_EXTRA = 32u,                                                              // CSG geometry, glossy Reflection, anaglyph camera
	HXR_F_RECT = 64u,                                                               // RectLight (every light is a PointLight otherwise)
	HXR_F_REFRACT = 128u,                                                           // glass and water: Refraction shader in the Whitted work list, Fresnel and Bumps textures
	HXR_F_ALL = 255u,
	// Whitted only: a hit's Reflection / Refraction rays are appended to a global queue (one ballot + prefix count + atomic per
	// group of lanes) instead of waiting on a per-lane stack; hxr_continue_kernel traces them, re-packed 64 to a wave.  The
	// render kernel then holds camera rays and their shadow rays only -- packets of equal length, no per-lane ray stack.
	HXR_F_DEFER = 256u,
	// Mesh::intersect walked by every lane for itself (vector loads, a per-lane stack in LDS) instead of by the wave as a packet:
	// for rays that have nothing in common -- the continuation pass re-packs secondary rays of unrelated pixels into one wave --
	// a packet walk visits the union of 64 paths one after the other, a per-lane walk visits them side by side.
	HXR_F_LANEWALK = 512u,
	// Whitted, in place: no hit of the scene spawns more than ONE secondary ray (a Layered shader with one Reflection or Refraction
	// layer, ...), so at most one ray is ever pending per lane.  It waits in LDS (18 dwords per lane, a row per dword) instead of on
	// the 12-entry per-lane stack, which is a dynamically indexed array and therefore lives in scratch memory.
	HXR_F_STACK1 = 1024u,
	// CSG operations whose operands are operations (catch-all variants only).  The recursion of CSGBase::intersect is unrolled HXR_CSG_DEPTH
	// levels deep inside two functions that are CALLED, not inlined: inlined into the flattened integrator they cost every catch-all
	// kernel its register allocation (scratch 1 KB -> 15-29 KB per lane, kdtree_test 11.7 -> 80 ms), and as calls in the ordinary
	// catch-alls they still cost 4x (the kernel then lives by the call ABI) -- so only scenes that nest CSG run these variants.
	HXR_F_CSGNEST = 2048u,
	// Whitted, in place: the first HXR_LDS_RAY_SLOTS entries of the per-lane ray stack live in LDS (18 dwords per entry and lane, a row of
	// 64 lanes per dword), the rest -- reached only by hits stacked deeper than that -- in the scratch array as before.  For the variants
	// compiled for two waves per SIMD, which leave the LDS of a CU idle (the heightfield variant: every water hit pushes a reflected and
	// a refracted ray, 2.4 GB of scratch writes per frame before this).
	HXR_F_LDSSTACK = 4096u,
	// Leaf loop of the packet walk: the hot record of triangle i + 1 is touched (one dword of each of its two cache lines, into two scratch scalar registers)
	// as soon as triangle i's products are formed, so that the next iteration's scalar loads hit the scalar cache instead of L2.  For the instances that run
	// launches a few packets deep (shards of a multi-GPU frame, first frames): there a packet's dependent chain is what the launch waits for and 28 % of a lone
	// wave's chain is the wait for hot records (profiles/r04_chain_*.txt): slowest 8-way rank 1.51 -> 1.36 ms, the 3-waves instance 7.04 -> 6.72 ms per whole
	// frame.  The throughput instance (four waves per SIMD) LOSES 6 % by it (128 spilled registers instead of 65) and stays without.
	HXR_F_TOUCH = 8192u,
	// Replicated packets (round 4; uninstrumented latency instances only).  A packet with few rays -- the four samples of a sample-split part, the single pixel
	// of a split chunk, the four taps of a flagged AA pixel -- is one long dependent chain on a wave that is mostly idle lanes.  Such a packet is traced with
	// every ray held by R adjacent lanes (R = 16 for up to 4 rays, 4 for up to 16): identical state, identical control flow, so nothing else changes -- except
	// in a kd leaf, where replica r of a ray tests triangles r, r + R, ... (its own vector loads of the hot records) and one prefix-minimum over the R
	// lanes reproduces the sequential loop's accept rule exactly (leaf_query_replicated).  A leaf of up to 16 triangles is one round instead of 16 dependent
	// iterations.
	HXR_F_REPL = 16384u
};
#ifdef HXR_LEAF_TOUCH   // every Whitted kernel of the unit (the path tracers lose by it)
#define HXR_TOUCH_ALL true
#else
#define HXR_TOUCH_ALL false
#endif
#ifdef HXR_REPLICATE    // ... every uninstrumented kernel of the unit
#define HXR_REPL_ALL true
#else
#define HXR_REPL_ALL false
#endif
template <bool COUNT, unsigned F>
constexpr bool replicating = !COUNT && ((F & HXR_F_REPL) != 0 || HXR_REPL_ALL) && (F & (HXR_F_GI | HXR_F_DEFER | HXR_F_LANEWALK)) == 0;
#ifndef HXR_LDS_RAY_SLOTS
#define HXR_LDS_RAY_SLOTS 2   // 2 x 18 dwords x 256 lanes = 36 KB per block: three blocks per CU (3 waves per SIMD) fit the 160 KB; 4 slots (72 KB) allow two
#endif

// counters (COUNT builds only); order = hxr_counters
enum {
	HC_SAMPLES, HC_RAYCASTS, HC_SHADOW, HC_NODE, HC_BBOX, HC_KD, HC_LEAFIDX, HC_TRI, HC_MESHHIT, HC_TEX, HC_ENV,
	HC_HFCALL, HC_HFHIGH, HC_HFHEIGHT, HC_HFTRI, HC_HFNORM, HC_AA, HC_BOXMISMATCH, HC_NUM
};

// lanes of the wave for which `pred` holds: the compare's own mask (HIP's __ballot() goes through an int and a second compare)
#define hxr_ballot(pred) ((unsigned long long)__builtin_amdgcn_ballot_w64(pred))

struct d3 { double x, y, z; };
struct f3 { float x, y, z; };

HXR_DEV d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
HXR_DEV d3 ld3(const double* p) { return d3{p[0], p[1], p[2]}; }
// Pointers the kernels find INSIDE records in HBM (a mesh's vertex arrays, a heightfield's tables, a bitmap's texels) are generic
// to the compiler, which then emits flat_load (both memory counters, LDS aperture check).  They all point into HBM: say so.
#define HXR_GLOBAL __attribute__((address_space(1)))
template <class T> HXR_DEV const HXR_GLOBAL T* in_hbm(const T* p) { return (const HXR_GLOBAL T*)p; }
HXR_DEV d3 ld3(const HXR_GLOBAL double* p) { return d3{p[0], p[1], p[2]}; }
// ... and tables that nothing writes while a kernel runs can be read through the CONSTANT address space: a wave-uniform load from it is a scalar load
// whatever stands between it and the kernel's entry (an inline asm, a store the compiler cannot tell apart)
#define HXR_CONSTANT __attribute__((address_space(4)))
HXR_DEV d3 ld3(const HXR_CONSTANT double* p) { return d3{p[0], p[1], p[2]}; }
template <class T>
HXR_DEV T ld_const(const HXR_CONSTANT T* p)   // a whole record (a multiple of 4 bytes) from the constant address space
{
	T r;
	const HXR_CONSTANT uint32_t* w = (const HXR_CONSTANT uint32_t*)p;
	uint32_t* o = (uint32_t*)&r;
#pragma unroll
	for (unsigned i = 0; i < sizeof(T) / 4; i++) o[i] = w[i];
	return r;
}
HXR_DEV d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
HXR_DEV d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
HXR_DEV d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
HXR_DEV d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
HXR_DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HXR_DEV d3 cross(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
HXR_DEV double len(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
HXR_DEV double len2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
HXR_DEV d3 unit(d3 a) { return a * (1.0 / len(a)); }                       // Vector::normalize, src/vector.h:79-83
// normalize(), src/vector.h:160-165: `l = length(a); if (fabs(l - 1) < 1e-6) return a; return a * (1 / l);`
// ON_SQUARE decides it on l^2 -- (1 - 1e-6)^2 < l^2 < (1 + 1e-6)^2 is the same interval -- and pays for the square root and the division
// only when some lane of the wave leaves it.  Every ray direction through a rotation-only or translate-only node lands inside (smallpt's
// nodes never take the slow path: -1.7 %); where nodes scale (the meshes of meshes.hexray) the extra test and ballot are pure cost
// (+1.8 % on the headline frame), so the variants choose (fast_transforms).
template <bool ON_SQUARE = false>
HXR_DEV d3 unit_lazy(d3 a)
{
	if constexpr (ON_SQUARE) {
		const double l2 = len2(a);
		const bool keep = l2 > 0.999998000001 && l2 < 1.000002000001;
		if (!hxr_ballot(!keep)) return a;
		const double l = sqrt(l2);
		return keep ? a : a * (1 / l);
	} else {
		const double l = len(a);
		if (fabs(l - 1.0) < 1e-6) return a;
		return a * (1 / l);
	}
}
// The path tracers and the variants without kd meshes take the two shortcuts of Node::intersect's transforms -- unit_lazy on the square,
// and no matrix products for translate-only nodes (mul_identity) -- which pay where most nodes are unscaled spheres and planes (smallpt
// -2 % and -1.7 %, heightfield -1.7 %, cornell box -2.4 %) and cost the Whitted mesh kernels 3 % + 1.8 % in registers and branches even
// when no node of the scene qualifies (measured against the builds before / after each, on one box: 8.85 -> 8.99 -> 9.28 ms).
template <unsigned F>
constexpr bool fast_transforms = (F & HXR_F_MESH) == 0 || (F & HXR_F_GI) != 0;
HXR_DEV double dist3(d3 a, d3 b) { return sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z)); }
HXR_DEV d3 faceforward(d3 ray, d3 n) { return dot(ray, n) < 0 ? n : -n; }  // src/vector.h:179-182
HXR_DEV d3 reflect3(d3 i, d3 n) { return n * (2 * dot(-i, n)) + i; }       // src/vector.h:184-187
HXR_DEV d3 mulm(d3 v, const double* m)                                     // v * M, src/matrix.h:53-60
{
	return d3{v.x * m[0] + v.y * m[3] + v.z * m[6], v.x * m[1] + v.y * m[4] + v.z * m[7], v.x * m[2] + v.y * m[5] + v.z * m[8]};
}
// v * I evaluated the way mulm evaluates it: x*1 + y*0 + z*0 = x + 0.0 (only the sign of a zero can change): nodes without rotation or
// scaling -- smallpt's spheres, every floor plane -- skip the 27 multiplications of Node::intersect's three matrix products, bit for bit
HXR_DEV d3 mul_identity(d3 v) { return d3{v.x + 0.0, v.y + 0.0, v.z + 0.0}; }
HXR_DEV double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

HXR_DEV f3 mkf(float x, float y, float z) { return f3{x, y, z}; }
HXR_DEV f3 ldf(const float* p) { return f3{p[0], p[1], p[2]}; }
HXR_DEV f3 ldf(const HXR_GLOBAL float* p) { return f3{p[0], p[1], p[2]}; }
HXR_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
HXR_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
HXR_DEV f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
HXR_DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
HXR_DEV f3 fdiv(f3 a, float d)                                             // Color / float, src/color.h:195-200
{
	if (d == 1) return a;
	float m = 1.0f / d;
	return f3{a.x * m, a.y * m, a.z * m};
}
HXR_DEV float intensity(f3 a) { return (a.x + a.y + a.z) / 3; }
HXR_DEV bool is_zero(f3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }

// Profiling builds (-DHXR_PHASE_TIMERS, tools/phase_profile.sh): every wave charges the shader-clock time between
// two `lap(phase)` calls to the phase that just ended; the totals land in hxr_counters.reserved_[phase].
#if !defined(HXR_PHASE_SET) || HXR_PHASE_SET == 1   // traversal in detail
enum { PH_DISPENSE, PH_GEN, PH_QUERY, PH_MESH, PH_LEAF, PH_SHADE, PH_NUM, PH_REDUCE = PH_GEN, PH_SURFACE = PH_SHADE, PH_ENDS = PH_SHADE, PH_MAP = PH_GEN, PH_LENS = PH_GEN };
#elif HXR_PHASE_SET == 2                            // everything around the traversal in detail
enum { PH_DISPENSE, PH_GEN, PH_REDUCE, PH_QUERY, PH_ENDS, PH_SHADE, PH_NUM, PH_MESH = PH_QUERY, PH_LEAF = PH_QUERY, PH_SURFACE = PH_SHADE, PH_MAP = PH_GEN, PH_LENS = PH_GEN };
#elif HXR_PHASE_SET == 3                            // sample generation in detail
enum { PH_DISPENSE, PH_MAP, PH_LENS, PH_GEN, PH_REDUCE, PH_QUERY, PH_NUM, PH_MESH = PH_QUERY, PH_LEAF = PH_QUERY, PH_SURFACE = PH_QUERY, PH_ENDS = PH_QUERY, PH_SHADE = PH_QUERY };
#else                                               // the dependent chain of the kd walk (round 4): what a step computes against what it waits for
// PH_DISPENSE = everything outside Mesh::intersect; inside it: PH4_KD_COMPUTE the children's box tests, ballots, pushes; PH4_KD_WAIT what is left of the
// wait for the children's records (in flight during the tests) / for a popped node's record and box; PH4_LEAF_WAIT the wait for a triangle's hot record;
// PH4_LEAF_COMPUTE its test; PH4_POP looking for the next pending visit
enum { PH_DISPENSE, PH4_KD_COMPUTE, PH4_KD_WAIT, PH4_LEAF_WAIT, PH4_LEAF_COMPUTE, PH4_POP, PH_NUM, PH_GEN = PH_DISPENSE, PH_QUERY = PH_DISPENSE, PH_MESH = PH4_POP,
       PH_LEAF = PH4_LEAF_COMPUTE, PH_SHADE = PH_DISPENSE, PH_REDUCE = PH_DISPENSE, PH_SURFACE = PH_DISPENSE, PH_ENDS = PH_DISPENSE, PH_MAP = PH_DISPENSE, PH_LENS = PH_DISPENSE };
#define HXR_CHAIN_LAPS 1
#endif
// HXR_CHAIN_LAP(tl, PHASE, words...): (chain profile only) first make the listed scalar words arrive -- an empty asm that takes them as inputs: the compiler
// must wait for their loads before it, which neither s_waitcnt nor a scheduling barrier achieves for constant-address-space loads -- then charge the time
// since the last lap to PHASE
#ifdef HXR_CHAIN_LAPS
#define HXR_CHAIN_ARRIVE2(a, b) asm volatile("" ::"s"(a), "s"(b))
#define HXR_CHAIN_LAP(tl, PHASE) (tl).template lap<PHASE>()
#else
#define HXR_CHAIN_ARRIVE2(a, b) ((void)0)
#define HXR_CHAIN_LAP(tl, PHASE) ((void)0)
#endif

template <bool COUNT>
struct Tally {
	uint32_t c[COUNT ? HC_NUM : 1];
	HXR_DEV void add(int which, uint32_t n = 1) { if (COUNT) c[which] += n; }
#ifdef HXR_PHASE_TIMERS
	// One set of accumulators per WAVE, in LDS, updated by the first active lane: a per-lane copy would only advance
	// while its lane is active and would charge the time a lane sits masked off to whatever phase it wakes up in.
	volatile unsigned long long* ph;   // PH_NUM phase sums + [PH_NUM] = time of the last lap
	HXR_DEV void clear() { for (int i = 0; i < (COUNT ? HC_NUM : 1); i++) c[i] = 0; }
	HXR_DEV void attach(unsigned long long* lds)
	{
		ph = lds;
		if ((threadIdx.x & 63) == 0) {
			for (int i = 0; i < PH_NUM; i++) lds[i] = 0;
			lds[PH_NUM] = __builtin_readcyclecounter();
		}
	}
#ifdef HXR_PACKET_CLOCKS
	template <int PHASE> HXR_DEV void lap() {}
#else
	template <int PHASE> HXR_DEV void lap()
	{
		const unsigned long long now = __builtin_readcyclecounter();
		const int me = threadIdx.x & 63;
		if (__builtin_amdgcn_readfirstlane(me) == me) {
			ph[PHASE] += now - ph[PH_NUM];
			ph[PH_NUM] = now;
		}
	}
#endif
#else
	HXR_DEV void clear() { for (int i = 0; i < (COUNT ? HC_NUM : 1); i++) c[i] = 0; }
	template <int PHASE> HXR_DEV void lap() {}
#endif
};

// ------------------------------------------------------------------------------------------------
// boxes
// ------------------------------------------------------------------------------------------------
struct Box { d3 lo, hi; };

HXR_DEV bool box_holds(const Box& b, d3 p)   // BBox::inside, src/bbox.h:78-85
{
	return b.lo.x - 1e-6 <= p.x && p.x <= b.hi.x + 1e-6 && b.lo.y - 1e-6 <= p.y && p.y <= b.hi.y + 1e-6 &&
	       b.lo.z - 1e-6 <= p.z && p.z <= b.hi.z + 1e-6;
}

// One axis of BBox::testIntersect (src/bbox.h:96-139), evaluated without branches: lanes of a wave leave
// the reference's loop at different points, so every exit is computed as a predicate instead.
//   rej : this axis proves a miss (origin beyond the slab, pointing away)           -> `return false`
//   hit : one of the two walls of this axis is crossed inside the box               -> `return true`
// The near wall behind the origin (t < 0) skips the far wall as well, like the reference's `continue`.
HXR_DEV void box_axis(double oa, double da, double ia, double loa, double hia, double ou, double du, double lou, double hiu, double ov,
                      double dv, double lov, double hiv, bool& rej, bool& hit)
{
	rej = (da < 0 && oa < loa) || (da > 0 && oa > hia);
	bool usable = !(fabs(da) < 1e-9);
	double t0 = (loa - oa) * ia, t1 = (hia - oa) * ia;
	bool near_ok = usable && !(t0 < 0);
	bool far_ok = near_ok && !(t1 < 0);
	double pu0 = ou + du * t0, pv0 = ov + dv * t0;
	double pu1 = ou + du * t1, pv1 = ov + dv * t1;
	bool h0 = near_ok && lou <= pu0 && pu0 <= hiu && lov <= pv0 && pv0 <= hiv;
	bool h1 = far_ok && lou <= pu1 && pu1 <= hiu && lov <= pv1 && pv1 <= hiv;
	hit = h0 || h1;
}

// BBox::testIntersect, literally; `inv` = (1/d.x, 1/d.y, 1/d.z), the reference's per-call `mul` hoisted out.
// Sequential meaning of the reference loop: inside || (!rejX && (hitX || (!rejY && (hitY || (!rejZ && hitZ))))).
HXR_DEV bool box_hit_literal(const Box& b, d3 o, d3 d, d3 inv)
{
	bool rx, hx, ry, hy, rz, hz;
	box_axis(o.x, d.x, inv.x, b.lo.x, b.hi.x, o.y, d.y, b.lo.y, b.hi.y, o.z, d.z, b.lo.z, b.hi.z, rx, hx);
	box_axis(o.y, d.y, inv.y, b.lo.y, b.hi.y, o.x, d.x, b.lo.x, b.hi.x, o.z, d.z, b.lo.z, b.hi.z, ry, hy);
	box_axis(o.z, d.z, inv.z, b.lo.z, b.hi.z, o.x, d.x, b.lo.x, b.hi.x, o.y, d.y, b.lo.y, b.hi.y, rz, hz);
	return box_holds(b, o) || (!rx && (hx || (!ry && (hy || (!rz && hz)))));
}

// BBox::testIntersect as a slab test.  The reference answers "does the ray reach the box?" by trying the
// six walls one after the other (src/bbox.h:96-139): with the origin outside, the ray reaches the box iff
// it ENTERS through a wall it faces (t >= 0) inside that wall's rectangle -- which is exactly
// max_a tnear_a <= min_a tfar_a with tfar >= 0, built from the same products (wall - o) * (1/d) the
// reference forms.  The 1e-6-inflated inside() shortcut (src/bbox.h:95) is kept as is.  The two procedures
// can only disagree when the entry point lies within one rounding error of a box edge; triangles that
// close to a kd cell's edge are also listed in the neighbouring cell, so the reported hit is unaffected.
// The instrumented kernel build evaluates BOTH and counts disagreements (hxr_counters::box_test_mismatch);
// the GPU tests require that count to be zero on the config scenes.
HXR_DEV bool box_hit(const Box& b, d3 o, d3 d, d3 inv)
{
	double tx0 = (b.lo.x - o.x) * inv.x, tx1 = (b.hi.x - o.x) * inv.x;
	double ty0 = (b.lo.y - o.y) * inv.y, ty1 = (b.hi.y - o.y) * inv.y;
	double tz0 = (b.lo.z - o.z) * inv.z, tz1 = (b.hi.z - o.z) * inv.z;
	double tn = fmax(fmax(fmin(tx0, tx1), fmin(ty0, ty1)), fmin(tz0, tz1));
	double tf = fmin(fmin(fmax(tx0, tx1), fmax(ty0, ty1)), fmax(tz0, tz1));
	bool r = tn <= tf && tf >= 0;
	if (hxr_ballot(!r)) r = r || box_holds(b, o);   // the inflated inside() shortcut only matters for lanes the slabs reject
	return r;
}

template <bool COUNT>
HXR_DEV bool box_test(const Box& b, d3 o, d3 d, d3 inv, Tally<COUNT>& tl)
{
	tl.add(HC_BBOX);
	bool r = box_hit(b, o, d, inv);
	if (COUNT && r != box_hit_literal(b, o, d, inv)) tl.add(HC_BOXMISMATCH);
	return r;
}

// one axis of BBox::closestIntersectionPeriphery (src/bbox.h:145-167); false = the ray cannot hit the box
HXR_DEV bool box_near_axis(double oa, double da, double loa, double hia, double ou, double du, double lou, double hiu, double ov,
                           double dv, double lov, double hiv, double& best)
{
	if ((da < 0 && oa < loa) || (da > 0 && oa > hia)) return false;
	if (fabs(da) < 1e-9) return true;
	double mul = 1 / da;
#pragma unroll
	for (int j = 0; j < 2; j++) {
		double t = ((j ? hia : loa) - oa) * mul;
		if (t < 0) continue;
		double pu = ou + du * t;
		if (lou <= pu && pu <= hiu) {
			double pv = ov + dv * t;
			if (lov <= pv && pv <= hiv) best = fmin(best, t);
		}
	}
	return true;
}

HXR_DEV double box_nearest(const Box& b, d3 o, d3 d)   // BBox::closestIntersection, src/bbox.h:169-174
{
	if (box_holds(b, o)) return 0;
	double best = HXR_INF;
	if (!box_near_axis(o.x, d.x, b.lo.x, b.hi.x, o.y, d.y, b.lo.y, b.hi.y, o.z, d.z, b.lo.z, b.hi.z, best)) return HXR_INF;
	if (!box_near_axis(o.y, d.y, b.lo.y, b.hi.y, o.x, d.x, b.lo.x, b.hi.x, o.z, d.z, b.lo.z, b.hi.z, best)) return HXR_INF;
	if (!box_near_axis(o.z, d.z, b.lo.z, b.hi.z, o.x, d.x, b.lo.x, b.hi.x, o.y, d.y, b.lo.y, b.hi.y, best)) return HXR_INF;
	return best;
}

// ------------------------------------------------------------------------------------------------
// geometry queries.  A query reports only what identifies the hit (distance along the object-space
// ray, triangle + barycentrics, cube face); attributes are derived later for the winning node only.
// ------------------------------------------------------------------------------------------------
struct GeomHit {
	double t;        // object-space distance (info.dist inside geometry code)
	int tri;         // mesh: index of the accepted triangle's hot record (scene-wide: HxrSceneDev::hot / hot_ids) / cube face (axis*2 + side)
	// The barycentrics of a mesh hit are not carried: surface_of() forms them again for the ONE hit a ray keeps, with the expressions
	// of the test (tri_products), instead of every candidate dragging two doubles through the node scan in two copies (8 VGPRs).
};

HXR_DEV bool plane_query(const double* p, d3 o, d3 d, GeomHit& g)   // Plane::intersect, src/geometry.cpp:31-50
{
	double py = p[0], limit = p[1];
	if (o.y > py && d.y >= 0) return false;
	if (o.y < py && d.y <= 0) return false;
	double t = (py - o.y) / d.y;
	d3 ip = o + d * t;
	if (fabs(ip.x) > limit || fabs(ip.z) > limit) return false;
	g.t = t;
	return true;
}

HXR_DEV bool sphere_query(const double* p, d3 o, d3 d, GeomHit& g)   // Sphere::intersect, src/geometry.cpp:52-85
{
	d3 O = ld3(p);
	double R = p[3];
	double A = len2(d);
	d3 H = o - O;
	double B = 2 * dot(d, H);
	double C = len2(H) - R * R;
	double D = B * B - 4 * A * C;
	if (D < 0) return false;
	double sD = sqrt(D);
	double p1 = (-B - sD) / (2 * A), p2 = (-B + sD) / (2 * A);
	if (p2 < 0) return false;
	g.t = (p1 < 0) ? p2 : p1;
	return true;
}

HXR_DEV bool cube_query(const double* p, d3 o, d3 d, GeomHit& g)   // Cube::intersect, src/geometry.cpp:87-147
{
	d3 O = ld3(p);
	double half = p[3];
	double best = HXR_INF;
	int face = -1;
#pragma unroll
	for (int a = 0; a < 3; a++)
#pragma unroll
		for (int s = 0; s < 2; s++) {
			double oa = comp(o, a), da = comp(d, a);
			double target = comp(O, a) + (s ? half : -half);
			if (fabs(da) < 1e-9) continue;
			if (oa < target && da < 0) continue;
			if (oa > target && da > 0) continue;
			double t = (target - oa) / da;
			if (!(t < best)) continue;
			d3 ip = o + d * t;
			if (!(ip.x > O.x - half - 1e-6 && ip.x < O.x + half + 1e-6)) continue;
			if (!(ip.y > O.y - half - 1e-6 && ip.y < O.y + half + 1e-6)) continue;
			if (!(ip.z > O.z - half - 1e-6 && ip.z < O.z + half + 1e-6)) continue;
			best = t;
			face = a * 2 + s;
		}
	g.t = best;
	g.tri = face;
	return face >= 0;
}

// ------------------------------------------------------------------------------------------------
// CSG over analytic operands (CSGUnion / CSGInter / CSGDiff, src/geometry.h:109-136, src/geometry.cpp:149-194)
// ------------------------------------------------------------------------------------------------
HXR_DEV bool analytic_query(const hxr_geometry& G, d3 o, d3 d, GeomHit& g)
{
	switch (G.type) {
		case HXR_GEOM_PLANE: return plane_query(G.p, o, d, g);
		case HXR_GEOM_SPHERE: return sphere_query(G.p, o, d, g);
		case HXR_GEOM_CUBE: return cube_query(G.p, o, d, g);
		default: return false;
	}
}

// object-space normal of an analytic primitive at the hit described by (o, d, g); also u, v
HXR_DEV d3 analytic_normal(const hxr_geometry& G, d3 o, d3 d, const GeomHit& g, double& u, double& v)
{
	d3 ip = o + d * g.t;
	if (G.type == HXR_GEOM_PLANE) { u = ip.x; v = ip.z; return mk3(0, (o.y > G.p[0]) ? 1 : -1, 0); }
	if (G.type == HXR_GEOM_SPHERE) {
		d3 n = unit(ip - ld3(G.p));
		double vv = asin(n.y), uu = atan2(n.z, n.x);
		vv = -(vv / HXR_PI + 0.5f);
		uu = uu / (2 * HXR_PI) + 0.5f;
		if (G.p[4] != 1) { uu *= G.p[4]; vv *= G.p[4]; }
		u = uu; v = vv;
		return n;
	}
	int a = g.tri >> 1;
	double sg = (g.tri & 1) ? 1 : -1;
	if (a == 0) { u = ip.y; v = ip.z; } else if (a == 1) { u = ip.x; v = ip.z; } else { u = ip.x; v = ip.y; }
	return mk3(a == 0 ? sg : 0, a == 1 ? sg : 0, a == 2 ? sg : 0);
}

#define HXR_CSG_MAX 30   // findAllIntersections re-shoots at most 30 times per operand (src/geometry.cpp:155)
#define HXR_CSG_DEPTH 3  // operations nested at most this deep (an operation over primitives = 1); validate() refuses deeper trees

HXR_DEV bool is_csg(int type) { return type >= HXR_GEOM_CSG_UNION && type <= HXR_GEOM_CSG_DIFF; }

// CSGBase::intersect is generic over Geometry* (src/geometry.cpp:149-194): an operand is an analytic primitive or another operation.
// LEVEL = how many more levels of operations may still appear below (compile-time recursion: the GPU has no call stack to spare).
template <int LEVEL>
HXR_DEV bool csg_query(const HxrSceneDev& S, const hxr_geometry& C, d3 o, d3 d, GeomHit& g, d3* ip_out);
template <int LEVEL>
HXR_DEV void csg_rehit(const HxrSceneDev& S, const hxr_geometry& G, d3 o, d3 d, int k, d3& ip, d3& norm, double& u, double& v, int& prim_type);

// findAllIntersections (src/geometry.cpp:149-165): distances (from the original origin) of the successive hits of operand G
template <int LEVEL>
HXR_DEV int csg_collect(const HxrSceneDev& S, const hxr_geometry& G, d3 o, d3 d, double* dist)
{
	d3 start = o;
	int n = 0;
	for (int k = 0; k < HXR_CSG_MAX; k++) {
		GeomHit g;
		g.t = HXR_INF;
		g.tri = -1;
		d3 ip;
		bool nested = false;
		if constexpr (LEVEL > 0) nested = is_csg(G.type);
		if (nested) {
			if constexpr (LEVEL > 0)
				if (!csg_query<LEVEL - 1>(S, G, start, d, g, &ip)) break;   // the operand's own IntersectionInfo::ip, not start + d * dist
		} else {
			if (!analytic_query(G, start, d, g)) break;
			ip = start + d * g.t;
		}
		dist[n++] = dist3(o, ip);
		start = ip + d * 1e-6;
	}
	return n;
}

// the k-th hit of findAllIntersections again, this time with its attributes (and the kind of primitive that was finally hit)
template <int LEVEL>
HXR_DEV void csg_rehit(const HxrSceneDev& S, const hxr_geometry& G, d3 o, d3 d, int k, d3& ip, d3& norm, double& u, double& v, int& prim_type)
{
	d3 start = o;
	GeomHit g{HXR_INF, -1};   // defined even for k < 0 (an inner query that found nothing hands on tri = -1): no attribute below is read from indeterminate bits
	bool nested = false;
	if constexpr (LEVEL > 0) nested = is_csg(G.type);
	for (int i = 0; i <= k; i++) {
		g.t = HXR_INF;
		g.tri = -1;
		if (nested) {
			if constexpr (LEVEL > 0) csg_query<LEVEL - 1>(S, G, start, d, g, &ip);
		} else {
			analytic_query(G, start, d, g);
			ip = start + d * g.t;
		}
		if (i < k) start = ip + d * 1e-6;
	}
	if (nested) {
		if constexpr (LEVEL > 0) {
			csg_rehit<LEVEL - 1>(S, S.geometries[(g.tri & 1) ? G.right : G.left], start, d, g.tri >> 1, ip, norm, u, v, prim_type);
			norm = faceforward(d, norm);   // the inner operation's own faceforward, src/geometry.cpp:187
		}
	} else {
		norm = analytic_normal(G, start, d, g, u, v);
		prim_type = G.type;
	}
}

// CSGBase::intersect (src/geometry.cpp:167-194).  Reports the hit as g.t = distance from the origin and
// g.tri = operand (bit 0) | ordinal of the operand's hit (bits 1..); surface_of() re-derives its attributes.
// ip_out (nested use): the hit point as the chosen operand computed it.
template <int LEVEL>
HXR_DEV bool csg_query(const HxrSceneDev& S, const hxr_geometry& C, d3 o, d3 d, GeomHit& g, d3* ip_out)
{
	const hxr_geometry& GL = S.geometries[C.left];
	const hxr_geometry& GR = S.geometries[C.right];
	double dl[HXR_CSG_MAX], dr[HXR_CSG_MAX];
	int nl = csg_collect<LEVEL>(S, GL, o, d, dl), nr = csg_collect<LEVEL>(S, GR, o, d, dr);
	bool inA = nl & 1, inB = nr & 1;
	auto inside = [&](bool a, bool b) { return C.type == HXR_GEOM_CSG_UNION ? (a || b) : C.type == HXR_GEOM_CSG_INTER ? (a && b) : (a && !b); };
	const bool initial = inside(inA, inB);
	// merge the two ascending-by-construction lists in distance order (std::sort of the concatenation, left
	// entries first: on equal distances the earlier element of the concatenation is not guaranteed to stay first
	// by std::sort; a stable merge is the oracle's choice as well)
	int il = 0, ir = 0;
	while (il < nl || ir < nr) {
		bool take_left = ir >= nr || (il < nl && dl[il] <= dr[ir]);
		if (take_left) inA = !inA; else inB = !inB;
		if (inside(inA, inB) != initial) {
			g.t = take_left ? dl[il] : dr[ir];
			g.tri = (take_left ? 0 : 1) | ((take_left ? il : ir) << 1);
			if (ip_out) {
				d3 n;
				double u, v;
				int prim;
				csg_rehit<LEVEL>(S, take_left ? GL : GR, o, d, take_left ? il : ir, *ip_out, n, u, v, prim);
			}
			return true;
		}
		if (take_left) il++; else ir++;
	}
	return false;
}

// the entry points of the HXR_F_CSGNEST variants: the whole unrolled recursion behind ONE call each
static __device__ __attribute__((noinline)) bool csg_query_nested(const HxrSceneDev& S, const hxr_geometry& C, d3 o, d3 d, GeomHit& g)
{
	return csg_query<HXR_CSG_DEPTH - 1>(S, C, o, d, g, nullptr);
}
static __device__ __attribute__((noinline)) void csg_rehit_nested(const HxrSceneDev& S, const hxr_geometry& G, d3 o, d3 d, int k, d3& ip, d3& norm, double& u, double& v, int& prim_type)
{
	csg_rehit<HXR_CSG_DEPTH - 1>(S, G, o, d, k, ip, norm, u, v, prim_type);
}
// ... and of every other variant: operands are primitives, everything inline as before
template <unsigned F>
HXR_DEV bool csg_query_any(const HxrSceneDev& S, const hxr_geometry& C, d3 o, d3 d, GeomHit& g)
{
	if constexpr ((F & HXR_F_CSGNEST) != 0) return csg_query_nested(S, C, o, d, g);
	else return csg_query<0>(S, C, o, d, g, nullptr);
}
template <unsigned F>
HXR_DEV void csg_rehit_any(const HxrSceneDev& S, const hxr_geometry& G, d3 o, d3 d, int k, d3& ip, d3& norm, double& u, double& v, int& prim_type)
{
	if constexpr ((F & HXR_F_CSGNEST) != 0) csg_rehit_nested(S, G, o, d, k, ip, norm, u, v, prim_type);
	else csg_rehit<0>(S, G, o, d, k, ip, norm, u, v, prim_type);
}

// Per-wave traversal stack in LDS: (node, lanes on whose behalf it is pending).
#define HXR_KD_STACK 200     // at most three entries per level of a depth-66 tree
struct WaveStack {
	int32_t node[HXR_KD_STACK];
	uint32_t mask_lo[HXR_KD_STACK], mask_hi[HXR_KD_STACK];   // lanes that enter the node when its turn comes (if still searching)
	uint32_t due_lo[HXR_KD_STACK], due_hi[HXR_KD_STACK];     // instrumented builds: lanes whose BBox::testIntersect of this node is counted then
};
// Per-lane traversal stack of the per-lane walk (mesh_query_lane): a column of 32-bit node numbers per lane in LDS
struct LaneStack {
	uint32_t* rows;     // this wave's `entries` rows x 64 lanes
	int entries;        // the deepest tree of the scene + 2
	uint32_t* overflow; // raised if a push would not fit (cannot happen: sized from the trees at launch)
};
// what a traversal may use: the packet walk's per-wave stack, or the per-lane walk's columns (HXR_F_LANEWALK variants)
struct Stacks {
	WaveStack* wave;
	LaneStack lane;
	uint32_t* pend;     // HXR_F_STACK1 variants: this wave's pending-ray slot, HXR_PEND_DWORDS rows x 64 lanes
	int repl = 0;       // HXR_F_REPL: replicas per ray of the packet being traced (wave-uniform; 0 = one sample per lane), set per packet
};
#define HXR_PEND_DWORDS 18   // RayItem: o, d (6 dwords each), weight (3), depth, flags, key

// The four products of Mesh::intersectTriangle (src/mesh.cpp:178-195): Dcr = (ABxAC).(-d), and the numerators of gamma, lambda2, lambda3.
// One definition for the test (record wave-uniform, in scalar registers) and for surface_of(), which forms the barycentrics of the
// accepted hit again from the same expressions.
HXR_DEV void tri_products(d3 N, d3 A, d3 AB, d3 AC, d3 o, d3 nd, double& Dcr, double& ga, double& gb, double& gc)
{
	Dcr = dot(N, nd);
	d3 H = o - A;
	ga = dot(N, H);
	// (H x AC) . nd = (nd x H) . AC  and  (AB x H) . nd = -(nd x H) . AB: one cross product serves both numerators (21 operations per
	// triangle instead of 27).  The same real numbers rounded along another path: gamma (= ga / Dcr, untouched) stays bit-identical, the
	// barycentrics move in their last bits -- a decision differs from the reference's only for a ray within an ulp of a triangle's edge.
	const d3 P = cross(nd, H);
	gb = dot(P, AC);
	gc = -dot(P, AB);
}

// BBox::inside (src/bbox.h:78-85) with the slack handed in: the leaf loop passes an OPAQUE 1e-6 so that the six inflated bounds are
// formed where a triangle is accepted (rare) instead of being hoisted out of the loop into twelve vector registers
HXR_DEV bool box_holds_eps(const Box& b, d3 p, double eps)
{
	return b.lo.x - eps <= p.x && p.x <= b.hi.x + eps && b.lo.y - eps <= p.y && p.y <= b.hi.y + eps && b.lo.z - eps <= p.z && p.z <= b.hi.z + eps;
}

// Mesh::intersectTriangle (src/mesh.cpp:178-195) against the running best hit `g`, and the leaf rule of Mesh::intersectKD
// (src/mesh.cpp:228): returns true when the triangle is accepted AND its hit point lies in the leaf's box.  One definition for the
// packet walk (record and index wave-uniform: scalar loads) and the per-lane walk (vector loads).  `hot` = the record's scene-wide index.
HXR_DEV bool triangle_decide(bool facing, double Dcr, double ga, double gb, double gc, int hot, const Box& box, d3 o, d3 d, GeomHit& g);

template <bool BACKFACE>
HXR_DEV bool triangle_test(const HxrSceneDev& S, const HxrMeshDev& M, const HxrTriHot& T, int hot, const Box& box, d3 o, d3 d, d3 nd, GeomHit& g)
{
	bool ok = true;
	if constexpr (BACKFACE) {
		const HXR_GLOBAL hxr_triangle& ct = in_hbm(M.tris)[S.hot_ids[hot]];
		ok = !(dot(d, ld3(ct.gnormal)) > 0);
	}
	// The four sign / magnitude rejections are evaluated without branching between them: the lanes of a
	// packet fail at different ones, so nested early-outs only add mask bookkeeping (8 levels of it).
	double Dcr, ga, gb, gc;
	tri_products(ld3(T.N), ld3(T.A), ld3(T.AB), ld3(T.AC), o, nd, Dcr, ga, gb, gc);
	return triangle_decide(ok, Dcr, ga, gb, gc, hot, box, o, d, g);
}

// The rest of Mesh::intersectTriangle once the four products are known (see triangle_test); `facing`: not culled as a back face
HXR_DEV bool triangle_decide(bool facing, double Dcr, double ga, double gb, double gc, int hot, const Box& box, d3 o, d3 d, GeomHit& g)
{
	// "x * rD < 0" <=> x is non-zero and its sign differs from Dcr's <=> x times (1 with Dcr's sign) is < 0 -- an exact product, and a zero
	// of either sign is not < 0: three multiplications and the comparisons, instead of two comparisons per numerator and mask arithmetic
	const double sgn = copysign(1.0, Dcr);
	const double sa = ga * sgn, sb = gb * sgn, sc = gc * sgn;
	// (the two spellings compile to different mask code; each translation unit takes the one its kernels measured faster with:
	// meshes16 7.52 -> 7.37 ms with the short-circuit form and 7.47 with the bitwise one, cornell16 53.4 -> 54.2 and 52.2)
#ifdef HXR_TRI_DECIDE_BITWISE
	bool ok = facing & !(fabs(Dcr) < 1e-12) & !(sa < 0) & !(sb < 0) & !(sc < 0);   // gamma, lambda2, lambda3 < 0
#else
	const bool behind = (sa < 0) || (sb < 0) || (sc < 0);   // gamma, lambda2, lambda3 < 0
	bool ok = facing && !(fabs(Dcr) < 1e-12) && !behind;
#endif
	bool inside = false;
	if (ok) {
		double rD = 1 / Dcr;
		double gamma = ga * rD, l2 = gb * rD, l3 = gc * rD;
		// gamma, l2, l3 are >= 0 here: rD has Dcr's sign, and a product has the product of the signs.  Then l1 = 1 - (l2 + l3) <= 1, and
		// l1 < 0 <=> l2 + l3 > 1 (for a sum above 1 the difference is exact), which also bounds l2 and l3 by 1 (a rounded sum is no smaller
		// than either non-negative term): the eight bounds of Mesh::intersectTriangle (src/mesh.cpp:189-193) come down to these two.
		bool take = !(gamma > g.t) & !(l2 + l3 > 1);
		if (take) {
			g.t = gamma;
			g.tri = hot;
			double eps = 1e-6;
			asm volatile("" : "+v"(eps));
			inside = box_holds_eps(box, o + d * gamma, eps);
		}
	}
	return inside;
}

// value of the lane `n` below in the same row of 16 lanes (DPP row_shr:n); a lane without one keeps its own value
template <int N>
HXR_DEV double row_shr(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x110 + N, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x110 + N, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

// The leaf of a REPLICATED packet (HXR_F_REPL): every ray of the wave is held by R adjacent, aligned lanes with identical state.  Replica r tests triangle
// r0 + r of every round of R triangles; the sequential loop's rule -- triangle i is taken iff it is valid and its distance is not above the running best,
// which is the smallest distance of any valid triangle before it (or the best the ray came with) -- is an exclusive prefix minimum over the R lanes.  The ray
// keeps the LAST taken triangle (ties go to the later one, like the loop's `!(gamma > best)`), and the leaf reports a hit iff any TAKEN triangle's point lies in
// the leaf's box (Mesh::intersectKD ORs that over every take, src/mesh.cpp:228).  Same products, same reciprocal, same comparisons as triangle_decide: same
// bits.  (A NaN distance, which the sequential loop would accept and keep, cannot occur: |Dcr| >= 1e-12 and finite operands.)
template <int R>
HXR_DEV unsigned long long leaf_query_replicated(const HxrSceneDev& S, const HxrMeshDev& M, int first, int count, unsigned long long mask, const Box& box, d3 o, d3 d, d3 nd, GeomHit& g)
{
	const int base = M.hot_first + first;
	const int lane = threadIdx.x & 63, rep = lane & (R - 1), group0 = lane & ~(R - 1);
	int hits = 0;
	if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
		for (int r0 = 0; r0 < count; r0 += R) {   // wave-uniform
			const int i = r0 + rep;
			bool valid = false, inside = false;
			double gamma = HXR_INF;
			if (i < count) {
				const HXR_GLOBAL HxrTriHot& hr = in_hbm(S.hot)[base + i];
				bool facing = true;
				if (M.backface_culling) facing = !(dot(d, ld3(in_hbm(M.tris)[in_hbm(S.hot_ids)[base + i]].gnormal)) > 0);   // src/mesh.cpp:180
				double Dcr, ga, gb, gc;
				tri_products(ld3(hr.N), ld3(hr.A), ld3(hr.AB), ld3(hr.AC), o, nd, Dcr, ga, gb, gc);
				const double sgn = copysign(1.0, Dcr);
				const double sa = ga * sgn, sb = gb * sgn, sc = gc * sgn;
				const bool behind = (sa < 0) || (sb < 0) || (sc < 0);
				if (facing && !(fabs(Dcr) < 1e-12) && !behind) {
					const double rD = 1 / Dcr;
					gamma = ga * rD;
					const double l2 = gb * rD, l3 = gc * rD;
					valid = !(l2 + l3 > 1);
					inside = box_holds(box, o + d * gamma);
				}
			}
			// inclusive prefix minimum of the valid distances over the group's lanes, then shifted by one: what the loop's running best would be at lane rep
			double pm = valid ? gamma : HXR_INF;
			{ const double t = row_shr<1>(pm); if (rep >= 1) pm = fmin(pm, t); }
			if (R > 2) { const double t = row_shr<2>(pm); if (rep >= 2) pm = fmin(pm, t); }
			if (R > 4) { const double t = row_shr<4>(pm); if (rep >= 4) pm = fmin(pm, t); }
			if (R > 8) { const double t = row_shr<8>(pm); if (rep >= 8) pm = fmin(pm, t); }
			double before = row_shr<1>(pm);
			if (rep == 0) before = HXR_INF;
			const bool taken = valid && !(gamma > fmin(g.t, before));
			const unsigned long long bt = hxr_ballot(taken), bi = hxr_ballot(taken && inside);
			const uint32_t gt = (uint32_t)(bt >> group0) & ((1u << R) - 1u), gi = (uint32_t)(bi >> group0) & ((1u << R) - 1u);
			const int last = 31 - __builtin_clz(gt | 1u);               // highest taken replica of this ray (0 when none: masked below)
			const double best = __shfl(gamma, group0 + last, 64);
			if (gt) {
				g.t = best;
				g.tri = base + r0 + last;
			}
			hits |= gi != 0 ? 1 : 0;
		}
	}
	return hxr_ballot(hits != 0);
}

// Leaf of Mesh::intersectKD (src/mesh.cpp:223-232): every triangle is tested against the running
// best distance; a hit counts for THIS leaf only if its point lies in the leaf's box.
// Mesh::intersectTriangle (src/mesh.cpp:178-195) accepts iff  0 <= gamma <= best, 0 <= l2 <= 1, 0 <= l3 <= 1,
// 0 <= l1 <= 1  with gamma = a*rD, l2 = b*rD, l3 = c*rD, rD = 1/Dcr.  The conjunction does not depend on
// the order of its terms, and  x*rD < 0  <=>  x and Dcr have opposite signs (x != 0), so the three
// "negative" rejections are decided from signs alone BEFORE the FP64 division is paid for.
// `first` / `count` are wave-uniform: the hot records (contiguous, leaf order) arrive through scalar loads
// and stay in SGPRs, so a triangle costs the lanes no vector registers.  The triangle NUMBER of a record (hot_ids) is not read
// here: a hit remembers its record, surface_of() looks the number up for the one hit a ray keeps.
// `mask`: the lanes taking part (wave-uniform); returns the lanes that found a hit inside the leaf's box.
template <bool COUNT, bool TOUCH, int REPL>   // REPL: 0 never, 1 the packets that say so (`repl`), 2 every packet of the kernel (a kernel that only takes replicated packets)
HXR_DEV unsigned long long leaf_query(const HxrSceneDev& S, const HxrMeshDev& M, int first, int count, unsigned long long mask, const Box& box, d3 o, d3 d, d3 nd,
                                      GeomHit& g, Tally<COUNT>& tl, int repl)
{
	tl.template lap<PH_MESH>();
	if constexpr (REPL != 0) {
		if (repl) {   // (wave-uniform) a replicated packet: the lane-parallel leaf
			const unsigned long long f = repl == 16 ? leaf_query_replicated<16>(S, M, first, count, mask, box, o, d, nd, g) : leaf_query_replicated<4>(S, M, first, count, mask, box, o, d, nd, g);
			tl.template lap<PH_LEAF>();
			return f;
		}
	}
	if constexpr (REPL != 2) {
#if defined(HXR_LEAF_STATS) || defined(HXR_LEAF_HIST)   // developer aids: how many lanes take part in a leaf visit (would testing ray x triangle pairs in parallel pay?)
	{
		const int k = __popcll(mask), me_ = threadIdx.x & 63;
		if (mask && __builtin_amdgcn_readfirstlane(me_) == me_ && S.dev_stats) {
#ifdef HXR_LEAF_STATS
			atomicAdd(S.dev_stats + 0, (unsigned long long)count);
			const int coop = 2 + (3 * k) / 2;
			atomicAdd(S.dev_stats + 1, (unsigned long long)(count < coop ? count : coop));
			atomicAdd(S.dev_stats + 2, 1ull);
			atomicAdd(S.dev_stats + 3, (unsigned long long)k);
			if (count == 0) atomicAdd(S.dev_stats + 4, 1ull);   // visits of EMPTY leaves
#else                   // leaf visits and their triangle iterations by the number of lanes taking part (<= 2, 3..8, >= 33)
			const int b = k <= 2 ? 0 : (k <= 8 ? 2 : (k >= 33 ? 4 : -1));
			if (b >= 0) { atomicAdd(S.dev_stats + b, 1ull); atomicAdd(S.dev_stats + b + 1, (unsigned long long)count); }
#endif
		}
	}
#endif
	const int base = M.hot_first + first;
	const HxrTriHot* T = S.hot + base;
	int hits = 0;   // per lane, touched only where a triangle is accepted: no mask bookkeeping per iteration
	[[maybe_unused]] uint32_t touch0 = 0, touch1 = 0;   // TOUCH: the scratch registers of the prefetch
	if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
		if (M.backface_culling) {
			for (int i = 0; i < count; i++) {
				tl.add(HC_TRI);
				tl.add(HC_LEAFIDX);
				hits |= (int)triangle_test<true>(S, M, T[i], base + i, box, o, d, nd, g);
			}
		} else {
			for (int i = 0; i < count; i++) {
				tl.add(HC_TRI);
				tl.add(HC_LEAFIDX);
#if defined(HXR_CHAIN_LAPS)
				const HxrTriHot rec = T[i];
				HXR_CHAIN_ARRIVE2(rec.A[0], rec.N[2]);   // first and last word of the 96-byte record
				HXR_CHAIN_LAP(tl, PH4_LEAF_WAIT);
				hits |= (int)triangle_test<false>(S, M, rec, base + i, box, o, d, nd, g);
				HXR_CHAIN_LAP(tl, PH4_LEAF_COMPUTE);
#else
				if constexpr (TOUCH) {
					// HXR_F_TOUCH.  The record is read through the constant address space, or the inline asm below (a possible store, to the compiler) would turn
					// the loads behind it into vector loads.  The touch registers are consumed one iteration later, behind that iteration's wait for its own
					// record (scalar loads return out of order: one counter, one wait for all), so they stay reserved until the touch has landed.
					double Dcr, ga, gb, gc;
					const HXR_CONSTANT HxrTriHot* Tc = (const HXR_CONSTANT HxrTriHot*)(T + i);
					tri_products(ld3(Tc->N), ld3(Tc->A), ld3(Tc->AB), ld3(Tc->AC), o, nd, Dcr, ga, gb, gc);
					asm volatile("" ::"s"(touch0), "s"(touch1), "v"(Dcr));
					if (i + 1 < count) asm volatile("s_load_dword %0, %2, 0x60\n\ts_load_dword %1, %2, 0xa0" : "=&s"(touch0), "=&s"(touch1) : "s"(T + i));
					hits |= (int)triangle_decide(true, Dcr, ga, gb, gc, base + i, box, o, d, g);
				} else {
#ifdef HXR_CONST_TABLES
					double Dcr, ga, gb, gc;
					const HXR_CONSTANT HxrTriHot* Tc = (const HXR_CONSTANT HxrTriHot*)(T + i);
					tri_products(ld3(Tc->N), ld3(Tc->A), ld3(Tc->AB), ld3(Tc->AC), o, nd, Dcr, ga, gb, gc);
					hits |= (int)triangle_decide(true, Dcr, ga, gb, gc, base + i, box, o, d, g);
#else
					hits |= (int)triangle_test<false>(S, M, T[i], base + i, box, o, d, nd, g);
#endif
				}
#endif
			}
		}
	}
	// formed OUTSIDE the divergent region: a wave-uniform value assigned under a lane mask is only defined for the lanes of that mask
	const unsigned long long found = hxr_ballot(hits != 0);
	tl.template lap<PH_LEAF>();
	return found;
	} else return 0ull;
}

// BBox::testIntersect (slab form: box_hit) of BOTH children of a kd node for one ray.  The children's boxes are the node's with one
// bound of axis A replaced by the split position (BBox::split, src/bbox.h:210-216), so five of the six wall distances are the
// node's own and are formed once; every value is the very product box_hit() would form for the child's box, and the maxima / minima
// are exact operations whose grouping does not matter: `left` / `right` are box_hit()'s answers bit for bit.
// `origin_in_root`: this lane's origin lies in the mesh's inflated root box.  Node boxes nest (HxrMeshDev::boxes_nested), so a lane
// outside the root's is outside every node's: the inflated inside() shortcut (src/bbox.h:95) is only evaluated for lanes inside.
template <int A>
HXR_DEV void kd_children_hit_axis(const Box& b, double split, d3 o, d3 inv, bool origin_in_root, unsigned long long mask, bool& left, bool& right)
{
	constexpr int U = (A + 1) % 3, V = (A + 2) % 3;
	const double lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z};
	const double oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
	const double tu0 = (lo[U] - oo[U]) * ii[U], tu1 = (hi[U] - oo[U]) * ii[U];
	const double tv0 = (lo[V] - oo[V]) * ii[V], tv1 = (hi[V] - oo[V]) * ii[V];
	const double ta0 = (lo[A] - oo[A]) * ii[A], ta1 = (hi[A] - oo[A]) * ii[A], ts = (split - oo[A]) * ii[A];
	const double near_uv = fmax(fmin(tu0, tu1), fmin(tv0, tv1)), far_uv = fmin(fmax(tu0, tu1), fmax(tv0, tv1));
	const double tnl = fmax(near_uv, fmin(ta0, ts)), tfl = fmin(far_uv, fmax(ta0, ts));   // left child: [lo, split] on axis A
	const double tnr = fmax(near_uv, fmin(ts, ta1)), tfr = fmin(far_uv, fmax(ts, ta1));   // right child: [split, hi]
	left = tnl <= tfl && tfl >= 0;
	right = tnr <= tfr && tfr >= 0;
	if (hxr_ballot(origin_in_root && !(left && right)) & mask) {   // BBox::inside of the child boxes, when it can matter to a lane of the visit
		const bool in_uv = lo[U] - 1e-6 <= oo[U] && oo[U] <= hi[U] + 1e-6 && lo[V] - 1e-6 <= oo[V] && oo[V] <= hi[V] + 1e-6;
		left = left || (origin_in_root && in_uv && lo[A] - 1e-6 <= oo[A] && oo[A] <= split + 1e-6);
		right = right || (origin_in_root && in_uv && split - 1e-6 <= oo[A] && oo[A] <= hi[A] + 1e-6);
	}
}

// Evaluated by every lane of the wave, whether or not it takes part in the visit (`mask`): the instructions cost the same, there is
// no branch, and the answers are wave-uniform ballots the caller restricts to the visit's lanes.
HXR_DEV void kd_children_hit(const Box& b, int axis, double split, d3 o, d3 inv, bool origin_in_root, unsigned long long mask, bool& left, bool& right)
{
	// `axis` is wave-uniform: three straight-line copies, no selects
	if (axis == 0) kd_children_hit_axis<0>(b, split, o, inv, origin_in_root, mask, left, right);
	else if (axis == 1) kd_children_hit_axis<1>(b, split, o, inv, origin_in_root, mask, left, right);
	else kd_children_hit_axis<2>(b, split, o, inv, origin_in_root, mask, left, right);
}

// Mesh::intersect (src/mesh.cpp:247-263) as a PACKET walk of the reference's kd-tree.
//
// The reference descends per ray: at a binary node it visits the child on the ray origin's side first,
// then the other, each only if the ray touches that child's box, and a ray stops at the first leaf that
// reports a hit inside the leaf's box (src/mesh.cpp:221-245).  The 64 rays of a wavefront come from one
// 2x2..8x8 pixel block and follow almost the same path, so the WAVE walks the tree once, with wave-uniform
// control flow, and every step carries the set of lanes on whose behalf it is taken:
//   * a lane takes part in a node's visit iff its own reference traversal would enter that node: it took part
//     in the parent's visit, is still searching when the node's turn comes, and its ray passes the node's box;
//   * the visiting order is the reference's PER LANE: lanes whose origin lies on the left of the split go
//     left-then-right, the others right-then-left.  When both groups exist the children are queued as
//     left(for the left-first lanes), right(for everybody), left(for the right-first lanes), so each lane still
//     sees its own near child before its own far child;
//   * a lane that finds a hit inside its leaf's box leaves the search, exactly like the reference's `return true`.
// Round 3: a node's two children are box-tested TOGETHER when the node is entered (kd_children_hit: five of the six wall distances
// are shared), for every lane of the visit.  The reference tests the far child's box only when the near subtree has come back
// empty-handed, but the test has no side effect: making it early changes nothing a lane can observe, a lane that leaves the search in
// the near subtree simply never uses its answer.  What is queued is therefore a visit together with the lanes that PASSED the node's
// box; a child nobody reaches is not queued at all, and a visit that resumes from the stack needs no box test (the old walk popped a
// node, loaded its box, tested it and -- for half of them -- found that no lane reached it).
// Node records are the same for all lanes of a step, so they are fetched by scalar loads into SGPRs -- both
// children of a binary node at once (they are adjacent, 32 B), as soon as the node is entered; per-lane state is just the ray and
// its best hit.  The work counters are per lane and equal the reference's per-ray counts: a lane's test of its NEAR child is
// counted when the node is entered, that of its FAR child when that child's turn comes and the lane is still searching (`due`
// masks, instrumented builds only) -- exactly when the reference makes it.
template <bool COUNT, bool TOUCH, int REPL>
HXR_DEV bool mesh_query(const HxrSceneDev& S, const HxrMeshDev& M, d3 o, d3 d, GeomHit& g, const Stacks& st, Tally<COUNT>& tl)
{
	tl.template lap<PH_QUERY>();
#ifdef HXR_CONST_TABLES
	const HXR_CONSTANT HxrKdSlim* kd = (const HXR_CONSTANT HxrKdSlim*)(S.kd + M.kd_first);
	const HXR_CONSTANT HxrKdBox* kd_box = (const HXR_CONSTANT HxrKdBox*)(S.kd_box + M.kd_first);
#else
	const HxrKdSlim* kd = S.kd + M.kd_first;
	const HxrKdBox* kd_box = S.kd_box + M.kd_first;
#endif
	Box cb{ld3(M.bbox_min), ld3(M.bbox_max)};   // box of the current node (scalar registers); the root's is the mesh's
	d3 inv = mk3(1 / d.x, 1 / d.y, 1 / d.z);
	d3 nd = -d;
	g.t = HXR_INF;
	g.tri = -1;
	bool in_root = box_test<COUNT>(cb, o, d, inv, tl);
	unsigned long long alive = hxr_ballot(in_root);   // lanes still searching
	if (!alive) { tl.template lap<PH_MESH>(); return false; }
	if (!M.has_kd) {   // brute force over every triangle (src/mesh.cpp:253-261)
		Box everything{mk3(-HXR_INF, -HXR_INF, -HXR_INF), mk3(HXR_INF, HXR_INF, HXR_INF)};
		leaf_query<COUNT, TOUCH, REPL>(S, M, 0, M.num_triangles, alive, everything, o, d, nd, g, tl, st.repl);
		tl.template lap<PH_MESH>();
		return in_root && g.t < HXR_INF;
	}
	const bool origin_in_root = !M.boxes_nested || box_holds(cb, o);
	unsigned long long found = 0;
	int sp = 0;
	unsigned long long mask = alive;   // lanes taking part in the visit of the current node
	if (in_root) tl.add(HC_KD);
#ifdef HXR_CONST_TABLES
	HxrKdSlim R = ld_const(kd);
#else
	HxrKdSlim R = kd[0];               // record of the current node (scalar registers)
#endif
	HXR_CHAIN_ARRIVE2(R.first, R.axis_count);
	HXR_CHAIN_LAP(tl, PH4_KD_WAIT);
	for (;;) {
		const int ac = R.axis_count;
		const int axis = ac & 3;
		bool descended = false;
#ifdef HXR_LEAF_STATS
		if (S.dev_stats && __builtin_amdgcn_readfirstlane((int)(threadIdx.x & 63)) == (int)(threadIdx.x & 63)) atomicAdd(S.dev_stats + 5, 1ull);   // walk steps
#endif
		if (axis == HXR_KD_LEAF) {
			const unsigned long long here = leaf_query<COUNT, TOUCH, REPL>(S, M, R.first, ac >> 2, mask, cb, o, d, nd, g, tl, st.repl);
			found |= here;
			alive &= ~here;
		} else {
			const double split = R.split;
			const int left = R.first;
#ifdef HXR_CONST_TABLES
			const HxrKdPair kids = ld_const((const HXR_CONSTANT HxrKdPair*)(kd + left));
#else
			const HxrKdPair kids = *(const HxrKdPair*)(kd + left);   // in flight while the children's boxes are being tested
#endif
			bool pl, pr;
			kd_children_hit(cb, axis, split, o, inv, origin_in_root, mask, pl, pr);
			if constexpr (COUNT) {   // instrumented builds: the children's slab answers against BBox::testIntersect taken literally, for the lanes of the visit (hxr_counters::box_test_mismatch)
				if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
					Box lb = cb, rb = cb;
					if (axis == 0) { lb.hi.x = split; rb.lo.x = split; } else if (axis == 1) { lb.hi.y = split; rb.lo.y = split; } else { lb.hi.z = split; rb.lo.z = split; }
					if (box_hit_literal(lb, o, d, inv) != pl) tl.add(HC_BOXMISMATCH);
					if (box_hit_literal(rb, o, d, inv) != pr) tl.add(HC_BOXMISMATCH);
				}
			}
			// lanes of the visit that reach the left / right child, and those whose near child is the right one
			const unsigned long long ml = hxr_ballot(pl) & mask, mr = hxr_ballot(pr) & mask, rm = hxr_ballot(comp(o, axis) > split) & mask;
			if (COUNT && __builtin_amdgcn_inverse_ballot_w64(mask)) tl.add(HC_BBOX);   // the near child's test; the far child's is counted when its turn comes
			const unsigned long long lm = mask & ~rm;
			// the visits of this node's children in execution order: left for the left-first lanes, right for everybody, left for the
			// right-first lanes -- each with the lanes that pass the child's box
			const unsigned long long v0 = lm & ml, v1 = mask & mr, v2 = rm & ml;
			const int first = v0 ? 0 : (v1 ? 1 : (v2 ? 2 : 3));
			if (COUNT) {   // far-child tests whose turn is NOW: those of the visit entered directly and of the empty visits before it
				const unsigned long long now_due = (first >= 1 ? lm : 0ull) | (first >= 2 ? rm : 0ull);
				if (__builtin_amdgcn_inverse_ballot_w64(now_due)) tl.add(HC_BBOX);
			}
			// later visits wait on the stack, last one first
			if (first < 2 && (v2 || (COUNT && rm))) {
				st.wave->node[sp] = left; st.wave->mask_lo[sp] = (uint32_t)v2; st.wave->mask_hi[sp] = (uint32_t)(v2 >> 32);
				if (COUNT) { st.wave->due_lo[sp] = (uint32_t)rm; st.wave->due_hi[sp] = (uint32_t)(rm >> 32); }
				sp++;
			}
			if (first < 1 && (v1 || (COUNT && lm))) {
				st.wave->node[sp] = left + 1; st.wave->mask_lo[sp] = (uint32_t)v1; st.wave->mask_hi[sp] = (uint32_t)(v1 >> 32);
				if (COUNT) { st.wave->due_lo[sp] = (uint32_t)lm; st.wave->due_hi[sp] = (uint32_t)(lm >> 32); }
				sp++;
			}
			HXR_CHAIN_LAP(tl, PH4_KD_COMPUTE);
			if (first < 3) {
				HXR_CHAIN_ARRIVE2(kids.kid[0].first, kids.kid[1].first);
				HXR_CHAIN_LAP(tl, PH4_KD_WAIT);
				// BBox::split (src/bbox.h:210-216): the child keeps the box and replaces one bound by the split position
				const bool go_left = first != 1;
				const int moved = go_left ? axis : axis + 3;   // which of the six bounds becomes the split position
				cb.hi.x = moved == 0 ? split : cb.hi.x;
				cb.hi.y = moved == 1 ? split : cb.hi.y;
				cb.hi.z = moved == 2 ? split : cb.hi.z;
				cb.lo.x = moved == 3 ? split : cb.lo.x;
				cb.lo.y = moved == 4 ? split : cb.lo.y;
				cb.lo.z = moved == 5 ? split : cb.lo.z;
				R = go_left ? kids.kid[0] : kids.kid[1];
				mask = first == 0 ? v0 : (first == 1 ? v1 : v2);
				if (COUNT && __builtin_amdgcn_inverse_ballot_w64(mask)) tl.add(HC_KD);
				descended = true;
			}
#ifdef HXR_WALK_STATS   // developer aid: [0] nodes whose first visit is entered directly, [1] pops, [2] pops nobody waits for any more, [4] inner nodes none of whose children is reached, [5] inner nodes
			if (S.dev_stats && __builtin_amdgcn_readfirstlane((int)(threadIdx.x & 63)) == (int)(threadIdx.x & 63)) { atomicAdd(S.dev_stats + (first < 3 ? 0 : 4), 1ull); atomicAdd(S.dev_stats + 5, 1ull); }
#endif
		}
		// otherwise the next pending visit some of whose lanes are still searching
		while (!descended && sp > 0) {
			sp--;
			const int cn = __builtin_amdgcn_readfirstlane(st.wave->node[sp]);
			const unsigned long long pm = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(st.wave->mask_hi[sp]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(st.wave->mask_lo[sp]);
			if (COUNT) {
				const unsigned long long due = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(st.wave->due_hi[sp]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(st.wave->due_lo[sp]);
				if (__builtin_amdgcn_inverse_ballot_w64(due & alive)) tl.add(HC_BBOX);
			}
			const unsigned long long m = pm & alive;
#ifdef HXR_WALK_STATS
			if (S.dev_stats && __builtin_amdgcn_readfirstlane((int)(threadIdx.x & 63)) == (int)(threadIdx.x & 63)) { atomicAdd(S.dev_stats + 1, 1ull); if (!m) atomicAdd(S.dev_stats + 2, 1ull); }
#endif
			if (!m) continue;
#ifdef HXR_CONST_TABLES
			R = ld_const(kd + cn);
			const HxrKdBox CB = ld_const(kd_box + cn);
#else
			R = kd[cn];
			const HxrKdBox CB = kd_box[cn];
#endif
			HXR_CHAIN_LAP(tl, PH4_POP);
			HXR_CHAIN_ARRIVE2(R.first, CB.hi[2]);
			HXR_CHAIN_LAP(tl, PH4_KD_WAIT);
			cb = Box{ld3(CB.lo), ld3(CB.hi)};
			mask = m;
			if (COUNT && __builtin_amdgcn_inverse_ballot_w64(m)) tl.add(HC_KD);
			descended = true;
		}
		if (!descended) break;
	}
	tl.template lap<PH_MESH>();
	return __builtin_amdgcn_inverse_ballot_w64(found);
}

// Mesh::intersect (src/mesh.cpp:247-263) walked by each lane for its own ray: the reference's recursion
// (src/mesh.cpp:221-245) with an explicit stack.  At a binary node the child on the origin's side is tried first; the other
// child waits on the stack UNTESTED and its box is only tested when the first subtree has come back empty-handed, like the
// reference's `if (childBB[i].testIntersect(ray) && intersectKD(...)) return true`, so the work counters are the per-ray
// counts of the reference and of the packet walk.  Node and triangle records arrive through per-lane vector loads; the stack
// is a column of 32-bit node numbers per lane in LDS (`entries` rows: the deepest tree of the scene + 2).
template <bool COUNT>
HXR_DEV bool mesh_query_lane(const HxrSceneDev& S, const HxrMeshDev& M, d3 o, d3 d, GeomHit& g, const LaneStack& ls, Tally<COUNT>& tl)
{
	const int lane = threadIdx.x & 63;
	const HxrKdSlim* kd = S.kd + M.kd_first;
	const HxrKdBox* kd_box = S.kd_box + M.kd_first;
	Box cb{ld3(M.bbox_min), ld3(M.bbox_max)};
	const d3 inv = mk3(1 / d.x, 1 / d.y, 1 / d.z), nd = -d;
	g.t = HXR_INF;
	g.tri = -1;
	if (!box_test<COUNT>(cb, o, d, inv, tl)) return false;
	if (!M.has_kd) {   // brute force over every triangle (src/mesh.cpp:253-261)
		const Box everything{mk3(-HXR_INF, -HXR_INF, -HXR_INF), mk3(HXR_INF, HXR_INF, HXR_INF)};
		for (int i = 0; i < M.num_triangles; i++) {
			tl.add(HC_TRI);
			tl.add(HC_LEAFIDX);
			if (M.backface_culling) triangle_test<true>(S, M, S.hot[M.hot_first + i], M.hot_first + i, everything, o, d, nd, g);
			else triangle_test<false>(S, M, S.hot[M.hot_first + i], M.hot_first + i, everything, o, d, nd, g);
		}
		return g.t < HXR_INF;
	}
	tl.add(HC_KD);
	int node = 0, sp = 0;
	for (;;) {
		const HxrKdSlim R = kd[node];
		const int ac = R.axis_count, axis = ac & 3;
		bool descended = false;
		if (axis == HXR_KD_LEAF) {
			bool found = false;
			const int first = R.first, count = ac >> 2;
			for (int i = 0; i < count; i++) {
				tl.add(HC_TRI);
				tl.add(HC_LEAFIDX);
				const int hi_ = M.hot_first + first + i;
				if (M.backface_culling ? triangle_test<true>(S, M, S.hot[hi_], hi_, cb, o, d, nd, g) : triangle_test<false>(S, M, S.hot[hi_], hi_, cb, o, d, nd, g)) found = true;
			}
			if (found) return true;
		} else {
			const double split = R.split;
			const bool right_first = comp(o, axis) > split;
			const int near = R.first + (right_first ? 1 : 0), far = R.first + (right_first ? 0 : 1);
			if (sp < ls.entries) ls.rows[sp * 64 + lane] = (uint32_t)far;
			else *ls.overflow = 1u;
			sp++;
			// BBox::split (src/bbox.h:210-216): the child keeps the box and replaces one bound by the split position
			const int moved = right_first ? axis + 3 : axis;
			Box nb;
			nb.hi.x = moved == 0 ? split : cb.hi.x;
			nb.hi.y = moved == 1 ? split : cb.hi.y;
			nb.hi.z = moved == 2 ? split : cb.hi.z;
			nb.lo.x = moved == 3 ? split : cb.lo.x;
			nb.lo.y = moved == 4 ? split : cb.lo.y;
			nb.lo.z = moved == 5 ? split : cb.lo.z;
			if (box_test<COUNT>(nb, o, d, inv, tl)) {
				tl.add(HC_KD);
				node = near;
				cb = nb;
				descended = true;
			}
		}
		while (!descended && sp > 0) {   // the next pending child whose box the ray reaches
			sp--;
			const int cn = sp < ls.entries ? (int)ls.rows[sp * 64 + lane] : 0;
			const HxrKdBox CB = kd_box[cn];
			const Box pb{ld3(CB.lo), ld3(CB.hi)};
			if (box_test<COUNT>(pb, o, d, inv, tl)) {
				tl.add(HC_KD);
				node = cn;
				cb = pb;
				descended = true;
			}
		}
		if (!descended) return false;
	}
}

// intersectTriangleFast, src/mesh.cpp:131-170
HXR_DEV bool tri_fast(d3 o, d3 d, d3 A, d3 B, d3 C, double& best)
{
	d3 AB = B - A, AC = C - A, D = -d, H = o - A;
	d3 N = cross(AB, AC);
	double Dcr = dot(N, D);
	if (fabs(Dcr) < 1e-12) return false;
	double l2 = dot(cross(H, AC), D) / Dcr;
	double l3 = dot(cross(AB, H), D) / Dcr;
	double gamma = dot(N, H) / Dcr;
	if (gamma < 0 || gamma > best) return false;
	if (l2 < 0 || l2 > 1 || l3 < 0 || l3 > 1 || l2 + l3 > 1) return false;
	best = gamma;
	return true;
}

HXR_DEV int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// Heightfield::intersect, src/heightfield.cpp:106-171
//
// Every lane follows the reference's marching sequence step for step; what the wave chooses is WHEN the two triangle
// tests of a cell run.  They are ~10x the cost of a marching step and only ~0.8 cells per ray need them, so a lane
// whose cell needs them parks (need_tri) while the other lanes go on marching, and the tests run for all parked lanes
// together once HXR_HF_TRI_BATCH of them wait or nobody can march any further.
#ifndef HXR_HF_TRI_BATCH
#define HXR_HF_TRI_BATCH 24
#endif
template <bool COUNT>
HXR_DEV bool heightfield_query(const HxrHeightfieldDev& F, d3 o, d3 d, GeomHit& g, Tally<COUNT>& tl)
{
	tl.add(HC_HFCALL);
	tl.template lap<PH_QUERY>();   // profiling builds: marching is charged to the "mesh walk" column, the triangle pairs to "leaf"
	Box box{ld3(F.bbox_min), ld3(F.bbox_max)};
	d3 step = d * (1.0 / sqrt(d.x * d.x + d.z * d.z));
	double dist = box_nearest(box, o, d);
	d3 p = o + d * (dist + 1e-6);
	double mx = 1.0 / d.x, mz = 1.0 / d.z;
	bool marching = box_holds(box, p), need_tri = false, found = false;
	int x0 = 0, z0 = 0;
	d3 pn = p;
	for (;;) {
#ifdef HXR_HF_STATS   // developer aid: lane-steps offered (kd_visits, unused without meshes) against lane-steps taken (box_test_mismatch) in a counting frame
		tl.add(HC_KD);
		if (marching && !need_tri) tl.add(HC_BOXMISMATCH);
#endif
		if (marching && !need_tri) {
			x0 = (int)floor(p.x);
			z0 = (int)floor(p.z);
			if (((unsigned)x0 >= (unsigned)F.W) | ((unsigned)z0 >= (unsigned)F.H)) {   // one comparison per coordinate, no nested masks
				marching = false;
			} else {
				const size_t cell = (size_t)z0 * F.W + x0;   // (x0, z0) is in range here: no clamping needed
				// ONE 16-byte load per step: the cell's maxH and pyramid levels 1..3 (step_rec: a table of its own, a quarter of the pyramid's
				// size, so that the dependent load of a marching step is served by L2)
				typedef float Levels4 __attribute__((ext_vector_type(4)));
				const Levels4 rec = *(const HXR_GLOBAL Levels4*)(in_hbm(F.step_rec) + cell * 4);
				const float cell_max = rec.x;
				int k = 0;
				if (F.use_optimization && F.max_k > 1) {
					// the reference probes levels 1, 2, ... one dependent load at a time; here levels 1-3 arrive with maxH and are decided in
					// registers (same comparisons, same order, same first failure), levels 4-7 -- one more 16-byte read of the pyramid -- follow
					// only when some lane got that far
					const HXR_GLOBAL float* hm = in_hbm(F.high_map) + cell * 16;
					const int top = F.max_k < 8 ? F.max_k : 8;
					bool passing = true;   // no branches between the levels of a group: the lanes of a wave stop at different ones
					k = 1;
					{
						const float lv[4] = {0.0f, rec.y, rec.z, rec.w};
#pragma unroll
						for (int j = 1; j < 4; j++) {
							passing = passing & (j < top) & (p.y + step.y * (1 << j) > lv[j]);
							k += passing ? 1 : 0;
						}
					}
					// two probes per step on average (hf_stats): levels 4..7 -- half of the record and of the comparisons -- only when some lane of the
					// wave has passed level 3, i.e. is about to skip eight cells or more
					if (hxr_ballot(passing))
					{
						const Levels4 hi = *(const HXR_GLOBAL Levels4*)(hm + 4);
						const float lv[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
						for (int j = 4; j < 8; j++) {
							passing = passing & (j < top) & (p.y + step.y * (1 << j) > lv[j - 4]);
							k += passing ? 1 : 0;
						}
					}
					while (k >= 8 && k < F.max_k && p.y + step.y * (1 << k) > hm[k]) k++;
					tl.add(HC_HFHIGH, k < F.max_k ? k : F.max_k - 1);   // probes made: every passed level plus the failed one
					k--;
				}
				// the skip of 2^k cells and the step to the next cell border are the same statement with another length: formed once, for
				// lanes of both kinds (a wave usually holds both, and then ran both branches)
				const double lx = d.x > 0 ? (ceil(p.x) - p.x) * mx : (floor(p.x) - p.x) * mx;
				const double lz = d.z > 0 ? (ceil(p.z) - p.z) * mz : (floor(p.z) - p.z) * mz;
				const double len = k > 0 ? (double)(1 << k) : fmin(lx, lz) + 1e-6;
				pn = p + step * len;
				if (k == 0) tl.add(HC_HFHEIGHT);
				need_tri = k == 0 && fmin(p.y, pn.y) < cell_max;
				if (!need_tri) {
					p = pn;
					marching = box_holds(box, p);
				}
			}
		}
		const uint64_t waiting = hxr_ballot(need_tri);
		if (hxr_ballot(marching && !need_tri) != 0 && __popcll(waiting) < HXR_HF_TRI_BATCH) continue;
		if (waiting == 0) break;
		tl.template lap<PH_MESH>();
		if (need_tri) {
			int x1 = clampi(x0 + 1, F.W - 1), z1 = clampi(z0 + 1, F.H - 1);   // getHeight clamps (src/heightfield.cpp:30-37)
			tl.add(HC_HFHEIGHT, 4);
			tl.add(HC_HFTRI, 2);
			const HXR_GLOBAL float* hs = in_hbm(F.heights);
			d3 A = mk3(x0, hs[(size_t)z0 * F.W + x0], z0);
			d3 B = mk3(x0 + 1, hs[(size_t)z0 * F.W + x1], z0);
			d3 C = mk3(x0 + 1, hs[(size_t)z1 * F.W + x1], z0 + 1);
			d3 D = mk3(x0, hs[(size_t)z1 * F.W + x0], z0 + 1);
			double best = HXR_INF;
			bool b1 = tri_fast(o, d, A, B, D, best);
			bool b2 = tri_fast(o, d, B, C, D, best);
			need_tri = false;
			if (b1 || b2) {
				g.t = best;
				found = true;
				marching = false;
			} else {
				p = pn;
				marching = box_holds(box, p);
			}
		}
		tl.template lap<PH_LEAF>();
	}
	tl.template lap<PH_MESH>();
	return found;
}

// Heightfield::getNormal, src/heightfield.cpp:83-104
HXR_DEV d3 heightfield_normal(const HxrHeightfieldDev& F, float x, float y)
{
	int x0 = (int)floorf(x), y0 = (int)floorf(y);
	float p = x - x0, q = y - y0;
	int x1 = min(F.W - 1, x0 + 1), y1 = min(F.H - 1, y0 + 1);
	x0 = clampi(x0, F.W - 1);
	y0 = clampi(y0, F.H - 1);
	const HXR_GLOBAL double* n = in_hbm(F.normals);
	d3 v = ld3(n + ((size_t)y0 * F.W + x0) * 3) * (double)((1 - p) * (1 - q)) + ld3(n + ((size_t)y0 * F.W + x1) * 3) * (double)(p * (1 - q)) +
	       ld3(n + ((size_t)y1 * F.W + x0) * 3) * (double)((1 - p) * q) + ld3(n + ((size_t)y1 * F.W + x1) * 3) * (double)(p * q);
	return unit(v);
}

// ------------------------------------------------------------------------------------------------
// nodes
// ------------------------------------------------------------------------------------------------
template <bool COUNT, unsigned F>
HXR_DEV bool geometry_query(const HxrSceneDev& S, const HxrNodeDev& N, d3 ol, d3 dl, GeomHit& g, const Stacks& st, Tally<COUNT>& tl)
{
	switch (N.geom_type) {
		case HXR_GEOM_PLANE: return plane_query(N.p, ol, dl, g);
		case HXR_GEOM_SPHERE:
			if constexpr ((F & HXR_F_SPHERE) != 0) return sphere_query(N.p, ol, dl, g);
			return false;
		case HXR_GEOM_CUBE:
			if constexpr ((F & HXR_F_CUBE) != 0) return cube_query(N.p, ol, dl, g);
			return false;
		case HXR_GEOM_MESH:
			if constexpr ((F & HXR_F_MESH) != 0) {
				if constexpr ((F & HXR_F_LANEWALK) != 0) return mesh_query_lane<COUNT>(S, S.meshes[N.ref], ol, dl, g, st.lane, tl);
				else return mesh_query<COUNT, (F & HXR_F_TOUCH) != 0 || (HXR_TOUCH_ALL && (F & HXR_F_GI) == 0), (replicating<COUNT, F> ? 1 : 0)>(S, S.meshes[N.ref], ol, dl, g, st, tl);
			}
			return false;
		case HXR_GEOM_HEIGHTFIELD:
			if constexpr ((F & HXR_F_HEIGHTFIELD) != 0) return heightfield_query<COUNT>(S.heightfields[N.ref], ol, dl, g, tl);
			return false;
		case HXR_GEOM_CSG_UNION:
		case HXR_GEOM_CSG_INTER:
		case HXR_GEOM_CSG_DIFF:
			if constexpr ((F & HXR_F_EXTRA) != 0) return csg_query_any<F>(S, S.geometries[N.geom], ol, dl, g);
			return false;
		default: return false;
	}
}

// object-space hit point of a query result: start + dir * dist, except for CSG whose point is the one the
// operand computed on its k-th re-shoot (src/geometry.cpp:158-161,186)
template <unsigned F>
HXR_DEV d3 local_hit_point(const HxrSceneDev& S, const HxrNodeDev& N, d3 ol, d3 dl, const GeomHit& g)
{
	if constexpr ((F & HXR_F_EXTRA) != 0) {
		if (N.geom_type >= HXR_GEOM_CSG_UNION && N.geom_type <= HXR_GEOM_CSG_DIFF) {
			const hxr_geometry& C = S.geometries[N.geom];
			d3 ip, n;
			double u, v;
			int prim;
			csg_rehit_any<F>(S, S.geometries[(g.tri & 1) ? C.right : C.left], ol, dl, g.tri >> 1, ip, n, u, v, prim);
			return ip;
		}
	}
	return ol + dl * g.t;
}

// Node::intersect (src/geometry.cpp:196-208) up to the world-space distance
template <bool COUNT, unsigned F>
HXR_DEV bool node_query(const HxrSceneDev& S, int ni, d3 o, d3 d, double reach_limit, GeomHit& g, double& wdist, const Stacks& st, Tally<COUNT>& tl)
{
	const HxrNodeDev& N = S.nodes[__builtin_amdgcn_readfirstlane(ni)];   // the node scan is wave-uniform: scalar loads
	tl.add(HC_NODE);
	if (N.cullable == 2) {
		// A plane node.  Plane::intersect (src/geometry.cpp:33-34) misses when the ray starts on one side and does not head for the
		// plane, and a hit farther than `reach_limit` along the ray cannot matter to the caller (visible(): beyond the segment;
		// raycast(): beyond the best hit so far).  Both are decided in world space from s(p) = p . c - k, the object-space height of a
		// point above the plane (c = the y-column of the node's inverse matrix, so s is the very sum Node::intersect forms for
		// ol.y - py): the ray is clearly heading away, or the piece of it that matters stays clearly on one side.  "Clearly" = by 1e-9
		// relative, seven orders above rounding; anything closer takes the exact path below.  It pays when EVERY lane of the wave
		// agrees (like the sphere cull): the shadow turns of a path tracer inside a room of planes (smallpt) skip all its walls.
		const d3 c = mk3(N.cull_lo[0], N.cull_lo[1], N.cull_lo[2]);
		const double k = N.cull_hi[0], cn = N.cull_hi[1];
		const double sA = dot(o, c) - k, h = dot(d, c);
		const double eps = 1e-9 * (fabs(sA) + fabs(k) + cn), eh = 1e-9 * cn;
		const bool away = (sA > eps && h > eh) || (sA < -eps && h < -eh);
		bool far_side = false;
		if (reach_limit < HXR_INF) {
			const double sB = sA + h * reach_limit, eb = eps + 1e-9 * fabs(h * reach_limit);
			far_side = (sA > eps && sB > eb) || (sA < -eps && sB < -eb);
		}
		if (!hxr_ballot(!(away || far_side))) return false;
	} else if (N.cullable == 1) {
		// World-space rejection: when NO ray of the wave reaches the node's (inflated) world box, the reference's
		// Node::intersect would return false for every one of them -- after transforming the ray and, for a mesh,
		// failing the root BBox::testIntersect (src/mesh.cpp:249), which is counted as if it had been made.
		// bounding sphere of that box: the ray's line passes farther than the radius from the centre, or the sphere
		// lies wholly behind the origin (`d` is a unit vector; no division needed)
		d3 oc = mk3(N.cull_lo[0], N.cull_lo[1], N.cull_lo[2]) - o;   // cull_lo = centre, cull_hi[0] = radius^2
		double along = dot(oc, d), oc2 = len2(oc), r2 = N.cull_hi[0];
		bool reach = !(oc2 - along * along > r2) && !(along < 0 && oc2 > r2);
		if (!hxr_ballot(reach)) {
			if (N.geom_type == HXR_GEOM_MESH) tl.add(HC_BBOX);
			if (N.geom_type == HXR_GEOM_HEIGHTFIELD) tl.add(HC_HFCALL);
			return false;
		}
	}
	d3 ol, dl;
	const bool identity = fast_transforms<F> && N.identity;   // wave-uniform: the matrices of this node are exactly the identity
	if (identity) {
		ol = mul_identity(o - ld3(N.T.offset));
		dl = unit_lazy<fast_transforms<F>>(mul_identity(d));
	} else {
		ol = mulm(o - ld3(N.T.offset), N.T.inv_m);
		dl = unit_lazy<fast_transforms<F>>(mulm(d, N.T.inv_m));
	}
	if (!geometry_query<COUNT, F>(S, N, ol, dl, g, st, tl)) return false;
	const d3 lp = local_hit_point<F>(S, N, ol, dl, g);
	d3 ipw = (identity ? mul_identity(lp) : mulm(lp, N.T.m)) + ld3(N.T.offset);
	wdist = dist3(o, ipw);
	return true;
}

// IntersectionInfo (src/geometry.h:33-40) of the accepted hit
struct Surface {
	double dist;
	d3 ip, norm, dNdx, dNdy;
	double u, v;
};

template <bool COUNT, unsigned F>
HXR_DEVX void surface_of(const HxrSceneDev& S, int ni, d3 o, d3 d, const GeomHit& g, double wdist, Surface& sf, Tally<COUNT>& tl)
{
	const HxrNodeDev& N = S.nodes[ni];
	d3 ol, dl;
	const bool identity = fast_transforms<F> && N.identity;
	if (identity) {
		ol = mul_identity(o - ld3(N.T.offset));
		dl = unit_lazy<fast_transforms<F>>(mul_identity(d));
	} else {
		ol = mulm(o - ld3(N.T.offset), N.T.inv_m);
		dl = unit_lazy<fast_transforms<F>>(mulm(d, N.T.inv_m));
	}
	d3 ip = local_hit_point<F>(S, N, ol, dl, g);
	d3 n = mk3(0, 0, 0);
	sf.dNdx = mk3(0, 0, 0);
	sf.dNdy = mk3(0, 0, 0);
	sf.u = sf.v = 0;
	switch (N.geom_type) {
		case HXR_GEOM_PLANE:
			n = mk3(0, (ol.y > N.p[0]) ? 1 : -1, 0);
			sf.u = ip.x;
			sf.v = ip.z;
			sf.dNdx = mk3(1, 0, 0);
			sf.dNdy = mk3(0, 0, 1);
			break;
		case HXR_GEOM_SPHERE: if constexpr ((F & HXR_F_SPHERE) != 0) {
			n = unit(ip - ld3(N.p));
			if (N.needs_uv) {   // asin / atan2 in FP64 cost more than the rest of the hit: only when a texture or bump map will read them
				double v = asin(n.y), u = atan2(n.z, n.x);
				v = -(v / HXR_PI + 0.5f);
				u = u / (2 * HXR_PI) + 0.5f;
				if (N.p[4] != 1) { u *= N.p[4]; v *= N.p[4]; }
				sf.u = u;
				sf.v = v;
			}
			break;
		}
		case HXR_GEOM_CUBE: if constexpr ((F & HXR_F_CUBE) != 0) {
			int a = g.tri >> 1;
			double sg = (g.tri & 1) ? 1 : -1;
			n = mk3(a == 0 ? sg : 0, a == 1 ? sg : 0, a == 2 ? sg : 0);
			if (a == 0) { sf.u = ip.y; sf.v = ip.z; }
			else if (a == 1) { sf.u = ip.x; sf.v = ip.z; }
			else { sf.u = ip.x; sf.v = ip.y; }
			break;
		}
		case HXR_GEOM_MESH: if constexpr ((F & HXR_F_MESH) != 0) {   // Mesh::intersectTriangle's attribute half, src/mesh.cpp:196-218
			const HxrMeshDev& M = S.meshes[N.ref];
			// the accepted hit: its hot record gives the barycentrics again (the products and the reciprocal of the test, tri_products),
			// its entry of hot_ids the triangle
			const HXR_GLOBAL HxrTriHot& hr = in_hbm(S.hot)[g.tri];
			double Dcr, ga, gb, gc;
			tri_products(ld3(hr.N), ld3(hr.A), ld3(hr.AB), ld3(hr.AC), ol, -dl, Dcr, ga, gb, gc);
			const double rD = 1 / Dcr, hit_l2 = gb * rD, hit_l3 = gc * rD;
			const HXR_GLOBAL hxr_triangle& t = in_hbm(M.tris)[in_hbm(S.hot_ids)[g.tri]];
			const HXR_GLOBAL double* uvs = in_hbm(M.uvs);
			tl.add(HC_MESHHIT);
			d3 tA = ld3(uvs + 3 * (size_t)t.t[0]), tB = ld3(uvs + 3 * (size_t)t.t[1]), tC = ld3(uvs + 3 * (size_t)t.t[2]);
			d3 tc = tA + (tB - tA) * hit_l2 + (tC - tA) * hit_l3;
			sf.u = tc.x;
			sf.v = tc.y;
			if (M.faceted) {
				n = ld3(t.gnormal);
			} else {
				const HXR_GLOBAL double* ns = in_hbm(M.normals);
				d3 nA = ld3(ns + 3 * (size_t)t.n[0]), nB = ld3(ns + 3 * (size_t)t.n[1]), nC = ld3(ns + 3 * (size_t)t.n[2]);
				n = unit(nA + (nB - nA) * hit_l2 + (nC - nA) * hit_l3);
			}
			sf.dNdx = ld3(t.dNdx);
			sf.dNdy = ld3(t.dNdy);
			break;
		}
		case HXR_GEOM_HEIGHTFIELD: if constexpr ((F & HXR_F_HEIGHTFIELD) != 0) {
			const HxrHeightfieldDev& Hf = S.heightfields[N.ref];
			tl.add(HC_HFNORM);
			n = heightfield_normal(Hf, (float)ip.x, (float)ip.z);
			sf.u = ip.x / Hf.W;
			sf.v = ip.z / Hf.H;
			sf.dNdx = mk3(1, 0, 0);
			sf.dNdy = mk3(0, 0, 1);
			break;
		}
		case HXR_GEOM_CSG_UNION:
		case HXR_GEOM_CSG_INTER:
		case HXR_GEOM_CSG_DIFF: if constexpr ((F & HXR_F_EXTRA) != 0) {
			const hxr_geometry& C = S.geometries[N.geom];
			const hxr_geometry& G = S.geometries[(g.tri & 1) ? C.right : C.left];
			d3 hit;
			int prim = 0;
			csg_rehit_any<F>(S, G, ol, dl, g.tri >> 1, hit, n, sf.u, sf.v, prim);
			n = faceforward(dl, n);   // src/geometry.cpp:187
			if (prim == HXR_GEOM_PLANE) { sf.dNdx = mk3(1, 0, 0); sf.dNdy = mk3(0, 0, 1); }
			break;
		}
		default: break;
	}
	sf.ip = (identity ? mul_identity(ip) : mulm(ip, N.T.m)) + ld3(N.T.offset);
	sf.norm = unit(identity ? mul_identity(n) : mulm(n, N.T.tinv_m));
	sf.dist = wdist;
}

// ------------------------------------------------------------------------------------------------
// lights
// ------------------------------------------------------------------------------------------------
// RectLight::intersect, src/lights.cpp:50-75 (PointLight::intersect returns 0, :27-30)
template <bool RECT>
HXR_DEV int light_query(const hxr_light& L, d3 o, d3 d, double& dist)
{
	if (!RECT || L.type != HXR_LIGHT_RECT) return 0;
	d3 s = mulm(o - ld3(L.T.offset), L.T.inv_m);
	d3 dl = unit_lazy(mulm(d, L.T.inv_m));
	if (fabs(dl.y) < 1e-12) return 0;
	double t = -(s.y / dl.y);
	if (t < 0) return 0;
	d3 p = s + dl * t;
	if (fabs(p.x) < 0.5 && fabs(p.z) < 0.5) {
		double w = len((mulm(p, L.T.m) + ld3(L.T.offset)) - o);
		if (w < dist) {
			dist = w;
			return s.y < 0 ? +1 : -1;
		}
	}
	return 0;
}

// Light::getNthSample: PointLight src/lights.h:88-92, RectLight src/lights.cpp:37-51 ((j1, j2) stand for its two randDouble())
template <bool RECT>
HXR_DEV void light_sample(const hxr_light& L, int idx, d3 shade_pos, double j1, double j2, d3& pos, f3& color)
{
	if (!RECT || L.type == HXR_LIGHT_POINT) {
		pos = ld3(L.pos);
		color = ldf(L.color);
		return;
	}
	double lx = ((idx % L.x_subd) + j1) / L.x_subd;
	double ly = ((idx / L.x_subd) + j2) / L.y_subd;
	pos = mulm(mk3(lx - 0.5, -1e-6, ly - 0.5), L.T.m) + ld3(L.T.offset);
	d3 ls = mulm(shade_pos - ld3(L.T.offset), L.T.inv_m);
	if (ls.y < 0) color = fdiv(ldf(L.color) * (float)(-ls.y), (float)len(ls));
	else color = mkf(0, 0, 0);
}

template <bool RECT>
HXR_DEV double light_solid_angle(const hxr_light& L, d3 p)   // RectLight::getSolidAngle, src/lights.cpp:91-102
{
	if (!RECT || L.type != HXR_LIGHT_RECT) return 0;
	d3 dir = unit_lazy(mulm(mk3(0, -1, 0), L.T.m));
	d3 to_p = p - (mulm(mk3(0, 0, 0), L.T.m) + ld3(L.T.offset));
	double c = dot(to_p, dir);
	if (c < 0) return 0;
	double dd = len(to_p);
	c /= dd;
	return L.area * c / ((1 + dd) * (1 + dd));
}

template <bool RECT>
HXR_DEV int light_num_samples(const hxr_light& L) { return RECT && L.type == HXR_LIGHT_RECT ? L.x_subd * L.y_subd : 1; }

// ------------------------------------------------------------------------------------------------
// textures / environment
// ------------------------------------------------------------------------------------------------
HXR_DEV f3 bitmap_bilinear(const HxrBitmapDev& b, float x, float y)   // Bitmap::getFilteredPixel, src/bitmap.cpp:66-81
{
	int w = b.width, h = b.height;
	if (w <= 0 || h <= 0 || x < 0 || x >= w || y < 0 || y >= h) return mkf(0, 0, 0);
	int tx = (int)floorf(x), ty = (int)floorf(y);
	if (tx < 0 || ty < 0) return mkf(0, 0, 0);
	int tx1 = (tx + 1) % w, ty1 = (ty + 1) % h;
	float p = x - tx, q = y - ty;
	const HXR_GLOBAL float* t = in_hbm(b.texels);
	return ldf(t + ((size_t)ty * w + tx) * 3) * ((1.0f - p) * (1.0f - q)) + ldf(t + ((size_t)ty * w + tx1) * 3) * (p * (1.0f - q)) +
	       ldf(t + ((size_t)ty1 * w + tx) * 3) * ((1.0f - p) * q) + ldf(t + ((size_t)ty1 * w + tx1) * 3) * (p * q);
}

template <bool COUNT, unsigned F>
HXR_DEVX f3 texture_sample(const HxrSceneDev& S, int ti, d3 dir, const Surface& sf, Tally<COUNT>& tl)
{
	const hxr_texture& t = S.textures[ti];
	switch (t.type) {
		case HXR_TEX_CHECKER: {   // src/shading.cpp:50-55
			int u1 = (int)floor(sf.u / t.scaling), v1 = (int)floor(sf.v / t.scaling);
			return ((u1 + v1) % 2 == 0) ? ldf(t.color1) : ldf(t.color2);
		}
		case HXR_TEX_BITMAP: {    // src/shading.cpp:147-157
			const HxrBitmapDev& b = S.bitmaps[t.bitmap];
			float u = (float)(sf.u / t.scaling), v = (float)(sf.v / t.scaling);
			u -= floorf(u);
			v -= floorf(v);
			u *= b.width;
			v *= b.height;
			tl.add(HC_TEX);
			return bitmap_bilinear(b, u, v);
		}
		case HXR_TEX_FRESNEL: if constexpr ((F & HXR_F_REFRACT) != 0) {   // src/shading.cpp:326-343, float arithmetic
			float eta = (float)t.ior;
			float ndi = (float)dot(dir, sf.norm);
			if (ndi > 0) eta = 1 / eta; else ndi = -ndi;
			double ratio = (double)((1.0f - eta) / (1.0f + eta));
			float f = (float)(ratio * ratio);
			float x = 1 - ndi;
			float fr = f + (1 - f) * powf(x, 5.0f);
			return mkf(fr, fr, fr);
		} else return mkf(0, 0, 0);
		default: return mkf(0, 0, 0);
	}
}

template <bool COUNT, unsigned F>
HXR_DEVX void texture_bump(const HxrSceneDev& S, int ti, Surface& sf, Tally<COUNT>& tl)
{
	const hxr_texture& t = S.textures[ti];
	if (t.type == HXR_TEX_BUMPTEX) {   // BumpTexture::modifyNormal, src/shading.cpp:345-356
		const HxrBitmapDev& b = S.bitmaps[t.bitmap];
		float x = (float)fmod(sf.u * t.scaling * b.width, (double)b.width);
		float y = (float)fmod(sf.v * t.scaling * b.height, (double)b.height);
		tl.add(HC_TEX);
		f3 bump = bitmap_bilinear(b, x, y);
		float dx = (float)(bump.x * t.strength), dy = (float)(bump.y * t.strength);
		sf.norm = unit(sf.norm + (sf.dNdx * (double)dx + sf.dNdy * (double)dy));
	} else if ((F & HXR_F_REFRACT) != 0 && t.type == HXR_TEX_BUMPS) {   // Bumps::modifyNormal, src/shading.cpp:363-377
		float strength = (float)t.strength;
		if (strength > 0) {
			const float fx[3] = {0.5f, 1.21f, 1.9f}, fz[3] = {0.4f, 1.13f, 1.81f}, amp[3] = {0.1f, 0.08f, 0.05f};
			const float fm = 0.2f;
			double dx = 0, dy = 0;
#pragma unroll
			for (int i = 0; i < 3; i++) {
				dx += sin((fm * fx[i]) * sf.u) * amp[i] * strength;
				dy += sin((fm * fz[i]) * sf.v) * amp[i] * strength;
			}
			sf.norm = unit(sf.norm + (sf.dNdx * dx + sf.dNdy * dy));
		}
	}
}

// CubemapEnvironment::getEnvironment, src/environment.cpp:54-81
template <bool COUNT>
HXR_DEVX f3 environment_lookup(const HxrSceneDev& S, d3 dir, Tally<COUNT>& tl)
{
	if (!S.environment.loaded) return mkf(0, 0, 0);
	int dim = 0;
	double big = fabs(dir.x);
	if (fabs(dir.y) > big) { dim = 1; big = fabs(dir.y); }
	if (fabs(dir.z) > big) dim = 2;
	double lead = comp(dir, dim);
	d3 s = dir * (1.0 / fabs(lead));
	int face = (lead > 0 ? 3 : 0) + dim;
	double fx, fy;
	switch (face) {
		case 0: fx = s.z; fy = -s.y; break;
		case 1: fx = s.x; fy = -s.z; break;
		case 2: fx = s.x; fy = s.y; break;
		case 3: fx = -s.z; fy = -s.y; break;
		case 4: fx = s.x; fy = s.z; break;
		default: fx = s.x; fy = -s.y; break;
	}
	const HxrBitmapDev& b = S.bitmaps[S.env_sides[face]];
	tl.add(HC_ENV);
	return bitmap_bilinear(b, (float)((fx + 1) * 0.5 * (b.width - 1)), (float)((fy + 1) * 0.5 * (b.height - 1)));
}

// ------------------------------------------------------------------------------------------------
// scene queries and the per-sample integrator
// ------------------------------------------------------------------------------------------------
// One query = the node scan shared by TraceContext::raycast (src/main.cpp:69-75: keep the node whose
// world-space hit distance is smallest) and visible() (src/main.cpp:179-184: stop at the first node
// hit closer than |AB|).  Both compare `info.dist < limit`; raycast lowers the limit as it goes,
// visible() leaves the scan at the first success.  Returns the node index or -1.
template <bool COUNT, unsigned F>
HXR_DEV int scene_query(const HxrSceneDev& S, d3 o, d3 d, double limit, bool any, GeomHit& bg, double& bw, const Stacks& st, Tally<COUNT>& tl)
{
	int node = -1;
	bw = limit;
	for (int ni = 0; ni < S.num_nodes; ni++) {
		GeomHit g;
		double w;
		if (node_query<COUNT, F>(S, ni, o, d, bw, g, w, st, tl) && w < bw) {
			bw = w;
			bg = g;
			node = ni;
			if (any) break;
		}
	}
	return node;
}

struct RayItem {
	d3 o, d;
	f3 weight;       // Whitted: product of the blend / reflectance factors on the way here; path tracing: throughput
	int depth;
	uint32_t flags, key;
};

// Where a deferring kernel variant appends secondary rays (wave-uniform; see HxrLaunch::q_*)
struct DeferCtx {
	HxrRayRec* records;    // this packet's sub-queue
	uint32_t* count;       // its append counter
	uint32_t* overflow;    // raised if a record would not fit (cannot happen: the host sizes the queue for the worst case)
	uint32_t capacity;
};

// Append this lane's ray.  Called from divergent code: the lanes that reach it together (same shader, same work-list step) are
// numbered with ballot + prefix count, the first of them reserves the slots with ONE atomic, each stores its own 80-byte record.
HXR_DEV void emit_ray(const DeferCtx& q, d3 o, d3 d, f3 weight, int depth, uint32_t flags, uint32_t key, uint32_t pixel, uint32_t sample)
{
	const unsigned long long group = hxr_ballot(true);
	const int lane = threadIdx.x & 63;
	const uint32_t rank = (uint32_t)__popcll(group & ((1ull << lane) - 1ull));
	uint32_t base = 0;
	if (rank == 0) base = atomicAdd(q.count, (uint32_t)__popcll(group));
	base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);   // rank 0 is the first active lane
	const uint32_t slot = base + rank;
	if (slot >= q.capacity) { *q.overflow = 1u; return; }
	HxrRayRec rec;
	rec.o[0] = o.x; rec.o[1] = o.y; rec.o[2] = o.z;
	rec.d[0] = d.x; rec.d[1] = d.y; rec.d[2] = d.z;
	rec.weight[0] = weight.x; rec.weight[1] = weight.y; rec.weight[2] = weight.z;
	rec.key = key;
	rec.pixel = pixel;
	rec.sample_depth = (sample & 0xffffu) | ((uint32_t)(depth & 0xff) << 16) | ((flags & 0xffu) << 24);
	rec.pad_[0] = rec.pad_[1] = 0u;
	q.records[slot] = rec;
}

// A pending entry of the per-lane ray stack.  With glossy Reflection compiled in, an entry can also be a
// GENERATOR: `left` more samples of shader `shader` are still to be shot from r.o (surface point), with r.d the
// incoming direction and n the face-forwarded normal (Reflection::computeColor, src/shading.cpp:165-184) -- the
// reference loops over numSamples recursive raytrace() calls; here the loop is unrolled over stack pops.
template <bool GLOSSY> struct StackItem;
template <> struct StackItem<false> { RayItem r; };
template <> struct StackItem<true> { RayItem r; d3 n; int left, shader; };

struct RngCtx {
	uint64_t seed;
	uint32_t pixel, sample;
};

HXR_DEV bool refract_dir(d3 i, d3 n, float ior, d3& out)   // refract(), src/shading.cpp:208-214 (float arithmetic)
{
	float ndi = (float)dot(i, n);
	float k = 1 - (ior * ior) * (1 - ndi * ndi);
	if (k < 0) return false;
	out = i * (double)ior - n * (double)(ior * ndi + sqrtf(k));
	return true;
}

HXR_DEV d3 hemisphere(double u, double v, d3 normal)   // hemisphereSample, src/util.cpp:109-129
{
	double theta = 2 * HXR_PI * u;
	double cphi = 2 * v - 1;
	double sphi = sqrt(1 - cphi * cphi);
	d3 dd = mk3(cos(theta) * sphi, cphi, sin(theta) * sphi);
	if (dot(dd, normal) < 0) dd = -dd;
	return dd;
}

// What stays of an IntersectionInfo once the bump map has been applied
struct HitPoint {
	d3 ip, norm;
	double u, v;
};

// The integrator.  raytrace() <-> Shader::computeColor() (src/main.cpp:105-113, src/shading.cpp) and
// pathtrace() (src/main.cpp:115-169) are mutually recursive in the reference and call visible() from
// inside the shaders.  Here one sample is a small state machine around ONE scene_query() site per loop
// turn, so that all lanes of the wave -- whether they need the closest hit of a camera / secondary ray
// or the any-hit of a shadow ray -- traverse together:
//   Q_CLOSEST  the current ray `cur` needs raycast()                                (src/main.cpp:63-103)
//   Q_SHADOW   a light sample of the shader being evaluated needs visible()          (src/main.cpp:171-190)
// Between two queries a lane finishes whatever needs no ray: light hits, environment, textures, the
// Layered expansion (src/shading.cpp:315-324), pushing Reflection / Refraction rays (:159-232) with their
// weights, BRDF spawn / eval for path tracing (:93-109,189-263).
//
// Table records (node, shader, light, texture) are selected by per-lane indices here.  Reading them with
// those indices would pull whole records -- a 240-byte Transform per node or light -- into vector registers
// of every lane.  Instead each distinct index present in the wave is handled in turn ("waterfall":
// readfirstlane picks an index, the lanes that hold it run the body with a wave-uniform index, the record
// arrives through scalar loads); a wave of a Whitted frame holds one to three distinct values in practice.
// The rays of a path tracer scatter after the first bounce -- a wave then holds nearly every node and shader of
// the scene and the waterfall would run its body once per value -- so the GI variants read the records with
// per-lane indices instead (`gather_records`): one pass, vector loads from tables that sit in L1.
enum { Q_NONE = 0, Q_CLOSEST = 1, Q_SHADOW = 2 };

#define HXR_WATERFALL(var, uni)                                                        \
	for (bool wf_done_ = false; !wf_done_;)                                            \
		for (const int uni = gather_records ? (var) : __builtin_amdgcn_readfirstlane(var); !wf_done_ && (var) == uni; wf_done_ = true)

// How the turn body (hxr_turn_body.inc) is compiled into a variant: see that file
template <unsigned F>
constexpr bool turn_through_lambda = (F & HXR_F_GI) != 0 || (F & HXR_F_MESH) == 0;

template <bool COUNT, unsigned F>
HXR_DEV f3 trace_rays(const HxrSceneDev& S, const RayItem& start, const RngCtx& rc, const Stacks& st, Tally<COUNT>& tl, const DeferCtx& dq)
{
	constexpr bool gi = (F & HXR_F_GI) != 0;   // the launcher picks the variant that matches settings.gi
	constexpr bool glossy = (F & HXR_F_EXTRA) != 0;
	constexpr bool gather_records = gi;
	constexpr bool rect = (F & HXR_F_RECT) != 0;   // RectLight code only where the scene has one
	constexpr bool defer = (F & HXR_F_DEFER) != 0; // secondary rays go to the queue: `top` stays 0, the stack is never touched
	// Whitted variants without RectLights: a light sample is the PointLight's position and colour -- functions of the light's index alone --
	// so they are read from the light's record where they are used instead of living in nine vector registers across the traversal
	constexpr bool lights_from_table = !gi && !rect;
	constexpr bool one = (F & HXR_F_STACK1) != 0;  // at most one pending ray: it waits in LDS (st.pend)
	constexpr int lds_slots = (F & HXR_F_LDSSTACK) != 0 ? HXR_LDS_RAY_SLOTS : 0;   // stack entries 0 .. lds_slots-1 live in LDS (st.pend), the rest in `stack`
	static_assert(lds_slots == 0 || (!gi && !glossy && !defer && !one), "the LDS stack entries are for the in-place Whitted variants without glossy generators");
	static_assert(!defer || (!gi && !glossy), "deferral is for the Whitted variants without glossy generators");
	static_assert(!one || (!gi && !glossy && !defer), "the one-slot stack is for the in-place Whitted variants");
	StackItem<glossy> stack[(defer || one) ? 1 : HXR_RAY_STACK - lds_slots];
	const int pend_lane = threadIdx.x & 63;
	auto pend_store = [&](const RayItem& r, int slot = 0) {   // HXR_F_STACK1 / HXR_F_LDSSTACK: a pending ray, one LDS row per dword (conflict-free)
		uint32_t* p = st.pend + slot * (HXR_PEND_DWORDS * 64) + pend_lane;
		const double v[6] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z};
#pragma unroll
		for (int k = 0; k < 6; k++) {
			const unsigned long long b = __builtin_bit_cast(unsigned long long, v[k]);
			p[(2 * k) * 64] = (uint32_t)b;
			p[(2 * k + 1) * 64] = (uint32_t)(b >> 32);
		}
		p[12 * 64] = __builtin_bit_cast(uint32_t, r.weight.x);
		p[13 * 64] = __builtin_bit_cast(uint32_t, r.weight.y);
		p[14 * 64] = __builtin_bit_cast(uint32_t, r.weight.z);
		p[15 * 64] = (uint32_t)r.depth;
		p[16 * 64] = r.flags;
		p[17 * 64] = r.key;
	};
	auto pend_load = [&](int slot = 0) {
		const uint32_t* p = st.pend + slot * (HXR_PEND_DWORDS * 64) + pend_lane;
		double v[6];
#pragma unroll
		for (int k = 0; k < 6; k++) v[k] = __builtin_bit_cast(double, ((unsigned long long)p[(2 * k + 1) * 64] << 32) | p[(2 * k) * 64]);
		return RayItem{mk3(v[0], v[1], v[2]), mk3(v[3], v[4], v[5]),
		               mkf(__builtin_bit_cast(float, p[12 * 64]), __builtin_bit_cast(float, p[13 * 64]), __builtin_bit_cast(float, p[14 * 64])),
		               (int)p[15 * 64], p[16 * 64], p[17 * 64]};
	};
	int sh_stack[HXR_SHADER_STACK];
	f3 w_stack[HXR_SHADER_STACK];
	int top = 0, stop = 0;
	f3 acc = mkf(0, 0, 0);
	RayItem cur = start;
	auto pop_ray = [&]() {   // top > 0: the next pending ray becomes `cur`
		if constexpr (one) {
			cur = pend_load();
			top = 0;
			return;
		}
		if constexpr (lds_slots > 0) {
			if (top <= lds_slots) {
				cur = pend_load(top - 1);
				top--;
				return;
			}
		}
		StackItem<glossy>& e = stack[top - 1 - lds_slots];
		if constexpr (glossy) {
			if (e.left > 0) {
				const hxr_shader& sh = S.shaders[e.shader];
				const int k = sh.num_samples - e.left;
				const float scaling = powf(10.0f, 2 - 8 * sh.glossiness);
				// othonormedBasis, src/vector.h:194-207
				d3 t0 = mk3(0.577350269, 0.577350269, 0.577350269), t1 = mk3(0.267261242, -0.534522484, 0.801783726);
				d3 tv = fabs(dot(e.n, t0)) > 0.9 ? t1 : t0;
				d3 bu = unit(cross(tv, e.n));
				d3 bv = cross(bu, e.n);
				d3 refl = e.n;
				for (uint32_t attempt = 0;; attempt++) {   // `do { ... } while (dot(reflected, n) < 0)`, bounded: 16 turns, as the oracle (both loops are unbounded in the reference; a disk draw fails 16 times in a row with p = 2e-11)
					double x = 0, y = 0;
					uint32_t inner = 0;
					do {                                   // unitDiskSample, src/util.cpp:101-107, bounded
						hxr_uniform2(hxr_philox(rc.seed, rc.pixel, rc.sample, e.r.key, HXR_DIM_GLOSSY + (uint32_t)k * 256u + ((attempt * 16u + inner) & 255u)), x, y);
						x = x * 2 - 1;
						y = y * 2 - 1;
						inner++;
					} while (x * x + y * y > 1 && inner < 16);
					x *= scaling;
					y *= scaling;
					d3 mn = unit(e.n + (bu * x + bv * y));
					refl = reflect3(e.r.d, mn);
					if (!(dot(refl, e.n) < 0) || attempt >= 15) break;
				}
				cur = e.r;
				cur.d = refl;
				cur.key = hxr_child_key(e.r.key, e.shader, (uint32_t)k);
				if (--e.left == 0) top--;
				return;
			}
		}
		cur = e.r;
		top--;
	};
	int want = Q_CLOSEST;
	// context of the hit being shaded
	HitPoint hp{};
	int cur_shader = -1;          // shader whose light loop is running (Whitted) / shader of the hit (path)
	f3 cur_w = mkf(0, 0, 0);      // weight of that shader instance
	f3 base = mkf(0, 0, 0), direct = mkf(0, 0, 0), sum = mkf(0, 0, 0), lc = mkf(0, 0, 0);
	d3 lp = mk3(0, 0, 0);         // light sample position
	int li = 0, ls = 0;           // light / sample cursor
	bool in_lights = false;

	bool shadow_turn = false;   // wave-uniform (path tracing only, see below)
	if constexpr (turn_through_lambda<F>) {
		auto turn = [&]() __attribute__((always_inline)) {
#define HXR_TURN_NEXT return
#include "hxr_turn_body.inc"
#undef HXR_TURN_NEXT
		};
		while (want != Q_NONE) turn();
	} else {
		while (want != Q_NONE) {
#define HXR_TURN_NEXT continue
#include "hxr_turn_body.inc"
#undef HXR_TURN_NEXT
		}
	}
	return acc;
}

// ------------------------------------------------------------------------------------------------
// camera (Camera::getScreenRay src/camera.cpp:65-76, getDOFScreenRay :78-86) and traceSingleRay (src/main.cpp:227-249)
// ------------------------------------------------------------------------------------------------
template <bool ON_SQUARE>
HXR_DEV void camera_ray(const HxrSceneDev& S, int W, int H, double x, double y, double lu, double lv, double stereo, d3& o, d3& d)
{
	const hxr_camera& c = S.camera;
	d3 pos = ld3(c.pos), tl = ld3(c.top_left);
	o = pos;
	d3 through = tl + (ld3(c.top_right) - tl) * (x / (double)W) + (ld3(c.bottom_left) - tl) * (y / (double)H);
	d = unit(through - o);
	if (stereo != 0) o = o + ld3(c.right) * (stereo * c.stereo_separation);
	if (c.dof) {
		double M = c.focal_plane_dist / dot(ld3(c.front), d);
		d3 T = o + d * M;
		o = pos + ld3(c.right) * (lu * c.aperture_size) + ld3(c.up) * (lv * c.aperture_size);
		d = unit_lazy<ON_SQUARE>(T - o);
	}
}

template <bool COUNT, unsigned F>
HXR_DEV f3 trace_sample(const HxrSceneDev& S, int W, int H, double x, double y, const RngCtx& rc, const Stacks& st, Tally<COUNT>& tl, const DeferCtx& dq)
{
	double lu = 0, lv = 0;
	tl.add(HC_SAMPLES);
	if (S.camera.dof) {   // unitDiskSample, src/util.cpp:101-107
		for (uint32_t attempt = 0; attempt < 64; attempt++) {
			hxr_uniform2(hxr_philox(rc.seed, rc.pixel, rc.sample, 0, HXR_DIM_LENS + attempt), lu, lv);
			lu = lu * 2 - 1;
			lv = lv * 2 - 1;
			if (!(lu * lu + lv * lv > 1)) break;
		}
	}
	tl.template lap<PH_LENS>();
	d3 o, d;
	// the anaglyph pair is compiled into the catch-all variants only (the launcher sends stereo cameras there): a second
	// inlined copy of the integrator doubles the code of the specialised kernels for a mode none of the config scenes uses
	if ((F & HXR_F_EXTRA) == 0 || S.camera.stereo_separation == 0.0) {
		camera_ray<(F & HXR_F_GI) != 0>(S, W, H, x, y, lu, lv, 0, o, d);
		tl.template lap<PH_GEN>();
#ifdef HXR_EXP_NO_TRACE   // developer aid: what a frame costs without the integrator (sample generation, dispenser, sample sums, stores)
		return mkf((float)d.x, (float)d.y, (float)(o.x + o.y));
#endif
		f3 c = trace_rays<COUNT, F>(S, RayItem{o, d, mkf(1, 1, 1), 0, 0u, 0u}, rc, st, tl, dq);
		tl.template lap<PH_SHADE>();
		return c;
	}
	f3 eye[2];
	for (int e = 0; e < 2; e++) {   // anaglyph pair (src/main.cpp:237-247)
		camera_ray<(F & HXR_F_GI) != 0>(S, W, H, x, y, lu, lv, e == 0 ? -0.5 : +0.5, o, d);
		eye[e] = trace_rays<COUNT, F>(S, RayItem{o, d, mkf(1, 1, 1), 0, 0u, 0u}, rc, st, tl, dq);
	}
	float ml = intensity(eye[0]), mr = intensity(eye[1]);
	return mkf(ml + (eye[0].x - ml) * 0.3f, mr + (eye[1].y - mr) * 0.3f, mr + (eye[1].z - mr) * 0.3f);
}

// ------------------------------------------------------------------------------------------------
// work mapping
// ------------------------------------------------------------------------------------------------
// Pixels of a shard are numbered tile by tile; inside a tile in 8x8 blocks (row-major), inside a block
// along a Morton curve, so every aligned group of 4 / 16 / 64 consecutive pixels is a 2x2 / 4x4 / 8x8
// square of the image: the lanes of a wave stay spatially coherent for every power-of-two spp.
// value of the lane below (lane 0 keeps 0): DPP wave_shr:1
HXR_DEV float wave_shr1(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}

HXR_DEV void morton6(uint32_t m, int& x, int& y)
{
	x = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4);
	y = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
}

HXR_DEV bool shard_pixel_general(const HxrLaunch& P, uint32_t q, int& x, int& y);

HXR_DEV bool shard_pixel(const HxrLaunch& P, uint32_t q, int& x, int& y)
{
	const int ts = P.tile_size;
	// Tile size and pixels per packet both powers of two (the default 64 and the shards' 8; every power-of-two spp): shifts and masks, and ONE
	// division, made once for the wave -- the pixels of such a packet lie in one tile.  (Three samples per pixel make packets of 21 pixels,
	// which straddle tiles: they take the general path below.)
	if ((ts & (ts - 1)) == 0 && (P.pixels_per_chunk & (P.pixels_per_chunk - 1)) == 0) {
		const int sh = __builtin_ctz(ts);
		const uint32_t lt = q >> (2 * sh), in = q & ((1u << (2 * sh)) - 1u);
		if (lt >= (uint32_t)P.num_shard_tiles) return false;
		// (a tile holds a multiple of 64 pixels, a packet an aligned power of two of them): the tile's position is computed for the first lane's
		const uint32_t lt0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lt);
		const uint32_t t0 = lt0 * (uint32_t)P.shard_count + (uint32_t)P.shard_rank;
		int tx, ty;
		hxr_tile_position((int)t0, P.tiles_x, P.shard_count, &tx, &ty);
		if (lt != lt0) return false;   // the lanes of a render-kernel packet, and 64 consecutive pixel numbers, lie in ONE tile; callers whose lanes do not use shard_pixel_general
		const uint32_t blk = in >> 6;
		int mx, my;
		morton6(in & 63u, mx, my);
		x = (tx << sh) + (int)(blk & ((1u << (sh - 3)) - 1u)) * 8 + mx;
		y = (ty << sh) + (int)(blk >> (sh - 3)) * 8 + my;
		return x < P.W && y < P.H;
	}
	return shard_pixel_general(P, q, x, y);
}

// ... for any tile size, every lane for itself
HXR_DEV bool shard_pixel_general(const HxrLaunch& P, uint32_t q, int& x, int& y)
{
	uint32_t per_tile = (uint32_t)(P.tile_size * P.tile_size);
	uint32_t lt = q / per_tile, in = q - lt * per_tile;
	if (lt >= (uint32_t)P.num_shard_tiles) return false;
	uint32_t t = lt * (uint32_t)P.shard_count + (uint32_t)P.shard_rank;   // row-major tile number
	int tx, ty;
	hxr_tile_position((int)t, P.tiles_x, P.shard_count, &tx, &ty);
	int bpr = P.tile_size / 8;
	uint32_t blk = in >> 6;
	int mx, my;
	morton6(in & 63u, mx, my);
	x = tx * P.tile_size + (int)(blk % (uint32_t)bpr) * 8 + mx;
	y = ty * P.tile_size + (int)(blk / (uint32_t)bpr) * 8 + my;
	return x < P.W && y < P.H;
}

HXR_DEV float* out_pixel(const HxrLaunch& P, uint32_t q, int x, int y)
{
	if (P.out_tiles) {
		uint32_t per_tile = (uint32_t)(P.tile_size * P.tile_size);
		uint32_t lt = q / per_tile;
		int lx = x % P.tile_size, ly = y % P.tile_size;
		return P.out + ((size_t)lt * per_tile + (size_t)ly * P.tile_size + lx) * 3;
	}
	return (float*)((char*)P.out + (size_t)y * P.out_stride) + (size_t)x * 3;
}

// ------------------------------------------------------------------------------------------------
// the render kernel: persistent wavefronts, one ray packet (64 camera samples) per dispenser ticket
// ------------------------------------------------------------------------------------------------
#ifndef HXR_BLOCK
#define HXR_BLOCK 256
#endif
#ifndef HXR_SCHED_URGENT_DIV
#define HXR_SCHED_URGENT_DIV 8u   // a whole chunk costing more than 1/8 of a wave's even share of the frame starts in the first round
#endif
#define HXR_SCHED_WHOLE 0x80000000u   // schedule item flag: the whole chunk that starts at this pixel
#define HXR_SCHED_PART 0x40000000u    // schedule item flag: bits 22..29 = which HXR_PART_LANES samples, bits 0..21 = part slot
#define HXR_SCHED_QUARTER 0x20000000u // schedule item flag (not a PART): the HXR_QUARTER_LANES pixels that start at this pixel (1-spp packets only)
#ifndef HXR_QUARTER_LANES
#define HXR_QUARTER_LANES 16
#endif
#ifndef HXR_PART_LANES
#define HXR_PART_LANES 4
#endif
#define HXR_PART_SLOTS 16384          // pixels that can be split by samples in one frame
#ifndef HXR_SHARDS
#define HXR_SHARDS 8          // dispenser counters (power of two)
#endif
#define HXR_SHARD_STRIDE 32   // uint32 between two counters: one 128-B line each
#ifndef HXR_BATCH
#define HXR_BATCH 1           // packets per ticket (larger batches unbalance the waves: expensive packets are spatially clustered)
#endif
// Waves per SIMD the register allocator must leave room for (template parameter W of the kernel; measured on
// MI355X): kernels that walk kd-trees are fastest at 3 (168 VGPRs, ~220 spilled: the packet walk
// needs the latency hiding more than the registers) -- mesh-only Whitted, and the catch-all variants on scenes
// with meshes (12 % faster than at 2); the heightfield marcher keeps more per-lane state alive and is 18 % faster at 2
// (256 VGPRs, almost nothing spilled).  The sphere path tracer was 21 % faster at 2 in round 1 and, since its turns (trace_rays), is
// 16 % faster at 3 (re-measured at the end of round 2 for every variant: hxr_variants.h).
// M: the render mode when known at compile time (1 = Monte-Carlo, the throughput mode), -1 = HxrLaunch::mode at run time
template <bool COUNT, unsigned F, int W, int M>
__global__ __launch_bounds__(HXR_BLOCK, W) void hxr_render_kernel(
    const HxrSceneDev* __restrict__ Sp, const HxrNodeDev* __restrict__ a_nodes, const HxrMeshDev* __restrict__ a_meshes,
    const HxrHeightfieldDev* __restrict__ a_hfs, const HxrBitmapDev* __restrict__ a_bitmaps, const hxr_texture* __restrict__ a_textures,
    const hxr_shader* __restrict__ a_shaders, const hxr_layer* __restrict__ a_layers, const hxr_light* __restrict__ a_lights,
    const HxrKdSlim* __restrict__ a_kd, const HxrKdBox* __restrict__ a_kd_box, const HxrTriHot* __restrict__ a_hot, const int32_t* __restrict__ a_hot_ids,
    const hxr_geometry* __restrict__ a_geoms, HxrLaunch P)
{
	// The small tables arrive as separate `const __restrict__` kernel arguments: only then can the compiler
	// prove that the frame / counter stores of this kernel never clobber them, and fetch their wave-uniform
	// records (node transforms, shaders, lights) with scalar loads into SGPRs instead of 64 identical vector loads.
	HxrSceneDev S = *Sp;
	S.nodes = a_nodes;
	S.meshes = a_meshes;
	S.heightfields = a_hfs;
	S.bitmaps = a_bitmaps;
	S.textures = a_textures;
	S.shaders = a_shaders;
	S.layers = a_layers;
	S.lights = a_lights;
	S.kd = a_kd;
	S.kd_box = a_kd_box;
	S.hot = a_hot;
	S.hot_ids = a_hot_ids;
	S.geometries = a_geoms;
	__shared__ WaveStack lds_stack[HXR_BLOCK / 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr int pend_slots = (F & HXR_F_STACK1) != 0 ? 1 : ((F & HXR_F_LDSSTACK) != 0 ? HXR_LDS_RAY_SLOTS : 0);   // ray stack entries per lane kept in LDS
	__shared__ uint32_t lds_pend[pend_slots > 0 ? (HXR_BLOCK / 64) * pend_slots * HXR_PEND_DWORDS * 64 : 1];
	const Stacks stacks{&lds_stack[wave], LaneStack{nullptr, 0, nullptr}, lds_pend + wave * pend_slots * HXR_PEND_DWORDS * 64};
	Tally<COUNT> tl;
	tl.clear();
#ifdef HXR_PHASE_TIMERS
	__shared__ unsigned long long lds_clock[HXR_BLOCK / 64][PH_NUM + 1];
	tl.attach(lds_clock[wave]);
#endif
	const int mode_ = M >= 0 ? M : P.mode;
	const int spp = P.spp;
	const int lanes_per_pixel = spp < 64 ? spp : 64;          // samples of one pixel resident in the wave
	const int ppc = P.pixels_per_chunk;                         // 64 / spp for spp <= 64, else 1
	const int rounds = (spp + 63) / 64;
	// pass 3 sizes itself from the device-side count of flagged pixels (no host round trip between the passes)
	const uint32_t num_chunks = mode_ == 2 ? ((uint32_t)*P.aa_count + (uint32_t)ppc - 1u) / (uint32_t)ppc : P.num_chunks;
	// Packet dispenser.  One device-scope counter saturates near 88 tickets/us (MI355X_MICROARCH.md, row
	// "dequeue") -- 518,400 single-packet tickets of the headline frame would cost 6 ms by themselves -- so the
	// packets are split into HXR_SHARDS contiguous ranges with a counter each (one per XCD under the
	// dispatcher's round-robin placement, a speed matter only), and a ticket is a batch of HXR_BATCH packets.
	// A wave drains its home range, then helps with the others.
	const uint32_t per_shard = ((num_chunks + HXR_SHARDS * HXR_BATCH - 1) / (HXR_SHARDS * HXR_BATCH)) * HXR_BATCH;
	uint32_t shard = blockIdx.x & (HXR_SHARDS - 1), shards_done = 0, batch_pos = 0, batch_end = 0;
	// Adaptive schedule: pixels of chunks that were expensive in the previous frame come first, ONE pixel per packet.  An
	// expensive packet is one whose rays have scattered (reflections inside the crease of a mesh ...): the packet walk then
	// serves its lanes one after the other, so its duration grows with the number of live lanes -- up to 8 ms for one packet
	// of the headline frame, which no other wave can help with.  Split four ways and started first, the same pixels finish
	// in a quarter of the time while the cheap packets fill the machine.  The image does not depend on any of this.
	const bool scheduled = P.sched_counts != nullptr && mode_ != 2;   // pass 3 only takes a list of single pixels (see the AA host path)
	const uint32_t listed_a = P.sched_counts ? P.sched_counts[0] : 0u, listed_b = P.sched_counts ? P.sched_counts[1] : 0u;
	bool listed_left = listed_a + listed_b > 0;
	const uint32_t listed2 = P.sched2_items ? *P.sched2_count : 0u;   // the second list (HxrLaunch::sched2_items)
	bool listed2_left = listed2 > 0;
	bool listed2_now = listed2 > P.sched2_early_above;               // ... before the ordinary packets, or after them
#ifdef HXR_WAVE_TIMELINE   // developer aid (tools/wave_timeline.py): when each wave started and left, and its longest packet, on the constant 100 MHz clock
	const unsigned long long wt_start = __builtin_amdgcn_s_memrealtime();
	unsigned long long wt_longest = 0, wt_longest_at = 0, wt_packets = 0, wt_longest_kind = 0;   // kind: 1 listed, 2 one pixel, 4 sample part, 8 from the front of the list
#endif
	for (;;) {
		if (P.cancel) {   // the host raised the cancel word (hxr_render): no further packets; every wave leaves through the common exit below.
			// The word lives in pinned HOST memory mapped into the device (a copy engine or blit kernel could starve behind the persistent
			// waves of this very kernel): a system-scope load, ~2 us, paid only by frames that are watched.
			uint32_t stop = 0;
			if (lane == 0) stop = __hip_atomic_load(P.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (__builtin_amdgcn_readfirstlane(stop)) break;
		}
		uint32_t single_pixel = 0xffffffffu;   // != ~0: this packet renders only that pixel
		bool from_front = false;               // ... handed out from the front of the schedule: split pixels / sample parts, chunks long against the frame
		if (listed_left) {
			uint32_t t = 0;
			if (lane == 0) t = atomicAdd(P.sched_ticket, 1u);
			t = __builtin_amdgcn_readfirstlane(t);
			if (t < listed_a) { single_pixel = P.sched_items[t]; from_front = true; }
			else if (t - listed_a < listed_b) single_pixel = P.sched_items[P.sched_capacity - 1u - (t - listed_a)];
			else listed_left = false;
		}
		if (single_pixel == 0xffffffffu && listed2_left && listed2_now) {
			uint32_t t = 0;
			if (lane == 0) t = atomicAdd(P.sched2_ticket, 1u);
			t = __builtin_amdgcn_readfirstlane(t);
			if (t < listed2) { single_pixel = P.sched2_items[t]; from_front = true; }
			else listed2_left = false;
		}
		if (single_pixel == 0xffffffffu) {
			if (batch_pos >= batch_end) {
				bool got = false;
				while (shards_done < HXR_SHARDS) {
					uint32_t t = 0;
					if (lane == 0) t = atomicAdd(P.work_counter + shard * HXR_SHARD_STRIDE, (uint32_t)HXR_BATCH);
					t = __builtin_amdgcn_readfirstlane(t);
					const uint32_t lo = shard * per_shard, hi = min(lo + per_shard, num_chunks);
					if (lo < hi && t < hi - lo) {
						batch_pos = lo + t;
						batch_end = min(batch_pos + HXR_BATCH, hi);
						got = true;
						break;
					}
					shard = (shard + 1) & (HXR_SHARDS - 1);
					shards_done++;
				}
				if (!got && listed2_left && !listed2_now) { listed2_now = true; continue; }   // the ordinary packets are through: what is left of the second list
				if (!got) break;
			}
			if (scheduled && P.sched_flags[batch_pos]) { batch_pos++; continue; }   // its pixels were handed out from the list
		}
		tl.template lap<PH_DISPENSE>();
		const unsigned long long packet_t0 = __builtin_readcyclecounter();
#ifdef HXR_WAVE_TIMELINE
		const unsigned long long wt_packet_began = __builtin_amdgcn_s_memrealtime();
#endif
		const bool listed = single_pixel != 0xffffffffu;
		if (P.front_priority) {   // (wave-uniform) see HxrLaunch::front_priority
			if (from_front || (P.front_priority > 1 && listed)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
		}
		const bool part = listed && (single_pixel & HXR_SCHED_PART);        // HXR_PART_LANES samples of one pixel
		const uint32_t part_slot = single_pixel & 0x3fffffu, part_group = (single_pixel >> 22) & 0xffu;
		if (part) single_pixel = P.sched_part_pixels[part_slot];
		const bool quarter = listed && !part && (single_pixel & HXR_SCHED_QUARTER);   // a run of HXR_QUARTER_LANES pixels of a 1-spp chunk
		const bool single = listed && !quarter && !(single_pixel & HXR_SCHED_WHOLE);   // one pixel only; a listed WHOLE chunk is an ordinary packet
		single_pixel &= ~(HXR_SCHED_WHOLE | (part ? 0u : HXR_SCHED_QUARTER));
		const uint32_t chunk = listed ? single_pixel / (uint32_t)P.pixels_per_chunk : batch_pos++;
		const int slot = lane / lanes_per_pixel;               // which pixel of the chunk this lane works for
		const int sub = lane - slot * lanes_per_pixel;         // which sample of that pixel (first round)
		const int quarter_first = quarter ? (int)(single_pixel - chunk * (uint32_t)P.pixels_per_chunk) : 0;
		bool live = single ? slot == 0 : quarter ? (slot >= quarter_first && slot < quarter_first + HXR_QUARTER_LANES && slot < ppc) : slot < ppc;   // lanes that belong to a pixel of this packet
		// which pixel
		int x = 0, y = 0;
		uint32_t q = 0;
		if (mode_ == 2) {        // AA taps: aa_list holds frame pixel numbers, 4 taps each
			uint32_t e = chunk * (uint32_t)ppc + (uint32_t)slot;
			if (!single) live = live && e < (uint32_t)*P.aa_count;
			if (live) { int pix = single ? (int)single_pixel : P.aa_list[e]; x = pix % P.W; y = pix / P.W; }   // a listed item IS the frame pixel
		} else if (mode_ == 3) { // coarseRender (src/main.cpp:387-414): a "pixel" of this pass is a 24x24 block, (x, y) its top-left corner
			q = chunk * (uint32_t)ppc + (uint32_t)slot;
			const uint32_t blocks_x = (uint32_t)(P.W + 23) / 24u, blocks = blocks_x * ((uint32_t)(P.H + 23) / 24u);
			live = live && q < blocks;
			x = (int)(q % blocks_x) * 24;
			y = (int)(q / blocks_x) * 24;
		} else {
			q = single ? single_pixel : chunk * (uint32_t)ppc + (uint32_t)slot;
			uint32_t tile_pixels = (uint32_t)P.num_shard_tiles * (uint32_t)(P.tile_size * P.tile_size);
			if (q < tile_pixels) live = live && shard_pixel(P, q, x, y);
			else {                // halo pixels of a sharded adaptive-AA frame
				uint32_t e = q - tile_pixels;
				live = live && e < (uint32_t)P.num_extra_pixels;
				if (live) { int pix = P.extra_pixels[e]; x = pix % P.W; y = pix / P.W; }
			}
		}
		tl.template lap<PH_MAP>();
		DeferCtx dq{nullptr, nullptr, nullptr, 0u};
		if constexpr ((F & HXR_F_DEFER) != 0) {   // the sub-queue of this packet's dispenser range
			const uint32_t qs = min(chunk / max(per_shard, 1u), (uint32_t)(HXR_QUEUE_SHARDS - 1));
			dq = DeferCtx{P.q_records + (size_t)qs * P.q_capacity, P.q_ctl + qs * HXR_SHARD_STRIDE, P.q_ctl + 2 * HXR_QUEUE_SHARDS * HXR_SHARD_STRIDE, P.q_capacity};
		}
		if (mode_ == 0 && P.fov_radius > 0 && !listed && chunk * 64u < (uint32_t)P.num_shard_tiles * (uint32_t)(P.tile_size * P.tile_size)) {
			// Foveated pass 1 (src/main.cpp:284-311).  A chunk of this pass IS one of the reference's 8x8 blocks (64 pixels in Morton
			// order, lane 0 = its top-left pixel).  Far from the point of interest the block gets the mean of five rays, traced by the
			// first five lanes at COARSE_KERNEL's positions (src/main.cpp:262-268) and added up in the reference's order.
			const int bx = __builtin_amdgcn_readlane(x, 0), by = __builtin_amdgcn_readlane(y, 0);
			const int dxc = bx + 4 - P.vip_x, dyc = by + 4 - P.vip_y;
			if (dxc * dxc + dyc * dyc > P.fov_radius * P.fov_radius && hxr_ballot(live) != 0) {
				f3 c = mkf(0, 0, 0);
				if (lane < 5) {
					const int sx = bx + (lane == 2 ? 4 : ((lane == 1 || lane == 4) ? 6 : 2)), sy = by + (lane < 2 ? 2 : (lane == 2 ? 4 : 6));
					const RngCtx rcc{P.seed, (uint32_t)(sy * P.W + sx), 0u};
					c = trace_sample<COUNT, F>(S, P.W, P.H, (double)sx, (double)sy, rcc, stacks, tl, dq);
				}
				f3 mean = mkf(__shfl(c.x, 0, 64), __shfl(c.y, 0, 64), __shfl(c.z, 0, 64));
				for (int j = 1; j < 5; j++) mean = mean + mkf(__shfl(c.x, j, 64), __shfl(c.y, j, 64), __shfl(c.z, j, 64));
				mean = mean * 0.2f;
				if (live) {
					float* dst = out_pixel(P, q, x, y);
					dst[0] = mean.x;
					dst[1] = mean.y;
					dst[2] = mean.z;
				}
				if (P.tile_done) {
					const unsigned long long wrote = hxr_ballot(live);
					if (wrote && lane == (int)__builtin_ctzll(wrote)) {
						__threadfence_system();   // the reader is a copy engine on another stream: outside this agent
						atomicAdd(P.tile_done + q / (uint32_t)(P.tile_size * P.tile_size), (uint32_t)__popcll(wrote));
					}
				}
				continue;
			}
		}
		f3 sum = mkf(0, 0, 0);
		int repl_shift = 0;   // HXR_F_REPL: log2 of the replicas per ray of this packet (0: one sample per lane)
		if constexpr (replicating<COUNT, F>) {
			// which sample this lane traces: its own (lane = pixel slot x samples per pixel + sample) -- or, in a replicated packet, the sample of the group of
			// R lanes it belongs to
			int tsub = sub, tx = x, ty = y;
			bool tlive = live && (!part || (uint32_t)sub / HXR_PART_LANES == part_group);
			Stacks rstacks = stacks;
			const int rays = part ? HXR_PART_LANES : lanes_per_pixel;   // rays of a one-pixel packet
			if (rounds == 1 && (part || single) && mode_ != 3 && rays <= 16 && !P.no_replication) {
				repl_shift = rays <= 4 ? 4 : 2;
				rstacks.repl = 1 << repl_shift;
				const int ray = lane >> repl_shift;
				tsub = (part ? (int)part_group * HXR_PART_LANES : 0) + ray;
				tx = __builtin_amdgcn_readlane(x, 0);      // lane 0 is the first lane of the packet's one pixel
				ty = __builtin_amdgcn_readlane(y, 0);
				tlive = (hxr_ballot(live) & 1ull) != 0 && ray < rays && tsub < spp;
			}
			if (tlive) {
				const Stacks& stacks = rstacks;
				const int x = tx, y = ty, sub = tsub;   // (shadow the packet's own: the loop below is written in their terms)
				RngCtx rc{P.seed, (uint32_t)(y * P.W + x), 0u};
				for (int rd = 0; rd < rounds; rd++) {
					int s = sub + rd * 64;
					if (s >= spp) break;
					double sx = x, sy = y;
					if (mode_ == 1) {          // renderWithMonteCarlo: x + randDouble(), y + randDouble() (src/main.cpp:373-375)
						rc.sample = (uint32_t)s;
						double jx, jy;
						hxr_uniform2(hxr_philox(rc.seed, rc.pixel, rc.sample, 0, HXR_DIM_PIXEL_JITTER), jx, jy);
						sx = x + jx;
						sy = y + jy;
					} else if (mode_ == 3) {   // coarseRender: x + randDouble() * BLK_SIZE, y + randDouble() * BLK_SIZE (src/main.cpp:396)
						rc.sample = (uint32_t)s;
						double jx, jy;
						hxr_uniform2(hxr_philox(rc.seed, rc.pixel, rc.sample, 0, HXR_DIM_PREPASS), jx, jy);
						sx = x + jx * 24;
						sy = y + jy * 24;
					} else if (mode_ == 2) {   // pass 3 taps 1..4 of AA_KERNEL; `x + 0.6f` is float arithmetic (src/main.cpp:255-261,336)
						const float kx[4] = {0.6f, 0.3f, 0.0f, 0.6f}, ky[4] = {0.0f, 0.3f, 0.6f, 0.6f};
						rc.sample = (uint32_t)(s + 1);
						sx = (double)((float)x + kx[s]);
						sy = (double)((float)y + ky[s]);
					}
					f3 c = trace_sample<COUNT, F>(S, P.W, P.H, sx, sy, rc, stacks, tl, dq);
					sum = (rd == 0) ? c : sum + c;
				}
			}
			if (repl_shift) {   // (wave-uniform) hand every sample's colour to the lane that owns it in the ordinary layout: replica 0 of its group
				const int src = ((sub - (part ? (int)part_group * HXR_PART_LANES : 0)) << repl_shift) & 63;
				sum = mkf(__shfl(sum.x, src, 64), __shfl(sum.y, src, 64), __shfl(sum.z, src, 64));
				if (!(live && (!part || (uint32_t)sub / HXR_PART_LANES == part_group))) sum = mkf(0, 0, 0);   // lanes that own no sample of this packet
			}
		} else if (live && (!part || (uint32_t)sub / HXR_PART_LANES == part_group)) {
			// (the same loop as above, spelled out a second time: shared through a lambda it costs the headline kernel 33 more spilled scalar registers)
			RngCtx rc{P.seed, (uint32_t)(y * P.W + x), 0u};
			for (int rd = 0; rd < rounds; rd++) {
				int s = sub + rd * 64;
				if (s >= spp) break;
				double sx = x, sy = y;
				if (mode_ == 1) {          // renderWithMonteCarlo: x + randDouble(), y + randDouble() (src/main.cpp:373-375)
					rc.sample = (uint32_t)s;
					double jx, jy;
					hxr_uniform2(hxr_philox(rc.seed, rc.pixel, rc.sample, 0, HXR_DIM_PIXEL_JITTER), jx, jy);
					sx = x + jx;
					sy = y + jy;
				} else if (mode_ == 3) {   // coarseRender: x + randDouble() * BLK_SIZE, y + randDouble() * BLK_SIZE (src/main.cpp:396)
					rc.sample = (uint32_t)s;
					double jx, jy;
					hxr_uniform2(hxr_philox(rc.seed, rc.pixel, rc.sample, 0, HXR_DIM_PREPASS), jx, jy);
					sx = x + jx * 24;
					sy = y + jy * 24;
				} else if (mode_ == 2) {   // pass 3 taps 1..4 of AA_KERNEL; `x + 0.6f` is float arithmetic (src/main.cpp:255-261,336)
					const float kx[4] = {0.6f, 0.3f, 0.0f, 0.6f}, ky[4] = {0.0f, 0.3f, 0.6f, 0.6f};
					rc.sample = (uint32_t)(s + 1);
					sx = (double)((float)x + kx[s]);
					sy = (double)((float)y + ky[s]);
				}
				f3 c = trace_sample<COUNT, F>(S, P.W, P.H, sx, sy, rc, stacks, tl, dq);
				sum = (rd == 0) ? c : sum + c;
			}
		}
		if (part) {
			// leave this part's sample colours where the pixel's last part will find them; only that one goes on to the sum
			float* side = P.sched_side + (size_t)part_slot * (size_t)lanes_per_pixel * 3;
			if (live && (uint32_t)sub / HXR_PART_LANES == part_group) {
				__hip_atomic_store(side + sub * 3 + 0, sum.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(side + sub * 3 + 1, sum.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(side + sub * 3 + 2, sum.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			__threadfence();
			uint32_t before = 0;
			if (lane == 0) before = __hip_atomic_fetch_add(P.sched_part_done + part_slot, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			before = __builtin_amdgcn_readfirstlane(before);
			const unsigned long long dt = (__builtin_readcyclecounter() - packet_t0) * (unsigned long long)(ppc * P.sched_part_groups);
			if (lane == 0) atomicMax(P.sched_cost + chunk, (uint32_t)(dt > 0xffffffffull ? 0xffffffffull : dt));
			if (before + 1u != (uint32_t)P.sched_part_groups) continue;
			__threadfence();
			if (live) {   // every lane of the pixel takes its sample's colour (past this CU's L1: other waves wrote them)
				sum.x = __hip_atomic_load(side + sub * 3 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				sum.y = __hip_atomic_load(side + sub * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				sum.z = __hip_atomic_load(side + sub * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		// Combine the samples of each pixel in sample order (the reference adds them one by one, src/main.cpp:372-377).
		// The lanes of a pixel are adjacent, so the running sum walks through them with one wave-shift per sample:
		// after step k the lane of sample k holds ((s0 + s1) + ...) + sk.  No LDS round trip, all lanes in lockstep.
		const bool in_tile = mode_ != 2 && mode_ != 3 && q < (uint32_t)P.num_shard_tiles * (uint32_t)(P.tile_size * P.tile_size);
		float* dst = nullptr;
		f3 acc = mkf(0, 0, 0);
		if (live && (sub == 0 || sub == lanes_per_pixel - 1)) {
			dst = in_tile ? out_pixel(P, q, x, y) : (float*)((char*)P.out + (size_t)y * P.out_stride) + (size_t)x * 3;
			if (mode_ == 2 && sub == 0) acc = mkf(dst[0], dst[1], dst[2]);   // pass 3 adds onto the pass-1 colour
		}
		acc = acc + sum;
		for (int k = 1; k < lanes_per_pixel; k++) {
			f3 left = mkf(wave_shr1(acc.x), wave_shr1(acc.y), wave_shr1(acc.z));
			if (sub == k) acc = left + sum;
		}
		if (live && sub == lanes_per_pixel - 1) {
			if (mode_ == 1 || mode_ == 3) acc = acc * (1.0f / spp);
			else if (mode_ == 2) acc = acc * (1.0f / 5);
			if (mode_ == 3) {   // drawRect(r, sum * mul) (src/main.cpp:399): the block's mean fills the block (a display-only preview: one lane writes it)
				const int x1 = min(x + 24, P.W), y1 = min(y + 24, P.H);
				for (int py = y; py < y1; py++) {
					float* row = (float*)((char*)P.out + (size_t)py * P.out_stride);
					for (int px = x; px < x1; px++) { row[px * 3 + 0] = acc.x; row[px * 3 + 1] = acc.y; row[px * 3 + 2] = acc.z; }
				}
			} else {
				dst[0] = acc.x;
				dst[1] = acc.y;
				dst[2] = acc.z;
			}
		}
		if (P.tile_done && in_tile) {   // progress for hxr_render's callback: the pixels of a packet lie in one tile
			const unsigned long long wrote = hxr_ballot(live && sub == lanes_per_pixel - 1);
			if (wrote && lane == (int)__builtin_ctzll(wrote)) {
				__threadfence_system();   // the pixel stores before the count; system scope: the tile is read by a host-initiated copy on another stream while this kernel runs
				atomicAdd(P.tile_done + q / (uint32_t)(P.tile_size * P.tile_size), (uint32_t)__popcll(wrote));
			}
		}
		if (scheduled && !part && lane == 0) {   // what this chunk cost (a single pixel stands for its ppc-pixel chunk)
			unsigned long long dt = (__builtin_readcyclecounter() - packet_t0) * (single ? (unsigned long long)ppc : quarter ? (unsigned long long)(ppc / HXR_QUARTER_LANES) : 1ull);
			atomicMax(P.sched_cost + chunk, (uint32_t)(dt > 0xffffffffull ? 0xffffffffull : dt));
		}
		tl.template lap<PH_REDUCE>();
#ifdef HXR_WAVE_TIMELINE
		{
			const unsigned long long now = __builtin_amdgcn_s_memrealtime(), began = wt_packet_began;
			wt_packets++;
			if (now - began > wt_longest) { wt_longest = now - began; wt_longest_at = began; wt_longest_kind = (listed ? 1u : 0u) | (single ? 2u : 0u) | (part ? 4u : 0u) | (from_front ? 8u : 0u); }
		}
#endif
#if defined(HXR_PHASE_TIMERS) && defined(HXR_PACKET_CLOCKS)
		{   // distribution of packet durations instead of the phase sums: slots = {sum, max, count, three thresholds}
			const unsigned long long dt = __builtin_readcyclecounter() - packet_t0;
			if (lane == 0) {
				tl.ph[0] += dt;
				if (dt > tl.ph[1]) tl.ph[1] = dt;
				tl.ph[2] += 1;
				if (dt > 1000000ull) tl.ph[3] += 1;    // > 1 M shader clocks (~0.5 ms)
				if (dt > 4000000ull) tl.ph[4] += 1;    // > 4 M (~2 ms)
				if (dt > 250000ull) tl.ph[5] += 1;     // > 250 k (~0.12 ms)
			}
#ifdef HXR_PACKET_CLOCK_IMAGE
			if (live && sub == lanes_per_pixel - 1) { dst[0] = (float)dt; dst[1] = (float)dt; dst[2] = (float)dt; }   // the frame becomes a cost map
#endif
		}
#endif
	}
#ifdef HXR_WAVE_TIMELINE
	if (P.sched_side && lane == 0) {   // (the far end of the side buffer of the sample parts: 8 MB in, four words per wave)
		// (the replicating blocks launched beside the main kernel write behind its 8192 slots)
		unsigned long long* w = (unsigned long long*)((char*)P.sched_side + (8u << 20)) + 4 * ((size_t)blockIdx.x * (HXR_BLOCK / 64) + wave + (replicating<COUNT, F> ? 8192u : 0u));
		w[0] = wt_start; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = wt_longest; w[3] = (wt_longest_at & ((1ull << 48) - 1)) | ((wt_packets & 0xfffull) << 48) | (wt_longest_kind << 60);
	}
#endif
#ifdef HXR_PHASE_TIMERS
	if (P.counters && lane == 0) {
#ifdef HXR_PACKET_CLOCKS
		atomicAdd(&P.counters[HC_NUM + 0], (unsigned long long)tl.ph[0]);
		atomicMax(&P.counters[HC_NUM + 1], (unsigned long long)tl.ph[1]);
		for (int i = 2; i < PH_NUM; i++) atomicAdd(&P.counters[HC_NUM + i], (unsigned long long)tl.ph[i]);
#else
		for (int i = 0; i < PH_NUM; i++) atomicAdd(&P.counters[HC_NUM + i], (unsigned long long)tl.ph[i]);
#endif
	}
#endif
	if (COUNT && P.counters) {
		for (int i = 0; i < HC_NUM; i++) {
			uint32_t v = tl.c[i];
			for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
			if (lane == 0 && v) atomicAdd(&P.counters[i], (unsigned long long)v);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// the continuation kernel: secondary rays deferred by the render kernel, 64 to a wave
// ------------------------------------------------------------------------------------------------
// A deferring render kernel (HXR_F_DEFER) leaves the Reflection / Refraction rays of its hits in HXR_QUEUE_SHARDS sub-queues.
// Here persistent waves take 64 consecutive records at a time -- rays of neighbouring pixels, whatever packet they came from,
// so a wave is full where the render kernel would have gone on with the few lanes of a packet that hit something shiny --
// and every lane runs raytrace() on its ray to the end (further bounces wait on the per-lane stack as before).  What the ray
// contributes to its pixel is `weight x colour` by linearity; it is added into HxrLaunch::q_accum as 64-bit fixed point, so the
// sum does not depend on the order in which waves finish.  F: the in-place variant of the integrator (no HXR_F_DEFER).
#define HXR_LANE_BLOCK 128   // threads per block of the per-lane-walk continuation kernels: two waves share 2 x entries x 256 B of LDS
template <bool COUNT, unsigned F, int W>
__global__ __launch_bounds__(((F & HXR_F_LANEWALK) != 0 ? HXR_LANE_BLOCK : HXR_BLOCK), W) void hxr_continue_kernel(
    const HxrSceneDev* __restrict__ Sp, const HxrNodeDev* __restrict__ a_nodes, const HxrMeshDev* __restrict__ a_meshes,
    const HxrHeightfieldDev* __restrict__ a_hfs, const HxrBitmapDev* __restrict__ a_bitmaps, const hxr_texture* __restrict__ a_textures,
    const hxr_shader* __restrict__ a_shaders, const hxr_layer* __restrict__ a_layers, const hxr_light* __restrict__ a_lights,
    const HxrKdSlim* __restrict__ a_kd, const HxrKdBox* __restrict__ a_kd_box, const HxrTriHot* __restrict__ a_hot, const int32_t* __restrict__ a_hot_ids,
    const hxr_geometry* __restrict__ a_geoms, HxrLaunch P)
{
	static_assert((F & HXR_F_DEFER) == 0, "the continuation pass traces in place");
	constexpr bool lanewalk = (F & HXR_F_LANEWALK) != 0;
	constexpr int block = lanewalk ? HXR_LANE_BLOCK : HXR_BLOCK;
	HxrSceneDev S = *Sp;
	S.nodes = a_nodes;
	S.meshes = a_meshes;
	S.heightfields = a_hfs;
	S.bitmaps = a_bitmaps;
	S.textures = a_textures;
	S.shaders = a_shaders;
	S.layers = a_layers;
	S.lights = a_lights;
	S.kd = a_kd;
	S.kd_box = a_kd_box;
	S.hot = a_hot;
	S.hot_ids = a_hot_ids;
	S.geometries = a_geoms;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	__shared__ WaveStack lds_stack[lanewalk ? 1 : block / 64];   // the packet walk's stacks (unused by the per-lane walk)
	extern __shared__ uint32_t lds_lane_rows[];                   // the per-lane walk's columns: P.lane_stack_entries rows x 64 lanes per wave
	const Stacks stacks{&lds_stack[lanewalk ? 0 : wave],
	                    LaneStack{lds_lane_rows + (size_t)wave * (size_t)P.lane_stack_entries * 64, P.lane_stack_entries, P.q_ctl + 2 * HXR_QUEUE_SHARDS * HXR_SHARD_STRIDE},
	                    nullptr};
	Tally<COUNT> tl;
	tl.clear();
#ifdef HXR_PHASE_TIMERS
	__shared__ unsigned long long lds_clock[block / 64][PH_NUM + 1];
	tl.attach(lds_clock[wave]);
#endif
	uint32_t shard = blockIdx.x & (HXR_QUEUE_SHARDS - 1), shards_done = 0;
	const DeferCtx none{nullptr, nullptr, nullptr, 0u};
	while (shards_done < HXR_QUEUE_SHARDS) {   // every wave leaves after at most HXR_QUEUE_SHARDS empty-handed tickets
		const uint32_t appended = min(P.q_ctl[shard * HXR_SHARD_STRIDE], P.q_capacity);
		uint32_t t = 0;
		const uint32_t ticket = P.cont_ticket > 0 ? (uint32_t)P.cont_ticket : 64u;   // rays per ticket (developer aid: fewer than a wave)
		if (lane == 0) t = atomicAdd(P.q_ctl + (HXR_QUEUE_SHARDS + shard) * HXR_SHARD_STRIDE, ticket);
		t = __builtin_amdgcn_readfirstlane(t);
		if (t >= appended) {
			shard = (shard + 1) & (HXR_QUEUE_SHARDS - 1);
			shards_done++;
			continue;
		}
		tl.template lap<PH_DISPENSE>();
#ifdef HXR_CONT_CLOCKS
		const unsigned long long ticket_t0 = __builtin_readcyclecounter();
#endif
		if ((uint32_t)lane < ticket && t + (uint32_t)lane < appended) {
			const HxrRayRec rec = P.q_records[(size_t)shard * P.q_capacity + t + (uint32_t)lane];
			const RngCtx rc{P.seed, rec.pixel, rec.sample_depth & 0xffffu};
			const RayItem ray{mk3(rec.o[0], rec.o[1], rec.o[2]), mk3(rec.d[0], rec.d[1], rec.d[2]), mkf(rec.weight[0], rec.weight[1], rec.weight[2]),
			                  (int)((rec.sample_depth >> 16) & 0xffu), rec.sample_depth >> 24, rec.key};
			tl.template lap<PH_GEN>();
			const f3 c = trace_rays<COUNT, F>(S, ray, rc, stacks, tl, none);
			tl.template lap<PH_SHADE>();
			unsigned long long* acc = (unsigned long long*)P.q_accum + (size_t)rec.pixel * 3;
			if (c.x != 0) atomicAdd(acc + 0, (unsigned long long)(long long)(c.x * HXR_ACCUM_SCALE));
			if (c.y != 0) atomicAdd(acc + 1, (unsigned long long)(long long)(c.y * HXR_ACCUM_SCALE));
			if (c.z != 0) atomicAdd(acc + 2, (unsigned long long)(long long)(c.z * HXR_ACCUM_SCALE));
		}
#ifdef HXR_CONT_CLOCKS   // developer aid: how long do the tickets (64 rays) of the continuation pass take?  words after the overflow flag:
		{                    // [1] longest, [2] count, [3] sum (all in units of 1024 clocks), [4 + k] tickets longer than 2^(17 + k) clocks
			const unsigned long long dt = __builtin_readcyclecounter() - ticket_t0;
			uint32_t* dbg = P.q_ctl + 2 * HXR_QUEUE_SHARDS * HXR_SHARD_STRIDE;
			if (lane == 0) {
				atomicMax(dbg + 1, (uint32_t)(dt >> 10));
				atomicAdd(dbg + 2, 1u);
				atomicAdd(dbg + 3, (uint32_t)(dt >> 10));
				for (int k = 0; k < 10; k++)
					if (dt > (1ull << (17 + k))) atomicAdd(dbg + 4 + k, 1u);
			}
		}
#endif
		tl.template lap<PH_REDUCE>();
	}
#ifdef HXR_PHASE_TIMERS
	if (P.counters && lane == 0) {
#ifndef HXR_PACKET_CLOCKS
		for (int i = 0; i < PH_NUM; i++) atomicAdd(&P.counters[HC_NUM + i], (unsigned long long)tl.ph[i]);
#endif
	}
#endif
	if (COUNT && P.counters) {
		for (int i = 0; i < HC_NUM; i++) {
			uint32_t v = tl.c[i];
			for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
			if (lane == 0 && v) atomicAdd(&P.counters[i], (unsigned long long)v);
		}
	}
}

#ifndef HXR_RENDER_KERNEL_ONLY   // the small kernels below live in hxr_api.hip only
// Adaptive schedule, step 1: total cost of the previous frame's chunks.
// `largest` (may be null): receives the largest chunk cost as well (hxr_sched_build_kernel's two phases)
__global__ void hxr_sched_sum_kernel(const uint32_t* __restrict__ cost, uint32_t n, unsigned long long* total, uint32_t* largest)
{
	unsigned long long s = 0;
	uint32_t m = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint32_t c = cost[i]; s += c; m = max(m, c); }
	for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); m = max(m, (uint32_t)__shfl_down((int)m, off, 64)); }
	if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
	if ((threadIdx.x & 63) == 0 && largest && m) atomicMax(largest, m);
}

// Step 2: chunks that cost more than `split_x` times the mean are listed pixel by pixel from the front of `items` (their
// rays have scattered: a packet of them costs in proportion to its live lanes, so four one-pixel packets on four waves finish
// in a quarter of the time); chunks above `first_x` times the mean are listed whole, from the back (expensive but coherent:
// only started early).  `items` has one entry per pixel, so the two lists cannot meet.  `cost` is cleared for the frame
// about to be rendered.
// `items_r` / `count_r` (may be null): a second list for the items that the replicating blocks take first (HXR_F_REPL): sample parts always, single pixels
// when `route_pixels` (a pixel's samples fit 16 lanes).  They then do not appear in `items`.
__global__ void hxr_sched_build_kernel(uint32_t* cost, uint32_t n, uint32_t ppc, uint32_t num_pixels, const unsigned long long* total, uint32_t split_x,
                                       uint32_t first_x, uint32_t resident_waves, uint32_t frame_div, uint32_t part_groups, uint8_t* flags, uint32_t* items, uint32_t capacity,
                                       uint32_t* part_pixels, uint32_t* counts, uint32_t* items_r, uint32_t* count_r, uint32_t route_pixels, uint32_t phase, const uint32_t* largest)
{
	// `phase` (round 4): the lists are filled in the order the atomics land, and in a launch a few packets deep per wave the front list is longer than one round
	// of the resident waves -- the one-pixel packets of the costliest chunk of a 1-spp frame (0.7 ms chains in a 0.9 ms frame) began 0.2 ms late.  Such launches run this
	// kernel twice: phase 1 lists the chunks that cost at least half of the costliest (their pixels / parts / themselves: the first entries of every list), phase 2 the
	// rest.  phase 0: everything in one go (whole frames: the tail is elsewhere).  Only phases 0 and 2 clear `cost`.
	const unsigned long long top_above = phase ? max(1ull, (unsigned long long)*largest / 2ull) : 0ull;
	uint32_t* part_items = items_r ? items_r : items;
	uint32_t* part_count = items_r ? count_r : counts + 0;
	uint32_t* pixel_items = items_r && route_pixels ? items_r : items;
	uint32_t* pixel_count = items_r && route_pixels ? count_r : counts + 0;
	const unsigned long long mean = *total / (n ? n : 1u);
	// a packet is worth splitting when it is long against the frame itself: as long as what one wave gets through when the work
	// is spread perfectly (a shard of an 8-GPU frame is eight times shorter than the whole frame, its limit eight times lower)
	const unsigned long long split_above = max((unsigned long long)split_x * mean, *total / (resident_waves ? resident_waves : 1u) / (frame_div ? frame_div : 1u));
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const unsigned long long c = cost[i];
		if (phase == 1 && c < top_above) continue;                      // its turn comes in phase 2
		if (phase != 1) cost[i] = 0;
		if (phase == 2 && c >= top_above) continue;                     // listed in phase 1
		const uint32_t first = i * ppc;
		const bool heavy = mean > 0 && c > (unsigned long long)first_x * mean && first < num_pixels;
		if (heavy) {
			const uint32_t cnt = min(ppc, num_pixels - first);
			if (part_groups > 0 && c > split_above * ppc) {           // even one pixel of it is too long: HXR_PART_LANES samples per packet
				for (uint32_t j = 0; j < cnt; j++) {
					const uint32_t ps = atomicAdd(counts + 2, 1u);
					if (ps < HXR_PART_SLOTS) {
						part_pixels[ps] = first + j;
						const uint32_t at = atomicAdd(part_count, part_groups);
						for (uint32_t g = 0; g < part_groups; g++) part_items[at + g] = HXR_SCHED_PART | (g << 22) | ps;
					} else pixel_items[atomicAdd(pixel_count, 1u)] = first + j;
				}
			} else if (c > split_above && ppc > 1) {
				const uint32_t at = atomicAdd(pixel_count, cnt);
				for (uint32_t j = 0; j < cnt; j++) pixel_items[at + j] = first + j;
			} else if (c > split_above / HXR_SCHED_URGENT_DIV) {
				// long against the frame, though not long enough to split: at the FRONT, behind the split ones -- the back list is thousands
				// of chunks in the order their atomics landed, two rounds of the resident waves deep, and a packet of a millisecond that
				// happens to sit in its second half ends the frame late (one frame in six, at four waves per SIMD)
				items[atomicAdd(counts + 0, 1u)] = first | HXR_SCHED_WHOLE;
			} else {
				const uint32_t at = atomicAdd(counts + 1, 1u);
				items[capacity - 1u - at] = first | HXR_SCHED_WHOLE;
			}
		}
		flags[i] = heavy ? 1 : 0;
	}
}

// Deferred secondary rays, last step: what the continuation pass added up for a pixel (HxrLaunch::q_accum, 2^-32 units) goes into
// the frame -- `scale` = 1 / spp for Monte-Carlo frames, 1 for pass 1, 1 / 5 for the AA taps, like the sums the render kernel wrote --
// and the accumulator is left cleared for the next pass.  Own tile pixels of the shard, then the halo pixels of a sharded AA pass 1.
__global__ __launch_bounds__(HXR_BLOCK) void hxr_resolve_kernel(HxrLaunch P, float scale)
{
	const uint32_t tile_pixels = (uint32_t)P.num_shard_tiles * (uint32_t)(P.tile_size * P.tile_size);
	const uint32_t total = tile_pixels + (uint32_t)P.num_extra_pixels;
	for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
		int x, y;
		float* dst;
		if (q < tile_pixels) {
			if (!shard_pixel(P, q, x, y)) continue;
			dst = out_pixel(P, q, x, y);
		} else {
			const int pix = P.extra_pixels[q - tile_pixels];
			x = pix % P.W;
			y = pix / P.W;
			dst = (float*)((char*)P.out + (size_t)y * P.out_stride) + (size_t)x * 3;
		}
		long long* acc = P.q_accum + ((size_t)y * P.W + x) * 3;
		const long long r = acc[0], g = acc[1], b = acc[2];
		if ((r | g | b) == 0) continue;
		acc[0] = 0; acc[1] = 0; acc[2] = 0;
		dst[0] += (float)((double)r * (1.0 / 4294967296.0)) * scale;
		dst[1] += (float)((double)g * (1.0 / 4294967296.0)) * scale;
		dst[2] += (float)((double)b * (1.0 / 4294967296.0)) * scale;
	}
}

// Adaptive schedule without history (the first frame of a scene, a new frame size ...): chunks inside the screen rectangles of nodes that
// spawn secondary rays -- where long reflection chains can only be -- are listed WHOLE at the back of `items`, i.e. started first, the way
// the cost map would list expensive-but-coherent chunks.  Monte-Carlo chunks are not split (that needs measured costs); the 64-pixel chunks
// of 1-spp passes are handed out in four runs of sixteen pixels, because a wave that holds 64 unrelated reflection chains IS the frame
// time (4 ms of a 1.3 ms pass).  The image does not depend on any of it.
struct HxrRects { int32_t n; int32_t r[8][4]; };   // x0, y0, x1, y1 (exclusive), frame pixels
__global__ void hxr_sched_rects_kernel(HxrLaunch P, HxrRects R, uint32_t n, uint32_t ppc, uint32_t num_pixels, uint32_t* cost, uint8_t* flags, uint32_t* items, uint32_t capacity,
                                       uint32_t* counts)
{
	const bool quarters = P.spp == 1 && ppc == 64 && P.mode == 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		cost[i] = 0;
		const uint32_t first = i * ppc;
		bool inside = false;
		int x, y;
		// (the general mapping: lane i asks for pixel i * ppc, so a wave spans 64 * ppc pixel numbers -- several tiles whenever that exceeds a tile, which
		// shard_pixel()'s one-tile-per-wave shortcut does not serve: with it a first frame of 8-pixel shard tiles listed one chunk in 64)
		if (first < num_pixels && shard_pixel_general(P, first, x, y))
			for (int k = 0; k < R.n; k++) inside = inside || (x >= R.r[k][0] && x < R.r[k][2] && y >= R.r[k][1] && y < R.r[k][3]);
		if (inside) {
			if (quarters) {   // 1-spp packets (64 pixels): sixteen pixels per packet, so that 64 unrelated chains do not share one wave
				const uint32_t at = atomicAdd(counts + 1, ppc / HXR_QUARTER_LANES);
				for (uint32_t k = 0; k < ppc / HXR_QUARTER_LANES; k++) items[capacity - 1u - (at + k)] = (first + k * HXR_QUARTER_LANES) | HXR_SCHED_QUARTER;
			} else items[capacity - 1u - atomicAdd(counts + 1, 1u)] = first | HXR_SCHED_WHOLE;
		}
		flags[i] = inside ? 1 : 0;
	}
}

// Autofocus probe of renderWithMonteCarlo (src/main.cpp:350-362): the distance to the closest node along the
// ray through the frame centre becomes the focal plane distance.  One wavefront, every lane the same ray.
__global__ __launch_bounds__(64) void hxr_autofocus_kernel(
    const HxrSceneDev* __restrict__ Sp, const HxrNodeDev* __restrict__ a_nodes, const HxrMeshDev* __restrict__ a_meshes,
    const HxrHeightfieldDev* __restrict__ a_hfs, const hxr_geometry* __restrict__ a_geoms, const HxrKdSlim* __restrict__ a_kd, const HxrKdBox* __restrict__ a_kd_box,
    const HxrTriHot* __restrict__ a_hot, const int32_t* __restrict__ a_hot_ids, int W, int H, double* focal_out)
{
	__shared__ WaveStack wave_stack;
	const Stacks st{&wave_stack, LaneStack{nullptr, 0, nullptr}, nullptr};
	HxrSceneDev S = *Sp;
	S.nodes = a_nodes;
	S.meshes = a_meshes;
	S.heightfields = a_hfs;
	S.geometries = a_geoms;
	S.kd = a_kd;
	S.kd_box = a_kd_box;
	S.hot = a_hot;
	S.hot_ids = a_hot_ids;
	Tally<false> tl;
#ifdef HXR_PHASE_TIMERS
	__shared__ unsigned long long lds_clock[PH_NUM + 1];
	tl.attach(lds_clock);
#endif
	d3 o = ld3(S.camera.pos), tlc = ld3(S.camera.top_left);
	d3 through = tlc + (ld3(S.camera.top_right) - tlc) * ((W * 0.5) / (double)W) + (ld3(S.camera.bottom_left) - tlc) * ((H * 0.5) / (double)H);
	d3 d = unit(through - o);
	GeomHit g;
	double w;
	int node = scene_query<false, (HXR_F_ALL & ~HXR_F_GI)>(S, o, d, HXR_INF, false, g, w, st, tl);
	if (threadIdx.x == 0 && node >= 0 && w < HXR_INF) *focal_out = w;
}

// detectAApixels, src/main.cpp:192-222: a pixel needs AA when any channel differs from any of its 8
// neighbours by more than 0.075 after clamping to 1.  Flagged pixels are appended to a compact list with
// one atomic per wave (ballot + prefix count).
// `cost` (may be null): what the 64-pixel chunks of pass 1 took; flagged pixels of chunks above `heavy_x` times the mean go to the BACK of
// the list (`heavy_count` of them), to be traced first and one pixel per packet by pass 3 -- the flagged set is small and made
// of the most expensive pixels of the frame, so a few 32-lane packets of them would otherwise decide when pass 3 ends.
#define HXR_DETECT_LDS 2048   // flagged pixels a block collects in LDS before it falls back to one atomic per pixel
__global__ __launch_bounds__(HXR_BLOCK) void hxr_detect_aa_kernel(HxrLaunch P, const float* __restrict__ frame, size_t stride, int32_t* list,
                                                                   int32_t* count, const uint32_t* __restrict__ cost, const unsigned long long* total,
                                                                   uint32_t cost_chunks, uint32_t* heavy_count, uint32_t capacity, uint32_t heavy_x)
{
	const unsigned long long heavy_above = cost ? (unsigned long long)heavy_x * (*total / (cost_chunks ? cost_chunks : 1u)) : ~0ull;
	uint32_t tile_pixels = (uint32_t)P.num_shard_tiles * (uint32_t)(P.tile_size * P.tile_size);
	__shared__ int32_t s_list[2][HXR_DETECT_LDS];   // this block's flagged pixels: [0] ordinary, [1] of heavy chunks
	__shared__ uint32_t s_n[2], s_base[2];
	if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
	__syncthreads();
	// An unsharded frame is walked ROW-MAJOR (round 4): a wave then reads three contiguous 768-byte runs of the frame instead of an 8x8 block's 24 short rows
	// (155 -> ~40 us at 1920x1080: the pass between the two render passes of an adaptive-AA frame); the pixel's number in shard order, which indexes the
	// pass-1 cost map, is formed from (x, y).  Shards keep the tile order: their pixels are not a rectangle.  The ORDER of the flagged list differs between
	// the two walks (it is unordered anyway: one atomic per wave); which pixels are flagged does not.
	const bool row_major = P.shard_count == 1;
	const uint32_t span = row_major ? (uint32_t)(P.W * P.H) : tile_pixels;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ((span + 63u) & ~63u); i += gridDim.x * blockDim.x) {
		int x = 0, y = 0;
		uint32_t q = i;
		bool mine = false;
		if (row_major) {
			mine = i < span;
			x = (int)(i % (uint32_t)P.W);
			y = (int)(i / (uint32_t)P.W);
			const int ts = P.tile_size, lx = x % ts, ly = y % ts;
			const uint32_t m = (uint32_t)((lx & 1) | ((ly & 1) << 1) | ((lx & 2) << 1) | ((ly & 2) << 2) | ((lx & 4) << 2) | ((ly & 4) << 3));
			q = (uint32_t)((y / ts) * P.tiles_x + x / ts) * (uint32_t)(ts * ts) + (uint32_t)((ly >> 3) * (ts >> 3) + (lx >> 3)) * 64u + m;
		} else {
			mine = i < tile_pixels && shard_pixel(P, i, x, y);
		}
		bool need = false;
		// src/main.cpp:201,205: with a fovea only pixels inside it are candidates
		if (mine && !((x - P.vip_x) * (x - P.vip_x) + (y - P.vip_y) * (y - P.vip_y) > (P.fov_radius > 0 ? P.fov_radius * P.fov_radius : 999666111))) {
			const float* me = (const float*)((const char*)frame + (size_t)y * stride) + (size_t)x * 3;
			const float m0 = fminf(1.0f, me[0]), m1 = fminf(1.0f, me[1]), m2 = fminf(1.0f, me[2]);
			float worst = 0.0f;   // the largest clamped difference to any neighbour in any channel: `> 0.075` for one of them <=> for the largest (no early exits: the lanes of a wave leave at different ones)
#pragma unroll
			for (int dy = -1; dy <= 1; dy++)
#pragma unroll
				for (int dx = -1; dx <= 1; dx++) {
					if (!dx && !dy) continue;
					const int nx = x + dx, ny = y + dy;
					if (nx < 0 || nx >= P.W || ny < 0 || ny >= P.H) continue;
					const float* nb = (const float*)((const char*)frame + (size_t)ny * stride) + (size_t)nx * 3;
					worst = fmaxf(worst, fmaxf(fabsf(m0 - fminf(1.0f, nb[0])), fmaxf(fabsf(m1 - fminf(1.0f, nb[1])), fabsf(m2 - fminf(1.0f, nb[2])))));
				}
			need = worst > 0.075f;
		}
		const bool heavy = need && cost && (unsigned long long)cost[q >> 6] > heavy_above;   // pass-1 chunks are 64 consecutive pixel numbers
		// collected per BLOCK in LDS (round 4): one global atomic per block and list instead of one per wave -- a single counter takes ~88 atomics per
		// microsecond, and half the waves of a 1080p frame hold a flagged pixel
		if (need && !heavy) { const uint32_t at = atomicAdd(&s_n[0], 1u); if (at < HXR_DETECT_LDS) s_list[0][at] = y * P.W + x; else list[atomicAdd(count, 1)] = y * P.W + x; }
		if (heavy) { const uint32_t at = atomicAdd(&s_n[1], 1u); if (at < HXR_DETECT_LDS) s_list[1][at] = y * P.W + x; else list[capacity - 1u - atomicAdd(heavy_count, 1u)] = y * P.W + x; }
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const uint32_t n0 = min(s_n[0], (uint32_t)HXR_DETECT_LDS), n1 = min(s_n[1], (uint32_t)HXR_DETECT_LDS);
		s_base[0] = n0 ? (uint32_t)atomicAdd(count, (int)n0) : 0u;
		s_base[1] = n1 ? atomicAdd(heavy_count, n1) : 0u;
	}
	__syncthreads();
	const uint32_t n0 = min(s_n[0], (uint32_t)HXR_DETECT_LDS), n1 = min(s_n[1], (uint32_t)HXR_DETECT_LDS);
	for (uint32_t k = threadIdx.x; k < n0; k += blockDim.x) list[s_base[0] + k] = s_list[0][k];
	for (uint32_t k = threadIdx.x; k < n1; k += blockDim.x) list[capacity - 1u - (s_base[1] + k)] = s_list[1][k];
}

// own tiles of a row-major scratch frame -> the caller's buffer: tile-major (`as_tiles`) or row-major with `dst_stride` bytes per row
// (sharded adaptive-AA frames render into a scratch frame; only the shard's own pixels are handed over)
__global__ void hxr_extract_tiles_kernel(HxrLaunch P, const float* __restrict__ frame, size_t stride, float* dst_base, size_t dst_stride, int as_tiles)
{
	uint32_t tile_pixels = (uint32_t)P.num_shard_tiles * (uint32_t)(P.tile_size * P.tile_size);
	for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < tile_pixels; q += gridDim.x * blockDim.x) {
		int x, y;
		if (!shard_pixel(P, q, x, y)) continue;
		const float* src = (const float*)((const char*)frame + (size_t)y * stride) + (size_t)x * 3;
		HxrLaunch T = P;
		T.out = dst_base;
		T.out_stride = dst_stride;
		T.out_tiles = as_tiles;
		float* dst = out_pixel(T, q, x, y);
		dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
	}
}

// tile-major shard buffers (rank-major, as gathered) -> row-major frame
__global__ void hxr_assemble_kernel(const float* __restrict__ gathered, size_t per_rank_floats, float* frame, size_t stride, int W, int H,
                                    int tile_size, int tiles_x, int tiles_y, int shard_count)
{
	size_t total = (size_t)tiles_x * tiles_y * tile_size * tile_size;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		int per_tile = tile_size * tile_size;
		int t = (int)(i / per_tile), in = (int)(i % per_tile);
		int tx, ty;
		hxr_tile_position(t, tiles_x, shard_count, &tx, &ty);
		int x = tx * tile_size + in % tile_size, y = ty * tile_size + in / tile_size;
		if (x >= W || y >= H) continue;
		int rank = t % shard_count, lt = t / shard_count;
		const float* src = gathered + (size_t)rank * per_rank_floats + ((size_t)lt * per_tile + in) * 3;
		float* dst = (float*)((char*)frame + (size_t)y * stride) + (size_t)x * 3;
		dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
	}
}

// Color::toRGB32 through the 4097-entry table (src/color.h:84-90, src/sdl.cpp:404-419)
__global__ void hxr_quantize_kernel(const float* __restrict__ frame, size_t stride, int W, int H, const uint8_t* __restrict__ lut, uint8_t* bgr)
{
	size_t total = (size_t)W * H;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		int x = (int)(i % W), y = (int)(i / W);
		const float* p = (const float*)((const char*)frame + (size_t)y * stride) + (size_t)x * 3;
#pragma unroll
		for (int c = 0; c < 3; c++) {
			float v = p[2 - c];
			uint8_t o = v <= 0 ? 0 : (v >= 1 ? 255 : lut[(int)(v * 4096.0f)]);
			bgr[i * 3 + c] = o;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// scene preparation on the device (the parts that are pure copies / maxima and therefore bit-identical to
// the host's): leaf-ordered hot triangle records, and the heightfield's maxima tables
// ------------------------------------------------------------------------------------------------
// One 96-byte hot record per entry of `ids` (triangle numbers local to the mesh, in leaf order): A, AB, AC, ABxAC -- the operands of
// Mesh::intersectTriangle's rejection tests (src/mesh.cpp:178-195) -- gathered from the cold records already in HBM.
__global__ void hxr_expand_hot_kernel(const hxr_triangle* __restrict__ tris, const double* __restrict__ verts, const int32_t* __restrict__ ids, int64_t n, HxrTriHot* out)
{
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const hxr_triangle& t = tris[ids[i]];
		HxrTriHot h;
#pragma unroll
		for (int c = 0; c < 3; c++) {
			h.A[c] = verts[(size_t)t.v[0] * 3 + c];
			h.AB[c] = t.AB[c];
			h.AC[c] = t.AC[c];
			h.N[c] = t.ABcrossAC[c];
		}
		out[i] = h;
	}
}

// hxr_heightfield::raw_heights -- the rest of Heightfield::fillProperties (src/heightfield.cpp:173-261) on the GPU, one texel per lane.
// The Gaussian blur (src/heightfield.cpp:189-212): taps outside the image count as 0; the taps of a texel are added up in the
// reference's loop order (dy outer, dx inner) with a fused multiply-add each.  The host loader's sums of the same taps are grouped by
// its compiler's vectoriser (eight partial sums per row), so the two agree to float rounding, not bit for bit.
__global__ void hxr_hf_blur_kernel(const float* __restrict__ grey, int W, int H, const float* __restrict__ g, int R, float* __restrict__ heights)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * H; i += gridDim.x * blockDim.x) {
		const int x = i % W, y = i / W;
		float sum = 0;
		for (int dy = -R + 1; dy < R; dy++) {
			const int yy = y + dy;
			if (yy < 0 || yy >= H) continue;
			const float* row = grey + (size_t)yy * W;
			const float* gr = g + (size_t)(dy < 0 ? -dy : dy) * R;
			for (int dx = -R + 1; dx < R; dx++) {
				const int xx = x + dx;
				if (xx >= 0 && xx < W) sum = fmaf(gr[dx < 0 ? -dx : dx], row[xx], sum);
			}
		}
		heights[i] = sum;
	}
}

// lowest and highest height (the y extent of the bounding box, src/heightfield.cpp:213-216): ONE workgroup, out[0] = min, out[1] = max
__global__ __launch_bounds__(1024) void hxr_hf_range_kernel(const float* __restrict__ heights, int n, float* out)
{
	__shared__ float s_lo[1024], s_hi[1024];
	float lo = 1e17f, hi = -1e17f;
	for (int i = threadIdx.x; i < n; i += 1024) { lo = fminf(lo, heights[i]); hi = fmaxf(hi, heights[i]); }
	s_lo[threadIdx.x] = lo;
	s_hi[threadIdx.x] = hi;
	__syncthreads();
	for (int step = 512; step > 0; step >>= 1) {
		if ((int)threadIdx.x < step) { s_lo[threadIdx.x] = fminf(s_lo[threadIdx.x], s_lo[threadIdx.x + step]); s_hi[threadIdx.x] = fmaxf(s_hi[threadIdx.x], s_hi[threadIdx.x + step]); }
		__syncthreads();
	}
	if (threadIdx.x == 0) { out[0] = s_lo[0]; out[1] = s_hi[0]; }
}

// the per-texel normals (src/heightfield.cpp:237-253): unit(cross((0, hy - h0, 1), (1, hx - h0, 0))) = unit(-(hx - h0), 1, -(hy - h0)); the last
// column and the last row repeat their neighbours
__global__ void hxr_hf_normals_kernel(const float* __restrict__ heights, int W, int H, double* __restrict__ normals)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * H; i += gridDim.x * blockDim.x) {
		const int x = min(i % W, W - 2), y = min(i / W, H - 2);
		double n[3] = {0, 0, 0};
		if (x >= 0 && y >= 0) {
			const float h0 = heights[(size_t)y * W + x];
			const double nx = (double)(h0 - heights[(size_t)y * W + x + 1]), nz = (double)(h0 - heights[(size_t)(y + 1) * W + x]);
			const double k = 1.0 / sqrt(fma(nx, nx, fma(nz, nz, 1.0)));
			n[0] = nx * k;
			n[1] = k;
			n[2] = nz * k;
		}
		normals[(size_t)i * 3 + 0] = n[0];
		normals[(size_t)i * 3 + 1] = n[1];
		normals[(size_t)i * 3 + 2] = n[2];
	}
}

// hxr_bitmap::prep -- the loader's two passes over a bitmap (src/bitmap.cpp:304-338), one channel value / one texel per lane.
// Bitmap::decompressGamma: a channel c > 0 becomes decompress_sRGB(c) (src/color.h:49-57) when the gamma is 2.2, pow(c, gamma) otherwise.
// `table` (an 8-bit source: the host's decode of q / 255.0f, q = 0..255) makes the result the host's bit for bit; without one
// (float texels) the curve is evaluated here, by the device's powf.
__global__ void hxr_bitmap_gamma_kernel(float* __restrict__ texels, size_t n, const float* __restrict__ table, int srgb, float gamma)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float c = texels[i];
		if (!(c > 0)) continue;
		float v;
		if (table) v = table[min(max((int)(c * 255.0f + 0.5f), 0), 255)];
		else if (!srgb) v = powf(c, gamma);
		else v = c >= 1 ? c : c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
		texels[i] = v;
	}
}

// Bitmap::differentiate: texel (x, y) becomes (I(x, y) - I(x + 1, y), I(x, y) - I(x, y + 1), 0) with wrap-around neighbours and
// I = Color::intensity = (r + g + b) / 3 -- which the reference's Release flags (-ffast-math: reciprocal-math) and the host loader built
// with them compile to a multiplication by the float nearest 1/3; so does this kernel, to be the host's table bit for bit.
__global__ void hxr_bitmap_differentiate_kernel(const float* __restrict__ src, int W, int H, float* __restrict__ dst)
{
#pragma clang fp contract(off)
	const float third = 1.0f / 3.0f;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * H; i += gridDim.x * blockDim.x) {
		const int x = i % W, y = i / W;
		const float* a = src + (size_t)i * 3;
		const float* bx = src + ((size_t)y * W + (x + 1) % W) * 3;
		const float* by = src + ((size_t)((y + 1) % H) * W + x) * 3;
		const float here = ((a[0] + a[1]) + a[2]) * third;
		dst[(size_t)i * 3 + 0] = here - ((bx[0] + bx[1]) + bx[2]) * third;
		dst[(size_t)i * 3 + 1] = here - ((by[0] + by[1]) + by[2]) * third;
		dst[(size_t)i * 3 + 2] = 0.0f;
	}
}

// The marcher's per-cell record: {maxH, highMap levels 1, 2, 3} (HxrHeightfieldDev::step_rec)
// (`high_map` is null for a heightfield without the pyramid -- useOptimization off: the descriptor then carries no such table)
__global__ void hxr_hf_pack_kernel(const float* __restrict__ max_h, const float* __restrict__ high_map, int cells, float* __restrict__ rec)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
		rec[(size_t)i * 4 + 0] = max_h[i];
		rec[(size_t)i * 4 + 1] = high_map ? high_map[(size_t)i * 16 + 1] : 0.0f;
		rec[(size_t)i * 4 + 2] = high_map ? high_map[(size_t)i * 16 + 2] : 0.0f;
		rec[(size_t)i * 4 + 3] = high_map ? high_map[(size_t)i * 16 + 3] : 0.0f;
	}
}

// Heightfield::maxH (src/heightfield.cpp:222-235): the highest of the four corners of a cell (fewer at the last row / column)
__global__ void hxr_hf_maxh_kernel(const float* __restrict__ heights, int W, int H, float* max_h)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * H; i += gridDim.x * blockDim.x) {
		const int x = i % W, y = i / W;
		float m = heights[i];
		if (x < W - 1) m = fmaxf(m, heights[i + 1]);
		if (y < H - 1) {
			m = fmaxf(m, heights[i + W]);
			if (x < W - 1) m = fmaxf(m, heights[i + W + 1]);
		}
		max_h[i] = m;
	}
}

// Heightfield::buildHighMap (src/heightfield.cpp:49-81): level 0 = the highest texel of the clamped 3x3 neighbourhood, level k = the highest
// of level k - 1 at the four corners (+-2^(k-1), +-2^(k-1)), clamped to the map.  One launch per level (each reads the previous one).
__global__ void hxr_hf_highmap_kernel(const float* __restrict__ heights, int W, int H, int k, float* high_map)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * H; i += gridDim.x * blockDim.x) {
		const int x = i % W, y = i / W;
		float r;
		if (k == 0) {
			r = heights[i];
			for (int dy = -1; dy <= 1; dy++)
				for (int dx = -1; dx <= 1; dx++) r = fmaxf(r, heights[(size_t)clampi(y + dy, H - 1) * W + clampi(x + dx, W - 1)]);
		} else {
			const int off = 1 << (k - 1);
			const int x0 = clampi(x - off, W - 1), x1 = clampi(x + off, W - 1), y0 = clampi(y - off, H - 1), y1 = clampi(y + off, H - 1);
			r = high_map[((size_t)y0 * W + x0) * 16 + (k - 1)];
			r = fmaxf(r, high_map[((size_t)y0 * W + x1) * 16 + (k - 1)]);
			r = fmaxf(r, high_map[((size_t)y1 * W + x0) * 16 + (k - 1)]);
			r = fmaxf(r, high_map[((size_t)y1 * W + x1) * 16 + (k - 1)]);
		}
		high_map[(size_t)i * 16 + k] = r;
	}
}

// displayVFB (src/sdl.cpp:79-90): Color::toRGB32 per pixel -- the three 8-bit sRGB channels shifted to where the display surface
// wants them (src/color.h:84-90) -- into a 32-bit surface with its own pitch
__global__ void hxr_quantize_rgb32_kernel(const float* __restrict__ frame, size_t stride, int W, int H, const uint8_t* __restrict__ lut, int rshift, int gshift, int bshift,
                                          uint32_t* pixels, size_t pitch)
{
	size_t total = (size_t)W * H;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		int x = (int)(i % W), y = (int)(i / W);
		const float* p = (const float*)((const char*)frame + (size_t)y * stride) + (size_t)x * 3;
		uint32_t c[3];
#pragma unroll
		for (int k = 0; k < 3; k++) {
			float v = p[k];
			c[k] = v <= 0 ? 0u : (v >= 1 ? 255u : (uint32_t)lut[(int)(v * 4096.0f)]);
		}
		((uint32_t*)((char*)pixels + (size_t)y * pitch))[x] = (c[2] << bshift) | (c[1] << gshift) | (c[0] << rshift);
	}
}
#endif   // HXR_RENDER_KERNEL_ONLY
