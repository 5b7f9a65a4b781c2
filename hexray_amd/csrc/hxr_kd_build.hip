// Mesh::buildKD on the GPU; see hxr_kd_build.h.
//
// One kernel launch per tree level.  A level is a list of tasks {node, box, triangle list, depth}; one workgroup takes
// one task: a task below 20 triangles or past depth 64 writes its leaf, any other splits its box at the midpoint, runs
// BBox::intersectTriangle of every triangle of its list against both halves (one triangle per lane) and writes the two
// child lists with a workgroup-wide stable partition (ballot + popcount ranks), so every list keeps the order the
// reference's push_back loop gives it (src/mesh.cpp:108-116).  Child tasks, child node pairs, list space and leaf space
// are claimed with one atomicAdd each; levels are launched back to back (a level reads its own task count), and the host looks
// at the control words every eight levels to see the end of the build and any capacity overflow, which restarts the build with larger buffers.  The atomics make
// the device-side numbering of nodes and leaf lists vary from run to run; the copy back renumbers the tree in the host
// builder's order (pre-order child pairs, depth-first leaf lists), which makes the result deterministic and directly
// comparable with a host-built tree.
//
// The first levels of a mesh are few tasks with long lists.  There (a task of KD_BIG triangles or more, in the first
// KD_CHUNKED_LEVELS levels) the costly half of a task -- the two BBox::intersectTriangle tests per triangle -- is spread over the
// whole GPU: a task's creator also lists it as chunks of KD_CHUNK triangles, kd_classify_kernel takes one chunk per workgroup and
// leaves two bits per triangle, and the task's own workgroup then only partitions by those bits.  Same tests, same answers, same tree;
// the root of a 120 000-triangle mesh is classified by 469 workgroups at once instead of in 118 rounds of one.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "hxr_kd_build.h"

namespace hxr_kd {
namespace {

#define KD_LEAF_BELOW 20   // TRIANGLES_PER_LEAF, src/mesh.cpp:97
#define KD_MAX_DEPTH 64    // MAX_TREE_DEPTH, src/mesh.cpp:97 (a node deeper than this is a leaf)

struct KdTask {   // 64 B
	double lo[3], hi[3];
	int32_t node, first, count, depth;
};

struct KdCtl {
	// per level, three sets in rotation (a level reads one, fills the next and clears the third for the level after it -- no memset between
	// the launches): {number of tasks, ints of list space claimed, chunks listed, unused}
	uint32_t side[3][4];
	uint32_t node_cursor, leaf_cursor, error, max_depth;
};

struct KdCaps { uint32_t tasks, list, nodes, leaf, chunks; };

#define KD_BIG 1024           // a task at least this long ...
#define KD_CHUNKED_LEVELS 12  // ... and above this depth is classified chunk by chunk (kd_classify_kernel)
#define KD_CHUNK 256
struct KdChunk { uint32_t task, index; };   // triangles index * KD_CHUNK ... of the task's list

__host__ __device__ inline bool kd_chunked(int32_t count, int32_t depth) { return count >= KD_BIG && depth < KD_CHUNKED_LEVELS; }

struct P3 { double v[3]; };

__device__ inline P3 sub(const P3& a, const P3& b) { return P3{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; }
__device__ inline P3 neg(const P3& a) { return P3{{-a.v[0], -a.v[1], -a.v[2]}}; }
__device__ inline double dot(const P3& a, const P3& b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
__device__ inline P3 cross(const P3& a, const P3& b)
{
	return P3{{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}

struct BoxD {
	double lo[3], hi[3];
	// BBox::inside, borders widened by 1e-6 (src/bbox.h:78-85)
	__device__ bool holds(const P3& p) const
	{
		return lo[0] - 1e-6 <= p.v[0] && p.v[0] <= hi[0] + 1e-6 && lo[1] - 1e-6 <= p.v[1] && p.v[1] <= hi[1] + 1e-6 && lo[2] - 1e-6 <= p.v[2] &&
		       p.v[2] <= hi[2] + 1e-6;
	}
	// BBox::testIntersect (src/bbox.h:94-141); `d` need not be unit length
	__device__ bool hit_by(const P3& o, const P3& d) const
	{
		if (holds(o)) return true;
		for (int a = 0; a < 3; a++) {
			if ((d.v[a] < 0 && o.v[a] < lo[a]) || (d.v[a] > 0 && o.v[a] > hi[a])) return false;
			if (fabs(d.v[a]) < 1e-9) continue;
			const double inv = 1 / d.v[a];
			const int u = (a == 0) ? 1 : 0, v = (a == 2) ? 1 : 2;
			for (int w = 0; w < 2; w++) {
				const double t = ((w ? hi[a] : lo[a]) - o.v[a]) * inv;
				if (t < 0) break;   // near wall behind the origin: the far wall of this axis is skipped too
				const double pu = o.v[u] + d.v[u] * t;
				if (lo[u] <= pu && pu <= hi[u]) {
					const double pv = o.v[v] + d.v[v] * t;
					if (lo[v] <= pv && pv <= hi[v]) return true;
				}
			}
		}
		return false;
	}
};

// intersectTriangleFast (src/mesh.cpp:131-170)
__device__ bool segment_hits_triangle(const P3& o, const P3& d, const P3& A, const P3& B, const P3& C, double& best)
{
	const P3 AB = sub(B, A), AC = sub(C, A), nd = neg(d), H = sub(o, A);
	const P3 N = cross(AB, AC);
	const double den = dot(N, nd);
	if (fabs(den) < 1e-12) return false;
	const double r = 1 / den;   // the host builder is compiled with the reference's Release flags (-ffast-math): one reciprocal, three products
	const double l2 = dot(cross(H, AC), nd) * r;
	const double l3 = dot(cross(AB, H), nd) * r;
	const double g = dot(N, H) * r;
	if (g < 0 || g > best) return false;
	if (l2 < 0 || l2 > 1 || l3 < 0 || l3 > 1 || l2 + l3 > 1) return false;
	best = g;
	return true;
}

__device__ inline double sign_of(double x) { return x > 0 ? +1 : -1; }

// BBox::intersectTriangle (src/bbox.h:177-207): a vertex inside, an edge through the box, a box edge through the triangle
__device__ bool box_overlaps_triangle(const BoxD& b, const P3& A, const P3& B, const P3& C)
{
	if (b.holds(A) || b.holds(B) || b.holds(C)) return true;
	const P3 t[3] = {A, B, C};
	for (int i = 0; i < 3; i++)
		for (int j = i + 1; j < 3; j++)
			if (b.hit_by(t[i], sub(t[j], t[i])) && b.hit_by(t[j], sub(t[i], t[j]))) return true;
	const P3 N = cross(sub(B, A), sub(C, A));
	const double D = dot(A, N);
	for (int mask = 0; mask < 7; mask++)
		for (int j = 0; j < 3; j++) {
			if (mask & (1 << j)) continue;
			const P3 s{{(mask & 1) ? b.hi[0] : b.lo[0], (mask & 2) ? b.hi[1] : b.lo[1], (mask & 4) ? b.hi[2] : b.lo[2]}};
			P3 e = s;
			e.v[j] = b.hi[j];
			if (sign_of(dot(s, N) - D) != sign_of(dot(e, N) - D)) {
				double reach = 1.0000001;
				if (segment_hits_triangle(s, sub(e, s), A, B, C, reach)) return true;
			}
		}
	return false;
}

__global__ void kd_iota_kernel(int32_t* list, uint32_t n)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) list[i] = (int32_t)i;
}

// the two halves of a task's box (BBox::split, src/bbox.h:210-216) at the midpoint of axis depth % 3 (src/mesh.cpp:100-104)
__device__ inline void kd_halves(const KdTask& T, int& axis, double& split, BoxD& L, BoxD& R)
{
	axis = T.depth % 3;
	split = T.lo[axis] + (T.hi[axis] - T.lo[axis]) * 0.5;
	for (int c = 0; c < 3; c++) { L.lo[c] = R.lo[c] = T.lo[c]; L.hi[c] = R.hi[c] = T.hi[c]; }
	L.hi[axis] = split;
	R.lo[axis] = split;
}

__device__ inline uint32_t kd_sides_of(const hxr_triangle* __restrict__ tris, const double* __restrict__ verts, int tri, const BoxD& L, const BoxD& R)
{
	const hxr_triangle& t = tris[tri];
	P3 A, B, C;
	for (int k = 0; k < 3; k++) {
		A.v[k] = verts[3 * (size_t)t.v[0] + k];
		B.v[k] = verts[3 * (size_t)t.v[1] + k];
		C.v[k] = verts[3 * (size_t)t.v[2] + k];
	}
	return (box_overlaps_triangle(L, A, B, C) ? 1u : 0u) | (box_overlaps_triangle(R, A, B, C) ? 2u : 0u);
}

// The long tasks of a level, one chunk of KD_CHUNK triangles per workgroup: which halves of the task's box each triangle touches
// (bit 0 left, bit 1 right), stored beside the task's list entry.
__global__ __launch_bounds__(KD_CHUNK) void kd_classify_kernel(const hxr_triangle* __restrict__ tris, const double* __restrict__ verts, const KdTask* __restrict__ tasks_in,
                                                               int side_in, const int32_t* __restrict__ list_in, const KdChunk* __restrict__ chunks_in,
                                                               uint8_t* __restrict__ sides, const KdCtl* __restrict__ ctl)
{
	const uint32_t n_chunks = ctl->side[side_in][2];
	for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
		const KdChunk ch = chunks_in[c];
		const KdTask T = tasks_in[ch.task];
		const uint32_t i = ch.index * KD_CHUNK + threadIdx.x;
		if (i >= (uint32_t)T.count) continue;
		int axis;
		double split;
		BoxD L, R;
		kd_halves(T, axis, split, L, R);
		sides[T.first + i] = (uint8_t)kd_sides_of(tris, verts, list_in[T.first + i], L, R);
	}
}

template <int KD_BLOCK>
__global__ __launch_bounds__(KD_BLOCK) void kd_level_kernel(const hxr_triangle* __restrict__ tris, const double* __restrict__ verts, const KdTask* __restrict__ tasks_in,
                                                            int side_in, const int32_t* __restrict__ list_in, const uint8_t* __restrict__ sides, KdTask* __restrict__ tasks_out,
                                                            int32_t* __restrict__ list_out, KdChunk* __restrict__ chunks_out, hxr_kd_node* __restrict__ nodes,
                                                            int32_t* __restrict__ leaf_idx, KdCtl* __restrict__ ctl, int side_out, KdCaps caps)
{
	__shared__ uint32_t s_claim[4];               // leaf: {first}; split: {list base, first child node, task slot}; [3] = 1 when a capacity ran out
	__shared__ uint32_t s_chunk_base;
	if (blockIdx.x == 0 && threadIdx.x < 4) ctl->side[(side_out + 1) % 3][threadIdx.x] = 0;   // the set the NEXT level fills: nobody reads or claims from it during this one
	__shared__ uint32_t s_left[KD_BLOCK / 64], s_right[KD_BLOCK / 64];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t n_in = ctl->side[side_in][0];   // written by the previous level (stream order): the host does not have to know it
	for (uint32_t ti = blockIdx.x; ti < n_in; ti += gridDim.x) {
		const KdTask T = tasks_in[ti];
		const uint32_t count = (uint32_t)T.count;
		__syncthreads();   // the previous task's claims have been read by everyone
		if (T.count < KD_LEAF_BELOW || T.depth > KD_MAX_DEPTH) {
			if (tid == 0) {
				const uint32_t first = atomicAdd(&ctl->leaf_cursor, count);
				const bool fits = first + count <= caps.leaf;
				if (!fits) atomicOr(&ctl->error, 1u);
				s_claim[0] = first;
				s_claim[3] = fits ? 0 : 1;
				atomicMax(&ctl->max_depth, (uint32_t)T.depth);
				hxr_kd_node& n = nodes[T.node];   // `parent` was written when the pair was claimed
				n.axis = 3;
				n.first = (int32_t)first;
				n.count = T.count;
				n.split = 0;
			}
			__syncthreads();
			if (s_claim[3]) continue;
			const uint32_t first = s_claim[0];
			for (uint32_t i = tid; i < count; i += KD_BLOCK) leaf_idx[first + i] = list_in[T.first + i];
			continue;
		}
		if (tid == 0) {
			const uint32_t base = atomicAdd(&ctl->side[side_out][1], 2 * count);
			const uint32_t kids = atomicAdd(&ctl->node_cursor, 2u);
			const uint32_t slot = atomicAdd(&ctl->side[side_out][0], 2u);
			const bool fits = base + 2 * count <= caps.list && kids + 2 <= caps.nodes && slot + 2 <= caps.tasks;
			if (!fits) atomicOr(&ctl->error, 1u);
			s_claim[0] = base;
			s_claim[1] = kids;
			s_claim[2] = slot;
			s_claim[3] = fits ? 0 : 1;
		}
		__syncthreads();
		if (s_claim[3]) continue;
		const uint32_t base = s_claim[0], kids = s_claim[1], slot = s_claim[2];
		int axis;
		double split;
		BoxD L, R;
		kd_halves(T, axis, split, L, R);
		const bool classified = kd_chunked(T.count, T.depth);   // kd_classify_kernel has been over this task's list
		uint32_t n_left = 0, n_right = 0;
		for (uint32_t c = 0; c < count; c += KD_BLOCK) {
			const uint32_t i = c + tid;
			int tri = 0;
			bool in_left = false, in_right = false;
			if (i < count) {
				tri = list_in[T.first + i];
				const uint32_t touched = classified ? sides[T.first + i] : kd_sides_of(tris, verts, tri, L, R);
				in_left = (touched & 1u) != 0;
				in_right = (touched & 2u) != 0;
			}
			const unsigned long long bl = __ballot(in_left), br = __ballot(in_right), below = (1ull << lane) - 1;
			if (lane == 0) { s_left[wave] = (uint32_t)__popcll(bl); s_right[wave] = (uint32_t)__popcll(br); }
			__syncthreads();
			uint32_t before_l = 0, before_r = 0, total_l = 0, total_r = 0;
			for (int w = 0; w < KD_BLOCK / 64; w++) {
				if (w < wave) { before_l += s_left[w]; before_r += s_right[w]; }
				total_l += s_left[w];
				total_r += s_right[w];
			}
			if (in_left) list_out[base + n_left + before_l + (uint32_t)__popcll(bl & below)] = tri;
			if (in_right) list_out[base + count + n_right + before_r + (uint32_t)__popcll(br & below)] = tri;
			n_left += total_l;
			n_right += total_r;
			__syncthreads();   // the wave sums are rewritten by the next round
		}
		if (tid == 0) {
			hxr_kd_node& n = nodes[T.node];
			n.axis = axis;
			n.first = (int32_t)kids;
			n.count = 0;
			n.split = split;
			nodes[kids].parent = nodes[kids + 1].parent = T.node;
			KdTask a, b;
			for (int c = 0; c < 3; c++) { a.lo[c] = L.lo[c]; a.hi[c] = L.hi[c]; b.lo[c] = R.lo[c]; b.hi[c] = R.hi[c]; }
			a.node = (int32_t)kids;      a.first = (int32_t)base;           a.count = (int32_t)n_left;  a.depth = T.depth + 1;
			b.node = (int32_t)kids + 1;  b.first = (int32_t)(base + count); b.count = (int32_t)n_right; b.depth = T.depth + 1;
			tasks_out[slot] = a;
			tasks_out[slot + 1] = b;
			// a long child is also listed chunk by chunk for the next level's kd_classify_kernel
			const uint32_t ch_a = kd_chunked(a.count, a.depth) ? ((uint32_t)a.count + KD_CHUNK - 1) / KD_CHUNK : 0u;
			const uint32_t ch_b = kd_chunked(b.count, b.depth) ? ((uint32_t)b.count + KD_CHUNK - 1) / KD_CHUNK : 0u;
			uint32_t cb = 0;
			if (ch_a + ch_b) {
				cb = atomicAdd(&ctl->side[side_out][2], ch_a + ch_b);
				if (cb + ch_a + ch_b > caps.chunks) { atomicOr(&ctl->error, 1u); cb = 0xffffffffu; }
			}
			s_claim[0] = ch_a;
			s_claim[1] = ch_b;
			s_chunk_base = cb;
		}
		__syncthreads();
		if (s_chunk_base != 0xffffffffu) {
			const uint32_t ch_a = s_claim[0], ch_b = s_claim[1], cb = s_chunk_base;
			for (uint32_t k = tid; k < ch_a + ch_b; k += KD_BLOCK) chunks_out[cb + k] = k < ch_a ? KdChunk{slot, k} : KdChunk{slot + 1, k - ch_a};
		}
	}
}

// Mesh::prepareTriangles (src/mesh.cpp:370-396): the derived fields of every triangle record, one triangle per lane, written in place.
//
// The records must come out bit for bit as the host loader's (hexray_amd/host/mesh_prep.cpp), which is compiled with the reference's
// Release flags (-O3 -ffast-math -march=haswell): every sum of products below is therefore written as the chain of fused multiply-adds
// that build executes (read off its code; tests/test_gpu_kd_build.py holds the two sides together), with contraction switched off so
// that nothing else fuses.  What -ffast-math contributes beyond fusing: Vector::normalize multiplies by ONE reciprocal of the length,
// and the two 2x2 solves of dNdx / dNdy (src/mesh.cpp:360-368) share one reciprocal of their determinant.
__global__ void prepare_triangles_kernel(hxr_triangle* __restrict__ tris, int64_t n, const double* __restrict__ verts, const double* __restrict__ uvs)
{
#pragma clang fp contract(off)
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		hxr_triangle& t = tris[i];
		double A[3], B[3], C[3], ta[2], tb[2], tc[2];
		for (int k = 0; k < 3; k++) {
			A[k] = verts[3 * (size_t)t.v[0] + k];
			B[k] = verts[3 * (size_t)t.v[1] + k];
			C[k] = verts[3 * (size_t)t.v[2] + k];
		}
		for (int k = 0; k < 2; k++) {
			ta[k] = uvs[3 * (size_t)t.t[0] + k];
			tb[k] = uvs[3 * (size_t)t.t[1] + k];
			tc[k] = uvs[3 * (size_t)t.t[2] + k];
		}
		const double AB[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, AC[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
		// AB x AC: the subtrahend is rounded, the minuend fused
		const double N[3] = {fma(AB[1], AC[2], -(AB[2] * AC[1])), fma(AB[2], AC[0], -(AB[0] * AC[2])), fma(AB[0], AC[1], -(AB[1] * AC[0]))};
		const double kn = 1.0 / sqrt(fma(N[2], N[2], fma(N[0], N[0], N[1] * N[1])));
		// px*tAB + qx*tAC = (1, 0), py*tAB + qy*tAC = (0, 1) by Cramer's rule
		const double tABx = tb[0] - ta[0], tABy = tb[1] - ta[1], tACx = tc[0] - ta[0], tACy = tc[1] - ta[1];
		const double inv = 1.0 / fma(tABx, tACy, -(tABy * tACx));
		const double px = tACy * inv, qx = (ta[1] - tb[1]) * inv, py = (ta[0] - tc[0]) * inv, qy = tABx * inv;   // the negated differences are taken anew (a zero keeps its sign)
		const double dx[3] = {fma(qx, AC[0], px * AB[0]), fma(qx, AC[1], px * AB[1]), fma(qx, AC[2], px * AB[2])};
		const double dy[3] = {fma(qy, AC[0], py * AB[0]), fma(py, AB[1], qy * AC[1]), fma(py, AB[2], qy * AC[2])};
		const double kx = 1.0 / sqrt(fma(dx[2], dx[2], fma(dx[0], dx[0], dx[1] * dx[1])));
		const double ky = 1.0 / sqrt(fma(dy[0], dy[0], fma(dy[1], dy[1], dy[2] * dy[2])));
		for (int k = 0; k < 3; k++) {
			t.AB[k] = AB[k];
			t.AC[k] = AC[k];
			t.ABcrossAC[k] = N[k];
			t.gnormal[k] = N[k] * kn;
			t.dNdx[k] = dx[k] * kx;
			t.dNdy[k] = dy[k] * ky;
		}
	}
}

void say(char* err, size_t n, const char* fmt, ...)
{
	if (!err || !n) return;
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(err, n, fmt, ap);
	va_end(ap);
}

struct DeviceBuffers {   // ONE allocation, sliced (a hipMalloc / hipFree pair costs more than a level of the build)
	void* block = nullptr;
	KdTask* tasks[2] = {nullptr, nullptr};
	int32_t* lists[2] = {nullptr, nullptr};
	KdChunk* chunks[2] = {nullptr, nullptr};
	uint8_t* sides = nullptr;   // kd_classify_kernel's two bits per entry of the level's input list
	hxr_kd_node* nodes = nullptr;
	int32_t* leaf = nullptr;
	KdCtl* ctl = nullptr;
	hipError_t allocate(const KdCaps& caps)
	{
		auto round = [](size_t n) { return (n + 255) & ~(size_t)255; };
		const size_t s_tasks = round((size_t)caps.tasks * sizeof(KdTask)), s_list = round((size_t)caps.list * sizeof(int32_t)),
		             s_nodes = round((size_t)caps.nodes * sizeof(hxr_kd_node)), s_leaf = round((size_t)caps.leaf * sizeof(int32_t)), s_ctl = round(sizeof(KdCtl)),
		             s_chunks = round((size_t)caps.chunks * sizeof(KdChunk)), s_sides = round((size_t)caps.list);
		const hipError_t e = hipMalloc(&block, 2 * s_tasks + 2 * s_list + 2 * s_chunks + s_sides + s_nodes + s_leaf + s_ctl);
		if (e != hipSuccess) return e;
		char* at = (char*)block;
		for (int i = 0; i < 2; i++) { tasks[i] = (KdTask*)at; at += s_tasks; }
		for (int i = 0; i < 2; i++) { lists[i] = (int32_t*)at; at += s_list; }
		for (int i = 0; i < 2; i++) { chunks[i] = (KdChunk*)at; at += s_chunks; }
		sides = (uint8_t*)at; at += s_sides;
		nodes = (hxr_kd_node*)at; at += s_nodes;
		leaf = (int32_t*)at; at += s_leaf;
		ctl = (KdCtl*)at;
		return hipSuccess;
	}
	~DeviceBuffers() { if (block) (void)hipFree(block); }
};

void renumber(const std::vector<hxr_kd_node>& dev_nodes, const std::vector<int32_t>& dev_leaf, DeviceBuiltTree& out);

#define KD_TRY(expr)                                                                                           \
	do {                                                                                                       \
		hipError_t e_ = (expr);                                                                                \
		if (e_ != hipSuccess) {                                                                                \
			say(err, err_len, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
			return HXR_E_HIP;                                                                                  \
		}                                                                                                      \
	} while (0)

// one attempt with the given capacities; *overflow tells the caller to try again with larger ones
int attempt(const hxr_triangle* d_tris, uint32_t n, const double* d_verts, const double lo[3], const double hi[3], const KdCaps& caps, DeviceBuiltTree& out,
            bool* overflow, char* err, size_t err_len)
{
	*overflow = false;
	DeviceBuffers B;
	KD_TRY(B.allocate(caps));

	const auto t0 = std::chrono::steady_clock::now();
	KdTask root;
	for (int c = 0; c < 3; c++) { root.lo[c] = lo[c]; root.hi[c] = hi[c]; }
	root.node = 0; root.first = 0; root.count = (int32_t)n; root.depth = 0;
	hxr_kd_node root_node;
	memset(&root_node, 0, sizeof root_node);
	root_node.parent = -1;
	KdCtl ctl;
	memset(&ctl, 0, sizeof ctl);
	ctl.side[0][0] = 1;
	ctl.side[0][1] = n;
	ctl.node_cursor = 1;
	if (kd_chunked(root.count, root.depth)) {
		std::vector<KdChunk> root_chunks((n + KD_CHUNK - 1) / KD_CHUNK);
		if (root_chunks.size() > caps.chunks) { *overflow = true; return 0; }
		for (size_t k = 0; k < root_chunks.size(); k++) root_chunks[k] = KdChunk{0u, (uint32_t)k};
		ctl.side[0][2] = (uint32_t)root_chunks.size();
		KD_TRY(hipMemcpy(B.chunks[0], root_chunks.data(), root_chunks.size() * sizeof(KdChunk), hipMemcpyHostToDevice));
	}
	KD_TRY(hipMemcpy(B.tasks[0], &root, sizeof root, hipMemcpyHostToDevice));
	KD_TRY(hipMemcpy(B.nodes, &root_node, sizeof root_node, hipMemcpyHostToDevice));
	KD_TRY(hipMemcpy(B.ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice));
	if (n) hipLaunchKernelGGL(kd_iota_kernel, dim3((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), dim3(256), 0, nullptr, B.lists[0], n);

	// Levels are launched back to back with a fixed grid (a level reads its task count on the device); the host looks at the control
	// words only every KD_LEVELS_PER_LOOK levels, to see the end of the build or a capacity overflow.  Launching a level that turns out
	// empty costs a few microseconds, a look ~20.
	enum { KD_LEVELS_PER_LOOK = 8, KD_GRID = 2048, KD_WIDE_LEVELS = 8 };
	int levels = 0;
	bool done = false;
	while (!done) {
		for (int k = 0; k < KD_LEVELS_PER_LOOK; k++) {
			const int buf_in = levels & 1, buf_out = buf_in ^ 1;              // task / list / chunk buffers: two, swapped
			const int set_in = levels % 3, set_out = (levels + 1) % 3;       // counter sets: three, rotated (the level clears the third itself)
			if (levels < KD_CHUNKED_LEVELS && n >= KD_BIG)   // no task of a later level (or of a small mesh) is listed in chunks
				hipLaunchKernelGGL(kd_classify_kernel, dim3(KD_GRID), dim3(KD_CHUNK), 0, nullptr, d_tris, d_verts, B.tasks[buf_in], set_in, B.lists[buf_in], B.chunks[buf_in],
				                   B.sides, B.ctl);
			// the first levels are few tasks with long lists: 1024 lanes per task there, 256 where the tasks are many and short
			if (levels < KD_WIDE_LEVELS)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_level_kernel<1024>), dim3(KD_GRID / 4), dim3(1024), 0, nullptr, d_tris, d_verts, B.tasks[buf_in], set_in, B.lists[buf_in], B.sides,
				                   B.tasks[buf_out], B.lists[buf_out], B.chunks[buf_out], B.nodes, B.leaf, B.ctl, set_out, caps);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_level_kernel<256>), dim3(KD_GRID), dim3(256), 0, nullptr, d_tris, d_verts, B.tasks[buf_in], set_in, B.lists[buf_in], B.sides,
				                   B.tasks[buf_out], B.lists[buf_out], B.chunks[buf_out], B.nodes, B.leaf, B.ctl, set_out, caps);
			levels++;
		}
		KD_TRY(hipGetLastError());
		KD_TRY(hipMemcpy(&ctl, B.ctl, sizeof ctl, hipMemcpyDeviceToHost));   // the look (synchronises)
		if (ctl.error) { *overflow = true; return 0; }
		done = ctl.side[levels % 3][0] == 0;
		if (levels > KD_MAX_DEPTH + 2 * KD_LEVELS_PER_LOOK) { say(err, err_len, "kd build did not terminate"); return HXR_E_INVALID; }
	}
	out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	out.levels = levels;
	out.max_depth = (int)ctl.max_depth;

	// copy back and renumber in the host builder's order (renumber())
	std::vector<hxr_kd_node> dev_nodes(ctl.node_cursor);
	std::vector<int32_t> dev_leaf(ctl.leaf_cursor);
	KD_TRY(hipMemcpy(dev_nodes.data(), B.nodes, dev_nodes.size() * sizeof(hxr_kd_node), hipMemcpyDeviceToHost));
	if (!dev_leaf.empty()) KD_TRY(hipMemcpy(dev_leaf.data(), B.leaf, dev_leaf.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
	renumber(dev_nodes, dev_leaf, out);
	return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// The surface-area-heuristic tree of hxr_accel.h (hxr_sah::SahBuilder) built on the device, node for node the same tree.
//
// The host builder sorts a node's events per axis and sweeps them.  What the sweep knows at a candidate position p is three counts --
//     A = triangles (not lying in a plane of this axis) that start left of p,      B = planar ones at or left of p,
//     C = non-planar ones that end at or left of p:       left count = A + B,       right count = N - C - B
// -- and those can be counted for every candidate independently: one lane per candidate (two per triangle and axis), every lane
// running over the node's triangles, which the workgroup stages through LDS 256 at a time.  N x 6N comparisons per node instead of a
// sort, but they are spread over the whole GPU in chunks of 256 candidates, whatever the node's size (the root of the teapot: 9120
// triangles, 214 chunks), and each level is two launches: the costs, then -- one workgroup per node -- the choice (the minimum in
// the host's order: smallest cost, then lowest axis, then smallest position, which is what its strict `<` over axes 0, 1, 2 and
// ascending positions keeps), the stable partition and the children.  The cost is formed with the host's expression, operation for
// operation and without contraction, so that ties and near-ties fall the same way: tests/test_gpu_kd_build.py compares the trees.
struct SahTask {   // 80 B
	double lo[3], hi[3];
	int32_t node, first, count, depth;
	uint32_t chunk_first, pad_[3];
};
struct SahChunk { uint32_t task, axis_sub; };   // axis in bits 30..31, which 256 candidates of that axis below
struct SahCtl {
	uint32_t side[2][4];   // per side: {tasks, ints of list space claimed, chunks, unused}
	uint32_t node_cursor, leaf_cursor, error, max_depth;
};
struct SahCaps { uint32_t tasks, list, nodes, leaf, chunks; };
struct SahCosts { double traversal, intersect, empty_bonus; int32_t max_depth, min_leaf; };

__global__ void sah_bounds_kernel(const hxr_triangle* __restrict__ tris, const double* __restrict__ verts, uint32_t n, double* __restrict__ blo, double* __restrict__ bhi)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		for (int c = 0; c < 3; c++) {
			const double a = verts[(size_t)tris[i].v[0] * 3 + c], b = verts[(size_t)tris[i].v[1] * 3 + c], d = verts[(size_t)tris[i].v[2] * 3 + c];
			blo[(size_t)i * 3 + c] = fmin(a, fmin(b, d));
			bhi[(size_t)i * 3 + c] = fmax(a, fmax(b, d));
		}
}

__device__ inline double sah_area(const double lo[3], const double hi[3])
{
#pragma clang fp contract(off)
	const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	return 2 * (dx * dy + dy * dz + dz * dx);
}

// chunks a task of `count` triangles is cut into: per axis, ceil(2 * count / 256)
__device__ __host__ inline uint32_t sah_chunks(uint32_t count) { return 3u * ((2u * count + 255u) / 256u); }

#define SAH_BLOCK 256
__global__ __launch_bounds__(SAH_BLOCK) void sah_cost_kernel(const SahTask* __restrict__ tasks_in, const SahChunk* __restrict__ chunks_in, int side_in,
                                                             const int32_t* __restrict__ list_in, const double* __restrict__ blo, const double* __restrict__ bhi,
                                                             double* __restrict__ costs, const SahCtl* __restrict__ ctl, SahCosts K)
{
#pragma clang fp contract(off)
	__shared__ double s_start[SAH_BLOCK], s_end[SAH_BLOCK];
	const uint32_t n_chunks = ctl->side[side_in][2];
	const int tid = threadIdx.x;
	for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
		const SahChunk ch = chunks_in[c];
		const SahTask T = tasks_in[ch.task];
		const int a = (int)(ch.axis_sub >> 30);
		const uint32_t N = (uint32_t)T.count, k = (ch.axis_sub & 0x3fffffffu) * SAH_BLOCK + (uint32_t)tid;
		const double lo_a = T.lo[a], hi_a = T.hi[a];
		double p = 0;
		if (k < 2 * N) {
			const int tri = list_in[T.first + (k >> 1)];
			const double s = fmax(blo[(size_t)tri * 3 + a], lo_a), e = fmin(bhi[(size_t)tri * 3 + a], hi_a);
			p = (k & 1) ? e : s;
		}
		uint32_t A = 0, B = 0, C = 0;
		for (uint32_t t0 = 0; t0 < N; t0 += SAH_BLOCK) {
			__syncthreads();
			if (t0 + tid < N) {
				const int tri = list_in[T.first + t0 + tid];
				s_start[tid] = fmax(blo[(size_t)tri * 3 + a], lo_a);
				s_end[tid] = fmin(bhi[(size_t)tri * 3 + a], hi_a);
			}
			__syncthreads();
			const uint32_t m = min((uint32_t)SAH_BLOCK, N - t0);
			for (uint32_t q = 0; q < m; q++) {
				const double ss = s_start[q], ee = s_end[q];
				const bool planar = ss == ee;
				A += (!planar && ss < p) ? 1u : 0u;
				B += (planar && ss <= p) ? 1u : 0u;
				C += (!planar && ee <= p) ? 1u : 0u;
			}
		}
		if (k < 2 * N) {
			double cost = HUGE_VAL;
			if (hi_a > lo_a && p > lo_a && p < hi_a) {
				const int cl = (int)(A + B), cr = (int)(N - C - B);
				double l_hi[3] = {T.hi[0], T.hi[1], T.hi[2]}, r_lo[3] = {T.lo[0], T.lo[1], T.lo[2]};
				l_hi[a] = p;
				r_lo[a] = p;
				const double parent_area = sah_area(T.lo, T.hi);
				cost = K.traversal + K.intersect * (sah_area(T.lo, l_hi) * cl + sah_area(r_lo, T.hi) * cr) / parent_area;
				if (cl == 0 || cr == 0) cost *= K.empty_bonus;
			}
			costs[6 * (size_t)T.first + (size_t)a * 2 * N + k] = cost;
		}
	}
}

template <int KD_BLOCK>
__global__ __launch_bounds__(KD_BLOCK) void sah_split_kernel(const SahTask* __restrict__ tasks_in, int side_in, const int32_t* __restrict__ list_in,
                                                             const double* __restrict__ blo, const double* __restrict__ bhi, const double* __restrict__ costs,
                                                             SahTask* __restrict__ tasks_out, SahChunk* __restrict__ chunks_out, int32_t* __restrict__ list_out,
                                                             hxr_kd_node* __restrict__ nodes, int32_t* __restrict__ leaf_idx, SahCtl* __restrict__ ctl, int side_out,
                                                             SahCaps caps, SahCosts K)
{
	__shared__ uint32_t s_claim[6];               // {list base / leaf first, first child node, task slot, overflow, chunk base of the left child, ... of the right child}
	__shared__ uint32_t s_left[KD_BLOCK / 64], s_right[KD_BLOCK / 64];
	__shared__ double s_cost[KD_BLOCK / 64][3], s_pos[KD_BLOCK / 64][3];
	__shared__ double s_best_pos;
	__shared__ int s_best_axis;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t n_in = ctl->side[side_in][0];
	for (uint32_t ti = blockIdx.x; ti < n_in; ti += gridDim.x) {
		const SahTask T = tasks_in[ti];
		const uint32_t count = (uint32_t)T.count;
		__syncthreads();   // the previous task's shared words have been read by everyone
		if (tid == 0) atomicMax(&ctl->max_depth, (uint32_t)T.depth);
		bool leaf = T.count <= K.min_leaf || T.depth >= K.max_depth;
		uint32_t n_left = 0, n_right = 0, base = 0;
		int axis = -1;
		double pos = 0;
		if (!leaf) {
			// the cheapest candidate in the host's order: per axis the smallest cost and, among equals, the smallest position ...
			double bc[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, bp[3] = {0, 0, 0};
			for (int a = 0; a < 3; a++) {
				const double lo_a = T.lo[a], hi_a = T.hi[a];
				for (uint32_t k = tid; k < 2 * count; k += KD_BLOCK) {
					const double c = costs[6 * (size_t)T.first + (size_t)a * 2 * count + k];
					if (c <= bc[a]) {
						const int tri = list_in[T.first + (k >> 1)];
						const double p = (k & 1) ? fmin(bhi[(size_t)tri * 3 + a], hi_a) : fmax(blo[(size_t)tri * 3 + a], lo_a);
						if (c < bc[a] || p < bp[a]) { bc[a] = c; bp[a] = p; }
					}
				}
				for (int off = 32; off > 0; off >>= 1) {
					const double oc = __shfl_down(bc[a], off, 64), op = __shfl_down(bp[a], off, 64);
					if (oc < bc[a] || (oc == bc[a] && op < bp[a])) { bc[a] = oc; bp[a] = op; }
				}
				if (lane == 0) { s_cost[wave][a] = bc[a]; s_pos[wave][a] = bp[a]; }
			}
			__syncthreads();
			if (tid == 0) {
				// ... then the axes in turn, each only if STRICTLY cheaper than what is held (hxr_sah::SahBuilder::split's `c < best_cost`),
				// starting from the cost of leaving the node a leaf
#pragma clang fp contract(off)
				double best = K.intersect * T.count;
				int best_axis = -1;
				double best_pos = 0;
				for (int a = 0; a < 3; a++) {
					double c = HUGE_VAL, p = 0;
					for (int w = 0; w < KD_BLOCK / 64; w++)
						if (s_cost[w][a] < c || (s_cost[w][a] == c && s_pos[w][a] < p)) { c = s_cost[w][a]; p = s_pos[w][a]; }
					if (c < best) { best = c; best_axis = a; best_pos = p; }
				}
				s_best_axis = best_axis;
				s_best_pos = best_pos;
				uint32_t lb = 0;
				bool fits = true;
				if (best_axis >= 0) {
					lb = atomicAdd(&ctl->side[side_out][1], 2 * count);
					fits = lb + 2 * count <= caps.list;
					if (!fits) atomicOr(&ctl->error, 1u);
				}
				s_claim[0] = lb;
				s_claim[3] = fits ? 0 : 1;
			}
			__syncthreads();
			if (s_claim[3]) continue;
			axis = s_best_axis;
			pos = s_best_pos;
			base = s_claim[0];
			if (axis < 0) leaf = true;
		}
		if (!leaf) {
			// the two lists (hxr_sah::SahBuilder::split): a triangle lying in the plane goes left; otherwise left if it starts left of the
			// plane, right if it ends right of it.  Stable, like the host's push_back loop.
			const double lo_a = T.lo[axis], hi_a = T.hi[axis];
			for (uint32_t c = 0; c < count; c += KD_BLOCK) {
				const uint32_t i = c + tid;
				int tri = 0;
				bool in_left = false, in_right = false;
				if (i < count) {
					tri = list_in[T.first + i];
					const double s = fmax(blo[(size_t)tri * 3 + axis], lo_a), e = fmin(bhi[(size_t)tri * 3 + axis], hi_a);
					if (s == e && s == pos) in_left = true;
					else { in_left = s < pos; in_right = e > pos; }
				}
				const unsigned long long bl = __ballot(in_left), br = __ballot(in_right), below = (1ull << lane) - 1;
				if (lane == 0) { s_left[wave] = (uint32_t)__popcll(bl); s_right[wave] = (uint32_t)__popcll(br); }
				__syncthreads();
				uint32_t before_l = 0, before_r = 0, total_l = 0, total_r = 0;
				for (int w = 0; w < KD_BLOCK / 64; w++) {
					if (w < wave) { before_l += s_left[w]; before_r += s_right[w]; }
					total_l += s_left[w];
					total_r += s_right[w];
				}
				if (in_left) list_out[base + n_left + before_l + (uint32_t)__popcll(bl & below)] = tri;
				if (in_right) list_out[base + count + n_right + before_r + (uint32_t)__popcll(br & below)] = tri;
				n_left += total_l;
				n_right += total_r;
				__syncthreads();
			}
			if (n_left == count && n_right == count) leaf = true;   // the split separates nothing: a leaf after all
		}
		if (leaf) {
			__syncthreads();   // every thread has read the split claim (s_claim, s_best_axis) before thread 0 rewrites the words for the leaf
			if (tid == 0) {
				const uint32_t first = atomicAdd(&ctl->leaf_cursor, count);
				const bool fits = first + count <= caps.leaf;
				if (!fits) atomicOr(&ctl->error, 1u);
				s_claim[0] = first;
				s_claim[3] = fits ? 0 : 1;
				hxr_kd_node& n = nodes[T.node];
				n.axis = 3;
				n.first = (int32_t)first;
				n.count = T.count;
				n.split = 0;
			}
			__syncthreads();
			if (s_claim[3]) continue;
			const uint32_t first = s_claim[0];
			for (uint32_t i = tid; i < count; i += KD_BLOCK) leaf_idx[first + i] = list_in[T.first + i];
			continue;
		}
		// the children: node pair, two tasks, and the chunks of those that will be split further
		const bool deeper = T.depth + 1 < K.max_depth;
		const uint32_t ch_l = (deeper && (int)n_left > K.min_leaf) ? sah_chunks(n_left) : 0u, ch_r = (deeper && (int)n_right > K.min_leaf) ? sah_chunks(n_right) : 0u;
		if (tid == 0) {
			const uint32_t kids = atomicAdd(&ctl->node_cursor, 2u);
			const uint32_t slot = atomicAdd(&ctl->side[side_out][0], 2u);
			const uint32_t cb = atomicAdd(&ctl->side[side_out][2], ch_l + ch_r);
			const bool fits = kids + 2 <= caps.nodes && slot + 2 <= caps.tasks && cb + ch_l + ch_r <= caps.chunks;
			if (!fits) atomicOr(&ctl->error, 1u);
			s_claim[1] = kids;
			s_claim[2] = slot;
			s_claim[3] = fits ? 0 : 1;
			s_claim[4] = cb;
			s_claim[5] = cb + ch_l;
			if (fits) {
				hxr_kd_node& n = nodes[T.node];
				n.axis = axis;
				n.first = (int32_t)kids;
				n.count = 0;
				n.split = pos;
				nodes[kids].parent = nodes[kids + 1].parent = T.node;
				SahTask a, b;
				for (int c = 0; c < 3; c++) { a.lo[c] = b.lo[c] = T.lo[c]; a.hi[c] = b.hi[c] = T.hi[c]; }
				a.hi[axis] = pos;
				b.lo[axis] = pos;
				a.node = (int32_t)kids;      a.first = (int32_t)base;           a.count = (int32_t)n_left;  a.depth = T.depth + 1;  a.chunk_first = cb;
				b.node = (int32_t)kids + 1;  b.first = (int32_t)(base + count); b.count = (int32_t)n_right; b.depth = T.depth + 1;  b.chunk_first = cb + ch_l;
				a.pad_[0] = a.pad_[1] = a.pad_[2] = b.pad_[0] = b.pad_[1] = b.pad_[2] = 0;
				tasks_out[slot] = a;
				tasks_out[slot + 1] = b;
			}
		}
		__syncthreads();
		if (s_claim[3]) continue;
		const uint32_t slot = s_claim[2];
		for (int side = 0; side < 2; side++) {
			const uint32_t nch = side ? ch_r : ch_l, first = s_claim[4 + side], per_axis = nch / 3;
			for (uint32_t i = tid; i < nch; i += KD_BLOCK) chunks_out[first + i] = SahChunk{slot + (uint32_t)side, ((i / per_axis) << 30) | (i % per_axis)};
		}
	}
}

struct SahBuffers {
	void* block = nullptr;
	SahTask* tasks[2] = {nullptr, nullptr};
	SahChunk* chunks[2] = {nullptr, nullptr};
	int32_t* lists[2] = {nullptr, nullptr};
	double* costs = nullptr;
	double *blo = nullptr, *bhi = nullptr;
	hxr_kd_node* nodes = nullptr;
	int32_t* leaf = nullptr;
	SahCtl* ctl = nullptr;
	hipError_t allocate(const SahCaps& caps, uint32_t n)
	{
		auto round = [](size_t v) { return (v + 255) & ~(size_t)255; };
		const size_t s_tasks = round((size_t)caps.tasks * sizeof(SahTask)), s_chunks = round((size_t)caps.chunks * sizeof(SahChunk)),
		             s_list = round((size_t)caps.list * sizeof(int32_t)), s_costs = round((size_t)caps.list * 6 * sizeof(double)), s_b = round((size_t)n * 3 * sizeof(double)),
		             s_nodes = round((size_t)caps.nodes * sizeof(hxr_kd_node)), s_leaf = round((size_t)caps.leaf * sizeof(int32_t)), s_ctl = round(sizeof(SahCtl));
		const hipError_t e = hipMalloc(&block, 2 * s_tasks + 2 * s_chunks + 2 * s_list + s_costs + 2 * s_b + s_nodes + s_leaf + s_ctl);
		if (e != hipSuccess) return e;
		char* at = (char*)block;
		costs = (double*)at; at += s_costs;
		blo = (double*)at; at += s_b;
		bhi = (double*)at; at += s_b;
		for (int i = 0; i < 2; i++) { tasks[i] = (SahTask*)at; at += s_tasks; }
		for (int i = 0; i < 2; i++) { chunks[i] = (SahChunk*)at; at += s_chunks; }
		for (int i = 0; i < 2; i++) { lists[i] = (int32_t*)at; at += s_list; }
		nodes = (hxr_kd_node*)at; at += s_nodes;
		leaf = (int32_t*)at; at += s_leaf;
		ctl = (SahCtl*)at;
		return hipSuccess;
	}
	~SahBuffers() { if (block) (void)hipFree(block); }
};

// host order of a device-built tree (shared by the two builders): a node's child pair is appended when the node is split, subtrees left first
void renumber(const std::vector<hxr_kd_node>& dev_nodes, const std::vector<int32_t>& dev_leaf, DeviceBuiltTree& out)
{
	out.nodes.clear();
	out.leaf_indices.clear();
	out.nodes.reserve(dev_nodes.size());
	out.leaf_indices.reserve(dev_leaf.size());
	out.nodes.resize(1);
	out.nodes[0].parent = -1;
	struct Visit { int32_t dev, host; };
	std::vector<Visit> todo{{0, 0}};
	while (!todo.empty()) {
		const Visit v = todo.back();
		todo.pop_back();
		const hxr_kd_node& src = dev_nodes[(size_t)v.dev];
		if (src.axis == 3) {
			hxr_kd_node& dst = out.nodes[(size_t)v.host];
			dst.axis = 3;
			dst.first = (int32_t)out.leaf_indices.size();
			dst.count = src.count;
			dst.split = 0;
			out.leaf_indices.insert(out.leaf_indices.end(), dev_leaf.begin() + src.first, dev_leaf.begin() + src.first + src.count);
			continue;
		}
		const int32_t first = (int32_t)out.nodes.size();
		out.nodes.resize(out.nodes.size() + 2);
		out.nodes[(size_t)first].parent = out.nodes[(size_t)first + 1].parent = v.host;
		hxr_kd_node& dst = out.nodes[(size_t)v.host];
		dst.axis = src.axis;
		dst.first = first;
		dst.count = 0;
		dst.split = src.split;
		todo.push_back({src.first + 1, first + 1});
		todo.push_back({src.first, first});   // left subtree first
	}
}

int sah_attempt(const hxr_triangle* d_tris, uint32_t n, const double* d_verts, const double lo[3], const double hi[3], const SahCaps& caps, const SahCosts& K,
                DeviceBuiltTree& out, bool* overflow, char* err, size_t err_len)
{
	*overflow = false;
	SahBuffers B;
	KD_TRY(B.allocate(caps, n));
	const auto t0 = std::chrono::steady_clock::now();
	SahTask root;
	memset(&root, 0, sizeof root);
	for (int c = 0; c < 3; c++) { root.lo[c] = lo[c]; root.hi[c] = hi[c]; }
	root.node = 0; root.first = 0; root.count = (int32_t)n; root.depth = 0; root.chunk_first = 0;
	const bool root_splits = (int)n > K.min_leaf && 0 < K.max_depth;
	const uint32_t root_chunks = root_splits ? sah_chunks(n) : 0u;
	if (root_chunks > caps.chunks) { *overflow = true; return 0; }
	std::vector<SahChunk> rc(root_chunks);
	for (uint32_t i = 0; i < root_chunks; i++) rc[i] = SahChunk{0u, ((i / (root_chunks / 3)) << 30) | (i % (root_chunks / 3))};
	hxr_kd_node root_node;
	memset(&root_node, 0, sizeof root_node);
	root_node.parent = -1;
	SahCtl ctl;
	memset(&ctl, 0, sizeof ctl);
	ctl.side[0][0] = 1;
	ctl.side[0][1] = n;
	ctl.side[0][2] = root_chunks;
	ctl.node_cursor = 1;
	KD_TRY(hipMemcpy(B.tasks[0], &root, sizeof root, hipMemcpyHostToDevice));
	if (root_chunks) KD_TRY(hipMemcpy(B.chunks[0], rc.data(), rc.size() * sizeof(SahChunk), hipMemcpyHostToDevice));
	KD_TRY(hipMemcpy(B.nodes, &root_node, sizeof root_node, hipMemcpyHostToDevice));
	KD_TRY(hipMemcpy(B.ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice));
	if (n) {
		const int g = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
		hipLaunchKernelGGL(kd_iota_kernel, dim3(g), dim3(256), 0, nullptr, B.lists[0], n);
		hipLaunchKernelGGL(sah_bounds_kernel, dim3(g), dim3(256), 0, nullptr, d_tris, d_verts, n, B.blo, B.bhi);
	}
	enum { LEVELS_PER_LOOK = 8, GRID = 2048, WIDE_LEVELS = 6 };
	int side_in = 0, levels = 0;
	bool done = false;
	while (!done) {
		for (int k = 0; k < LEVELS_PER_LOOK; k++) {
			const int side_out = side_in ^ 1;
			KD_TRY(hipMemsetAsync(&B.ctl->side[side_out][0], 0, 4 * sizeof(uint32_t), nullptr));
			hipLaunchKernelGGL(sah_cost_kernel, dim3(GRID), dim3(SAH_BLOCK), 0, nullptr, B.tasks[side_in], B.chunks[side_in], side_in, B.lists[side_in], B.blo, B.bhi, B.costs,
			                   B.ctl, K);
			if (levels < WIDE_LEVELS)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(sah_split_kernel<1024>), dim3(GRID / 4), dim3(1024), 0, nullptr, B.tasks[side_in], side_in, B.lists[side_in], B.blo, B.bhi,
				                   B.costs, B.tasks[side_out], B.chunks[side_out], B.lists[side_out], B.nodes, B.leaf, B.ctl, side_out, caps, K);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(sah_split_kernel<256>), dim3(GRID), dim3(256), 0, nullptr, B.tasks[side_in], side_in, B.lists[side_in], B.blo, B.bhi, B.costs,
				                   B.tasks[side_out], B.chunks[side_out], B.lists[side_out], B.nodes, B.leaf, B.ctl, side_out, caps, K);
			side_in = side_out;
			levels++;
		}
		KD_TRY(hipGetLastError());
		KD_TRY(hipMemcpy(&ctl, B.ctl, sizeof ctl, hipMemcpyDeviceToHost));   // the look (synchronises)
		if (ctl.error) { *overflow = true; return 0; }
		done = ctl.side[side_in][0] == 0;
		if (levels > K.max_depth + 2 * LEVELS_PER_LOOK + 2) { say(err, err_len, "SAH build did not terminate"); return HXR_E_INVALID; }
	}
	out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	out.levels = levels;
	out.max_depth = (int)ctl.max_depth;
	std::vector<hxr_kd_node> dev_nodes(ctl.node_cursor);
	std::vector<int32_t> dev_leaf(ctl.leaf_cursor);
	KD_TRY(hipMemcpy(dev_nodes.data(), B.nodes, dev_nodes.size() * sizeof(hxr_kd_node), hipMemcpyDeviceToHost));
	if (!dev_leaf.empty()) KD_TRY(hipMemcpy(dev_leaf.data(), B.leaf, dev_leaf.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
	renumber(dev_nodes, dev_leaf, out);
	return 0;
}

}  // namespace

int prepare_triangles(hxr_triangle* d_tris, int64_t num_triangles, const double* d_vertices, const double* d_uvs)
{
	if (num_triangles <= 0) return 0;
	const int grid = (int)std::min<int64_t>((num_triangles + 255) / 256, 4096);
	hipLaunchKernelGGL(prepare_triangles_kernel, dim3(grid), dim3(256), 0, nullptr, d_tris, num_triangles, d_vertices, d_uvs);
	return hipGetLastError() == hipSuccess ? 0 : HXR_E_HIP;
}

int build_reference_tree(const hxr_triangle* d_tris, int64_t num_triangles, const double* d_vertices, const double bbox_min[3], const double bbox_max[3],
                         DeviceBuiltTree& out, char* err, size_t err_len)
{
	if (num_triangles < 0 || num_triangles > (1ll << 26)) { say(err, err_len, "kd build: %lld triangles not supported", (long long)num_triangles); return HXR_E_INVALID; }
	const uint32_t n = (uint32_t)num_triangles;
	// the reference's meshes end at 0.2-0.7 nodes and 2.5-5 leaf entries per triangle; a level's lists hold at most twice its parent level's
	uint64_t scale = 1;
	for (int tries = 0; tries < 4; tries++, scale *= 4) {
		const uint64_t tasks = std::min<uint64_t>((4ull * n + 4096) * scale, 1ull << 27), list = std::min<uint64_t>((24ull * n + 65536) * scale, 1ull << 31),
		               nodes = std::min<uint64_t>((8ull * n + 4096) * scale, 1ull << 28), leaf = std::min<uint64_t>((32ull * n + 65536) * scale, 1ull << 31);
		KdCaps caps{(uint32_t)tasks, (uint32_t)(list - 1), (uint32_t)nodes, (uint32_t)(leaf - 1), (uint32_t)(list / (KD_CHUNK / 2) + 4096)};
		bool overflow = false;
		const int rc = attempt(d_tris, n, d_vertices, bbox_min, bbox_max, caps, out, &overflow, err, err_len);
		if (rc) return rc;
		if (!overflow) return 0;
	}
	say(err, err_len, "kd build: the tree of this mesh outgrows the build buffers (a degenerate mesh?)");
	return HXR_E_INVALID;
}

int build_sah_tree(const hxr_triangle* d_tris, int64_t num_triangles, const double* d_vertices, const double bbox_min[3], const double bbox_max[3],
                   double traversal_cost, double intersect_cost, double empty_bonus, int max_depth, int min_leaf, DeviceBuiltTree& out, char* err, size_t err_len)
{
	// quadratic in a node's triangles: the callers send larger meshes to the host builder (HXR_SAH_DEVICE_MAX_TRIANGLES, hxr_api.hip); 2^17 is the most ever run here
	if (num_triangles < 0 || num_triangles > (1ll << 17)) { say(err, err_len, "SAH build on the device: %lld triangles not supported", (long long)num_triangles); return HXR_E_INVALID; }
	const uint32_t n = (uint32_t)num_triangles;
	const SahCosts K{traversal_cost, intersect_cost, empty_bonus, max_depth, min_leaf};
	uint64_t scale = 1;
	for (int tries = 0; tries < 4; tries++, scale *= 4) {
		// a level's lists hold at most twice its parent level's; SAH trees of the reference's meshes end below 2 nodes and 4 leaf entries per triangle
		const uint64_t tasks = std::min<uint64_t>((8ull * n + 4096) * scale, 1ull << 26), list = std::min<uint64_t>((16ull * n + 65536) * scale, 1ull << 29),
		               nodes = std::min<uint64_t>((16ull * n + 4096) * scale, 1ull << 27), leaf = std::min<uint64_t>((16ull * n + 65536) * scale, 1ull << 30);
		const uint64_t chunks = 3 * (2 * list / 256 + tasks) + 64;
		SahCaps caps{(uint32_t)tasks, (uint32_t)(list - 1), (uint32_t)nodes, (uint32_t)(leaf - 1), (uint32_t)std::min<uint64_t>(chunks, 1ull << 30)};
		bool overflow = false;
		const int rc = sah_attempt(d_tris, n, d_vertices, bbox_min, bbox_max, caps, K, out, &overflow, err, err_len);
		if (rc) return rc;
		if (!overflow) return 0;
	}
	say(err, err_len, "SAH build: the tree of this mesh outgrows the build buffers (a degenerate mesh?)");
	return HXR_E_INVALID;
}

}  // namespace hxr_kd
