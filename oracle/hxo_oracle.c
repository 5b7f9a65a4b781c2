/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * hxo_oracle.c: a plain-C, CPU restatement of heX-Ray's per-pixel / per-sample raytrace loop, used
 * ONLY as the checker in tests/, in __graft_entry__.smoke() and as the `cpu_baseline` leg of
 * bench.py.  Nothing under hexray_amd/ links, loads or calls it; the product path is the HIP module
 * (hexray_amd/csrc) and fails loudly without a GPU.
 *
 * It consumes the same flattened scene (include/hexray_hip.h: hxr_scene_desc) the HIP module
 * consumes, and follows the reference's evaluation structure closely (recursive shading, recursive
 * kd descent that carries the child box, linear scans over nodes and lights), unlike the GPU path
 * which is iterative / stackless / re-laid-out.  Every function cites the reference code it restates
 * (paths relative to the reference checkout, /root/reference).
 *
 * Pinning status (details in DESIGN.md section "Oracle"):
 *   - the full reference program cannot be built here (it needs SDL2 and OpenEXR headers that the
 *     image lacks, and no stand-ins may be written), so whole-frame parity against the reference
 *     binary is UNPINNED;
 *   - pinned pieces, against the reference's own translation units compiled in place (oracle/_ref, golden vectors in
 *     tests/golden/): vector/matrix/transform/colour/sRGB math, BBox::testIntersect / inside / closestIntersection / split, the
 *     analytic primitives, CSG, Node::intersect, the lights (ref_harness.cpp); CheckerTexture / Fresnel / Bumps, Reflection (mirror
 *     and glossy) / Refraction / Layered ::computeColor with the secondary rays they ask for, the specular BRDFs' eval / spawnRay,
 *     unitDiskSample, hemisphereSample (ref_harness_shading.cpp, linked against the reference's shading.cpp and util.cpp);
 *   - count-pinned only (the reference's recorded kd statistics and per-frame call counts): the kd
 *     builder, Mesh::intersectTriangle / intersectKD, Lambert / Phong::computeColor and the whole traversal / shading pipeline;
 *   - unpinned: Heightfield, Bitmap-backed lookups (BitmapTexture, BumpTexture, cubemap), Camera -- their translation units need
 *     SDL2 / OpenEXR headers.
 *
 * Random numbers: see hxo_rng.h (counter-based replacement of the thread-local mt19937).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "hexray_hip.h"
#include "hxo_rng.h"

#define HXO_PI 3.141592653589793238 /* src/constants.h:28 */
#define HXO_INF 1e99                /* src/constants.h:29 */
#define RF_GI_DIFFUSE 0x0002u       /* src/vector.h:176 */

typedef struct { double x, y, z; } vec;
typedef struct { float r, g, b; } col;

typedef struct {
	vec start, dir;
	int depth;
	unsigned flags;
	uint32_t key; /* RNG ray key, hxo_rng.h */
} ray_t;

/* IntersectionInfo, src/geometry.h:33-40; `geom` is the geometry table index */
typedef struct {
	double dist;
	vec ip, norm, dNdx, dNdy;
	double u, v;
	int geom;
} hit_t;

typedef struct {
	const hxr_scene_desc* sc;
	int W, H;
	uint64_t seed;
	uint32_t pixel, sample; /* RNG counter words of the camera sample being traced */
	double focal_plane_dist; /* Camera::focalPlaneDist, possibly replaced by the autofocus probe */
	hxr_counters cnt;
	/* unit-level probing (hxo_shade_probe): raytrace() calls are recorded instead of traced, like the recording raytrace() the
	   reference's shading.cpp is linked against in oracle/ref_harness_shading.cpp */
	double* probe_rays; /* 8 doubles per recorded ray: start, dir, depth, flags */
	int probe_max, probe_count, probe_on;
	/* ... and visible() calls are recorded and answered from a pattern, draws come from a list (hxo_shade_probe2: the recording
	   visible() and the recorded generator of `ref_harness_shading diffuse`) */
	uint64_t probe_vis_pattern;
	double* probe_vis_points; /* 6 doubles per recorded call: A, B */
	int probe_vis_max, probe_vis_count;
	const double* probe_draws; /* the numbers randDouble() handed out, in order; NULL = the counter-based generator */
	int probe_draws_len, probe_draws_used;
} ctx_t;

/* two uniform draws: from the probe's list when there is one (unit-level tests replay the reference generator's numbers), else from the
   counter-based generator under the given key / dimension */
static inline void draw2(ctx_t* cx, uint32_t key, uint32_t dim, double* u, double* v);

/* ------------------------------------------------------------------ vector / colour helpers */
static inline vec V(double x, double y, double z) { vec r = {x, y, z}; return r; }
static inline vec vld(const double* p) { return V(p[0], p[1], p[2]); }
static inline vec vadd(vec a, vec b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec vsub(vec a, vec b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec vneg(vec a) { return V(-a.x, -a.y, -a.z); }
static inline vec vmul(vec a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline double vdot(vec a, vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec vcross(vec a, vec b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline double vlen(vec a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline double vlen2(vec a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline double vget(vec a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
/* Vector::normalize, src/vector.h:79-83 */
static inline vec vunit(vec a) { return vmul(a, 1.0 / vlen(a)); }
/* free normalize(), src/vector.h:160-165 */
static inline vec vunit_lazy(vec a)
{
	double len = vlen(a);
	if (fabs(len - 1.0) < 1e-6) return a;
	return vmul(a, 1 / len);
}
static inline double vdist(vec a, vec b)
{
	return sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z));
}
/* faceforward / reflect, src/vector.h:179-187 */
static inline vec vfaceforward(vec ray, vec n) { return vdot(ray, n) < 0 ? n : vneg(n); }
static inline vec vreflect(vec i, vec n) { return vadd(vmul(n, 2 * vdot(vneg(i), n)), i); }
/* v * M for a row-major 3x3, src/matrix.h:53-60 */
static inline vec vxm(vec v, const double* m)
{
	return V(v.x * m[0] + v.y * m[3] + v.z * m[6], v.x * m[1] + v.y * m[4] + v.z * m[7], v.x * m[2] + v.y * m[5] + v.z * m[8]);
}

static inline col C(float r, float g, float b) { col c = {r, g, b}; return c; }
static inline col cld(const float* p) { return C(p[0], p[1], p[2]); }
static inline col cadd(col a, col b) { return C(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline col csub(col a, col b) { return C(a.r - b.r, a.g - b.g, a.b - b.b); }
static inline col cmul(col a, col b) { return C(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline col cscale(col a, float s) { return C(a.r * s, a.g * s, a.b * s); }
/* Color / float, src/color.h:195-200 */
static inline col cdiv(col a, float d)
{
	if (d == 1) return a;
	float m = 1.0f / d;
	return C(a.r * m, a.g * m, a.b * m);
}
static inline float cintensity(col a) { return (a.r + a.g + a.b) / 3; }
static inline int czero(col a) { return a.r == 0 && a.g == 0 && a.b == 0; }

/* ------------------------------------------------------------------ transforms */
/* Transform::untransformPoint / untransformDir / transformPoint / normal / transformDir, src/matrix.cpp:156-187 */
static inline vec xf_unpoint(const hxr_transform* T, vec p) { return vxm(vsub(p, vld(T->offset)), T->inv_m); }
static inline vec xf_undir(const hxr_transform* T, vec d) { return vunit_lazy(vxm(d, T->inv_m)); }
static inline vec xf_point(const hxr_transform* T, vec p) { return vadd(vxm(p, T->m), vld(T->offset)); }
static inline vec xf_normal(const hxr_transform* T, vec n) { return vxm(n, T->tinv_m); }
static inline vec xf_dir(const hxr_transform* T, vec d) { return vunit_lazy(vxm(d, T->m)); }

/* ------------------------------------------------------------------ boxes */
typedef struct { vec lo, hi; } box_t;

/* BBox::inside, src/bbox.h:78-85 */
static inline int box_holds(const box_t* b, vec p)
{
	return b->lo.x - 1e-6 <= p.x && p.x <= b->hi.x + 1e-6 && b->lo.y - 1e-6 <= p.y && p.y <= b->hi.y + 1e-6 &&
	       b->lo.z - 1e-6 <= p.z && p.z <= b->hi.z + 1e-6;
}

/* BBox::testIntersect, src/bbox.h:94-141 */
static int box_hit(ctx_t* cx, const box_t* b, const ray_t* ray)
{
	cx->cnt.bbox_tests++;
	if (box_holds(b, ray->start)) return 1;
	for (int a = 0; a < 3; a++) {
		double d = vget(ray->dir, a), o = vget(ray->start, a);
		double lo = vget(b->lo, a), hi = vget(b->hi, a);
		if ((d < 0 && o < lo) || (d > 0 && o > hi)) return 0;
		if (fabs(d) < 1e-9) continue;
		double mul = 1 / d;
		int u = (a == 0) ? 1 : 0, v = (a == 2) ? 1 : 2;
		double t = (lo - o) * mul;
		if (t < 0) continue; /* also skips the far wall of this axis */
		double pu = vget(ray->start, u) + vget(ray->dir, u) * t;
		if (vget(b->lo, u) <= pu && pu <= vget(b->hi, u)) {
			double pv = vget(ray->start, v) + vget(ray->dir, v) * t;
			if (vget(b->lo, v) <= pv && pv <= vget(b->hi, v)) return 1;
		}
		t = (hi - o) * mul;
		if (t < 0) continue;
		pu = vget(ray->start, u) + vget(ray->dir, u) * t;
		if (vget(b->lo, u) <= pu && pu <= vget(b->hi, u)) {
			double pv = vget(ray->start, v) + vget(ray->dir, v) * t;
			if (vget(b->lo, v) <= pv && pv <= vget(b->hi, v)) return 1;
		}
	}
	return 0;
}

/* BBox::closestIntersection (+Periphery), src/bbox.h:145-174 */
static double box_nearest(const box_t* b, const ray_t* ray)
{
	if (box_holds(b, ray->start)) return 0;
	double best = HXO_INF;
	for (int a = 0; a < 3; a++) {
		double d = vget(ray->dir, a), o = vget(ray->start, a);
		if ((d < 0 && o < vget(b->lo, a)) || (d > 0 && o > vget(b->hi, a))) return HXO_INF;
		if (fabs(d) < 1e-9) continue;
		double mul = 1 / d;
		double walls[2] = {vget(b->lo, a), vget(b->hi, a)};
		int u = (a == 0) ? 1 : 0, v = (a == 2) ? 1 : 2;
		for (int j = 0; j < 2; j++) {
			double t = (walls[j] - o) * mul;
			if (t < 0) continue;
			double pu = vget(ray->start, u) + vget(ray->dir, u) * t;
			if (vget(b->lo, u) <= pu && pu <= vget(b->hi, u)) {
				double pv = vget(ray->start, v) + vget(ray->dir, v) * t;
				if (vget(b->lo, v) <= pv && pv <= vget(b->hi, v)) best = t < best ? t : best;
			}
		}
	}
	return best;
}

/* ------------------------------------------------------------------ analytic primitives */
/* Plane::intersect, src/geometry.cpp:31-50 */
static int plane_hit(const hxr_geometry* g, int gi, const ray_t* ray, hit_t* h)
{
	double py = g->p[0], limit = g->p[1];
	if (ray->start.y > py && ray->dir.y >= 0) return 0;
	if (ray->start.y < py && ray->dir.y <= 0) return 0;
	double t = (py - ray->start.y) / ray->dir.y;
	h->dist = t;
	h->ip = vadd(ray->start, vmul(ray->dir, t));
	if (fabs(h->ip.x) > limit || fabs(h->ip.z) > limit) return 0;
	h->norm = V(0, (ray->start.y > py) ? 1 : -1, 0);
	h->u = h->ip.x;
	h->v = h->ip.z;
	h->dNdx = V(1, 0, 0);
	h->dNdy = V(0, 0, 1);
	h->geom = gi;
	return 1;
}

/* Sphere::intersect, src/geometry.cpp:52-85 */
static int sphere_hit(const hxr_geometry* g, int gi, const ray_t* ray, hit_t* h)
{
	vec O = vld(g->p);
	double R = g->p[3], uvscaling = g->p[4];
	double A = vlen2(ray->dir);
	vec Hh = vsub(ray->start, O);
	double B = 2 * vdot(ray->dir, Hh);
	double Cc = vlen2(Hh) - R * R;
	double D = B * B - 4 * A * Cc;
	if (D < 0) return 0;
	double sD = sqrt(D);
	double p1 = (-B - sD) / (2 * A), p2 = (-B + sD) / (2 * A);
	if (p2 < 0) return 0;
	double p = (p1 < 0) ? p2 : p1;
	h->dist = p;
	h->ip = vadd(ray->start, vmul(ray->dir, p));
	h->norm = vunit(vsub(h->ip, O));
	h->v = asin(h->norm.y);
	h->u = atan2(h->norm.z, h->norm.x);
	h->v = -(h->v / HXO_PI + 0.5f);
	h->u = h->u / (2 * HXO_PI) + 0.5f;
	if (uvscaling != 1) {
		h->u *= uvscaling;
		h->v *= uvscaling;
	}
	h->geom = gi;
	return 1;
}

/* Cube::intersect + intersectCubeSide, src/geometry.cpp:87-147 */
static int cube_hit(const hxr_geometry* g, int gi, const ray_t* ray, hit_t* h)
{
	vec O = vld(g->p);
	double half = g->p[3];
	int found = 0;
	h->dist = HXO_INF;
	for (int a = 0; a < 3; a++)
		for (int s = 0; s < 2; s++) {
			double o = vget(ray->start, a), d = vget(ray->dir, a);
			double target = vget(O, a) + (s ? half : -half);
			if (fabs(d) < 1e-9) continue;
			if (o < target && d < 0) continue;
			if (o > target && d > 0) continue;
			double p = (target - o) / d;
			if (!(p < h->dist)) continue;
			vec ip = vadd(ray->start, vmul(ray->dir, p));
			if (!(ip.x > O.x - half - 1e-6 && ip.x < O.x + half + 1e-6)) continue;
			if (!(ip.y > O.y - half - 1e-6 && ip.y < O.y + half + 1e-6)) continue;
			if (!(ip.z > O.z - half - 1e-6 && ip.z < O.z + half + 1e-6)) continue;
			h->dist = p;
			h->ip = ip;
			h->norm = V(a == 0 ? (s ? 1 : -1) : 0, a == 1 ? (s ? 1 : -1) : 0, a == 2 ? (s ? 1 : -1) : 0);
			if (a == 0) { h->u = ip.y; h->v = ip.z; }
			else if (a == 1) { h->u = ip.x; h->v = ip.z; }
			else { h->u = ip.x; h->v = ip.y; }
			h->geom = gi;
			found++;
		}
	return found > 0;
}

/* ------------------------------------------------------------------ meshes */
static inline vec pool_vec(const hxr_scene_desc* sc, int64_t first, int64_t e) { return vld(sc->f64_pool + (first + e) * 3); }

/* Mesh::intersectTriangle, src/mesh.cpp:178-219 */
static int mesh_triangle(ctx_t* cx, const hxr_mesh* m, const hxr_triangle* t, const ray_t* ray, hit_t* h)
{
	const hxr_scene_desc* sc = cx->sc;
	cx->cnt.tri_tests++;
	if (m->backface_culling && vdot(ray->dir, vld(t->gnormal)) > 0) return 0;
	vec A = pool_vec(sc, m->first_vertex, t->v[0]);
	vec AB = vld(t->AB), AC = vld(t->AC), N = vld(t->ABcrossAC);
	vec Hh = vsub(ray->start, A);
	vec nd = vneg(ray->dir);
	double Dcr = vdot(N, nd);
	if (fabs(Dcr) < 1e-12) return 0;
	double rD = 1 / Dcr;
	double gamma = vdot(N, Hh) * rD;
	if (gamma < 0 || gamma > h->dist) return 0;
	double l2 = vdot(vcross(Hh, AC), nd) * rD;
	if (l2 < 0 || l2 > 1) return 0;
	double l3 = vdot(vcross(AB, Hh), nd) * rD;
	if (l3 < 0 || l3 > 1) return 0;
	double l1 = 1 - (l2 + l3);
	if (l1 < 0 || l1 > 1) return 0;
	cx->cnt.mesh_hits++;
	h->dist = gamma;
	h->ip = vadd(ray->start, vmul(ray->dir, gamma));
	vec tA = pool_vec(sc, m->first_uv, t->t[0]), tB = pool_vec(sc, m->first_uv, t->t[1]), tC = pool_vec(sc, m->first_uv, t->t[2]);
	vec tc = vadd(vadd(tA, vmul(vsub(tB, tA), l2)), vmul(vsub(tC, tA), l3));
	h->u = tc.x;
	h->v = tc.y;
	if (m->faceted) {
		h->norm = vld(t->gnormal);
	} else {
		vec nA = pool_vec(sc, m->first_normal, t->n[0]), nB = pool_vec(sc, m->first_normal, t->n[1]), nC = pool_vec(sc, m->first_normal, t->n[2]);
		h->norm = vunit(vadd(vadd(nA, vmul(vsub(nB, nA), l2)), vmul(vsub(nC, nA), l3)));
	}
	h->dNdx = vld(t->dNdx);
	h->dNdy = vld(t->dNdy);
	return 1;
}

/* Mesh::intersectKD, src/mesh.cpp:221-245: recursive, carries the node's box */
/* BBox::split, src/bbox.h:210-216 */
static void box_split(const box_t* box, int axis, double where, box_t* left, box_t* right)
{
	*left = *box;
	*right = *box;
	if (axis == 0) { left->hi.x = where; right->lo.x = where; }
	else if (axis == 1) { left->hi.y = where; right->lo.y = where; }
	else { left->hi.z = where; right->lo.z = where; }
}

static int mesh_kd(ctx_t* cx, const hxr_mesh* m, int node, const box_t* box, const ray_t* ray, hit_t* h)
{
	const hxr_scene_desc* sc = cx->sc;
	const hxr_kd_node* n = sc->kd_nodes + m->first_kd_node + node;
	cx->cnt.kd_visits++;
	if (n->axis == 3) {
		int found = 0;
		const int32_t* list = sc->leaf_indices + m->first_leaf_index + n->first;
		for (int i = 0; i < n->count; i++) {
			cx->cnt.leaf_indices++;
			const hxr_triangle* t = sc->triangles + m->first_triangle + list[i];
			if (mesh_triangle(cx, m, t, ray, h) && box_holds(box, h->ip)) found = 1;
		}
		return found;
	}
	box_t kid[2];
	box_split(box, n->axis, n->split, &kid[0], &kid[1]);
	int first = vget(ray->start, n->axis) > n->split ? 1 : 0;
	for (int i = 0; i < 2; i++) {
		int c = i == 0 ? first : 1 - first;
		if (box_hit(cx, &kid[c], ray))
			if (mesh_kd(cx, m, n->first + c, &kid[c], ray, h)) return 1;
	}
	return 0;
}

/* Mesh::intersect, src/mesh.cpp:247-263 */
static int mesh_hit(ctx_t* cx, const hxr_geometry* g, int gi, const ray_t* ray, hit_t* h)
{
	const hxr_mesh* m = cx->sc->meshes + g->ref;
	box_t root = {vld(m->bbox_min), vld(m->bbox_max)};
	if (!box_hit(cx, &root, ray)) return 0;
	h->dist = HXO_INF;
	if (m->has_kd) return mesh_kd(cx, m, 0, &root, ray, h);
	for (int64_t i = 0; i < m->num_triangles; i++) mesh_triangle(cx, m, cx->sc->triangles + m->first_triangle + i, ray, h);
	if (h->dist < HXO_INF) {
		h->geom = gi;
		return 1;
	}
	return 0;
}

/* ------------------------------------------------------------------ heightfield */
/* intersectTriangleFast, src/mesh.cpp:131-170 */
static int tri_fast(const ray_t* ray, vec A, vec B, vec Cv, double* dist)
{
	vec AB = vsub(B, A), AC = vsub(Cv, A), D = vneg(ray->dir), Hh = vsub(ray->start, A);
	vec N = vcross(AB, AC);
	double Dcr = vdot(N, D);
	if (fabs(Dcr) < 1e-12) return 0;
	double l2 = vdot(vcross(Hh, AC), D) / Dcr;
	double l3 = vdot(vcross(AB, Hh), D) / Dcr;
	double gamma = vdot(N, Hh) / Dcr;
	if (gamma < 0 || gamma > *dist) return 0;
	if (l2 < 0 || l2 > 1 || l3 < 0 || l3 > 1 || l2 + l3 > 1) return 0;
	*dist = gamma;
	return 1;
}

static inline int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* Heightfield::getHeight / getHighest, src/heightfield.cpp:30-47 */
static inline float hf_height(ctx_t* cx, const hxr_heightfield* f, int x, int y)
{
	cx->cnt.hf_height++;
	return cx->sc->f32_pool[f->first_height + (int64_t)clampi(y, f->H - 1) * f->W + clampi(x, f->W - 1)];
}
static inline float hf_highest(ctx_t* cx, const hxr_heightfield* f, int x, int y, int k)
{
	cx->cnt.hf_highest++;
	return cx->sc->f32_pool[f->first_highmap + ((int64_t)clampi(y, f->H - 1) * f->W + clampi(x, f->W - 1)) * 16 + k];
}

/* Heightfield::getNormal, src/heightfield.cpp:83-104 */
static vec hf_normal(ctx_t* cx, const hxr_heightfield* f, float x, float y)
{
	cx->cnt.hf_normals++;
	int x0 = (int)floorf(x), y0 = (int)floorf(y);
	float p = x - x0, q = y - y0;
	int x1 = x0 + 1 < f->W - 1 ? x0 + 1 : f->W - 1;
	int y1 = y0 + 1 < f->H - 1 ? y0 + 1 : f->H - 1;
	x0 = clampi(x0, f->W - 1);
	y0 = clampi(y0, f->H - 1);
	const hxr_scene_desc* sc = cx->sc;
	vec n00 = pool_vec(sc, f->first_normal, (int64_t)y0 * f->W + x0), n01 = pool_vec(sc, f->first_normal, (int64_t)y0 * f->W + x1);
	vec n10 = pool_vec(sc, f->first_normal, (int64_t)y1 * f->W + x0), n11 = pool_vec(sc, f->first_normal, (int64_t)y1 * f->W + x1);
	/* the reference sums left to right: ((a + b) + c) + d */
	vec v = vadd(vadd(vadd(vmul(n00, (1 - p) * (1 - q)), vmul(n01, p * (1 - q))), vmul(n10, (1 - p) * q)), vmul(n11, p * q));
	return vunit(v);
}

/* Heightfield::intersect, src/heightfield.cpp:106-171 */
static int heightfield_hit(ctx_t* cx, const hxr_geometry* g, int gi, const ray_t* ray, hit_t* h)
{
	const hxr_heightfield* f = cx->sc->heightfields + g->ref;
	cx->cnt.hf_calls++;
	box_t box = {vld(f->bbox_min), vld(f->bbox_max)};
	vec step = ray->dir;
	double horiz = sqrt(step.x * step.x + step.z * step.z);
	step = vmul(step, 1.0 / horiz);
	double dist = box_nearest(&box, ray);
	vec p = vadd(ray->start, vmul(ray->dir, dist + 1e-6));
	double mx = 1.0 / ray->dir.x, mz = 1.0 / ray->dir.z;
	while (box_holds(&box, p)) {
		int x0 = (int)floor(p.x), z0 = (int)floor(p.z);
		if (x0 < 0 || x0 >= f->W || z0 < 0 || z0 >= f->H) break;
		if (f->use_optimization) {
			int k = 1;
			while (k < f->max_k && p.y + step.y * (1 << k) > hf_highest(cx, f, x0, z0, k)) k++;
			k--;
			if (k > 0) {
				p = vadd(p, vmul(step, (1 << k)));
				continue;
			}
		}
		double lx = ray->dir.x > 0 ? (ceil(p.x) - p.x) * mx : (floor(p.x) - p.x) * mx;
		double lz = ray->dir.z > 0 ? (ceil(p.z) - p.z) * mz : (floor(p.z) - p.z) * mz;
		vec pn = vadd(p, vmul(step, (lx < lz ? lx : lz) + 1e-6));
		cx->cnt.hf_height++;
		if ((p.y < pn.y ? p.y : pn.y) < cx->sc->f32_pool[f->first_maxh + (int64_t)z0 * f->W + x0]) {
			double best = HXO_INF;
			vec A = V(x0, hf_height(cx, f, x0, z0), z0);
			vec B = V(x0 + 1, hf_height(cx, f, x0 + 1, z0), z0);
			vec Cv = V(x0 + 1, hf_height(cx, f, x0 + 1, z0 + 1), z0 + 1);
			vec D = V(x0, hf_height(cx, f, x0, z0 + 1), z0 + 1);
			cx->cnt.hf_tri_tests += 2;
			int b1 = tri_fast(ray, A, B, D, &best);
			int b2 = tri_fast(ray, B, Cv, D, &best);
			if (b1 || b2) {
				h->dist = best;
				h->ip = vadd(ray->start, vmul(ray->dir, best));
				h->norm = hf_normal(cx, f, (float)h->ip.x, (float)h->ip.z);
				h->u = h->ip.x / f->W;
				h->v = h->ip.z / f->H;
				h->dNdx = V(1, 0, 0);
				h->dNdy = V(0, 0, 1);
				h->geom = gi;
				return 1;
			}
		}
		p = pn;
	}
	return 0;
}

/* ------------------------------------------------------------------ CSG */
static int geom_hit(ctx_t* cx, int gi, const ray_t* ray, hit_t* h);

#define CSG_MAX 30
/* findAllIntersections, src/geometry.cpp:149-165 */
static int csg_collect(ctx_t* cx, int gi, ray_t ray, hit_t* out)
{
	vec origin = ray.start;
	int n = 0;
	for (int counter = 0; counter < CSG_MAX; counter++) {
		hit_t h;
		memset(&h, 0, sizeof h);
		h.dist = HXO_INF;
		if (!geom_hit(cx, gi, &ray, &h)) break;
		out[n++] = h;
		ray.start = vadd(h.ip, vmul(ray.dir, 1e-6));
	}
	for (int i = 0; i < n; i++) out[i].dist = vdist(origin, out[i].ip);
	return n;
}

/* CSGBase::intersect, src/geometry.cpp:167-194 */
static int csg_hit(ctx_t* cx, const hxr_geometry* g, int gi, const ray_t* ray, hit_t* h)
{
	hit_t all[2 * CSG_MAX];
	int nl = csg_collect(cx, g->left, *ray, all);
	int nr = csg_collect(cx, g->right, *ray, all + nl);
	/* stable insertion sort by dist: std::sort on distinct keys; ties are not ordered by the reference either */
	for (int i = 1; i < nl + nr; i++) {
		hit_t hv = all[i];
		int j = i - 1;
		while (j >= 0 && all[j].dist > hv.dist) { all[j + 1] = all[j]; j--; }
		all[j + 1] = hv;
	}
	int inA = nl % 2, inB = nr % 2;
#define CSG_INSIDE(a, b) (g->type == HXR_GEOM_CSG_UNION ? ((a) || (b)) : g->type == HXR_GEOM_CSG_INTER ? ((a) && (b)) : ((a) && !(b)))
	int initial = CSG_INSIDE(inA, inB);
	for (int i = 0; i < nl + nr; i++) {
		/* the reference tells the operands apart by info.geom == left */
		if (all[i].geom == g->left) inA = !inA; else inB = !inB;
		if (CSG_INSIDE(inA, inB) != initial) {
			*h = all[i];
			h->norm = vfaceforward(ray->dir, h->norm);
			h->geom = gi;
			return 1;
		}
	}
#undef CSG_INSIDE
	return 0;
}

static int geom_hit(ctx_t* cx, int gi, const ray_t* ray, hit_t* h)
{
	const hxr_geometry* g = cx->sc->geometries + gi;
	switch (g->type) {
		case HXR_GEOM_PLANE: return plane_hit(g, gi, ray, h);
		case HXR_GEOM_SPHERE: return sphere_hit(g, gi, ray, h);
		case HXR_GEOM_CUBE: return cube_hit(g, gi, ray, h);
		case HXR_GEOM_CSG_UNION:
		case HXR_GEOM_CSG_INTER:
		case HXR_GEOM_CSG_DIFF: return csg_hit(cx, g, gi, ray, h);
		case HXR_GEOM_MESH: return mesh_hit(cx, g, gi, ray, h);
		case HXR_GEOM_HEIGHTFIELD: return heightfield_hit(cx, g, gi, ray, h);
	}
	return 0;
}

/* Node::intersect, src/geometry.cpp:196-208 */
static int node_hit(ctx_t* cx, int ni, const ray_t* ray, hit_t* h)
{
	const hxr_node* n = cx->sc->nodes + ni;
	cx->cnt.node_tests++;
	ray_t local = *ray;
	local.start = xf_unpoint(&n->T, ray->start);
	local.dir = xf_undir(&n->T, ray->dir);
	if (!geom_hit(cx, n->geometry, &local, h)) return 0;
	h->ip = xf_point(&n->T, h->ip);
	h->norm = vunit(xf_normal(&n->T, h->norm));
	h->dist = vdist(ray->start, h->ip);
	return 1;
}

/* ------------------------------------------------------------------ lights */
/* RectLight::intersect, src/lights.cpp:50-75; PointLight::intersect returns 0 (src/lights.cpp:27-30) */
static int light_hit(const hxr_light* L, const ray_t* ray, double* dist)
{
	if (L->type != HXR_LIGHT_RECT) return 0;
	vec s = xf_unpoint(&L->T, ray->start), d = xf_undir(&L->T, ray->dir);
	if (fabs(d.y) < 1e-12) return 0;
	double t = -(s.y / d.y);
	if (t < 0) return 0;
	vec p = vadd(s, vmul(d, t));
	if (fabs(p.x) < 0.5 && fabs(p.z) < 0.5) {
		double w = vlen(vsub(xf_point(&L->T, p), ray->start));
		if (w < *dist) {
			*dist = w;
			return s.y < 0 ? +1 : -1;
		}
	}
	return 0;
}

/* Light::getNthSample, src/lights.h:88-92 (point), src/lights.cpp:37-51 (rect); (j1, j2) replace the two randDouble() */
static void light_sample(const hxr_light* L, int idx, vec shade_pos, double j1, double j2, vec* pos, col* color)
{
	if (L->type == HXR_LIGHT_POINT) {
		*pos = vld(L->pos);
		*color = cld(L->color);
		return;
	}
	double lx = ((idx % L->x_subd) + j1) / L->x_subd;
	double ly = ((idx / L->x_subd) + j2) / L->y_subd;
	*pos = xf_point(&L->T, V(lx - 0.5, -1e-6, ly - 0.5));
	vec ls = xf_unpoint(&L->T, shade_pos);
	if (ls.y < 0) *color = cdiv(cscale(cld(L->color), (float)(-ls.y)), (float)vlen(ls));
	else *color = C(0, 0, 0);
}

/* RectLight::getSolidAngle, src/lights.cpp:91-102; PointLight returns 0 */
static double light_solid_angle(const hxr_light* L, vec p)
{
	if (L->type != HXR_LIGHT_RECT) return 0;
	vec dir = xf_dir(&L->T, V(0, -1, 0));
	vec to_p = vsub(p, xf_point(&L->T, V(0, 0, 0)));
	double c = vdot(to_p, dir);
	if (c < 0) return 0;
	double d = vlen(to_p);
	c /= d;
	return L->area * c / ((1 + d) * (1 + d));
}

static inline int light_num_samples(const hxr_light* L) { return L->type == HXR_LIGHT_RECT ? L->x_subd * L->y_subd : 1; }
static inline col light_color(const hxr_light* L) { return cscale(cld(L->color), L->power); } /* Light::getColor */

/* ------------------------------------------------------------------ textures / environment */
/* Bitmap::getFilteredPixel, src/bitmap.cpp:66-81 */
static col bitmap_bilinear(const hxr_scene_desc* sc, int bi, float x, float y)
{
	if (bi < 0) return C(0, 0, 0);
	const hxr_bitmap* b = sc->bitmaps + bi;
	int w = b->width, hh = b->height;
	if (w <= 0 || hh <= 0 || x < 0 || x >= w || y < 0 || y >= hh) return C(0, 0, 0);
	int tx = (int)floorf(x), ty = (int)floorf(y);
	if (tx < 0 || ty < 0) return C(0, 0, 0);
	int tx1 = (tx + 1) % w, ty1 = (ty + 1) % hh;
	float p = x - tx, q = y - ty;
	const float* t = sc->f32_pool + b->first_texel;
	col c00 = cld(t + ((int64_t)ty * w + tx) * 3), c01 = cld(t + ((int64_t)ty * w + tx1) * 3);
	col c10 = cld(t + ((int64_t)ty1 * w + tx) * 3), c11 = cld(t + ((int64_t)ty1 * w + tx1) * 3);
	return cadd(cadd(cadd(cscale(c00, (1.0f - p) * (1.0f - q)), cscale(c01, p * (1.0f - q))), cscale(c10, (1.0f - p) * q)), cscale(c11, p * q));
}

/* fresnelSchlickApprox + Fresnel::sample, src/shading.cpp:326-343 (float arithmetic) */
static col fresnel_sample(const hxr_texture* t, vec dir, const hit_t* h)
{
	float eta = (float)t->ior;
	float ndi = (float)vdot(dir, h->norm);
	if (ndi > 0) eta = 1 / eta; else ndi = -ndi;
	double ratio = (double)((1.0f - eta) / (1.0f + eta));
	float f = (float)(ratio * ratio);
	float x = 1 - ndi;
	float fr = f + (1 - f) * powf(x, 5.0f);
	return C(fr, fr, fr);
}

/* Texture::sample dispatch: CheckerTexture src/shading.cpp:50-55, BitmapTexture :147-157, Fresnel :333-343 */
static col texture_sample(ctx_t* cx, int ti, vec dir, const hit_t* h)
{
	const hxr_texture* t = cx->sc->textures + ti;
	switch (t->type) {
		case HXR_TEX_CHECKER: {
			int u1 = (int)floor(h->u / t->scaling), v1 = (int)floor(h->v / t->scaling);
			return ((u1 + v1) % 2 == 0) ? cld(t->color1) : cld(t->color2);
		}
		case HXR_TEX_BITMAP: {
			const hxr_bitmap* b = cx->sc->bitmaps + t->bitmap;
			float u = (float)(h->u / t->scaling), v = (float)(h->v / t->scaling);
			u -= floorf(u);
			v -= floorf(v);
			u *= b->width;
			v *= b->height;
			cx->cnt.tex_bilinear++;
			return bitmap_bilinear(cx->sc, t->bitmap, u, v);
		}
		case HXR_TEX_FRESNEL: return fresnel_sample(t, dir, h);
		default: return C(0, 0, 0); /* BumpTexture / Bumps sample() return black, src/shading.h:191,220 */
	}
}

/* Texture::modifyNormal: BumpTexture src/shading.cpp:345-356, Bumps :363-377 */
static void texture_bump(ctx_t* cx, int ti, hit_t* h)
{
	const hxr_texture* t = cx->sc->textures + ti;
	if (t->type == HXR_TEX_BUMPTEX) {
		const hxr_bitmap* b = cx->sc->bitmaps + t->bitmap;
		float x = (float)fmod(h->u * t->scaling * b->width, b->width);
		float y = (float)fmod(h->v * t->scaling * b->height, b->height);
		cx->cnt.tex_bilinear++;
		col bump = bitmap_bilinear(cx->sc, t->bitmap, x, y);
		float dx = (float)(bump.r * t->strength), dy = (float)(bump.g * t->strength);
		h->norm = vunit(vadd(h->norm, vadd(vmul(h->dNdx, dx), vmul(h->dNdy, dy))));
	} else if (t->type == HXR_TEX_BUMPS) {
		float strength = (float)t->strength;
		if (strength > 0) {
			static const float fx[3] = {0.5f, 1.21f, 1.9f}, fz[3] = {0.4f, 1.13f, 1.81f}, amp[3] = {0.1f, 0.08f, 0.05f};
			const float fm = 0.2f;
			double dx = 0, dy = 0;
			for (int i = 0; i < 3; i++) {
				dx += sin(fm * fx[i] * h->u) * amp[i] * strength;
				dy += sin(fm * fz[i] * h->v) * amp[i] * strength;
			}
			h->norm = vunit(vadd(h->norm, vadd(vmul(h->dNdx, dx), vmul(h->dNdy, dy))));
		}
	}
}

/* CubemapEnvironment::getEnvironment + getSide, src/environment.cpp:54-81; Vector::maxDimension src/vector.h:86-98 */
static col environment_lookup(ctx_t* cx, vec dir)
{
	const hxr_environment* e = &cx->sc->environment;
	if (!e->loaded) return C(0, 0, 0);
	int dim = 0;
	double big = fabs(dir.x);
	if (fabs(dir.y) > big) { dim = 1; big = fabs(dir.y); }
	if (fabs(dir.z) > big) dim = 2;
	double lead = vget(dir, dim);
	vec s = vmul(dir, 1.0 / fabs(lead));
	int face = (lead > 0 ? 3 : 0) + dim;
	double fx, fy;
	switch (face) {
		case 0: fx = s.z; fy = -s.y; break;  /* NEGX */
		case 1: fx = s.x; fy = -s.z; break;  /* NEGY */
		case 2: fx = s.x; fy = s.y; break;   /* NEGZ */
		case 3: fx = -s.z; fy = -s.y; break; /* POSX */
		case 4: fx = s.x; fy = s.z; break;   /* POSY */
		default: fx = s.x; fy = -s.y; break; /* POSZ */
	}
	const hxr_bitmap* b = cx->sc->bitmaps + e->side_bitmap[face];
	cx->cnt.env_bilinear++;
	return bitmap_bilinear(cx->sc, e->side_bitmap[face], (float)((fx + 1) * 0.5 * (b->width - 1)), (float)((fy + 1) * 0.5 * (b->height - 1)));
}

/* ------------------------------------------------------------------ the integrators */
static col trace_ray(ctx_t* cx, const ray_t* ray);

/* visible(), src/main.cpp:171-190 */
static int visible(ctx_t* cx, vec A, vec B)
{
	cx->cnt.shadow_queries++;
	if (cx->probe_on) { /* unit-level probing: record the two points, answer with the bit the reference harness's visible() answers */
		int k = cx->probe_vis_count++;
		if (cx->probe_vis_points && k < cx->probe_vis_max) {
			double* q = cx->probe_vis_points + 6 * k;
			q[0] = A.x; q[1] = A.y; q[2] = A.z; q[3] = B.x; q[4] = B.y; q[5] = B.z;
		}
		return (int)((cx->probe_vis_pattern >> (k & 63)) & 1u);
	}
	double D = vdist(A, B);
	ray_t ray;
	memset(&ray, 0, sizeof ray);
	ray.start = A;
	ray.dir = vunit(vsub(B, A));
	for (int64_t i = 0; i < cx->sc->num_nodes; i++) {
		hit_t h;
		if (node_hit(cx, (int)i, &ray, &h) && h.dist < D) { cx->cnt.reserved_[0]++; return 0; }   /* reserved_[0]: shadow queries that found an occluder */
	}
	for (int64_t i = 0; i < cx->sc->num_lights; i++) {
		double d = D;
		if (light_hit(cx->sc->lights + i, &ray, &d)) { cx->cnt.reserved_[0]++; return 0; }
	}
	return 1;
}

/* TraceContext::raycast, src/main.cpp:63-103.  Returns 1 with *early set when the ray ends without shading. */
static int raycast(ctx_t* cx, const ray_t* ray, hit_t* best, int* best_node, col* early)
{
	const hxr_scene_desc* sc = cx->sc;
	cx->cnt.raycasts++;
	if (ray->depth > sc->settings.max_trace_depth) { *early = C(0, 0, 0); return 1; }
	best->dist = HXO_INF;
	*best_node = -1;
	for (int64_t i = 0; i < sc->num_nodes; i++) {
		/* The reference's `IntersectionInfo info;` is uninitialised here (Vector() {} writes nothing), and Sphere / Cube / CSG never set
		   dNdx / dNdy: a bump map on such a node reads stack garbage there.  This restatement (and the kernels) define them as zero. */
		hit_t h;
		memset(&h, 0, sizeof h);
		if (node_hit(cx, (int)i, ray, &h) && h.dist < best->dist) {
			*best = h;
			*best_node = (int)i;
		}
	}
	int hit_light = 0;
	col light_col = C(0, 0, 0);
	for (int64_t i = 0; i < sc->num_lights; i++) {
		const hxr_light* L = sc->lights + i;
		int s = light_hit(L, ray, &best->dist);
		if (s == -1) { hit_light = 1; light_col = C(0, 0, 0); }
		else if (s == 1) { hit_light = 1; light_col = cscale(light_color(L), L->scale_factor); }
	}
	if (hit_light) { *early = (ray->flags & RF_GI_DIFFUSE) ? C(0, 0, 0) : light_col; return 1; }
	if (best->dist >= HXO_INF) {
		*early = sc->environment.present ? environment_lookup(cx, ray->dir) : cld(sc->settings.background);
		return 1;
	}
	int bump = sc->nodes[*best_node].bump;
	if (bump >= 0) texture_bump(cx, bump, best);
	return 0;
}

static inline void draw2(ctx_t* cx, uint32_t key, uint32_t dim, double* u, double* v)
{
	if (cx->probe_draws) {
		*u = cx->probe_draws_used < cx->probe_draws_len ? cx->probe_draws[cx->probe_draws_used] : 0.5;
		*v = cx->probe_draws_used + 1 < cx->probe_draws_len ? cx->probe_draws[cx->probe_draws_used + 1] : 0.5;
		cx->probe_draws_used += 2;
		return;
	}
	hxo_uniform2(hxo_philox(cx->seed, cx->pixel, cx->sample, key, dim), u, v);
}

/* getLambertTerm, src/shading.cpp:57-65 */
static float lambert_term(const hit_t* h, double* dist2, vec light_pos)
{
	vec to_ip = vsub(h->ip, light_pos);
	*dist2 = vlen2(to_ip);
	vec to_light = vunit(vneg(to_ip));
	double d = vdot(h->norm, to_light);
	return (float)(d > 0.0 ? d : 0.0);
}

/* Lambert::computeColor src/shading.cpp:67-91 and Phong::computeColor :112-145 */
static col shade_diffuse(ctx_t* cx, const hxr_shader* s, const ray_t* ray, const hit_t* h)
{
	const hxr_scene_desc* sc = cx->sc;
	col base = s->texture >= 0 ? texture_sample(cx, s->texture, ray->dir, h) : cld(s->color);
	col direct = C(0, 0, 0);
	for (int64_t li = 0; li < sc->num_lights; li++) {
		const hxr_light* L = sc->lights + li;
		int n = light_num_samples(L);
		col sum = C(0, 0, 0);
		for (int i = 0; i < n; i++) {
			double j1 = 0, j2 = 0;
			if (L->type == HXR_LIGHT_RECT) draw2(cx, ray->key, 0x10000u + (uint32_t)li * 4096u + (uint32_t)i, &j1, &j2);
			vec lp;
			col lc;
			light_sample(L, i, h->ip, j1, j2, &lp, &lc);
			if (czero(lc) || !visible(cx, lp, vadd(h->ip, vmul(h->norm, 1e-6)))) continue;
			double d2;
			float lt = lambert_term(h, &d2, lp);
			if (s->type == HXR_SHADER_PHONG) {
				vec from_light = vunit(vsub(h->ip, lp));
				vec refl = vreflect(from_light, vfaceforward(from_light, h->norm));
				float cg = (float)vdot(vneg(ray->dir), refl);
				if (cg > 0) sum = cadd(sum, cscale(cmul(cld(s->specular), lc), powf(cg, s->exponent)));
			}
			sum = cadd(sum, cscale(cmul(base, lc), (float)(lt * L->power / d2)));
		}
		direct = cadd(direct, cdiv(sum, (float)n));
	}
	return cadd(direct, cmul(base, cld(sc->settings.ambient)));
}

/* refract(), src/shading.cpp:208-214 (float arithmetic); returns 0 on total internal reflection */
static int refract_dir(vec i, vec n, float ior, vec* out)
{
	float ndi = (float)vdot(i, n);
	float k = 1 - (ior * ior) * (1 - ndi * ndi);
	if (k < 0) return 0;
	*out = vsub(vmul(i, ior), vmul(n, ior * ndi + sqrtf(k)));
	return 1;
}

static col shade(ctx_t* cx, int si, const ray_t* ray, const hit_t* h);

/* one turn of unitDiskSample's loop (src/util.cpp:101-107): two uniform draws -> a point of [-1,1]^2; returns 1 when it is accepted */
static int disk_map(double u, double v, double* x, double* y)
{
	*x = u * 2 - 1;
	*y = v * 2 - 1;
	return !(*x * *x + *y * *y > 1);
}

/* one turn of the glossy loop of Reflection::computeColor (src/shading.cpp:165-179) for an accepted disk point (x, y):
   perturb the face-forwarded normal `n` in its othonormedBasis (src/vector.h:194-207) and reflect `dir` about it */
static vec glossy_dir(vec n, vec dir, float glossiness, double x, double y)
{
	float scaling = powf(10.0f, 2 - 8 * glossiness);
	vec t0 = V(0.577350269, 0.577350269, 0.577350269), t1 = V(0.267261242, -0.534522484, 0.801783726);
	vec tv = fabs(vdot(n, t0)) > 0.9 ? t1 : t0;
	vec bu = vunit(vcross(tv, n));
	vec bv = vcross(bu, n);
	x *= scaling;
	y *= scaling;
	vec mn = vunit(vadd(n, vadd(vmul(bu, x), vmul(bv, y))));
	return vreflect(dir, mn);
}

/* Reflection::computeColor, src/shading.cpp:159-187 */
static col shade_reflection(ctx_t* cx, int si, const hxr_shader* s, const ray_t* ray, const hit_t* h)
{
	vec n = vfaceforward(ray->dir, h->norm);
	ray_t nr = *ray;
	nr.start = vadd(h->ip, vmul(n, 1e-6));
	nr.depth = ray->depth + 1;
	if (s->glossiness < 1.0f) {
		col sum = C(0, 0, 0);
		for (int i = 0; i < s->num_samples; i++) {
			vec refl = n;
			for (uint32_t attempt = 0;; attempt++) {
				/* unitDiskSample inside the rejection loop: each outer attempt draws one accepted disk point.  Both loops are
				   unbounded in the reference; here they stop after 16 turns (a disk draw fails 16 times in a row with p = 2e-11; the outer
				   test fails that often only at grazing incidence), and the HIP kernel does the same */
				double x = 0, y = 0, u, v;
				uint32_t inner = 0;
				int accepted;
				do {
					hxo_uniform2(hxo_philox(cx->seed, cx->pixel, cx->sample, ray->key, 0x2000000u + (uint32_t)i * 256u + ((attempt * 16u + inner) & 255u)), &u, &v);
					accepted = disk_map(u, v, &x, &y);
					inner++;
				} while (!accepted && inner < 16);
				refl = glossy_dir(n, ray->dir, s->glossiness, x, y);
				if (!(vdot(refl, n) < 0) || attempt >= 15) break;
			}
			nr.dir = refl;
			nr.key = hxo_child_key(ray->key, si, (uint32_t)i);
			sum = cadd(sum, cmul(trace_ray(cx, &nr), cld(s->color)));
		}
		return cdiv(sum, (float)s->num_samples);
	}
	nr.dir = vreflect(ray->dir, n);
	nr.key = hxo_child_key(ray->key, si, 0);
	return cmul(trace_ray(cx, &nr), cld(s->color));
}

/* Refraction::computeColor, src/shading.cpp:216-232 */
static col shade_refraction(ctx_t* cx, int si, const hxr_shader* s, const ray_t* ray, const hit_t* h)
{
	vec out;
	int ok;
	if (vdot(ray->dir, h->norm) < 0) ok = refract_dir(ray->dir, h->norm, (float)(1 / s->ior), &out);
	else ok = refract_dir(ray->dir, vneg(h->norm), (float)s->ior, &out);
	if (!ok) return C(0, 0, 0);
	ray_t nr = *ray;
	nr.start = vsub(h->ip, vmul(vfaceforward(ray->dir, h->norm), 0.000001));
	nr.dir = out;
	nr.depth++;
	nr.key = hxo_child_key(ray->key, si, 0);
	return cmul(trace_ray(cx, &nr), cld(s->color));
}

/* Shader::computeColor dispatch; Layered::computeColor src/shading.cpp:315-324; Const src/shading.h:206-214 */
static col shade(ctx_t* cx, int si, const ray_t* ray, const hit_t* h)
{
	const hxr_shader* s = cx->sc->shaders + si;
	switch (s->type) {
		case HXR_SHADER_LAMBERT:
		case HXR_SHADER_PHONG: return shade_diffuse(cx, s, ray, h);
		case HXR_SHADER_REFLECTION: return shade_reflection(cx, si, s, ray, h);
		case HXR_SHADER_REFRACTION: return shade_refraction(cx, si, s, ray, h);
		case HXR_SHADER_CONST: return cld(s->color);
		case HXR_SHADER_LAYERED: {
			col acc = C(0, 0, 0);
			for (int i = 0; i < s->num_layers; i++) {
				const hxr_layer* L = cx->sc->layers + s->first_layer + i;
				col blend = L->texture >= 0 ? texture_sample(cx, L->texture, ray->dir, h) : cld(L->blend);
				col from = shade(cx, L->shader, ray, h);
				acc = cadd(cmul(acc, csub(C(1, 1, 1), blend)), cmul(from, blend));
			}
			return acc;
		}
	}
	return C(0, 0, 0);
}

/* raytrace(), src/main.cpp:105-113 */
static col trace_ray(ctx_t* cx, const ray_t* ray)
{
	if (cx->probe_on) { /* unit-level probing: record the ray, answer with the colour the reference harness's raytrace() answers */
		if (cx->probe_count < cx->probe_max) {
			double* r = cx->probe_rays + 8 * cx->probe_count;
			r[0] = ray->start.x; r[1] = ray->start.y; r[2] = ray->start.z;
			r[3] = ray->dir.x; r[4] = ray->dir.y; r[5] = ray->dir.z;
			r[6] = ray->depth; r[7] = ray->flags;
		}
		cx->probe_count++;
		return C((float)(0.2 + 0.6 * fabs(ray->dir.x)), (float)(0.2 + 0.6 * fabs(ray->dir.y)), (float)(0.2 + 0.6 * fabs(ray->dir.z)));
	}
	hit_t h;
	int node;
	col early;
	if (raycast(cx, ray, &h, &node, &early)) return early;
	return shade(cx, cx->sc->nodes[node].shader, ray, &h);
}

/* BRDF::eval: Lambert src/shading.cpp:93-97, Reflection :189-194, Refraction :234-239, default :32-35 */
static col brdf_eval(ctx_t* cx, const hxr_shader* s, const hit_t* h, vec w_in, vec w_out)
{
	switch (s->type) {
		case HXR_SHADER_LAMBERT:
		case HXR_SHADER_PHONG: {
			col base = s->texture >= 0 ? texture_sample(cx, s->texture, w_in, h) : cld(s->color);
			return cscale(cscale(base, (float)(1 / HXO_PI)), (float)fabs(vdot(w_out, h->norm)));
		}
		case HXR_SHADER_REFLECTION:
			return vdot(vreflect(w_in, vfaceforward(w_in, h->norm)), w_out) > 0.999999999 ? cld(s->color) : C(0, 0, 0);
		case HXR_SHADER_REFRACTION:
			return vdot(vreflect(w_in, vfaceforward(w_in, h->norm)), w_out) > 0.99999999 ? cld(s->color) : C(0, 0, 0);
		default: return C(1, 0, 0);
	}
}

/* hemisphereSample, src/util.cpp:109-129 */
static vec hemisphere(double u, double v, vec normal)
{
	double theta = 2 * HXO_PI * u;
	double cphi = 2 * v - 1;
	double sphi = sqrt(1 - cphi * cphi);
	vec d = V(cos(theta) * sphi, cphi, sin(theta) * sphi);
	if (vdot(d, normal) < 0) d = vneg(d);
	return d;
}

/* BRDF::spawnRay: Lambert src/shading.cpp:99-109, Reflection :196-205, Refraction :241-263, default :37-43 */
static void brdf_spawn(ctx_t* cx, int si, const hit_t* h, const ray_t* in, ray_t* out, col* color, float* pdf)
{
	const hxr_shader* s = cx->sc->shaders + si;
	vec w_in = in->dir;
	out->key = hxo_child_key(in->key, si, 0);
	switch (s->type) {
		case HXR_SHADER_LAMBERT:
		case HXR_SHADER_PHONG: {
			vec N = vfaceforward(w_in, h->norm);
			out->start = vadd(h->ip, vmul(N, 1e-6));
			double u, v;
			draw2(cx, in->key, 2, &u, &v);
			out->dir = hemisphere(u, v, N);
			out->flags |= RF_GI_DIFFUSE;
			col base = s->texture >= 0 ? texture_sample(cx, s->texture, w_in, h) : cld(s->color);
			*color = cscale(cscale(base, (float)(1 / HXO_PI)), (float)vdot(out->dir, N));
			*pdf = (float)(1 / (2 * HXO_PI));
			return;
		}
		case HXR_SHADER_REFLECTION: {
			vec N = vfaceforward(w_in, h->norm);
			out->start = vadd(h->ip, vmul(N, 1e-6));
			out->dir = vreflect(w_in, N);
			out->flags &= ~RF_GI_DIFFUSE;
			*color = cscale(cld(s->color), 1e+6f);
			*pdf = 1e+6f;
			return;
		}
		case HXR_SHADER_REFRACTION: {
			vec r;
			int ok;
			if (vdot(w_in, h->norm) < 0) ok = refract_dir(w_in, h->norm, (float)(1 / s->ior), &r);
			else ok = refract_dir(w_in, vneg(h->norm), (float)s->ior, &r);
			if (!ok) { *color = C(1, 0, 0); *pdf = -1; return; }
			out->start = vsub(h->ip, vmul(vfaceforward(w_in, h->norm), 0.000001));
			out->dir = r;
			out->flags &= ~RF_GI_DIFFUSE;
			*color = cscale(cld(s->color), 1e+6f);
			*pdf = 1e+6f;
			return;
		}
		default:
			out->dir = V(1, 0, 0);
			*color = C(1, 0, 0);
			*pdf = -1;
	}
}

/* pathtrace(), src/main.cpp:115-169 (recursive, like the reference) */
static col trace_path(ctx_t* cx, const ray_t* ray, col mult)
{
	const hxr_scene_desc* sc = cx->sc;
	if (cintensity(mult) < 0.001f) return C(0, 0, 0);
	hit_t h;
	int node;
	col early;
	if (raycast(cx, ray, &h, &node, &early)) return cmul(early, mult);
	int si = sc->nodes[node].shader;
	ray_t next = *ray;
	col bc;
	float pdf;
	next.depth++;
	brdf_spawn(cx, si, &h, ray, &next, &bc, &pdf);
	if (pdf <= 0) return C(0, 0, 0);
	col from_gi = trace_path(cx, &next, cdiv(cmul(mult, bc), pdf));
	if (sc->num_lights == 0) return from_gi;
	hxo_rand128 pick = hxo_philox(cx->seed, cx->pixel, cx->sample, ray->key, 3);
	const hxr_light* L = sc->lights + hxo_below(pick.w[0], (uint32_t)sc->num_lights);
	double solid = light_solid_angle(L, h.ip);
	if (solid <= 0) return from_gi;
	int idx = (int)hxo_below(pick.w[2], (uint32_t)light_num_samples(L));
	double j1, j2;
	hxo_uniform2(hxo_philox(cx->seed, cx->pixel, cx->sample, ray->key, 4), &j1, &j2);
	vec on_light;
	col unused;
	light_sample(L, idx, h.ip, j1, j2, &on_light, &unused);
	if (!visible(cx, vadd(h.ip, vmul(h.norm, 1e-6)), on_light)) return from_gi;
	vec w_out = vunit(vsub(on_light, h.ip));
	col from_light = cmul(cscale(light_color(L), L->scale_factor), brdf_eval(cx, sc->shaders + si, &h, ray->dir, w_out));
	if (cintensity(from_light) <= 0) return from_gi;
	float p_choose = 1.0f / (float)sc->num_lights;
	float p_hit = (float)(1.0f / solid);
	float p_path = p_choose * p_hit;
	return cadd(from_gi, cdiv(cmul(from_light, mult), p_path));
}

/* ------------------------------------------------------------------ camera + per-sample entry */
/* Camera::getScreenRay src/camera.cpp:65-76, getDOFScreenRay :78-86 */
static ray_t camera_ray(const ctx_t* cx, double x, double y, int dof, double lu, double lv, double stereo)
{
	const hxr_camera* c = &cx->sc->camera;
	ray_t r;
	memset(&r, 0, sizeof r);
	vec pos = vld(c->pos), tl = vld(c->top_left);
	r.start = pos;
	vec through = vadd(vadd(tl, vmul(vsub(vld(c->top_right), tl), x / (double)cx->W)), vmul(vsub(vld(c->bottom_left), tl), y / (double)cx->H));
	r.dir = vunit(vsub(through, r.start));
	if (stereo != 0) r.start = vadd(r.start, vmul(vld(c->right), stereo * c->stereo_separation));
	if (dof) {
		double M = cx->focal_plane_dist / vdot(vld(c->front), r.dir);
		vec T = vadd(r.start, vmul(r.dir, M));
		r.start = vadd(vadd(pos, vmul(vld(c->right), lu * c->aperture_size)), vmul(vld(c->up), lv * c->aperture_size));
		r.dir = vunit_lazy(vsub(T, r.start));
	}
	return r;
}

/* traceSingleRay, src/main.cpp:227-249 */
static col trace_sample(ctx_t* cx, double x, double y)
{
	const hxr_scene_desc* sc = cx->sc;
	double lu = 0, lv = 0;
	cx->cnt.camera_samples++;
	if (sc->camera.dof) { /* unitDiskSample, src/util.cpp:101-107 */
		for (uint32_t attempt = 0; attempt < 64; attempt++) {
			double u, v;
			hxo_uniform2(hxo_philox(cx->seed, cx->pixel, cx->sample, 0, 16 + attempt), &u, &v);
			if (disk_map(u, v, &lu, &lv)) break;
		}
	}
	int passes = sc->camera.stereo_separation == 0.0 ? 1 : 2;
	col eye[2];
	for (int e = 0; e < passes; e++) {
		double off = passes == 1 ? 0 : (e == 0 ? -0.5 : +0.5);
		ray_t r = camera_ray(cx, x, y, sc->camera.dof, lu, lv, off);
		eye[e] = sc->settings.gi ? trace_path(cx, &r, C(1, 1, 1)) : trace_ray(cx, &r);
	}
	if (passes == 1) return eye[0];
	float ml = cintensity(eye[0]), mr = cintensity(eye[1]);
	return C(ml + (eye[0].r - ml) * 0.3f, mr + (eye[1].g - mr) * 0.3f, mr + (eye[1].b - mr) * 0.3f);
}

/* ------------------------------------------------------------------ frame drivers */
static void counters_add(hxr_counters* a, const hxr_counters* b)
{
	uint64_t* pa = (uint64_t*)a;
	const uint64_t* pb = (const uint64_t*)b;
	for (size_t i = 0; i < sizeof(hxr_counters) / sizeof(uint64_t); i++) pa[i] += pb[i];
}

static inline float* px(float* out, size_t stride, int x, int y) { return (float*)((char*)out + (size_t)y * stride) + (size_t)x * 3; }

/*
 * hxo_render: the oracle's render() (src/main.cpp:416-426).
 *   mode / spp / seed / width / height as in hxr_frame_params (tile sharding is ignored: whole frame).
 *   threads <= 0: all OpenMP threads.  Buckets are 64x64 and handed out dynamically like the
 *   reference's atomic cursor (src/main.cpp:273-275); results do not depend on the schedule.
 * Returns 0, or -1 on bad arguments.
 */
int hxo_render(const hxr_scene_desc* sc, const hxr_frame_params* fp, float* out, size_t row_stride_bytes, int threads,
               hxr_counters* counters)
{
	if (!sc || !fp || !out || sc->abi_version != HXR_ABI_VERSION) return -1;
	for (int64_t i = 0; i < sc->num_heightfields; i++)
		if (sc->heightfields[i].raw_heights || sc->heightfields[i].first_maxh < 0) return -1;   /* a descriptor whose tables were left to the GPU module */
	for (int64_t i = 0; i < sc->num_bitmaps; i++)
		if (sc->bitmaps[i].prep) return -1;                                                     /* likewise: gamma decode / differentiate left to it */
	int W = fp->width > 0 ? fp->width : sc->settings.frame_width;
	int H = fp->height > 0 ? fp->height : sc->settings.frame_height;
	if (W <= 0 || H <= 0 || W > 1920 || H > 1920) return -1;
	int mode = fp->mode, spp = fp->spp;
	if (mode == HXR_MODE_AUTO) {
		int rpp = 0;
		if (sc->camera.dof) rpp = sc->camera.num_samples;
		if (sc->settings.gi) rpp = rpp > sc->settings.num_paths ? rpp : sc->settings.num_paths;
		if (rpp > 0) { mode = HXR_MODE_MONTE_CARLO; if (spp <= 0) spp = rpp; }
		else mode = sc->settings.want_aa ? HXR_MODE_ADAPTIVE_AA : HXR_MODE_SINGLE;
	}
	if (mode == HXR_MODE_MONTE_CARLO && spp <= 0) {
		if (sc->camera.dof) spp = sc->camera.num_samples;
		if (sc->settings.gi) spp = spp > sc->settings.num_paths ? spp : sc->settings.num_paths;
		if (spp <= 0) spp = 1;
	}
	uint64_t seed = fp->rng_seed;
	double focal = sc->camera.focal_plane_dist;
	if (mode == HXR_MODE_MONTE_CARLO && sc->camera.dof && sc->camera.auto_focus) {
		/* autofocus, src/main.cpp:350-362: distance of the closest node along the ray through the frame centre */
		ctx_t cx;
		memset(&cx, 0, sizeof cx);
		cx.sc = sc; cx.W = W; cx.H = H; cx.focal_plane_dist = focal;
		ray_t mid = camera_ray(&cx, W * 0.5, H * 0.5, 0, 0, 0, 0);
		double closest = HXO_INF;
		for (int64_t i = 0; i < sc->num_nodes; i++) {
			hit_t h;
			if (node_hit(&cx, (int)i, &mid, &h) && h.dist < closest) closest = h.dist;
		}
		if (closest < HXO_INF) focal = closest;
	}
	const int B = 64;
	int BW = (W - 1) / B + 1, BH = (H - 1) / B + 1, nb = BW * BH;
	hxr_counters total;
	memset(&total, 0, sizeof total);
#ifdef _OPENMP
	if (threads > 0) omp_set_num_threads(threads);
#endif
	unsigned char* flags = NULL;

	if (mode == HXR_MODE_PREPASS) { /* coarseRender, src/main.cpp:387-414 (the reference paints the blocks on the window; here they fill the frame) */
		if (spp <= 0) return -1;
		float mul = 1.0f / spp;
		ctx_t cx;
		memset(&cx, 0, sizeof cx);
		cx.sc = sc; cx.W = W; cx.H = H; cx.seed = seed; cx.focal_plane_dist = focal;
		for (int y = 0; y < H; y += 24)
			for (int x = 0; x < W; x += 24) {
				col sum = C(0, 0, 0);
				cx.pixel = (uint32_t)(y * W + x);
				for (int i = 0; i < spp; i++) {
					double jx, jy;
					cx.sample = (uint32_t)i;
					hxo_uniform2(hxo_philox(seed, cx.pixel, cx.sample, 0, 5), &jx, &jy);
					sum = cadd(sum, trace_sample(&cx, x + jx * 24, y + jy * 24));
				}
				int x1 = x + 24 < W ? x + 24 : W, y1 = y + 24 < H ? y + 24 : H;
				for (int py = y; py < y1; py++)
					for (int qx = x; qx < x1; qx++) {
						float* o = px(out, row_stride_bytes, qx, py);
						o[0] = sum.r * mul; o[1] = sum.g * mul; o[2] = sum.b * mul;
					}
			}
		counters_add(&total, &cx.cnt);
	} else if (mode == HXR_MODE_MONTE_CARLO) { /* renderWithMonteCarlo, src/main.cpp:347-384 */
		float mul = 1.0f / spp;
#pragma omp parallel
		{
			ctx_t cx;
			memset(&cx, 0, sizeof cx);
			cx.sc = sc; cx.W = W; cx.H = H; cx.seed = seed; cx.focal_plane_dist = focal;
#pragma omp for schedule(dynamic, 1)
			for (int b = 0; b < nb; b++) {
				int x0 = (b % BW) * B, y0 = (b / BW) * B;
				int x1 = x0 + B < W ? x0 + B : W, y1 = y0 + B < H ? y0 + B : H;
				for (int y = y0; y < y1; y++)
					for (int x = x0; x < x1; x++) {
						col sum = C(0, 0, 0);
						cx.pixel = (uint32_t)(y * W + x);
						for (int s = 0; s < spp; s++) {
							double jx, jy;
							cx.sample = (uint32_t)s;
							hxo_uniform2(hxo_philox(seed, cx.pixel, cx.sample, 0, 0), &jx, &jy);
							sum = cadd(sum, trace_sample(&cx, x + jx, y + jy));
						}
						float* o = px(out, row_stride_bytes, x, y);
						o[0] = sum.r * mul; o[1] = sum.g * mul; o[2] = sum.b * mul;
					}
			}
#pragma omp critical
			counters_add(&total, &cx.cnt);
		}
	} else { /* renderWithoutMonteCarlo, src/main.cpp:253-345 */
		const int fovea = fp->foveated_radius, vip_x = fp->vip_x, vip_y = fp->vip_y;
		const int fovea_thresh = fovea * fovea;
		static const int coarse[5][2] = {{2, 2}, {2, 6}, {4, 4}, {6, 2}, {6, 6}}; /* COARSE_KERNEL, src/main.cpp:262-268: {dy, dx} */
#pragma omp parallel
		{
			ctx_t cx;
			memset(&cx, 0, sizeof cx);
			cx.sc = sc; cx.W = W; cx.H = H; cx.seed = seed; cx.focal_plane_dist = focal;
#pragma omp for schedule(dynamic, 1)
			for (int b = 0; b < nb; b++) {
				int x0 = (b % BW) * B, y0 = (b / BW) * B;
				int x1 = x0 + B < W ? x0 + B : W, y1 = y0 + B < H ? y0 + B : H;
				if (fovea <= 0) { /* src/main.cpp:277-283 */
					for (int y = y0; y < y1; y++)
						for (int x = x0; x < x1; x++) {
							cx.pixel = (uint32_t)(y * W + x);
							cx.sample = 0;
							col c = trace_sample(&cx, x, y);
							float* o = px(out, row_stride_bytes, x, y);
							o[0] = c.r; o[1] = c.g; o[2] = c.b;
						}
					continue;
				}
				/* foveated rendering, src/main.cpp:284-311: 8x8 blocks far from (vipX, vipY) get the mean of five rays */
				for (int by = y0; by < y1; by += 8) {
					int bye = by + 8 < y1 ? by + 8 : y1;
					for (int bx = x0; bx < x1; bx += 8) {
						int bxe = bx + 8 < x1 ? bx + 8 : x1;
						int cxp = bx + 4, cyp = by + 4;
						if ((cxp - vip_x) * (cxp - vip_x) + (cyp - vip_y) * (cyp - vip_y) > fovea_thresh) {
							col sum = C(0, 0, 0);
							for (int j = 0; j < 5; j++) {
								int sx = bx + coarse[j][1], sy = by + coarse[j][0];
								cx.pixel = (uint32_t)(sy * W + sx);
								cx.sample = 0;
								sum = cadd(sum, trace_sample(&cx, sx, sy));
							}
							sum = cscale(sum, 0.2f);
							for (int y = by; y < bye; y++)
								for (int x = bx; x < bxe; x++) {
									float* o = px(out, row_stride_bytes, x, y);
									o[0] = sum.r; o[1] = sum.g; o[2] = sum.b;
								}
						} else {
							for (int y = by; y < bye; y++)
								for (int x = bx; x < bxe; x++) {
									cx.pixel = (uint32_t)(y * W + x);
									cx.sample = 0;
									col c = trace_sample(&cx, x, y);
									float* o = px(out, row_stride_bytes, x, y);
									o[0] = c.r; o[1] = c.g; o[2] = c.b;
								}
						}
					}
				}
			}
#pragma omp critical
			counters_add(&total, &cx.cnt);
		}
		if (mode == HXR_MODE_ADAPTIVE_AA) {
			/* detectAApixels, src/main.cpp:192-222 */
			flags = (unsigned char*)calloc((size_t)W * H, 1);
			if (!flags) return -1;
			static const int nb8[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
			uint64_t flagged = 0;
			for (int y = 0; y < H; y++)
				for (int x = 0; x < W; x++) {
					const float* me = px(out, row_stride_bytes, x, y);
					int need = 0;
					/* src/main.cpp:201,205: outside the fovea nothing is anti-aliased */
					if ((x - vip_x) * (x - vip_x) + (y - vip_y) * (y - vip_y) > (fovea > 0 ? fovea_thresh : 999666111)) { flags[(size_t)y * W + x] = 0; continue; }
					for (int k = 0; k < 8 && !need; k++) {
						int nx = x + nb8[k][0], ny = y + nb8[k][1];
						if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
						const float* nbp = px(out, row_stride_bytes, nx, ny);
						for (int c = 0; c < 3; c++) {
							float a = me[c] < 1.0f ? me[c] : 1.0f, bb = nbp[c] < 1.0f ? nbp[c] : 1.0f;
							if (fabsf(a - bb) > 0.075f) { need = 1; break; }
						}
					}
					flags[(size_t)y * W + x] = (unsigned char)need;
					flagged += (uint64_t)need;
				}
			total.aa_pixels = flagged;
			/* pass 3, src/main.cpp:324-342: four more taps, then * 1/5 */
			static const float taps[5][2] = {{0.0f, 0.0f}, {0.6f, 0.0f}, {0.3f, 0.3f}, {0.0f, 0.6f}, {0.6f, 0.6f}};
#pragma omp parallel
			{
				ctx_t cx;
				memset(&cx, 0, sizeof cx);
				cx.sc = sc; cx.W = W; cx.H = H; cx.seed = seed; cx.focal_plane_dist = focal;
				float mul = 1.0f / 5;
#pragma omp for schedule(dynamic, 1)
				for (int b = 0; b < nb; b++) {
					int x0 = (b % BW) * B, y0 = (b / BW) * B;
					int x1 = x0 + B < W ? x0 + B : W, y1 = y0 + B < H ? y0 + B : H;
					for (int y = y0; y < y1; y++)
						for (int x = x0; x < x1; x++) {
							if (!flags[(size_t)y * W + x]) continue;
							float* o = px(out, row_stride_bytes, x, y);
							cx.pixel = (uint32_t)(y * W + x);
							for (int i = 1; i < 5; i++) {
								cx.sample = (uint32_t)i;
								/* int + float arithmetic happens in float, like `x + AA_KERNEL[i][0]` */
								col c = trace_sample(&cx, (double)((float)x + taps[i][0]), (double)((float)y + taps[i][1]));
								o[0] += c.r; o[1] += c.g; o[2] += c.b;
							}
							o[0] *= mul; o[1] *= mul; o[2] *= mul;
						}
				}
#pragma omp critical
				counters_add(&total, &cx.cnt);
			}
			free(flags);
		}
	}
	if (counters) *counters = total;
	return 0;
}

/* ------------------------------------------------------------------ unit-level entry points (golden-vector tests) */

/* one object-space geometry query; returns hit flag, fills out[12] = dist, ip, norm, u, v, (pad) */
int hxo_geometry_intersect(const hxr_scene_desc* sc, int geometry, const double* start, const double* dir, double* out)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	ray_t r;
	memset(&r, 0, sizeof r);
	r.start = vld(start);
	r.dir = vld(dir);
	hit_t h;
	memset(&h, 0, sizeof h);
	int ok = geom_hit(&cx, geometry, &r, &h);
	out[0] = h.dist; out[1] = h.ip.x; out[2] = h.ip.y; out[3] = h.ip.z;
	out[4] = h.norm.x; out[5] = h.norm.y; out[6] = h.norm.z; out[7] = h.u; out[8] = h.v;
	return ok;
}

int hxo_node_intersect(const hxr_scene_desc* sc, int node, const double* start, const double* dir, double* out)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	ray_t r;
	memset(&r, 0, sizeof r);
	r.start = vld(start);
	r.dir = vld(dir);
	hit_t h;
	memset(&h, 0, sizeof h);
	int ok = node_hit(&cx, node, &r, &h);
	out[0] = h.dist; out[1] = h.ip.x; out[2] = h.ip.y; out[3] = h.ip.z;
	out[4] = h.norm.x; out[5] = h.norm.y; out[6] = h.norm.z; out[7] = h.u; out[8] = h.v;
	return ok;
}

int hxo_box_test(const double* lo, const double* hi, const double* start, const double* dir, double* nearest)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	box_t b = {vld(lo), vld(hi)};
	ray_t r;
	memset(&r, 0, sizeof r);
	r.start = vld(start);
	r.dir = vld(dir);
	if (nearest) *nearest = box_nearest(&b, &r);
	return box_hit(&cx, &b, &r);
}

int hxo_light_intersect(const hxr_scene_desc* sc, int light, const double* start, const double* dir, double* dist_inout)
{
	ray_t r;
	memset(&r, 0, sizeof r);
	r.start = vld(start);
	r.dir = vld(dir);
	return light_hit(sc->lights + light, &r, dist_inout);
}

double hxo_light_solid_angle(const hxr_scene_desc* sc, int light, const double* p) { return light_solid_angle(sc->lights + light, vld(p)); }

void hxo_light_sample(const hxr_scene_desc* sc, int light, int idx, const double* shade_pos, double j1, double j2, double* pos, float* color)
{
	vec p;
	col c;
	light_sample(sc->lights + light, idx, vld(shade_pos), j1, j2, &p, &c);
	pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
	color[0] = c.r; color[1] = c.g; color[2] = c.b;
}

void hxo_camera_ray(const hxr_scene_desc* sc, int W, int H, double x, double y, int dof, double lu, double lv, double* out6)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc; cx.W = W; cx.H = H; cx.focal_plane_dist = sc->camera.focal_plane_dist;
	ray_t r = camera_ray(&cx, x, y, dof, lu, lv, 0);
	out6[0] = r.start.x; out6[1] = r.start.y; out6[2] = r.start.z;
	out6[3] = r.dir.x; out6[4] = r.dir.y; out6[5] = r.dir.z;
}

/* ---- shading members pinned by tests/golden/reference_shading_vectors.json (oracle/ref_harness_shading.cpp) ---- */
typedef struct {   /* the fields of IntersectionInfo a shader / texture reads */
	double ip[3], norm[3], dNdx[3], dNdy[3], u, v;
} hxo_probe_hit;

static hit_t probe_hit(const hxo_probe_hit* p)
{
	hit_t h;
	memset(&h, 0, sizeof h);
	h.ip = vld(p->ip); h.norm = vld(p->norm); h.dNdx = vld(p->dNdx); h.dNdy = vld(p->dNdy);
	h.u = p->u; h.v = p->v;
	return h;
}

void hxo_box_split(const double* lo, const double* hi, int axis, double where, double* out12)
{
	box_t b = {vld(lo), vld(hi)}, l, r;
	box_split(&b, axis, where, &l, &r);
	const vec v[4] = {l.lo, l.hi, r.lo, r.hi};
	for (int i = 0; i < 4; i++) { out12[3 * i] = v[i].x; out12[3 * i + 1] = v[i].y; out12[3 * i + 2] = v[i].z; }
}

void hxo_hemisphere(double u, double v, const double* normal, double* out3)
{
	vec d = hemisphere(u, v, vld(normal));
	out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

int hxo_disk_map(double u, double v, double* xy) { return disk_map(u, v, &xy[0], &xy[1]); }

void hxo_glossy_dir(const double* n, const double* dir, float glossiness, double x, double y, double* out3)
{
	vec d = glossy_dir(vld(n), vld(dir), glossiness, x, y);
	out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

void hxo_texture_sample(const hxr_scene_desc* sc, int texture, const double* dir, const hxo_probe_hit* hit, float* out3)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	hit_t h = probe_hit(hit);
	col c = texture_sample(&cx, texture, vld(dir), &h);
	out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}

void hxo_texture_bump(const hxr_scene_desc* sc, int texture, hxo_probe_hit* hit)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	hit_t h = probe_hit(hit);
	texture_bump(&cx, texture, &h);
	hit->norm[0] = h.norm.x; hit->norm[1] = h.norm.y; hit->norm[2] = h.norm.z;
}

/* Shader::computeColor of shader `shader` with raytrace() replaced by a recorder: returns the number of rays it asked for,
   the first `max_rays` of them in rays[8 * i] = start, dir, depth, flags */
int hxo_shade_probe(const hxr_scene_desc* sc, int shader, const double* start, const double* dir, int depth, unsigned flags, const hxo_probe_hit* hit, float* out3,
                    double* rays, int max_rays)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	cx.probe_on = 1;
	cx.probe_rays = rays;
	cx.probe_max = max_rays;
	ray_t r;
	memset(&r, 0, sizeof r);
	r.start = vld(start); r.dir = vld(dir); r.depth = depth; r.flags = flags;
	hit_t h = probe_hit(hit);
	col c = shade(&cx, shader, &r, &h);
	out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
	return cx.probe_count;
}

/* hxo_shade_probe with the hooks Lambert / Phong need: visible() answers bit k of `vis_pattern` for the k-th call and records its two
   points (vis_points[6 * k], up to max_vis; *n_vis = calls made), RectLight jitter is taken from `draws` (two per light sample, in
   call order; *draws_used = numbers consumed) */
int hxo_shade_probe2(const hxr_scene_desc* sc, int shader, const double* start, const double* dir, int depth, unsigned flags, const hxo_probe_hit* hit, float* out3,
                     double* rays, int max_rays, uint64_t vis_pattern, double* vis_points, int max_vis, int* n_vis, const double* draws, int n_draws, int* draws_used)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	cx.probe_on = 1;
	cx.probe_rays = rays;
	cx.probe_max = max_rays;
	cx.probe_vis_pattern = vis_pattern;
	cx.probe_vis_points = vis_points;
	cx.probe_vis_max = max_vis;
	cx.probe_draws = draws;
	cx.probe_draws_len = n_draws;
	ray_t r;
	memset(&r, 0, sizeof r);
	r.start = vld(start); r.dir = vld(dir); r.depth = depth; r.flags = flags;
	hit_t h = probe_hit(hit);
	col c = shade(&cx, shader, &r, &h);
	out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
	if (n_vis) *n_vis = cx.probe_vis_count;
	if (draws_used) *draws_used = cx.probe_draws_used;
	return cx.probe_count;
}

void hxo_brdf_eval(const hxr_scene_desc* sc, int shader, const hxo_probe_hit* hit, const double* w_in, const double* w_out, float* out3)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	hit_t h = probe_hit(hit);
	col c = brdf_eval(&cx, sc->shaders + shader, &h, vld(w_in), vld(w_out));
	out3[0] = c.r; out3[1] = c.g; out3[2] = c.b;
}

/* BRDF::spawnRay; out8 = start, dir, depth, flags of w_out (which starts as a copy of the incoming ray, src/main.cpp:124-128) */
void hxo_brdf_spawn(const hxr_scene_desc* sc, int shader, const hxo_probe_hit* hit, const double* start, const double* dir, int depth, unsigned flags, double* out8,
                    float* color3, float* pdf)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	ray_t in, out;
	memset(&in, 0, sizeof in);
	in.start = vld(start); in.dir = vld(dir); in.depth = depth; in.flags = flags;
	out = in;
	hit_t h = probe_hit(hit);
	col c;
	brdf_spawn(&cx, shader, &h, &in, &out, &c, pdf);
	out8[0] = out.start.x; out8[1] = out.start.y; out8[2] = out.start.z; out8[3] = out.dir.x; out8[4] = out.dir.y; out8[5] = out.dir.z;
	out8[6] = out.depth; out8[7] = out.flags;
	color3[0] = c.r; color3[1] = c.g; color3[2] = c.b;
}

/* hxo_brdf_spawn with the two draws of hemisphereSample handed in (Lambert::spawnRay) */
void hxo_brdf_spawn2(const hxr_scene_desc* sc, int shader, const hxo_probe_hit* hit, const double* start, const double* dir, int depth, unsigned flags, double u, double v,
                     double* out8, float* color3, float* pdf)
{
	ctx_t cx;
	memset(&cx, 0, sizeof cx);
	cx.sc = sc;
	const double uv[2] = {u, v};
	cx.probe_draws = uv;
	cx.probe_draws_len = 2;
	ray_t in, out;
	memset(&in, 0, sizeof in);
	in.start = vld(start); in.dir = vld(dir); in.depth = depth; in.flags = flags;
	out = in;
	hit_t h = probe_hit(hit);
	col c;
	brdf_spawn(&cx, shader, &h, &in, &out, &c, pdf);
	out8[0] = out.start.x; out8[1] = out.start.y; out8[2] = out.start.z; out8[3] = out.dir.x; out8[4] = out.dir.y; out8[5] = out.dir.z;
	out8[6] = out.depth; out8[7] = out.flags;
	color3[0] = c.r; color3[1] = c.g; color3[2] = c.b;
}

void hxo_philox_words(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t key, uint32_t dim, uint32_t* out4)
{
	hxo_rand128 r = hxo_philox(seed, pixel, sample, key, dim);
	memcpy(out4, r.w, sizeof r.w);
}

int hxo_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
