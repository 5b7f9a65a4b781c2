// Developer micro-benchmark of the packet kd traversal: camera-like coherent packets against one mesh node.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihexray_amd/csrc -Ihexray_amd/host tools/bench_mesh.hip -Lhexray_amd/lib -lhexray_host -Wl,-rpath,$PWD/hexray_amd/lib -o /tmp/bench_mesh
#include "../hexray_amd/csrc/hxr_api.hip"
#include "host_api.h"
#include <chrono>

__global__ __launch_bounds__(256) void kmesh_bench(const HxrSceneDev* __restrict__ Sp, const HxrNodeDev* __restrict__ a_nodes, const HxrMeshDev* __restrict__ a_meshes,
    const HxrKdWide* __restrict__ a_kd, const HxrTriHot* __restrict__ a_hot, const int32_t* __restrict__ a_hot_ids, int node, int W, int H, int reps, double* out,
    unsigned long long* cycles)
{
	__shared__ WaveStack st[4];
	HxrSceneDev S = *Sp; S.nodes = a_nodes; S.meshes = a_meshes; S.kd = a_kd; S.hot = a_hot; S.hot_ids = a_hot_ids;
	Tally<false> tl;
	const int lane = threadIdx.x & 63;
	int packet = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	int npackets = (W / 8) * (H / 8);
	double acc = 0;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int r = 0; r < reps; r++)
	for (int p = packet; p < npackets; p += (gridDim.x * blockDim.x) >> 6) {
		int bx = p % (W / 8), by = p / (W / 8), mx, my;
		morton6(lane, mx, my);
		d3 o, d;
		camera_ray(S, W, H, bx * 8 + mx, by * 8 + my, 0, 0, 0, o, d);
		const HxrNodeDev& N = S.nodes[node];
		d3 ol = mulm(o - ld3(N.T.offset), N.T.inv_m);
		d3 dl = unit_lazy(mulm(d, N.T.inv_m));
		GeomHit g;
		bool h = mesh_query<false>(S, S.meshes[N.ref], ol, dl, g, st[threadIdx.x >> 6], tl);
		acc += h ? g.t : 0.0;
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
	if (lane == 0) atomicAdd(cycles, t1 - t0);
}

int main(int argc, char** argv)
{
	const char* scene = argc > 1 ? argv[1] : "data/meshes.hexray";
	int node = argc > 2 ? atoi(argv[2]) : 2;
	const char* ov = "GlobalSettings {\nframeWidth 1920\nframeHeight 1080\n}\nCamera {\n pos (104, 30, 80)\n yaw 0\n pitch 0\n}\n";
	char err[512];
	hxh_scene* hs = hxh_scene_load(scene, argc > 3 ? argv[3] : ov, err, sizeof err);
	if (!hs) { printf("%s\n", err); return 1; }
	hxr_init(0);
	hxr_scene* s = hxr_scene_create(hxh_scene_desc(hs));
	if (!s) { printf("%s\n", hxr_last_error()); return 1; }
	int W = 1920, H = 1080, grid = 2048;
	double* out; unsigned long long* cyc;
	hipMalloc(&out, (size_t)grid * 256 * 8); hipMalloc(&cyc, 8);
	const HxrSceneDev& V = s->host_view;
	for (int it = 0; it < 3; it++) {
		hipMemset(cyc, 0, 8);
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		hipEventRecord(a);
		hipLaunchKernelGGL(kmesh_bench, dim3(grid), dim3(256), 0, 0, s->dev, V.nodes, V.meshes, V.kd, V.hot, V.hot_ids, node, W, H, 4, out, cyc);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
		double packets = 4.0 * (W / 8) * (H / 8);
		printf("node %d: %.3f ms for %.0f packets  -> %.1f ns/packet/wave-slot, %.0f wave-cycles(100MHz ticks?)/packet\n", node, ms, packets, ms * 1e6 / packets, (double)c / packets);
	}
	return 0;
}
