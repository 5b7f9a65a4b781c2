// Heightfield tables; see heightfield_prep.h.  Float accumulation order follows the reference loops
// because the heights feed cell-index decisions downstream.
#include "heightfield_prep.h"
#include <algorithm>
#include <cmath>

namespace hxh {

void HeightfieldData::build(Image& bmp, double blur, bool maxima_tables, bool raw)
{
	W = bmp.w;
	H = bmp.h;
	heights.assign(size_t(W) * H, 0.0f);
	float lo = 1e17f, hi = -1e17f;
	blur_kernel.clear();
	blur_radius = 0;
	if (raw) {   // hxr_heightfield::raw_heights: everything below happens on the GPU
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++) heights[size_t(y) * W + x] = bmp.intensity(x, y);
		if (blur > 0) {
			blur_radius = std::min(128, int(std::floor(float(3 * blur) + 0.5f)));
			// 0 < blur < 1/6: the radius rounds to 0, the reference's tap loops (src/heightfield.cpp:204-212) are empty and every height
			// becomes 0 -- not "no blur", which is what a radius of 0 means to the module.  Flatten the field here.
			if (blur_radius == 0) std::fill(heights.begin(), heights.end(), 0.0f);
			blur_kernel.resize(size_t(blur_radius) * blur_radius);
			for (int y = 0; y < blur_radius; y++)
				for (int x = 0; x < blur_radius; x++)
					blur_kernel[size_t(y) * blur_radius + x] = float(std::exp(-(double(x) * x + double(y) * y) / (2 * blur * blur)) / (2 * kPi * (blur * blur)));
		}
		bbox_min = V3(0, 0, 0);
		bbox_max = V3(W, 0, H);
		max_h.clear();
		normals.clear();
		high_map.clear();
		max_k = 0;
		return;
	}
	if (blur <= 0) {
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++) {
				float h = bmp.intensity(x, y);
				heights[size_t(y) * W + x] = h;
				lo = std::min(lo, h);
				hi = std::max(hi, h);
			}
	} else {
		std::vector<float> grey(size_t(W) * H);
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++) grey[size_t(y) * W + x] = bmp.intensity(x, y);
		auto grey_at = [&](int x, int y) -> float {
			return (x < 0 || x >= W || y < 0 || y >= H) ? 0.0f : grey[size_t(y) * W + x];
		};
		int R = std::min(128, int(std::floor(float(3 * blur) + 0.5f)));
		std::vector<float> g(size_t(R) * R);
		for (int y = 0; y < R; y++)
			for (int x = 0; x < R; x++)
				g[size_t(y) * R + x] = float(std::exp(-(double(x) * x + double(y) * y) / (2 * blur * blur)) / (2 * kPi * (blur * blur)));
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++) {
				float sum = 0;
				for (int dy = -R + 1; dy < R; dy++)
					for (int dx = -R + 1; dx < R; dx++)
						sum += g[size_t(std::abs(dy)) * R + std::abs(dx)] * grey_at(x + dx, y + dy);
				heights[size_t(y) * W + x] = sum;
				lo = std::min(lo, sum);
				hi = std::max(hi, sum);
			}
	}
	bbox_min = V3(0, lo, 0);
	bbox_max = V3(W, hi, H);

	max_h.clear();
	if (maxima_tables) max_h.assign(size_t(W) * H, 0.0f);
	for (int y = 0; maxima_tables && y < H; y++)
		for (int x = 0; x < W; x++) {
			float m = heights[size_t(y) * W + x];
			if (x < W - 1) m = std::max(m, heights[size_t(y) * W + x + 1]);
			if (y < H - 1) {
				m = std::max(m, heights[size_t(y + 1) * W + x]);
				if (x < W - 1) m = std::max(m, heights[size_t(y + 1) * W + x + 1]);
			}
			max_h[size_t(y) * W + x] = m;
		}

	normals.assign(size_t(W) * H, V3());
	for (int y = 0; y < H - 1; y++)
		for (int x = 0; x < W - 1; x++) {
			float h0 = heights[size_t(y) * W + x];
			float hx = heights[size_t(y) * W + x + 1];
			float hy = heights[size_t(y + 1) * W + x];
			V3 along_x(1, hx - h0, 0), along_z(0, hy - h0, 1);
			normals[size_t(y) * W + x] = unit(cross(along_z, along_x));
		}
	for (int y = 0; y < H; y++) normals[size_t(y) * W + W - 1] = normals[size_t(y) * W + W - 2];
	for (int x = 0; x < W; x++) normals[size_t(H - 1) * W + x] = normals[size_t(H - 2) * W + x];

	high_map.clear();
	max_k = 0;
	if (!use_optimization || !maxima_tables) return;
	high_map.assign(size_t(W) * H * 16, 0.0f);
	max_k = int(std::ceil(std::log(W) / std::log(2.0)));
	auto clampi = [](int v, int hi_) { return std::max(0, std::min(hi_, v)); };
	auto height_at = [&](int x, int y) { return heights[size_t(clampi(y, H - 1)) * W + clampi(x, W - 1)]; };
	auto highest_at = [&](int x, int y, int k) { return high_map[(size_t(clampi(y, H - 1)) * W + clampi(x, W - 1)) * 16 + k]; };
	for (int y = 0; y < H; y++)
		for (int x = 0; x < W; x++) {
			float r = height_at(x, y);
			for (int dy = -1; dy <= 1; dy++)
				for (int dx = -1; dx <= 1; dx++) r = std::max(r, height_at(x + dx, y + dy));
			high_map[(size_t(y) * W + x) * 16] = r;
		}
	for (int k = 1; k < max_k; k++) {
		int off = 1 << (k - 1);
		for (int y = 0; y < H; y++)
			for (int x = 0; x < W; x++) {
				float r = highest_at(x - off, y - off, k - 1);
				r = std::max(r, highest_at(x + off, y - off, k - 1));
				r = std::max(r, highest_at(x - off, y + off, k - 1));
				r = std::max(r, highest_at(x + off, y + off, k - 1));
				high_map[(size_t(y) * W + x) * 16 + k] = r;
			}
	}
}

}  // namespace hxh
