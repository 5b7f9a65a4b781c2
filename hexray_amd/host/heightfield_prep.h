// Heightfield preprocessing on the host: heights (optionally Gaussian-blurred), per-cell 2x2 maxima,
// per-texel normals and the 2^k-radius maxima pyramid used to skip empty space.
// Reference: src/heightfield.cpp:49-81 (buildHighMap), :173-261 (fillProperties).
#pragma once
#include <vector>
#include "image_io.h"
#include "linalg.h"

namespace hxh {

struct HeightfieldData {
	int W = 0, H = 0, max_k = 0;
	bool use_optimization = true;
	std::vector<float> heights, max_h;
	std::vector<V3> normals;
	std::vector<float> high_map;   // 16 floats per cell
	V3 bbox_min, bbox_max;

	std::vector<float> blur_kernel;   // raw mode: the R x R Gaussian table
	int blur_radius = 0;
	// `raw`: stop after the image: heights = unblurred intensities + the Gaussian table; blur, box, normals and maxima are left to the render module
	void build(Image& bmp, double blur, bool maxima_tables = true, bool raw = false);   // false: max_h / high_map / max_k are left to the render module (hxr_heightfield::first_maxh = -1)
};

}  // namespace hxh
