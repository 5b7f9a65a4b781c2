// Counter-based random numbers for the Monte-Carlo render modes (device side).
//
// The reference draws from a thread_local std::mt19937 in bucket-scheduling order (reference
// src/util.cpp:81-99), which is neither reproducible across threads nor mappable onto 2^25 concurrent
// lanes.  The module replaces the GENERATOR, not the samplers, with Philox-4x32-10 (Salmon et al.,
// SC'11): every draw is a pure function of (seed, pixel, sample, ray key, dimension), so the image is
// independent of how samples are dealt to wavefronts or GPUs.  The dimension / ray-key map is fixed
// in DESIGN.md ("Random numbers"); the CPU oracle implements the same function separately.
#pragma once
#include <stdint.h>

struct HxrRand128 { uint32_t w0, w1, w2, w3; };

__device__ __forceinline__ HxrRand128 hxr_philox(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t ray_key, uint32_t dim)
{
	uint32_t c0 = pixel, c1 = sample, c2 = ray_key, c3 = dim;
	uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
	for (int r = 0; r < 10; r++) {
		uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
		uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
		uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
		c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	return HxrRand128{c0, c1, c2, c3};
}

// two uniform doubles in [0,1), 53 random bits each
__device__ __forceinline__ void hxr_uniform2(HxrRand128 r, double& a, double& b)
{
	uint64_t x = ((uint64_t)r.w0 << 32) | r.w1, y = ((uint64_t)r.w2 << 32) | r.w3;
	a = (double)(x >> 11) * (1.0 / 9007199254740992.0);
	b = (double)(y >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ uint32_t hxr_below(uint32_t word, uint32_t n) { return __umulhi(word, n); }

__device__ __forceinline__ uint32_t hxr_child_key(uint32_t key, int shader, uint32_t k)
{
	return (key ^ (uint32_t)(shader + 1)) * 2654435761u + k;
}

enum : uint32_t {
	HXR_DIM_PIXEL_JITTER = 0,
	HXR_DIM_HEMISPHERE = 2,
	HXR_DIM_LIGHT_PICK = 3,
	HXR_DIM_LIGHT_JITTER = 4,
	HXR_DIM_PREPASS = 5,            // sample position inside a 24x24 block of the coarse prepass
	HXR_DIM_LENS = 16,              // + attempt
	HXR_DIM_DIRECT_LIGHT = 0x10000, // + light * 4096 + sample
	HXR_DIM_GLOSSY = 0x2000000      // + sample * 256 + outer attempt * 16 + disk draw
};
