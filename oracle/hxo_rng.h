/*
 * TEST INFRASTRUCTURE -- part of the CPU oracle (see hxo_oracle.c).  Never linked into the product.
 *
 * Counter-based random numbers for the Monte-Carlo modes.  The reference draws from a
 * `thread_local std::mt19937` in bucket-scheduling order (reference src/util.cpp:81-99), which is not
 * reproducible across threads, so the build replaces the GENERATOR (not the
 * samplers) by Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
 * keyed on what the draw is for:
 *
 *     counter = (pixel index y*W+x, sample index, ray key, dimension)      key = (seed lo, seed hi)
 *
 * One call yields 128 bits = two uniform doubles in [0,1) with 53 random bits each, or four 32-bit
 * words.  The HIP kernels implement the same function independently (hexray_amd/csrc/hxr_rng.hip.h);
 * the GPU-vs-oracle Monte-Carlo parity tests compare the two bit for bit through the image.
 *
 * Dimension map (DESIGN.md section "Random numbers"):
 *     0                  pixel jitter (x + rand, y + rand), src/main.cpp:373-375
 *     16 + attempt       lens sample rejection loop (unitDiskSample, src/util.cpp:101-107), ray key 0
 *     2                  hemisphereSample at this ray's hit (src/util.cpp:109-129)
 *     3                  light choice (word 0) and light sample index (word 2), src/main.cpp:146,152
 *     4                  jitter of that light sample (RectLight::getNthSample, src/lights.cpp:37-41)
 *     0x10000 + L*4096+i jitter of sample i of light L in direct lighting (src/shading.cpp:78,123)
 *     0x2000000 + k*256 + a*16 + i  glossy reflection sample k, rejection attempt a, disk draw i (src/shading.cpp:171-179)
 * Ray keys: the camera ray has key 0; a ray spawned by shader S (table index) as its k-th ray from a
 * ray with key K gets (K ^ (S + 1)) * 2654435761 + k  (mod 2^32).
 */
#ifndef HXO_RNG_H
#define HXO_RNG_H
#include <stdint.h>

typedef struct { uint32_t w[4]; } hxo_rand128;

static inline hxo_rand128 hxo_philox(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t ray_key, uint32_t dim)
{
	uint32_t c0 = pixel, c1 = sample, c2 = ray_key, c3 = dim;
	uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
	for (int round = 0; round < 10; round++) {
		uint64_t p0 = (uint64_t)0xD2511F53u * c0;
		uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
		uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
		uint32_t n1 = (uint32_t)p1;
		uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		uint32_t n3 = (uint32_t)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	hxo_rand128 r = {{c0, c1, c2, c3}};
	return r;
}

/* two doubles in [0,1): 53 high bits of each 64-bit half */
static inline void hxo_uniform2(hxo_rand128 r, double* a, double* b)
{
	uint64_t x = ((uint64_t)r.w[0] << 32) | r.w[1];
	uint64_t y = ((uint64_t)r.w[2] << 32) | r.w[3];
	*a = (double)(x >> 11) * (1.0 / 9007199254740992.0);
	*b = (double)(y >> 11) * (1.0 / 9007199254740992.0);
}

/* integer in [0, n) from one 32-bit word */
static inline uint32_t hxo_below(uint32_t word, uint32_t n) { return (uint32_t)(((uint64_t)word * n) >> 32); }

static inline uint32_t hxo_child_key(uint32_t key, int shader, uint32_t k) { return (key ^ (uint32_t)(shader + 1)) * 2654435761u + k; }

#endif
