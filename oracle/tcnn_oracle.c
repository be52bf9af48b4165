/*
 * tcnn_oracle.c -- CPU restatement (plain C + OpenMP) of the tiny-cuda-nn hot path.
 * TEST INFRASTRUCTURE ONLY -- see tcnn_oracle.h for who may use it and for the pinning status.
 * Build with -ffp-contract=off: every float operation below is meant to round exactly once,
 * in the order written.
 */
#include "tcnn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* ------------------------------------------------------------------ fp16 / bfloat16 */

/* The product builds with IEEE fp16 or (libtcnn_hip_bf16.so, -DTCNN_BF16) with bfloat16 as its 16-bit type.  The oracle
 * follows with a run-time switch: every "half" below is the selected format.  Not thread-safe; set it before a test. */
static int g_bf16 = 0;
void orc_set_half_format(int bf16) { g_bf16 = bf16 ? 1 : 0; }
int orc_get_half_format(void) { return g_bf16; }

static uint16_t d2bf(double d) {
	/* double -> bfloat16 (1 + 8 + 7 bits), round-to-nearest-even, directly */
	union { double d; uint64_t u; } v;
	v.d = d;
	uint16_t sign = (uint16_t)((v.u >> 48) & 0x8000u);
	uint64_t absu = v.u & 0x7fffffffffffffffULL;
	if (absu >= 0x7ff0000000000000ULL) return (uint16_t)(sign | 0x7f80u | ((absu > 0x7ff0000000000000ULL) ? 0x40u : 0));
	if (absu == 0) return sign;
	int e = (int)(absu >> 52) - 1023;
	uint64_t mant = (absu & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
	if (e > 127) return (uint16_t)(sign | 0x7f80u);
	int shift, he;
	if (e >= -126) {
		shift = 45; /* keep 8 bits (1 + 7) */
		he = e + 127;
	} else {
		shift = 45 + (-126 - e);
		he = 0;
		if (shift > 63) return sign;
	}
	uint64_t kept = mant >> shift, rem = mant & (((uint64_t)1 << shift) - 1), half = (uint64_t)1 << (shift - 1);
	if (rem > half || (rem == half && (kept & 1))) kept++;
	uint32_t out = he == 0 ? (uint32_t)kept : ((uint32_t)he << 7) + (uint32_t)(kept - 0x80);
	if (out >= 0x7f80u) out = 0x7f80u;
	return (uint16_t)(sign | out);
}

static uint16_t d2h(double d) {
	/* double -> binary16, round-to-nearest-even, directly (no intermediate float rounding) */
	if (g_bf16) return d2bf(d);
	union { double d; uint64_t u; } v;
	v.d = d;
	uint16_t sign = (uint16_t)((v.u >> 48) & 0x8000u);
	uint64_t absu = v.u & 0x7fffffffffffffffULL;
	if (absu >= 0x7ff0000000000000ULL) { /* inf / nan */
		return (uint16_t)(sign | 0x7c00u | ((absu > 0x7ff0000000000000ULL) ? 0x200u : 0));
	}
	int e = (int)(absu >> 52) - 1023;
	uint64_t mant = (absu & 0x000fffffffffffffULL) | 0x0010000000000000ULL; /* 53 bits, implicit 1 */
	if (absu == 0) return sign;
	if (e > 15) return (uint16_t)(sign | 0x7c00u);
	int shift; /* number of low bits to drop from the 53-bit mantissa */
	int he;    /* biased half exponent */
	if (e >= -14) {
		shift = 42; /* keep 11 bits (1 + 10) */
		he = e + 15;
	} else {
		shift = 42 + (-14 - e); /* subnormal */
		he = 0;
		if (shift > 63) return sign; /* far below the smallest subnormal: rounds to zero */
	}
	uint64_t kept = mant >> shift;
	uint64_t rem = mant & (((uint64_t)1 << shift) - 1);
	uint64_t half = (uint64_t)1 << (shift - 1);
	if (rem > half || (rem == half && (kept & 1))) kept++;
	uint32_t out;
	if (he == 0) {
		out = (uint32_t)kept; /* may carry into exponent 1 naturally */
	} else {
		out = ((uint32_t)he << 10) + (uint32_t)(kept - 0x400); /* carry propagates into exponent */
	}
	if (out >= 0x7c00u) out = 0x7c00u;
	return (uint16_t)(sign | out);
}

uint16_t orc_f2h(float f) { return d2h((double)f); }

float orc_h2f(uint16_t h) {
	if (g_bf16) {
		union { uint32_t u; float f; } b;
		b.u = (uint32_t)h << 16;
		return b.f;
	}
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1f;
	uint32_t m = h & 0x3ff;
	union { uint32_t u; float f; } v;
	if (e == 0) {
		if (m == 0) {
			v.u = sign;
		} else {
			/* subnormal: m * 2^-24 */
			float f = (float)m * 5.9604644775390625e-08f;
			v.f = f;
			v.u |= sign;
		}
	} else if (e == 31) {
		v.u = sign | 0x7f800000u | (m << 13);
	} else {
		v.u = sign | ((e + 112) << 23) | (m << 13);
	}
	return v.f;
}

void orc_f2h_array(const float* in, uint16_t* out, size_t n) {
	for (size_t i = 0; i < n; ++i) out[i] = orc_f2h(in[i]);
}
void orc_h2f_array(const uint16_t* in, float* out, size_t n) {
	for (size_t i = 0; i < n; ++i) out[i] = orc_h2f(in[i]);
}

/* half fma with a single rounding: a*b is exact in double (22 significant bits); the sum is exact in
 * double unless the exponents are > 30 apart, in which case the small term cannot move the result
 * across a half rounding boundary (c is itself a half). */
static inline uint16_t hfma(uint16_t a, uint16_t b, uint16_t c) {
	/* bfloat16: gfx950 has no bf16 fma; the kernel's chain is an fp32 fma rounded to bf16 (tcnn_device.h fma_h) */
	if (g_bf16) return d2h((double)fmaf(orc_h2f(a), orc_h2f(b), orc_h2f(c)));
	return d2h((double)orc_h2f(a) * (double)orc_h2f(b) + (double)orc_h2f(c));
}
static inline uint16_t hmul(uint16_t a, uint16_t b) {
	return d2h((double)orc_h2f(a) * (double)orc_h2f(b));
}

/* ------------------------------------------------------------------ pcg32 */

#define PCG32_MULT 0x5851f42d4c957f2dULL

uint32_t orc_pcg32_next_uint(orc_pcg32* r) {
	uint64_t oldstate = r->state;
	r->state = oldstate * PCG32_MULT + r->inc;
	uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
	uint32_t rot = (uint32_t)(oldstate >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
	r->state = 0U;
	r->inc = (initseq << 1u) | 1u;
	orc_pcg32_next_uint(r);
	r->state += initstate;
	orc_pcg32_next_uint(r);
}

float orc_pcg32_next_float(orc_pcg32* r) {
	union { uint32_t u; float f; } x;
	x.u = (orc_pcg32_next_uint(r) >> 9) | 0x3f800000u;
	return x.f - 1.0f;
}

void orc_pcg32_advance(orc_pcg32* r, int64_t delta_) {
	uint64_t cur_mult = PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
	uint64_t delta = (uint64_t)delta_;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	r->state = acc_mult * r->state + acc_plus;
}

/* C++ [rand.util.seedseq] generate() for a one-element seed sequence and n = 2 output words. */
uint32_t orc_seed_seq_first(uint32_t seed) {
	const uint32_t n = 2, s = 1;
	uint32_t b[2] = {0x8b8b8b8bu, 0x8b8b8b8bu};
	const uint32_t v[1] = {seed};
	const uint32_t t = (n - 1) / 2; /* n < 7 */
	const uint32_t p = (n - t) / 2;
	const uint32_t q = p + t;
	const uint32_t m = (s + 1 > n) ? s + 1 : n;
	for (uint32_t k = 0; k < m; ++k) {
		uint32_t x = b[k % n] ^ b[(k + p) % n] ^ b[(k + n - 1) % n];
		uint32_t r1 = 1664525u * (x ^ (x >> 27));
		uint32_t r2 = r1;
		if (k == 0) r2 += s;
		else if (k <= s) r2 += (k % n) + v[k - 1];
		else r2 += (k % n);
		b[(k + p) % n] += r1;
		b[(k + q) % n] += r2;
		b[k % n] = r2;
	}
	for (uint32_t k = m; k < m + n; ++k) {
		uint32_t x = b[k % n] + b[(k + p) % n] + b[(k + n - 1) % n];
		uint32_t r3 = 1566083941u * (x ^ (x >> 27));
		uint32_t r4 = r3 - (k % n);
		b[(k + p) % n] ^= r3;
		b[(k + q) % n] ^= r4;
		b[k % n] = r4;
	}
	return b[0];
}

void orc_generate_random_uniform(orc_pcg32* rng, size_t n, float* out, float lower, float upper) {
	/* random.h:39-65.  N_TO_GENERATE = 4, 128-thread blocks (common.h:247).  Element
	 * idx = i + n_threads*j receives stream position 4*i + j.  The transform is
	 * val*(upper-lower)+lower, which the device compiler contracts to one fma. */
	const size_t n_gen = 4;
	size_t n_threads_needed = (n + n_gen - 1) / n_gen;
	size_t n_blocks = (n_threads_needed + 127) / 128;
	size_t n_threads = n_blocks * 128;
	const float range = upper - lower;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n_threads; ++ii) {
		size_t i = (size_t)ii;
		if (i >= n) continue;
		orc_pcg32 r = *rng;
		orc_pcg32_advance(&r, (int64_t)(i * n_gen));
		for (size_t j = 0; j < n_gen; ++j) {
			size_t idx = i + n_threads * j;
			if (idx >= n) break;
			out[idx] = fmaf(orc_pcg32_next_float(&r), range, lower);
		}
	}
	orc_pcg32_advance(rng, (int64_t)n);
}

/* ------------------------------------------------------------------ grid */

static uint32_t powi_u32(uint32_t base, uint32_t exponent) {
	uint32_t result = 1;
	for (uint32_t i = 0; i < exponent; ++i) result *= base;
	return result;
}

static uint32_t next_multiple_u32(uint32_t val, uint32_t divisor) {
	return ((val + divisor - 1) / divisor) * divisor;
}

int orc_grid_init(orc_grid* g, uint32_t n_dims, uint32_t n_levels, uint32_t n_features_per_level,
                  uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale,
                  int grid_type, int interpolation) {
	if (n_dims < 1 || n_dims > ORC_MAX_DIMS) return -1;
	if (n_levels > ORC_MAX_LEVELS) return -2;
	memset(g, 0, sizeof(*g));
	g->n_dims = n_dims;
	g->n_levels = n_levels;
	g->n_features_per_level = n_features_per_level;
	g->log2_hashmap_size = log2_hashmap_size;
	g->base_resolution = base_resolution;
	g->per_level_scale = per_level_scale;
	g->grid_type = grid_type;
	g->interpolation = interpolation;

	/* grid.h:699-727; grid_scale / grid_resolution: common_device.h:886-895 */
	const float log2_per_level_scale = log2f(per_level_scale);
	uint32_t offset = 0;
	for (uint32_t i = 0; i < n_levels; ++i) {
		const float scale = exp2f((float)i * log2_per_level_scale) * (float)base_resolution - 1.0f;
		const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
		g->scale[i] = scale;
		g->resolution[i] = resolution;

		const uint32_t max_params = 0xFFFFFFFFu / 2;
		uint32_t params_in_level =
			powf((float)resolution, (float)n_dims) > (float)max_params ? max_params : powi_u32(resolution, n_dims);
		params_in_level = next_multiple_u32(params_in_level, 8u);
		if (grid_type == ORC_GRID_DENSE) {
		} else if (grid_type == ORC_GRID_TILED) {
			uint32_t t = powi_u32(base_resolution, n_dims);
			if (t < params_in_level) params_in_level = t;
		} else if (grid_type == ORC_GRID_HASH) {
			uint32_t t = 1u << log2_hashmap_size;
			if (t < params_in_level) params_in_level = t;
		} else {
			return -3;
		}
		g->offsets[i] = offset;
		offset += params_in_level;
	}
	g->offsets[n_levels] = offset;
	g->n_params = offset * n_features_per_level;
	return 0;
}

static const uint32_t ORC_PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
static const uint32_t ORC_MAX_BASES[11] = {0x0, 0xFFFFFFFF, 0xFFFF, 0x659, 0xFF, 0x54, 0x28, 0x17, 0xF, 0xB, 0x9};

uint32_t orc_grid_index(const orc_grid* g, uint32_t level, const uint32_t* pos_grid) {
	/* common_device.h:847-884 with HashType::CoherentPrime (:787-791) */
	const uint32_t hashmap_size = g->offsets[level + 1] - g->offsets[level];
	const uint32_t res = g->resolution[level];
	uint32_t stride = 1;
	uint32_t index = 0;
	if (res <= ORC_MAX_BASES[g->n_dims]) {
		for (uint32_t d = 0; d < g->n_dims; ++d) {
			index += pos_grid[d] * stride;
			stride *= res;
		}
	} else {
		stride = 0xFFFFFFFFu;
	}
	if (g->grid_type == ORC_GRID_HASH && hashmap_size < stride) {
		uint32_t h = 0;
		for (uint32_t d = 0; d < g->n_dims; ++d) h ^= pos_grid[d] * ORC_PRIMES[d];
		index = h;
	}
	return index % hashmap_size;
}

static inline float smoothstep_f(float v) { return v * v * (3.0f - 2.0f * v); }
static inline float smoothstep_d(float v) { return 6 * v * (1.0f - v); }

/* common_device.h:1016-1043 pos_fract */
static inline void pos_fract(const orc_grid* g, float input, float scale, float* pos, float* pos_derivative,
                             uint32_t* pos_grid) {
	float p = fmaf(scale, input, 0.5f);
	float tmp = floorf(p);
	*pos_grid = (uint32_t)(int)tmp;
	p -= tmp;
	if (g->interpolation == ORC_INTERP_SMOOTHSTEP) {
		*pos_derivative = smoothstep_d(p);
		*pos = smoothstep_f(p);
	} else {
		*pos_derivative = 1.0f;
		*pos = p;
	}
}

void orc_grid_indices(const orc_grid* g, const float* positions, uint32_t n, uint32_t* indices, float* weights) {
	const uint32_t D = g->n_dims, L = g->n_levels, C = 1u << D;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		for (uint32_t level = 0; level < L; ++level) {
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) pos_fract(g, positions[(size_t)i * D + d], g->scale[level], &pos[d], &pd[d], &pg[d]);
			for (uint32_t idx = 0; idx < C; ++idx) {
				float weight = 1;
				uint32_t local[ORC_MAX_DIMS];
				for (uint32_t d = 0; d < D; ++d) {
					if ((idx & (1u << d)) == 0) {
						weight *= 1 - pos[d];
						local[d] = pg[d];
					} else {
						weight *= pos[d];
						local[d] = pg[d] + 1;
					}
				}
				size_t o = ((size_t)i * L + level) * C + idx;
				indices[o] = orc_grid_index(g, level, local);
				if (weights) weights[o] = weight;
			}
		}
	}
}

void orc_grid_forward(const orc_grid* g, const uint16_t* params, const float* positions, uint32_t n,
                      uint16_t* out, uint32_t out_stride, float* dy_dx) {
	const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features_per_level, C = 1u << D;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		uint16_t* o = out + (size_t)i * out_stride;
		for (uint32_t c = L * F; c < out_stride; ++c) o[c] = 0; /* grid.h:757-766 */
		for (uint32_t level = 0; level < L; ++level) {
			const uint16_t* grid = params + (size_t)g->offsets[level] * F;
			const float scale = g->scale[level];
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) pos_fract(g, positions[(size_t)i * D + d], scale, &pos[d], &pd[d], &pg[d]);

			if (g->interpolation == ORC_INTERP_NEAREST) {
				uint32_t index = orc_grid_index(g, level, pg) * F;
				for (uint32_t f = 0; f < F; ++f) o[level * F + f] = grid[index + f];
				if (dy_dx)
					for (uint32_t f = 0; f < F; ++f)
						for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)i * L * F + level * F + f) * D + d] = 0.0f;
				continue;
			}

			uint16_t result[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (uint32_t idx = 0; idx < C; ++idx) {
				float weight = 1;
				uint32_t local[ORC_MAX_DIMS];
				for (uint32_t d = 0; d < D; ++d) {
					if ((idx & (1u << d)) == 0) {
						weight *= 1 - pos[d];
						local[d] = pg[d];
					} else {
						weight *= pos[d];
						local[d] = pg[d] + 1;
					}
				}
				uint32_t index = orc_grid_index(g, level, local) * F;
				uint16_t wh = orc_f2h(weight);
				for (uint32_t f = 0; f < F; ++f) result[f] = hfma(wh, grid[index + f], result[f]); /* grid.h:162 */
			}
			for (uint32_t f = 0; f < F; ++f) o[level * F + f] = result[f];

			if (dy_dx) { /* grid.h:172-211 */
				for (uint32_t f = 0; f < F; ++f)
					for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)i * L * F + level * F + f) * D + d] = 0.0f;
				for (uint32_t gd = 0; gd < D; ++gd) {
					for (uint32_t idx = 0; idx < (C >> 1); ++idx) {
						float weight = scale;
						uint32_t local[ORC_MAX_DIMS];
						for (uint32_t ngd = 0; ngd + 1 < D; ++ngd) {
							const uint32_t dim = ngd >= gd ? (ngd + 1) : ngd;
							if ((idx & (1u << ngd)) == 0) {
								weight *= 1 - pos[dim];
								local[dim] = pg[dim];
							} else {
								weight *= pos[dim];
								local[dim] = pg[dim] + 1;
							}
						}
						local[gd] = pg[gd];
						uint32_t il = orc_grid_index(g, level, local) * F;
						local[gd] = pg[gd] + 1;
						uint32_t ir = orc_grid_index(g, level, local) * F;
						for (uint32_t f = 0; f < F; ++f) {
							float* dst = &dy_dx[((size_t)i * L * F + level * F + f) * D + gd];
							float diff = orc_h2f(grid[ir + f]) - orc_h2f(grid[il + f]);
							float t = weight * diff;
							t = t * pd[gd];
							*dst = *dst + t;
						}
					}
				}
			}
		}
	}
}

/* grid.h:284-299: stochastic interpolation -- the whole gradient of (sample, level) goes, unweighted, to ONE corner chosen
 * by a single uniform variate: per dimension the upper neighbour iff sample < pos[d].
 * random_val(1337, i + level * n): common_device.h:469-473 (pcg32{seed}, advance(idx), next_float). */
void orc_grid_backward_stochastic(const orc_grid* g, const float* positions, uint32_t n, const uint16_t* dL_dy,
                                  uint32_t dy_stride, double* grad) {
	const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features_per_level;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		for (uint32_t level = 0; level < L; ++level) {
			double* gg = grad + (size_t)g->offsets[level] * F;
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) pos_fract(g, positions[(size_t)i * D + d], g->scale[level], &pos[d], &pd[d], &pg[d]);
			const uint16_t* dy = dL_dy + (size_t)i * dy_stride + level * F;
			if (g->interpolation != ORC_INTERP_NEAREST) { /* Nearest returns before the stochastic branch (grid.h:279-282) */
				orc_pcg32 rng;
				orc_pcg32_seed(&rng, 1337u, 1u);
				orc_pcg32_advance(&rng, (int64_t)(uint32_t)(i + level * n));
				const float sample = orc_pcg32_next_float(&rng);
				for (uint32_t d = 0; d < D; ++d) {
					if (!(sample >= pos[d])) pg[d] += 1;
				}
			}
			uint32_t index = orc_grid_index(g, level, pg) * F;
			for (uint32_t f = 0; f < F; ++f) {
				double c = (double)orc_h2f(dy[f]);
#pragma omp atomic
				gg[index + f] += c;
			}
		}
	}
}

void orc_grid_backward(const orc_grid* g, const float* positions, uint32_t n, const uint16_t* dL_dy,
                       uint32_t dy_stride, double* grad) {
	const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features_per_level, C = 1u << D;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		for (uint32_t level = 0; level < L; ++level) {
			double* gg = grad + (size_t)g->offsets[level] * F;
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) pos_fract(g, positions[(size_t)i * D + d], g->scale[level], &pos[d], &pd[d], &pg[d]);
			const uint16_t* dy = dL_dy + (size_t)i * dy_stride + level * F;
			if (g->interpolation == ORC_INTERP_NEAREST) {
				uint32_t index = orc_grid_index(g, level, pg) * F;
				for (uint32_t f = 0; f < F; ++f) {
					double c = (double)orc_h2f(dy[f]);
#pragma omp atomic
					gg[index + f] += c;
				}
				continue;
			}
			for (uint32_t idx = 0; idx < C; ++idx) {
				float weight = 1;
				uint32_t local[ORC_MAX_DIMS];
				for (uint32_t d = 0; d < D; ++d) {
					if ((idx & (1u << d)) == 0) {
						weight *= 1 - pos[d];
						local[d] = pg[d];
					} else {
						weight *= pos[d];
						local[d] = pg[d] + 1;
					}
				}
				uint32_t index = orc_grid_index(g, level, local) * F;
				for (uint32_t f = 0; f < F; ++f) {
					double c;
					if (F == 1) {
						c = (double)(weight * orc_h2f(dy[f])); /* grad_t == float when F == 1 (grid.h:665) */
					} else {
						c = (double)orc_h2f(hmul(orc_f2h(weight), dy[f])); /* (GRAD_T)weight * grad, grid.h:254 */
					}
#pragma omp atomic
					gg[index + f] += c;
				}
			}
		}
	}
}

/* Second order: gradients of dL_dx = sum_k dL_dy[k] dy_dx[k] w.r.t. the grid parameters, dL_dy and the positions, given
 * ddx = dL/d(dL_dx).  Restates grid.h:352-455 (backward_input_backward_grid), :457-620 (backward_input_backward_input) and
 * :623-653 (backward_dLdoutput), following the reference's loop structure (gradient dimension outer, the 2^(D-1) corners
 * of the remaining dimensions inner, left / right along the gradient dimension).  grad_params (double, accumulated into),
 * dL_ddLdy (half [n][dy_stride], first L*F columns written) and dL_dx (fp32 [n][D], overwritten) may each be NULL.
 * dy_dx: orc_grid_forward's [i][k][d] layout, needed for dL_ddLdy only. */
void orc_grid_backward_backward_input(const orc_grid* g, const uint16_t* params, const float* positions, const float* ddx, uint32_t n,
                                      const uint16_t* dL_dy, uint32_t dy_stride, const float* dy_dx, double* grad_params,
                                      uint16_t* dL_ddLdy, float* dL_dx) {
	const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features_per_level, C2 = 1u << (D - 1);
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		if (dL_ddLdy) { /* grid.h:637-648 */
			for (uint32_t k = 0; k < L * F; ++k) {
				float result = 0;
				for (uint32_t d = 0; d < D; ++d) result += dy_dx[((size_t)i * L * F + k) * D + d] * ddx[(size_t)i * D + d];
				dL_ddLdy[(size_t)i * dy_stride + k] = orc_f2h(result);
			}
		}
		float out[ORC_MAX_DIMS] = {0};
		for (uint32_t level = 0; level < L && g->interpolation != ORC_INTERP_NEAREST; ++level) {
			const float scale = g->scale[level];
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS], pd2[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) {
				pos_fract(g, positions[(size_t)i * D + d], scale, &pos[d], &pd[d], &pg[d]);
				float p = fmaf(scale, positions[(size_t)i * D + d], 0.5f);
				p -= floorf(p);
				pd2[d] = g->interpolation == ORC_INTERP_SMOOTHSTEP ? 6.0f - 12.0f * p : 0.0f; /* common_device.h:984-986 */
			}
			const uint16_t* dy = dL_dy + (size_t)i * dy_stride + level * F;
			const uint16_t* table = params ? params + (size_t)g->offsets[level] * F : NULL;
			double* gg = grad_params ? grad_params + (size_t)g->offsets[level] * F : NULL;
			for (uint32_t gd = 0; gd < D; ++gd) {
				const float grad_in = scale * ddx[(size_t)i * D + gd] * pd[gd];                       /* grid.h:429 */
				const float grad_in_diag = scale * scale * ddx[(size_t)i * D + gd] * pd2[gd];           /* grid.h:546 */
				float grad_out = 0;
				for (uint32_t idx = 0; idx < C2; ++idx) {
					/* corners of the dimensions other than gd */
					float weight = 1;
					uint32_t local[ORC_MAX_DIMS];
					for (uint32_t nd = 0; nd + 1 < D; ++nd) {
						const uint32_t dim = nd >= gd ? nd + 1 : nd;
						if ((idx & (1u << nd)) == 0) { weight *= 1 - pos[dim]; local[dim] = pg[dim]; }
						else { weight *= pos[dim]; local[dim] = pg[dim] + 1; }
					}
					for (int side = 0; side < 2; ++side) { /* left, right along gd */
						local[gd] = pg[gd] + (uint32_t)side;
						const float sgn = side ? 1.0f : -1.0f;
						const uint32_t index = orc_grid_index(g, level, local) * F;
						if (gg) { /* grid.h:448-453: (GRAD_T)weight * grad */
							for (uint32_t f = 0; f < F; ++f) {
								const float w = sgn * grad_in * weight;
								double c = F == 1 ? (double)(w * orc_h2f(dy[f])) : (double)orc_h2f(hmul(orc_f2h(w), dy[f]));
#pragma omp atomic
								gg[index + f] += c;
							}
						}
						if (dL_dx && table && g->interpolation == ORC_INTERP_SMOOTHSTEP) { /* diagonal of the Hessian, grid.h:561-582 */
							float v = 0;
							for (uint32_t f = 0; f < F; ++f) v += orc_h2f(table[index + f]) * orc_h2f(dy[f]) * (sgn * grad_in_diag * weight);
							grad_out += v;
						}
					}
				}
				if (dL_dx && table) { /* mixed terms, grid.h:585-616: d(dy/dx[other]) / dx[gd] */
					for (uint32_t od = 0; od < D; ++od) {
						if (od == gd) continue;
						const float base = scale * scale * ddx[(size_t)i * D + od] * pd[od] * pd[gd];
						for (uint32_t idx = 0; idx < C2; ++idx) {
							float weight = base;
							uint32_t local[ORC_MAX_DIMS];
							for (uint32_t nd = 0; nd + 1 < D; ++nd) {
								const uint32_t dim = nd >= od ? nd + 1 : nd; /* dimensions other than od */
								if ((idx & (1u << nd)) == 0) {
									if (dim != gd) weight *= 1 - pos[dim]; else weight *= -1;
									local[dim] = pg[dim];
								} else {
									if (dim != gd) weight *= pos[dim];
									local[dim] = pg[dim] + 1;
								}
							}
							for (int side = 0; side < 2; ++side) {
								local[od] = pg[od] + (uint32_t)side;
								const uint32_t index = orc_grid_index(g, level, local) * F;
								float v = 0;
								for (uint32_t f = 0; f < F; ++f) v += orc_h2f(table[index + f]) * orc_h2f(dy[f]) * ((side ? 1.0f : -1.0f) * weight);
								grad_out += v;
							}
						}
					}
				}
				out[gd] += grad_out;
			}
		}
		if (dL_dx) for (uint32_t d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = out[d];
	}
}

void orc_grid_backward_input(const orc_grid* g, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride,
                             const float* dy_dx, float* dL_dx) {
	const uint32_t D = g->n_dims, K = g->n_levels * g->n_features_per_level;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		float result[ORC_MAX_DIMS] = {0, 0, 0, 0};
		for (uint32_t k = 0; k < K; ++k) {
			float dl = orc_h2f(dL_dy[(size_t)i * dy_stride + k]);
			for (uint32_t d = 0; d < D; ++d) {
				float t = dl * dy_dx[((size_t)i * K + k) * D + d];
				result[d] = result[d] + t;
			}
		}
		for (uint32_t d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = result[d];
	}
}

/* ---- GridEncodingTemplated<float> (Encoding<float>, cpp_api.cu:165-168): the same three kernels with T = float.  Parameters, encoded
 * features and gradients are fp32; the interpolation is result = fma((T)weight, value, result) in fp32 (grid.h:144-163), the scatter
 * adds (T)weight * grad per corner and feature (grid.h:252-255) -- accumulated here in double (the exact sum of the fp32 products; the
 * reference's fp32 atomics round after every addition, in launch order).  Layouts as in the 16-bit functions above. ---- */
void orc_grid_forward_f32(const orc_grid* g, const float* params, const float* positions, uint32_t n, float* out, uint32_t out_stride, float* dy_dx) {
	const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features_per_level, C = 1u << D;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		float* o = out + (size_t)i * out_stride;
		for (uint32_t c = L * F; c < out_stride; ++c) o[c] = 0.0f; /* grid.h:757-766 */
		for (uint32_t level = 0; level < L; ++level) {
			const float* grid = params + (size_t)g->offsets[level] * F;
			const float scale = g->scale[level];
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) pos_fract(g, positions[(size_t)i * D + d], scale, &pos[d], &pd[d], &pg[d]);
			if (dy_dx)
				for (uint32_t f = 0; f < F; ++f)
					for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)i * L * F + level * F + f) * D + d] = 0.0f;
			if (g->interpolation == ORC_INTERP_NEAREST) {
				uint32_t index = orc_grid_index(g, level, pg) * F;
				for (uint32_t f = 0; f < F; ++f) o[level * F + f] = grid[index + f];
				continue;
			}
			float result[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (uint32_t idx = 0; idx < C; ++idx) {
				float weight = 1;
				uint32_t local[ORC_MAX_DIMS];
				for (uint32_t d = 0; d < D; ++d) {
					if ((idx & (1u << d)) == 0) {
						weight *= 1 - pos[d];
						local[d] = pg[d];
					} else {
						weight *= pos[d];
						local[d] = pg[d] + 1;
					}
				}
				uint32_t index = orc_grid_index(g, level, local) * F;
				for (uint32_t f = 0; f < F; ++f) result[f] = fmaf(weight, grid[index + f], result[f]); /* grid.h:162, T = float */
			}
			for (uint32_t f = 0; f < F; ++f) o[level * F + f] = result[f];
			if (dy_dx) { /* grid.h:172-211 */
				for (uint32_t gd = 0; gd < D; ++gd) {
					for (uint32_t idx = 0; idx < (C >> 1); ++idx) {
						float weight = scale;
						uint32_t local[ORC_MAX_DIMS];
						for (uint32_t ngd = 0; ngd + 1 < D; ++ngd) {
							const uint32_t dim = ngd >= gd ? (ngd + 1) : ngd;
							if ((idx & (1u << ngd)) == 0) {
								weight *= 1 - pos[dim];
								local[dim] = pg[dim];
							} else {
								weight *= pos[dim];
								local[dim] = pg[dim] + 1;
							}
						}
						local[gd] = pg[gd];
						uint32_t il = orc_grid_index(g, level, local) * F;
						local[gd] = pg[gd] + 1;
						uint32_t ir = orc_grid_index(g, level, local) * F;
						for (uint32_t f = 0; f < F; ++f) {
							float* dst = &dy_dx[((size_t)i * L * F + level * F + f) * D + gd];
							float diff = grid[ir + f] - grid[il + f];
							float t = weight * diff;
							t = t * pd[gd];
							*dst = *dst + t;
						}
					}
				}
			}
		}
	}
}
void orc_grid_backward_f32(const orc_grid* g, const float* positions, uint32_t n, const float* dL_dy, uint32_t dy_stride, double* grad) {
	const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features_per_level, C = 1u << D;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		for (uint32_t level = 0; level < L; ++level) {
			double* gg = grad + (size_t)g->offsets[level] * F;
			float pos[ORC_MAX_DIMS], pd[ORC_MAX_DIMS];
			uint32_t pg[ORC_MAX_DIMS];
			for (uint32_t d = 0; d < D; ++d) pos_fract(g, positions[(size_t)i * D + d], g->scale[level], &pos[d], &pd[d], &pg[d]);
			const float* dy = dL_dy + (size_t)i * dy_stride + level * F;
			const uint32_t n_corners = g->interpolation == ORC_INTERP_NEAREST ? 1u : C;
			for (uint32_t idx = 0; idx < n_corners; ++idx) {
				float weight = 1;
				uint32_t local[ORC_MAX_DIMS];
				for (uint32_t d = 0; d < D; ++d) {
					if (g->interpolation == ORC_INTERP_NEAREST) {
						local[d] = pg[d];
					} else if ((idx & (1u << d)) == 0) {
						weight *= 1 - pos[d];
						local[d] = pg[d];
					} else {
						weight *= pos[d];
						local[d] = pg[d] + 1;
					}
				}
				uint32_t index = orc_grid_index(g, level, local) * F;
				for (uint32_t f = 0; f < F; ++f) {
					const double c = (double)(weight * dy[f]); /* (T)weight * grad, T = float (grid.h:254) */
#pragma omp atomic
					gg[index + f] += c;
				}
			}
		}
	}
}
void orc_grid_backward_input_f32(const orc_grid* g, uint32_t n, const float* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx) {
	const uint32_t D = g->n_dims, K = g->n_levels * g->n_features_per_level;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		float result[ORC_MAX_DIMS] = {0, 0, 0, 0};
		for (uint32_t k = 0; k < K; ++k) {
			float dl = dL_dy[(size_t)i * dy_stride + k];
			for (uint32_t d = 0; d < D; ++d) {
				float t = dl * dy_dx[((size_t)i * K + k) * D + d];
				result[d] = result[d] + t;
			}
		}
		for (uint32_t d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = result[d];
	}
}

/* ------------------------------------------------------------------ MLP */

int orc_mlp_init(orc_mlp* m, uint32_t in_width, uint32_t width, uint32_t out_width, uint32_t n_hidden,
                 int activation, int output_activation) {
	if (n_hidden < 1) return -1; /* fully_fused_mlp.cu:650-652 */
	if (in_width % 16 != 0) return -2;
	m->in_width = in_width;
	m->width = width;
	m->out_width = out_width;
	m->padded_out = next_multiple_u32(out_width, 16);
	m->n_hidden = n_hidden;
	m->activation = activation;
	m->output_activation = output_activation;
	m->n_params = width * in_width + (n_hidden - 1) * width * width + m->padded_out * width;
	return 0;
}

static void xavier(orc_pcg32* rng, float* w, uint32_t rows, uint32_t cols, float scale) {
	scale *= sqrtf(6.0f / (float)(rows + cols));
	for (size_t i = 0; i < (size_t)rows * cols; ++i) {
		float t = orc_pcg32_next_float(rng) * 2.0f;
		t = t * scale;
		w[i] = t - scale;
	}
}

void orc_mlp_init_params(const orc_mlp* m, orc_pcg32* rng, float* p, float scale) {
	xavier(rng, p, m->width, m->in_width, scale);
	p += (size_t)m->width * m->in_width;
	for (uint32_t i = 0; i + 1 < m->n_hidden; ++i) {
		xavier(rng, p, m->width, m->width, scale);
		p += (size_t)m->width * m->width;
	}
	xavier(rng, p, m->padded_out, m->width, scale);
}

/* common_device.h:108-186 (K_ACT = 10) */
static inline float act_fwd(int act, float x) {
	switch (act) {
		case ORC_ACT_RELU: return x > 0.0f ? x : 0.0f;
		case ORC_ACT_LEAKY_RELU: return x * (x > 0.0f ? 1.0f : orc_h2f(orc_f2h(0.01f)));  /* common_device.h:127: the slope is a (T) constant */
		case ORC_ACT_EXPONENTIAL: return expf(x);
		case ORC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
		case ORC_ACT_SQUAREPLUS: { float y = x * 10.0f; return 0.5f * (y + sqrtf(y * y + 4.0f)) / 10.0f; }
		case ORC_ACT_SOFTPLUS: return logf(expf(x * 10.0f) + 1.0f) / 10.0f;
		case ORC_ACT_TANH: return tanhf(x);
		default: return x;
	}
}

/* common_device.h:363-418: v * f'(x) through the POST-activation value y; the factor is a half in the reference */
static inline float act_bwd(int act, float v, float y) {
	float factor;
	switch (act) {
		case ORC_ACT_RELU: return y > 0.0f ? v : 0.0f;
		case ORC_ACT_LEAKY_RELU: factor = y > 0.0f ? 1.0f : 0.01f; break;
		case ORC_ACT_EXPONENTIAL: factor = y; break;
		case ORC_ACT_SIGMOID: factor = orc_h2f(orc_f2h(y * orc_h2f(orc_f2h(1.0f - y)))); break;
		case ORC_ACT_SQUAREPLUS: { float t = y * 10.0f; factor = t * t / (t * t + 1.0f); break; }
		case ORC_ACT_SOFTPLUS: factor = 1.0f - expf(-y * 10.0f); break;
		case ORC_ACT_TANH: factor = 1.0f - y * y; break;
		default: return v;
	}
	return v * orc_h2f(orc_f2h(factor));
}

/* the two functions above on arrays of halves (tests/test_oracle_ref.py pins them against the reference's warp_activation /
 * warp_activation_backward compiled for the host): y = (half)f((float)x);  out = (half)(v * (half)f'(y)) */
void orc_activation_forward(int act, uint32_t n, const uint16_t* x, uint16_t* y) {
	for (uint32_t i = 0; i < n; ++i) y[i] = orc_f2h(act_fwd(act, orc_h2f(x[i])));
}
void orc_activation_backward(int act, uint32_t n, const uint16_t* v, const uint16_t* y, uint16_t* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = orc_f2h(act_bwd(act, orc_h2f(v[i]), orc_h2f(y[i])));
}

/* one dense layer for one sample: out[o] = act(sum_i W[o][i] * in[i]); W pre-converted to float */
static void layer_fwd(const float* W, uint32_t n_out, uint32_t n_in, const float* in, int act, int accum_fp16,
                      uint16_t* out_h, float* out_f) {
	for (uint32_t o = 0; o < n_out; ++o) {
		const float* w = W + (size_t)o * n_in;
		float acc = 0.0f;
		if (!accum_fp16) {
			for (uint32_t i = 0; i < n_in; ++i) acc += w[i] * in[i];
		} else {
			for (uint32_t i0 = 0; i0 < n_in; i0 += 16) {
				float part = acc;
				for (uint32_t i = i0; i < i0 + 16 && i < n_in; ++i) part += w[i] * in[i];
				acc = orc_h2f(orc_f2h(part));
			}
		}
		uint16_t h = orc_f2h(act_fwd(act, acc));
		out_h[o] = h;
		if (out_f) out_f[o] = orc_h2f(h);
	}
}

void orc_mlp_forward(const orc_mlp* m, const uint16_t* params, const uint16_t* input, uint32_t n,
                     uint16_t* hidden, uint16_t* output, int accum_fp16) {
	const uint32_t W = m->width, IN = m->in_width, OUT = m->padded_out, H = m->n_hidden;
	float* Wf = (float*)malloc(sizeof(float) * m->n_params);
	orc_h2f_array(params, Wf, m->n_params);
#pragma omp parallel
	{
		float* a = (float*)malloc(sizeof(float) * (IN > W ? IN : W));
		float* b = (float*)malloc(sizeof(float) * (IN > W ? IN : W));
		uint16_t* hb = (uint16_t*)malloc(sizeof(uint16_t) * (W > OUT ? W : OUT));
#pragma omp for schedule(static)
		for (long long ii = 0; ii < (long long)n; ++ii) {
			size_t i = (size_t)ii;
			for (uint32_t k = 0; k < IN; ++k) a[k] = orc_h2f(input[i * IN + k]);
			const float* Wl = Wf;
			uint32_t n_in = IN;
			for (uint32_t l = 0; l < H; ++l) {
				uint16_t* dst = hidden ? hidden + ((size_t)l * n + i) * W : hb;
				layer_fwd(Wl, W, n_in, a, m->activation, accum_fp16, dst, b);
				Wl += (size_t)W * n_in;
				n_in = W;
				float* t = a; a = b; b = t;
			}
			if (output) layer_fwd(Wl, OUT, W, a, m->output_activation, accum_fp16, output + i * OUT, NULL);
		}
		free(a); free(b); free(hb);
	}
	free(Wf);
}

void orc_mlp_backward(const orc_mlp* m, const uint16_t* params, const uint16_t* input, const uint16_t* hidden,
                      const uint16_t* output, const uint16_t* dL_doutput, uint32_t n, double* grad_params,
                      uint16_t* dL_dinput) {
	orc_mlp_backward_ex(m, params, input, hidden, output, dL_doutput, n, grad_params, dL_dinput, 0);
}

/* accum_fp16 != 0: the reference's accumulator type (wmma half fragments fully_fused_mlp.cu:68,198; CUTLASS
 * ElementAccumulator = half, cutlass_matmul.h:67): the running sum is rounded to half after every 16 k-steps (one
 * 16x16x16 MMA).  For the weight gradients k runs over the SAMPLES: each thread rounds its running sums every 16
 * samples; the per-thread sums are combined in double (the reference's split-K reduction is not modelled). */
void orc_mlp_backward_ex(const orc_mlp* m, const uint16_t* params, const uint16_t* input, const uint16_t* hidden,
                         const uint16_t* output, const uint16_t* dL_doutput, uint32_t n, double* grad_params,
                         uint16_t* dL_dinput, int accum_fp16) {
	const uint32_t W = m->width, IN = m->in_width, OUT = m->padded_out, H = m->n_hidden;
	float* Wf = (float*)malloc(sizeof(float) * m->n_params);
	orc_h2f_array(params, Wf, m->n_params);
	/* parameter offsets per matrix */
	size_t off_in = 0, off_hidden = (size_t)W * IN, off_out = off_hidden + (size_t)(H - 1) * W * W;
	int nthreads = orc_num_threads();
	double* partial = grad_params ? (double*)calloc((size_t)nthreads * m->n_params, sizeof(double)) : NULL;
#pragma omp parallel
	{
#ifdef _OPENMP
		int tid = omp_get_thread_num();
#else
		int tid = 0;
#endif
		double* gp = partial ? partial + (size_t)tid * m->n_params : NULL;
		uint32_t mx = W > IN ? W : IN;
		if (OUT > mx) mx = OUT;
		float* d_cur = (float*)malloc(sizeof(float) * mx);
		float* d_nxt = (float*)malloc(sizeof(float) * mx);
		float* act = (float*)malloc(sizeof(float) * mx);
		uint32_t since_round = 0;
#pragma omp for schedule(static)
		for (long long ii = 0; ii < (long long)n; ++ii) {
			size_t i = (size_t)ii;
			if (accum_fp16 && gp && ++since_round == 16) { /* one MMA k-step of samples done: half accumulators */
				since_round = 0;
				for (size_t k = 0; k < m->n_params; ++k) gp[k] = (double)orc_h2f(d2h(gp[k]));
			}
			/* output layer: dY (half) */
			for (uint32_t o = 0; o < OUT; ++o) {
				float d = orc_h2f(dL_doutput[i * OUT + o]);
				/* output activation: continue from dL/d(pre-activation), a half (fully_fused_mlp.cu:760-763) */
				if (m->output_activation != ORC_ACT_NONE) d = orc_h2f(orc_f2h(act_bwd(m->output_activation, d, orc_h2f(output[i * OUT + o]))));
				d_cur[o] = d;
			}
			uint32_t n_cur = OUT;            /* width of d_cur */
			const float* Wl = Wf + off_out;  /* matrix producing d_cur's layer: [n_cur][W] */
			size_t goff = off_out;
			for (int l = (int)H - 1; l >= -1; --l) {
				/* activation feeding this matrix: hidden[l] for l>=0, else the network input */
				const uint32_t n_prev = (l >= 0) ? W : IN;
				const uint16_t* prev_h = (l >= 0) ? hidden + ((size_t)l * n + i) * W : input + i * IN;
				for (uint32_t k = 0; k < n_prev; ++k) act[k] = orc_h2f(prev_h[k]);
				if (gp) { /* dW[o][k] += d[o] * act[k]  (cutlass_mlp.cu:272,292,308) */
					for (uint32_t o = 0; o < n_cur; ++o) {
						double d = d_cur[o];
						if (d == 0.0) continue;
						double* row = gp + goff + (size_t)o * n_prev;
						for (uint32_t k = 0; k < n_prev; ++k) row[k] += d * (double)act[k];
					}
				}
				if (l >= 0 || dL_dinput) { /* d_prev = W^T d, then activation transfer on post-activation values */
					for (uint32_t k = 0; k < n_prev; ++k) d_nxt[k] = 0.0f;
					for (uint32_t o = 0; o < n_cur; ++o) {
						float d = d_cur[o];
						const float* w = Wl + (size_t)o * n_prev;
						for (uint32_t k = 0; k < n_prev; ++k) d_nxt[k] += w[k] * d;
						if (accum_fp16 && (o % 16 == 15 || o + 1 == n_cur)) {
							for (uint32_t k = 0; k < n_prev; ++k) d_nxt[k] = orc_h2f(orc_f2h(d_nxt[k]));
						}
					}
					if (l >= 0) {
						for (uint32_t k = 0; k < n_prev; ++k) {
							d_nxt[k] = orc_h2f(orc_f2h(act_bwd(m->activation, d_nxt[k], act[k]))); /* common_device.h:363-418 */
						}
					} else {
						for (uint32_t k = 0; k < n_prev; ++k) dL_dinput[i * IN + k] = orc_f2h(d_nxt[k]);
					}
				}
				if (l < 0) break;
				/* step to the previous matrix */
				n_cur = W;
				if (l >= 1) {
					goff = off_hidden + (size_t)(l - 1) * W * W;
				} else {
					goff = off_in;
				}
				Wl = Wf + goff;
				float* t = d_cur; d_cur = d_nxt; d_nxt = t;
			}
		}
		free(d_cur); free(d_nxt); free(act);
	}
	if (partial) {
		for (int t = 0; t < nthreads; ++t)
			for (size_t k = 0; k < m->n_params; ++k) grad_params[k] += partial[(size_t)t * m->n_params + k];
		free(partial);
	}
	free(Wf);
}

/* ------------------------------------------------------------------ loss */

void orc_loss(int loss_type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale,
              const uint16_t* prediction, const float* target, const float* data_pdf, float* values,
              uint16_t* gradients, uint64_t n_total_override) {
	const uint32_t n_elements = n * stride;
	const uint32_t n_total_u = n_total_override ? (uint32_t)n_total_override : n_elements / stride * dims;
	const float n_total = (float)n_total_u;
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n_elements; ++ii) {
		uint32_t i = (uint32_t)ii;
		const uint32_t intra = i % stride, inter = i / stride;
		if (intra >= dims) {
			if (values) values[i] = 0;
			gradients[i] = 0;
			continue;
		}
		const uint32_t target_idx = inter * dims + intra;
		const float p = orc_h2f(prediction[i]);
		const float pdf = data_pdf ? data_pdf[target_idx] : 1;
		const float tg = target[target_idx];
		const float difference = p - tg;
		float value, gradient; /* gradient: dL/dprediction * n_total */
		switch (loss_type) {
			case ORC_LOSS_RELATIVE_L2_LUMINANCE: { /* relative_l2_luminance.h:66-86 */
				const uint16_t* row = prediction + (size_t)inter * stride;
				float r = orc_h2f(row[0]), g = orc_h2f(row[1]), b = orc_h2f(row[2]);
				if (dims >= 6) { r += orc_h2f(row[3]); g += orc_h2f(row[4]); b += orc_h2f(row[5]); }
				const float lum = 0.299f * r + 0.587f * g + 0.114f * b;
				const float psq = lum * lum + 0.01f;
				value = difference * difference / psq / pdf / n_total;
				gradient = 2 * difference / psq / pdf;
				break;
			}
			case ORC_LOSS_RELATIVE_L2: { /* relative_l2.h:70-80 */
				const float psq = p * p + 0.01f;
				value = difference * difference / psq / pdf / n_total;
				gradient = 2 * difference / psq / pdf;
				break;
			}
			case ORC_LOSS_L1: /* l1.h:69-74 */
				value = fabsf(difference) / pdf / n_total;
				gradient = copysignf(1.0f / pdf, difference);
				break;
			case ORC_LOSS_RELATIVE_L1: { /* relative_l1.h:69-76 */
				const float scale = 1.0f / (fabsf(p) + 1e-2f) / pdf;
				value = fabsf(difference) * scale / n_total;
				gradient = copysignf(scale, difference);
				break;
			}
			case ORC_LOSS_MAPE: { /* mape.h:69-77 */
				const float scale = 1.0f / (fabsf(tg) + 1e-2f) / pdf;
				value = fabsf(difference) * scale / n_total;
				gradient = copysignf(scale, difference);
				break;
			}
			case ORC_LOSS_SMAPE: { /* smape.h:69-77 */
				const float scale = 1.0f / (0.5f * (fabsf(tg) + fabsf(p)) + 1e-2f) / pdf;
				value = fabsf(difference) * scale / n_total;
				gradient = copysignf(scale, difference);
				break;
			}
			case ORC_LOSS_CROSS_ENTROPY: { /* cross_entropy.h:66-76 */
				const float factor = -tg / pdf / n_total;
				value = factor * logf(p);
				if (values) values[i] = value;
				gradients[i] = orc_f2h(loss_scale * (factor / p));
				continue;
			}
			case ORC_LOSS_VARIANCE: { /* variance_is.h:66-76 */
				const float factor = tg * tg / pdf / n_total;
				value = factor / p - factor / pdf;
				if (values) values[i] = value;
				gradients[i] = orc_f2h(loss_scale * (-factor / (p * p)));
				continue;
			}
			default: /* L2, l2.h:69-74 */
				value = difference * difference / pdf / n_total;
				gradient = 2 * difference / pdf;
				break;
		}
		if (values) values[i] = value;
		gradients[i] = orc_f2h(loss_scale * gradient / n_total);
	}
}

/* ------------------------------------------------------------------ Adam */

void orc_adam_defaults(orc_adam_hparams* h) {
	/* optimizers/adam.h:330-351 member defaults */
	h->learning_rate = 1e-3f;
	h->beta1 = 0.9f;
	h->beta2 = 0.999f;
	h->epsilon = 1e-8f;
	h->l2_reg = 1e-8f;
	h->non_matrix_l2_reg = 0.0f;
	h->relative_weight_decay = 0.0f;
	h->absolute_weight_decay = 0.0f;
	h->weight_clipping_magnitude = 0.0f;
	h->gradient_clipping_magnitude = 0.0f;
	h->non_matrix_learning_rate_factor = 1.0f;
	h->adabound = 0;
	h->optimize_matrix_params = 1;
	h->optimize_non_matrix_params = 1;
	h->skip_zero_grad_non_matrix_params = 1;
}

void orc_adam_step(const orc_adam_hparams* h, uint32_t n, uint32_t n_matrix_weights, float loss_scale,
                   uint32_t current_step, float* weights_fp32, uint16_t* weights_half,
                   const uint16_t* gradients, float* m1, float* m2, uint32_t* param_steps) {
	float lower_lr_bound = 0;
	float upper_lr_bound = 3.402823466e+38f;
	if (h->adabound) { /* adam.h:165-168 */
		lower_lr_bound = 0.1f - 0.1f / ((1 - h->beta2) * (float)current_step + 1);
		upper_lr_bound = 0.1f + 0.1f / ((1 - h->beta2) * (float)current_step);
	}
#pragma omp parallel for schedule(static)
	for (long long ii = 0; ii < (long long)n; ++ii) {
		uint32_t i = (uint32_t)ii;
		float gradient = orc_h2f(gradients[i]) / loss_scale;
		if (i >= n_matrix_weights) {
			if (!h->optimize_non_matrix_params || (gradient == 0 && h->skip_zero_grad_non_matrix_params)) continue;
		} else {
			if (!h->optimize_matrix_params) continue;
		}
		const float weight_fp = weights_fp32[i];
		if (i < n_matrix_weights) gradient += h->l2_reg * weight_fp;
		else gradient += h->non_matrix_l2_reg * weight_fp;
		if (h->gradient_clipping_magnitude != 0.0f) {
			gradient = copysignf(fminf(fabsf(gradient), h->gradient_clipping_magnitude), gradient);
		}
		const float gradient_sq = gradient * gradient;
		float first_moment = m1[i] = h->beta1 * m1[i] + (1 - h->beta1) * gradient;
		const float second_moment = m2[i] = h->beta2 * m2[i] + (1 - h->beta2) * gradient_sq;
		float learning_rate = h->learning_rate;
		if (i >= n_matrix_weights) learning_rate *= h->non_matrix_learning_rate_factor;
		const uint32_t step = ++param_steps[i];
		learning_rate *= sqrtf(1 - powf(h->beta2, (float)step)) / (1 - powf(h->beta1, (float)step));
		const float effective_lr =
			fminf(fmaxf(learning_rate / (sqrtf(second_moment) + h->epsilon), lower_lr_bound), upper_lr_bound);
		/* common_device.h:1045-1048 weight_decay */
		const float rel = h->relative_weight_decay * learning_rate, ab = h->absolute_weight_decay * learning_rate;
		const float decayed = (1 - rel) * weight_fp - copysignf(ab, weight_fp);
		float new_weight = decayed - effective_lr * first_moment;
		if (h->weight_clipping_magnitude != 0.0f) {
			new_weight = fminf(fmaxf(new_weight, -h->weight_clipping_magnitude), h->weight_clipping_magnitude);
		}
		weights_fp32[i] = new_weight;
		weights_half[i] = orc_f2h(new_weight);
	}
}

/* ------------------------------------------------------------------ identity */

/* ---- frequency encoding: encodings/frequency.h:46-104.  sinf / cosf of the reference's fp32 argument (the reference uses the
 * __sinf / __cosf approximations, see the kernel's header: parity with the reference itself is to that intrinsic's accuracy). ---- */
#define ORC_PI_F 3.14159265358979323846f
void orc_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint16_t* out) {
	const uint32_t fan_out_encoded = n_dims * n_frequencies * 2;
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < padded; ++j) {
			if (j >= fan_out_encoded) {
				out[(size_t)i * padded + j] = orc_f2h(1.0f);
				continue;
			}
			const uint32_t d = j / (n_frequencies * 2), log2_frequency = (j / 2) % n_frequencies;
			const float phase_shift = (float)(j % 2) * (ORC_PI_F / 2);
			const float x = scalbnf(in[(size_t)i * n_dims + d], (int)log2_frequency);
			const float input = x * ORC_PI_F + phase_shift;
			out[(size_t)i * padded + j] = orc_f2h(sinf(input));
		}
	}
}
void orc_frequency_backward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, const uint16_t* dL_dy, float* dL_dx) {
	const uint32_t outputs_per_input = n_frequencies * 2;
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t d = 0; d < n_dims; ++d) {
			float result = 0;
			for (uint32_t k = 0; k < outputs_per_input; ++k) {
				const uint32_t j = d * outputs_per_input + k, log2_frequency = k / 2;
				const float phase_shift = (float)(k % 2) * (ORC_PI_F / 2);
				const float input = scalbnf(in[(size_t)i * n_dims + d], (int)log2_frequency) * ORC_PI_F + phase_shift;
				const float dy_dx = scalbnf(1.0f, (int)log2_frequency) * ORC_PI_F * cosf(input);
				result += orc_h2f(dL_dy[(size_t)i * padded + j]) * dy_dx;
			}
			dL_dx[(size_t)i * n_dims + d] = result;
		}
	}
}

/* ---- one-blob encoding: encodings/oneblob.h:84-164 (kernel_one_blob_soa / kernel_one_blob_backward) with the quartic
 * kernel of common_device.h:1076-1095.  Outputs fp16, padding value 1 (oneblob.h:214-216). ---- */
static float orc_quartic(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float tmp = fmaxf(1 - u * u, 0.0f);
	return ((float)15 / 16) * tmp * tmp;
}
static float orc_quartic_cdf_deriv(float x, float inv_radius) { return orc_quartic(x, inv_radius) * inv_radius; }
static float orc_quartic_cdf(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float u2 = u * u;
	const float u4 = u2 * u2;
	return fmaxf(0.0f, fminf(1.0f, ((float)15 / 16) * u * (1 - ((float)2 / 3) * u2 + ((float)1 / 5) * u4) + 0.5f));
}
void orc_oneblob_forward(uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint16_t* out) {
	uint32_t log2_bins = 0;
	while ((1u << log2_bins) < n_bins) ++log2_bins;
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < n_dims; ++j) {
			const float x = in[(size_t)i * n_dims + j];
			float left_cdf = orc_quartic_cdf(-x, (float)n_bins) + orc_quartic_cdf(-x - 1.0f, (float)n_bins) + orc_quartic_cdf(-x + 1.0f, (float)n_bins);
			for (uint32_t k = 0; k < n_bins; ++k) {
				const float right_boundary = scalbnf((float)(k + 1), -(int)log2_bins);
				const float right_cdf = orc_quartic_cdf(right_boundary - x, (float)n_bins) + orc_quartic_cdf(right_boundary - x - 1.0f, (float)n_bins) +
				                        orc_quartic_cdf(right_boundary - x + 1.0f, (float)n_bins);
				out[(size_t)i * padded + j * n_bins + k] = orc_f2h(right_cdf - left_cdf);
				left_cdf = right_cdf;
			}
		}
		for (uint32_t k = n_dims * n_bins; k < padded; ++k) out[(size_t)i * padded + k] = orc_f2h(1.0f);
	}
}
void orc_oneblob_backward(uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, const uint16_t* dL_dy, float* dL_dx) {
	uint32_t log2_bins = 0;
	while ((1u << log2_bins) < n_bins) ++log2_bins;
	for (uint32_t i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < n_dims; ++j) {
			const float x = in[(size_t)i * n_dims + j];
			float result = 0;
			float left_cdf = orc_quartic_cdf_deriv(-x, (float)n_bins) + orc_quartic_cdf_deriv(-x - 1.0f, (float)n_bins) + orc_quartic_cdf_deriv(-x + 1.0f, (float)n_bins);
			for (uint32_t k = 0; k < n_bins; ++k) {
				const float right_boundary = scalbnf((float)(k + 1), -(int)log2_bins);
				const float right_cdf = orc_quartic_cdf_deriv(right_boundary - x, (float)n_bins) + orc_quartic_cdf_deriv(right_boundary - x - 1.0f, (float)n_bins) +
				                        orc_quartic_cdf_deriv(right_boundary - x + 1.0f, (float)n_bins);
				const float deriv = left_cdf - right_cdf;
				left_cdf = right_cdf;
				result += orc_h2f(dL_dy[(size_t)i * padded + j * n_bins + k]) * deriv;
			}
			dL_dx[(size_t)i * n_dims + j] = result;
		}
	}
}

void orc_identity_forward(uint32_t n, uint32_t n_dims, uint32_t padded, const float* in, uint16_t* out) {
	for (size_t i = 0; i < n; ++i)
		for (uint32_t j = 0; j < padded; ++j)
			out[i * padded + j] = j < n_dims ? orc_f2h(in[i * n_dims + j] * 1.0f + 0.0f) : orc_f2h(1.0f);
}

/* ------------------------------------------------------------------ whole model */

int orc_model_init(orc_model* md, uint32_t n_in, uint32_t n_out, const orc_grid* g, uint32_t width,
                   uint32_t n_hidden, int loss_type, const orc_adam_hparams* adam) {
	(void)n_in;
	md->grid = *g;
	/* network_with_input_encoding.h:47: encoding output padded to the network's alignment (16) */
	uint32_t enc_out = next_multiple_u32(g->n_levels * g->n_features_per_level, 16);
	int r = orc_mlp_init(&md->mlp, enc_out, width, n_out, n_hidden, ORC_ACT_RELU, ORC_ACT_NONE);
	if (r) return r;
	md->loss_type = loss_type;
	md->adam = *adam;
	md->n_params = md->mlp.n_params + g->n_params;
	md->n_out = n_out;
	return 0;
}

double orc_training_step(const orc_model* md, uint32_t n, const float* positions, const float* targets,
                         float* params_fp32, uint16_t* params_half, uint16_t* grads_half, float* m1,
                         float* m2, uint32_t* steps, uint32_t current_step, float loss_scale,
                         int run_optimizer, uint16_t* out_prediction) {
	return orc_training_step_ex(md, n, positions, targets, params_fp32, params_half, grads_half, m1, m2, steps, current_step,
	                            loss_scale, run_optimizer, out_prediction, NULL);
}

/* Trainer::training_step with its optional arguments (trainer.h:254-357): data_pdf (losses divide by it),
 * external_dL_dy (replaces the loss gradient, trainer.h:124-128; the returned loss is then 0), dL_dinput
 * (network_with_input_encoding.h:83-113 -> grid.h:897-907), the data-parallel n_total, and the fp16-accumulate
 * bracket of the network numerics. */
double orc_training_step_ex(const orc_model* md, uint32_t n, const float* positions, const float* targets,
                            float* params_fp32, uint16_t* params_half, uint16_t* grads_half, float* m1,
                            float* m2, uint32_t* steps, uint32_t current_step, float loss_scale,
                            int run_optimizer, uint16_t* out_prediction, const orc_step_options* opt) {
	const orc_mlp* m = &md->mlp;
	const uint32_t IN = m->in_width, W = m->width, OUT = m->padded_out, H = m->n_hidden;
	const uint16_t* mlp_params = params_half;                 /* network first ... */
	const uint16_t* grid_params = params_half + m->n_params;  /* ... then encoding (nwie.h:115-122) */
	const int accum_fp16 = opt ? opt->accum_fp16 : 0;
	const uint32_t K = md->grid.n_levels * md->grid.n_features_per_level;

	uint16_t* enc = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * IN);
	uint16_t* hidden = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * W * H);
	uint16_t* out = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * OUT);
	uint16_t* dout = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * OUT);
	float* L = (float*)calloc((size_t)n * OUT, sizeof(float));
	uint16_t* denc = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * IN);
	double* gmlp = (double*)calloc(m->n_params, sizeof(double));
	double* ggrid = (double*)calloc(md->grid.n_params, sizeof(double));
	float* dy_dx = (opt && opt->dL_dinput) ? (float*)malloc(sizeof(float) * (size_t)n * K * md->grid.n_dims) : NULL;

	memset(enc, 0, sizeof(uint16_t) * (size_t)n * IN); /* padded encoding columns are zero (grid.h:757-766) */
	orc_grid_forward(&md->grid, grid_params, positions, n, enc, IN, dy_dx);
	orc_mlp_forward(m, mlp_params, enc, n, hidden, out, accum_fp16);
	const uint16_t* dL_dy = dout;
	if (opt && opt->external_dL_dy) {
		dL_dy = opt->external_dL_dy;
	} else {
		orc_loss(md->loss_type, n, OUT, md->n_out, loss_scale, out, targets, opt ? opt->data_pdf : NULL, L, dout,
		         opt ? opt->n_total_override : 0);
	}
	orc_mlp_backward_ex(m, mlp_params, enc, hidden, out, dL_dy, n, gmlp, denc, accum_fp16);
	orc_grid_backward(&md->grid, positions, n, denc, IN, ggrid);
	if (dy_dx) orc_grid_backward_input(&md->grid, n, denc, IN, dy_dx, opt->dL_dinput);

	double loss = 0.0;
	for (size_t i = 0; i < (size_t)n * OUT; ++i) loss += (double)L[i];
	for (size_t i = 0; i < m->n_params; ++i) grads_half[i] = d2h(gmlp[i]);
	for (size_t i = 0; i < md->grid.n_params; ++i) grads_half[m->n_params + i] = d2h(ggrid[i]);
	if (out_prediction) memcpy(out_prediction, out, sizeof(uint16_t) * (size_t)n * OUT);

	if (run_optimizer) {
		orc_adam_step(&md->adam, md->n_params, m->n_params, loss_scale, current_step, params_fp32, params_half,
		              grads_half, m1, m2, steps);
	}
	free(enc); free(hidden); free(out); free(dout); free(L); free(denc); free(gmlp); free(ggrid); free(dy_dx);
	return loss;
}

void orc_inference(const orc_model* md, uint32_t n, const float* positions, const uint16_t* params_half,
                   float* outf) {
	const orc_mlp* m = &md->mlp;
	uint16_t* enc = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * m->in_width);
	uint16_t* out = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)n * m->padded_out);
	orc_grid_forward(&md->grid, params_half + m->n_params, positions, n, enc, m->in_width, NULL);
	orc_mlp_forward(m, params_half, enc, n, NULL, out, 0);
	for (size_t i = 0; i < n; ++i) /* object.cu:61-67 trim_and_cast */
		for (uint32_t j = 0; j < md->n_out; ++j) outf[i * md->n_out + j] = orc_h2f(out[i * m->padded_out + j]);
	free(enc); free(out);
}
