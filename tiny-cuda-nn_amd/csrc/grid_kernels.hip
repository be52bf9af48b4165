// grid_kernels.hip -- see grid_kernels.h for the design notes and the reference lines restated.
// Build with -ffp-contract=off: the fp32 position/weight arithmetic must round exactly as written so
// that indices AND interpolated features are bit-identical to the CPU oracle.
#include "grid_kernels.h"

#include <stdexcept>
#include <string>

namespace tcnn_hip {

constexpr uint32_t GRID_THREADS = 256;
constexpr uint32_t GRID_SPT = 4;  // samples per thread (independent gathers in flight per lane)
constexpr uint32_t GRID_TILE = GRID_THREADS * GRID_SPT;

// Work distribution: block b -> (level, tile) with level % 8 == b % 8.  Blocks are dispatched
// round-robin over the 8 XCDs (observed, not guaranteed): every XCD then gathers from ceil(L/8)
// level tables only, in level order, so one table at a time is hot in its private L2.
TCNN_DEVICE bool grid_work_item(uint32_t n_levels, uint32_t tiles, uint32_t& level, uint32_t& tile) {
	const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
	if (xcd >= n_levels) return false;
	const uint32_t levels_here = (n_levels - xcd + 7u) / 8u;
	if (slot >= levels_here * tiles) return false;
	level = xcd + 8u * (slot / tiles);
	tile = slot % tiles;
	return true;
}

static uint32_t grid_n_blocks(uint32_t n_levels, uint32_t n) {
	return 8u * div_round_up(n_levels, 8u) * div_round_up(n, GRID_TILE);
}

TCNN_DEVICE float smoothstep(float v) { return v * v * (3.0f - 2.0f * v); }
TCNN_DEVICE float smoothstep_derivative(float v) { return 6 * v * (1.0f - v); }

// reference common_device.h:1016-1043
TCNN_DEVICE void pos_fract(float input, float scale, bool smooth, float& pos, float& pos_derivative, uint32_t& pos_grid) {
	float p = __builtin_fmaf(scale, input, 0.5f);
	const float tmp = __builtin_floorf(p);
	pos_grid = (uint32_t)(int)tmp;
	p -= tmp;
	if (smooth) {
		pos_derivative = smoothstep_derivative(p);
		pos = smoothstep(p);
	} else {
		pos_derivative = 1.0f;
		pos = p;
	}
}

// F halves at `p` -> NP packed pairs (F == 1: {x, 0})
template <uint32_t F>
TCNN_DEVICE void load_features(const half_t* p, h2 (&v)[(F + 1) / 2]) {
	if constexpr (F == 1) {
		v[0] = h2{p[0], (half_t)0.0f};
	} else if constexpr (F == 2) {
		v[0] = *(const h2*)p;
	} else if constexpr (F == 4) {
		const h4 t = *(const h4*)p;
		v[0] = h2{t[0], t[1]};
		v[1] = h2{t[2], t[3]};
	} else {
		static_assert(F == 8, "n_features_per_level must be 1, 2, 4 or 8 (grid.h:1811-1821)");
		const h8 t = *(const h8*)p;
		v[0] = h2{t[0], t[1]};
		v[1] = h2{t[2], t[3]};
		v[2] = h2{t[4], t[5]};
		v[3] = h2{t[6], t[7]};
	}
}

template <uint32_t D, uint32_t F>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_forward(const GridMeta meta, const GridIO io, const half_t* __restrict__ params,
                                                                half_t* __restrict__ out, float* __restrict__ dy_dx) {
	uint32_t level, tile;
	if (!grid_work_item(meta.n_levels, div_round_up(io.n, GRID_TILE), level, tile)) return;

	constexpr uint32_t NP = (F + 1) / 2;
	const uint32_t hashmap_size = meta.offset[level + 1] - meta.offset[level];
	const half_t* __restrict__ grid = params + (size_t)meta.offset[level] * F;
	const float scale = meta.scale[level];
	const uint32_t resolution = meta.resolution[level];
	const bool is_hash = meta.grid_type == (uint32_t)GridType::Hash;
	const bool smooth = meta.interp == (uint32_t)InterpolationType::Smoothstep;
	const bool nearest = meta.interp == (uint32_t)InterpolationType::Nearest;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;  // grid.h:72
	const bool level_off = (float)level >= max_level + 1e-3f;               // grid.h:75

#pragma unroll
	for (uint32_t s = 0; s < GRID_SPT; ++s) {
		const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
		if (i >= io.n) continue;

		h2 result[NP];
#pragma unroll
		for (uint32_t p = 0; p < NP; ++p) result[p] = h2{(half_t)0.0f, (half_t)0.0f};
		float grads[F][D];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f)
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) grads[f][d] = 0.0f;

		if (!level_off) {
			float pos[D], pos_derivative[D];
			uint32_t pos_grid[D];
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) {
				pos_fract(io.positions[(size_t)i * io.pos_stride_i + (size_t)d * io.pos_stride_d], scale, smooth, pos[d], pos_derivative[d], pos_grid[d]);
			}

			if (nearest) {
				const uint32_t index = grid_index<D>(is_hash, hashmap_size, resolution, pos_grid);
				load_features<F>(grid + (size_t)index * F, result);
			} else {
				// N-linear interpolation, corner order and fp16 fma chain of grid.h:144-163
#pragma unroll
				for (uint32_t idx = 0; idx < (1u << D); ++idx) {
					float weight = 1;
					uint32_t local[D];
#pragma unroll
					for (uint32_t d = 0; d < D; ++d) {
						if ((idx & (1u << d)) == 0) {
							weight *= 1 - pos[d];
							local[d] = pos_grid[d];
						} else {
							weight *= pos[d];
							local[d] = pos_grid[d] + 1;
						}
					}
					const uint32_t index = grid_index<D>(is_hash, hashmap_size, resolution, local);
					h2 val[NP];
					load_features<F>(grid + (size_t)index * F, val);
					const half_t wh = to_half_rn(weight);
					const h2 w2 = h2{wh, wh};
#pragma unroll
					for (uint32_t p = 0; p < NP; ++p) result[p] = fma_h2(w2, val[p], result[p]);
				}

				if (dy_dx) {  // grid.h:172-211
#pragma unroll
					for (uint32_t gd = 0; gd < D; ++gd) {
#pragma unroll
						for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
							float weight = scale;
							uint32_t local[D];
#pragma unroll
							for (uint32_t ngd = 0; ngd < D - 1; ++ngd) {
								const uint32_t dim = ngd >= gd ? (ngd + 1) : ngd;
								if ((idx & (1u << ngd)) == 0) {
									weight *= 1 - pos[dim];
									local[dim] = pos_grid[dim];
								} else {
									weight *= pos[dim];
									local[dim] = pos_grid[dim] + 1;
								}
							}
							local[gd] = pos_grid[gd];
							h2 vl[NP], vr[NP];
							load_features<F>(grid + (size_t)grid_index<D>(is_hash, hashmap_size, resolution, local) * F, vl);
							local[gd] = pos_grid[gd] + 1;
							load_features<F>(grid + (size_t)grid_index<D>(is_hash, hashmap_size, resolution, local) * F, vr);
#pragma unroll
							for (uint32_t f = 0; f < F; ++f) {
								const float diff = (float)vr[f / 2][f % 2] - (float)vl[f / 2][f % 2];
								float t = weight * diff;
								t = t * pos_derivative[gd];
								grads[f][gd] = grads[f][gd] + t;
							}
						}
					}
				}
			}
		}

#pragma unroll
		for (uint32_t f = 0; f < F; ++f) {
			const uint32_t k = level * F + f;
			if (out) out[(size_t)k * io.stride_k + (size_t)i * io.stride_i] = result[f / 2][f % 2];
			if (dy_dx) {
#pragma unroll
				for (uint32_t d = 0; d < D; ++d) dy_dx[((size_t)k * io.n + i) * D + d] = grads[f][d];
			}
		}
	}
}

struct SmallLevels {
	uint32_t mask[4];    // bit l set: level l is handled by the LDS kernel
	uint32_t count;
	uint32_t levels[16]; // first `count` small levels
};

template <uint32_t F, typename GRAD_T>
TCNN_DEVICE void scatter_add(GRAD_T* grad, uint32_t index, const h2 (&g)[(F + 1) / 2], float weight) {
	if constexpr (F == 1) {
		// grad_t == float when F == 1 (grid.h:665): fp32 product, fp32 atomic
		atomic_add_f32(grad + index, weight * (float)g[0][0]);
	} else {
		const half_t wh = to_half_rn(weight);
		const h2 w2 = h2{wh, wh};
#pragma unroll
		for (uint32_t p = 0; p < F / 2; ++p) atomic_add_h2(grad + (size_t)index * F + 2 * p, w2 * g[p]);  // (GRAD_T)weight * grad, grid.h:254
	}
}

template <uint32_t D, uint32_t F, typename GRAD_T>
__global__ void __launch_bounds__(GRID_THREADS) k_grid_backward(const GridMeta meta, const GridIO io, const SmallLevels small,
                                                                 const half_t* __restrict__ dL_dy, GRAD_T* __restrict__ grid_gradient) {
	uint32_t level, tile;
	if (!grid_work_item(meta.n_levels, div_round_up(io.n, GRID_TILE), level, tile)) return;
	if (small.mask[level >> 5] & (1u << (level & 31u))) return;

	constexpr uint32_t NP = (F + 1) / 2;
	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	if ((float)level > max_level + 1e-3f) return;  // grid.h:242 (sic: '>' here, '>=' in forward)

	const uint32_t hashmap_size = meta.offset[level + 1] - meta.offset[level];
	GRAD_T* __restrict__ grad = grid_gradient + (size_t)meta.offset[level] * F;
	const float scale = meta.scale[level];
	const uint32_t resolution = meta.resolution[level];
	const bool is_hash = meta.grid_type == (uint32_t)GridType::Hash;
	const bool smooth = meta.interp == (uint32_t)InterpolationType::Smoothstep;
	const bool nearest = meta.interp == (uint32_t)InterpolationType::Nearest;

#pragma unroll
	for (uint32_t s = 0; s < GRID_SPT; ++s) {
		const uint32_t i = tile * GRID_TILE + s * GRID_THREADS + threadIdx.x;
		if (i >= io.n) continue;

		float pos[D], pos_derivative[D];
		uint32_t pos_grid[D];
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) {
			pos_fract(io.positions[(size_t)i * io.pos_stride_i + (size_t)d * io.pos_stride_d], scale, smooth, pos[d], pos_derivative[d], pos_grid[d]);
		}
		h2 g[NP];
#pragma unroll
		for (uint32_t p = 0; p < NP; ++p) g[p] = h2{(half_t)0.0f, (half_t)0.0f};
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) g[f / 2][f % 2] = dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];

		if (nearest) {
			scatter_add<F, GRAD_T>(grad, grid_index<D>(is_hash, hashmap_size, resolution, pos_grid), g, 1.0f);
			continue;
		}
#pragma unroll
		for (uint32_t idx = 0; idx < (1u << D); ++idx) {
			float weight = 1;
			uint32_t local[D];
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) {
				if ((idx & (1u << d)) == 0) {
					weight *= 1 - pos[d];
					local[d] = pos_grid[d];
				} else {
					weight *= pos[d];
					local[d] = pos_grid[d] + 1;
				}
			}
			scatter_add<F, GRAD_T>(grad, grid_index<D>(is_hash, hashmap_size, resolution, local), g, weight);
		}
	}
}

// Small levels: the whole level table lives in LDS as fp32; one flush of packed-half atomics per
// workgroup.  Removes the atomic hot-spotting of coarse levels (level 0 of the headline config has
// 4096 entries receiving 2^21 updates per step).
constexpr uint32_t GRID_LDS_THREADS = 512;
constexpr uint32_t GRID_LDS_SAMPLES_PER_BLOCK = 16384;

template <uint32_t D, uint32_t F, typename GRAD_T>
__global__ void __launch_bounds__(GRID_LDS_THREADS) k_grid_backward_lds(const GridMeta meta, const GridIO io, const SmallLevels small,
                                                                          const half_t* __restrict__ dL_dy, GRAD_T* __restrict__ grid_gradient) {
	TCNN_DYN_LDS(lds_raw);
	float* lds_table = (float*)lds_raw;
	const uint32_t chunks = div_round_up(io.n, GRID_LDS_SAMPLES_PER_BLOCK);
	const uint32_t level = small.levels[blockIdx.x / chunks];
	const uint32_t chunk = blockIdx.x % chunks;

	const uint32_t n_features = meta.n_levels * F;
	const float max_level = (meta.max_level * (float)n_features) / (float)F;
	if ((float)level > max_level + 1e-3f) return;

	const uint32_t hashmap_size = meta.offset[level + 1] - meta.offset[level];
	GRAD_T* __restrict__ grad = grid_gradient + (size_t)meta.offset[level] * F;
	const float scale = meta.scale[level];
	const uint32_t resolution = meta.resolution[level];
	const bool is_hash = meta.grid_type == (uint32_t)GridType::Hash;
	const bool smooth = meta.interp == (uint32_t)InterpolationType::Smoothstep;
	const bool nearest = meta.interp == (uint32_t)InterpolationType::Nearest;

	for (uint32_t e = threadIdx.x; e < hashmap_size * F; e += GRID_LDS_THREADS) lds_table[e] = 0.0f;
	__syncthreads();

	const uint32_t begin = chunk * GRID_LDS_SAMPLES_PER_BLOCK;
	const uint32_t end = min(begin + GRID_LDS_SAMPLES_PER_BLOCK, io.n);
	for (uint32_t i = begin + threadIdx.x; i < end; i += GRID_LDS_THREADS) {
		float pos[D], pos_derivative[D];
		uint32_t pos_grid[D];
#pragma unroll
		for (uint32_t d = 0; d < D; ++d) {
			pos_fract(io.positions[(size_t)i * io.pos_stride_i + (size_t)d * io.pos_stride_d], scale, smooth, pos[d], pos_derivative[d], pos_grid[d]);
		}
		float g[F];
#pragma unroll
		for (uint32_t f = 0; f < F; ++f) g[f] = (float)dL_dy[(size_t)(level * F + f) * io.stride_k + (size_t)i * io.stride_i];

		if (nearest) {
			const uint32_t index = grid_index<D>(is_hash, hashmap_size, resolution, pos_grid);
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) lds_atomic_add_f32(&lds_table[index * F + f], g[f]);
			continue;
		}
#pragma unroll
		for (uint32_t idx = 0; idx < (1u << D); ++idx) {
			float weight = 1;
			uint32_t local[D];
#pragma unroll
			for (uint32_t d = 0; d < D; ++d) {
				if ((idx & (1u << d)) == 0) {
					weight *= 1 - pos[d];
					local[d] = pos_grid[d];
				} else {
					weight *= pos[d];
					local[d] = pos_grid[d] + 1;
				}
			}
			const uint32_t index = grid_index<D>(is_hash, hashmap_size, resolution, local);
			// same operand rounding as the atomic path: weight rounded to half first (grid.h:254)
			const float wq = F == 1 ? weight : (float)to_half_rn(weight);
#pragma unroll
			for (uint32_t f = 0; f < F; ++f) lds_atomic_add_f32(&lds_table[index * F + f], wq * g[f]);
		}
	}
	__syncthreads();

	if constexpr (F == 1) {
		for (uint32_t e = threadIdx.x; e < hashmap_size; e += GRID_LDS_THREADS) {
			const float v = lds_table[e];
			if (v != 0.0f) atomic_add_f32((float*)grad + e, v);
		}
	} else {
		for (uint32_t e = threadIdx.x; e < hashmap_size * F / 2; e += GRID_LDS_THREADS) {
			const float a = lds_table[2 * e], b = lds_table[2 * e + 1];
			if (a != 0.0f || b != 0.0f) atomic_add_h2((half_t*)grad + 2 * (size_t)e, h2{(half_t)a, (half_t)b});
		}
	}
}

__global__ void k_grid_backward_input(uint32_t n_dims, uint32_t n_features, GridIO io, const half_t* __restrict__ dL_dy,
                                      const float* __restrict__ dy_dx, float* __restrict__ dL_dx, uint32_t dx_stride_i,
                                      uint32_t dx_stride_d) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= io.n) return;
	float result[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	for (uint32_t k = 0; k < n_features; ++k) {
		const float dl = (float)dL_dy[(size_t)k * io.stride_k + (size_t)i * io.stride_i];
		for (uint32_t d = 0; d < n_dims; ++d) {
			const float t = dl * dy_dx[((size_t)k * io.n + i) * n_dims + d];
			result[d] = result[d] + t;
		}
	}
	for (uint32_t d = 0; d < n_dims; ++d) dL_dx[(size_t)i * dx_stride_i + (size_t)d * dx_stride_d] = result[d];
}

template <uint32_t D>
__global__ void k_grid_indices(const GridMeta meta, const GridIO io, uint32_t* __restrict__ indices) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= io.n) return;
	const bool is_hash = meta.grid_type == (uint32_t)GridType::Hash;
	const bool smooth = meta.interp == (uint32_t)InterpolationType::Smoothstep;
	for (uint32_t level = 0; level < meta.n_levels; ++level) {
		const uint32_t hashmap_size = meta.offset[level + 1] - meta.offset[level];
		float pos[D], pd[D];
		uint32_t pos_grid[D];
		for (uint32_t d = 0; d < D; ++d) {
			pos_fract(io.positions[(size_t)i * io.pos_stride_i + (size_t)d * io.pos_stride_d], meta.scale[level], smooth, pos[d], pd[d], pos_grid[d]);
		}
		for (uint32_t idx = 0; idx < (1u << D); ++idx) {
			uint32_t local[D];
			for (uint32_t d = 0; d < D; ++d) local[d] = pos_grid[d] + ((idx >> d) & 1u);
			indices[((size_t)i * meta.n_levels + level) * (1u << D) + idx] = grid_index<D>(is_hash, hashmap_size, meta.resolution[level], local);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------

#define TCNN_GRID_DISPATCH_F(D_, MACRO)                                                             \
	switch (meta.n_feat) {                                                                          \
		case 1: MACRO(D_, 1); break;                                                                \
		case 2: MACRO(D_, 2); break;                                                                \
		case 4: MACRO(D_, 4); break;                                                                \
		case 8: MACRO(D_, 8); break;                                                                \
		default: throw std::runtime_error("GridEncoding: n_features_per_level must be 1, 2, 4, or 8."); \
	}

#define TCNN_GRID_DISPATCH(MACRO)                                                                \
	switch (meta.n_dims) {                                                                       \
		case 2: TCNN_GRID_DISPATCH_F(2, MACRO); break;                                           \
		case 3: TCNN_GRID_DISPATCH_F(3, MACRO); break;                                           \
		case 4: TCNN_GRID_DISPATCH_F(4, MACRO); break;                                           \
		default: throw std::runtime_error("GridEncoding: number of input dims must be 2, 3 or 4."); \
	}

void grid_forward(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* params, half_t* out, float* dy_dx) {
	if (io.n == 0) return;
	const uint32_t blocks = grid_n_blocks(meta.n_levels, io.n);
#define FWD(D_, F_) TCNN_LAUNCH((k_grid_forward<D_, F_>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, params, out, dy_dx)
	TCNN_GRID_DISPATCH(FWD)
#undef FWD
}

void grid_backward(hipStream_t stream, const GridMeta& meta, const GridIO& io, const half_t* dL_dy, half_t* grid_gradient,
                   float* grad_f32, uint32_t lds_level_budget_bytes) {
	if (io.n == 0) return;
	SmallLevels small = {};
	uint32_t max_lds_bytes = 0;
	for (uint32_t l = 0; l < meta.n_levels && small.count < 16; ++l) {
		const uint32_t bytes = (meta.offset[l + 1] - meta.offset[l]) * meta.n_feat * (uint32_t)sizeof(float);
		if (bytes <= lds_level_budget_bytes) {
			small.mask[l >> 5] |= 1u << (l & 31u);
			small.levels[small.count++] = l;
			if (bytes > max_lds_bytes) max_lds_bytes = bytes;
		}
	}
	const uint32_t blocks = grid_n_blocks(meta.n_levels, io.n);
	const uint32_t lds_blocks = small.count * div_round_up(io.n, GRID_LDS_SAMPLES_PER_BLOCK);
	if (meta.n_feat == 1) {
		if (!grad_f32) throw std::runtime_error("grid_backward: F == 1 needs the fp32 gradient accumulation buffer");
	} else if (!grid_gradient) {
		throw std::runtime_error("grid_backward: missing gradient buffer");
	}
#define BWD(D_, F_)                                                                                                              \
	{                                                                                                                            \
		using G = std::conditional_t<F_ == 1, float, half_t>;                                                                    \
		G* gp = (G*)(F_ == 1 ? (void*)grad_f32 : (void*)grid_gradient);                                                          \
		if (lds_blocks > 0) TCNN_SET_MAX_DYN_LDS((k_grid_backward_lds<D_, F_, G>), max_lds_bytes);                                \
		if (lds_blocks > 0)                                                                                                      \
			TCNN_LAUNCH((k_grid_backward_lds<D_, F_, G>), dim3(lds_blocks), dim3(GRID_LDS_THREADS), max_lds_bytes, stream, meta, \
			            io, small, dL_dy, gp);                                                                                   \
		if (small.count < meta.n_levels)                                                                                         \
			TCNN_LAUNCH((k_grid_backward<D_, F_, G>), dim3(blocks), dim3(GRID_THREADS), 0, stream, meta, io, small, dL_dy, gp);  \
	}
	TCNN_GRID_DISPATCH(BWD)
#undef BWD
}

void grid_backward_input(hipStream_t stream, uint32_t n_dims, uint32_t n_features, const GridIO& io, const half_t* dL_dy,
                         const float* dy_dx, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_d) {
	if (io.n == 0) return;
	TCNN_LAUNCH(k_grid_backward_input, dim3(div_round_up(io.n, 128u)), dim3(128), 0, stream, n_dims, n_features, io, dL_dy, dy_dx,
	            dL_dx, dx_stride_i, dx_stride_d);
}

void grid_indices(hipStream_t stream, const GridMeta& meta, const GridIO& io, uint32_t* indices) {
	if (io.n == 0) return;
	const uint32_t blocks = div_round_up(io.n, 128u);
	switch (meta.n_dims) {
		case 2: TCNN_LAUNCH((k_grid_indices<2>), dim3(blocks), dim3(128), 0, stream, meta, io, indices); break;
		case 3: TCNN_LAUNCH((k_grid_indices<3>), dim3(blocks), dim3(128), 0, stream, meta, io, indices); break;
		case 4: TCNN_LAUNCH((k_grid_indices<4>), dim3(blocks), dim3(128), 0, stream, meta, io, indices); break;
		default: throw std::runtime_error("GridEncoding: number of input dims must be 2, 3 or 4.");
	}
}

}  // namespace tcnn_hip
