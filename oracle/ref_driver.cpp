// ref_driver.cpp -- extern "C" face of the REFERENCE'S OWN kernels compiled for the host (oracle/build_ref.py).
// TEST INFRASTRUCTURE: nothing under tiny-cuda-nn_amd/ links, loads or calls this; tests/test_oracle_ref.py uses it to pin the
// restated oracle (oracle/tcnn_oracle.c) against the reference's arithmetic bit for bit.
//
// The kernel bodies come from /root/reference, where they lie (ref_extracted_kernels.inc is produced by build_ref.py in a temporary
// directory for the duration of the compile).  What is OURS here is only the launch emulation: for every kernel the grid / block
// shape the reference's host code uses (cited per wrapper), one host call per (blockIdx, threadIdx).  Threads run one after the
// other in ascending (blockIdx.y, blockIdx.x, threadIdx.x) order, so atomically accumulated fp16 sums are those of that order.
#include <tiny-cuda-nn/common.h>
// PTX inline assembly (lane_id(), common_device.h:38-42) cannot be parsed by a host compiler: `asm volatile(...)` -> `((void)0)`
#define asm
#define volatile(...) ((void)0)
#include <tiny-cuda-nn/common_device.h>
#undef asm
#undef volatile

#include "ref_extracted_kernels.inc"

#include <cmath>
#include <cstdint>
#include <stdexcept>

using namespace tcnn;

namespace {

template <typename F>
void launch(uint32_t blocks_x, uint32_t blocks_y, uint32_t threads, F&& body) {
	gridDim.x = blocks_x;
	gridDim.y = blocks_y;
	blockDim.x = threads;
	for (uint32_t by = 0; by < blocks_y; ++by) {
		for (uint32_t bx = 0; bx < blocks_x; ++bx) {
			for (uint32_t t = 0; t < threads; ++t) {
				blockIdx.x = bx;
				blockIdx.y = by;
				threadIdx.x = t;
				body();
			}
		}
	}
}
// linear_kernel (common_host.h:194-201): N_THREADS_LINEAR = 128 threads, n_blocks_linear(n) blocks
template <typename F>
void launch_linear(size_t n, F&& body) {
	if (n == 0) return;
	launch((uint32_t)((n + 127) / 128), 1, 128, body);
}

ParamsOffsetTable offset_table(uint32_t n_levels, const uint32_t* offsets) {
	ParamsOffsetTable t;
	for (uint32_t l = 0; l <= n_levels; ++l) t.data[l] = offsets[l];
	t.size = n_levels + 1;
	return t;
}

struct GridArgs {
	uint32_t n_dims, n_feat, n, n_levels;
	const uint32_t* offsets;
	uint32_t base_resolution;
	float log2_per_level_scale, max_level;
	int interpolation, grid_type, stochastic;
};

// GridEncodingTemplated::forward_impl, grid.h:773-801: blocks (ceil(n / 512), n_levels), 512 threads; positions as the
// reference's MatrixView over a column-major (n_dims x n) matrix: element (dim, i) at [dim + i * n_dims]
template <uint32_t D, uint32_t F>
void grid_forward(const GridArgs& a, const __half* grid, const float* positions, __half* encoded, float* dy_dx) {
	const ParamsOffsetTable table = offset_table(a.n_levels, a.offsets);
	MatrixView<const float> pos(positions, 1u, a.n_dims);
	launch((a.n + 511) / 512, a.n_levels, 512, [&] {
		kernel_grid<__half, D, F, HashType::CoherentPrime>(a.n, a.n_levels * F, table, a.base_resolution, a.log2_per_level_scale, a.max_level, nullptr,
		                                                   (InterpolationType)a.interpolation, (GridType)a.grid_type, grid, pos, encoded, dy_dx);
	});
}
// backward_impl, grid.h:853-895: grad_t = float for one feature per level, else T (:665); N_FEATURES_PER_THREAD = min(2, F);
// blocks (ceil(n * F / N_FEATURES_PER_THREAD / 256), n_levels), 256 threads
template <uint32_t D, uint32_t F>
void grid_backward(const GridArgs& a, const float* positions, const __half* dL_dy, void* grid_gradient) {
	const ParamsOffsetTable table = offset_table(a.n_levels, a.offsets);
	MatrixView<const float> pos(positions, 1u, a.n_dims);
	constexpr uint32_t FPT = F < 2 ? F : 2;
	using grad_t = std::conditional_t<F == 1, float, __half>;
	const uint32_t n_threads_total = a.n * F / FPT;
	launch((n_threads_total + 255) / 256, a.n_levels, 256, [&] {
		kernel_grid_backward<__half, grad_t, D, F, FPT, HashType::CoherentPrime>(a.n, a.n_levels * F, table, a.base_resolution, a.log2_per_level_scale, a.max_level, nullptr,
		                                                                        a.stochastic != 0, (InterpolationType)a.interpolation, (GridType)a.grid_type,
		                                                                        (grad_t*)grid_gradient, pos, dL_dy);
	});
}

// ---- the same kernels instantiated with T = float: GridEncodingTemplated<float>, what create_encoding(..., Precision::Fp32) builds
// (cpp_api.cu:165-168); grad_t = float as well (grid.h:665 only switches to float for F == 1 -- with T = float it is float anyway)
template <uint32_t D, uint32_t F>
void grid_forward_f32(const GridArgs& a, const float* grid, const float* positions, float* encoded, float* dy_dx) {
	const ParamsOffsetTable table = offset_table(a.n_levels, a.offsets);
	MatrixView<const float> pos(positions, 1u, a.n_dims);
	launch((a.n + 511) / 512, a.n_levels, 512, [&] {
		kernel_grid<float, D, F, HashType::CoherentPrime>(a.n, a.n_levels * F, table, a.base_resolution, a.log2_per_level_scale, a.max_level, nullptr,
		                                                  (InterpolationType)a.interpolation, (GridType)a.grid_type, grid, pos, encoded, dy_dx);
	});
}
template <uint32_t D, uint32_t F>
void grid_backward_f32(const GridArgs& a, const float* positions, const float* dL_dy, float* grid_gradient) {
	const ParamsOffsetTable table = offset_table(a.n_levels, a.offsets);
	MatrixView<const float> pos(positions, 1u, a.n_dims);
	constexpr uint32_t FPT = F < 2 ? F : 2;
	const uint32_t n_threads_total = a.n * F / FPT;
	launch((n_threads_total + 255) / 256, a.n_levels, 256, [&] {
		kernel_grid_backward<float, float, D, F, FPT, HashType::CoherentPrime>(a.n, a.n_levels * F, table, a.base_resolution, a.log2_per_level_scale, a.max_level, nullptr,
		                                                                      a.stochastic != 0, (InterpolationType)a.interpolation, (GridType)a.grid_type, grid_gradient, pos, dL_dy);
	});
}
// backward_backward_input_impl, grid.h:907-1042: dL_d(dL_dx) -> grid gradient (blocks (ceil(n * F / FPT / 256), n_levels), 256 threads;
// grad_t as in backward_impl) and -> dL_dx (same launch shape; dL_dx zeroed first, :1011-1016)
template <uint32_t D, uint32_t F>
void grid_backward_backward(const GridArgs& a, const float* dL_ddLdx, const float* positions, const __half* dL_dy, const __half* grid, void* grid_gradient, float* dL_dx) {
	const ParamsOffsetTable table = offset_table(a.n_levels, a.offsets);
	MatrixView<const float> pos(positions, 1u, a.n_dims), ddx(dL_ddLdx, 1u, a.n_dims);
	constexpr uint32_t FPT = F < 2 ? F : 2;
	using grad_t = std::conditional_t<F == 1, float, __half>;
	const uint32_t blocks = (a.n * F / FPT + 255) / 256;
	if (grid_gradient) {
		launch(blocks, a.n_levels, 256, [&] {
			kernel_grid_backward_input_backward_grid<__half, grad_t, D, F, FPT, HashType::CoherentPrime>(a.n, a.n_levels * F, table, a.base_resolution, a.log2_per_level_scale, a.max_level,
			                                                                                            nullptr, (InterpolationType)a.interpolation, (GridType)a.grid_type, ddx, pos, dL_dy,
			                                                                                            (grad_t*)grid_gradient);
		});
	}
	if (dL_dx) {
		for (uint32_t i = 0; i < a.n * a.n_dims; ++i) dL_dx[i] = 0.0f;
		MatrixView<float> out(dL_dx, 1u, a.n_dims);
		launch(blocks, a.n_levels, 256, [&] {
			kernel_grid_backward_input_backward_input<__half, D, F, FPT, HashType::CoherentPrime>(a.n, a.n_levels * F, table, a.base_resolution, a.log2_per_level_scale, a.max_level, nullptr,
			                                                                                     (InterpolationType)a.interpolation, (GridType)a.grid_type, ddx, pos, dL_dy, grid, out);
		});
	}
}

// the 2-D launches of the OneBlob encoding (oneblob.h:209-224, 250-262): threads (x, ceil(128 / x)), blockIdx.x over ceil(n * x / (threads.x * threads.y))
template <typename F>
void launch_xy(uint32_t n_work, uint32_t tx, F&& body) {
	const uint32_t ty = (128 + tx - 1) / tx, per_block = tx * ty, blocks = (n_work + per_block - 1) / per_block;
	gridDim.x = blocks;
	gridDim.y = 1;
	blockDim.x = tx;
	blockDim.y = ty;
	blockIdx.y = 0;
	for (uint32_t b = 0; b < blocks; ++b) {
		for (uint32_t y = 0; y < ty; ++y) {
			for (uint32_t x = 0; x < tx; ++x) {
				blockIdx.x = b;
				threadIdx.x = x;
				threadIdx.y = y;
				body();
			}
		}
	}
	threadIdx.y = 0;
	blockDim.y = 1;
}

template <typename Fn>
void dispatch_grid(uint32_t D, uint32_t F, Fn&& fn) {
#define CASE(D_, F_) if (D == D_ && F == F_) { fn(std::integral_constant<uint32_t, D_>{}, std::integral_constant<uint32_t, F_>{}); return; }
	CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(2, 8) CASE(3, 1) CASE(3, 2) CASE(3, 4) CASE(3, 8) CASE(4, 1) CASE(4, 2) CASE(4, 4) CASE(4, 8)
#undef CASE
	throw std::runtime_error("ref: unsupported (n_dims, n_features_per_level)");
}

struct Frag8 {  // what warp_activation sees of a wmma fragment (common_device.h:108-186): x[] and num_elements
	__half x[8];
	static constexpr int num_elements = 8;
};

}  // namespace

extern "C" {

// GridEncodingTemplated's constructor, grid.h:701: m_per_level_scale -> std::log2(per_level_scale), an fp32 on the host
float ref_log2_per_level_scale(float per_level_scale) { return std::log2(per_level_scale); }
// grid_scale / grid_resolution, common_device.h:886-895
void ref_grid_level(uint32_t level, float log2_per_level_scale, uint32_t base_resolution, float* scale, uint32_t* resolution) {
	*scale = grid_scale(level, log2_per_level_scale, base_resolution);
	*resolution = grid_resolution(*scale);
}
// grid_index<N_POS_DIMS, CoherentPrime>, common_device.h:855-884
uint32_t ref_grid_index(uint32_t n_dims, int grid_type, uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos) {
	switch (n_dims) {
		case 2: return grid_index<2, HashType::CoherentPrime>((GridType)grid_type, hashmap_size, resolution, uvec<2>{pos[0], pos[1]});
		case 3: return grid_index<3, HashType::CoherentPrime>((GridType)grid_type, hashmap_size, resolution, uvec<3>{pos[0], pos[1], pos[2]});
		case 4: return grid_index<4, HashType::CoherentPrime>((GridType)grid_type, hashmap_size, resolution, uvec<4>{pos[0], pos[1], pos[2], pos[3]});
	}
	return 0xFFFFFFFFu;
}
// pos_fract with the derivative, common_device.h:1016-1030 (Linear: identity_fun; Smoothstep: smoothstep)
void ref_pos_fract(float input, float scale, int smooth, float* pos, float* pos_derivative, uint32_t* pos_grid) {
	if (smooth) pos_fract(input, pos, pos_derivative, pos_grid, scale, smoothstep, smoothstep_derivative);
	else pos_fract(input, pos, pos_derivative, pos_grid, scale, identity_fun, identity_derivative);
}

int ref_grid_forward(uint32_t n_dims, uint32_t n_feat, uint32_t n, uint32_t n_levels, const uint32_t* offsets, uint32_t base_resolution,
                     float log2_per_level_scale, float max_level, int interpolation, int grid_type, const void* grid, const float* positions,
                     void* encoded, float* dy_dx) {
	try {
		GridArgs a = {n_dims, n_feat, n, n_levels, offsets, base_resolution, log2_per_level_scale, max_level, interpolation, grid_type, 0};
		dispatch_grid(n_dims, n_feat, [&](auto d, auto f) { grid_forward<decltype(d)::value, decltype(f)::value>(a, (const __half*)grid, positions, (__half*)encoded, dy_dx); });
	} catch (...) {
		return 1;
	}
	return 0;
}
// grid_gradient: fp16 [n_params] for n_feat >= 2, fp32 [n_params] for n_feat == 1 (grid.h:665); the caller zeroes it (grid.h:865-867)
int ref_grid_backward(uint32_t n_dims, uint32_t n_feat, uint32_t n, uint32_t n_levels, const uint32_t* offsets, uint32_t base_resolution,
                      float log2_per_level_scale, float max_level, int stochastic, int interpolation, int grid_type, const float* positions,
                      const void* dL_dy, void* grid_gradient) {
	try {
		GridArgs a = {n_dims, n_feat, n, n_levels, offsets, base_resolution, log2_per_level_scale, max_level, interpolation, grid_type, stochastic};
		dispatch_grid(n_dims, n_feat, [&](auto d, auto f) { grid_backward<decltype(d)::value, decltype(f)::value>(a, positions, (const __half*)dL_dy, grid_gradient); });
	} catch (...) {
		return 1;
	}
	return 0;
}
// ---- the same kernels with T = float (the templates are above, outside the extern "C" block)
int ref_grid_forward_f32(uint32_t n_dims, uint32_t n_feat, uint32_t n, uint32_t n_levels, const uint32_t* offsets, uint32_t base_resolution,
                         float log2_per_level_scale, float max_level, int interpolation, int grid_type, const float* grid, const float* positions,
                         float* encoded, float* dy_dx) {
	try {
		GridArgs a = {n_dims, n_feat, n, n_levels, offsets, base_resolution, log2_per_level_scale, max_level, interpolation, grid_type, 0};
		dispatch_grid(n_dims, n_feat, [&](auto d, auto f) { grid_forward_f32<decltype(d)::value, decltype(f)::value>(a, grid, positions, encoded, dy_dx); });
	} catch (...) {
		return 1;
	}
	return 0;
}
// grid_gradient: fp32 [n_params]; the caller zeroes it (grid.h:865-867)
int ref_grid_backward_f32(uint32_t n_dims, uint32_t n_feat, uint32_t n, uint32_t n_levels, const uint32_t* offsets, uint32_t base_resolution,
                          float log2_per_level_scale, float max_level, int interpolation, int grid_type, const float* positions, const float* dL_dy,
                          float* grid_gradient) {
	try {
		GridArgs a = {n_dims, n_feat, n, n_levels, offsets, base_resolution, log2_per_level_scale, max_level, interpolation, grid_type, 0};
		dispatch_grid(n_dims, n_feat, [&](auto d, auto f) { grid_backward_f32<decltype(d)::value, decltype(f)::value>(a, positions, dL_dy, grid_gradient); });
	} catch (...) {
		return 1;
	}
	return 0;
}
int ref_grid_backward_input_f32(uint32_t n_dims, uint32_t n, uint32_t n_features, const float* dL_dy, const float* dy_dx, float* dL_dx) {
	MatrixView<float> out(dL_dx, 1u, n_dims);
	switch (n_dims) {
		case 2: launch_linear(n, [&] { kernel_grid_backward_input<float, 2>(n, n_features, dL_dy, dy_dx, out); }); return 0;
		case 3: launch_linear(n, [&] { kernel_grid_backward_input<float, 3>(n, n_features, dL_dy, dy_dx, out); }); return 0;
		case 4: launch_linear(n, [&] { kernel_grid_backward_input<float, 4>(n, n_features, dL_dy, dy_dx, out); }); return 0;
	}
	return 1;
}
// kernel_grid_backward_input through linear_kernel (grid.h:897-905); dL_dx column-major (n_dims x n)
int ref_grid_backward_input(uint32_t n_dims, uint32_t n, uint32_t n_features, const void* dL_dy, const float* dy_dx, float* dL_dx) {
	MatrixView<float> out(dL_dx, 1u, n_dims);
	switch (n_dims) {
		case 2: launch_linear(n, [&] { kernel_grid_backward_input<__half, 2>(n, n_features, (const __half*)dL_dy, dy_dx, out); }); return 0;
		case 3: launch_linear(n, [&] { kernel_grid_backward_input<__half, 3>(n, n_features, (const __half*)dL_dy, dy_dx, out); }); return 0;
		case 4: launch_linear(n, [&] { kernel_grid_backward_input<__half, 4>(n, n_features, (const __half*)dL_dy, dy_dx, out); }); return 0;
	}
	return 1;
}

// kernel_grid_backward_input_backward_grid / _backward_input (grid.h:351-620): dL_ddLdx and positions column-major (n_dims x n), dL_dy
// feature-major [k][i], grid_gradient fp16 (fp32 for one feature per level) zeroed by the caller, dL_dx column-major; either output may be NULL
int ref_grid_backward_backward(uint32_t n_dims, uint32_t n_feat, uint32_t n, uint32_t n_levels, const uint32_t* offsets, uint32_t base_resolution,
                               float log2_per_level_scale, float max_level, int interpolation, int grid_type, const float* dL_ddLdx, const float* positions,
                               const void* dL_dy, const void* grid, void* grid_gradient, float* dL_dx) {
	try {
		GridArgs a = {n_dims, n_feat, n, n_levels, offsets, base_resolution, log2_per_level_scale, max_level, interpolation, grid_type, 0};
		dispatch_grid(n_dims, n_feat, [&](auto d, auto f) {
			grid_backward_backward<decltype(d)::value, decltype(f)::value>(a, dL_ddLdx, positions, (const __half*)dL_dy, (const __half*)grid, grid_gradient, dL_dx);
		});
	} catch (...) {
		return 1;
	}
	return 0;
}
// kernel_grid_backward_input_backward_dLdoutput through linear_kernel (grid.h:622-653, 994-1006); dL_ddLdy column-major ((n_features + n_to_pad) x n)
int ref_grid_backward_backward_dLdoutput(uint32_t n_dims, uint32_t n, uint32_t n_features, uint32_t n_to_pad, const float* dL_ddLdx, const float* dy_dx, const void* dL_dy,
                                         void* dL_ddLdy) {
	MatrixView<const float> ddx(dL_ddLdx, 1u, n_dims);
	MatrixView<__half> out((__half*)dL_ddLdy, 1u, n_features + n_to_pad);
	switch (n_dims) {
		case 2: launch_linear(n, [&] { kernel_grid_backward_input_backward_dLdoutput<__half, 2>(n, n_features, n_to_pad, ddx, dy_dx, (const __half*)dL_dy, out); }); return 0;
		case 3: launch_linear(n, [&] { kernel_grid_backward_input_backward_dLdoutput<__half, 3>(n, n_features, n_to_pad, ddx, dy_dx, (const __half*)dL_dy, out); }); return 0;
		case 4: launch_linear(n, [&] { kernel_grid_backward_input_backward_dLdoutput<__half, 4>(n, n_features, n_to_pad, ddx, dy_dx, (const __half*)dL_dy, out); }); return 0;
	}
	return 1;
}

// frequency encoding, frequency.h:45-105 (linear_kernel over n * padded outputs / n * n_dims inputs); in / out / dL_dy / dL_dx column-major,
// dy_dx [i][n_dims * n_frequencies * 2].  __sinf / __cosf are approximations on the device; this host build evaluates sinf / cosf
void ref_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t n_to_pad, const float* in, void* out, float* dy_dx) {
	const uint32_t fan_out = n_dims * n_frequencies * 2 + n_to_pad;
	MatrixView<const float> vin(in, 1u, n_dims);
	MatrixView<__half> vout((__half*)out, 1u, fan_out);
	launch_linear((size_t)n * fan_out, [&] { frequency_encoding<__half>(n * fan_out, n_frequencies, n_dims, n_to_pad, vin, vout, dy_dx); });
}
void ref_frequency_backward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const void* dL_dy, const float* dy_dx, float* dL_dx) {
	MatrixView<const __half> vdy((const __half*)dL_dy, 1u, padded);
	MatrixView<float> vdx(dL_dx, 1u, n_dims);
	launch_linear((size_t)n * n_dims, [&] { frequency_encoding_backward<__half>(n * n_dims, n_dims, n_frequencies, vdy, dy_dx, vdx); });
}

// OneBlob, the structure-of-arrays forward kernel and the backward kernel (oneblob.h:98-164; launches :209-224, :250-262).  `out`: feature-major
// [n_dims * n_bins][n]; dL_dy column-major (padded x n); in / dL_dx column-major (n_dims x n).  (The array-of-structures forward kernel
// gets a bin's right boundary from the neighbouring lane with __shfl_sync, oneblob.h:46-67: not compiled.)
void ref_oneblob_forward_soa(uint32_t n, uint32_t n_dims, uint32_t num_bins_log2, const float* in, void* out) {
	MatrixView<const float> vin(in, 1u, n_dims);
	launch_xy(n * n_dims, n_dims, [&] { kernel_one_blob_soa<__half>(n, num_bins_log2, n_dims, vin, (__half*)out); });
}
void ref_oneblob_backward(uint32_t n, uint32_t n_dims, uint32_t num_bins_log2, uint32_t padded, const void* dL_dy, const float* in, float* dL_dx) {
	MatrixView<const __half> vdy((const __half*)dL_dy, 1u, padded);
	MatrixView<const float> vin(in, 1u, n_dims);
	MatrixView<float> vdx(dL_dx, 1u, n_dims);
	launch_xy(n * n_dims, n_dims, [&] { kernel_one_blob_backward<__half>(n, n_dims, num_bins_log2, vdy, vin, vdx); });
}

// AdamOptimizer::step, adam.h:158-198 (linear_kernel over n_weights)
void ref_adam_step(uint32_t n_elements, uint32_t n_matrix_weights, float relative_weight_decay, float absolute_weight_decay, float weight_clipping_magnitude,
                   float gradient_clipping_magnitude, float loss_scale, float learning_rate, float non_matrix_learning_rate_factor, int optimize_matrix_params,
                   int optimize_non_matrix_params, int skip_zero_grad_non_matrix_params, float beta1, float beta2, float epsilon, float lower_lr_bound, float upper_lr_bound,
                   float l2_reg, float non_matrix_l2_reg, float* weights_full_precision, void* weights, const void* gradients, float* first_moments,
                   float* second_moments, uint32_t* param_steps) {
	launch_linear(n_elements, [&] {
		adam_step<__half>(n_elements, n_matrix_weights, relative_weight_decay, absolute_weight_decay, weight_clipping_magnitude, gradient_clipping_magnitude, loss_scale,
		                  learning_rate, non_matrix_learning_rate_factor, optimize_matrix_params != 0, optimize_non_matrix_params != 0,
		                  skip_zero_grad_non_matrix_params != 0, beta1, beta2, epsilon, lower_lr_bound, upper_lr_bound, l2_reg, non_matrix_l2_reg,
		                  weights_full_precision, (__half*)weights, (const __half*)gradients, first_moments, second_moments, param_steps);
	});
}

// EmaOptimizer::step after the nested step, ema.h:103-136: the debias factors are computed there on the host (std::pow(float, unsigned) in
// double, narrowed to float); tmp != NULL selects the full-precision variant (fp32 running average next to the half one)
void ref_ema_step(uint32_t n_elements, float ema_decay, uint32_t current_step, const void* weights, void* weights_ema, float* tmp) {
	const float ema_debias_old = 1 - (float)std::pow(ema_decay, current_step - 1);
	const float ema_debias_new = 1.0f / (1 - (float)std::pow(ema_decay, current_step));
	if (tmp) {
		launch_linear(n_elements, [&] { ema_step_full_precision<__half>(n_elements, ema_decay, ema_debias_old, ema_debias_new, (const __half*)weights, (__half*)weights_ema, tmp); });
	} else {
		launch_linear(n_elements, [&] { ema_step_half_precision<__half>(n_elements, ema_decay, ema_debias_old, ema_debias_new, (const __half*)weights, (__half*)weights_ema); });
	}
}

// Loss::evaluate of the element-wise losses (e.g. relative_l2.h:92-106): linear_kernel over n_elements = batch * stride
// which: 0 L2, 1 RelativeL2, 2 L1, 3 RelativeL1, 4 Mape, 5 Smape, 6 RelativeL2Luminance, 7 CrossEntropy, 8 Variance
int ref_loss(int which, uint32_t n_elements, uint32_t stride, uint32_t dims, float loss_scale, const void* predictions, const float* targets, float* values,
             void* gradients, const float* data_pdf) {
	const __half* p = (const __half*)predictions;
	__half* g = (__half*)gradients;
	switch (which) {
		case 0: launch_linear(n_elements, [&] { l2_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 1: launch_linear(n_elements, [&] { relative_l2_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 2: launch_linear(n_elements, [&] { l1_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 3: launch_linear(n_elements, [&] { relative_l1_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 4: launch_linear(n_elements, [&] { mape_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 5: launch_linear(n_elements, [&] { smape_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 6: launch_linear(n_elements, [&] { relative_l2_luminance_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 7: launch_linear(n_elements, [&] { cross_entropy_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
		case 8: launch_linear(n_elements, [&] { variance_is_loss<__half>(n_elements, stride, dims, loss_scale, p, targets, values, g, data_pdf); }); return 0;
	}
	return 1;
}

// generate_random_uniform<float>(rng, n, out, lower, upper), random.h:53-69: N_TO_GENERATE = 4, linear_kernel over ceil(n / 4) threads,
// transform val * (upper - lower) + lower; afterwards rng.advance(n).  `position`: draws consumed before the call.
void ref_generate_random_uniform(uint64_t seed, uint64_t position, size_t n, float* out, float lower, float upper) {
	pcg32 rng{seed};
	rng.advance((int64_t)position);
	const size_t n_threads = (n + 3) / 4;
	auto transform = [upper, lower](float val) { return val * (upper - lower) + lower; };
	launch_linear(n_threads, [&] { generate_random_kernel<float, pcg32, 4>(n, rng, out, transform); });
}

// warp_activation / warp_activation_backward on 8 values (common_device.h:108-186, 363-440): the activation arithmetic of the fused
// network kernels.  activation: the reference's enum Activation (common.h:134-145).  Returns 1 for an unknown activation.
int ref_activation(int activation, int backward, uint32_t n, const void* in, const void* forward_value, void* out) {
	const __half* x = (const __half*)in;
	const __half* fwd = (const __half*)forward_value;
	__half* y = (__half*)out;
	for (uint32_t base = 0; base < n; base += 8) {
		Frag8 a, f, r;
		for (uint32_t k = 0; k < 8; ++k) {
			a.x[k] = base + k < n ? x[base + k] : __half(0.0f);
			f.x[k] = backward && base + k < n ? fwd[base + k] : __half(0.0f);
		}
		if (backward) warp_activation_backward<__half>((Activation)activation, a, f, r);
		else warp_activation<__half>((Activation)activation, a, r);
		for (uint32_t k = 0; k < 8 && base + k < n; ++k) y[base + k] = r.x[k];
	}
	return 0;
}

// identity encoding, identity.h:46-90 (linear_kernel over n * padded outputs / n * n_dims inputs)
void ref_identity_forward(uint32_t n, uint32_t n_dims, uint32_t n_to_pad, float scale, float offset, const float* in, void* out) {
	const uint32_t fan_out = n_dims + n_to_pad;
	MatrixView<const float> vin(in, 1u, n_dims);
	MatrixView<__half> vout((__half*)out, 1u, fan_out);
	launch_linear((size_t)n * fan_out, [&] { identity<__half>(n * fan_out, n_dims, n_to_pad, scale, offset, vin, vout); });
}
void ref_identity_backward(uint32_t n, uint32_t n_dims, uint32_t padded, float scale, const void* dL_dy, float* dL_dx) {
	MatrixView<const __half> vdy((const __half*)dL_dy, 1u, padded);
	MatrixView<float> vdx(dL_dx, 1u, n_dims);
	launch_linear((size_t)n * n_dims, [&] { identity_backward<__half>(n * n_dims, n_dims, scale, vdy, vdx); });
}

}  // extern "C"
