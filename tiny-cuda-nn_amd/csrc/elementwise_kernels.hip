// elementwise_kernels.hip -- see elementwise_kernels.h.  All kernels here are pure HBM streams:
// 16-byte accesses per lane, grid-stride where it matters, no host synchronisation.
// Build with -ffp-contract=off (loss / Adam arithmetic is checked against the CPU oracle).
#include "elementwise_kernels.h"
#include <type_traits>
#include "adam_device.h"

#include <cmath>

#include <stdexcept>

namespace tcnn_hip {

constexpr uint32_t EW_THREADS = 256;

// ------------------------------------------------------------------------------------------ rng
// pcg32's skip-ahead by 2^k draws, k = 0 .. 39, as affine maps state -> mult[k] * state + plus[k] (the squarings of Pcg32::advance, done once on
// the host for the stream's increment).  A thread that has to skip 4 i draws applies the maps of the set bits of 4 i: one 64-bit multiply per
// set bit instead of the four per bit POSITION of the generic loop -- the same state, bit for bit (the maps are powers of one affine map and
// commute), at a quarter of the quarter-rate multiplies this kernel consists of (it is what a benchmark step that draws its batch pays first).
struct Pcg32Skip {
	uint64_t mult[40], plus[40];
};
static Pcg32Skip make_pcg32_skip(const Pcg32& rng) {
	Pcg32Skip t;
	uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = rng.inc;
	for (int k = 0; k < 40; ++k) {
		t.mult[k] = cur_mult;
		t.plus[k] = cur_plus;
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
	}
	return t;
}
__global__ void k_generate_random_uniform(size_t n_elements, Pcg32 rng, const Pcg32Skip skip, float* __restrict__ out, float lower, float range) {
	const size_t i = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
	const size_t n_threads = (size_t)blockDim.x * gridDim.x;
	{
		uint64_t delta = (uint64_t)i * 4u;  // < 2^40: the host checked the element count
		for (uint32_t k = 0; delta != 0; ++k, delta >>= 1) {
			if (delta & 1u) rng.state = skip.mult[k] * rng.state + skip.plus[k];
		}
	}
#pragma unroll
	for (size_t j = 0; j < 4; ++j) {
		const size_t idx = i + n_threads * j;
		if (idx >= n_elements) return;
		out[idx] = __builtin_fmaf(rng.next_float(), range, lower);
	}
}

// The same element <-> draw mapping (element i + n_threads * j is draw 4 i + j of the stream, n_threads the reference's launch width), four i per
// thread: one skip-ahead per 16 consecutive draws instead of per 4, and the four elements of one j are adjacent, so they leave as one 16-byte store.
__global__ void k_generate_random_uniform_x4(size_t n_elements, size_t n_threads, Pcg32 rng, const Pcg32Skip skip, float* __restrict__ out, float lower, float range) {
	const size_t i = (threadIdx.x + (size_t)blockIdx.x * blockDim.x) * 4;
	if (i >= n_threads) return;
	{
		uint64_t delta = (uint64_t)i * 4u;
		for (uint32_t k = 0; delta != 0; ++k, delta >>= 1) {
			if (delta & 1u) rng.state = skip.mult[k] * rng.state + skip.plus[k];
		}
	}
	float draw[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) draw[k] = __builtin_fmaf(rng.next_float(), range, lower);
#pragma unroll
	for (size_t j = 0; j < 4; ++j) {
		const size_t idx = i + n_threads * j;
		if (idx + 3 < n_elements) {
			*(f4*)(out + idx) = f4{draw[j], draw[4 + j], draw[8 + j], draw[12 + j]};
		} else {
			for (size_t a = 0; a < 4 && idx + a < n_elements; ++a) out[idx + a] = draw[4 * a + j];
		}
	}
}

void generate_random_uniform(hipStream_t stream, Pcg32& rng, size_t n, float* out, float lower, float upper) {
	if (n > 0) {
		const size_t n_threads = div_round_up(n, (size_t)4);
		const uint32_t blocks = (uint32_t)div_round_up(n_threads, (size_t)128);  // N_THREADS_LINEAR = 128 (common.h:247)
		if (n >= (1ull << 40)) throw std::runtime_error("generate_random_uniform: more than 2^40 elements");
		if (((uintptr_t)out & 15u) == 0) {
			const size_t width = (size_t)blocks * 128;  // the launch width the mapping is defined on
			TCNN_LAUNCH(k_generate_random_uniform_x4, dim3((uint32_t)div_round_up(width / 4, (size_t)64)), dim3(64), 0, stream, n, width, rng, make_pcg32_skip(rng), out, lower,
			            upper - lower);
		} else {
			TCNN_LAUNCH(k_generate_random_uniform, dim3(blocks), dim3(128), 0, stream, n, rng, make_pcg32_skip(rng), out, lower, upper - lower);
		}
	}
	rng.advance((int64_t)n);
}

// ------------------------------------------------------------------------------------------ casts
__global__ void k_cast_f32_to_f16(size_t n, const float* __restrict__ in, half_t* __restrict__ out) {
	const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (i4 + 3 < n) {
		const f4 v = *(const f4*)(in + i4);
		*(h4*)(out + i4) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
	} else {
		for (size_t i = i4; i < n; ++i) out[i] = (half_t)in[i];
	}
}
__global__ void k_cast_f16_to_f32(size_t n, const half_t* __restrict__ in, float* __restrict__ out) {
	const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (i4 + 3 < n) {
		const h4 v = *(const h4*)(in + i4);
		*(f4*)(out + i4) = f4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
	} else {
		for (size_t i = i4; i < n; ++i) out[i] = (float)in[i];
	}
}
// master[i] := (float)half[i] wherever the 16-bit value is no longer the rounded master weight (tcnn_trainer_params_written: a host
// wrote the 16-bit parameters directly); untouched parameters keep the master's extra bits
__global__ void k_resync_master_from_half(size_t n, const half_t* __restrict__ in, float* __restrict__ master) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const half_t h = in[i];
	if (__builtin_bit_cast(uint16_t, to_half_rn(master[i])) != __builtin_bit_cast(uint16_t, h)) master[i] = (float)h;
}
// fp32 <-> 16-bit with a power-of-two scale (the fp32 encodings of cpp_api.cu:165-174 computed in the 16-bit type: the
// gradients entering the backward pass are scaled into its range and the results scaled back, exactly)
__global__ void k_cast_scaled_f32_to_f16(size_t n, const float* __restrict__ in, half_t* __restrict__ out, float scale) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = to_half_rn(in[i] * scale);
}
__global__ void k_cast_scaled_f16_to_f32(size_t n, const half_t* __restrict__ in, float* __restrict__ out, float scale) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = (float)in[i] * scale;
}
__global__ void k_scale_f32(size_t n, float* __restrict__ data, float scale) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) data[i] = data[i] * scale;
}
__global__ void k_fill_f16(size_t n, half_t* __restrict__ out, float value) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = (half_t)value;
}

void cast_f32_to_f16(hipStream_t stream, size_t n, const float* in, half_t* out) {
	if (n == 0) return;
	TCNN_LAUNCH(k_cast_f32_to_f16, dim3((uint32_t)div_round_up(div_round_up(n, (size_t)4), (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, out);
}
// Regression targets of the synthetic benchmark workloads, evaluated on the device inside every timed step the way the reference's
// sample evaluates its image (samples/mlp_learning_an_image.cu:263-271: draw positions, `eval_image` them): a smooth analytic
// n_in-D -> n_out function, products of sinusoids of frequencies 1..4.  One thread per (sample, output).
__global__ void k_sinusoid_targets(uint32_t n, uint32_t n_in, uint32_t n_out, const float* __restrict__ positions, float* __restrict__ targets) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n * n_out) return;
	const uint32_t i = e / n_out, c = e % n_out;
	const float* x = positions + (size_t)i * n_in;
	const float two_pi = 6.28318530717958647692f, f = (float)(c % 4u + 1u);
	targets[e] = 0.5f + 0.5f * __builtin_sinf(two_pi * f * x[0]) * __builtin_cosf(two_pi * f * x[1u % n_in]) * __builtin_sinf(two_pi * x[2u % n_in] + (float)c);
}
void sinusoid_targets(hipStream_t stream, uint32_t n, uint32_t n_in, uint32_t n_out, const float* positions, float* targets) {
	if (n == 0 || n_out == 0) return;
	TCNN_LAUNCH(k_sinusoid_targets, dim3(div_round_up(n * n_out, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_in, n_out, positions, targets);
}
void resync_master_from_half(hipStream_t stream, size_t n, const half_t* in, float* master) {
	if (n == 0) return;
	TCNN_LAUNCH(k_resync_master_from_half, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, master);
}
void cast_f16_to_f32(hipStream_t stream, size_t n, const half_t* in, float* out) {
	if (n == 0) return;
	TCNN_LAUNCH(k_cast_f16_to_f32, dim3((uint32_t)div_round_up(div_round_up(n, (size_t)4), (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, out);
}
// ---- fp32 gradients entering the 16-bit kernels (fp32 encodings, cpp_api.cu:165-174): a per-call power-of-two scale from max |x|, computed
// and consumed on the device (no host round trip).  pair = {scale, 1 / scale}; word 2 of the buffer is the running maximum's bit pattern.
__global__ void k_absmax_f32(size_t n, const float* __restrict__ in, uint32_t* __restrict__ absmax_bits) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t b = 0;
	if (i < n) {
		b = __builtin_bit_cast(uint32_t, in[i]) & 0x7FFFFFFFu;
		if (b > 0x7F800000u) b = 0;  // NaN: leaves the scale alone (it propagates as NaN through the cast)
	}
	// wave maximum first: one atomic per wavefront
	for (uint32_t d = 32; d > 0; d >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, (int)d, 64));
	if ((threadIdx.x & 63u) == 0 && b) atomicMax(absmax_bits, b);
}
__global__ void k_gradient_scale(const uint32_t* __restrict__ absmax_bits, float* __restrict__ pair, float cap, float target) {
	const float m = __builtin_bit_cast(float, absmax_bits[0]);
	float scale = cap;
	if (m > 0.0f && m < __builtin_inff()) {
		int e;
		(void)__builtin_frexpf(target / m, &e);       // target / m = f * 2^e, f in [0.5, 1)
		scale = __builtin_fminf(__builtin_scalbnf(1.0f, e - 1), cap);  // the largest power of two <= target / m, capped
		scale = __builtin_fmaxf(scale, 1.0f / 16777216.0f);
	}
	pair[0] = scale;
	pair[1] = 1.0f / scale;
}
__global__ void k_cast_scaled_f32_to_f16_dev(size_t n, const float* __restrict__ in, half_t* __restrict__ out, const float* __restrict__ scale) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = to_half_rn(in[i] * scale[0]);
}
__global__ void k_cast_scaled_f16_to_f32_dev(size_t n, const half_t* __restrict__ in, float* __restrict__ out, const float* __restrict__ scale) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = (float)in[i] * scale[0];
}
__global__ void k_scale_f32_dev(size_t n, float* __restrict__ data, const float* __restrict__ scale) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) data[i] = data[i] * scale[0];
}
void gradient_scale_from_absmax(hipStream_t stream, size_t n, const float* in, float* pair_and_scratch, float cap, float target) {
	uint32_t* bits = (uint32_t*)(pair_and_scratch + 2);
	if (hipMemsetAsync(bits, 0, sizeof(uint32_t), stream) != hipSuccess) throw std::runtime_error("gradient_scale_from_absmax: memset failed");
	if (n) TCNN_LAUNCH(k_absmax_f32, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, bits);
	TCNN_LAUNCH(k_gradient_scale, dim3(1), dim3(1), 0, stream, (const uint32_t*)bits, pair_and_scratch, cap, target);
}
void cast_scaled_f32_to_f16(hipStream_t stream, size_t n, const float* in, half_t* out, const float* scale_dev) {
	if (n == 0) return;
	TCNN_LAUNCH(k_cast_scaled_f32_to_f16_dev, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, out, scale_dev);
}
void cast_scaled_f16_to_f32(hipStream_t stream, size_t n, const half_t* in, float* out, const float* scale_dev) {
	if (n == 0) return;
	TCNN_LAUNCH(k_cast_scaled_f16_to_f32_dev, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, out, scale_dev);
}
void scale_f32(hipStream_t stream, size_t n, float* data, const float* scale_dev) {
	if (n == 0) return;
	TCNN_LAUNCH(k_scale_f32_dev, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, data, scale_dev);
}
void cast_scaled_f32_to_f16(hipStream_t stream, size_t n, const float* in, half_t* out, float scale) {
	if (n == 0) return;
	TCNN_LAUNCH(k_cast_scaled_f32_to_f16, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, out, scale);
}
void cast_scaled_f16_to_f32(hipStream_t stream, size_t n, const half_t* in, float* out, float scale) {
	if (n == 0) return;
	TCNN_LAUNCH(k_cast_scaled_f16_to_f32, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, in, out, scale);
}
void scale_f32(hipStream_t stream, size_t n, float* data, float scale) {
	if (n == 0) return;
	TCNN_LAUNCH(k_scale_f32, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, data, scale);
}
void fill_f16(hipStream_t stream, size_t n, half_t* out, float value) {
	if (n == 0) return;
	TCNN_LAUNCH(k_fill_f16, dim3((uint32_t)div_round_up(n, (size_t)EW_THREADS)), dim3(EW_THREADS), 0, stream, n, out, value);
}

__global__ void k_trim_and_cast(uint32_t n, uint32_t padded, uint32_t dims, const half_t* __restrict__ in, float* __restrict__ out,
                                uint32_t stride_i, uint32_t stride_j) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n * dims) return;
	const uint32_t i = e / dims, j = e - i * dims;
	out[(size_t)i * stride_i + (size_t)j * stride_j] = (float)in[(size_t)i * padded + j];
}
void trim_and_cast(hipStream_t stream, uint32_t n, uint32_t padded, uint32_t dims, const half_t* in, float* out, uint32_t stride_i,
                   uint32_t stride_j) {
	if (n == 0) return;
	TCNN_LAUNCH(k_trim_and_cast, dim3(div_round_up(n * dims, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, padded, dims, in, out, stride_i, stride_j);
}

// ------------------------------------------------------------------------------------------ loss
// One thread per 8 consecutive elements of the [n][stride] half prediction matrix (16-byte accesses).
__global__ void __launch_bounds__(EW_THREADS) k_loss(const LossType type, uint32_t n_groups, uint32_t stride, uint32_t dims, float loss_scale,
                                                      const half_t* __restrict__ predictions, const float* __restrict__ targets,
                                                      const float* __restrict__ data_pdf, float* __restrict__ values,
                                                      half_t* __restrict__ gradients, float* __restrict__ block_sums, uint32_t n_total_u) {
	__shared__ float red[EW_THREADS];
	const uint32_t gidx = blockIdx.x * EW_THREADS + threadIdx.x;
	float local_sum = 0.0f;
	if (gidx < n_groups) {
		const uint32_t e0 = gidx * 8;
		const uint32_t inter = e0 / stride, intra0 = e0 - inter * stride;
		const h8 p8 = *(const h8*)(predictions + e0);
		h8 g8;
		float v8[8];
		const float n_total = (float)n_total_u;
		float luminance = 0.0f;
		if (type == LossType::RelativeL2Luminance) {  // relative_l2_luminance.h:68-76: from the first 3 (dims >= 6: 3 + 3) outputs of the row
			const h8 row = *(const h8*)(predictions + (size_t)inter * stride);
			float r = (float)row[0], g = (float)row[1], b = (float)row[2];
			if (dims >= 6) {
				r += (float)row[3];
				g += (float)row[4];
				b += (float)row[5];
			}
			luminance = loss_row_luminance(r, g, b);
		}
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) {
			const uint32_t intra = intra0 + j;
			if (intra >= dims) {  // relative_l2.h:57-61
				v8[j] = 0.0f;
				g8[j] = (half_t)0.0f;
				continue;
			}
			const uint32_t target_idx = inter * dims + intra;
			const float prediction = (float)p8[j];
			const float pdf = data_pdf ? data_pdf[target_idx] : 1.0f;
			float value;
			g8[j] = type == LossType::RelativeL2Luminance ? loss_element_luminance(prediction, luminance, targets[target_idx], pdf, n_total, loss_scale, value)
			                                              : loss_element(type, prediction, targets[target_idx], pdf, n_total, loss_scale, value);
			v8[j] = value;
			local_sum += value;
		}
		*(h8*)(gradients + e0) = g8;
		if (values) {
			*(f4*)(values + e0) = f4{v8[0], v8[1], v8[2], v8[3]};
			*(f4*)(values + e0 + 4) = f4{v8[4], v8[5], v8[6], v8[7]};
		}
	}
	if (block_sums) {
		red[threadIdx.x] = local_sum;
		__syncthreads();
		for (uint32_t s = EW_THREADS / 2; s > 0; s >>= 1) {
			if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
			__syncthreads();
		}
		if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
	}
}

uint32_t loss_n_blocks(uint32_t n, uint32_t stride) { return div_round_up(n * stride / 8u, EW_THREADS); }

void loss_evaluate(hipStream_t stream, LossType type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale,
                   const half_t* prediction, const float* target, const float* data_pdf, float* values, half_t* gradients,
                   float* block_sums, uint32_t n_total) {
	if (n == 0) return;
	if (stride % 8 != 0) throw std::runtime_error("loss: padded output width must be a multiple of 8");
	if (type == LossType::RelativeL2Luminance && dims < 3) throw std::runtime_error("RelativeL2Luminance needs at least 3 output dimensions");
	const uint32_t n_groups = n * stride / 8u;
	const uint32_t blocks = div_round_up(n_groups, EW_THREADS);
	TCNN_LAUNCH(k_loss, dim3(blocks), dim3(EW_THREADS), 0, stream, type, n_groups, stride, dims, loss_scale, prediction, target, data_pdf, values, gradients,
	            block_sums, n_total);
}

// ------------------------------------------------------------------------------------------ reduce
__global__ void __launch_bounds__(EW_THREADS) k_reduce_partial(const float* __restrict__ in, size_t n, float* __restrict__ out) {
	__shared__ float red[EW_THREADS];
	float s = 0.0f;
	for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * EW_THREADS) s += in[i];
	red[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t k = EW_THREADS / 2; k > 0; k >>= 1) {
		if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
		__syncthreads();
	}
	if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

void reduce_sum(hipStream_t stream, const float* in, size_t n, float* workspace, float* out) {
	const uint32_t blocks = (uint32_t)(n == 0 ? 1 : (div_round_up(n, (size_t)EW_THREADS) < 1024 ? div_round_up(n, (size_t)EW_THREADS) : 1024));
	TCNN_LAUNCH(k_reduce_partial, dim3(blocks), dim3(EW_THREADS), 0, stream, in, n, workspace);
	TCNN_LAUNCH(k_reduce_partial, dim3(1), dim3(EW_THREADS), 0, stream, (const float*)workspace, (size_t)blocks, out);
}

// ------------------------------------------------------------------------------------------ EMA of the weights
// optimizers/ema.h:45-81: debiased exponential moving average of the fp16 weights, kept in fp16 (tmp == nullptr) or
// in an fp32 shadow (full_precision).  8 weights per lane.
__global__ void __launch_bounds__(EW_THREADS) k_ema_step(uint32_t n, float ema_decay, float ema_debias_old, float ema_debias_new,
                                                          const half_t* __restrict__ weights, half_t* __restrict__ weights_ema, float* __restrict__ tmp) {
	const uint32_t i0 = (blockIdx.x * EW_THREADS + threadIdx.x) * 8u;
	if (i0 >= n) return;
	if (i0 + 8u <= n) {
		const h8 w = *(const h8*)(weights + i0);
		h8 e = *(const h8*)(weights_ema + i0);
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) {
			const float old = tmp ? tmp[i0 + j] : (float)e[j];
			const float filtered = (old * ema_decay * ema_debias_old + (float)w[j] * (1 - ema_decay)) * ema_debias_new;
			if (tmp) tmp[i0 + j] = filtered;
			e[j] = (half_t)filtered;
		}
		*(h8*)(weights_ema + i0) = e;
	} else {
		for (uint32_t i = i0; i < n; ++i) {
			const float old = tmp ? tmp[i] : (float)weights_ema[i];
			const float filtered = (old * ema_decay * ema_debias_old + (float)weights[i] * (1 - ema_decay)) * ema_debias_new;
			if (tmp) tmp[i] = filtered;
			weights_ema[i] = (half_t)filtered;
		}
	}
}

void ema_step(hipStream_t stream, uint32_t n, float ema_decay, uint32_t current_step, const half_t* weights, half_t* weights_ema, float* tmp,
              uint32_t begin, uint32_t end) {
	if (end > n) end = n;
	if (begin >= end) return;
	if (begin % 8u != 0u) throw std::runtime_error("ema_step: a parameter range must start at a multiple of 8");
	weights += begin;
	weights_ema += begin;
	if (tmp) tmp += begin;
	n = end - begin;
	// ema.h:113-114 (float pow through double, as std::pow(float, unsigned) does)
	const float ema_debias_old = 1 - (float)std::pow((double)ema_decay, (double)(current_step - 1));
	const float ema_debias_new = 1.0f / (1 - (float)std::pow((double)ema_decay, (double)current_step));
	TCNN_LAUNCH(k_ema_step, dim3(div_round_up(div_round_up(n, 8u), EW_THREADS)), dim3(EW_THREADS), 0, stream, n, ema_decay, ema_debias_old,
	            ema_debias_new, weights, weights_ema, tmp);
}

// ------------------------------------------------------------------------------------------ Adam
struct AdamArgs : AdamCore {
	uint32_t begin, n_elements;  // parameters [begin, n_elements) are stepped (begin % 4 == 0)
	MlpMeta mlp;                 // layout of the matrix weights (for weights_t)
};

// Byte form of the step-counter deficits (AdamCore::deficit == 2): deficits8[p] = steps_done - counter[p] while that is < 255; 255
// means "the counter itself is in param_steps[p]" (a parameter that has been skipped 255 times since it last fitted a byte stays in
// counter form).  A parameter that is stepped every time then costs ONE byte of step bookkeeping per step instead of four.
// count-before-this-step of parameter p with deficit byte d:
TCNN_DEVICE uint32_t adam_count8(const AdamCore& a, uint32_t d, const uint32_t* param_steps, uint32_t p) { return d == 255u ? param_steps[p] : a.steps_done - d; }
// the byte after this step: stepped -> unchanged (a saturated one writes its new count); skipped -> one more missed step
TCNN_DEVICE uint32_t adam_next8(const AdamCore& a, uint32_t d, bool stepped, uint32_t new_count, uint32_t* param_steps, uint32_t p) {
	if (stepped) {
		if (d == 255u) param_steps[p] = new_count;
		return d;
	}
	if (d == 254u) param_steps[p] = a.steps_done - 254u;  // leaves the byte's range: its counter from now on
	return d < 255u ? d + 1u : 255u;
}

// one parameter on its own (the ragged end of a range; the network weights a finalize block has just summed)
TCNN_DEVICE void adam_single(const AdamArgs& a, uint32_t i, half_t gradient, float* __restrict__ weights_fp32, half_t* __restrict__ weights,
                             float* __restrict__ first_moments, float* __restrict__ second_moments, uint32_t* __restrict__ param_steps,
                             half_t* __restrict__ weights_t, uint8_t* __restrict__ deficits8) {
	float wj = weights_fp32[i], m1j = first_moments[i], m2j = second_moments[i];
	const bool bytes = a.deficit == 2;
	const uint32_t dj = bytes ? deficits8[i] : 0u;
	uint32_t sj = bytes ? adam_count8(a, dj, param_steps, i) : (a.deficit ? a.steps_done - param_steps[i] : param_steps[i]);
	const bool stepped = adam_one(a, i, (float)gradient, wj, m1j, m2j, sj);
	if (bytes) deficits8[i] = (uint8_t)adam_next8(a, dj, stepped, sj, param_steps, i);
	if (stepped) {
		weights_fp32[i] = wj;
		first_moments[i] = m1j;
		second_moments[i] = m2j;
		if (!a.deficit) param_steps[i] = sj;
		weights[i] = to_half_rn(wj);
		if (weights_t && i < a.n_matrix_weights) weights_t[mlp_transposed_index(a.mlp, i)] = weights[i];
	} else if (a.deficit == 1) {
		param_steps[i] += 1u;
	}
}

// The network's weight-gradient slabs summed INSIDE the optimizer's launch (AdamFinalize): the first `blocks` workgroups of the launch each
// take 32 slab positions, add up the training kernel's fp32 slabs exactly as k_mlp_finalize_gradients does -- 32 groups of slabs, group g
// sums slabs g, g + 32, ... in that order, then the groups in order: the same bits -- store the 16-bit gradient and step the parameter
// right there; the other workgroups step the encoding's parameters as always.  One launch (and its dependent boundary) less per training
// step, and the 14.7 MB of slabs travel beside the optimizer's state stream instead of in a 224-workgroup kernel that is all latency.
// 256 threads: lane = thread & 31 (slab position), the thread's group of 32 = G = thread >> 5, which carries groups G, G + 8, G + 16, G + 24
// in separate accumulators, four loads of each in flight.
TCNN_DEVICE void adam_finalize_block(const AdamArgs& a, const AdamFinalize& fin, float* __restrict__ weights_fp32, half_t* __restrict__ weights,
                                     half_t* __restrict__ gradients, float* __restrict__ first_moments, float* __restrict__ second_moments,
                                     uint32_t* __restrict__ param_steps, half_t* __restrict__ weights_t, uint8_t* __restrict__ deficits8) {
	constexpr uint32_t GROUPS = 32, PER_THREAD = GROUPS / (EW_THREADS / 32), U = 16, CHUNK = 4;
	static_assert(EW_THREADS % 32 == 0 && GROUPS % (EW_THREADS / 32) == 0 && U % CHUNK == 0, "finalize geometry");
	__shared__ float red[GROUPS][32];
	const uint32_t lane = threadIdx.x & 31u, G = threadIdx.x >> 5;
	const uint32_t i = blockIdx.x * 32u + lane, n_params = a.n_matrix_weights;
	float s[PER_THREAD];
#pragma unroll
	for (uint32_t k = 0; k < PER_THREAD; ++k) s[k] = 0.0f;
	if (i < n_params) {
		// (PER_THREAD x CHUNK loads in flight and no more -- the loops are NOT unrolled further: this role shares its kernel, and with it its
		// register allocation, with the optimizer's streaming workgroups, whose eight waves per SIMD are what keeps HBM busy)
		const float* __restrict__ column = fin.partials + i;  // element offsets of a slab fit 32 bits (the host checks n_partials * n_params)
		for (uint32_t b0 = 0; b0 < fin.n_partials; b0 += GROUPS * U) {
#pragma clang loop unroll(disable)
			for (uint32_t c = 0; c < U; c += CHUNK) {
				float v[PER_THREAD][CHUNK];
#pragma unroll
				for (uint32_t k = 0; k < PER_THREAD; ++k) {
#pragma unroll
					for (uint32_t u = 0; u < CHUNK; ++u) {
						const uint32_t bb = b0 + (G + (EW_THREADS / 32) * k) + (c + u) * GROUPS;
						v[k][u] = bb < fin.n_partials ? column[bb * n_params] : 0.0f;
					}
				}
#pragma unroll
				for (uint32_t k = 0; k < PER_THREAD; ++k) {
#pragma unroll
					for (uint32_t u = 0; u < CHUNK; ++u) s[k] += v[k][u];
				}
			}
		}
	}
#pragma unroll
	for (uint32_t k = 0; k < PER_THREAD; ++k) red[G + (EW_THREADS / 32) * k][lane] = s[k];
	__syncthreads();
	if (G == 0 && i < n_params) {
		float t = red[0][lane];
#pragma unroll
		for (uint32_t k = 1; k < GROUPS; ++k) t += red[k][lane];
		const uint32_t param = fin.order == (uint32_t)SlabOrder::WaveRegisters ? mlp_wave_slab_param(a.mlp, i) : i;
		const half_t g = to_half_rn(t);
		gradients[param] = g;
		adam_single(a, param, g, weights_fp32, weights, first_moments, second_moments, param_steps, weights_t, deficits8);
	}
}

// 4 parameters per lane: 8 B of gradients decide whether the 16-byte state loads happen at all, so
// untouched stretches of a hash table cost 2 B/param as in the reference (adam.h:79-82).
template <bool STREAM, bool FINALIZE>
__global__ void __launch_bounds__(EW_THREADS) k_adam_step(const AdamArgs a, float* __restrict__ weights_fp32, half_t* __restrict__ weights,
                                                           half_t* __restrict__ gradients, float* __restrict__ first_moments,
                                                           float* __restrict__ second_moments, uint32_t* __restrict__ param_steps,
                                                           half_t* __restrict__ weights_t, uint8_t* __restrict__ deficits8, const AdamFinalize fin) {
	uint32_t block = blockIdx.x;
	if constexpr (FINALIZE) {  // (a.begin == the first parameter behind the network's: the host checked)
		if (block < fin.blocks) {
			adam_finalize_block(a, fin, weights_fp32, weights, gradients, first_moments, second_moments, param_steps, weights_t, deficits8);
			return;
		}
		block -= fin.blocks;
	}
	const uint32_t i0 = a.begin + (block * EW_THREADS + threadIdx.x) * 4;
	const bool four = i0 < a.n_elements && i0 + 3 < a.n_elements;
	h4 g = h4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
	if (four) g = *(const h4*)(gradients + i0);
	bool skip_all = four && i0 >= a.n_matrix_weights && a.skip_zero_grad_non_matrix_params && g[0] == (half_t)0.0f && g[1] == (half_t)0.0f &&
	                g[2] == (half_t)0.0f && g[3] == (half_t)0.0f;
	if (a.dense_store) {
		// the 8 lanes that share a 128-byte line of each fp32 state array decide together: a line nobody steps is not touched
		// (2 B per parameter, as the reference), a line somebody steps is written whole (a partly written line costs HBM a
		// read-modify-write).  Every lane of the wave votes, also those beyond the end (they never skip).
		const uint64_t skipping = __ballot(skip_all);
		skip_all = ((skipping >> (threadIdx.x & 56u)) & 0xFFull) == 0xFFull;
	}
	if (i0 >= a.n_elements) return;
	if (four) {
		if (skip_all) {
			if (a.deficit == 2) {  // all four skipped: one more missed step each
				const uint32_t b4 = *(const uint32_t*)(deficits8 + i0);
				uint32_t n4 = 0;
#pragma unroll
				for (uint32_t j = 0; j < 4; ++j) n4 |= adam_next8(a, (b4 >> (8u * j)) & 255u, false, 0u, param_steps, i0 + j) << (8u * j);
				if (n4 != b4) *(uint32_t*)(deficits8 + i0) = n4;
			} else if (a.deficit) {
				u4 st = adam_load<STREAM>((const u4*)(param_steps + i0));
				st += 1u;
				adam_store<STREAM>((u4*)(param_steps + i0), st);
			}
			return;
		}
		f4 w = adam_load<STREAM>((const f4*)(weights_fp32 + i0));
		f4 m1 = adam_load<STREAM>((const f4*)(first_moments + i0));
		f4 m2 = adam_load<STREAM>((const f4*)(second_moments + i0));
		const bool bytes = a.deficit == 2;
		const uint32_t b4 = bytes ? *(const uint32_t*)(deficits8 + i0) : 0u;
		uint32_t n4 = 0;
		u4 st = u4{0u, 0u, 0u, 0u};
		if (!bytes) st = adam_load<STREAM>((const u4*)(param_steps + i0));
		h4 wh = h4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
		uint32_t updated = 0;  // bit j: parameter i0 + j was stepped
		bool any = false;
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			float wj = w[j], m1j = m1[j], m2j = m2[j];
			const uint32_t dj = (b4 >> (8u * j)) & 255u;
			uint32_t sj = bytes ? adam_count8(a, dj, param_steps, i0 + j) : (a.deficit ? a.steps_done - st[j] : st[j]);
			const bool stepped = adam_one(a, i0 + j, (float)g[j], wj, m1j, m2j, sj);
			if (bytes) n4 |= adam_next8(a, dj, stepped, sj, param_steps, i0 + j) << (8u * j);
			if (stepped) {
				w[j] = wj;
				m1[j] = m1j;
				m2[j] = m2j;
				if (!a.deficit) st[j] = sj;
				wh[j] = to_half_rn(wj);
				if (weights_t && i0 + j < a.n_matrix_weights) weights_t[mlp_transposed_index(a.mlp, i0 + j)] = wh[j];
				any = true;
				updated |= 1u << j;
			} else if (a.deficit == 1) {
				st[j] += 1u;
			}
		}
		if (bytes && n4 != b4) *(uint32_t*)(deficits8 + i0) = n4;
		if (a.deficit == 1 && updated != 0xFu) adam_store<STREAM>((u4*)(param_steps + i0), st);
		if (any || a.dense_store) {
			if (updated != 0xFu) {  // keep the fp16 weights of the parameters that were skipped
				if (a.half_follows_master) {  // (they are the rounded master weights, AdamCore::half_follows_master: nothing to read)
#pragma unroll
					for (uint32_t j = 0; j < 4; ++j) {
						if (!((updated >> j) & 1u)) wh[j] = to_half_rn(w[j]);
					}
				} else {
					const h4 old = *(const h4*)(weights + i0);
#pragma unroll
					for (uint32_t j = 0; j < 4; ++j) {
						if (!((updated >> j) & 1u)) wh[j] = old[j];
					}
				}
			}
			adam_store<STREAM>((f4*)(weights_fp32 + i0), w);
			adam_store<STREAM>((f4*)(first_moments + i0), m1);
			adam_store<STREAM>((f4*)(second_moments + i0), m2);
			if (!a.deficit) adam_store<STREAM>((u4*)(param_steps + i0), st);
			*(h4*)(weights + i0) = wh;
		}
	} else {
		for (uint32_t i = i0; i < a.n_elements; ++i) adam_single(a, i, gradients[i], weights_fp32, weights, first_moments, second_moments, param_steps, weights_t, deficits8);
	}
}

constexpr size_t ADAM_STREAM_THRESHOLD_BYTES = 192u << 20;  // optimizer state beyond this cannot stay in the 256 MiB Infinity Cache

AdamCore make_adam_core(const AdamHyper& h, uint32_t n_matrix_weights, float loss_scale, uint32_t current_step, int steps_form) {
	AdamCore a;
	a.n_matrix_weights = n_matrix_weights;
	a.relative_weight_decay = h.relative_weight_decay;
	a.absolute_weight_decay = h.absolute_weight_decay;
	a.weight_clipping_magnitude = h.weight_clipping_magnitude;
	a.gradient_clipping_magnitude = h.gradient_clipping_magnitude;
	a.loss_scale = loss_scale;
	a.learning_rate = h.learning_rate;
	a.non_matrix_learning_rate_factor = h.non_matrix_learning_rate_factor;
	a.optimize_matrix_params = h.optimize_matrix_params;
	a.optimize_non_matrix_params = h.optimize_non_matrix_params;
	a.skip_zero_grad_non_matrix_params = h.skip_zero_grad_non_matrix_params;
	a.beta1 = h.beta1;
	a.beta2 = h.beta2;
	a.epsilon = h.epsilon;
	a.lower_lr_bound = 0;
	a.upper_lr_bound = 3.402823466e+38f;
	if (h.adabound) {  // adam.h:165-168
		a.lower_lr_bound = 0.1f - 0.1f / ((1 - h.beta2) * (float)current_step + 1);
		a.upper_lr_bound = 0.1f + 0.1f / ((1 - h.beta2) * (float)current_step);
	}
	a.l2_reg = h.l2_reg;
	a.non_matrix_l2_reg = h.non_matrix_l2_reg;
	a.steps_done = current_step - 1u;
	a.deficit = steps_form;
	a.dense_store = 1;
	a.half_follows_master = 0;
	return a;
}
bool adam_streams_its_state(uint32_t n) { return (size_t)n * 32u > ADAM_STREAM_THRESHOLD_BYTES; }

void adam_step(hipStream_t stream, const AdamHyper& h, uint32_t n, uint32_t n_matrix_weights, float loss_scale, uint32_t current_step,
               float* weights_fp32, half_t* weights, half_t* gradients, float* m1, float* m2, uint32_t* param_steps, half_t* weights_t,
               const MlpMeta* mlp, uint32_t begin, uint32_t end, int steps_form, uint8_t* deficits8, bool half_follows_master, const AdamFinalize* finalize) {
	if (end > n) end = n;
	if (begin >= end) return;
	if (begin % 4u != 0u) throw std::runtime_error("adam_step: a parameter range must start at a multiple of 4");
	if (weights_t && !mlp) throw std::runtime_error("adam_step: weights_t needs the network layout");
	AdamArgs a;
	if (steps_form == ADAM_STEPS_DEFICITS8 && !deficits8) throw std::runtime_error("adam_step: the byte form of the step deficits needs its array");
	(AdamCore&)a = make_adam_core(h, n_matrix_weights, loss_scale, current_step, steps_form);
	a.half_follows_master = half_follows_master ? 1 : 0;
	a.begin = begin;
	a.n_elements = end;
	a.mlp = mlp ? *mlp : MlpMeta{};
	AdamFinalize fin;
	if (finalize && finalize->partials) {
		// the finalize workgroups sum AND step the network's weights; the others start behind them
		if (begin != 0 || !mlp || mlp->n_params() != n_matrix_weights || n_matrix_weights % 4u != 0u || end < n_matrix_weights || finalize->n_partials == 0 ||
		    (uint64_t)finalize->n_partials * n_matrix_weights >= (1ull << 32)) {
			throw std::runtime_error("adam_step: the weight-gradient slabs can only be summed inside an optimizer step over the whole network");
		}
		fin = *finalize;
		fin.blocks = div_round_up(n_matrix_weights, 32u);
		a.begin = n_matrix_weights;
	}
	const dim3 grid(fin.blocks + div_round_up(div_round_up(end - a.begin, 4u), EW_THREADS));
#define TCNN_ADAM_LAUNCH(STREAM_, FINALIZE_) \
	TCNN_LAUNCH((k_adam_step<STREAM_, FINALIZE_>), grid, dim3(EW_THREADS), 0, stream, a, weights_fp32, weights, gradients, m1, m2, param_steps, weights_t, deficits8, fin)
	if (adam_streams_its_state(n)) {
		if (fin.blocks) TCNN_ADAM_LAUNCH(true, true); else TCNN_ADAM_LAUNCH(true, false);
	} else {
		if (fin.blocks) TCNN_ADAM_LAUNCH(false, true); else TCNN_ADAM_LAUNCH(false, false);
	}
#undef TCNN_ADAM_LAUNCH
}

// counters <-> deficits: x -> steps_done - x is its own inverse (mod 2^32)
__global__ void __launch_bounds__(EW_THREADS) k_adam_flip_steps(uint32_t n, uint32_t steps_done, uint32_t* __restrict__ param_steps) {
	const uint32_t i = blockIdx.x * EW_THREADS + threadIdx.x;
	if (i < n) param_steps[i] = steps_done - param_steps[i];
}
void adam_flip_step_representation(hipStream_t stream, uint32_t n, uint32_t steps_done, uint32_t* param_steps) {
	if (n == 0) return;
	TCNN_LAUNCH(k_adam_flip_steps, dim3(div_round_up(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, steps_done, param_steps);
}
// counters -> byte deficits (to_bytes) and back; a counter more than 254 steps behind keeps living in param_steps (byte 255)
__global__ void __launch_bounds__(EW_THREADS) k_adam_convert_steps8(uint32_t n, uint32_t steps_done, uint32_t* __restrict__ param_steps, uint8_t* __restrict__ deficits8,
                                                                     int to_bytes) {
	const uint32_t i = blockIdx.x * EW_THREADS + threadIdx.x;
	if (i >= n) return;
	if (to_bytes) {
		const uint32_t d = steps_done - param_steps[i];
		deficits8[i] = (uint8_t)(d < 255u ? d : 255u);
	} else {
		const uint32_t d = deficits8[i];
		if (d != 255u) param_steps[i] = steps_done - d;
	}
}
void adam_convert_step_representation(hipStream_t stream, uint32_t n, uint32_t steps_done, uint32_t* param_steps, uint8_t* deficits8, int from, int to) {
	if (n == 0 || from == to) return;
	// everything goes through the counters
	if (from == ADAM_STEPS_DEFICITS32) adam_flip_step_representation(stream, n, steps_done, param_steps);
	if (from == ADAM_STEPS_DEFICITS8) TCNN_LAUNCH(k_adam_convert_steps8, dim3(div_round_up(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, steps_done, param_steps, deficits8, 0);
	if (to == ADAM_STEPS_DEFICITS32) adam_flip_step_representation(stream, n, steps_done, param_steps);
	if (to == ADAM_STEPS_DEFICITS8) {
		if (!deficits8) throw std::runtime_error("adam_convert_step_representation: missing byte array");
		TCNN_LAUNCH(k_adam_convert_steps8, dim3(div_round_up(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, steps_done, param_steps, deficits8, 1);
	}
}

// The element-wise encodings write their value type VAL_T: the library's 16-bit type (rounded to nearest even, as the reference's
// (T) casts do), or float for the fp32 encodings of create_encoding(..., Precision::Fp32) (Encoding<float>, cpp_api.cu:165-168).
template <typename VAL_T>
TCNN_DEVICE VAL_T encoded_value(float v) {
	if constexpr (std::is_same<VAL_T, float>::value) return v;
	else return to_half_rn(v);
}
// ------------------------------------------------------------------------------------------ identity
template <typename VAL_T>
__global__ void k_identity_forward(uint32_t n, uint32_t n_dims, uint32_t padded, float scale, float offset, const float* __restrict__ in,
                                   uint32_t in_stride_i, uint32_t in_stride_j, VAL_T* __restrict__ out, uint32_t stride_k, uint32_t stride_i) {
	// thread -> (k, i) with i fastest: coalesced for the feature-major output the MLP kernels read
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n * padded) return;
	const uint32_t k = e / n, i = e - k * n;
	VAL_T v;
	if (k >= n_dims) {
		v = (VAL_T)1.0f;  // identity.h:62-64
	} else {
		float t = in[(size_t)i * in_stride_i + (size_t)k * in_stride_j] * scale;
		t = t + offset;
		v = encoded_value<VAL_T>(t);
	}
	out[(size_t)k * stride_k + (size_t)i * stride_i] = v;
}
// The layout the trainer runs: sample-major fp32 input [n][n_dims] -> feature-major half output [padded][n].  A workgroup
// transposes a tile of 256 samples through LDS: 16-byte reads along the features of consecutive samples, 8-byte writes that make
// every wave store one 512-byte run of a feature row (the element-wise form above reads one 128-byte line per lane: 80 us for
// 2^18 x 64 inputs; 64-sample tiles with 2-byte accesses: 30 us).
constexpr uint32_t ID_TILE = 256, ID_LD = ID_TILE + 4;  // row pitch in halves: 8-byte aligned rows, 2 banks apart
__global__ void __launch_bounds__(EW_THREADS) k_identity_forward_transpose(uint32_t n, uint32_t n_dims, uint32_t padded, float scale, float offset,
                                                                           const float* __restrict__ in, half_t* __restrict__ out) {
	TCNN_DYN_LDS(lds_raw);
	half_t* tile = (half_t*)lds_raw;  // [n_dims][ID_LD]
	const uint32_t first = blockIdx.x * ID_TILE;
	const uint32_t rows = min(ID_TILE, n - first);
	const float* src = in + (size_t)first * n_dims;
	if (n_dims % 4u == 0u && ((uintptr_t)src & 15u) == 0u) {
		for (uint32_t e4 = threadIdx.x; e4 < rows * n_dims / 4u; e4 += EW_THREADS) {
			const uint32_t e = 4u * e4, s = e / n_dims, k = e - s * n_dims;  // four features of one sample
			const f4 v = *(const f4*)(src + e);
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) {
				float t = v[j] * scale;
				t = t + offset;
				tile[(k + j) * ID_LD + s] = to_half_rn(t);
			}
		}
	} else {
		for (uint32_t e = threadIdx.x; e < rows * n_dims; e += EW_THREADS) {
			const uint32_t s = e / n_dims, k = e - s * n_dims;
			float t = src[e] * scale;
			t = t + offset;
			tile[k * ID_LD + s] = to_half_rn(t);
		}
	}
	__syncthreads();
	if (rows == ID_TILE && n % 4u == 0u && ((uintptr_t)out & 7u) == 0u) {  // four samples of one feature per lane
		const h4 ones = h4{(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};  // identity.h:62-64: padding features are 1
		for (uint32_t e4 = threadIdx.x; e4 < padded * (ID_TILE / 4u); e4 += EW_THREADS) {
			const uint32_t k = e4 / (ID_TILE / 4u), s = 4u * (e4 % (ID_TILE / 4u));
			*(h4*)(out + (size_t)k * n + first + s) = k < n_dims ? *(const h4*)(tile + k * ID_LD + s) : ones;
		}
	} else {
		for (uint32_t e = threadIdx.x; e < padded * ID_TILE; e += EW_THREADS) {
			const uint32_t k = e / ID_TILE, s = e % ID_TILE;
			if (s < rows) out[(size_t)k * n + first + s] = k < n_dims ? tile[k * ID_LD + s] : (half_t)1.0f;  // identity.h:62-64
		}
	}
}
template <typename VAL_T>
__global__ void k_identity_backward(uint32_t n, uint32_t n_dims, float scale, const VAL_T* __restrict__ dL_dy, uint32_t stride_k,
                                    uint32_t stride_i, float* __restrict__ dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n * n_dims) return;
	const uint32_t k = e / n, i = e - k * n;
	// identity.h:83: (T)((float)dL_dy * scale) -- rounded through half, then widened to the fp32 dL_dx
	dL_dx[(size_t)i * dx_stride_i + (size_t)k * dx_stride_j] = (float)encoded_value<VAL_T>((float)dL_dy[(size_t)k * stride_k + (size_t)i * stride_i] * scale);
}

void identity_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t padded, float scale, float offset, const float* in,
                      uint32_t in_stride_i, uint32_t in_stride_j, half_t* out, uint32_t stride_k, uint32_t stride_i) {
	if (n == 0) return;
	if (in_stride_j == 1 && in_stride_i == n_dims && stride_i == 1 && stride_k == n) {
		const uint32_t lds = n_dims * ID_LD * (uint32_t)sizeof(half_t);  // 66.5 KB at the widest input (128)
		TCNN_SET_MAX_DYN_LDS(k_identity_forward_transpose, lds);
		TCNN_LAUNCH(k_identity_forward_transpose, dim3(div_round_up(n, ID_TILE)), dim3(EW_THREADS), lds, stream, n, n_dims, padded, scale, offset, in, out);
		return;
	}
	TCNN_LAUNCH(k_identity_forward<half_t>, dim3(div_round_up(n * padded, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, padded, scale, offset, in, in_stride_i, in_stride_j, out, stride_k, stride_i);
}
void identity_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t padded, float scale, float offset, const float* in,
                      uint32_t in_stride_i, uint32_t in_stride_j, float* out, uint32_t stride_k, uint32_t stride_i) {
	if (n == 0) return;
	TCNN_LAUNCH(k_identity_forward<float>, dim3(div_round_up(n * padded, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, padded, scale, offset, in, in_stride_i, in_stride_j, out, stride_k, stride_i);
}
void identity_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, float scale, const half_t* dL_dy, uint32_t stride_k, uint32_t stride_i,
                       float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	if (n == 0) return;
	TCNN_LAUNCH(k_identity_backward<half_t>, dim3(div_round_up(n * n_dims, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, scale, dL_dy, stride_k, stride_i, dL_dx, dx_stride_i, dx_stride_j);
}
void identity_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, float scale, const float* dL_dy, uint32_t stride_k, uint32_t stride_i,
                       float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	if (n == 0) return;
	TCNN_LAUNCH(k_identity_backward<float>, dim3(div_round_up(n * n_dims, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, scale, dL_dy, stride_k, stride_i, dL_dx, dx_stride_i, dx_stride_j);
}

// ------------------------------------------------------------------------------------------ frequency
// encodings/frequency.h:46-79, 82-104: output j of sample i = sin(2^f pi x_d + (j & 1) pi / 2), d = j / (2 n_frequencies),
// f = (j / 2) % n_frequencies; padding 1.  The reference evaluates __sinf / __cosf (hardware approximations whose error grows
// with the argument, up to 2^11 pi here); this kernel and the oracle evaluate sinf / cosf of the SAME fp32 argument
// x 2^f * pi + phase, so parity with the reference is to its intrinsic's accuracy (a fp16 ulp at these magnitudes), parity
// between kernel and oracle to libm rounding.  One thread per output element (k-major, i fastest).
#define TCNN_PI_F 3.14159265358979323846f
template <typename VAL_T>
__global__ void __launch_bounds__(EW_THREADS) k_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* __restrict__ in,
                                                                  uint32_t in_stride_i, uint32_t in_stride_j, VAL_T* __restrict__ out, uint32_t stride_k,
                                                                  uint32_t stride_i) {
	const uint32_t e = blockIdx.x * EW_THREADS + threadIdx.x;
	if (e >= n * padded) return;
	const uint32_t j = e / n, i = e - j * n;
	VAL_T v = (VAL_T)1.0f;
	if (j < n_dims * n_frequencies * 2u) {
		const uint32_t d = j / (n_frequencies * 2u), log2_frequency = (j / 2u) % n_frequencies;
		const float phase_shift = (float)(j % 2u) * (TCNN_PI_F / 2);
		const float x = __builtin_scalbnf(in[(size_t)i * in_stride_i + (size_t)d * in_stride_j], (int)log2_frequency);
		const float input = x * TCNN_PI_F + phase_shift;
		v = encoded_value<VAL_T>(sinf(input));
	}
	out[(size_t)j * stride_k + (size_t)i * stride_i] = v;
}
template <typename VAL_T>
__global__ void __launch_bounds__(EW_THREADS) k_frequency_backward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const VAL_T* __restrict__ dL_dy,
                                                                   uint32_t stride_k, uint32_t stride_i, const float* __restrict__ in, uint32_t in_stride_i,
                                                                   uint32_t in_stride_j, float* __restrict__ dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	const uint32_t e = blockIdx.x * EW_THREADS + threadIdx.x;
	if (e >= n * n_dims) return;
	const uint32_t d = e / n, i = e - d * n, outputs_per_input = n_frequencies * 2u;
	const float x0 = in[(size_t)i * in_stride_i + (size_t)d * in_stride_j];
	float result = 0;
	for (uint32_t k = 0; k < outputs_per_input; ++k) {
		const uint32_t j = d * outputs_per_input + k, log2_frequency = k / 2u;
		const float phase_shift = (float)(k % 2u) * (TCNN_PI_F / 2);
		const float input = __builtin_scalbnf(x0, (int)log2_frequency) * TCNN_PI_F + phase_shift;
		const float dy_dx = __builtin_scalbnf(1.0f, (int)log2_frequency) * TCNN_PI_F * cosf(input);  // what the reference's forward pass stores
		result += (float)dL_dy[(size_t)j * stride_k + (size_t)i * stride_i] * dy_dx;
	}
	dL_dx[(size_t)i * dx_stride_i + (size_t)d * dx_stride_j] = result;
}
template <typename VAL_T>
static void frequency_forward_t(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint32_t in_stride_i,
                                uint32_t in_stride_j, VAL_T* out, uint32_t stride_k, uint32_t stride_i) {
	if (n == 0) return;
	TCNN_LAUNCH(k_frequency_forward<VAL_T>, dim3(div_round_up(n * padded, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, n_frequencies, padded, in, in_stride_i,
	            in_stride_j, out, stride_k, stride_i);
}
template <typename VAL_T>
static void frequency_backward_t(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const VAL_T* dL_dy, uint32_t stride_k, uint32_t stride_i,
                                 const float* in, uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	if (n == 0) return;
	TCNN_LAUNCH(k_frequency_backward<VAL_T>, dim3(div_round_up(n * n_dims, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, n_frequencies, dL_dy, stride_k, stride_i, in,
	            in_stride_i, in_stride_j, dL_dx, dx_stride_i, dx_stride_j);
}
void frequency_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint32_t in_stride_i,
                       uint32_t in_stride_j, half_t* out, uint32_t stride_k, uint32_t stride_i) {
	frequency_forward_t<half_t>(stream, n, n_dims, n_frequencies, padded, in, in_stride_i, in_stride_j, out, stride_k, stride_i);
}
void frequency_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint32_t in_stride_i,
                       uint32_t in_stride_j, float* out, uint32_t stride_k, uint32_t stride_i) {
	frequency_forward_t<float>(stream, n, n_dims, n_frequencies, padded, in, in_stride_i, in_stride_j, out, stride_k, stride_i);
}
void frequency_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const half_t* dL_dy, uint32_t stride_k, uint32_t stride_i,
                        const float* in, uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	frequency_backward_t<half_t>(stream, n, n_dims, n_frequencies, dL_dy, stride_k, stride_i, in, in_stride_i, in_stride_j, dL_dx, dx_stride_i, dx_stride_j);
}
void frequency_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* dL_dy, uint32_t stride_k, uint32_t stride_i,
                        const float* in, uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	frequency_backward_t<float>(stream, n, n_dims, n_frequencies, dL_dy, stride_k, stride_i, in, in_stride_i, in_stride_j, dL_dx, dx_stride_i, dx_stride_j);
}

// ------------------------------------------------------------------------------------------ one-blob
// encodings/oneblob.h:84-164 (the SoA kernels' arithmetic) with common_device.h:1076-1095 (quartic kernel).  One thread per
// (dimension j, sample i), i fastest: the n_bins bin integrals of a quartic blob centred at x, wrapped around [0, 1).
TCNN_DEVICE float quartic(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float tmp = __builtin_fmaxf(1 - u * u, 0.0f);
	return ((float)15 / 16) * tmp * tmp;
}
TCNN_DEVICE float quartic_cdf_deriv(float x, float inv_radius) { return quartic(x, inv_radius) * inv_radius; }
TCNN_DEVICE float quartic_cdf(float x, float inv_radius) {
	const float u = x * inv_radius;
	const float u2 = u * u;
	const float u4 = u2 * u2;
	return __builtin_fmaxf(0.0f, __builtin_fminf(1.0f, ((float)15 / 16) * u * (1 - ((float)2 / 3) * u2 + ((float)1 / 5) * u4) + 0.5f));
}
template <typename VAL_T>
__global__ void __launch_bounds__(EW_THREADS) k_oneblob_forward(uint32_t n, uint32_t n_dims, uint32_t log2_bins, uint32_t padded, const float* __restrict__ in,
                                                                uint32_t in_stride_i, uint32_t in_stride_j, VAL_T* __restrict__ out, uint32_t stride_k,
                                                                uint32_t stride_i) {
	const uint32_t e = blockIdx.x * EW_THREADS + threadIdx.x;
	const uint32_t n_bins = 1u << log2_bins, n_out = n_dims * n_bins;
	if (e < n * n_dims) {
		const uint32_t j = e / n, i = e - j * n;
		const float x = in[(size_t)i * in_stride_i + (size_t)j * in_stride_j];
		const float nb = (float)n_bins, inv_bins = 1.0f / nb;  // scalbnf(k, -log2_bins) == k * inv_bins exactly
		float left_cdf = quartic_cdf(-x, nb) + quartic_cdf(-x - 1.0f, nb) + quartic_cdf(-x + 1.0f, nb);
		for (uint32_t k = 0; k < n_bins; ++k) {
			const float right_boundary = (float)(k + 1) * inv_bins;
			const float right_cdf = quartic_cdf(right_boundary - x, nb) + quartic_cdf(right_boundary - x - 1.0f, nb) + quartic_cdf(right_boundary - x + 1.0f, nb);
			out[(size_t)(j * n_bins + k) * stride_k + (size_t)i * stride_i] = encoded_value<VAL_T>(right_cdf - left_cdf);
			left_cdf = right_cdf;
		}
	} else if (e < n * n_dims + n * (padded - n_out)) {  // oneblob.h:214-216, 232-234: padding is 1
		const uint32_t q = e - n * n_dims, k = n_out + q / n, i = q % n;
		out[(size_t)k * stride_k + (size_t)i * stride_i] = (VAL_T)1.0f;
	}
}
template <typename VAL_T>
__global__ void __launch_bounds__(EW_THREADS) k_oneblob_backward(uint32_t n, uint32_t n_dims, uint32_t log2_bins, const VAL_T* __restrict__ dL_dy, uint32_t stride_k,
                                                                 uint32_t stride_i, const float* __restrict__ in, uint32_t in_stride_i, uint32_t in_stride_j,
                                                                 float* __restrict__ dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	const uint32_t e = blockIdx.x * EW_THREADS + threadIdx.x;
	if (e >= n * n_dims) return;
	const uint32_t j = e / n, i = e - j * n, n_bins = 1u << log2_bins;
	const float x = in[(size_t)i * in_stride_i + (size_t)j * in_stride_j];
	const float nb = (float)n_bins, inv_bins = 1.0f / nb;
	float result = 0;
	float left_cdf = quartic_cdf_deriv(-x, nb) + quartic_cdf_deriv(-x - 1.0f, nb) + quartic_cdf_deriv(-x + 1.0f, nb);
	for (uint32_t k = 0; k < n_bins; ++k) {
		const float right_boundary = (float)(k + 1) * inv_bins;
		const float right_cdf = quartic_cdf_deriv(right_boundary - x, nb) + quartic_cdf_deriv(right_boundary - x - 1.0f, nb) + quartic_cdf_deriv(right_boundary - x + 1.0f, nb);
		const float deriv = left_cdf - right_cdf;
		left_cdf = right_cdf;
		result += (float)dL_dy[(size_t)(j * n_bins + k) * stride_k + (size_t)i * stride_i] * deriv;
	}
	dL_dx[(size_t)i * dx_stride_i + (size_t)j * dx_stride_j] = result;
}

template <typename VAL_T>
static void oneblob_forward_t(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint32_t in_stride_i,
                              uint32_t in_stride_j, VAL_T* out, uint32_t stride_k, uint32_t stride_i) {
	if (n == 0) return;
	uint32_t log2_bins = 0;
	while ((1u << log2_bins) < n_bins) ++log2_bins;
	const uint32_t work = n * n_dims + n * (padded - n_dims * n_bins);
	TCNN_LAUNCH(k_oneblob_forward<VAL_T>, dim3(div_round_up(work, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, log2_bins, padded, in, in_stride_i, in_stride_j, out,
	            stride_k, stride_i);
}
template <typename VAL_T>
static void oneblob_backward_t(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, const VAL_T* dL_dy, uint32_t stride_k, uint32_t stride_i, const float* in,
                               uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	if (n == 0) return;
	uint32_t log2_bins = 0;
	while ((1u << log2_bins) < n_bins) ++log2_bins;
	TCNN_LAUNCH(k_oneblob_backward<VAL_T>, dim3(div_round_up(n * n_dims, EW_THREADS)), dim3(EW_THREADS), 0, stream, n, n_dims, log2_bins, dL_dy, stride_k, stride_i, in,
	            in_stride_i, in_stride_j, dL_dx, dx_stride_i, dx_stride_j);
}
void oneblob_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint32_t in_stride_i,
                     uint32_t in_stride_j, half_t* out, uint32_t stride_k, uint32_t stride_i) {
	oneblob_forward_t<half_t>(stream, n, n_dims, n_bins, padded, in, in_stride_i, in_stride_j, out, stride_k, stride_i);
}
void oneblob_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint32_t in_stride_i,
                     uint32_t in_stride_j, float* out, uint32_t stride_k, uint32_t stride_i) {
	oneblob_forward_t<float>(stream, n, n_dims, n_bins, padded, in, in_stride_i, in_stride_j, out, stride_k, stride_i);
}
void oneblob_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, const half_t* dL_dy, uint32_t stride_k, uint32_t stride_i, const float* in,
                      uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	oneblob_backward_t<half_t>(stream, n, n_dims, n_bins, dL_dy, stride_k, stride_i, in, in_stride_i, in_stride_j, dL_dx, dx_stride_i, dx_stride_j);
}
void oneblob_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, const float* dL_dy, uint32_t stride_k, uint32_t stride_i, const float* in,
                      uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j) {
	oneblob_backward_t<float>(stream, n, n_dims, n_bins, dL_dy, stride_k, stride_i, in, in_stride_i, in_stride_j, dL_dx, dx_stride_i, dx_stride_j);
}

}  // namespace tcnn_hip
