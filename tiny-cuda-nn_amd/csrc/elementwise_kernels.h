// elementwise_kernels.h -- streaming (HBM-bound) kernels of the training step: loss, Adam, casts,
// PCG32 fills, reductions, identity encoding.  Reference lines restated are cited per function.
#pragma once
#include "loss_device.h"
#include "mlp_kernels.h"
#include "tcnn_device.h"

namespace tcnn_hip {

// LossType lives in loss_device.h (shared with the fused MLP training kernel)

struct Pcg32 {  // reference dependencies/pcg32/pcg32.h:40-170
	uint64_t state, inc;
	TCNN_HOST_DEVICE Pcg32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
	TCNN_HOST_DEVICE explicit Pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }
	TCNN_HOST_DEVICE void seed(uint64_t initstate, uint64_t initseq = 1) {
		state = 0U;
		inc = (initseq << 1u) | 1u;
		next_uint();
		state += initstate;
		next_uint();
	}
	TCNN_HOST_DEVICE uint32_t next_uint() {
		const uint64_t oldstate = state;
		state = oldstate * 0x5851f42d4c957f2dULL + inc;
		const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
		const uint32_t rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	TCNN_HOST_DEVICE float next_float() {
		union { uint32_t u; float f; } x;
		x.u = (next_uint() >> 9) | 0x3f800000u;
		return x.f - 1.0f;
	}
	TCNN_HOST_DEVICE void advance(int64_t delta_) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
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
		state = acc_mult * state + acc_plus;
	}
};

// random.h:39-75 -- fills out[0..n) with U[lower, upper) and advances rng by n (same idx <-> stream map)
void generate_random_uniform(hipStream_t stream, Pcg32& rng, size_t n, float* out, float lower, float upper);

void cast_f32_to_f16(hipStream_t stream, size_t n, const float* in, half_t* out);  // trainer.h:415-417
void cast_f16_to_f32(hipStream_t stream, size_t n, const half_t* in, float* out);  // trainer.h:430-432
void sinusoid_targets(hipStream_t stream, uint32_t n, uint32_t n_in, uint32_t n_out, const float* positions, float* targets);  // bench / sample targets
void resync_master_from_half(hipStream_t stream, size_t n, const half_t* in, float* master);  // master := half where round(master) != half
void cast_scaled_f32_to_f16(hipStream_t stream, size_t n, const float* in, half_t* out, float scale);
void cast_scaled_f16_to_f32(hipStream_t stream, size_t n, const half_t* in, float* out, float scale);
void scale_f32(hipStream_t stream, size_t n, float* data, float scale);
// the same with the scale read from device memory.  gradient_scale_from_absmax fills pair_and_scratch (3 floats of device memory: scale,
// 1 / scale, scratch) with the largest power of two s <= cap such that s * max |in| <= target (cap when the input is all zero)
void gradient_scale_from_absmax(hipStream_t stream, size_t n, const float* in, float* pair_and_scratch, float cap, float target);
void cast_scaled_f32_to_f16(hipStream_t stream, size_t n, const float* in, half_t* out, const float* scale_dev);
void cast_scaled_f16_to_f32(hipStream_t stream, size_t n, const half_t* in, float* out, const float* scale_dev);
void scale_f32(hipStream_t stream, size_t n, float* data, const float* scale_dev);
void fill_f16(hipStream_t stream, size_t n, half_t* out, float value);

// object.cu:61-67 trim_and_cast_from: in half AoS [n][padded] -> out float element (j, i) at out[i*stride_i + j*stride_j]
void trim_and_cast(hipStream_t stream, uint32_t n, uint32_t padded, uint32_t dims, const half_t* in, float* out,
                   uint32_t stride_i, uint32_t stride_j);

// losses/relative_l2.h:40-76, losses/l2.h:40-76.  prediction/gradients: half AoS [n][stride];
// target/data_pdf: fp32, element (dim j, sample i) at [i*dims + j].  values (fp32 [n][stride]) may be
// null.  block_sums (may be null): one fp32 partial loss sum per 256-thread block, n_blocks returned.
// n_total: normalisation count (n*dims on one GPU; the GLOBAL batch*dims under data parallelism).
uint32_t loss_n_blocks(uint32_t n, uint32_t stride);
void loss_evaluate(hipStream_t stream, LossType type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale,
                   const half_t* prediction, const float* target, const float* data_pdf, float* values,
                   half_t* gradients, float* block_sums, uint32_t n_total);

// reduce_sum.h:117-156 (deterministic two-pass): *out = sum(in[0..n))
void reduce_sum(hipStream_t stream, const float* in, size_t n, float* workspace /* >= 1024 floats */, float* out);

struct AdamHyper {  // optimizers/adam.h:330-351 defaults
	float learning_rate = 1e-3f, beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-8f, l2_reg = 1e-8f, non_matrix_l2_reg = 0.0f;
	float relative_weight_decay = 0.0f, absolute_weight_decay = 0.0f;
	float weight_clipping_magnitude = 0.0f, gradient_clipping_magnitude = 0.0f;
	float non_matrix_learning_rate_factor = 1.0f;
	bool adabound = false;
	bool optimize_matrix_params = true, optimize_non_matrix_params = true, skip_zero_grad_non_matrix_params = true;
};
// optimizers/adam.h:48-127, 158-199.  current_step = optimizer step AFTER the increment.
// optimizers/ema.h:45-141: weights_ema <- debiased EMA of weights after optimizer step `current_step` (>= 1);
// tmp: fp32 shadow of the average (full_precision) or nullptr.
void ema_step(hipStream_t stream, uint32_t n, float ema_decay, uint32_t current_step, const half_t* weights, half_t* weights_ema, float* tmp,
              uint32_t begin = 0, uint32_t end = 0xFFFFFFFFu);

// the per-parameter arithmetic and its argument block (adam_device.h), for kernels that apply the step themselves
struct AdamCore;
// Representation of the per-parameter step counters (adam.h:84 m_param_steps), see adam_device.h AdamCore::deficit:
// the counters themselves; their deficits steps_done - counter as 32-bit words; the deficits as BYTES (255 = "the counter itself is in
// the 32-bit array"): a parameter that is stepped every time costs 4 / 1 bytes of bookkeeping per step and no write.
enum AdamStepsForm : int { ADAM_STEPS_COUNTERS = 0, ADAM_STEPS_DEFICITS32 = 1, ADAM_STEPS_DEFICITS8 = 2 };
AdamCore make_adam_core(const AdamHyper& h, uint32_t n_matrix_weights, float loss_scale, uint32_t current_step, int steps_form);
bool adam_streams_its_state(uint32_t n_params);  // optimizer state too large for the Infinity Cache: non-temporal accesses

// weights_t (nullable) + mlp: also keep the transposed copy of the network weights (mlp_transposed_index) current, so
// that the next training step does not need a transposition pass.
// finalize (nullable): the network's weight gradients are still fp32 slabs of the training kernel (mlp_train's `partials`, `order` as it
// returned them): the first workgroups of THIS launch sum them -- the same additions in the same order as mlp_finalize_gradients --, write the
// 16-bit gradients into `gradients` and step those parameters; only for a step over the whole network (begin == 0, GradientMode::Overwrite).
struct AdamFinalize {
	const float* partials = nullptr;
	uint32_t n_partials = 0, order = 0;
	uint32_t blocks = 0;  // (set by adam_step)
};
void adam_step(hipStream_t stream, const AdamHyper& h, uint32_t n, uint32_t n_matrix_weights, float loss_scale,
               uint32_t current_step, float* weights_fp32, half_t* weights, half_t* gradients, float* m1, float* m2,
               uint32_t* param_steps, half_t* weights_t = nullptr, const MlpMeta* mlp = nullptr, uint32_t begin = 0, uint32_t end = 0xFFFFFFFFu,
               int steps_form = ADAM_STEPS_COUNTERS, uint8_t* deficits8 = nullptr, bool half_follows_master = false,  // AdamCore::half_follows_master
               const AdamFinalize* finalize = nullptr);
// param_steps holds either the per-parameter step counters (adam.h:84) or, with steps_are_deficits, their deficit
// steps_done - counter (steps_done = current_step - 1): a stepped parameter then reads its 4 bytes and writes nothing, a
// skipped one is incremented -- cheaper when most parameters are stepped every time (the headline table: 98 %), dearer
// when most are skipped.  This converts one representation into the other, in place (its own inverse).
void adam_flip_step_representation(hipStream_t stream, uint32_t n, uint32_t steps_done, uint32_t* param_steps);
// any form -> any form (AdamStepsForm), in place; deficits8 (n bytes) is needed when the byte form is involved
void adam_convert_step_representation(hipStream_t stream, uint32_t n, uint32_t steps_done, uint32_t* param_steps, uint8_t* deficits8, int from, int to);

// encodings/identity.h:46-84.  in: fp32 element (dim j, sample i) at in[i*in_stride_i + j*in_stride_j];
// out: half element (k, i) at out[k*stride_k + i*stride_i], k < padded, padding value 1.
void identity_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t padded, float scale, float offset,
                      const float* in, uint32_t in_stride_i, uint32_t in_stride_j, half_t* out, uint32_t stride_k,
                      uint32_t stride_i);
void identity_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, float scale, const half_t* dL_dy, uint32_t stride_k,
                       uint32_t stride_i, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j);

// encodings/frequency.h:46-104: sin / cos of 2^f pi x, padding value 1; dL/dinput recomputes the derivative the reference stores.
void frequency_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint32_t in_stride_i,
                       uint32_t in_stride_j, half_t* out, uint32_t stride_k, uint32_t stride_i);
void frequency_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const half_t* dL_dy, uint32_t stride_k, uint32_t stride_i,
                        const float* in, uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j);

// encodings/oneblob.h:84-164 (n_bins a power of two; padding value 1).  Same addressing conventions as the identity encoding.
void oneblob_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint32_t in_stride_i,
                     uint32_t in_stride_j, half_t* out, uint32_t stride_k, uint32_t stride_i);
void oneblob_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, const half_t* dL_dy, uint32_t stride_k, uint32_t stride_i, const float* in,
                      uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j);

// The same three encodings with fp32 values (Encoding<float>: create_encoding(..., Precision::Fp32), cpp_api.cu:165-168): encoded features
// and incoming gradients are float, nothing is rounded to 16 bits.
void identity_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t padded, float scale, float offset, const float* in, uint32_t in_stride_i,
                      uint32_t in_stride_j, float* out, uint32_t stride_k, uint32_t stride_i);
void identity_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, float scale, const float* dL_dy, uint32_t stride_k, uint32_t stride_i, float* dL_dx,
                       uint32_t dx_stride_i, uint32_t dx_stride_j);
void frequency_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint32_t in_stride_i,
                       uint32_t in_stride_j, float* out, uint32_t stride_k, uint32_t stride_i);
void frequency_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const float* dL_dy, uint32_t stride_k, uint32_t stride_i,
                        const float* in, uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j);
void oneblob_forward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint32_t in_stride_i,
                     uint32_t in_stride_j, float* out, uint32_t stride_k, uint32_t stride_i);
void oneblob_backward(hipStream_t stream, uint32_t n, uint32_t n_dims, uint32_t n_bins, const float* dL_dy, uint32_t stride_k, uint32_t stride_i, const float* in,
                      uint32_t in_stride_i, uint32_t in_stride_j, float* dL_dx, uint32_t dx_stride_i, uint32_t dx_stride_j);

}  // namespace tcnn_hip
