// adam_device.h -- Adam's per-parameter arithmetic (optimizers/adam.h:48-127), shared by the stand-alone optimizer kernel
// (elementwise_kernels.hip) and the grid backward's owner pass, which can apply the step straight from the exact gradient
// sums it holds in LDS (grid_kernels.hip).
#pragma once
#include "tcnn_device.h"

namespace tcnn_hip {

struct AdamCore {
	uint32_t n_matrix_weights;  // parameters [0, n_matrix_weights) are network weights ("matrix params"), the rest grid entries
	float relative_weight_decay, absolute_weight_decay, weight_clipping_magnitude, gradient_clipping_magnitude;
	float loss_scale, learning_rate, non_matrix_learning_rate_factor;
	int optimize_matrix_params, optimize_non_matrix_params, skip_zero_grad_non_matrix_params;
	float beta1, beta2, epsilon, lower_lr_bound, upper_lr_bound, l2_reg, non_matrix_l2_reg;
	// Representation of the per-parameter step counters (adam.h:84 m_param_steps).  deficit == 0: the counters themselves.
	// deficit != 0: the array holds steps_done - counter, the number of optimizer steps that SKIPPED the parameter; a
	// parameter that is stepped then costs a 4-byte read and no write (its deficit does not change), only skipped ones
	// are written.  steps_done = optimizer steps before this one.
	uint32_t steps_done;
	int deficit;
	// lanes whose four parameters are all skipped (untouched hash-table entries) still load and store their unchanged state,
	// so that every 128-byte line of the optimizer state is written whole: a line is fetched as soon as one of its 32
	// parameters is stepped anyway, and partially written lines cost the memory system more than full ones (T = 2^22,
	// 60 % of the entries untouched per step: 0.68 -> 0.53 ms, 4.7 -> 6.0 TB/s; profiles/r02_exp_notes.txt)
	int dense_store;
	// the 16-bit weights are the rounded fp32 master weights for EVERY parameter of this call (the trainer's own buffers: both are only
	// ever written together -- Adam itself, the casts of set_params / snapshots).  A lane that steps some of its four parameters and skips
	// others then re-derives the skipped ones' 16-bit weights from the master weights it holds anyway instead of reading them back
	// (8 B per lane; the reference leaves them untouched, adam.h:79-82: the same bits).  Late in a long run, when converged gradients
	// underflow to fp16 zero here and there, most lanes are such mixed lanes.  0: read them back (buffers a caller may write separately).
	int half_follows_master;
};

// One parameter, exactly the arithmetic of adam.h:66-126.  Returns false if the parameter is skipped.
TCNN_DEVICE bool adam_one(const AdamCore& a, uint32_t i, float gradient_raw, float& weight_fp, float& m1, float& m2, uint32_t& step) {
	float gradient = gradient_raw / a.loss_scale;
	if (i >= a.n_matrix_weights) {
		if (!a.optimize_non_matrix_params || (gradient == 0 && a.skip_zero_grad_non_matrix_params)) return false;
	} else {
		if (!a.optimize_matrix_params) return false;
	}
	if (i < a.n_matrix_weights) {
		gradient += a.l2_reg * weight_fp;
	} else {
		gradient += a.non_matrix_l2_reg * weight_fp;
	}
	if (a.gradient_clipping_magnitude != 0.0f) {
		gradient = __builtin_copysignf(__builtin_fminf(__builtin_fabsf(gradient), a.gradient_clipping_magnitude), gradient);
	}
	const float gradient_sq = gradient * gradient;
	const float first_moment = m1 = a.beta1 * m1 + (1 - a.beta1) * gradient;
	const float second_moment = m2 = a.beta2 * m2 + (1 - a.beta2) * gradient_sq;
	float learning_rate = a.learning_rate;
	if (i >= a.n_matrix_weights) learning_rate *= a.non_matrix_learning_rate_factor;
	const uint32_t current_step = ++step;
	learning_rate *= __builtin_sqrtf(1 - __builtin_powf(a.beta2, (float)current_step)) / (1 - __builtin_powf(a.beta1, (float)current_step));
	const float effective_learning_rate =
		__builtin_fminf(__builtin_fmaxf(learning_rate / (__builtin_sqrtf(second_moment) + a.epsilon), a.lower_lr_bound), a.upper_lr_bound);
	// common_device.h:1045-1048 weight_decay
	const float rel = a.relative_weight_decay * learning_rate, ab = a.absolute_weight_decay * learning_rate;
	const float decayed_weight = (1 - rel) * weight_fp - __builtin_copysignf(ab, weight_fp);
	float new_weight = decayed_weight - effective_learning_rate * first_moment;
	if (a.weight_clipping_magnitude != 0.0f) {
		new_weight = __builtin_fminf(__builtin_fmaxf(new_weight, -a.weight_clipping_magnitude), a.weight_clipping_magnitude);
	}
	weight_fp = new_weight;
	return true;
}

// STREAM: the optimizer state (fp32 master, moments, step counters: 32 B per parameter) is read and written with
// non-temporal accesses.  For tables far larger than the caches it cannot survive until the next step anyway, and
// streaming it leaves L2 / Infinity Cache to the grid tables and the backward pass (measured at the headline size:
// Adam +5 us, record scatter -8 us, owner pass -2 us).  Small models keep the cached path: their state stays resident.
template <bool STREAM, typename T>
TCNN_DEVICE T adam_load(const T* p) {
#if !defined(TCNN_HOST_EMU)
	if constexpr (STREAM) return __builtin_nontemporal_load(p);
#endif
	return *p;
}
template <bool STREAM, typename T>
TCNN_DEVICE void adam_store(T* p, T v) {
#if !defined(TCNN_HOST_EMU)
	if constexpr (STREAM) {
		__builtin_nontemporal_store(v, p);
		return;
	}
#endif
	*p = v;
}


}  // namespace tcnn_hip
