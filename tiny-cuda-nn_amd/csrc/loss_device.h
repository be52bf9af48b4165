// loss_device.h -- per-element loss value and scaled gradient, shared by the stand-alone loss kernel
// (elementwise_kernels.hip) and the fused MLP training kernel (mlp_kernels.hip) so that both produce the same bits.
// Reference: include/tiny-cuda-nn/losses/relative_l2.h:40-86, l2.h:40-83.
#pragma once
#include "tcnn_device.h"

namespace tcnn_hip {

enum class LossType : int { L2 = 0, RelativeL2 = 1 };

// prediction: the fp16 network output widened to fp32.  Returns the fp16 gradient loss_scale * dL/dprediction / n_total
// (relative_l2.h:80), `value` receives this element's share of the mean loss (relative_l2.h:77).
template <LossType LOSS>
TCNN_DEVICE half_t loss_element(float prediction, float target, float pdf, float n_total, float loss_scale, float& value) {
	const float difference = prediction - target;
	float gradient;
	if (LOSS == LossType::RelativeL2) {
		const float prediction_sq_plus_epsilon = prediction * prediction + 0.01f;
		value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
		gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
	} else {
		value = difference * difference / pdf / n_total;
		gradient = 2 * difference / pdf;
	}
	return to_half_rn(loss_scale * gradient / n_total);
}

}  // namespace tcnn_hip
