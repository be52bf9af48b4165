// loss_device.h -- per-element loss value and scaled gradient, shared by the stand-alone loss kernel
// (elementwise_kernels.hip) and the fused MLP training kernel (mlp_kernels.hip) so that both produce the same bits.
// Reference: include/tiny-cuda-nn/losses/{l2,relative_l2,l1,relative_l1,mape,smape,cross_entropy,variance_is}.h
// (the element kernels at :40-80 of each); names as registered in src/loss.cu:57-65.
#pragma once
#include "tcnn_device.h"
#if defined(TCNN_HOST_EMU)
#include <math.h>
#endif

namespace tcnn_hip {

enum class LossType : int { L2 = 0, RelativeL2 = 1, L1 = 2, RelativeL1 = 3, Mape = 4, Smape = 5, CrossEntropy = 6, Variance = 7, RelativeL2Luminance = 8 };
constexpr int N_LOSS_TYPES = 9;
// RelativeL2Luminance (relative_l2_luminance.h:66-86) normalises by the luminance of the sample's first three (six) outputs:
// it needs the whole row, so only the stand-alone loss kernel evaluates it (loss_row_luminance + loss_element_luminance).

// prediction: the fp16 network output widened to fp32.  Returns the fp16 gradient loss_scale * dL/dprediction / n_total,
// `value` receives this element's share of the mean loss.  `type` is uniform over the launch.
#if defined(TCNN_HOST_EMU)
#define TCNN_LOSS_NOINLINE inline
#else
#define TCNN_LOSS_NOINLINE __device__ __attribute__((noinline))
#endif
TCNN_LOSS_NOINLINE half_t loss_element_general(LossType type, float prediction, float target, float pdf, float n_total, float loss_scale, float& value) {
	const float difference = prediction - target;
	float gradient;  // dL/dprediction before the 1 / n_total of the mean
	switch (type) {
		case LossType::RelativeL2: {  // relative_l2.h:70-80
			const float prediction_sq_plus_epsilon = prediction * prediction + 0.01f;
			value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
			gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
			break;
		}
		case LossType::L1:  // l1.h:69-74
			value = fabsf(difference) / pdf / n_total;
			gradient = copysignf(1.0f / pdf, difference);
			break;
		case LossType::RelativeL1: {  // relative_l1.h:69-76
			const float scale = 1.0f / (fabsf(prediction) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
			break;
		}
		case LossType::Mape: {  // mape.h:69-77
			const float scale = 1.0f / (fabsf(target) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
			break;
		}
		case LossType::Smape: {  // smape.h:69-77
			const float scale = 1.0f / (0.5f * (fabsf(target) + fabsf(prediction)) + 1e-2f) / pdf;
			value = fabsf(difference) * scale / n_total;
			gradient = copysignf(scale, difference);
			break;
		}
		case LossType::CrossEntropy: {  // cross_entropy.h:66-76: the factor already holds 1 / n_total
			const float factor = -target / pdf / n_total;
			value = factor * logf(prediction);
			return to_half_rn(loss_scale * (factor / prediction));
		}
		case LossType::Variance: {  // variance_is.h:66-76
			const float factor = target * target / pdf / n_total;
			value = factor / prediction - factor / pdf;
			return to_half_rn(loss_scale * (-factor / (prediction * prediction)));
		}
		default:  // L2, l2.h:69-74
			value = difference * difference / pdf / n_total;
			gradient = 2 * difference / pdf;
			break;
	}
	return to_half_rn(loss_scale * gradient / n_total);
}

TCNN_DEVICE float loss_row_luminance(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }
TCNN_DEVICE half_t loss_element_luminance(float prediction, float luminance, float target, float pdf, float n_total, float loss_scale, float& value) {
	const float prediction_sq_plus_epsilon = luminance * luminance + 0.01f;
	const float difference = prediction - target;
	value = difference * difference / prediction_sq_plus_epsilon / pdf / n_total;
	const float gradient = 2 * difference / prediction_sq_plus_epsilon / pdf;
	return to_half_rn(loss_scale * gradient / n_total);
}

// RelativeL2 / L2 (the defaults) inline, the rest through one out-of-line copy (see activation_device.h)
TCNN_HOST_DEVICE bool loss_is_simple(LossType type) { return type == LossType::RelativeL2 || type == LossType::L2; }
TCNN_HOST_DEVICE bool loss_is_elementwise(LossType type) { return type != LossType::RelativeL2Luminance; }
template <bool GENERAL = true>
TCNN_DEVICE half_t loss_element(LossType type, float prediction, float target, float pdf, float n_total, float loss_scale, float& value) {
	if (!GENERAL || type == LossType::RelativeL2 || type == LossType::L2) {
		const float difference = prediction - target;
		const float denom = type == LossType::RelativeL2 ? prediction * prediction + 0.01f : 1.0f;
		value = type == LossType::RelativeL2 ? difference * difference / denom / pdf / n_total : difference * difference / pdf / n_total;
		const float gradient = type == LossType::RelativeL2 ? 2 * difference / denom / pdf : 2 * difference / pdf;
		return to_half_rn(loss_scale * gradient / n_total);
	}
	if constexpr (GENERAL) return loss_element_general(type, prediction, target, pdf, n_total, loss_scale, value);
}

}  // namespace tcnn_hip
