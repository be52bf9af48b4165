// activation_device.h -- the activations FullyFusedMLP offers for its hidden layers (fully_fused_mlp.cu:690-697) and,
// together with None, for the output layer; reference arithmetic: common_device.h:108-186 (forward) and :363-418
// (backward, expressed through the POST-activation value -- which is all that is stored).
#pragma once
#include "tcnn_device.h"
#if defined(TCNN_HOST_EMU)
#include <math.h>
#endif

namespace tcnn_hip {

enum class Activation : int { None = 0, ReLU = 1, LeakyReLU = 2, Exponential = 3, Sigmoid = 4, Squareplus = 5, Softplus = 6, Tanh = 7 };
constexpr float K_ACT = 10.0f;  // common_device.h:108

// activation of a pre-activation accumulator (fp32); the caller rounds the result to fp16 once
TCNN_DEVICE float act_forward(uint32_t act, float x) {
	switch ((Activation)act) {
		case Activation::ReLU: return x > 0.0f ? x : 0.0f;
		case Activation::LeakyReLU: return x * (x > 0.0f ? 1.0f : 0.01f);
		case Activation::Exponential: return expf(x);
		case Activation::Sigmoid: return 1.0f / (1.0f + expf(-x));
		case Activation::Squareplus: {
			const float y = x * K_ACT;
			return 0.5f * (y + sqrtf(y * y + 4.0f)) / K_ACT;
		}
		case Activation::Softplus: return logf(expf(x * K_ACT) + 1.0f) / K_ACT;
		case Activation::Tanh: return tanhf(x);
		default: return x;
	}
}

// dL/d(pre-activation) = v * f'(x) with f' written in terms of the stored fp16 post-activation value; the factor is
// rounded to fp16 like the reference's (T)(...) before the multiply.  ReLU keeps the select form (exact, no -0).
TCNN_DEVICE float act_backward(uint32_t act, float v, half_t forward_value) {
	const float y = (float)forward_value;
	float factor;
	switch ((Activation)act) {
		case Activation::ReLU: return forward_value > (half_t)0.0f ? v : 0.0f;
		case Activation::LeakyReLU: factor = forward_value > (half_t)0.0f ? 1.0f : 0.01f; break;
		case Activation::Exponential: factor = y; break;
		case Activation::Sigmoid: factor = (float)(half_t)(y * (float)(half_t)(1.0f - y)); break;  // common_device.h:389
		case Activation::Squareplus: {
			const float t = y * K_ACT;
			factor = t * t / (t * t + 1.0f);
			break;
		}
		case Activation::Softplus: factor = 1.0f - expf(-y * K_ACT); break;
		case Activation::Tanh: factor = 1.0f - y * y; break;
		default: return v;
	}
	return v * (float)(half_t)factor;
}

}  // namespace tcnn_hip
