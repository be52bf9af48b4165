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

// The fused kernels evaluate activations at dozens of unrolled sites: ReLU / None stay inline, everything else is ONE
// out-of-line copy (inlining the transcendental bodies everywhere doubled the training kernel's run time through
// instruction-cache misses).
#if defined(TCNN_HOST_EMU)
#define TCNN_DEVICE_NOINLINE inline
#else
#define TCNN_DEVICE_NOINLINE __device__ __attribute__((noinline))
#endif

// activation of a pre-activation accumulator (fp32); the caller rounds the result to fp16 once
TCNN_DEVICE_NOINLINE float act_forward_general(uint32_t act, float x) {
	switch ((Activation)act) {
		case Activation::ReLU: return x > 0.0f ? x : 0.0f;
		case Activation::LeakyReLU: return x * (x > 0.0f ? 1.0f : (float)(half_t)0.01f);  // common_device.h:127: the slope is a (T) constant
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

// GENERAL == false: the kernel instance only ever sees ReLU / None (no call sites at all in its body)
template <bool GENERAL>
TCNN_DEVICE float act_forward(uint32_t act, float x) {
	if (act == (uint32_t)Activation::ReLU) return x > 0.0f ? x : 0.0f;
	if constexpr (GENERAL) {
		if (act != (uint32_t)Activation::None) return act_forward_general(act, x);
	}
	return x;
}
TCNN_HOST_DEVICE bool act_is_simple(uint32_t act) { return act == (uint32_t)Activation::ReLU || act == (uint32_t)Activation::None; }

// dL/d(pre-activation) = v * f'(x) with f' written in terms of the stored fp16 post-activation value; the factor is
// rounded to fp16 like the reference's (T)(...) before the multiply.  ReLU keeps the select form (exact, no -0).
TCNN_DEVICE_NOINLINE float act_backward_general(uint32_t act, float v, half_t forward_value) {
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

template <bool GENERAL>
TCNN_DEVICE float act_backward(uint32_t act, float v, half_t forward_value) {
	if (act == (uint32_t)Activation::ReLU) return forward_value > (half_t)0.0f ? v : 0.0f;
	if constexpr (GENERAL) {
		if (act != (uint32_t)Activation::None) return act_backward_general(act, v, forward_value);
	}
	return v;
}

}  // namespace tcnn_hip
