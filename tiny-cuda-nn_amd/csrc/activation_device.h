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

// An MFMA result that is about to be read on the far side of a branch: hipcc pads the wait states between an MFMA and the first reader of its
// result in straight-line code, but can miss them on a path that leaves the MFMA through a TAKEN branch (scripts/check_mfma_branch_hazard.py,
// profiles/r03_mfma_branch_hazard.txt: stale accumulators, no fault, no message).  This spends the wait states (12 >= the 8-pass shapes' need)
// in front of the branch, tied to the fragment so that it cannot move above the MFMA.
TCNN_DEVICE void mfma_settle(f4& fragment) {
#if !defined(TCNN_HOST_EMU)
	// (between scheduling barriers: the wait states have to stand directly in front of the branch -- tied to the fragment alone, the scheduler
	// moved independent MFMAs of the surrounding loop between them and the branch)
	__builtin_amdgcn_sched_barrier(0);
	asm volatile("s_nop 7\n\ts_nop 3" : "+v"(fragment));
	__builtin_amdgcn_sched_barrier(0);
#endif
}

// ---- four values at a time: what every call site of the fused kernels has in hand (an MFMA accumulator fragment).  The out-of-line body is
// entered ONCE per fragment with the switch outside the four evaluations: per-element calls cost the 128 x 4 Sigmoid network 0.45 ms of
// call overhead and spills around the calls per step (0.667 against 0.218 ms with ReLU; scripts/exp_spilling_instances.py, round 5).
// Same arithmetic per element as the scalar forms above, same rounding points.
TCNN_DEVICE_NOINLINE f4 act_forward4_general(uint32_t act, f4 x) {
	f4 y;
	switch ((Activation)act) {
		case Activation::LeakyReLU:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) y[j] = x[j] * (x[j] > 0.0f ? 1.0f : (float)(half_t)0.01f);
			break;
		case Activation::Exponential:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) y[j] = expf(x[j]);
			break;
		case Activation::Sigmoid:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) y[j] = 1.0f / (1.0f + expf(-x[j]));
			break;
		case Activation::Squareplus:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) {
				const float t = x[j] * K_ACT;
				y[j] = 0.5f * (t + sqrtf(t * t + 4.0f)) / K_ACT;
			}
			break;
		case Activation::Softplus:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) y[j] = logf(expf(x[j] * K_ACT) + 1.0f) / K_ACT;
			break;
		case Activation::Tanh:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) y[j] = tanhf(x[j]);
			break;
		case Activation::ReLU:
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) y[j] = x[j] > 0.0f ? x[j] : 0.0f;
			break;
		default: y = x; break;
	}
	return y;
}
// activation of an accumulator fragment, rounded to the 16-bit type once per value
template <bool GENERAL>
TCNN_DEVICE h4 act_forward4(uint32_t act, f4 x) {
	if constexpr (GENERAL) {
		mfma_settle(x);  // the fragment is read on both sides of the branch below
		if (act != (uint32_t)Activation::None && act != (uint32_t)Activation::ReLU) x = act_forward4_general(act, x);
	}
	// ReLU as a SELECT on a uniform flag, not as a branch around four selects: these fragments come straight out of an MFMA, and a taken
	// branch between an MFMA and the first reader of its result is where hipcc under-pads the wait states
	// (scripts/check_mfma_branch_hazard.py, profiles/r03_mfma_branch_hazard.txt)
	const bool relu = act == (uint32_t)Activation::ReLU;
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) x[j] = (relu && !(x[j] > 0.0f)) ? 0.0f : x[j];
	return h4{(half_t)x[0], (half_t)x[1], (half_t)x[2], (half_t)x[3]};
}
TCNN_DEVICE_NOINLINE f4 act_backward4_general(uint32_t act, f4 v, h4 forward_value) {
	f4 out;
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) {
		const float y = (float)forward_value[j];
		float factor = 1.0f;
		bool plain = false;
		switch ((Activation)act) {  // (uniform: the compiler hoists it out of the unrolled loop)
			case Activation::ReLU: out[j] = forward_value[j] > (half_t)0.0f ? v[j] : 0.0f; plain = true; break;
			case Activation::LeakyReLU: factor = forward_value[j] > (half_t)0.0f ? 1.0f : 0.01f; break;
			case Activation::Exponential: factor = y; break;
			case Activation::Sigmoid: factor = (float)(half_t)(y * (float)(half_t)(1.0f - y)); break;
			case Activation::Squareplus: {
				const float t = y * K_ACT;
				factor = t * t / (t * t + 1.0f);
				break;
			}
			case Activation::Softplus: factor = 1.0f - expf(-y * K_ACT); break;
			case Activation::Tanh: factor = 1.0f - y * y; break;
			default: out[j] = v[j]; plain = true; break;
		}
		if (!plain) out[j] = v[j] * (float)(half_t)factor;
	}
	return out;
}
template <bool GENERAL>
TCNN_DEVICE h4 act_backward4(uint32_t act, f4 v, h4 forward_value) {
	if constexpr (GENERAL) {
		mfma_settle(v);
		if (act != (uint32_t)Activation::None && act != (uint32_t)Activation::ReLU) v = act_backward4_general(act, v, forward_value);
	}
	const bool relu = act == (uint32_t)Activation::ReLU;  // (a select, see act_forward4)
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) v[j] = (relu && !(forward_value[j] > (half_t)0.0f)) ? 0.0f : v[j];
	return h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
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
