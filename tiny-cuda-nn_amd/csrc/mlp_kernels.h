// mlp_kernels.h -- fully fused MLP (16/32/64/128 wide, fp16 storage, fp32 MFMA accumulation) on gfx950.
//
// Restates the BEHAVIOUR of reference src/fully_fused_mlp.cu:499-837 (kernel_mlp_fused,
// kernel_mlp_fused_backward and the four CUTLASS call sites :786,:820,:829,:835) -- not its tiling.
// MI355X-first structure (DESIGN.md "MLP kernels"):
//   * wave64 + v_mfma_f32_16x16x32_f16; a workgroup = WIDTH/16 waves, wave w owns 16 neurons.
//   * forward:  D[neuron][sample] = W * X   (weights are the A operand in their natural row-major
//     layout, activations the B operand straight out of a [sample][feature] LDS tile); the
//     accumulator fragment IS the next layer's [sample][4 features] store -- no shuffles.
//   * backward: D[sample][neuron] = dA * W  (pre-transposed weights as B operand); the accumulator
//     fragment (neuron per lane, 4 samples per register) is directly the A operand of the
//     weight-gradient MFMA  dW[out][in] = sum_s dA[out][s] * A_prev[in][s], whose B operand comes
//     from feature-major (transposed) LDS tiles of the forward activations.
//   * weight gradients are accumulated in fp32 MFMA accumulators held in registers across ALL
//     sample tiles a persistent workgroup processes, written once per workgroup as fp32 partials
//     and summed by a tiny deterministic finalize kernel (replaces the reference's 64-way split-K
//     CUTLASS GEMMs and their aux streams).
// Data layout at the kernel boundary:
//   input / dL_dinput : half, feature-major  [in_width][n]   (the reference's SoA, grid.h:1070)
//   output / dL_doutput: half, sample-major  [n][16]         (the reference's CM padded output)
//   hidden (saved)    : half, [n_hidden][n][WIDTH] post-activation (fully_fused_mlp.cu:841-854)
#pragma once
#include "activation_device.h"
#include "loss_device.h"
#include "tcnn_device.h"

namespace tcnn_hip {


struct MlpMeta {
	uint32_t in_width;          // multiple of 16
	uint32_t width;             // 16 / 32 / 64 / 128
	uint32_t padded_out;        // multiple of 16, <= MLP_MAX_OUT_WIDTH
	uint32_t n_hidden_matmuls;  // n_hidden_layers - 1
	uint32_t activation;        // Activation of the hidden layers
	uint32_t output_activation; // Activation of the output layer (default None)
	TCNN_HOST_DEVICE uint32_t n_params() const { return width * in_width + n_hidden_matmuls * width * width + padded_out * width; }
};

// position of parameter i (natural layout: [W][IN] | HM x [W][W] | [OUTP][W], row-major) in the transposed scratch the
// backward kernels read ([IN][W] | HM x [W][W]^T | [W][OUTP])
TCNN_HOST_DEVICE uint32_t mlp_transposed_index(const MlpMeta& m, uint32_t i) {
	const uint32_t W = m.width, IN = m.in_width;
	const uint32_t n_in = W * IN, n_hid = m.n_hidden_matmuls * W * W;
	if (i < n_in) {
		const uint32_t o = i / IN, k = i % IN;
		return k * W + o;
	}
	if (i < n_in + n_hid) {
		const uint32_t local = i - n_in, j = local / (W * W), e = local % (W * W);
		const uint32_t o = e / W, k = e % W;
		return n_in + j * W * W + k * W + o;
	}
	const uint32_t local = i - n_in - n_hid;
	const uint32_t o = local / W, k = local % W;
	return n_in + n_hid + k * m.padded_out + o;
}

enum class SlabOrder : uint32_t { Params = 0, WaveRegisters = 1 };
// Order of the fp32 weight-gradient slabs a training kernel writes: the parameter layout, or k_mlp_train_wave's accumulator
// registers as they lie (coalesced 16-byte stores; mlp_train_wave.hip).  The parameter a slab position of that kernel stands for
// (its tile order and accumulator lane map): position = 256 * tile + 4 * lane + r, tiles per neuron block b: FB input tiles, HM * NB hidden tiles, one output tile.
TCNN_HOST_DEVICE uint32_t mlp_wave_slab_param(const MlpMeta& m, uint32_t position) {
	const uint32_t width = m.width, in_width = m.in_width, n_hidden_matmuls = m.n_hidden_matmuls;
	const uint32_t NB = width / 16u, FB = in_width / 16u, tiles_per_block = FB + n_hidden_matmuls * NB + 1u;
	const uint32_t t = position >> 8, lane = (position >> 2) & 63u, r = position & 3u, lr = lane & 15u, g = lane >> 4;
	const uint32_t b = t / tiles_per_block, u = t % tiles_per_block;
	auto perm = [](uint32_t blk, uint32_t row) { return 32u * (blk >> 1) + 8u * (row >> 2) + 4u * (blk & 1u) + (row & 3u); };  // perm32
	if (u < FB) return perm(b, 4u * g + r) * in_width + perm(u, lr);
	const uint32_t off_hid = width * in_width;
	if (u < FB + n_hidden_matmuls * NB) {
		const uint32_t e = u - FB, j = e / NB, i = e % NB;
		return off_hid + j * width * width + perm(b, 4u * g + r) * width + perm(i, lr);
	}
	return off_hid + n_hidden_matmuls * width * width + (4u * r + g) * width + perm(b, lr);  // accumulator row 4g+r <-> output 4r+g
}

constexpr uint32_t MLP_MAX_HIDDEN_MATMULS_TRAIN = 3;  // the register-resident backward / training kernels are instantiated for 0..3
constexpr uint32_t MLP_MAX_IN_WIDTH = 128;
constexpr uint32_t MLP_MAX_OUT_WIDTH = 128;  // padded; more than 16 outputs train through the layer-by-layer backward

// Forward.  hidden == nullptr -> inference (nothing saved).
void mlp_forward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* input, half_t* hidden,
                 half_t* output);

// weights -> transposed scratch ([in_width][W] | HM x [W][W] | [W][16]); n_params halves
void mlp_transpose_weights(hipStream_t stream, const MlpMeta& m, const half_t* params, half_t* params_t);

// Number of fp32 partial gradient slabs mlp_backward writes (== its grid size) for batch n.
uint32_t mlp_backward_n_partials(const MlpMeta& m, uint32_t n);

// dL/d(pre-activation of the output layer) from dL/doutput and the (post-activation) output; only needed when the
// output activation is not None (fully_fused_mlp.cu:760-763, common_device.h:923-932).  [n][16] each.
void mlp_output_activation_backward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* output, const half_t* dL_doutput,
                                    half_t* dL_dpreact);

// Backward.  dL_doutput is the gradient w.r.t. the output layer's PRE-activation (== dL/doutput when the output
// activation is None).  params_t from mlp_transpose_weights.  dL_dinput may be null.  partials: fp32
// [mlp_backward_n_partials][n_params] or null (GradientMode::Ignore).
// Networks with more than MLP_MAX_HIDDEN_MATMULS_TRAIN + 1 hidden layers run a layer-by-layer formulation that
// needs `workspace` (mlp_backward_workspace_bytes, 0 for the shallower ones).
size_t mlp_backward_workspace_bytes(const MlpMeta& m, uint32_t n);
void mlp_backward(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params_t, const half_t* input, const half_t* hidden,
                  const half_t* dL_doutput, half_t* dL_dinput, float* partials, void* workspace = nullptr);

// Fused training pass of Trainer::training_step (trainer.h:254-357): forward + loss + backward per sample tile in one
// kernel -- the hidden activations stay in LDS, prediction / dL_doutput are written for the caller's ForwardContext
// (either may be null), block_sums receives mlp_backward_n_partials() partial loss sums.  Bit-identical to
// mlp_forward -> loss_evaluate -> mlp_backward.  Widths 16/32/64 (and 128 through mlp_train_wide), up to 4 hidden layers (mlp_train_supported).
struct MlpLossArgs {
	LossType type;
	const float* targets;   // fp32 [n][dims]
	const float* data_pdf;  // nullable, like targets
	uint32_t dims;          // unpadded output width
	float loss_scale;
	uint32_t n_total;       // elements the mean runs over (global batch x dims)
	// Not null: no loss is evaluated -- dL/doutput [n][16] comes from the caller (Trainer's external_dL_dy, trainer.h:124-128; the
	// backward pass of a module, which RECOMPUTES the forward pass from the encoded input instead of reading saved activations:
	// the three matrix products cost less than writing and re-reading every hidden activation).  targets / data_pdf unused.
	const half_t* external_dL_doutput = nullptr;
};
// The network kernel reads the caller's fp32 matrix itself: an Identity encoding without padding (encodings/identity.h:46-66) is
// `(T)(x * scale + offset)` per element, and the register-resident training kernel loads its input one 32-sample strip ahead anyway -- so it
// can load x[i][k] (sample-major, x[i * in_width + k]) instead of the encoded half matrix, convert with k_identity_forward's arithmetic, and, if
// `enc_out` is given, leave the encoded input [in_width][n] there for the caller's context (what encoding_forward would have written).
// Saves the transpose kernel and its 100 MB of traffic on BASELINE configs[1].  Same bits as the two-kernel path.
struct MlpF32Input {
	const float* x = nullptr;
	float scale = 1.0f, offset = 0.0f;
	half_t* enc_out = nullptr;
};
bool mlp_train_f32_input_supported(const MlpMeta& m, uint32_t n, LossType loss);
bool mlp_train_supported(const MlpMeta& m);
// The register-resident instances of the training pass (mlp_train_wave.hip): one wavefront per strip of 32 samples, for
// 32 inputs x {64 neurons, 1-2 hidden layers | 32 neurons, 1-3 hidden layers} x 16 padded outputs, ReLU / None activations,
// (Relative)L2 loss.  mlp_train() picks them
// when available; same contract as mlp_train().
bool mlp_train_wave_supported(const MlpMeta& m, uint32_t n, LossType loss);
uint32_t mlp_train_wave_n_partials(const MlpMeta& m, uint32_t n);
void mlp_train_wave(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                    const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums,
                    const MlpF32Input* f32_input = nullptr);

// The fused training pass of 128-neuron networks (mlp_train_wide.hip): hidden weights resident in LDS (one copy, the backward
// operands come out of it through the hardware transpose read), sample-major activation tiles of 32 samples, weight gradients
// in registers.  32 or 64 inputs, up to 4 hidden layers, 16 padded outputs, any activation / element-wise loss.  mlp_train()
// picks it.
bool mlp_train_wide_supported(const MlpMeta& m, uint32_t n);
uint32_t mlp_train_wide_n_partials(uint32_t n);
void mlp_train_wide(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
                    const MlpLossArgs& la, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums);

// Inference (no saved activations) of the same shapes, also with 64 inputs, with up to 3 (64 neurons) / 4 (32 neurons) hidden layers: the forward
// half of the register-resident kernel.  mlp_forward() picks it when `hidden` is null.
bool mlp_infer_wave_supported(const MlpMeta& m, uint32_t n);
// where an inference call wants its result as the caller's fp32 matrix (object.h:269-270: the first `dims` of the padded outputs, cast):
// element (sample i, output j) at out[i * stride_i + j * stride_j].  The kernel then stores that instead of the padded 16-bit matrix.
struct MlpF32Output {
	float* out = nullptr;
	uint32_t dims = 0, stride_i = 0, stride_j = 0;
};
void mlp_infer_wave(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* input, half_t* output, const MlpF32Output& f32 = {},
                    const MlpF32Input* f32_input = nullptr);  // f32_input (16-byte aligned, mlp_infer_f32_input_supported): `input` is ignored
bool mlp_infer_f32_input_supported(const MlpMeta& m, uint32_t n);

// number of fp32 slabs / loss partial sums mlp_train() writes for this shape and batch (<= mlp_backward_n_partials)
uint32_t mlp_train_n_partials(const MlpMeta& m, uint32_t n, LossType loss);
SlabOrder mlp_train(hipStream_t stream, const MlpMeta& m, uint32_t n, const half_t* params, const half_t* params_t, const half_t* input,
               const MlpLossArgs& loss, half_t* output, half_t* dL_doutput, half_t* dL_dinput, float* partials, float* block_sums,
               const MlpF32Input* f32_input = nullptr);  // f32_input: only where mlp_train_f32_input_supported (then `input` is ignored)

// grads[i] = (accumulate ? grads[i] : 0) + sum_b partials[b][i]   (fully_fused_mlp.cu:770 beta)
// `order`: how the slabs are laid out -- what mlp_train() returned for them (mlp_backward writes parameter order)
void mlp_finalize_gradients(hipStream_t stream, const MlpMeta& m, uint32_t n_partials, const float* partials, half_t* grads, bool accumulate,
                            SlabOrder order = SlabOrder::Params);

}  // namespace tcnn_hip
