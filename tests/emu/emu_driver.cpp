// emu_driver.cpp -- builds the gfx950 kernel SOURCES for the host under the SIMT emulator
// (hip_emu.h) and exposes them to pytest through a flat C interface (host pointers only).
// TEST INFRASTRUCTURE ONLY -- see hip_emu.h.
#define TCNN_HOST_EMU 1
#include "../../tiny-cuda-nn_amd/csrc/grid_kernels.hip"
#include "../../tiny-cuda-nn_amd/csrc/elementwise_kernels.hip"
#include "../../tiny-cuda-nn_amd/csrc/mlp_kernels.hip"
#include "../../tiny-cuda-nn_amd/csrc/mlp_train_wave.hip"
#include "../../tiny-cuda-nn_amd/csrc/mlp_train_wide.hip"

using namespace tcnn_hip;

extern "C" {

struct EmuGrid {
	uint32_t n_dims, n_levels, n_feat, grid_type, interp;
	float max_level;
	const uint32_t* offset;      // [L+1]
	const float* scale;          // [L]
	const uint32_t* resolution;  // [L]
};

static GridMeta make_meta(const EmuGrid* e) {
	GridMeta m = {};
	m.n_dims = e->n_dims;
	m.n_levels = e->n_levels;
	m.n_feat = e->n_feat;
	m.grid_type = e->grid_type;
	m.interp = e->interp & 0xFFu;
	m.stochastic = (e->interp >> 8) & 1u;  // emu.py packs the flag into the interpolation word
	m.max_level = e->max_level;
	for (uint32_t l = 0; l <= e->n_levels; ++l) m.offset[l] = e->offset[l];
	for (uint32_t l = 0; l < e->n_levels; ++l) {
		m.scale[l] = e->scale[l];
		m.resolution[l] = e->resolution[l];
	}
	return m;
}

// positions: [n][D] ; out: feature-major [L*F][n] when soa != 0 else sample-major [n][out_stride]
int emu_grid_forward(const EmuGrid* e, const float* positions, uint32_t n, const uint16_t* params, uint16_t* out, int soa,
                     uint32_t out_stride, float* dy_dx) {
	try {
		GridIO io = {positions, e->n_dims, 1, n, soa ? n : 1u, soa ? 1u : out_stride};
		grid_forward(nullptr, make_meta(e), io, (const half_t*)params, (half_t*)out, dy_dx);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_grid_forward: %s\n", ex.what());
		return 1;
	}
	return 0;
}

// The tiled gather's work plan (make_forward_plan, host code of the library): segments[x][k] = {level, tile_begin, tile_end} for XCD x.
// out: [8][FWD_MAX_SEGMENTS][3] u32, n_segments: [8]; returns the sample tiles per level (0 on failure).
uint32_t emu_grid_forward_plan(const EmuGrid* e, uint32_t n, uint32_t tile_samples, uint32_t* n_segments, uint32_t* out, uint32_t max_segments) {
	try {
		const ForwardPlan plan = make_forward_plan(make_meta(e), n, tile_samples);
		for (uint32_t x = 0; x < 8; ++x) {
			n_segments[x] = plan.n_segments[x];
			for (uint32_t k = 0; k < plan.n_segments[x] && k < max_segments; ++k) {
				out[(x * max_segments + k) * 3 + 0] = plan.segments[x][k].level;
				out[(x * max_segments + k) * 3 + 1] = plan.segments[x][k].tile_begin;
				out[(x * max_segments + k) * 3 + 2] = plan.segments[x][k].tile_end;
			}
		}
		return plan.tiles;
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_grid_forward_plan: %s\n", ex.what());
		return 0;
	}
}

void emu_set_grid_owner_mode(int mode) { grid_owner_mode() = mode; }
// slices the packed owner kernel finished from its packed table / redid with 64 bits per value, since the last call
void emu_grid_owner_stats(unsigned long* packed_wide) {
	for (int i = 0; i < 2; ++i) {
		packed_wide[i] = owner_slice_stats[i];
		owner_slice_stats[i] = 0;
	}
}

int emu_grid_backward(const EmuGrid* e, const float* positions, uint32_t n, const uint16_t* dL_dy, int soa, uint32_t dy_stride,
                      uint16_t* grad_half, int accumulate, int mode, uint32_t lds_budget) {
	try {
		GridIO io = {positions, e->n_dims, 1, n, soa ? n : 1u, soa ? 1u : dy_stride};
		const GridMeta meta = make_meta(e);
		GridBackwardWorkspace ws = grid_backward_workspace_size(meta, n, (GridBackwardMode)mode, lds_budget);
		std::vector<unsigned char> scratch(ws.scratch_bytes + 16, 0xCD);  // garbage-filled: the call must not rely on clean scratch
		static std::vector<uint32_t> counters;                            // zero on entry, zero on exit (checked below)
		if (counters.size() < ws.n_counters) counters.assign(ws.n_counters, 0u);
		if (ws.scratch_bytes) {
			ws.scratch = scratch.data();
			ws.counters = counters.data();
		}
		grid_backward(nullptr, meta, io, (const half_t*)dL_dy, (half_t*)grad_half, accumulate != 0, (GridBackwardMode)mode, lds_budget, ws);
		for (uint32_t c : counters) {
			if (c != 0u) throw std::runtime_error("bucketed backward left a non-zero counter behind");
		}
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_grid_backward: %s\n", ex.what());
		return 1;
	}
	return 0;
}

// second order (grid.h:352-655): any of grad_half / dL_ddLdy / dL_dx may be null.  dL_dy and dL_ddLdy feature-major [K][n].
int emu_grid_backward_backward(const EmuGrid* e, const float* positions, const float* ddx, uint32_t n, const uint16_t* dL_dy, const uint16_t* params,
                               const float* dy_dx, uint16_t* grad_half, uint16_t* dL_ddLdy, float* dL_dx) {
	try {
		const GridMeta meta = make_meta(e);
		GridIO io = {positions, e->n_dims, 1, n, n, 1u};
		io.ddx = ddx;
		io.ddx_stride_i = e->n_dims;
		io.ddx_stride_d = 1;
		if (grad_half) {
			GridBackwardWorkspace ws = grid_backward_workspace_size(meta, n, GridBackwardMode::Bucketed, 0);
			std::vector<unsigned char> scratch(ws.scratch_bytes + 16, 0xCD);
			std::vector<uint32_t> counters(ws.n_counters + 1, 0u);
			ws.scratch = scratch.data();
			ws.counters = counters.data();
			grid_backward(nullptr, meta, io, (const half_t*)dL_dy, (half_t*)grad_half, false, GridBackwardMode::Bucketed, 0, ws);
		}
		if (dL_ddLdy) grid_backward_backward_dLdoutput(nullptr, e->n_dims, e->n_levels * e->n_feat, 0, io, dy_dx, (half_t*)dL_ddLdy);
		if (dL_dx) grid_backward_backward_input(nullptr, meta, io, (const half_t*)dL_dy, (const half_t*)params, dL_dx, e->n_dims, 1);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_grid_backward_backward: %s\n", ex.what());
		return 1;
	}
	return 0;
}

int emu_grid_backward_input(const EmuGrid* e, uint32_t n, const uint16_t* dL_dy, int soa, uint32_t dy_stride, const float* dy_dx,
                            float* dL_dx) {
	GridIO io = {nullptr, e->n_dims, 1, n, soa ? n : 1u, soa ? 1u : dy_stride};
	grid_backward_input(nullptr, e->n_dims, e->n_levels * e->n_feat, io, (const half_t*)dL_dy, dy_dx, dL_dx, e->n_dims, 1);
	return 0;
}

// GridEncodingTemplated<float>: fp32 parameters / features / gradients (sample-major [n][stride] features and gradients)
int emu_grid_forward_f32(const EmuGrid* e, const float* positions, uint32_t n, const float* params, float* out, uint32_t out_stride, float* dy_dx) {
	try {
		GridIO io = {positions, e->n_dims, 1, n, 1u, out_stride};
		grid_forward_f32(nullptr, make_meta(e), io, params, out, dy_dx);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_grid_forward_f32: %s\n", ex.what());
		return 1;
	}
	return 0;
}
int emu_grid_backward_f32(const EmuGrid* e, const float* positions, uint32_t n, const float* dL_dy, uint32_t dy_stride, float* grad, int accumulate) {
	try {
		GridIO io = {positions, e->n_dims, 1, n, 1u, dy_stride};
		grid_backward_f32(nullptr, make_meta(e), io, dL_dy, grad, accumulate != 0);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_grid_backward_f32: %s\n", ex.what());
		return 1;
	}
	return 0;
}
int emu_grid_backward_input_f32(const EmuGrid* e, uint32_t n, const float* dL_dy, uint32_t dy_stride, const float* dy_dx, float* dL_dx) {
	GridIO io = {nullptr, e->n_dims, 1, n, 1u, dy_stride};
	grid_backward_input_f32(nullptr, e->n_dims, e->n_levels * e->n_feat, io, dL_dy, dy_dx, dL_dx, e->n_dims, 1);
	return 0;
}

int emu_grid_indices(const EmuGrid* e, const float* positions, uint32_t n, uint32_t* indices) {
	GridIO io = {positions, e->n_dims, 1, n, n, 1};
	grid_indices(nullptr, make_meta(e), io, indices);
	return 0;
}

struct EmuMlp {
	uint32_t in_width, width, padded_out, n_hidden_matmuls, activation, output_activation;
};
static MlpMeta make_mlp(const EmuMlp* e) {
	return MlpMeta{e->in_width, e->width, e->padded_out, e->n_hidden_matmuls, e->activation, e->output_activation};
}

int emu_mlp_forward(const EmuMlp* e, uint32_t n, const uint16_t* params, const uint16_t* input_soa, uint16_t* hidden, uint16_t* output) {
	try {
		mlp_forward(nullptr, make_mlp(e), n, (const half_t*)params, (const half_t*)input_soa, (half_t*)hidden, (half_t*)output);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_mlp_forward: %s\n", ex.what());
		return 1;
	}
	return 0;
}

// the register-resident inference kernel reading an unpadded Identity encoding's fp32 input itself (MlpF32Input); out_half [n][16] or, if
// out_f32 is given, the caller's fp32 matrix [n][dims].  2: no instance.
int emu_mlp_infer_f32_input(const EmuMlp* e, uint32_t n, const uint16_t* params, const float* x, float scale, float offset, uint16_t* out_half, float* out_f32,
                            uint32_t dims) {
	try {
		const MlpMeta m = make_mlp(e);
		if (!mlp_infer_f32_input_supported(m, n)) return 2;
		MlpF32Input fin;
		fin.x = x;
		fin.scale = scale;
		fin.offset = offset;
		MlpF32Output f32;
		if (out_f32) f32 = {out_f32, dims, dims, 1u};
		mlp_infer_wave(nullptr, m, n, (const half_t*)params, nullptr, (half_t*)out_half, f32, &fin);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_mlp_infer_f32_input: %s\n", ex.what());
		return 1;
	}
	return 0;
}

// grads: half [n_params] (Overwrite unless accumulate); scratch sizes are handled here.
int emu_mlp_backward(const EmuMlp* e, uint32_t n, const uint16_t* params, const uint16_t* input_soa, const uint16_t* hidden,
                     const uint16_t* dL_doutput, uint16_t* dL_dinput_soa, uint16_t* grads, int accumulate, const uint16_t* output) {
	try {
		const MlpMeta m = make_mlp(e);
		std::vector<uint16_t> params_t(m.n_params());
		mlp_transpose_weights(nullptr, m, (const half_t*)params, (half_t*)params_t.data());
		const uint32_t np = mlp_backward_n_partials(m, n);
		std::vector<float> partials(grads ? (size_t)np * m.n_params() : 0, -12345.0f);
		std::vector<uint16_t> dpre;
		if (m.output_activation != (uint32_t)Activation::None) {
			if (!output) throw std::runtime_error("output required");
			dpre.resize((size_t)n * m.padded_out);
			mlp_output_activation_backward(nullptr, m, n, (const half_t*)output, (const half_t*)dL_doutput, (half_t*)dpre.data());
			dL_doutput = dpre.data();
		}
		std::vector<unsigned char> deep(mlp_backward_workspace_bytes(m, n) + 16, 0xCD);
		mlp_backward(nullptr, m, n, (const half_t*)params_t.data(), (const half_t*)input_soa, (const half_t*)hidden,
		             (const half_t*)dL_doutput, (half_t*)dL_dinput_soa, grads ? partials.data() : nullptr, deep.data());
		if (grads) mlp_finalize_gradients(nullptr, m, np, partials.data(), (half_t*)grads, accumulate != 0);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_mlp_backward: %s\n", ex.what());
		return 1;
	}
	return 0;
}

// fused forward + loss + backward; returns prediction, dL_doutput, dL_dinput, grads (Overwrite) and the loss sum
int emu_mlp_train(const EmuMlp* e, uint32_t n, const uint16_t* params, const uint16_t* input_soa, int loss_type, const float* target,
                  const float* data_pdf, uint32_t dims, float loss_scale, uint32_t n_total, uint16_t* output, uint16_t* dL_doutput,
                  uint16_t* dL_dinput_soa, uint16_t* grads, float* loss_sum, const uint16_t* external_dL_doutput) {
	try {
		const MlpMeta m = make_mlp(e);
		if (!mlp_train_supported(m)) return 2;
		std::vector<uint16_t> params_t(m.n_params());
		mlp_transpose_weights(nullptr, m, (const half_t*)params, (half_t*)params_t.data());
		if (external_dL_doutput) loss_type = (int)LossType::L2;  // no loss is evaluated
		const uint32_t np = mlp_train_n_partials(m, n, (LossType)loss_type);
		std::vector<float> partials(grads ? (size_t)np * m.n_params() : 0, -12345.0f), block_sums(np, -777.0f), ws(1024);
		MlpLossArgs la = {(LossType)loss_type, target, data_pdf, dims, loss_scale, n_total};
		la.external_dL_doutput = (const half_t*)external_dL_doutput;
		const SlabOrder order = mlp_train(nullptr, m, n, (const half_t*)params, (const half_t*)params_t.data(), (const half_t*)input_soa, la, (half_t*)output,
		          (half_t*)dL_doutput, (half_t*)dL_dinput_soa, grads ? partials.data() : nullptr, block_sums.data());
		if (grads) mlp_finalize_gradients(nullptr, m, np, partials.data(), (half_t*)grads, false, order);
		if (loss_sum) reduce_sum(nullptr, block_sums.data(), block_sums.size(), ws.data(), loss_sum);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_mlp_train: %s\n", ex.what());
		return 1;
	}
	return 0;
}

// the same with the network kernel loading an unpadded Identity encoding's fp32 input itself (MlpF32Input); 2: no instance offers it.
// enc_out [in_width][n] receives the encoded input the kernel leaves behind.
int emu_mlp_train_f32_input(const EmuMlp* e, uint32_t n, const uint16_t* params, const float* x, float scale, float offset, int loss_type, const float* target,
                            const float* data_pdf, uint32_t dims, float loss_scale, uint32_t n_total, uint16_t* output, uint16_t* dL_doutput,
                            uint16_t* dL_dinput_soa, uint16_t* grads, float* loss_sum, uint16_t* enc_out) {
	try {
		const MlpMeta m = make_mlp(e);
		if (!mlp_train_supported(m) || !mlp_train_f32_input_supported(m, n, (LossType)loss_type)) return 2;
		std::vector<uint16_t> params_t(m.n_params());
		mlp_transpose_weights(nullptr, m, (const half_t*)params, (half_t*)params_t.data());
		const uint32_t np = mlp_train_n_partials(m, n, (LossType)loss_type);
		std::vector<float> partials(grads ? (size_t)np * m.n_params() : 0, -12345.0f), block_sums(np, -777.0f), ws(1024);
		MlpLossArgs la = {(LossType)loss_type, target, data_pdf, dims, loss_scale, n_total};
		MlpF32Input fin;
		fin.x = x;
		fin.scale = scale;
		fin.offset = offset;
		fin.enc_out = (half_t*)enc_out;
		const SlabOrder order = mlp_train(nullptr, m, n, (const half_t*)params, (const half_t*)params_t.data(), nullptr, la, (half_t*)output, (half_t*)dL_doutput,
		                                  (half_t*)dL_dinput_soa, grads ? partials.data() : nullptr, block_sums.data(), &fin);
		if (grads) mlp_finalize_gradients(nullptr, m, np, partials.data(), (half_t*)grads, false, order);
		if (loss_sum) reduce_sum(nullptr, block_sums.data(), block_sums.size(), ws.data(), loss_sum);
	} catch (const std::exception& ex) {
		fprintf(stderr, "emu_mlp_train_f32_input: %s\n", ex.what());
		return 1;
	}
	return 0;
}

int emu_loss(int type, uint32_t n, uint32_t stride, uint32_t dims, float loss_scale, const uint16_t* prediction, const float* target,
             const float* data_pdf, float* values, uint16_t* gradients, float* loss_sum, uint32_t n_total) {
	std::vector<float> block_sums(loss_n_blocks(n, stride));
	std::vector<float> ws(1024);
	loss_evaluate(nullptr, (LossType)type, n, stride, dims, loss_scale, (const half_t*)prediction, target, data_pdf, values,
	              (half_t*)gradients, block_sums.data(), n_total);
	if (loss_sum) reduce_sum(nullptr, block_sums.data(), block_sums.size(), ws.data(), loss_sum);
	return 0;
}

struct EmuAdam {
	float learning_rate, beta1, beta2, epsilon, l2_reg, non_matrix_l2_reg, relative_weight_decay, absolute_weight_decay;
	float weight_clipping_magnitude, gradient_clipping_magnitude, non_matrix_learning_rate_factor;
	int adabound, optimize_matrix_params, optimize_non_matrix_params, skip_zero_grad_non_matrix_params;
};

int emu_adam_flip_steps(uint32_t n, uint32_t steps_done, uint32_t* steps) {
	adam_flip_step_representation(nullptr, n, steps_done, steps);
	return 0;
}

// any AdamStepsForm -> any other, in place (deficits8: n bytes, needed when the byte form is involved)
int emu_adam_convert_steps(uint32_t n, uint32_t steps_done, uint32_t* steps, uint8_t* deficits8, int from, int to) {
	adam_convert_step_representation(nullptr, n, steps_done, steps, deficits8, from, to);
	return 0;
}

static bool g_adam_half_follows_master = false;
void emu_set_adam_half_follows_master(int on) { g_adam_half_follows_master = on != 0; }  // AdamCore::half_follows_master of the calls below
int emu_adam_step(const EmuAdam* e, uint32_t n, uint32_t n_matrix, float loss_scale, uint32_t current_step, float* w32, uint16_t* w16,
                  const uint16_t* grads, float* m1, float* m2, uint32_t* steps, int steps_form, uint8_t* deficits8) {
	AdamHyper h;
	h.learning_rate = e->learning_rate;
	h.beta1 = e->beta1;
	h.beta2 = e->beta2;
	h.epsilon = e->epsilon;
	h.l2_reg = e->l2_reg;
	h.non_matrix_l2_reg = e->non_matrix_l2_reg;
	h.relative_weight_decay = e->relative_weight_decay;
	h.absolute_weight_decay = e->absolute_weight_decay;
	h.weight_clipping_magnitude = e->weight_clipping_magnitude;
	h.gradient_clipping_magnitude = e->gradient_clipping_magnitude;
	h.non_matrix_learning_rate_factor = e->non_matrix_learning_rate_factor;
	h.adabound = e->adabound != 0;
	h.optimize_matrix_params = e->optimize_matrix_params != 0;
	h.optimize_non_matrix_params = e->optimize_non_matrix_params != 0;
	h.skip_zero_grad_non_matrix_params = e->skip_zero_grad_non_matrix_params != 0;
	adam_step(nullptr, h, n, n_matrix, loss_scale, current_step, w32, (half_t*)w16, (half_t*)grads, m1, m2, steps, nullptr, nullptr, 0, 0xFFFFFFFFu,
	          steps_form, deficits8, g_adam_half_follows_master);
	return 0;
}

// rng state is passed in and written back
int emu_generate_random_uniform(uint64_t* state, uint64_t* inc, uint64_t n, float* out, float lower, float upper) {
	Pcg32 r;
	r.state = *state;
	r.inc = *inc;
	generate_random_uniform(nullptr, r, (size_t)n, out, lower, upper);
	*state = r.state;
	*inc = r.inc;
	return 0;
}

int emu_cast_f32_to_f16(uint64_t n, const float* in, uint16_t* out) {
	cast_f32_to_f16(nullptr, (size_t)n, in, (half_t*)out);
	return 0;
}

int emu_identity_forward(uint32_t n, uint32_t n_dims, uint32_t padded, const float* in, uint16_t* out_soa) {
	identity_forward(nullptr, n, n_dims, padded, 1.0f, 0.0f, in, n_dims, 1, (half_t*)out_soa, n, 1);
	return 0;
}

int emu_frequency_forward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, uint32_t padded, const float* in, uint16_t* out_soa) {
	frequency_forward(nullptr, n, n_dims, n_frequencies, padded, in, n_dims, 1, (half_t*)out_soa, n, 1);
	return 0;
}
int emu_frequency_backward(uint32_t n, uint32_t n_dims, uint32_t n_frequencies, const uint16_t* dL_dy_soa, const float* in, float* dL_dx) {
	frequency_backward(nullptr, n, n_dims, n_frequencies, (const half_t*)dL_dy_soa, n, 1, in, n_dims, 1, dL_dx, n_dims, 1);
	return 0;
}
// one-blob encoding: feature-major output (as the network consumes it) and dL/dinput
int emu_oneblob_forward(uint32_t n, uint32_t n_dims, uint32_t n_bins, uint32_t padded, const float* in, uint16_t* out_soa) {
	oneblob_forward(nullptr, n, n_dims, n_bins, padded, in, n_dims, 1, (half_t*)out_soa, n, 1);
	return 0;
}
int emu_oneblob_backward(uint32_t n, uint32_t n_dims, uint32_t n_bins, const uint16_t* dL_dy_soa, const float* in, float* dL_dx) {
	oneblob_backward(nullptr, n, n_dims, n_bins, (const half_t*)dL_dy_soa, n, 1, in, n_dims, 1, dL_dx, n_dims, 1);
	return 0;
}

int emu_trim_and_cast(uint32_t n, uint32_t padded, uint32_t dims, const uint16_t* in, float* out) {
	trim_and_cast(nullptr, n, padded, dims, (const half_t*)in, out, dims, 1);
	return 0;
}

}  // extern "C"

// ---- trainer snapshot (host-only code of the product: csrc/snapshot_msgpack.h) --------------------------------
#include "../../tiny-cuda-nn_amd/csrc/snapshot_msgpack.h"

extern "C" {

// Encodes a snapshot from host arrays; returns the byte count (call with out == nullptr to size the buffer).
long emu_snapshot_encode(uint64_t n_params, const void* params_fp16, int with_optimizer, uint32_t current_step, float base_lr,
                         const float* m1, const float* m2, const uint32_t* steps, uint8_t* out, size_t capacity) {
	try {
		Snapshot s;
		s.n_params = n_params;
		s.params.data = (const uint8_t*)params_fp16;
		s.params.size = n_params * 2;
		s.has_optimizer = with_optimizer != 0;
		s.current_step = current_step;
		s.base_learning_rate = base_lr;
		s.first_moments.data = (const uint8_t*)m1; s.first_moments.size = n_params * 4;
		s.second_moments.data = (const uint8_t*)m2; s.second_moments.size = n_params * 4;
		s.param_steps.data = (const uint8_t*)steps; s.param_steps.size = n_params * 4;
		const size_t needed = snapshot_encoded_size(s);
		if (!out) return (long)needed;
		const std::vector<uint8_t> bytes = snapshot_encode(s);
		if (bytes.size() != needed || capacity < needed) return -2;
		std::memcpy(out, bytes.data(), bytes.size());
		return (long)needed;
	} catch (const std::exception&) { return -1; }
}

// Decodes; copies each blob into the caller's array when non-null.  meta = {n_params, has_optimizer, current_step,
// params_bytes, m1_bytes, m2_bytes, steps_bytes, params_is_float}.
int emu_snapshot_decode(const uint8_t* data, size_t size, uint64_t* meta, float* base_lr, void* params, void* m1, void* m2, void* steps) {
	try {
		const Snapshot s = snapshot_decode(data, size);
		meta[0] = s.n_params; meta[1] = s.has_optimizer; meta[2] = s.current_step; meta[3] = s.params.size;
		meta[4] = s.first_moments.size; meta[5] = s.second_moments.size; meta[6] = s.param_steps.size; meta[7] = s.params_type == "float";
		*base_lr = s.base_learning_rate;
		if (params) std::memcpy(params, s.params.data, s.params.size);
		if (m1 && s.first_moments.present()) std::memcpy(m1, s.first_moments.data, s.first_moments.size);
		if (m2 && s.second_moments.present()) std::memcpy(m2, s.second_moments.data, s.second_moments.size);
		if (steps && s.param_steps.present()) std::memcpy(steps, s.param_steps.data, s.param_steps.size);
		return 0;
	} catch (const std::exception&) { return -1; }
}

}  // extern "C"
