/*
 * tiny-cuda-nn/trainer.h -- Trainer<T, PARAMS_T, COMPUTE_T> (reference trainer.h:40-503) over the C ABI: the class a
 * native host drives the hot path with.  Same member functions, argument order and defaults as the reference's for
 * training_step / forward / backward / optimizer_step / loss, the parameter accessors, (de)serialisation, hyper-parameter
 * updates.  The model, loss and optimizer objects it is built from carry configuration; the state lives in the library.
 */
#pragma once
#include <tiny-cuda-nn/loss.h>
#include <tiny-cuda-nn/network_with_input_encoding.h>
#include <tiny-cuda-nn/optimizer.h>

namespace tcnn {

template <typename T, typename PARAMS_T, typename COMPUTE_T = PARAMS_T>
class Trainer {
	static_assert(sizeof(T) == sizeof(float) && sizeof(PARAMS_T) == 2 && sizeof(COMPUTE_T) == 2, "this build provides Trainer<float, half, half> (the reference's TCNN_HALF_PRECISION configuration)");

public:
	// trainer.h:89-95; the padded fp16 prediction and dL_doutput stay on the device
	struct ForwardContext {
		tcnn_train_context_t* ctx = nullptr;
		uint32_t padded_output_width = 0, batch_size = 0;
		ForwardContext() = default;
		ForwardContext(const ForwardContext&) = delete;
		ForwardContext& operator=(const ForwardContext&) = delete;
		~ForwardContext() {
			if (ctx) tcnn_train_context_destroy(ctx);
		}
		GPUMatrix<COMPUTE_T> output() const { return GPUMatrix<COMPUTE_T>((COMPUTE_T*)tcnn_train_context_output(ctx), padded_output_width, batch_size); }
		GPUMatrix<COMPUTE_T> dL_doutput() const { return GPUMatrix<COMPUTE_T>((COMPUTE_T*)tcnn_train_context_dL_doutput(ctx), padded_output_width, batch_size); }
	};

	// trainer.h:51-62 (default seed 1337)
	Trainer(std::shared_ptr<NetworkWithInputEncoding<PARAMS_T>> model, std::shared_ptr<Optimizer<PARAMS_T>> optimizer, std::shared_ptr<Loss<COMPUTE_T>> loss,
	        uint32_t seed = 1337)
	    : m_model(std::move(model)), m_optimizer(std::move(optimizer)), m_loss(std::move(loss)), m_h(std::make_shared<detail::ModelHandle>()) {
		json config = json::object();
		config["loss"] = m_loss->hyperparams();
		config["optimizer"] = m_optimizer->hyperparams();
		config["encoding"] = m_model->encoding_config();
		config["network"] = m_model->network_config();
		check(tcnn_create_from_config(m_model->input_width(), m_model->output_width(), json_text(config).c_str(), seed, &m_h->tm));
		m_model->bind(m_h);
		m_optimizer->bind(m_h->tm);
	}

	// trainer.h:254-264 (+ the stream-less overload :359-370)
	std::unique_ptr<ForwardContext> training_step(hipStream_t stream, const GPUMatrixDynamic<T>& input, const GPUMatrix<float>& target,
	                                               const GPUMatrix<float>* data_pdf = nullptr, bool run_optimizer = true, GPUMatrixDynamic<T>* dL_dinput = nullptr,
	                                               bool use_inference_params = false, GradientMode param_gradients_mode = GradientMode::Overwrite,
	                                               const GPUMatrix<COMPUTE_T>* external_dL_dy = nullptr) {
		auto out = new_context(input.n());
		const tcnn_matrix_t in = input.c_matrix(), tg = target.c_matrix();
		tcnn_matrix_t pdf{}, dx{}, ext{};
		if (data_pdf) pdf = data_pdf->c_matrix();
		if (dL_dinput) dx = dL_dinput->c_matrix();
		if (external_dL_dy) ext = external_dL_dy->c_matrix();
		check(tcnn_trainer_training_step_matrices(m_h->tm, stream, &in, &tg, data_pdf ? &pdf : nullptr, run_optimizer, dL_dinput ? &dx : nullptr,
		                                          use_inference_params, static_cast<int>(param_gradients_mode), external_dL_dy ? &ext : nullptr, &out->ctx));
		return out;
	}
	std::unique_ptr<ForwardContext> training_step(const GPUMatrixDynamic<T>& input, const GPUMatrix<float>& target, const GPUMatrix<float>* data_pdf = nullptr,
	                                               bool run_optimizer = true, GPUMatrixDynamic<T>* dL_dinput = nullptr, bool use_inference_params = false,
	                                               GradientMode param_gradients_mode = GradientMode::Overwrite, const GPUMatrix<COMPUTE_T>* external_dL_dy = nullptr) {
		return training_step(nullptr, input, target, data_pdf, run_optimizer, dL_dinput, use_inference_params, param_gradients_mode, external_dL_dy);
	}

	// trainer.h:97-148: `input` / `dL_dinput` are GPUMatrixDynamic of either layout, as in the reference
	std::unique_ptr<ForwardContext> forward(hipStream_t stream, const float loss_scale, const GPUMatrixDynamic<T>& input, const GPUMatrix<float>& target,
	                                         const GPUMatrix<float>* data_pdf = nullptr, bool use_inference_params = false, bool prepare_input_gradients = false,
	                                         const GPUMatrix<COMPUTE_T>* external_dL_dy = nullptr) {
		auto out = new_context(input.n());
		const tcnn_matrix_t in = input.c_matrix(), tg = target.c_matrix();
		tcnn_matrix_t pdf{}, ext{};
		if (data_pdf) pdf = data_pdf->c_matrix();
		if (external_dL_dy) ext = external_dL_dy->c_matrix();
		check(tcnn_trainer_forward_matrices(m_h->tm, stream, loss_scale, &in, &tg, data_pdf ? &pdf : nullptr, use_inference_params, prepare_input_gradients,
		                                    external_dL_dy ? &ext : nullptr, &out->ctx));
		return out;
	}
	void backward(hipStream_t stream, const ForwardContext& ctx, const GPUMatrixDynamic<T>& input, GPUMatrixDynamic<T>* dL_dinput = nullptr,
	              bool use_inference_params = false, GradientMode param_gradients_mode = GradientMode::Overwrite) {
		const tcnn_matrix_t in = input.c_matrix();
		tcnn_matrix_t dx{};
		if (dL_dinput) dx = dL_dinput->c_matrix();
		check(tcnn_trainer_backward_matrices(m_h->tm, stream, ctx.ctx, &in, dL_dinput ? &dx : nullptr, use_inference_params, static_cast<int>(param_gradients_mode)));
	}
	void optimizer_step(hipStream_t stream, float loss_scale) { check(tcnn_trainer_optimizer_step(m_h->tm, stream, loss_scale)); }  // trainer.h:150-152
	void optimizer_step(float loss_scale) { optimizer_step(nullptr, loss_scale); }

	float loss(hipStream_t stream, const ForwardContext& ctx) {  // trainer.h:372-374
		float v = 0.f;
		check(tcnn_trainer_loss(m_h->tm, stream, ctx.ctx, &v));
		return v;
	}

	// trainer.h:376-440
	size_t n_params() const { return tcnn_trainer_n_params(m_h->tm); }
	float* params_full_precision() const { return tcnn_trainer_params_full_precision(m_h->tm); }
	PARAMS_T* params() const { return static_cast<PARAMS_T*>(tcnn_trainer_params(m_h->tm)); }
	PARAMS_T* params_inference() const { return static_cast<PARAMS_T*>(tcnn_trainer_params_inference(m_h->tm)); }
	PARAMS_T* param_gradients() const { return static_cast<PARAMS_T*>(tcnn_trainer_param_gradients(m_h->tm)); }
	void set_params_full_precision(const float* params, size_t n_params, bool device_ptr = false) { check(tcnn_trainer_set_params_full_precision(m_h->tm, params, n_params, device_ptr)); }
	void set_params(const PARAMS_T* params, size_t n_params, bool device_ptr = false) { check(tcnn_trainer_set_params(m_h->tm, params, n_params, device_ptr)); }
	void update_hyperparams(const json& params) { check(tcnn_trainer_update_hyperparams(m_h->tm, json_text(params).c_str())); }
	json hyperparams() const { return json::parse(std::string(tcnn_trainer_hyperparams_json(m_h->tm))); }
	// trainer.h:442-481.  The snapshot crosses the C ABI as the MessagePack bytes of the reference's snapshot document
	// (json::to_msgpack of the reference's serialize()): serialize_msgpack / deserialize_msgpack in every build; with an nlohmann::json
	// that has binary values (3.8 or newer; the reference ships 3.10.4) serialize() returns the document itself and deserialize() takes
	// it, exactly as the reference's members do, otherwise (older nlohmann, the bundled json_mini.h) they are the byte forms.
	std::vector<uint8_t> serialize_msgpack(bool serialize_optimizer = false) const {
		size_t n = 0;
		check(tcnn_trainer_serialize(m_h->tm, serialize_optimizer, nullptr, 0, &n));
		std::vector<uint8_t> blob(n);
		check(tcnn_trainer_serialize(m_h->tm, serialize_optimizer, blob.data(), blob.size(), &n));
		return blob;
	}
	void deserialize_msgpack(const std::vector<uint8_t>& blob) { check(tcnn_trainer_deserialize(m_h->tm, blob.data(), blob.size())); }
#if defined(TCNN_JSON_HAS_BINARY)
	json serialize(bool serialize_optimizer = false) const { return json::from_msgpack(serialize_msgpack(serialize_optimizer)); }
	void deserialize(const json& data) { deserialize_msgpack(json::to_msgpack(data)); }
#else
	std::vector<uint8_t> serialize(bool serialize_optimizer = false) const { return serialize_msgpack(serialize_optimizer); }
#endif
	void deserialize(const std::vector<uint8_t>& blob) { deserialize_msgpack(blob); }

	std::shared_ptr<NetworkWithInputEncoding<PARAMS_T>> model() const { return m_model; }
	std::shared_ptr<Optimizer<PARAMS_T>> optimizer() const { return m_optimizer; }
	std::shared_ptr<Loss<COMPUTE_T>> loss_function() const { return m_loss; }

	// data parallelism (no reference counterpart; see tcnn_hip.h)
	void set_global_batch_size(uint64_t n) { check(tcnn_trainer_set_global_batch_size(m_h->tm, n)); }
	void set_gradient_exchange(void (*exchange)(void*, void*, size_t, tcnn_stream_t), void* user) { check(tcnn_trainer_set_gradient_exchange(m_h->tm, exchange, user)); }
	// exchange overlapped with the backward pass: `ready` is called inside training_step for every gradient range as soon as its kernels
	// are enqueued (network weights, then the encoding's level groups); or hand the library this rank's ncclComm_t and let it all-reduce
	void set_backward_level_groups(uint32_t n_groups) { check(tcnn_trainer_set_backward_level_groups(m_h->tm, n_groups)); }
	void set_gradient_ready_callback(void (*ready)(void*, size_t, size_t, tcnn_stream_t), void* user) { check(tcnn_trainer_set_gradient_ready_callback(m_h->tm, ready, user)); }
	void enable_rccl(void* nccl_comm, int n_ranks) { check(tcnn_trainer_enable_rccl(m_h->tm, nccl_comm, n_ranks)); }
	void enable_rccl_sharded(void* nccl_comm, int n_ranks, int rank) { check(tcnn_trainer_enable_rccl_sharded(m_h->tm, nccl_comm, n_ranks, rank)); }
	// exchange over peer-mapped memory (tcnn_hip.h: tcnn_trainer_direct_*): publish direct_export() to every rank, direct_open() with all of them
	std::vector<uint8_t> direct_export() const {
		size_t n = 0;
		check(tcnn_trainer_direct_export(m_h->tm, nullptr, 0, &n));
		std::vector<uint8_t> record(n);
		check(tcnn_trainer_direct_export(m_h->tm, record.data(), record.size(), &n));
		return record;
	}
	void direct_open(int rank, int n_ranks, const std::vector<uint8_t>& all_records) { check(tcnn_trainer_direct_open(m_h->tm, rank, n_ranks, all_records.data(), all_records.size() / (size_t)n_ranks)); }
	void direct_close() { check(tcnn_trainer_direct_close(m_h->tm)); }
	// link check of the opened exchange (collective, between steps; overwrites the gradient buffer only): wrong elements of this rank's buffer
	uint64_t direct_selftest(hipStream_t stream, uint32_t rounds = 3, uint32_t seed = 0, int* status = nullptr) {
		uint64_t bad = 0;
		int st = 0;
		check(tcnn_trainer_direct_selftest(m_h->tm, stream, rounds, seed, &bad, &st));
		if (status) *status = st;
		return bad;
	}
	tcnn_trainable_model_t* c_handle() const { return m_h->tm; }

private:
	std::unique_ptr<ForwardContext> new_context(uint32_t batch_size) const {
		auto out = std::make_unique<ForwardContext>();
		out->padded_output_width = tcnn_trainer_padded_output_width(m_h->tm);
		out->batch_size = batch_size;
		return out;
	}
	std::shared_ptr<NetworkWithInputEncoding<PARAMS_T>> m_model;
	std::shared_ptr<Optimizer<PARAMS_T>> m_optimizer;
	std::shared_ptr<Loss<COMPUTE_T>> m_loss;
	std::shared_ptr<detail::ModelHandle> m_h;
};

}  // namespace tcnn
