/*
 * tiny-cuda-nn/config.h -- header-only C++ facade over the C ABI (include/tcnn_hip.h, libtcnn_hip.so).
 *
 * Gives a C++ application written against the reference's hot-path surface
 *     auto model = tcnn::create_from_config(n_in, n_out, config);           (config.h:53-63)
 *     auto ctx   = model.trainer->training_step(stream, inputs, targets);   (trainer.h:254-357)
 *     float loss = model.trainer->loss(stream, *ctx);                       (trainer.h:372-374)
 *     model.network->inference(stream, inputs, outputs);                    (object.h:214-271)
 * the same spelling on MI355X.  Only what that path needs is mirrored: GPUMatrix<float> (column-major,
 * gpu_matrix.h:253-330), TrainableModel {network, trainer}, Trainer, the network's inference / parameter
 * accessors.  `config` is JSON TEXT (std::string); an application that already holds an nlohmann::json
 * passes config.dump().  Errors surface as std::runtime_error carrying tcnn_last_error(), as in the
 * reference (common_host.h:71-110).  No kernels or numerics live here -- everything forwards to the C ABI.
 */
#pragma once

#include <hip/hip_runtime_api.h>
#include <tcnn_hip.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace tcnn {

using network_precision_t = uint16_t;  // raw IEEE fp16 bits on the host side of the boundary

inline void check(int rc) {
	if (rc != TCNN_OK) throw std::runtime_error(tcnn_last_error());
}
inline void hip_check(hipError_t e, const char* what) {
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

inline uint32_t batch_size_granularity() { return tcnn_batch_size_granularity(); }
inline uint32_t next_multiple(uint32_t v, uint32_t d) { return (v + d - 1) / d * d; }

enum class GradientMode { Ignore = TCNN_GRADIENT_IGNORE, Overwrite = TCNN_GRADIENT_OVERWRITE, Accumulate = TCNN_GRADIENT_ACCUMULATE };

// rows x cols, column-major: element (r, c) at data()[c * rows + r]; i.e. one sample per column.
template <typename T>
class GPUMatrix {
public:
	GPUMatrix(uint32_t rows, uint32_t cols) : m_rows(rows), m_cols(cols), m_owned(true) {
		hip_check(hipMalloc(reinterpret_cast<void**>(&m_data), n_bytes()), "GPUMatrix: hipMalloc");
	}
	GPUMatrix(T* device_ptr, uint32_t rows, uint32_t cols) : m_data(device_ptr), m_rows(rows), m_cols(cols), m_owned(false) {}
	GPUMatrix(const GPUMatrix&) = delete;
	GPUMatrix& operator=(const GPUMatrix&) = delete;
	GPUMatrix(GPUMatrix&& o) noexcept : m_data(o.m_data), m_rows(o.m_rows), m_cols(o.m_cols), m_owned(o.m_owned) { o.m_data = nullptr; o.m_owned = false; }
	~GPUMatrix() { if (m_owned && m_data) (void)hipFree(m_data); }

	T* data() const { return m_data; }
	uint32_t rows() const { return m_rows; }
	uint32_t cols() const { return m_cols; }
	uint32_t m() const { return m_rows; }
	uint32_t n() const { return m_cols; }
	size_t n_elements() const { return size_t(m_rows) * m_cols; }
	size_t n_bytes() const { return n_elements() * sizeof(T); }

	void memset_async(hipStream_t stream, int value) { hip_check(hipMemsetAsync(m_data, value, n_bytes(), stream), "GPUMatrix: memset"); }
	void copy_from_host(const T* host) { hip_check(hipMemcpy(m_data, host, n_bytes(), hipMemcpyHostToDevice), "GPUMatrix: h2d"); }
	void copy_from_host(const std::vector<T>& host) {
		if (host.size() < n_elements()) throw std::runtime_error("GPUMatrix::copy_from_host: host buffer too small");
		copy_from_host(host.data());
	}
	std::vector<T> to_cpu_vector() const {
		std::vector<T> out(n_elements());
		hip_check(hipMemcpy(out.data(), m_data, n_bytes(), hipMemcpyDeviceToHost), "GPUMatrix: d2h");
		return out;
	}

private:
	T* m_data = nullptr;
	uint32_t m_rows, m_cols;
	bool m_owned;
};

namespace detail {
struct ModelHandle {
	tcnn_trainable_model_t* tm = nullptr;
	uint32_t n_input_dims = 0, n_output_dims = 0;
	~ModelHandle() { if (tm) tcnn_trainable_model_destroy(tm); }
};
inline void check_batch(const ModelHandle& h, const GPUMatrix<float>& input, const GPUMatrix<float>* target) {
	if (input.rows() != h.n_input_dims) throw std::runtime_error("input has " + std::to_string(input.rows()) + " rows, model expects " + std::to_string(h.n_input_dims));
	if (target && (target->rows() != h.n_output_dims || target->cols() != input.cols())) throw std::runtime_error("target shape does not match (n_output_dims x batch_size)");
}
}  // namespace detail

// NetworkWithInputEncoding as the hot path sees it: inference + parameter views.
class Network {
public:
	explicit Network(std::shared_ptr<detail::ModelHandle> h) : m_h(std::move(h)) {}

	void inference(hipStream_t stream, const GPUMatrix<float>& input, GPUMatrix<float>& output, bool use_inference_params = true) {
		detail::check_batch(*m_h, input, nullptr);
		if (output.rows() != m_h->n_output_dims || output.cols() != input.cols()) throw std::runtime_error("inference: output must be n_output_dims x batch_size");
		check(tcnn_network_inference(m_h->tm, stream, input.cols(), input.data(), output.data(), use_inference_params));
	}
	void inference(const GPUMatrix<float>& input, GPUMatrix<float>& output) { inference(nullptr, input, output); }

	uint32_t input_width() const { return m_h->n_input_dims; }
	uint32_t output_width() const { return m_h->n_output_dims; }
	uint32_t padded_output_width() const { return tcnn_trainer_padded_output_width(m_h->tm); }
	size_t n_params() const { return tcnn_trainer_n_params(m_h->tm); }
	network_precision_t* params() const { return static_cast<network_precision_t*>(tcnn_trainer_params(m_h->tm)); }
	network_precision_t* inference_params() const { return static_cast<network_precision_t*>(tcnn_trainer_params_inference(m_h->tm)); }
	network_precision_t* gradients() const { return static_cast<network_precision_t*>(tcnn_trainer_param_gradients(m_h->tm)); }

private:
	std::shared_ptr<detail::ModelHandle> m_h;
};

class Trainer {
public:
	// Trainer::ForwardContext (trainer.h:89-95); padded fp16 output / dL_doutput stay on the device.
	struct ForwardContext {
		tcnn_train_context_t* ctx = nullptr;
		ForwardContext() = default;
		ForwardContext(const ForwardContext&) = delete;
		ForwardContext& operator=(const ForwardContext&) = delete;
		~ForwardContext() { if (ctx) tcnn_train_context_destroy(ctx); }
		const network_precision_t* output() const { return static_cast<const network_precision_t*>(tcnn_train_context_output(ctx)); }
		const network_precision_t* dL_doutput() const { return static_cast<const network_precision_t*>(tcnn_train_context_dL_doutput(ctx)); }
	};

	explicit Trainer(std::shared_ptr<detail::ModelHandle> h) : m_h(std::move(h)) {}

	std::unique_ptr<ForwardContext> training_step(hipStream_t stream, const GPUMatrix<float>& input, const GPUMatrix<float>& target,
	                                               const GPUMatrix<float>* data_pdf = nullptr, bool run_optimizer = true,
	                                               GPUMatrix<float>* dL_dinput = nullptr, bool use_inference_params = false,
	                                               GradientMode param_gradients_mode = GradientMode::Overwrite) {
		detail::check_batch(*m_h, input, &target);
		auto out = std::make_unique<ForwardContext>();
		check(tcnn_trainer_training_step(m_h->tm, stream, input.cols(), input.data(), target.data(), data_pdf ? data_pdf->data() : nullptr,
		                                 run_optimizer, dL_dinput ? dL_dinput->data() : nullptr, use_inference_params,
		                                 static_cast<int>(param_gradients_mode), nullptr, &out->ctx));
		return out;
	}
	std::unique_ptr<ForwardContext> training_step(const GPUMatrix<float>& input, const GPUMatrix<float>& target) { return training_step(nullptr, input, target); }

	std::unique_ptr<ForwardContext> forward(hipStream_t stream, float loss_scale, const GPUMatrix<float>& input, const GPUMatrix<float>& target,
	                                         const GPUMatrix<float>* data_pdf = nullptr, bool use_inference_params = false, bool prepare_input_gradients = false) {
		detail::check_batch(*m_h, input, &target);
		auto out = std::make_unique<ForwardContext>();
		check(tcnn_trainer_forward(m_h->tm, stream, loss_scale, input.cols(), input.data(), target.data(), data_pdf ? data_pdf->data() : nullptr,
		                           use_inference_params, prepare_input_gradients, nullptr, &out->ctx));
		return out;
	}
	void backward(hipStream_t stream, const ForwardContext& ctx, const GPUMatrix<float>& input, GPUMatrix<float>* dL_dinput = nullptr,
	              bool use_inference_params = false, GradientMode param_gradients_mode = GradientMode::Overwrite) {
		check(tcnn_trainer_backward(m_h->tm, stream, ctx.ctx, input.cols(), input.data(), dL_dinput ? dL_dinput->data() : nullptr,
		                            use_inference_params, static_cast<int>(param_gradients_mode)));
	}
	void optimizer_step(hipStream_t stream, float loss_scale) { check(tcnn_trainer_optimizer_step(m_h->tm, stream, loss_scale)); }

	float loss(hipStream_t stream, const ForwardContext& ctx) {
		float v = 0.f;
		check(tcnn_trainer_loss(m_h->tm, stream, ctx.ctx, &v));
		return v;
	}

	size_t n_params() const { return tcnn_trainer_n_params(m_h->tm); }
	float* params_full_precision() const { return tcnn_trainer_params_full_precision(m_h->tm); }
	network_precision_t* params() const { return static_cast<network_precision_t*>(tcnn_trainer_params(m_h->tm)); }
	network_precision_t* params_inference() const { return static_cast<network_precision_t*>(tcnn_trainer_params_inference(m_h->tm)); }
	network_precision_t* param_gradients() const { return static_cast<network_precision_t*>(tcnn_trainer_param_gradients(m_h->tm)); }
	void set_params_full_precision(const float* params, size_t n, bool device_ptr = false) { check(tcnn_trainer_set_params_full_precision(m_h->tm, params, n, device_ptr)); }
	void update_hyperparams(const std::string& json_text) { check(tcnn_trainer_update_hyperparams(m_h->tm, json_text.c_str())); }
	std::string hyperparams() const { return tcnn_trainer_hyperparams_json(m_h->tm); }
	// MessagePack bytes of the reference's snapshot document (trainer.h:442-455)
	std::string serialize(bool serialize_optimizer = false) const {
		size_t n = 0;
		check(tcnn_trainer_serialize(m_h->tm, serialize_optimizer, nullptr, 0, &n));
		std::string blob(n, '\0');
		check(tcnn_trainer_serialize(m_h->tm, serialize_optimizer, blob.data(), blob.size(), &n));
		return blob;
	}
	void deserialize(const std::string& blob) { check(tcnn_trainer_deserialize(m_h->tm, blob.data(), blob.size())); }
	void set_global_batch_size(uint64_t n) { check(tcnn_trainer_set_global_batch_size(m_h->tm, n)); }

private:
	std::shared_ptr<detail::ModelHandle> m_h;
};

struct TrainableModel {
	std::shared_ptr<Network> network;
	std::shared_ptr<Trainer> trainer;
};

// config.h:53-63.  The trainer seed is the reference's default (trainer.h:53, 1337).
inline TrainableModel create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const std::string& config_json, uint32_t seed = 1337) {
	auto h = std::make_shared<detail::ModelHandle>();
	h->n_input_dims = n_input_dims;
	h->n_output_dims = n_output_dims;
	check(tcnn_create_from_config(n_input_dims, n_output_dims, config_json.c_str(), seed, &h->tm));
	return {std::make_shared<Network>(h), std::make_shared<Trainer>(h)};
}

}  // namespace tcnn
