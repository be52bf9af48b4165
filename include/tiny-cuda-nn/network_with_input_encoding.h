/*
 * tiny-cuda-nn/network_with_input_encoding.h -- NetworkWithInputEncoding<T> as the hot path's callers see it
 * (reference network_with_input_encoding.h:40-130, object.h:166-271): construction from the encoding / network JSON,
 * inference(), widths, parameter views.  Parameters belong to the Trainer that is built around the network
 * (trainer.h:76, 489-503); inference before that throws, as calling a parameter-less network does in the reference.
 */
#pragma once
#include <tiny-cuda-nn/gpu_matrix.h>

namespace tcnn {

namespace detail {
struct ModelHandle {
	tcnn_trainable_model_t* tm = nullptr;
	~ModelHandle() {
		if (tm) tcnn_trainable_model_destroy(tm);
	}
};
}  // namespace detail

template <typename T>
class NetworkWithInputEncoding {
public:
	NetworkWithInputEncoding(uint32_t n_dims_to_encode, uint32_t n_output_dims, const json& encoding, const json& network)
	    : m_n_input_dims(n_dims_to_encode), m_n_output_dims(n_output_dims), m_encoding(encoding), m_network(network) {}

	// DifferentiableObject::inference (object.h:214-271): fp32 in, fp32 out (trimmed to n_output_dims)
	void inference(hipStream_t stream, const GPUMatrixDynamic<float>& input, GPUMatrixDynamic<float>& output, bool use_inference_params = true) {
		const tcnn_matrix_t in = input.c_matrix(), out = output.c_matrix();
		check(tcnn_network_inference_matrices(handle(), stream, &in, &out, use_inference_params));
	}
	void inference(const GPUMatrixDynamic<float>& input, GPUMatrixDynamic<float>& output, bool use_inference_params = true) {
		inference(nullptr, input, output, use_inference_params);
	}

	uint32_t input_width() const { return m_n_input_dims; }
	uint32_t output_width() const { return m_n_output_dims; }
	uint32_t padded_output_width() const { return tcnn_trainer_padded_output_width(handle()); }
	size_t n_params() const { return tcnn_trainer_n_params(handle()); }
	T* params() const { return static_cast<T*>(tcnn_trainer_params(handle())); }
	T* inference_params() const { return static_cast<T*>(tcnn_trainer_params_inference(handle())); }
	T* gradients() const { return static_cast<T*>(tcnn_trainer_param_gradients(handle())); }
	json hyperparams() const {  // network_with_input_encoding.h:200-206
		json j = json::object();
		j["otype"] = "NetworkWithInputEncoding";
		j["encoding"] = m_encoding;
		j["network"] = m_network;
		return j;
	}
	const json& encoding_config() const { return m_encoding; }
	const json& network_config() const { return m_network; }
	bool jit_fusion() const { return false; }  // no RTC path in this build (north_star); the statically fused kernels always run
	void set_jit_fusion(bool) {}

	void bind(std::shared_ptr<detail::ModelHandle> h) { m_h = std::move(h); }  // called by Trainer

private:
	tcnn_trainable_model_t* handle() const {
		if (!m_h || !m_h->tm) throw std::runtime_error("NetworkWithInputEncoding: no parameters yet -- construct a Trainer around the network first");
		return m_h->tm;
	}
	uint32_t m_n_input_dims, m_n_output_dims;
	json m_encoding, m_network;
	std::shared_ptr<detail::ModelHandle> m_h;
};

}  // namespace tcnn
