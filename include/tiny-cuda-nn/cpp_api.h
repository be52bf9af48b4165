/*
 * tiny-cuda-nn/cpp_api.h -- `tcnn::cpp::Module` and its factories (reference include/tiny-cuda-nn/cpp_api.h:40-123, src/cpp_api.cu:40-174)
 * over the C ABI: the type-erased interface the reference's bindings are written against (bindings/torch/tinycudann/bindings.cpp:49
 * holds a `std::unique_ptr<tcnn::cpp::Module>` and calls nothing else).  A binding that includes this header instead of the
 * reference's, with `hipStream_t` where the reference says `cudaStream_t` (what PyTorch-ROCm's build does to the binding's own
 * sources anyway), compiles against libtcnn_hip.so unchanged: same class, same virtual functions, same factories, same free functions.
 *
 * Header-only; the one concrete Module is `HipModule`, a handle to a tcnn_module_t.  `tcnn::cpp::Context` owns the forward pass's
 * saved state the way the reference's does (cpp_api.h:87-89): a unique_ptr to a `tcnn::Context`, here the wrapper of a tcnn_context_t.
 */
#pragma once
#include <tiny-cuda-nn/common.h>

#include <functional>

namespace tcnn {

// object.h:40-47 of the reference: the base of whatever a forward pass keeps for its backward pass
struct Context {
	Context() = default;
	virtual ~Context() {}
	Context(const Context&) = delete;
	Context& operator=(const Context&) = delete;
};

namespace cpp {

enum class LogSeverity { Info, Debug, Warning, Error, Success };  // cpp_api.h:52-58 (== TCNN_LOG_*)
enum class Precision { Fp32, Fp16 };                              // cpp_api.h:71-74; the bfloat16 build reports Fp16 here: "the 16-bit type"

using json = ::tcnn::json;

inline uint32_t batch_size_granularity() { return tcnn_batch_size_granularity(); }
inline int hip_device() { return tcnn_hip_device(); }  // cuda_device(), cpp_api.h:64
inline void set_hip_device(int device) { check(tcnn_set_hip_device(device)); }
inline void free_temporary_memory() { tcnn_free_temporary_memory(); }
inline bool has_networks() { return tcnn_has_networks() != 0; }
inline float default_loss_scale(Precision p) { return tcnn_default_loss_scale(p == Precision::Fp32 ? TCNN_PRECISION_FP32 : TCNN_PRECISION_FP16); }
inline Precision preferred_precision() { return tcnn_preferred_precision() == TCNN_PRECISION_FP32 ? Precision::Fp32 : Precision::Fp16; }
inline bool supports_jit_fusion(int device = -1) { return tcnn_supports_jit_fusion(device) != 0; }  // always false: no run-time compilation
inline void rtc_set_cache_dir(const std::string&) {}                                                 // (accepted, nothing to configure)
inline void rtc_set_include_dir(const std::string&) {}
inline void set_log_callback(const std::function<void(LogSeverity, const std::string&)>& callback) {  // cpp_api.h:85
	static std::function<void(LogSeverity, const std::string&)> current;
	current = callback;
	tcnn_set_log_callback(callback ? +[](int severity, const char* message) { current(static_cast<LogSeverity>(severity), message); } : nullptr);
}

struct Context {  // cpp_api.h:87-89
	std::unique_ptr<tcnn::Context> ctx;
};

class Module {  // cpp_api.h:91-119, member for member
public:
	Module(Precision param_precision, Precision output_precision) : m_param_precision{param_precision}, m_output_precision{output_precision} {}
	virtual ~Module() {}

	virtual void inference(hipStream_t stream, uint32_t n_elements, const float* input, void* output, void* params) = 0;
	virtual Context forward(hipStream_t stream, uint32_t n_elements, const float* input, void* output, void* params, bool prepare_input_gradients) = 0;
	virtual void backward(hipStream_t stream, const Context& ctx, uint32_t n_elements, float* dL_dinput, const void* dL_doutput, void* dL_dparams, const float* input,
	                      const void* output, const void* params) = 0;
	virtual void backward_backward_input(hipStream_t stream, const Context& ctx, uint32_t n_elements, const float* dL_ddLdinput, const float* input, const void* dL_doutput,
	                                     void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput, const void* params) = 0;

	virtual uint32_t n_input_dims() const = 0;
	virtual uint32_t n_output_dims() const = 0;
	Precision output_precision() const { return m_output_precision; }

	virtual size_t n_params() const = 0;
	Precision param_precision() const { return m_param_precision; }

	virtual void initialize_params(size_t seed, float* params_full_precision, float scale = 1.0f) = 0;

	virtual json hyperparams() const = 0;
	virtual std::string name() const = 0;

	virtual bool jit_fusion() const = 0;
	virtual void set_jit_fusion(bool val) = 0;

private:
	Precision m_param_precision;
	Precision m_output_precision;
};

// what stands where src/cpp_api.cu:72-153's DifferentiableObject wrapper stands
class HipModule : public Module {
	struct HipContext : public tcnn::Context {
		explicit HipContext(tcnn_context_t* h_) : h{h_} {}
		~HipContext() override { tcnn_context_destroy(h); }
		tcnn_context_t* h;
	};
	static Precision precision_of(int p) { return p == TCNN_PRECISION_FP32 ? Precision::Fp32 : Precision::Fp16; }
	static const tcnn_context_t* handle_of(const Context& ctx) {
		auto* c = dynamic_cast<const HipContext*>(ctx.ctx.get());
		if (!c) throw std::runtime_error("Module::backward called with invalid context. Did you call forward() in inference mode?");  // cpp_api.cu:100-102
		return c->h;
	}

public:
	explicit HipModule(tcnn_module_t* m) : Module{precision_of(tcnn_module_param_precision(m)), precision_of(tcnn_module_output_precision(m))}, m_{m} {}
	~HipModule() override { tcnn_module_destroy(m_); }
	HipModule(const HipModule&) = delete;
	HipModule& operator=(const HipModule&) = delete;

	void inference(hipStream_t stream, uint32_t n_elements, const float* input, void* output, void* params) override {
		check(tcnn_module_inference(m_, stream, n_elements, input, output, params));
	}
	Context forward(hipStream_t stream, uint32_t n_elements, const float* input, void* output, void* params, bool prepare_input_gradients) override {
		tcnn_context_t* c = nullptr;
		check(tcnn_module_forward(m_, stream, n_elements, input, output, params, prepare_input_gradients, &c));
		Context result;
		result.ctx = std::make_unique<HipContext>(c);
		return result;
	}
	void backward(hipStream_t stream, const Context& ctx, uint32_t n_elements, float* dL_dinput, const void* dL_doutput, void* dL_dparams, const float* input,
	              const void* output, const void* params) override {
		check(tcnn_module_backward(m_, stream, handle_of(ctx), n_elements, dL_dinput, dL_doutput, dL_dparams, input, output, params));
	}
	void backward_backward_input(hipStream_t stream, const Context& ctx, uint32_t n_elements, const float* dL_ddLdinput, const float* input, const void* dL_doutput,
	                             void* dL_dparams, void* dL_ddLdoutput, float* dL_dinput, const void* params) override {
		check(tcnn_module_backward_backward_input(m_, stream, handle_of(ctx), n_elements, dL_ddLdinput, input, dL_doutput, dL_dparams, dL_ddLdoutput, dL_dinput, params));
	}
	uint32_t n_input_dims() const override { return tcnn_module_n_input_dims(m_); }
	uint32_t n_output_dims() const override { return tcnn_module_n_output_dims(m_); }  // the PADDED width, as cpp_api.cu:137
	size_t n_params() const override { return tcnn_module_n_params(m_); }
	void initialize_params(size_t seed, float* params_full_precision, float scale = 1.0f) override {
		check(tcnn_module_initialize_params(m_, seed, params_full_precision, scale));
	}
	json hyperparams() const override { return json::parse(std::string(tcnn_module_hyperparams_json(m_))); }
	std::string name() const override { return tcnn_module_name(m_); }
	bool jit_fusion() const override { return tcnn_module_jit_fusion(m_) != 0; }
	void set_jit_fusion(bool val) override { check(tcnn_module_set_jit_fusion(m_, val)); }

	tcnn_module_t* c_handle() const { return m_; }

private:
	tcnn_module_t* m_;
};

// cpp_api.h:121-123 (the caller owns the result, as in the reference)
inline Module* create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const json& encoding, const json& network) {
	tcnn_module_t* m = nullptr;
	check(tcnn_create_network_with_input_encoding(n_input_dims, n_output_dims, json_text(encoding).c_str(), json_text(network).c_str(), &m));
	return new HipModule(m);
}
inline Module* create_network(uint32_t n_input_dims, uint32_t n_output_dims, const json& network) {
	tcnn_module_t* m = nullptr;
	check(tcnn_create_network(n_input_dims, n_output_dims, json_text(network).c_str(), &m));
	return new HipModule(m);
}
inline Module* create_encoding(uint32_t n_input_dims, const json& encoding, Precision requested_precision) {
	tcnn_module_t* m = nullptr;
	check(tcnn_create_encoding(n_input_dims, json_text(encoding).c_str(), requested_precision == Precision::Fp32 ? TCNN_PRECISION_FP32 : TCNN_PRECISION_FP16, &m));
	return new HipModule(m);
}

}  // namespace cpp
}  // namespace tcnn
