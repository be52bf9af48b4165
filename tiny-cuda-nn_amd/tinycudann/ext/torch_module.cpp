// Compiled PyTorch binding of the tinycudann package: the reference's pybind11 module (bindings/torch/tinycudann/bindings.cpp:75-343 -- `Module`
// with fwd / bwd / bwd_bwd_input / initial_params and the accessors, the three factories, the free functions) as plain C++ over the C ABI of
// include/tcnn_hip.h, plus the autograd function pair that modules.py implements in Python (reference modules.py:132-201), so that a training
// step costs ONE Python -> C++ transition per direction instead of a chain of ctypes calls and two Python autograd.Function frames: on
// small tables (data/config_hash.json as shipped) the ctypes binding's loop was bound by host time, 2.4 - 2.8 x the native step
// (profiles/r05_exp_notes.txt section 5).  No kernels here and nothing hipified: device work happens behind the C ABI, on torch's current stream
// (getCurrentHIPStreamMasqueradingAsCUDA, bindings.cpp:96) and under a device guard (bindings.cpp:95).
//
// The 16-bit type is a property of the library that is loaded (libtcnn_hip.so / libtcnn_hip_bf16.so, TCNN_PRECISION): this extension links
// neither; bind_library(path) resolves the entry points from the library tinycudann/_C.py has loaded already.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>  // (ROCm builds of PyTorch call their devices "cuda": the guard and the stream
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>     //  accessor that accept that device type)
#include <dlfcn.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>

namespace {

struct tcnn_module;
struct tcnn_context;
struct Abi {  // include/tcnn_hip.h, the entries this binding calls
	const char* (*last_error)() = nullptr;
	int (*create_nwie)(uint32_t, uint32_t, const char*, const char*, tcnn_module**) = nullptr;
	int (*create_network)(uint32_t, uint32_t, const char*, tcnn_module**) = nullptr;
	int (*create_encoding)(uint32_t, const char*, int, tcnn_module**) = nullptr;
	void (*module_destroy)(tcnn_module*) = nullptr;
	int (*inference)(tcnn_module*, void*, uint32_t, const float*, void*, void*) = nullptr;
	int (*forward)(tcnn_module*, void*, uint32_t, const float*, void*, void*, int, tcnn_context**) = nullptr;
	int (*backward)(tcnn_module*, void*, const tcnn_context*, uint32_t, float*, const void*, void*, const float*, const void*, const void*) = nullptr;
	int (*backward_backward_input)(tcnn_module*, void*, const tcnn_context*, uint32_t, const float*, const float*, const void*, void*, void*, float*,
	                               const void*) = nullptr;
	void (*context_destroy)(tcnn_context*) = nullptr;
	uint32_t (*n_input_dims)(const tcnn_module*) = nullptr;
	uint32_t (*n_output_dims)(const tcnn_module*) = nullptr;
	size_t (*n_params)(const tcnn_module*) = nullptr;
	int (*param_precision)(const tcnn_module*) = nullptr;
	int (*output_precision)(const tcnn_module*) = nullptr;
	int (*initialize_params)(tcnn_module*, size_t, float*, float) = nullptr;
	const char* (*hyperparams_json)(const tcnn_module*) = nullptr;
	const char* (*name)(const tcnn_module*) = nullptr;
	int (*jit_fusion)(const tcnn_module*) = nullptr;
	int (*set_jit_fusion)(tcnn_module*, int) = nullptr;
	uint32_t (*batch_size_granularity)() = nullptr;
	float (*default_loss_scale)(int) = nullptr;
	bool bound = false;
	std::string path;
} capi;

template <typename F>
void resolve(void* lib, F& fn, const char* symbol) {
	fn = (F)dlsym(lib, symbol);
	if (!fn) throw std::runtime_error(std::string("tinycudann extension: ") + symbol + " not found in the native library");
}

void bind_library(const std::string& path) {
	// (RTLD_NOLOAD first: the library is mapped already -- _C.py loaded it -- and must not be mapped a second time under another name)
	void* lib = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
	if (!lib) lib = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
	if (!lib) throw std::runtime_error("tinycudann extension: cannot open " + path + ": " + dlerror());
	resolve(lib, capi.last_error, "tcnn_last_error");
	resolve(lib, capi.create_nwie, "tcnn_create_network_with_input_encoding");
	resolve(lib, capi.create_network, "tcnn_create_network");
	resolve(lib, capi.create_encoding, "tcnn_create_encoding");
	resolve(lib, capi.module_destroy, "tcnn_module_destroy");
	resolve(lib, capi.inference, "tcnn_module_inference");
	resolve(lib, capi.forward, "tcnn_module_forward");
	resolve(lib, capi.backward, "tcnn_module_backward");
	resolve(lib, capi.backward_backward_input, "tcnn_module_backward_backward_input");
	resolve(lib, capi.context_destroy, "tcnn_context_destroy");
	resolve(lib, capi.n_input_dims, "tcnn_module_n_input_dims");
	resolve(lib, capi.n_output_dims, "tcnn_module_n_output_dims");
	resolve(lib, capi.n_params, "tcnn_module_n_params");
	resolve(lib, capi.param_precision, "tcnn_module_param_precision");
	resolve(lib, capi.output_precision, "tcnn_module_output_precision");
	resolve(lib, capi.initialize_params, "tcnn_module_initialize_params");
	resolve(lib, capi.hyperparams_json, "tcnn_module_hyperparams_json");
	resolve(lib, capi.name, "tcnn_module_name");
	resolve(lib, capi.jit_fusion, "tcnn_module_jit_fusion");
	resolve(lib, capi.set_jit_fusion, "tcnn_module_set_jit_fusion");
	resolve(lib, capi.batch_size_granularity, "tcnn_batch_size_granularity");
	resolve(lib, capi.default_loss_scale, "tcnn_default_loss_scale");
	capi.bound = true;
	capi.path = path;
}

void check(int code) {  // the reference throws std::runtime_error (common_host.h:71-110) -> Python RuntimeError
	if (code != 0) throw std::runtime_error(capi.last_error ? capi.last_error() : "tinycudann: native call failed");
}

// include/tcnn_hip.h tcnn_precision_t: 0 = fp32, 1 = fp16, 2 = bf16
c10::ScalarType torch_type(int precision) {
	switch (precision) {
		case 0: return torch::kFloat32;
		case 1: return torch::kHalf;
		case 2: return torch::kBFloat16;
		default: throw std::runtime_error("Unknown precision tcnn->torch");
	}
}

void* current_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

#define CHECK_INPUT(x)                                                                   \
	do {                                                                                 \
		if (!(x).device().is_cuda()) throw std::runtime_error(#x " must be a CUDA tensor"); \
		if (!(x).is_contiguous()) throw std::runtime_error(#x " must be contiguous");       \
	} while (0)

// tcnn::cpp::Context (cpp_api.h:87-89): owns the saved activations of one forward call
struct Context {
	tcnn_context* h = nullptr;
	Context() = default;
	explicit Context(tcnn_context* h_) : h(h_) {}
	Context(const Context&) = delete;
	Context& operator=(const Context&) = delete;
	Context(Context&& o) noexcept : h(o.h) { o.h = nullptr; }
	~Context() {
		if (h && capi.context_destroy) capi.context_destroy(h);
	}
	bool valid() const { return h != nullptr; }
};

class Module {
public:
	explicit Module(tcnn_module* m) : m_(m) {
		n_input_dims_ = capi.n_input_dims(m_);
		n_output_dims_ = capi.n_output_dims(m_);
		n_params_ = capi.n_params(m_);
		param_precision_ = capi.param_precision(m_);
		output_precision_ = capi.output_precision(m_);
	}
	Module(const Module&) = delete;
	Module& operator=(const Module&) = delete;
	~Module() {
		if (m_ && capi.module_destroy) capi.module_destroy(m_);
	}

	// bindings.cpp:79-110
	std::pair<std::shared_ptr<Context>, torch::Tensor> fwd(torch::Tensor input, torch::Tensor params) {
		CHECK_INPUT(input);
		CHECK_INPUT(params);
		if (input.scalar_type() != torch::kFloat32) throw std::runtime_error("input must be float32");
		if (params.scalar_type() != torch_type(param_precision_)) throw std::runtime_error("params have the wrong precision");
		if (input.dim() != 2 || input.size(1) != n_input_dims_ || params.size(0) != (int64_t)n_params_) throw std::runtime_error("input / params have the wrong size");
		if (input.device() != params.device()) throw std::runtime_error("input and params must be on the same device");
		const c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
		const uint32_t batch_size = (uint32_t)input.size(0);
		torch::Tensor output = torch::empty({(int64_t)batch_size, (int64_t)n_output_dims_}, torch::TensorOptions().dtype(torch_type(output_precision_)).device(input.device()));
		if (!input.requires_grad() && !params.requires_grad()) {
			check(capi.inference(m_, current_stream(), batch_size, input.data_ptr<float>(), output.data_ptr(), params.data_ptr()));
			return {std::make_shared<Context>(), output};
		}
		tcnn_context* h = nullptr;
		check(capi.forward(m_, current_stream(), batch_size, input.data_ptr<float>(), output.data_ptr(), params.data_ptr(), input.requires_grad() ? 1 : 0, &h));
		return {std::make_shared<Context>(h), output};
	}

	// bindings.cpp:112-171
	std::pair<c10::optional<torch::Tensor>, c10::optional<torch::Tensor>> bwd(const std::shared_ptr<Context>& ctx, torch::Tensor input, torch::Tensor params, torch::Tensor output,
	                                                                          torch::Tensor dL_doutput) {
		if (!ctx || !ctx->valid()) throw std::runtime_error("Module::bwd: called with invalid context. fwd likely (mistakenly) ran in inference mode.");
		CHECK_INPUT(input);
		CHECK_INPUT(params);
		CHECK_INPUT(output);
		CHECK_INPUT(dL_doutput);
		if (input.scalar_type() != torch::kFloat32 || params.scalar_type() != torch_type(param_precision_) || output.scalar_type() != torch_type(output_precision_) ||
		    dL_doutput.scalar_type() != torch_type(output_precision_)) {
			throw std::runtime_error("bwd: wrong tensor precision");
		}
		if (input.size(1) != n_input_dims_ || output.size(1) != n_output_dims_ || params.size(0) != (int64_t)n_params_ || output.size(0) != input.size(0) ||
		    dL_doutput.size(0) != input.size(0)) {
			throw std::runtime_error("bwd: wrong tensor size");
		}
		const c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
		const uint32_t batch_size = (uint32_t)input.size(0);
		c10::optional<torch::Tensor> dL_dinput, dL_dparams;
		if (input.requires_grad()) dL_dinput = torch::empty({(int64_t)batch_size, input.size(1)}, torch::TensorOptions().dtype(torch::kFloat32).device(input.device()));
		if (params.requires_grad()) dL_dparams = torch::empty({(int64_t)n_params_}, torch::TensorOptions().dtype(torch_type(param_precision_)).device(input.device()));
		if (input.requires_grad() || params.requires_grad()) {
			check(capi.backward(m_, current_stream(), ctx->h, batch_size, dL_dinput ? dL_dinput->data_ptr<float>() : nullptr, dL_doutput.data_ptr(),
			                   dL_dparams ? dL_dparams->data_ptr() : nullptr, input.data_ptr<float>(), output.data_ptr(), params.data_ptr()));
		}
		return {dL_dinput, dL_dparams};
	}

	// bindings.cpp:173-241 -> (dL_ddLdoutput, dL_dparams, dL_dinput)
	std::tuple<c10::optional<torch::Tensor>, c10::optional<torch::Tensor>, c10::optional<torch::Tensor>> bwd_bwd_input(const std::shared_ptr<Context>& ctx, torch::Tensor input,
	                                                                                                                     torch::Tensor params, torch::Tensor dL_ddLdinput,
	                                                                                                                     torch::Tensor dL_doutput) {
		if (!ctx || !ctx->valid()) throw std::runtime_error("Module::bwd_bwd_input: called with invalid context. fwd likely (mistakenly) ran in inference mode.");
		CHECK_INPUT(input);
		CHECK_INPUT(params);
		CHECK_INPUT(dL_ddLdinput);
		CHECK_INPUT(dL_doutput);
		if (input.scalar_type() != torch::kFloat32 || dL_ddLdinput.scalar_type() != torch::kFloat32 || dL_doutput.scalar_type() != torch_type(output_precision_)) {
			throw std::runtime_error("bwd_bwd_input: wrong tensor dtype");
		}
		if (input.size(1) != n_input_dims_ || dL_doutput.size(1) != n_output_dims_ || dL_ddLdinput.sizes() != input.sizes() || params.size(0) != (int64_t)n_params_ ||
		    dL_doutput.size(0) != input.size(0)) {
			throw std::runtime_error("bwd_bwd_input: wrong tensor size");
		}
		const c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
		const uint32_t batch_size = (uint32_t)input.size(0);
		c10::optional<torch::Tensor> dL_ddLdoutput, dL_dparams, dL_dinput;
		if (dL_doutput.requires_grad()) dL_ddLdoutput = torch::zeros({(int64_t)batch_size, (int64_t)n_output_dims_}, torch::TensorOptions().dtype(torch_type(output_precision_)).device(input.device()));
		if (params.requires_grad()) dL_dparams = torch::zeros({(int64_t)n_params_}, torch::TensorOptions().dtype(torch_type(param_precision_)).device(input.device()));
		if (input.requires_grad()) dL_dinput = torch::zeros({(int64_t)batch_size, input.size(1)}, torch::TensorOptions().dtype(torch::kFloat32).device(input.device()));
		if (dL_doutput.requires_grad() || params.requires_grad() || input.requires_grad()) {
			check(capi.backward_backward_input(m_, current_stream(), ctx->h, batch_size, dL_ddLdinput.data_ptr<float>(), input.data_ptr<float>(), dL_doutput.data_ptr(),
			                                  dL_dparams ? dL_dparams->data_ptr() : nullptr, dL_ddLdoutput ? dL_ddLdoutput->data_ptr() : nullptr,
			                                  dL_dinput ? dL_dinput->data_ptr<float>() : nullptr, params.data_ptr()));
		}
		return {dL_ddLdoutput, dL_dparams, dL_dinput};
	}

	torch::Tensor initial_params(size_t seed) {  // bindings.cpp:284-289
		torch::Tensor out = torch::zeros({(int64_t)n_params_}, torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA));
		check(capi.initialize_params(m_, seed, out.data_ptr<float>(), 1.0f));
		return out;
	}

	uint32_t n_input_dims() const { return n_input_dims_; }
	size_t n_params() const { return n_params_; }
	int param_precision() const { return param_precision_; }
	uint32_t n_output_dims() const { return n_output_dims_; }
	int output_precision() const { return output_precision_; }
	std::string hyperparams_json() const { return capi.hyperparams_json(m_); }
	std::string name() const { return capi.name(m_); }
	bool jit_fusion() const { return capi.jit_fusion(m_) != 0; }
	void set_jit_fusion(bool val) { check(capi.set_jit_fusion(m_, val ? 1 : 0)); }
	uintptr_t handle() const { return (uintptr_t)m_; }  // for the ctypes helpers of _C.py (grid_indices, level tables)

private:
	tcnn_module* m_;
	uint32_t n_input_dims_, n_output_dims_;
	size_t n_params_;
	int param_precision_, output_precision_;
};

// ---- the autograd function pair (reference modules.py:132-201): forward -> Module::fwd; backward scales dL/doutput by the loss scale, calls
// Module::bwd and divides the gradients by it; under create_graph the backward is itself an autograd node whose backward is bwd_bwd_input.
struct FwdState : torch::CustomClassHolder {  // what the backward nodes keep of a forward call (not a tensor: lives in AutogradContext::saved_data)
	std::shared_ptr<Module> module;
	std::shared_ptr<Context> native_ctx;
	double loss_scale = 1.0;
};

torch::Tensor scaled_doutput(const torch::Tensor& dy, double loss_scale, c10::ScalarType type) {
	return (dy * loss_scale).to(type).contiguous();  // modules.py:167
}

struct NativeBackwardFunction : public torch::autograd::Function<NativeBackwardFunction> {
	// inputs: dy, x, params, y (+ the forward call's state) -> (dx, dparams); a 0-dim tensor stands for "no gradient" (outputs must be tensors)
	static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, c10::intrusive_ptr<FwdState> st, torch::Tensor dy, torch::Tensor x, torch::Tensor params,
	                                              torch::Tensor y) {
		ctx->saved_data["state"] = st;
		ctx->save_for_backward({x, params, dy});
		at::AutoGradMode no_grad(false);
		auto r = st->module->bwd(st->native_ctx, x, params, y, scaled_doutput(dy, st->loss_scale, y.scalar_type()));
		torch::Tensor dx = r.first ? *r.first / st->loss_scale : torch::empty({}, x.options());
		torch::Tensor dparams = r.second ? *r.second / st->loss_scale : torch::empty({}, params.options());
		return {dx, dparams};
	}
	static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
		auto st = ctx->saved_data["state"].toCustomClass<FwdState>();
		const auto saved = ctx->get_saved_variables();
		const torch::Tensor &x = saved[0], &params = saved[1], &dy = saved[2];
		const torch::Tensor& ddx = grads[0];
		if (!ddx.defined() || ddx.dim() == 0) return {torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
		torch::Tensor scaled;
		{
			at::AutoGradMode grad(true);  // keeps dy's requires_grad flag
			scaled = scaled_doutput(dy, st->loss_scale, torch_type(st->module->output_precision()));
		}
		at::AutoGradMode no_grad(false);
		auto r = st->module->bwd_bwd_input(st->native_ctx, x, params, ddx.to(torch::kFloat32).contiguous(), scaled);
		// d_dy depends on ddx only; the other two carry one factor of the loss scale through the scaled dL/doutput
		torch::Tensor d_dy = std::get<0>(r) ? *std::get<0>(r) : torch::Tensor();
		torch::Tensor d_params = std::get<1>(r) ? *std::get<1>(r) / st->loss_scale : torch::Tensor();
		torch::Tensor d_x = std::get<2>(r) ? *std::get<2>(r) / st->loss_scale : torch::Tensor();
		return {torch::Tensor(), d_dy, d_x, d_params, torch::Tensor()};
	}
};

struct NativeFunction : public torch::autograd::Function<NativeFunction> {
	static torch::Tensor forward(torch::autograd::AutogradContext* ctx, c10::intrusive_ptr<FwdState> st, torch::Tensor x, torch::Tensor params) {
		ctx->set_materialize_grads(false);
		auto r = st->module->fwd(x, params);
		st->native_ctx = r.first;
		ctx->saved_data["state"] = st;
		ctx->save_for_backward({x, params, r.second});
		return r.second;
	}
	static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
		torch::Tensor dy = grads[0];
		if (!dy.defined()) return {torch::Tensor(), torch::Tensor(), torch::Tensor()};
		auto st = ctx->saved_data["state"].toCustomClass<FwdState>();
		const auto saved = ctx->get_saved_variables();
		const torch::Tensor &x = saved[0], &params = saved[1], &y = saved[2];
		if (!dy.device().is_cuda()) {
			TORCH_WARN("doutput must be a GPU tensor, but isn't. This indicates suboptimal performance.");
			dy = dy.to(y.device());
		}
		if (!at::GradMode::is_enabled()) {  // the ordinary backward pass (no create_graph): the native call right here
			auto r = st->module->bwd(st->native_ctx, x, params, y, scaled_doutput(dy, st->loss_scale, y.scalar_type()));
			return {torch::Tensor(), r.first ? *r.first / st->loss_scale : torch::Tensor(), r.second ? *r.second / st->loss_scale : torch::Tensor()};
		}
		// a node of its own, so that the input gradient can be differentiated again (eikonal / SDF losses)
		auto r = NativeBackwardFunction::apply(st, dy, x, params, y);
		return {torch::Tensor(), r[0].dim() == 0 ? torch::Tensor() : r[0], r[1].dim() == 0 ? torch::Tensor() : r[1]};
	}
};

// y = module(x, params) with gradients, params already in the module's precision (modules.py:230)
torch::Tensor module_apply(const std::shared_ptr<Module>& module, torch::Tensor x, torch::Tensor params, double loss_scale) {
	auto st = c10::make_intrusive<FwdState>();
	st->module = module;
	st->loss_scale = loss_scale;
	return NativeFunction::apply(st, x, params);
}

std::shared_ptr<Module> create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims, const std::string& encoding_json, const std::string& network_json) {
	tcnn_module* m = nullptr;
	check(capi.create_nwie(n_input_dims, n_output_dims, encoding_json.c_str(), network_json.c_str(), &m));
	return std::make_shared<Module>(m);
}
std::shared_ptr<Module> create_network(uint32_t n_input_dims, uint32_t n_output_dims, const std::string& network_json) {
	tcnn_module* m = nullptr;
	check(capi.create_network(n_input_dims, n_output_dims, network_json.c_str(), &m));
	return std::make_shared<Module>(m);
}
std::shared_ptr<Module> create_encoding(uint32_t n_input_dims, const std::string& encoding_json, int precision) {
	tcnn_module* m = nullptr;
	check(capi.create_encoding(n_input_dims, encoding_json.c_str(), precision, &m));
	return std::make_shared<Module>(m);
}

}  // namespace

TORCH_LIBRARY(tcnn_amd, m) { m.class_<FwdState>("FwdState"); }  // (registered so that it can ride in an IValue; never constructed from Python)

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
	m.def("bind_library", &bind_library, "resolve the C ABI from the native library at this path (tinycudann/_C.py has loaded it)");
	m.def("library_path", [] { return capi.path; });
	m.def("batch_size_granularity", [] { return capi.batch_size_granularity(); });
	m.def("default_loss_scale", [](int precision) { return capi.default_loss_scale(precision); });
	py::class_<Context, std::shared_ptr<Context>>(m, "Context").def_property_readonly("valid", &Context::valid);
	py::class_<Module, std::shared_ptr<Module>>(m, "Module")
		.def("fwd", &Module::fwd)
		.def("bwd", &Module::bwd)
		.def("bwd_bwd_input", &Module::bwd_bwd_input)
		.def("initial_params", &Module::initial_params)
		.def("n_input_dims", &Module::n_input_dims)
		.def("n_params", &Module::n_params)
		.def("param_precision", &Module::param_precision)
		.def("n_output_dims", &Module::n_output_dims)
		.def("output_precision", &Module::output_precision)
		.def("hyperparams_json", &Module::hyperparams_json)
		.def("name", &Module::name)
		.def("handle", &Module::handle)
		.def_property("jit_fusion", &Module::jit_fusion, &Module::set_jit_fusion);
	m.def("create_network_with_input_encoding", &create_network_with_input_encoding);
	m.def("create_network", &create_network);
	m.def("create_encoding", &create_encoding);
	m.def("apply", &module_apply, "y = module(x, params) as an autograd node (first and second order), dL/doutput scaled by loss_scale on the way down");
}
