/*
 * tiny-cuda-nn/optimizer.h -- Optimizer<T> + create_optimizer<T>(json) (reference optimizer.h:40-99,
 * src/optimizer.cu:50-86).  Adam, optionally wrapped in Ema / ExponentialDecay, runs inside the library; this object
 * carries the configuration to the Trainer built from it and, once bound, reads the live state back.  Used on its own
 * (allocate() + step(), optimizer.h:52-60) it owns an Adam state over the caller's weight buffers (tcnn_create_optimizer).
 */
#pragma once
#include <tiny-cuda-nn/common.h>

#include <utility>
#include <vector>

namespace tcnn {

template <typename T>
class Optimizer {
public:
	explicit Optimizer(const json& params) : m_params(params) {}
	json hyperparams() const { return m_params; }
	void update_hyperparams(const json& params) {  // optimizer.h:77
		m_params = params;
		if (m_tm) {
			json wrapper = json::object();
			wrapper["optimizer"] = params;
			check(tcnn_trainer_update_hyperparams(m_tm, json_text(wrapper).c_str()));
		}
		if (m_own) check(tcnn_optimizer_update_hyperparams(m_own, json_text(params).c_str()));
	}
	uint32_t step() const { return m_tm ? tcnn_trainer_optimizer_step_count(m_tm) : (m_own ? tcnn_optimizer_step_count(m_own) : 0u); }  // optimizer.h:66

	// on its own (optimizer.h:52-60): layer_sizes = (rows, cols) of the weight matrices that lead the parameter buffer
	void allocate(uint32_t n_weights, const std::vector<std::pair<uint32_t, uint32_t>>& layer_sizes = {}) {
		size_t n_matrix = 0;
		for (const auto& l : layer_sizes) n_matrix += (size_t)l.first * l.second;
		if (!m_own) check(tcnn_create_optimizer(json_text(m_params).c_str(), &m_own));
		check(tcnn_optimizer_allocate(m_own, n_weights, n_matrix));
	}
	void step(hipStream_t stream, float loss_scale, float* weights_full_precision, T* weights, const T* gradients) {
		if (!m_own) throw std::runtime_error("Optimizer::step: allocate() first (an optimizer bound to a Trainer is stepped by the Trainer)");
		check(tcnn_optimizer_step(m_own, (tcnn_stream_t)stream, loss_scale, weights_full_precision, weights, gradients));
	}
	~Optimizer() {
		if (m_own) tcnn_optimizer_destroy(m_own);
	}
	Optimizer(const Optimizer&) = delete;
	Optimizer& operator=(const Optimizer&) = delete;

	void bind(tcnn_trainable_model_t* tm) { m_tm = tm; }  // called by Trainer

private:
	json m_params;
	tcnn_trainable_model_t* m_tm = nullptr;
	tcnn_optimizer_t* m_own = nullptr;
};

template <typename T>
Optimizer<T>* create_optimizer(const json& params) { return new Optimizer<T>(params); }

}  // namespace tcnn
