/*
 * tiny-cuda-nn/optimizer.h -- Optimizer<T> + create_optimizer<T>(json) (reference optimizer.h:40-99,
 * src/optimizer.cu:50-86).  Adam, optionally wrapped in Ema / ExponentialDecay, runs inside the library; this object
 * carries the configuration to the Trainer built from it and, once bound, reads the live state back.
 */
#pragma once
#include <tiny-cuda-nn/common.h>

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
	}
	uint32_t step() const { return m_tm ? tcnn_trainer_optimizer_step_count(m_tm) : 0u; }  // optimizer.h:66

	void bind(tcnn_trainable_model_t* tm) { m_tm = tm; }  // called by Trainer

private:
	json m_params;
	tcnn_trainable_model_t* m_tm = nullptr;
};

template <typename T>
Optimizer<T>* create_optimizer(const json& params) { return new Optimizer<T>(params); }

}  // namespace tcnn
