/*
 * tiny-cuda-nn/loss.h -- Loss<T> + create_loss<T>(json) (reference loss.h:40-77, src/loss.cu:49-88).  The loss lives
 * inside the library's training kernels; this object carries its configuration to the Trainer that is built from it.
 */
#pragma once
#include <tiny-cuda-nn/common.h>

namespace tcnn {

template <typename T>
class Loss {
public:
	explicit Loss(const json& params) : m_params(params) {}
	void update_hyperparams(const json& params) { m_params = params; }
	json hyperparams() const { return m_params; }

private:
	json m_params;
};

template <typename T>
Loss<T>* create_loss(const json& params) { return new Loss<T>(params); }

}  // namespace tcnn
