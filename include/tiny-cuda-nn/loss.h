/*
 * tiny-cuda-nn/loss.h -- Loss<T> + create_loss<T>(json) (reference loss.h:40-77, src/loss.cu:49-88).  Inside a training step the
 * loss lives in the library's fused kernels; this object carries its configuration to the Trainer that is built from it, and
 * evaluate() runs it on its own for hosts that call it directly (tcnn_loss_evaluate).
 */
#pragma once
#include <tiny-cuda-nn/common.h>
#include <tiny-cuda-nn/gpu_matrix.h>

namespace tcnn {

template <typename T>
class Loss {
public:
	explicit Loss(const json& params) : m_params(params) {}
	void update_hyperparams(const json& params) { m_params = params; }
	json hyperparams() const { return m_params; }

	// loss.h:42-50: prediction / gradients `padded width` x n, target (and data_pdf) `dims` x n, values `padded width` x n, column-major
	void evaluate(hipStream_t stream, const float loss_scale, const GPUMatrix<T>& prediction, const GPUMatrix<float>& target, GPUMatrix<float>& values,
	              GPUMatrix<T>& gradients, const GPUMatrix<float>* data_pdf = nullptr) const {
		if (prediction.n() != target.n() || gradients.m() != prediction.m() || gradients.n() != prediction.n() || values.m() != prediction.m() || values.n() != prediction.n()) {
			throw std::runtime_error("Loss::evaluate: matrix sizes do not match");
		}
		check(tcnn_loss_evaluate(m_params.value("otype", std::string("RelativeL2")).c_str(), (tcnn_stream_t)stream, prediction.n(), prediction.m(), target.m(), loss_scale,
		                         prediction.data(), target.data(), data_pdf ? data_pdf->data() : nullptr, values.data(), gradients.data()));
	}
	void evaluate(const float loss_scale, const GPUMatrix<T>& prediction, const GPUMatrix<float>& target, GPUMatrix<float>& values, GPUMatrix<T>& gradients,
	              const GPUMatrix<float>* data_pdf = nullptr) const {
		evaluate(nullptr, loss_scale, prediction, target, values, gradients, data_pdf);
	}

private:
	json m_params;
};

template <typename T>
Loss<T>* create_loss(const json& params) { return new Loss<T>(params); }

}  // namespace tcnn
