// learn_function.cpp -- the hot path through the C++ facade (include/tiny-cuda-nn/*.h), host code only, spelled the way a
// reference application spells it (README.md:40-67 / samples/mlp_learning_an_image.cu:214-311 of the reference):
//   create_from_config -> trainer->training_step -> trainer->loss -> network->inference.
// Learns a smooth 3-D -> 4-D function from synthetic samples, prints the loss curve, checks a snapshot round trip.
// Build: make -C samples      Run (MI355X): samples/learn_function [n_steps] [batch_size]
#include <tiny-cuda-nn/config.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

static const char* CONFIG = R"({
	"loss": {"otype": "RelativeL2"},
	"optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
	"encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
	"network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}
})";

int main(int argc, char** argv) {
	try {
		const uint32_t n_steps = argc > 1 ? uint32_t(atoi(argv[1])) : 200;
		const uint32_t batch_size = tcnn::next_multiple(argc > 2 ? uint32_t(atoi(argv[2])) : (1u << 16), tcnn::batch_size_granularity());
		const uint32_t n_input_dims = 3, n_output_dims = 4;

		auto model = tcnn::create_from_config(n_input_dims, n_output_dims, CONFIG);

		std::mt19937 rng(42);
		std::uniform_real_distribution<float> uni(0.f, 1.f);
		std::vector<float> xs(size_t(n_input_dims) * batch_size), ys(size_t(n_output_dims) * batch_size);
		for (uint32_t i = 0; i < batch_size; ++i) {
			float* x = &xs[size_t(i) * n_input_dims];
			for (uint32_t d = 0; d < n_input_dims; ++d) x[d] = uni(rng);
			float* y = &ys[size_t(i) * n_output_dims];
			y[0] = 0.5f + 0.5f * std::sin(6.2831853f * x[0]) * std::cos(6.2831853f * x[1]);
			y[1] = x[0] * x[1] + x[2];
			y[2] = std::exp(-8.f * ((x[0] - .5f) * (x[0] - .5f) + (x[1] - .5f) * (x[1] - .5f) + (x[2] - .5f) * (x[2] - .5f)));
			y[3] = 1.f;
		}
		tcnn::GPUMatrix<float> training_batch(n_input_dims, batch_size), training_target(n_output_dims, batch_size), prediction(n_output_dims, batch_size);
		training_batch.copy_from_host(xs);
		training_target.copy_from_host(ys);

		hipStream_t stream;
		tcnn::hip_check(hipStreamCreate(&stream), "hipStreamCreate");
		float first_loss = 0.f, last_loss = 0.f;
		for (uint32_t i = 0; i < n_steps; ++i) {
			auto ctx = model.trainer->training_step(stream, training_batch, training_target);
			if (i == 0 || (i + 1) % 50 == 0 || i + 1 == n_steps) {
				last_loss = model.trainer->loss(stream, *ctx);
				if (i == 0) first_loss = last_loss;
				std::printf("step=%u loss=%g\n", i + 1, last_loss);
			}
		}
		model.network->inference(stream, training_batch, prediction);
		tcnn::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
		const std::vector<float> pred = prediction.to_cpu_vector();
		double mse = 0.0;
		for (size_t k = 0; k < pred.size(); ++k) mse += (pred[k] - ys[k]) * (pred[k] - ys[k]);
		mse /= double(pred.size());
		std::printf("inference mse=%g\n", mse);

		// snapshot -> fresh model -> identical inference
		const auto snapshot = model.trainer->serialize(true);  // the snapshot document (nlohmann::json present) or its MessagePack bytes
		auto restored = tcnn::create_from_config(n_input_dims, n_output_dims, CONFIG, /*seed=*/7);
		restored.trainer->deserialize(snapshot);
		tcnn::GPUMatrix<float> prediction2(n_output_dims, batch_size);
		restored.network->inference(stream, training_batch, prediction2);
		tcnn::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
		const bool same = prediction2.to_cpu_vector() == pred;
		std::printf("snapshot bytes=%zu restored_inference_identical=%d\n", snapshot.size(), int(same));

		// GPUMatrixDynamic layouts (gpu_matrix.h:106-250): the same batch row-major (SoA), and column-major with a padded stride,
		// must give the same inference bits; the output is written row-major with a stride as well
		bool layouts_ok = true;
		{
			std::vector<float> soa(xs.size()), padded(size_t(n_input_dims + 1) * batch_size, -1.0f);
			for (uint32_t i = 0; i < batch_size; ++i)
				for (uint32_t d = 0; d < n_input_dims; ++d) {
					soa[size_t(d) * batch_size + i] = xs[size_t(i) * n_input_dims + d];
					padded[size_t(i) * (n_input_dims + 1) + d] = xs[size_t(i) * n_input_dims + d];
				}
			tcnn::GPUMatrixDynamic<float> in_rm(n_input_dims, batch_size, tcnn::RM);
			in_rm.copy_from_host(soa);
			tcnn::GPUMemory<float> padded_mem(padded.size());
			padded_mem.copy_from_host(padded);
			tcnn::GPUMatrixDynamic<float> in_strided(padded_mem.data(), n_input_dims, batch_size, tcnn::CM, n_input_dims + 1);
			tcnn::GPUMatrixDynamic<float> out_a(n_output_dims, batch_size), out_rm(n_output_dims, batch_size, tcnn::RM);
			model.network->inference(stream, in_rm, out_a);
			model.network->inference(stream, in_strided, out_rm);
			tcnn::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
			const std::vector<float> a = out_a.to_cpu_vector(), b = out_rm.to_cpu_vector();
			for (uint32_t i = 0; i < batch_size && layouts_ok; ++i)
				for (uint32_t j = 0; j < n_output_dims; ++j)
					layouts_ok = layouts_ok && a[size_t(i) * n_output_dims + j] == pred[size_t(i) * n_output_dims + j] && b[size_t(j) * batch_size + i] == pred[size_t(i) * n_output_dims + j];
			// a training step from the row-major batch gives the same loss as from the column-major one (no optimizer step)
			auto c1 = model.trainer->training_step(stream, training_batch, training_target, nullptr, /*run_optimizer=*/false);
			auto c2 = model.trainer->training_step(stream, in_rm, training_target, nullptr, /*run_optimizer=*/false);
			layouts_ok = layouts_ok && model.trainer->loss(stream, *c1) == model.trainer->loss(stream, *c2);
			std::printf("layouts_identical=%d\n", int(layouts_ok));
			// Loss<T>::evaluate on its own (loss.h:42-50) reproduces the gradient the fused training step computed from the same prediction
			using T = tcnn::network_precision_t;
			tcnn::GPUMatrix<T> prediction = c1->output(), reference_gradient = c1->dL_doutput();
			tcnn::GPUMatrix<T> gradient(prediction.m(), prediction.n());
			tcnn::GPUMatrix<float> values(prediction.m(), prediction.n());
			model.loss->evaluate(stream, 128.0f, prediction, training_target, values, gradient);
			HIP_CHECK_THROW(hipStreamSynchronize(stream));
			std::vector<uint16_t> g1(gradient.n_elements()), g2(gradient.n_elements());
			HIP_CHECK_THROW(hipMemcpy(g1.data(), gradient.data(), g1.size() * 2, hipMemcpyDeviceToHost));
			HIP_CHECK_THROW(hipMemcpy(g2.data(), reference_gradient.data(), g2.size() * 2, hipMemcpyDeviceToHost));
			const bool loss_ok = g1 == g2;
			std::printf("loss_evaluate_matches_training_step=%d\n", int(loss_ok));
			layouts_ok = layouts_ok && loss_ok;
			// Optimizer<T> on its own (optimizer.h:52-60): Adam over buffers this host owns -- one step moves every weight against its gradient
			{
				const uint32_t n_w = 1024;
				tcnn::json opt_cfg = tcnn::json::parse(R"({"otype": "Adam", "learning_rate": 1e-2})");
				tcnn::Optimizer<T> opt(opt_cfg);
				opt.allocate(n_w, {{16, 32}});
				tcnn::GPUMemory<float> w_fp(n_w);
				tcnn::GPUMemory<T> w_half(n_w), g_half(n_w);
				std::vector<float> w0(n_w, 0.5f);
				std::vector<T> h0(n_w, (T)0.5f), g0(n_w, (T)(128.0f * 0.25f));  // gradient 0.25 at loss scale 128
				w_fp.copy_from_host(w0);
				w_half.copy_from_host(h0);
				g_half.copy_from_host(g0);
				opt.step(stream, 128.0f, w_fp.data(), w_half.data(), g_half.data());
				HIP_CHECK_THROW(hipStreamSynchronize(stream));
				std::vector<float> w1(n_w);
				w_fp.copy_to_host(w1);
				bool moved = opt.step() == 1;
				for (uint32_t i = 0; i < n_w; ++i) moved = moved && w1[i] < 0.5f && w1[i] > 0.48f;  // first Adam step: -lr * sign(g)
				std::printf("optimizer_on_its_own=%d\n", int(moved));
				layouts_ok = layouts_ok && moved;
			}
			// GPUMatrix(m, n, stream) (gpu_matrix.h:141-152): temporaries out of the stream-ordered arena -- the block a destroyed matrix
			// hands back is what the next one of that size on the same stream receives (no driver allocation per iteration)
			{
				void* first = nullptr;
				bool reused = true;
				for (int it = 0; it < 4; ++it) {
					tcnn::GPUMatrix<float> tmp(n_output_dims, batch_size, stream);
					model.network->inference(stream, training_batch, tmp);
					if (it == 0) first = tmp.data();
					reused = reused && tmp.data() == first;
				}
				HIP_CHECK_THROW(hipStreamSynchronize(stream));
				std::printf("arena_block_reused=%d\n", int(reused));
				layouts_ok = layouts_ok && reused;
			}
		}

		// error behaviour: the reference throws std::runtime_error for a batch that is not a multiple of 256
		bool threw = false;
		try {
			tcnn::GPUMatrix<float> bad_in(n_input_dims, 100), bad_out(n_output_dims, 100);
			model.network->inference(stream, bad_in, bad_out);
		} catch (const std::runtime_error& e) { threw = true; std::printf("expected error: %s\n", e.what()); }

		(void)hipStreamDestroy(stream);
		const bool ok = std::isfinite(last_loss) && last_loss < 0.5f * first_loss && same && threw && layouts_ok;
		std::printf(ok ? "OK\n" : "FAILED\n");
		return ok ? 0 : 1;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 2;
	}
}
