/*
 * The training loop of this file (the part between the markers "reference lines 214-311") reproduces, line for line, the caller code
 * of the reference's samples/mlp_learning_an_image.cu, as a fixture that proves the facade headers are a drop-in for it.  That code is
 *
 * Copyright (c) 2020-2025, NVIDIA CORPORATION.  All rights reserved.
 *
 * Redistribution and use in source and binary forms, with or without modification, are permitted
 * provided that the following conditions are met:
 *     * Redistributions of source code must retain the above copyright notice, this list of
 *       conditions and the following disclaimer.
 *     * Redistributions in binary form must reproduce the above copyright notice, this list of
 *       conditions and the following disclaimer in the documentation and/or other materials
 *       provided with the distribution.
 *     * Neither the name of the NVIDIA CORPORATION nor the names of its contributors may be used
 *       to endorse or promote products derived from this software without specific prior written
 *       permission.
 *
 * THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR
 * IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND
 * FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED. IN NO EVENT SHALL NVIDIA CORPORATION BE LIABLE
 * FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING,
 * BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS;
 * OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT,
 * STRICT LIABILITY, OR TOR (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE
 * OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
 */
// mlp_learning_an_image.hip -- the reference's demo (samples/mlp_learning_an_image.cu) against the MI355X facade:
// a 2-D encoding + fully fused MLP learns an RGB image from random pixel lookups, then renders it back.
//   samples/mlp_learning_an_image [image.ppm | -] [config.json | -] [n_training_steps] [output.ppm]
// The training loop below (reference lines 214-311) is the reference's, line for line; what differs is what has to:
//   * `hip*` for `cuda*`; the bilinear lookup is an explicit device function instead of a CUDA texture fetch
//     (same filter: normalised coordinates, texel centres at (i + 0.5) / size, clamp addressing);
//   * images are binary PPM (P6) -- the reference decodes JPEG / EXR through third-party loaders (stbi, tinyexr); without
//     an image a procedural test card is used; the result is written as PPM and its PSNR against the image is printed.
#include <tiny-cuda-nn/config.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

using namespace tcnn;
using precision_t = network_precision_t;

struct Texture {  // stands where cudaTextureObject_t stands: RGBA float texels, linear filter, clamp, normalised coordinates
	const float* texels;
	int width, height;
};
__device__ inline void tex2D(const Texture& t, float u, float v, float (&out)[4]) {
	const float x = fminf(fmaxf(u * t.width - 0.5f, 0.0f), (float)(t.width - 1)), y = fminf(fmaxf(v * t.height - 0.5f, 0.0f), (float)(t.height - 1));
	const int x0 = (int)x, y0 = (int)y, x1 = min(x0 + 1, t.width - 1), y1 = min(y0 + 1, t.height - 1);
	const float fx = x - x0, fy = y - y0;
	for (int c = 0; c < 4; ++c) {
		const float top = t.texels[((size_t)y0 * t.width + x0) * 4 + c] * (1 - fx) + t.texels[((size_t)y0 * t.width + x1) * 4 + c] * fx;
		const float bot = t.texels[((size_t)y1 * t.width + x0) * 4 + c] * (1 - fx) + t.texels[((size_t)y1 * t.width + x1) * 4 + c] * fx;
		out[c] = top * (1 - fy) + bot * fy;
	}
}

template <uint32_t stride>
__global__ void eval_image(uint32_t n_elements, Texture texture, float* __restrict__ xs_and_ys, float* __restrict__ result) {  // reference lines 82-98
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	uint32_t output_idx = i * stride;
	uint32_t input_idx = i * 2;
	float val[4];
	tex2D(texture, xs_and_ys[input_idx], xs_and_ys[input_idx + 1], val);
	result[output_idx + 0] = val[0];
	result[output_idx + 1] = val[1];
	result[output_idx + 2] = val[2];
	for (uint32_t i = 3; i < stride; ++i) result[output_idx + i] = 1;
}

static GPUMemory<float> load_image(const std::string& filename, int& width, int& height) {  // RGBA float texels, as the reference's loader returns
	std::vector<float> rgba;
	if (filename == "-") {  // procedural test card: smooth gradients + rings + a checker (low and high frequencies)
		width = height = 512;
		rgba.resize((size_t)width * height * 4);
		for (int y = 0; y < height; ++y)
			for (int x = 0; x < width; ++x) {
				const float u = (x + 0.5f) / width, v = (y + 0.5f) / height, r = std::hypot(u - 0.5f, v - 0.5f);
				float* p = &rgba[((size_t)y * width + x) * 4];
				p[0] = 0.5f + 0.5f * std::sin(40.0f * r);
				p[1] = u * (1 - v) + (((x / 16) + (y / 16)) % 2 ? 0.25f : 0.0f);
				p[2] = 0.5f + 0.5f * std::cos(12.0f * u) * std::sin(9.0f * v);
				p[3] = 1.0f;
			}
	} else {
		std::ifstream f(filename, std::ios::binary);
		if (!f) throw std::runtime_error("cannot open " + filename);
		std::string magic;
		int maxval = 0;
		auto next_token = [&](auto& value) {
			for (;;) {
				f >> std::ws;
				if (f.peek() == '#') { std::string line; std::getline(f, line); continue; }
				f >> value;
				return;
			}
		};
		next_token(magic);
		next_token(width);
		next_token(height);
		next_token(maxval);
		if (magic != "P6" || maxval != 255 || width <= 0 || height <= 0) throw std::runtime_error("expected a binary 8-bit PPM (P6)");
		f.get();
		std::vector<unsigned char> bytes((size_t)width * height * 3);
		f.read(reinterpret_cast<char*>(bytes.data()), (std::streamsize)bytes.size());
		if (!f) throw std::runtime_error("truncated PPM");
		rgba.resize((size_t)width * height * 4);
		for (size_t i = 0; i < (size_t)width * height; ++i) {
			for (int c = 0; c < 3; ++c) rgba[i * 4 + c] = bytes[i * 3 + c] / 255.0f;
			rgba[i * 4 + 3] = 1.0f;
		}
	}
	GPUMemory<float> result(rgba.size());
	result.copy_from_host(rgba);
	return result;
}

template <typename T>
static std::vector<float> save_image(const T* image, int width, int height, int n_channels, int channel_stride, const std::string& filename) {  // reference lines 63-80
	std::vector<T> host_data((size_t)width * height * channel_stride);
	HIP_CHECK_THROW(hipMemcpy(host_data.data(), image, host_data.size() * sizeof(T), hipMemcpyDeviceToHost));
	std::vector<float> rgb((size_t)width * height * n_channels);
	for (size_t i = 0; i < (size_t)width * height; ++i)
		for (int k = 0; k < n_channels; ++k) rgb[i * n_channels + k] = (float)host_data[i * channel_stride + k];
	if (!filename.empty()) {
		std::ofstream f(filename, std::ios::binary);
		f << "P6\n" << width << " " << height << "\n255\n";
		for (float v : rgb) f.put((char)std::lround(std::min(std::max(v, 0.0f), 1.0f) * 255.0f));
	}
	return rgb;
}

int main(int argc, char* argv[]) {
	try {
		json config = json::parse(std::string(R"({
			"loss": {"otype": "RelativeL2"},
			"optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
			"encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5},
			"network": {"otype": "FullyFusedMLP", "n_neurons": 64, "n_hidden_layers": 2, "activation": "ReLU", "output_activation": "None"}
		})"));  // data/config_hash.json

		if (argc >= 3 && std::string(argv[2]) != "-") {
			std::cout << "Loading custom json config '" << argv[2] << "'." << std::endl;
			std::ifstream f{argv[2]};
			std::stringstream ss;
			ss << f.rdbuf();
			config = json::parse(ss.str());
		}

		// First step: load an image that we'd like to learn
		int width, height;
		GPUMemory<float> image = load_image(argc >= 2 ? argv[1] : "-", width, height);

		// Second step: the image as a texture; it is used to generate training data efficiently on the fly
		Texture texture{image.data(), width, height};

		// Third step: sample a reference image.  Comparison of this reference image and the learned function is possible at the end.
		int sampling_width = width;
		int sampling_height = height;

		uint32_t n_coords = sampling_width * sampling_height;
		uint32_t n_coords_padded = next_multiple(n_coords, BATCH_SIZE_GRANULARITY);

		GPUMemory<float> sampled_image(n_coords * 3);
		GPUMemory<float> xs_and_ys(n_coords_padded * 2);

		std::vector<float> host_xs_and_ys(n_coords_padded * 2, 0.0f);
		for (int y = 0; y < sampling_height; ++y) {
			for (int x = 0; x < sampling_width; ++x) {
				int idx = (y * sampling_width + x) * 2;
				host_xs_and_ys[idx + 0] = (float)(x + 0.5) / (float)sampling_width;
				host_xs_and_ys[idx + 1] = (float)(y + 0.5) / (float)sampling_height;
			}
		}

		xs_and_ys.copy_from_host(host_xs_and_ys.data());

		linear_kernel(eval_image<3>, 0, nullptr, n_coords, texture, xs_and_ys.data(), sampled_image.data());

		const std::vector<float> reference_rgb = save_image(sampled_image.data(), sampling_width, sampling_height, 3, 3, "");

		// Fourth step: train the model by sampling the above image and optimizing an error metric

		// Various constants for the network and optimization
		const uint32_t batch_size = 1 << 18;
		const uint32_t n_training_steps = argc >= 4 ? atoi(argv[3]) : 1000;
		const uint32_t n_input_dims = 2;   // 2-D image coordinate
		const uint32_t n_output_dims = 3;  // RGB color

		hipStream_t inference_stream;
		HIP_CHECK_THROW(hipStreamCreate(&inference_stream));
		hipStream_t training_stream = inference_stream;

		default_rng_t rng{1337};

		// Auxiliary matrices for training
		GPUMatrix<float> training_target(n_output_dims, batch_size);
		GPUMatrix<float> training_batch(n_input_dims, batch_size);

		// Auxiliary matrices for evaluation
		GPUMatrix<float> prediction(n_output_dims, n_coords_padded);
		GPUMatrix<float> inference_batch(xs_and_ys.data(), n_input_dims, n_coords_padded);

		json encoding_opts = config.value("encoding", json::object());
		json loss_opts = config.value("loss", json::object());
		json optimizer_opts = config.value("optimizer", json::object());
		json network_opts = config.value("network", json::object());

		std::shared_ptr<Loss<precision_t>> loss{create_loss<precision_t>(loss_opts)};
		std::shared_ptr<Optimizer<precision_t>> optimizer{create_optimizer<precision_t>(optimizer_opts)};
		std::shared_ptr<NetworkWithInputEncoding<precision_t>> network = std::make_shared<NetworkWithInputEncoding<precision_t>>(n_input_dims, n_output_dims, encoding_opts, network_opts);
		network->set_jit_fusion(tcnn::supports_jit_fusion());

		auto trainer = std::make_shared<Trainer<float, precision_t, precision_t>>(network, optimizer, loss);

		std::chrono::steady_clock::time_point begin = std::chrono::steady_clock::now();

		float tmp_loss = 0;
		uint32_t tmp_loss_counter = 0;

		std::cout << "Beginning optimization with " << n_training_steps << " training steps." << std::endl;

		uint32_t interval = 10;

		for (uint32_t i = 0; i < n_training_steps; ++i) {
			bool print_loss = i % interval == 0;

			// Compute reference values at random coordinates
			{
				generate_random_uniform<float>(training_stream, rng, batch_size * n_input_dims, training_batch.data());
				linear_kernel(eval_image<n_output_dims>, 0, training_stream, batch_size, texture, training_batch.data(), training_target.data());
			}

			// Training step
			{
				auto ctx = trainer->training_step(training_stream, training_batch, training_target);

				if (i % std::min(interval, (uint32_t)100) == 0) {
					tmp_loss += trainer->loss(training_stream, *ctx);
					++tmp_loss_counter;
				}
			}

			// Debug outputs
			{
				if (print_loss) {
					std::chrono::steady_clock::time_point end = std::chrono::steady_clock::now();
					std::cout << "Step#" << i << ": " << "loss=" << tmp_loss / (float)tmp_loss_counter << " time=" << std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count() << "[µs]" << std::endl;

					tmp_loss = 0;
					tmp_loss_counter = 0;
					begin = std::chrono::steady_clock::now();
				}
			}

			if (print_loss && i > 0 && interval < 1000) {
				interval *= 10;
			}
		}

		// Dump the final image and compare it with the reference
		network->inference(inference_stream, inference_batch, prediction);
		HIP_CHECK_THROW(hipStreamSynchronize(inference_stream));
		const std::string out_name = argc >= 5 ? argv[4] : "learned_image.ppm";
		const std::vector<float> learned_rgb = save_image(prediction.data(), sampling_width, sampling_height, 3, n_output_dims, out_name);
		double mse = 0.0;
		for (size_t k = 0; k < learned_rgb.size(); ++k) mse += double(learned_rgb[k] - reference_rgb[k]) * double(learned_rgb[k] - reference_rgb[k]);
		mse /= double(learned_rgb.size());
		const double psnr = 10.0 * std::log10(1.0 / std::max(mse, 1e-12));
		std::printf("psnr=%.2f dB (mse %g); wrote %s\n", psnr, mse, out_name.c_str());

		free_all_gpu_memory_arenas();
		HIP_CHECK_THROW(hipStreamDestroy(inference_stream));
		return psnr > 20.0 ? EXIT_SUCCESS : EXIT_FAILURE;
	} catch (const std::exception& e) {
		std::cout << "Uncaught exception: " << e.what() << std::endl;
		return 2;
	}
}
