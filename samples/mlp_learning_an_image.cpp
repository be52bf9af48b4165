// mlp_learning_an_image.cpp -- the reference's demo (samples/mlp_learning_an_image.cu:130-330) through the C++ facade:
// a 2-D hash-grid + fully fused MLP learns an RGB image from random pixel lookups, then renders it back.
//   samples/mlp_learning_an_image [image.ppm] [n_training_steps] [batch_size] [config.json]
// Input: a binary PPM (P6, 8 bit); without one a procedural test card is used.  The reference decodes JPEG / EXR through
// third-party loaders and samples the image with a bilinear texture fetch; here the lookup is done on the host side of
// the demo with the same bilinear filter, uploaded per batch.  Output: learned_image.ppm next to the binary, and the PSNR.
#include <tiny-cuda-nn/config.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>

static const char* DEFAULT_CONFIG = R"({
	"loss": {"otype": "RelativeL2"},
	"optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
	"encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5},
	"network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}
})";

struct Image {
	int w = 0, h = 0;
	std::vector<float> rgb;  // [h][w][3], 0..1
	void bilinear(float u, float v, float* out) const {  // u, v in [0, 1): texel centres at (i + 0.5) / w
		const float x = std::min(std::max(u * w - 0.5f, 0.0f), float(w - 1)), y = std::min(std::max(v * h - 0.5f, 0.0f), float(h - 1));
		const int x0 = int(x), y0 = int(y), x1 = std::min(x0 + 1, w - 1), y1 = std::min(y0 + 1, h - 1);
		const float fx = x - x0, fy = y - y0;
		for (int c = 0; c < 3; ++c) {
			const float top = rgb[(size_t(y0) * w + x0) * 3 + c] * (1 - fx) + rgb[(size_t(y0) * w + x1) * 3 + c] * fx;
			const float bot = rgb[(size_t(y1) * w + x0) * 3 + c] * (1 - fx) + rgb[(size_t(y1) * w + x1) * 3 + c] * fx;
			out[c] = top * (1 - fy) + bot * fy;
		}
	}
};

static Image load_ppm(const char* path) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error(std::string("cannot open ") + path);
	std::string magic;
	int maxval = 0;
	Image img;
	auto next_token = [&](auto& value) {
		for (;;) {
			f >> std::ws;
			if (f.peek() == '#') { std::string line; std::getline(f, line); continue; }
			f >> value;
			return;
		}
	};
	next_token(magic);
	next_token(img.w);
	next_token(img.h);
	next_token(maxval);
	if (magic != "P6" || maxval != 255 || img.w <= 0 || img.h <= 0) throw std::runtime_error("expected a binary 8-bit PPM (P6)");
	f.get();
	std::vector<unsigned char> bytes(size_t(img.w) * img.h * 3);
	f.read(reinterpret_cast<char*>(bytes.data()), std::streamsize(bytes.size()));
	if (!f) throw std::runtime_error("truncated PPM");
	img.rgb.resize(bytes.size());
	for (size_t i = 0; i < bytes.size(); ++i) img.rgb[i] = bytes[i] / 255.0f;
	return img;
}

static Image test_card(int w, int h) {  // smooth gradients + rings + a checker: low and high frequencies
	Image img;
	img.w = w;
	img.h = h;
	img.rgb.resize(size_t(w) * h * 3);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			const float u = (x + 0.5f) / w, v = (y + 0.5f) / h, r = std::hypot(u - 0.5f, v - 0.5f);
			float* p = &img.rgb[(size_t(y) * w + x) * 3];
			p[0] = 0.5f + 0.5f * std::sin(40.0f * r);
			p[1] = u * (1 - v) + (((x / 16) + (y / 16)) % 2 ? 0.25f : 0.0f);
			p[2] = 0.5f + 0.5f * std::cos(12.0f * u) * std::sin(9.0f * v);
		}
	return img;
}

static void save_ppm(const char* path, int w, int h, const std::vector<float>& rgb_rows3) {
	std::ofstream f(path, std::ios::binary);
	f << "P6\n" << w << " " << h << "\n255\n";
	for (float v : rgb_rows3) f.put(char(std::lround(std::min(std::max(v, 0.0f), 1.0f) * 255.0f)));
}

int main(int argc, char** argv) {
	try {
		const Image image = (argc > 1 && std::string(argv[1]) != "-") ? load_ppm(argv[1]) : test_card(512, 512);
		const uint32_t n_training_steps = argc > 2 ? uint32_t(atoi(argv[2])) : 1000;
		const uint32_t batch_size = tcnn::next_multiple(argc > 3 ? uint32_t(atoi(argv[3])) : (1u << 16), tcnn::batch_size_granularity());
		std::string config = DEFAULT_CONFIG;
		if (argc > 4) {
			std::ifstream cf(argv[4]);
			if (!cf) throw std::runtime_error(std::string("cannot open ") + argv[4]);
			std::stringstream ss;
			ss << cf.rdbuf();
			config = ss.str();
		}
		const uint32_t n_input_dims = 2, n_output_dims = 3;
		auto model = tcnn::create_from_config(n_input_dims, n_output_dims, config);
		std::printf("image %dx%d, %zu parameters, batch %u, %u steps\n", image.w, image.h, model.network->n_params(), batch_size, n_training_steps);

		hipStream_t stream;
		tcnn::hip_check(hipStreamCreate(&stream), "hipStreamCreate");
		tcnn::GPUMatrix<float> training_batch(n_input_dims, batch_size), training_target(n_output_dims, batch_size);
		std::vector<float> xs(size_t(n_input_dims) * batch_size), ys(size_t(n_output_dims) * batch_size);
		std::mt19937 rng(1337);
		std::uniform_real_distribution<float> uni(0.f, 1.f);
		for (uint32_t i = 0; i < n_training_steps; ++i) {
			for (uint32_t k = 0; k < batch_size; ++k) {  // random lookups into the image (mlp_learning_an_image.cu:262-268)
				const float u = uni(rng), v = uni(rng);
				xs[size_t(k) * 2] = u;
				xs[size_t(k) * 2 + 1] = v;
				image.bilinear(u, v, &ys[size_t(k) * 3]);
			}
			training_batch.copy_from_host(xs);
			training_target.copy_from_host(ys);
			auto ctx = model.trainer->training_step(stream, training_batch, training_target);
			if (i == 0 || (i + 1) % 250 == 0 || i + 1 == n_training_steps) std::printf("step=%u loss=%g\n", i + 1, model.trainer->loss(stream, *ctx));
		}

		// render every pixel centre (mlp_learning_an_image.cu:229-252, 311-325)
		const uint32_t n_coords = uint32_t(image.w) * image.h, n_padded = tcnn::next_multiple(n_coords, tcnn::batch_size_granularity());
		std::vector<float> coords(size_t(n_padded) * 2, 0.0f);
		for (int y = 0; y < image.h; ++y)
			for (int x = 0; x < image.w; ++x) {
				coords[(size_t(y) * image.w + x) * 2] = (x + 0.5f) / image.w;
				coords[(size_t(y) * image.w + x) * 2 + 1] = (y + 0.5f) / image.h;
			}
		tcnn::GPUMatrix<float> inference_batch(n_input_dims, n_padded), prediction(n_output_dims, n_padded);
		inference_batch.copy_from_host(coords);
		model.network->inference(stream, inference_batch, prediction);
		tcnn::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
		std::vector<float> out = prediction.to_cpu_vector();
		out.resize(size_t(n_coords) * 3);
		double mse = 0.0;
		for (size_t k = 0; k < out.size(); ++k) mse += double(out[k] - image.rgb[k]) * double(out[k] - image.rgb[k]);
		mse /= double(out.size());
		const double psnr = 10.0 * std::log10(1.0 / std::max(mse, 1e-12));
		save_ppm("learned_image.ppm", image.w, image.h, out);
		std::printf("psnr=%.2f dB (mse %g); wrote learned_image.ppm\n", psnr, mse);
		(void)hipStreamDestroy(stream);
		return psnr > 20.0 ? 0 : 1;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 2;
	}
}
