/*
 * tiny-cuda-nn/config.h -- TrainableModel + create_from_config (reference config.h:46-63): the one call a host makes to
 * get {loss, optimizer, network, trainer} for the HashGrid + FullyFusedMLP hot path on MI355X.
 *
 *     auto model = tcnn::create_from_config(n_in, n_out, config);           (config.h:53-63)
 *     auto ctx   = model.trainer->training_step(stream, inputs, targets);   (trainer.h:254-357)
 *     float loss = model.trainer->loss(stream, *ctx);                       (trainer.h:372-374)
 *     model.network->inference(stream, inputs, outputs);                    (object.h:214-271)
 *
 * `config` is a tcnn::json (nlohmann::json when its header is on the include path, see common.h) or JSON text.
 */
#pragma once
#include <tiny-cuda-nn/gpu_memory.h>
#include <tiny-cuda-nn/random.h>
#include <tiny-cuda-nn/trainer.h>

namespace tcnn {

using precision_t = network_precision_t;

struct TrainableModel {  // config.h:46-51
	std::shared_ptr<Loss<network_precision_t>> loss;
	std::shared_ptr<Optimizer<network_precision_t>> optimizer;
	std::shared_ptr<NetworkWithInputEncoding<network_precision_t>> network;
	std::shared_ptr<Trainer<float, network_precision_t, network_precision_t>> trainer;
};

inline TrainableModel create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, json config, uint32_t seed = 1337) {  // config.h:53-63
	json loss_opts = config.value("loss", json::object());
	json optimizer_opts = config.value("optimizer", json::object());
	json network_opts = config.value("network", json::object());
	json encoding_opts = config.value("encoding", json::object());
	std::shared_ptr<Loss<network_precision_t>> loss{create_loss<network_precision_t>(loss_opts)};
	std::shared_ptr<Optimizer<network_precision_t>> optimizer{create_optimizer<network_precision_t>(optimizer_opts)};
	auto network = std::make_shared<NetworkWithInputEncoding<network_precision_t>>(n_input_dims, n_output_dims, encoding_opts, network_opts);
	auto trainer = std::make_shared<Trainer<float, network_precision_t, network_precision_t>>(network, optimizer, loss, seed);
	return {loss, optimizer, network, trainer};
}
inline TrainableModel create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const std::string& config_json_text, uint32_t seed = 1337) {
	return create_from_config(n_input_dims, n_output_dims, json::parse(config_json_text), seed);
}
inline TrainableModel create_from_config(uint32_t n_input_dims, uint32_t n_output_dims, const char* config_json_text, uint32_t seed = 1337) {
	return create_from_config(n_input_dims, n_output_dims, json::parse(std::string(config_json_text)), seed);
}

}  // namespace tcnn
