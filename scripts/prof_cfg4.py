import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch, tinycudann as tcnn
ADAM = {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}
enc = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 22, "base_resolution": 16, "per_level_scale": 1.5}
net = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 4}
tm = tcnn.create_from_config(3, 16, {"loss": {"otype": "RelativeL2"}, "optimizer": ADAM, "encoding": enc, "network": net})
n = 1 << 18
x = torch.rand((n, 3), device="cuda"); t = torch.rand((n, 16), device="cuda")
for _ in range(10): tm.training_step(x, t, want_context=False)
tm.set_profiling(True)
for _ in range(20): tm.training_step(x, t, want_context=False)
torch.cuda.synchronize()
print({k: round(ms / max(c, 1), 4) for k, (ms, c) in tm.stage_times().items()})
