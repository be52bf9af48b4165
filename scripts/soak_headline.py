import os, sys, time
sys.path.insert(0, "tiny-cuda-nn_amd")
import torch, tinycudann as tcnn, math
cfg = {"loss": {"otype": "RelativeL2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
       "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
       "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}
tm = tcnn.create_from_config(3, 4, cfg)
n = 1 << 18
g = torch.Generator(device="cuda"); g.manual_seed(0)
def batch():
    x = torch.rand((n, 3), generator=g, device="cuda")
    t = torch.stack([0.5 + 0.5 * torch.sin(6.2831853 * (c + 1) * x[:, 0]) * torch.cos(6.2831853 * (c % 3 + 1) * x[:, 1]) * torch.sin(3.1 * x[:, 2] + c) for c in range(4)], 1).contiguous()
    return x, t
t0 = time.perf_counter()
for step in range(10001):
    x, t = batch()
    ctx = tm.training_step(x, t, want_context=(step % 2000 == 0))
    if step % 2000 == 0:
        l = tm.loss(ctx); print(step, l, flush=True); assert math.isfinite(l)
torch.cuda.synchronize(); print("seconds", time.perf_counter() - t0)
w = tm.params_full_precision; print("finite params", bool(torch.isfinite(w).all()), float(w.abs().max()))
