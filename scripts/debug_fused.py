import sys, os
sys.path.insert(0, "tiny-cuda-nn_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import tinycudann as T
from conftest import config_hash
from test_gpu_parity import positions, targets_for
n, log2_t = 1 << 16, 15
cfg = config_hash(log2_hashmap_size=log2_t, per_level_scale=1.5)
a, b = T.create_from_config(3, 4, cfg, seed=3), T.create_from_config(3, 4, cfg, seed=3)
b.set_fused_optimizer(False)
for tm in (a, b):
    w = tm.params_full_precision.clone(); w[tm.n_mlp_params:] *= 1.0e3; tm.set_params_full_precision(w)
pos = positions(n, 3, seed=41)
x, t = torch.from_numpy(pos).cuda(), torch.from_numpy(targets_for(pos, 4)).cuda()
a.training_step(x, t); b.training_step(x, t)
nm = a.n_mlp_params
for name, u, v in (("master", a.params_full_precision, b.params_full_precision), ("grads", a.param_gradients.float(), b.param_gradients.float()),
                   ("m1", a.optimizer_state()[0], b.optimizer_state()[0]), ("m2", a.optimizer_state()[1], b.optimizer_state()[1]), ("steps", a.optimizer_state()[2], b.optimizer_state()[2])):
    d = (u != v).nonzero().flatten().cpu().numpy()
    print(name, "mismatches", d.size, "mlp part", int((d < nm).sum()), "first", d[:8], "vals", u[d[:4]].cpu().numpy() if d.size else "", v[d[:4]].cpu().numpy() if d.size else "")
    if d.size and name == "master":
        ulp = (u[d].view(torch.int32) - v[d].view(torch.int32)).abs().max().item(); print("  max ulp diff", ulp)
