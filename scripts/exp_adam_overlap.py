"""Upper bound for overlapping the optimizer step with the grid backward of OTHER levels: Adam of one model (HBM-bound) on one
stream, the grid backward (record scatter + owner pass: LDS-bound) of an independent encoding of the same size on another."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
C = tcnn._C
n = 1 << 18
ENC = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}
cfg = {"loss": {"otype": "RelativeL2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
       "encoding": ENC, "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}
x = torch.rand((n, 3), device="cuda")
t = torch.rand((n, 4), device="cuda")
tm = tcnn.create_from_config(3, 4, cfg)
for _ in range(5): tm.training_step(x, t, want_context=False)          # gradients in place, Adam state warm
m = C.create_encoding(3, ENC)
p = (torch.rand(m.n_params(), device="cuda") - 0.5).half().requires_grad_(True)
ctx, y = m.fwd(x, p)
dy = (torch.randn_like(y.float()) * 0.01).half()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(sa, sb, what):
    if what in ("both", "backward"):
        with torch.cuda.stream(sa): m.bwd(ctx, x, p, y, dy)
    if what in ("both", "adam"):
        with torch.cuda.stream(sb): tm.optimizer_step()
for sa, sb, what, name in ((s1, s1, "backward", "grid backward alone"), (s1, s1, "adam", "Adam alone"), (s1, s1, "both", "serial (one stream)"), (s1, s2, "both", "concurrent (two streams)")):
    for _ in range(5): run(sa, sb, what)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run(sa, sb, what)
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms")
