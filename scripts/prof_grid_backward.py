"""Headline-config grid backward alone, 30 launches (for rocprofv3 --kernel-trace --stats per-kernel averages)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
C = tcnn._C
n = 1 << 18
enc = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}
x = torch.rand((n, 3), device="cuda")
m = C.create_encoding(3, enc)
p = (torch.rand(m.n_params(), device="cuda") - 0.5).half().requires_grad_(True)
ctx, y = m.fwd(x, p)
dy = (torch.randn_like(y.float()) * 0.01).half()
for _ in range(30):
    m.bwd(ctx, x, p, y, dy)
torch.cuda.synchronize()
