"""Grid backward (record scatter + owner pass) against the number of levels: time = fixed + per-level?  All levels hashed (T = 2^19), N = 2^18.
usage: python scripts/exp_bwd_levels.py L    (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
C = tcnn._C
L = int(sys.argv[1])
n = 1 << 18
x = torch.rand((n, 3), device="cuda")
enc = {"otype": "HashGrid", "n_levels": L, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 512, "per_level_scale": 1.15}
m = C.create_encoding(3, enc)
p = (torch.rand(m.n_params(), device="cuda") - 0.5).half().requires_grad_(True)
ctx, y = m.fwd(x, p)
dy = (torch.randn_like(y.float()) * 0.01).half()
for _ in range(5): m.bwd(ctx, x, p, y, dy)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30): m.bwd(ctx, x, p, y, dy)
b.record(); torch.cuda.synchronize()
print(f"L={L} backward (scatter + owner + input-gradient kernels) {a.elapsed_time(b) / 30:.4f} ms")
