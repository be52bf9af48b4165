"""Forward gather cost per kind of level (is a dense coarse level cheaper than a hashed one?).  8 copies of one level fill all XCDs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
C = tcnn._C
n = 1 << 18
x = torch.rand((n, 3), device="cuda")
def enc(base, levels=8, scale=1.0): return {"otype": "HashGrid", "n_levels": levels, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": base, "per_level_scale": scale}
cases = {"8 x level 0 (res 17, 19.6 KB)": enc(16), "8 x level 1 (res 33, 143 KB)": enc(32), "8 x level 2 (res 65, 1.1 MB dense)": enc(64),
         "8 x hashed (res 513)": enc(512), "8 x hashed (res 8193)": enc(8192),
         "headline 16 levels": enc(16, 16, 2.0), "16 hashed levels (base 128)": enc(128, 16, 1.3)}
for name, e in cases.items():
    m = C.create_encoding(3, e)
    p = (torch.rand(m.n_params(), device="cuda") - 0.5).half()
    for _ in range(5): m.fwd(x, p)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): m.fwd(x, p)
    b.record(); torch.cuda.synchronize()
    print(f"{name:40s} {a.elapsed_time(b) / 20:8.4f} ms (module forward: gather + output allocation)")
