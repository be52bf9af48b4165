"""Where does the grid-backward time go?  Times the encoding backward alone for several level mixes / modes."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
C = tcnn._C
n = 1 << 18
x = torch.rand((n, 3), device="cuda")
cases = {
    "headline L16 base16 T19": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
    "hashed-only 13 levels (base 128)": {"otype": "HashGrid", "n_levels": 13, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 128, "per_level_scale": 2.0},
    "level 0 only (4096 entries)": {"otype": "HashGrid", "n_levels": 1, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
    "levels 0-2 (dense)": {"otype": "HashGrid", "n_levels": 3, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
    "level 1 only (32768 entries)": {"otype": "HashGrid", "n_levels": 1, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 32, "per_level_scale": 2.0},
    "level 2 only (262144 dense)": {"otype": "HashGrid", "n_levels": 1, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 64, "per_level_scale": 2.0},
    "4 hashed levels (base 512)": {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 512, "per_level_scale": 2.0},
    "one hashed level (base 512)": {"otype": "HashGrid", "n_levels": 1, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 512, "per_level_scale": 2.0},
}
for name, enc in cases.items():
    m = C.create_encoding(3, enc)
    p = (torch.rand(m.n_params(), device="cuda") - 0.5).half().requires_grad_(True)
    xx = x.clone()
    ctx, y = m.fwd(xx, p)
    dy = (torch.randn_like(y.float()) * 0.01).half()
    for mode, mname in ((3, "bucketed"),):
        C.set_grid_backward_mode(mode)
        for _ in range(3):
            m.bwd(ctx, xx, p, y, dy)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            m.bwd(ctx, xx, p, y, dy)
        b.record(); torch.cuda.synchronize()
        print(f"{name:36s} {mname:10s} {a.elapsed_time(b)/10:8.4f} ms (incl. torch.empty of grads)")
