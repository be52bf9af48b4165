"""Would pipelining the grid backward over two level groups on two streams pay?  Two independent 8-level encodings'
backward passes, serial on one stream vs concurrent on two (an upper bound for scatter(group 2) || owner(group 1))."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
C = tcnn._C
n = 1 << 18
x = torch.rand((n, 3), device="cuda")
def enc(base): return {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": base, "per_level_scale": 2.0}
mods = []
for base in (16, 4096):  # levels 0-7 and 8-15 of the headline encoding
    m = C.create_encoding(3, enc(base))
    p = (torch.rand(m.n_params(), device="cuda") - 0.5).half().requires_grad_(True)
    ctx, y = m.fwd(x, p)
    dy = (torch.randn_like(y.float()) * 0.01).half()
    mods.append((m, p, ctx, y, dy))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(streams):
    for (m, p, ctx, y, dy), s in zip(mods, streams):
        with torch.cuda.stream(s):
            m.bwd(ctx, x, p, y, dy)
for streams, name in (((s1, s1), "serial (one stream)"), ((s1, s2), "concurrent (two streams)")):
    for _ in range(5): run(streams)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): run(streams)
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per pair of backward passes")
