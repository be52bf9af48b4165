#!/usr/bin/env python3
"""The PyTorch-binding training loop alone (bench.py's torch_binding leg: tcnn.NetworkWithInputEncoding -> RelativeL2 in torch -> backward ->
torch.optim.Adam), for `rocprofv3 --kernel-trace --stats`: wall time per step next to the kernels' total says whether the loop is bound by the
GPU or by the host's launch rate.   usage: prof_torch_binding.py [hash|hash_shipped] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("TCNN_PKG_DIR") or os.path.join(ROOT, "tiny-cuda-nn_amd"))  # (TCNN_PKG_DIR: A/B against another copy of the Python package)
import torch  # noqa: E402
import tinycudann as tcnn  # noqa: E402  (before bench: bench.py puts the in-tree package first on sys.path)
import bench  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hash"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
rng = tcnn._C.Pcg32(1337)
batches = bench.make_batches(w, bench.BATCH, 4, rng, device=dev, tcnn=tcnn)
fresh = bench.make_batches(w, bench.BATCH, 1, rng, device=dev, tcnn=tcnn)[0]
out = bench.torch_binding_leg(w, tcnn, batches, fresh, rng, True, steps, 20, 1.0)
print({k: v for k, v in out.items() if k in ("ms_per_step", "gpu_ms_per_step", "final_loss")})
# host-side cost of the pieces (synchronised around each: upper bounds, not additive under asynchronous execution)
cfg = w["config"]
model = tcnn.NetworkWithInputEncoding(w["n_in"], w["n_out"], cfg["encoding"], cfg["network"], seed=1337)
opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
x, t = batches[0]
acc = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    a = acc.setdefault(name, [0.0, 0.0])
    a[0] += host
    a[1] += total
    return r


for i in range(60):
    o = timed("forward", lambda: model(x))
    loss = timed("loss", lambda: ((o - t.to(o.dtype)) ** 2 / (o.detach() ** 2 + 0.01)).mean())
    timed("zero_grad", lambda: opt.zero_grad())
    timed("backward", lambda: loss.backward())
    timed("adam", lambda: opt.step())
    if i == 9:
        acc.clear()
print({k: (round(v[0] / 50 * 1e3, 4), round(v[1] / 50 * 1e3, 4)) for k, v in acc.items()}, "(host ms to enqueue, ms until the GPU is done), per step")
