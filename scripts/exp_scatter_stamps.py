"""The record scatter under a clock: k_grid_bucket_scatter stamped per workgroup (entry, first tile loaded, end of each of its first four tiles, level).
The instrumentation is NOT in the tree; scripts/exp_scatter_stamps.patch adds it to a working copy:
    git apply scripts/exp_scatter_stamps.patch && bash scripts/build_variant_one.sh stampsc grid_kernels "" && git apply -R scripts/exp_scatter_stamps.patch
    TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/stampsc.so python scripts/exp_scatter_stamps.py
Results: profiles/r04_exp_notes.txt section 15."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import bench  # noqa: E402
import tinycudann as tcnn  # noqa: E402

w = bench.WORKLOADS["hash"]
tm = tcnn.create_from_config(w["n_in"], w["n_out"], w["config"], seed=1337)
rng = tcnn._C.Pcg32(1337)
batches = bench.make_batches(w, bench.BATCH, 4, rng, device=torch.device("cuda", 0), tcnn=tcnn)
for i in range(30):
    tm.training_step(*batches[i % 4], want_context=False)
torch.cuda.synchronize()
buf = np.zeros((4096, 8), dtype=np.uint64)
assert tcnn._C._lib.tcnn_experiment_read_owner_stamps(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes)) == 0
live = buf[:, 5] > 0
s = buf[live].astype(np.int64)
rel = (s[:, :6] - s[:, 0].min()) * 0.01
lvl = s[:, 7]
print("workgroups", live.sum(), "span %.1f us" % rel[:, 5].max())
d = np.diff(rel, axis=1)
print("phase means: entry -> first tile loaded %.2f, tiles %s" % (d[:, 0].mean(), np.round(d[:, 1:].mean(axis=0), 2)))
for l in sorted(set(lvl.tolist())):
    m = lvl == l
    print("level %2d: n %4d entry mean %5.1f (min %5.1f max %5.1f)  per tile %s  life %5.2f  end max %5.1f" % (
        l, m.sum(), rel[m, 0].mean(), rel[m, 0].min(), rel[m, 0].max(), np.round(d[m, 1:].mean(axis=0), 2), (rel[m, 5] - rel[m, 0]).mean(), rel[m, 5].max()))
ts = np.arange(0, rel[:, 5].max(), 3.0)
print("resident at t:", "  ".join("%d:%d" % (t, ((rel[:, 0] <= t) & (rel[:, 5] > t)).sum()) for t in ts))
