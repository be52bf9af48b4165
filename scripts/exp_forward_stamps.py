"""Where the gather's time goes, per XCD and level: k_grid_forward_tiles stamped per workgroup (100 MHz wall clock at entry and exit, level, XCD slot).
The instrumentation is NOT in the tree; scripts/exp_forward_stamps.patch adds it to a working copy:
    git apply scripts/exp_forward_stamps.patch && bash scripts/build_variant_one.sh stampfwd grid_kernels "" && git apply -R scripts/exp_forward_stamps.patch
    TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/stampfwd.so python scripts/exp_forward_stamps.py
Results: profiles/r04_exp_notes.txt section 18."""
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

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hash"]
tm = tcnn.create_from_config(w["n_in"], w["n_out"], w["config"], seed=1337)
rng = tcnn._C.Pcg32(1337)
batches = bench.make_batches(w, bench.BATCH, 4, rng, device=torch.device("cuda", 0), tcnn=tcnn)
for i in range(30):
    tm.training_step(*batches[i % 4], want_context=False)
torch.cuda.synchronize()
buf = np.zeros((32768, 4), dtype=np.uint64)
assert tcnn._C._lib.tcnn_experiment_read_owner_stamps(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes)) == 0
live = buf[:, 1] > 0
s = buf[live].astype(np.int64)
idx = np.nonzero(live)[0]
t0 = s[:, 0].min()
st, en, lvl = (s[:, 0] - t0) * 0.01, (s[:, 1] - t0) * 0.01, s[:, 2]
print("workgroups", live.sum(), "span %.1f us" % en.max())
for x in range(8):
    m = (idx % 8) == x
    print("XCD slot %d: %4d workgroups, levels %s, last exit %.1f us, mean life %.2f us" % (x, m.sum(), sorted(set(lvl[m].tolist())), en[m].max(), (en[m] - st[m]).mean()))
for l in sorted(set(lvl.tolist())):
    m = lvl == l
    print("level %2d: n %4d  life mean %.2f  p90 %.2f   entries %.1f .. %.1f  last exit %.1f" % (l, m.sum(), (en[m] - st[m]).mean(), np.percentile(en[m] - st[m], 90), st[m].min(), st[m].max(), en[m].max()))
ts = np.arange(0, en.max(), 5.0)
print("resident at t:", "  ".join("%d:%d" % (t, ((st <= t) & (en > t)).sum()) for t in ts))
