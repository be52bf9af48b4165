"""Where an owner workgroup's time goes: per-workgroup wall-clock stamps (100 MHz) written by an instrumented build of k_grid_bucket_owner
(stamps at entry / before and after the table-clear barrier / after the queue stream / after the second barrier / after conversion +
stores / after the sign-off).  The instrumentation is NOT in the tree; scripts/exp_owner_stamps.patch adds it to a working copy:
    git apply scripts/exp_owner_stamps.patch && bash scripts/build_variant_one.sh stamps grid_kernels "" && git apply -R scripts/exp_owner_stamps.patch
    TCNN_HIP_LIBRARY=$PWD/tiny-cuda-nn_amd/lib/variants/stamps.so python scripts/exp_owner_stamps.py
Results: profiles/r04_exp_notes.txt section 14."""
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
lib = tcnn._C._lib
buf = np.zeros((4096, 8), dtype=np.uint64)
rc = lib.tcnn_experiment_read_owner_stamps(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes))
assert rc == 0, rc
live = buf[:, 6] > 0
s = buf[live].astype(np.int64)
t0 = s[:, 0].min()
rel = (s[:, :7] - t0) * 0.01  # us
print("workgroups", live.sum(), "kernel span (first entry -> last sign-off) %.1f us" % rel[:, 6].max())
names = ["entry->clear issued+preload", "barrier 1", "queue stream", "bound + barrier 2", "convert + store", "sign-off"]
d = np.diff(rel, axis=1)
for k, n in enumerate(names):
    print("  %-28s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (n, d[:, k].mean(), *np.percentile(d[:, k], [10, 50, 90]), d[:, k].max()))
print("  whole workgroup              mean %6.2f  p50 %6.2f  p90 %6.2f us" % ((rel[:, 6] - rel[:, 0]).mean(), *np.percentile(rel[:, 6] - rel[:, 0], [50, 90])))
order = np.argsort(rel[:, 0])
starts = rel[order, 0]
print("entry times: first 8 %s ... #256 %.1f #512 %.1f #513 %.1f #768 %.1f last %.1f us" % (np.round(starts[:8], 1), starts[255], starts[511], starts[min(512, len(starts) - 1)], starts[min(767, len(starts) - 1)], starts[-1]))
first = rel[:, 0] < 5.0
print("workgroups entering in the first 5 us: %d; their stream phase mean %.2f us, the others' %.2f us; records per queue mean %.0f" % (first.sum(), d[first, 2].mean(), d[~first, 2].mean() if (~first).any() else 0.0, s[:, 7].mean()))
# how many workgroups are in their stream phase at time t
ts = np.arange(0, rel[:, 6].max(), 2.0)
conc = [(int(((rel[:, 2] <= t) & (rel[:, 3] > t)).sum()), int(((rel[:, 0] <= t) & (rel[:, 6] > t)).sum())) for t in ts]
print("t (us): streaming / resident workgroups")
print("  " + "  ".join("%d:%d/%d" % (t, a, b) for t, (a, b) in zip(ts, conc)))
# who is slow?  stream time by block-index range (items in plan order), by XCD (block % 8), and against the queue length
idx = np.nonzero(live)[0]
st = d[:, 2]
print("stream time by block range:", "  ".join("%d-%d: %.1f" % (a, a + 127, st[(idx >= a) & (idx < a + 128)].mean()) for a in range(0, 1024, 128)))
print("stream time by XCD (block %% 8):", np.round([st[idx % 8 == x].mean() for x in range(8)], 2))
print("first generation, stream time by XCD:", np.round([st[(idx % 8 == x) & (idx < 512)].mean() for x in range(8)], 2))
cnt = s[:, 7]
print("records per queue: min %d p10 %d p50 %d p90 %d max %d; correlation(stream time, records) %.2f" % (cnt.min(), *np.percentile(cnt, [10, 50, 90]), cnt.max(), np.corrcoef(st, cnt)[0, 1]))
slow = np.argsort(-st)[:12]
print("slowest:", [(int(idx[k]), round(float(st[k]), 1), int(cnt[k]), round(float(rel[k, 0]), 1)) for k in slow], "(block, stream us, records, entry us)")
life = rel[:, 6] - rel[:, 0]
print("lifetime: first generation mean %.1f max %.1f; second mean %.1f max %.1f" % (life[idx < 512].mean(), life[idx < 512].max(), life[idx >= 512].mean(), life[idx >= 512].max()))
