"""Where do the extra ~0.3 ms of a 20-step measurement come from?  Per-step GPU time of the first steps after a synchronize."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import bench
import tinycudann as tcnn
w = bench.WORKLOADS["hash"]
tm = tcnn.create_from_config(w["n_in"], w["n_out"], w["config"], seed=1337)
n = 1 << 18
x = torch.rand((n, 3), device="cuda"); t = torch.rand((n, 4), device="cuda")
for _ in range(5): tm.training_step(x, t, want_context=False)
torch.cuda.synchronize()
for trial in range(3):
    time.sleep(0.2 * trial)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(30):
        h0 = time.perf_counter(); tm.training_step(x, t, want_context=False); host.append(time.perf_counter() - h0); ev[i + 1].record()
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(30)]
    print(f"trial {trial}: wall {wall * 1e3 / 30:.4f} ms/step; first-20 wall-equivalent {sum(gpu[:20]) / 20:.4f}; gpu per step:", " ".join(f"{g:.3f}" for g in gpu[:12]), "... host per call (us):", " ".join(f"{h * 1e6:.0f}" for h in host[:8]))
