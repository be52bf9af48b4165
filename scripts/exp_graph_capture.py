"""training_step as one graph launch (tcnn_trainer_set_graph_capture) against the plain step, on a side stream, over batch sizes:
   python scripts/exp_graph_capture.py  ->  one line per batch size: plain ms, captured ms, graph launches / instantiations."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tiny-cuda-nn_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import tinycudann as tcnn
from tinycudann import native
import bench

w = bench.WORKLOADS["hash"]
side = torch.cuda.Stream()
for n in (256, 4096, 65536, 1 << 18):
    res = {}
    for mode in ("plain", "graph"):
        tm = native.create_from_config(w["n_in"], w["n_out"], w["config"], seed=1337)
        tm.set_graph_capture(mode == "graph")
        x = torch.rand(n, w["n_in"], device="cuda")
        t = torch.rand(n, w["n_out"], device="cuda")
        steps = 300 if n < 65536 else 100
        with torch.cuda.stream(side):
            for _ in range(20):
                tm.training_step(x, t, want_context=False)
            side.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tm.training_step(x, t, want_context=False)
            side.synchronize()
            res[mode] = (time.perf_counter() - t0) / steps * 1e3
        stats = tm.graph_capture_stats()
    print(f"batch {n:7d}: plain {res['plain']:.4f} ms  captured {res['graph']:.4f} ms  (graph launches {stats[0]}, instantiations {stats[1]})", flush=True)
