"""Is a whole training step capturable into a HIP graph (torch.cuda.CUDAGraph) after warm-up, and what does replay cost at small batches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn
cfg = {"loss": {"otype": "RelativeL2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
       "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
       "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}
for n in (256, 4096, 1 << 18):
    tm = tcnn.create_from_config(3, 4, cfg, seed=1)
    ref = tcnn.create_from_config(3, 4, cfg, seed=1)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand((n, 3), generator=g, device="cuda"); t = torch.rand((n, 4), generator=g, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            tm.training_step(x, t, want_context=False)
    torch.cuda.synchronize()
    for _ in range(5): ref.training_step(x, t, want_context=False)
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, stream=s):
            tm.training_step(x, t, want_context=False)
    except Exception as e:
        print(n, "capture failed:", repr(e)[:300]); continue
    # the capture itself did not run the step: replay k times == k eager steps (the host-side step counter has to follow)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): graph.replay()
    torch.cuda.synchronize(); ms_graph = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50): ref.training_step(x, t, want_context=False)
    torch.cuda.synchronize(); ms_eager = (time.perf_counter() - t0) / 50 * 1e3
    print(f"n={n}: graph replay {ms_graph:.4f} ms/step, eager {ms_eager:.4f} ms/step")
